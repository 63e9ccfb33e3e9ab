# fused decoder entry: is the low phase bound by the inputs' latency (8: cached low inputs) or by DMA bytes (16: half of the low weights)?
O=gpurun_out/r5v; mkdir -p $O
for ab in 0 8 16 24; do
WSU_QU_ABLATE=$ab timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
done
grep -v amdgpu.ids $O/probe.log
