# fused decoder entry: which DMA kind costs the time (0x100 IN_S, 0x200 W_S, 0x400 IN_L, 0x800 W_L not fetched; 32: skip inputs without the class gather)
O=gpurun_out/r5w; mkdir -p $O
for ab in 0 32 256 512 1024 2048 768 3072; do
WSU_QU_ABLATE=$ab timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
done
grep -v amdgpu.ids $O/probe.log
