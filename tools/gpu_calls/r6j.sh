# fused decoder entry: the loaders' weight pieces paced with s_sleep (64: 64 cycles, 128: 192 cycles behind every piece) against the burst, one box
O=gpurun_out/r6j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_qu.py -x -q -k "emulation" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for ab in 0 64 128 192 0 64; do
WSU_QU_ABLATE=$ab timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
done
grep -v amdgpu.ids $O/probe.log
