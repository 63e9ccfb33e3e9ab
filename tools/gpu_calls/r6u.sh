# experiment: accumulators initialised with the bias (libwsu_qbiasinit.so = -DWSU_Q_BIAS_INIT=1) against the product build: parity of the conv tests, layer probe, forward A/B
O=gpurun_out/r6u; mkdir -p $O
WSU_LIB=$PWD/ws_unet_amd/libwsu_qbiasinit.so timeout -k 10 600 python -m pytest tests/test_gpu_q.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
timeout -k 10 300 python tools/probe_q_layer.py 2>&1 | grep -v amdgpu >> $O/probe.log || exit 1
WSU_LIB=$PWD/ws_unet_amd/libwsu_qbiasinit.so timeout -k 10 300 python tools/probe_q_layer.py 2>&1 | grep -v amdgpu >> $O/probe.log || exit 1
done
cat $O/probe.log | cut -c1-120
B="--no-cpu-baseline --no-train-step --no-latency --no-trained-mae --no-other-modes"
for i in 1 2; do
timeout -k 10 300 python bench.py $B > $O/bench_base_$i.log 2>&1 || exit 1
WSU_LIB=$PWD/ws_unet_amd/libwsu_qbiasinit.so timeout -k 10 300 python bench.py $B > $O/bench_biasinit_$i.log 2>&1 || exit 1
done
for f in base_1 biasinit_1 base_2 biasinit_2; do echo -n "$f: "; tail -1 $O/bench_$f.log | cut -c1-130; done
