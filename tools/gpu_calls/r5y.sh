# fused decoder entry with the low chunks first and the last (skip) step split over the epilogue: tests, then the layer probe and the forward A/B
O=gpurun_out/r5y; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_qu.py -x -q > $O/pytest_qu.log 2>&1 || { tail -30 $O/pytest_qu.log; exit 1; }
tail -2 $O/pytest_qu.log
timeout -k 10 200 python tools/probe_qu_layer.py > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=2 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
grep -v amdgpu.ids $O/probe.log
B="--no-cpu-baseline --no-train-step --no-latency --no-trained-mae --no-other-modes"
WSU_FUSE_UP=0 timeout -k 10 300 python bench.py $B > $O/bench_two_kernel.log 2>&1 && timeout -k 10 300 python bench.py $B --detail $O/bench_fused_detail.json > $O/bench_fused.log 2>&1
for f in two_kernel fused; do tail -1 $O/bench_$f.log | cut -c1-200; done
