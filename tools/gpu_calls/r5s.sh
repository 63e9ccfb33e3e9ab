# fused upsample + concat + conv (conv3x3_qu.hip): its own tests first, then the whole-net tests of the default mode, then a same-box A/B of the forward
set -e
mkdir -p gpurun_out/r5s
timeout -k 10 300 python -m pytest tests/test_gpu_qu.py -x -q > gpurun_out/r5s/pytest_qu.log 2>&1 || { tail -30 gpurun_out/r5s/pytest_qu.log; exit 1; }
tail -3 gpurun_out/r5s/pytest_qu.log
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round4.py tests/test_gpu_evaluate.py -x -q > gpurun_out/r5s/pytest_net.log 2>&1 || { tail -30 gpurun_out/r5s/pytest_net.log; exit 1; }
tail -3 gpurun_out/r5s/pytest_net.log
B="--no-cpu-baseline --no-train-step --no-latency --no-trained-mae --no-other-modes"
WSU_FUSE_UP=0 timeout -k 10 300 python bench.py $B --detail gpurun_out/r5s/bench_two_kernel_detail.json > gpurun_out/r5s/bench_two_kernel.log 2>&1
timeout -k 10 300 python bench.py $B --detail gpurun_out/r5s/bench_fused_detail.json > gpurun_out/r5s/bench_fused.log 2>&1
WSU_FUSE_UP=0 timeout -k 10 300 python bench.py $B > gpurun_out/r5s/bench_two_kernel_2.log 2>&1
timeout -k 10 300 python bench.py $B > gpurun_out/r5s/bench_fused_2.log 2>&1
for f in two_kernel fused two_kernel_2 fused_2; do tail -1 gpurun_out/r5s/bench_$f.log | cut -c1-400; done
