# e11 + e12 + pool in one launch (conv3x3_q kernel variant F1): parity, whole-net tests, forward A/B on one box
O=gpurun_out/r6e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_q.py tests/test_gpu_qu.py -x -q > $O/pytest_q.log 2>&1 || { tail -40 $O/pytest_q.log; exit 1; }
tail -2 $O/pytest_q.log
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round4.py tests/test_gpu_evaluate.py -x -q > $O/pytest_net.log 2>&1 || { tail -40 $O/pytest_net.log; exit 1; }
tail -2 $O/pytest_net.log
B="--no-cpu-baseline --no-train-step --no-latency --no-trained-mae --no-other-modes"
WSU_FUSE_FIRST_Q=0 timeout -k 10 300 python bench.py $B > $O/bench_separate.log 2>&1 && timeout -k 10 300 python bench.py $B --detail $O/bench_fused_first_detail.json > $O/bench_fused_first.log 2>&1 && WSU_FUSE_FIRST_Q=0 timeout -k 10 300 python bench.py $B --detail $O/bench_separate_detail.json > $O/bench_separate_2.log 2>&1 && timeout -k 10 300 python bench.py $B > $O/bench_fused_first_2.log 2>&1
for f in separate fused_first separate_2 fused_first_2; do tail -1 $O/bench_$f.log | cut -c1-170; done
