# round 4, call f: planar input gradient, multi-plane default-mode nets, hipGraph replay of the per-image API; evaluate loop bench
O=gpurun_out/r5f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_planar_train.py tests/test_gpu_round4.py tests/test_gpu_evaluate.py tests/test_gpu_backward.py tests/test_gpu_q.py -x -q -s 2>&1 | grep -v "^$" | tail -30 | tee $O/pytest.log
timeout -k 10 400 python tools/bench_evaluate.py --images 512 2>/dev/null | grep "^{" | tee $O/evaluate_loop.json.log
