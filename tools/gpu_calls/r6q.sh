# uploads on a side stream (batched chunks and per-image launches): evaluate tests, two-rank tests, evaluate-loop bench
O=gpurun_out/r6q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_dp.py tests/test_gpu_ws_attack.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 400 python tools/bench_evaluate.py --images 1024 > $O/evaluate_loop.log 2>&1 || { tail -5 $O/evaluate_loop.log; exit 1; }
tail -1 $O/evaluate_loop.log | cut -c1-420
timeout -k 10 400 python tools/profile_per_image.py --images 768 2>&1 | grep "per-image API"
