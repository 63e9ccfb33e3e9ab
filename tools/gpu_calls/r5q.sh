# round 4, call q: the evaluate driver with --u8-shards (CLI test) and the evaluate tests
O=gpurun_out/r5q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_dp.py -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest.log
