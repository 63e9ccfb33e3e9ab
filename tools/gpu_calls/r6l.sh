# after the last host-side clean-ups: evaluate tests, the fused entry's tests (incl. the four-tiles-per-workgroup case), the 1024x1024 default-mode test
O=gpurun_out/r6l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_qu.py tests/test_gpu_q.py tests/test_gpu_dp.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
