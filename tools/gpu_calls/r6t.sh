# the default inference path at 1024x1024, non-square and 2048x1536 against the CPU oracle
O=gpurun_out/r6t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_qu.py -x -q -k "pair_training_resolution" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
