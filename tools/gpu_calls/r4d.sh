O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "ragged" > $O/pytest.log 2>&1; rc=$?; tail -8 $O/pytest.log | cut -c1-250; exit $rc
