O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_planar.py -x -q -m gpu -k q4 > $O/pytest_q4.log 2>&1; rc=$?; tail -15 $O/pytest_q4.log | cut -c1-250; exit $rc
