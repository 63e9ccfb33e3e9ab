O=gpurun_out/r3p; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log | cut -c1-200; exit $rc
