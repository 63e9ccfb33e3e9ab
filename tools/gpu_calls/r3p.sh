O=gpurun_out/r3p; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; rc=$?; tail -4 $O/smoke.log; exit $rc
