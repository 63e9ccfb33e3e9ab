O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planar_train.py -x -q -m gpu > $O/pytest_planar_train.log 2>&1; rc=$?; tail -4 $O/pytest_planar_train.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; rc=$?; tail -4 $O/smoke.log; exit $rc
