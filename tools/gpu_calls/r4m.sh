# one-off: 300 random shapes of the fp4 variant against its CPU emulation (the test draws 12 by default)
O=gpurun_out/r4m; mkdir -p $O
WSU_TEST_SWEEP=300 timeout -k 10 1000 python -m pytest tests/test_gpu_planar.py -x -q -m gpu -k "q4_random_shapes" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-300; exit $rc
