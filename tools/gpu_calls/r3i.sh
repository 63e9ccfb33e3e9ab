# f16 products in the backward matrix kernels: parity tests, whole-net gradient bands, same-box A/B of the train step
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_planar_train.py -x -q -m gpu > $O/pytest_planar_train.log 2>&1; rc=$?; tail -5 $O/pytest_planar_train.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_products.py 64 512 3 > $O/ab_products.log 2>&1; rc=$?; cat $O/ab_products.log | tail -12; [ $rc -eq 0 ] || exit 1

