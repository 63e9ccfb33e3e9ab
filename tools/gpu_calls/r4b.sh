# sum kernels of the training step (two pixels per thread; head backward without the residual plane in f16 products): parity, then the step
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planar_train.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-250; [ $rc -eq 0 ] || exit 1
WSU_TIME_TRAIN_LAUNCHES=1 timeout -k 10 300 python tools/time_train.py f16f8p 64 512 > $O/train.log 2>&1 || { tail -5 $O/train.log; exit 1; }
grep -i "chansum\|head_bwd\|pool_bwd\|first_pl_bwd\|colsum" $O/train.log | head; tail -c 400 $O/train.log
