# round 4, call c: the whole GPU suite on the new default path (planar Q storage, conv3x3_q) + the round-4 pinning tests
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest_gpu.log
