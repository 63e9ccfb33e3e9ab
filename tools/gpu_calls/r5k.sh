# round 4, call k: race screen of conv3x3_q (300 launches per shape, results compared bitwise), both organisations; full GPU suite
O=gpurun_out/r5k; mkdir -p $O
timeout -k 10 500 python tools/stress_pl.py 300 --q4 2>&1 | tail -8 | tee $O/stress_q_300_launches.log
WSU_Q_ROWS=4 timeout -k 10 500 python tools/stress_pl.py 100 --q4 2>&1 | tail -8 | tee $O/stress_q_rows4_100_launches.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest_gpu.log
