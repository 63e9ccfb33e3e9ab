O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 300 python tools/bench_ws_attack.py 2>/dev/null | grep "^{" | tee $O/ws_attack.log
timeout -k 10 300 python tools/bench_ws_attack.py --correct-bias 2>/dev/null | grep "^{" | tee -a $O/ws_attack.log
