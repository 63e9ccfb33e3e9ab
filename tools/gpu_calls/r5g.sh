O=gpurun_out/r5g; mkdir -p $O
timeout -k 10 400 python tools/bench_evaluate.py --images 512 2>/dev/null | grep "^{" | tee $O/evaluate_loop.json.log
