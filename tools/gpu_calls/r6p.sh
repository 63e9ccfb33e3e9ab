# evaluate-loop bench on the final host code (per-image API with 16 images per launch, two launches queued ahead)
O=gpurun_out/r6p; mkdir -p $O
timeout -k 10 400 python tools/bench_evaluate.py --images 1024 > $O/evaluate_loop.log 2>&1 || { tail -5 $O/evaluate_loop.log; exit 1; }
tail -1 $O/evaluate_loop.log | cut -c1-900
