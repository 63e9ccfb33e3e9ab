# per-image API: rows riding along + one launch queued ahead (pipelined micro-batches): tests, evaluate-loop bench
O=gpurun_out/r6b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python tools/bench_evaluate.py --images 512 > $O/evaluate_loop.log 2>&1 || { tail -5 $O/evaluate_loop.log; exit 1; }
tail -1 $O/evaluate_loop.log | cut -c1-700
