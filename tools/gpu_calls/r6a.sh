# per-image API with rows riding along (micro-batches), race screen of the fused decoder entry, evaluate-loop bench
O=gpurun_out/r6a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_qu.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python tools/stress_qu.py 60 > $O/stress_qu.log 2>&1 || { tail -5 $O/stress_qu.log; exit 1; }
grep -v amdgpu $O/stress_qu.log
timeout -k 10 400 python tools/bench_evaluate.py --images 512 > $O/evaluate_loop.log 2>&1 || { tail -5 $O/evaluate_loop.log; exit 1; }
tail -1 $O/evaluate_loop.log | cut -c1-900
