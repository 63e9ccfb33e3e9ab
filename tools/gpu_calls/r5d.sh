# round 4, call d: Q-format tests, round-4 pinning tests, shards test; decode micro-benchmark and the evaluate loop's host budget on the GPU box
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_q.py tests/test_gpu_round4.py tests/test_gpu_evaluate.py tests/test_gpu_dp.py tests/test_gpu_planar.py -x -q -s 2>&1 | grep -v "^$" | tail -30 | tee $O/pytest.log
python tools/bench_png_decode.py 128 2>&1 | tee $O/png_decode.log
timeout -k 10 400 python tools/bench_evaluate.py --images 512 2>/dev/null | grep "^{" | tee $O/evaluate_loop.json.log
