# per-image evaluate API with the next file decoded one row ahead: evaluate / sharded tests, then the rates (256 files)
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_dp.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/bench_evaluate.py --images 256 --batch 32 2>/dev/null | grep "^{" | tee $O/evaluate_256.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['batched_images_per_s'],1), 'img/s batched;', round(d['per_image_api_images_per_s'],1), 'per-image API; max |beta diff|', d['max_abs_beta_diff_batched_vs_per_image'])"
