# batched iterator builds its rows from records, per chunk: evaluate tests, then the loop's throughput at 256 / 2048 images
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_dp.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
for cfg in "256 32" "2048 32"; do set -- $cfg
  timeout -k 10 400 python tools/bench_evaluate.py --images $1 --batch $2 2>/dev/null | grep "^{" | tee $O/evaluate_$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('images $1 batch $2:', round(d['batched_images_per_s'],1), 'img/s batched;', round(d['per_image_api_images_per_s'],1), 'per-image API')" | tee -a $O/evaluate_sizes.log
done
