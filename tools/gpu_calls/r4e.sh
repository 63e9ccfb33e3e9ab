# evaluate loop from PNG files: 256 / 1024 / 2048 images (how much of the 256-image figure is start-up), batch 32 and 64
O=gpurun_out/r4e; mkdir -p $O
for cfg in "256 32" "1024 32" "2048 32" "2048 64" "2048 128"; do set -- $cfg
  timeout -k 10 400 python tools/bench_evaluate.py --images $1 --batch $2 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('images $1 batch $2:', round(d['batched_images_per_s'],1), 'img/s batched;', round(d['per_image_api_images_per_s'],1), 'per-image API; decode', round(d['png_decode_ms_per_image_1thread'],2), 'ms/img/thread')" | tee -a $O/evaluate_sizes.log
done
nproc
