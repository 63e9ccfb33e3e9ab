# evaluate loop: decode threads 8 / 16 / 32 (is the 2840 images/s steady state decode-bound?)
O=gpurun_out/r4f; mkdir -p $O
for th in 8 16 32 64; do
  WSU_IO_THREADS=$th timeout -k 10 400 python tools/bench_evaluate.py --images 2048 --batch 32 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $th:', round(d['batched_images_per_s'],1), 'img/s batched')" | tee -a $O/evaluate_threads.log
done
