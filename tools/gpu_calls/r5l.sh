# round 4, call l: first_pl with hand-packed FMAs and a single-plane specialisation
O=gpurun_out/r5l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_planar.py tests/test_gpu_q.py tests/test_gpu_forward.py tests/test_gpu_round4.py tests/test_gpu_planar_train.py -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest.log || exit 1
B="--no-other-modes --no-latency --no-trained-mae --steps 20 --warmup 5"
for i in 1 2; do timeout -k 10 400 python bench.py $B --detail $O/detail_$i.json 2>/dev/null | grep "^{" > $O/bench_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5l/detail_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), d['kernel_ms_per_step'], 'train', d['train_step']['ms_per_step'], d['train_step']['kernels_ms_per_step'].get('conv3x3_first_pl'), d['train_step']['kernels_ms_per_step'].get('convt2x2_pl'))
PY
