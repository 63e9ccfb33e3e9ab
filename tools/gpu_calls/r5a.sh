# round 4, call a: the new Q-format conv (csrc/conv3x3_q.hip) -- kernel tests, whole-net tests, same-box A/B against round 3's organisation
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_q.py -x -q 2>&1 | tail -25 | tee $O/pytest_q.log
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_planar.py tests/test_gpu_evaluate.py -x -q 2>&1 | tail -25 | tee $O/pytest_fwd.log
B="--no-other-modes --no-train-step --no-latency --steps 20 --warmup 5"
for i in 1 2; do
  WSU_Q4_R3=1 timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_r3_$i.json
  timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_q4_$i.json
  WSU_Q_ROWS=2 timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_q2_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5a/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), 'avg ms', round(r['avg_launch_ms'],4), d['kernel_ms_per_step'])
    print('   ', [(x['layer'], round(x['ms'],3)) for x in r['per_layer']['layers']])
PY
