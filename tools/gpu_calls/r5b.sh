# round 4, call b: Q-format tests after the fix; A/B: round-3 organisation / 8 matrix waves / 4 matrix waves with the explicit fragment pipeline; 16x16-shape timing probes
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_q.py -x -q 2>&1 | tail -25 | tee $O/pytest_q.log
B="--no-other-modes --no-train-step --no-latency --steps 20 --warmup 5"
for i in 1 2; do
  WSU_Q4_R3=1 timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_r3_$i.json
  timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_q4_$i.json
  WSU_Q_ROWS=2 timeout -k 10 300 python bench.py $B 2>/dev/null | grep "^{" > $O/bench_q2_$i.json
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qprobe16.so WSU_Q_ROWS=2 timeout -k 10 300 python bench.py $B --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_p16q2_$i.json
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qprobe16.so timeout -k 10 300 python bench.py $B --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_p16q4_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5b/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), 'avg ms', round(r['avg_launch_ms'],4), d['kernel_ms_per_step'])
    print('   ', [(x['layer'], round(x['ms'],3)) for x in r['per_layer']['layers']])
PY
