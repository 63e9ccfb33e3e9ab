# round 4, call i: the transposed conv and the first layer with compile-time output formats (the run-time switch spilled 33 registers in convt2x2_pl)
O=gpurun_out/r5i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_q.py tests/test_gpu_planar.py tests/test_gpu_forward.py -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest.log || exit 1
B="--no-other-modes --no-train-step --no-latency --no-trained-mae --steps 20 --warmup 5"
for i in 1 2; do timeout -k 10 300 python bench.py $B --detail $O/detail_$i.json 2>/dev/null | grep "^{" > $O/bench_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5i/detail_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), d['kernel_ms_per_step'])
    print('   ', [(x['layer'], round(x['ms'],3)) for x in r['per_layer']['layers']])
PY
