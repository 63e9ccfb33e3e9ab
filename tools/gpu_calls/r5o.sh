# round 4, call o: the ninth tap's cross terms paired across two chunks -- tests, race screen, bench against a build without the pairing
O=gpurun_out/r5o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_q.py tests/test_gpu_forward.py tests/test_gpu_evaluate.py tests/test_gpu_round4.py -x -q 2>&1 | grep -v "^$" | tail -8 | tee $O/pytest.log || exit 1
timeout -k 10 400 python tools/stress_pl.py 100 --q4 2>&1 | tail -3 | tee $O/stress.log
B="--no-other-modes --no-train-step --no-latency --no-trained-mae --steps 20 --warmup 5"
for i in 1 2; do
  timeout -k 10 300 python bench.py $B --detail $O/pair_$i.json > /dev/null 2>&1
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qnopair.so timeout -k 10 300 python bench.py $B --detail $O/nopair_$i.json > /dev/null 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5o/*pair_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), d['kernel_ms_per_step'])
    print('   ', [(x['layer'], round(x['ms'],3)) for x in r['per_layer']['layers']])
PY
