# round 4, call j: the explicit fragment pipeline on the 8-wave organisation (depth 1 and 2) and depth 2 on the 4-wave one, interleaved with the product build
O=gpurun_out/r5j; mkdir -p $O
B="--no-other-modes --no-train-step --no-latency --no-trained-mae --steps 20 --warmup 5"
for i in 1 2; do
  timeout -k 10 300 python bench.py $B --detail $O/base_$i.json > /dev/null 2>&1
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qpipe2_d1.so timeout -k 10 300 python bench.py $B --detail $O/pipe2d1_$i.json > /dev/null 2>&1
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qpipe2_d2.so timeout -k 10 300 python bench.py $B --detail $O/pipe2d2_$i.json > /dev/null 2>&1
  WSU_Q_ROWS=4 WSU_LIB=$PWD/ws_unet_amd/libwsu_qpipe2_d2.so timeout -k 10 300 python bench.py $B --detail $O/rows4d2_$i.json > /dev/null 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5j/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'conv frac', round(r['frac'],4), d['kernel_ms_per_step'])
    print('   ', [(x['layer'], round(x['ms'],3)) for x in r['per_layer']['layers']])
PY
