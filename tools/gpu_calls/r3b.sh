# round-3 GPU call B: lean loaders -- parity of the planar kernels, then timing; the r02 hazard repro rides along
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planar.py tests/test_gpu_planar_train.py tests/test_gpu_forward.py -q -x > $O/pytest_pl.log 2>&1; rc=$?; echo "pytest planar rc=$rc"; tail -4 $O/pytest_pl.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/stress_pl.py 100 > $O/stress.log 2>&1 || { echo stress failed; tail -3 $O/stress.log; exit 1; }
tail -1 $O/stress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.log 2>&1 || { echo bench failed; tail -5 $O/bench.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3b/bench.log').read().strip().split('\n')[-1])
print('img/s',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4),'per_layer',d['roofline']['per_layer']['frac'])
print(' '.join(f"{r['layer']}:{r['ms']}" for r in d['roofline']['per_layer']['layers']))
print('train', round(d['train_step']['ms_per_step'],2), d['train_step']['kernels_ms_per_step'])
P
timeout -k 10 120 ./tools/pk_hazard_probe 4000 10 > $O/probe.log 2>&1; cat $O/probe.log
cd repro_r02 && for v in slp slpnop8 slpnop; do WSU_LIB=ws_unet_amd/libwsu_$v.so timeout -k 10 200 python tools/stress_pl.py 200 > ../$O/r02_stress_$v.log 2>&1; echo "r02 tree $v: $(tail -1 ../$O/r02_stress_$v.log)"; done
