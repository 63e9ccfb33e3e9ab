# round-3 GPU call C: hazard four-way A/B on the failing round-2 build + probe; full suite on the all-no-SLP build; bench; unit probes on the planar kernel
O=gpurun_out/r3c; mkdir -p $O
( cd repro_r02 && for v in slp slpnop_w slpnop_a slpnop; do WSU_LIB=ws_unet_amd/libwsu_$v.so timeout -k 10 200 python tools/stress_pl.py 200 > ../$O/r02_stress_$v.log 2>&1; echo "r02 tree $v: $(grep head=True ../$O/r02_stress_$v.log)"; done )
timeout -k 10 200 ./tools/pk_hazard_probe 4000 10 > $O/probe.log 2>&1; cat $O/probe.log
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.log 2>&1 || { echo bench failed; tail -5 $O/bench.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3c/bench.log').read().strip().split('\n')[-1])
print('img/s',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4),'per_layer',d['roofline']['per_layer']['frac'])
print(' '.join(f"{r['layer']}:{r['ms']}" for r in d['roofline']['per_layer']['layers']))
print('train', round(d['train_step']['ms_per_step'],2), d['train_step']['kernels_ms_per_step'])
print({k:round(v['images_per_s']) for k,v in d['other_modes'].items()})
P
for lib in libwsu.so libwsu_plprobe2.so libwsu_plprobe4.so libwsu_plprobe3.so; do timeout -k 10 120 python tools/probe_units_pl.py $lib >> $O/units_pl.log 2>&1 || exit 1; done
timeout -k 10 120 python tools/probe_units_pl.py libwsu.so --xres0 >> $O/units_pl.log 2>&1
cat $O/units_pl.log | grep -v amdgpu.ids
