# round-3 GPU call D: full suite (no-contract loss kernels, lean convT loaders); bench; unit probes on the planar kernel; SQ counters of the new build
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.log 2>&1 || { echo bench failed; tail -5 $O/bench.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3d/bench.log').read().strip().split('\n')[-1])
print('img/s',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4),'per_layer',d['roofline']['per_layer']['frac'])
print(' '.join(f"{r['layer']}:{r['ms']}" for r in d['roofline']['per_layer']['layers']))
print('train', round(d['train_step']['ms_per_step'],2), d['train_step']['kernels_ms_per_step'])
print({k:round(v['images_per_s']) for k,v in d['other_modes'].items()})
P
for lib in libwsu.so libwsu_plprobe2.so libwsu_plprobe4.so libwsu_plprobe3.so; do timeout -k 10 120 python tools/probe_units_pl.py $lib >> $O/units_pl.log 2>&1 || exit 1; done
timeout -k 10 120 python tools/probe_units_pl.py libwsu.so --xres0 >> $O/units_pl.log 2>&1
grep -v amdgpu.ids $O/units_pl.log
tools/profile_sq.sh lean > $O/sq.log 2>&1 || { echo "sq failed"; tail -5 $O/sq.log; }
head -20 gpurun_out/sq_lean/summary.md
