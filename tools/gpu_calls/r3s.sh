# epilogue overlap (EPO) of the fp4 variant: parity tests, then product vs the same tree built with -DWSU_PL_EPO=0 (libwsu_noepo.so), interleaved
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_planar.py -x -q -m gpu -k "f16f4p or q4" > $O/pytest_f4.log 2>&1; rc=$?; tail -6 $O/pytest_f4.log | cut -c1-250; [ $rc -eq 0 ] || exit 1
for r in 1 2; do
  timeout -k 10 200 python tools/probe_units_pl.py --q4 > $O/epo_$r.log 2>&1 || exit 1
  timeout -k 10 200 python tools/probe_units_pl.py --q4 libwsu_noepo.so > $O/noepo_$r.log 2>&1 || exit 1
done
for f in epo_1 noepo_1 epo_2 noepo_2; do echo "== $f"; grep -o "cin=.*us" $O/$f.log | tr '\n' ';'; echo; done
for lib in libwsu.so libwsu_noepo.so libwsu.so libwsu_noepo.so; do
  WSU_LIB=$PWD/ws_unet_amd/$lib timeout -k 10 200 python bench.py --no-other-modes --no-cpu-baseline --no-train-step --no-latency --steps 20 --warmup 5 > $O/bench_$lib.log 2>&1 || { tail -3 $O/bench_$lib.log; exit 1; }
  python -c "
import json,sys
d=json.loads(open('$O/bench_$lib.log').read().strip().split('\n')[-1]); print('$lib', round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'frac', round(d['roofline']['frac'],4), {r['layer']: r['ms'] for r in d['roofline']['per_layer']['layers']})"
done
