# whole-net A/B of the epilogue-overlap builds (bench.py, default mode, interleaved on one box)
O=gpurun_out/r3v; mkdir -p $O
for r in 1 2; do for lib in libwsu_noepo.so libwsu_noepo_w.so libwsu_epo_w.so libwsu_epo_nfw.so libwsu_epo_w_nohead.so; do
  WSU_LIB=$PWD/ws_unet_amd/$lib timeout -k 10 200 python bench.py --no-other-modes --no-cpu-baseline --no-train-step --no-latency --steps 20 --warmup 5 > $O/bench_${lib}_$r.log 2>&1 || { tail -3 $O/bench_${lib}_$r.log; exit 1; }
  python -c "
import json,sys
d=json.loads(open('$O/bench_${lib}_$r.log').read().strip().split('\n')[-1]); print('$lib', round(d['value'],1), 'img/s', 'frac', round(d['roofline']['frac'],4), {r['layer']: round(r['ms'],3) for r in d['roofline']['per_layer']['layers'] if r['layer'] in ('e12','e22','e32','d31','d41','d42+outconv')})"
done; done
