# priority swap between the two matrix waves of a SIMD inside a step (fp4 variants), interleaved on one box
O=gpurun_out/r3x; mkdir -p $O; rm -f $O/*.log
LIBS="libwsu_base.so libwsu_ps2.so libwsu_ps3.so"
for r in 1 2 3; do for l in $LIBS; do
  timeout -k 10 200 python tools/probe_units_pl.py --q4 $l > $O/${l}_$r.log 2>&1 || { tail -3 $O/${l}_$r.log; exit 1; }
done; done
python - <<'P'
import re,glob,collections
t=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r3x/*.log')):
    lib=f.split('/')[-1].rsplit('_',1)[0]
    for m in re.finditer(r'cin=(\d+) cout=(\d+) hw=(\d+) concat=(\d+) pool=(\d): (\d+) us', open(f).read()):
        t[lib]['%s>%s@%s'%(m.group(1),m.group(2),m.group(3))].append(int(m.group(6)))
for lib,d in t.items():
    print(lib, {k: min(v) for k,v in d.items()}, 'sum', sum(min(v) for v in d.values()))
P
