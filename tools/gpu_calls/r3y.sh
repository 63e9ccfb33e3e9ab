# fragment loads one unit ahead (libwsu_pipe.so = -DWSU_PL_PIPE=1) against the product: parity of the variant, then per-layer times interleaved on one box
O=gpurun_out/r3y; mkdir -p $O; rm -f $O/*.log
WSU_LIB=$PWD/ws_unet_amd/libwsu_pipe.so timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_planar.py -x -q -m gpu -k "f16f4p or q4" > $O/pytest_f4.log 2>&1; rc=$?; tail -3 $O/pytest_f4.log | cut -c1-250; [ $rc -eq 0 ] || exit 1
LIBS="libwsu.so libwsu_pipe.so"
for r in 1 2 3; do for l in $LIBS; do
  timeout -k 10 200 python tools/probe_units_pl.py --q4 $l > $O/${l}_$r.log 2>&1 || { tail -3 $O/${l}_$r.log; exit 1; }
done; done
python - <<'P'
import re,glob,collections
t=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r3y/libwsu*.log')):
    lib=f.split('/')[-1].rsplit('_',1)[0]
    for m in re.finditer(r'cin=(\d+) cout=(\d+) hw=(\d+) concat=(\d+) pool=(\d): (\d+) us', open(f).read()):
        t[lib]['%s>%s@%s'%(m.group(1),m.group(2),m.group(3))].append(int(m.group(6)))
for lib,d in t.items():
    print(lib, {k: min(v) for k,v in d.items()}, 'sum', sum(min(v) for v in d.values()))
P
