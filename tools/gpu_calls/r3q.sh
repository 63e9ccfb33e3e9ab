# e4m3 product vs the same with per-step recomputed tap-pair offsets (libwsu_plopq.so), and the whole net in both modes
O=gpurun_out/r3q; mkdir -p $O
for r in 1 2; do
  timeout -k 10 200 python tools/probe_units_pl.py > $O/product_$r.log 2>&1 || exit 1
  timeout -k 10 200 python tools/probe_units_pl.py libwsu_plopq.so > $O/opq_$r.log 2>&1 || exit 1
done
for f in product_1 opq_1 product_2 opq_2; do echo "== $f"; grep -o "cin=.*us" $O/$f.log | tr '\n' ';'; echo; done
bash tools/gpu_calls/r3n.sh || exit 1
bash tools/gpu_calls/r3o.sh
