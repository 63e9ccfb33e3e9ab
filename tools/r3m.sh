# best-case fp4-configuration probe of the forward conv against the product, interleaved on one box
O=gpurun_out/r3m; mkdir -p $O
for r in 1 2; do
  timeout -k 10 200 python tools/probe_units_pl.py > $O/product_$r.log 2>&1 || { tail -3 $O/product_$r.log; exit 1; }
  timeout -k 10 200 python tools/probe_units_pl.py libwsu_plprobe6.so > $O/probe6_$r.log 2>&1 || { tail -3 $O/probe6_$r.log; exit 1; }
  timeout -k 10 200 python tools/probe_units_pl.py libwsu_plprobe2.so > $O/probe2_$r.log 2>&1 || { tail -3 $O/probe2_$r.log; exit 1; }
done
paste -d'|' $O/product_2.log $O/probe6_2.log | cut -c1-200
