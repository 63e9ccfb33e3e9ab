O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 200 python tools/probe_units_pl.py > $O/product.log 2>&1 || exit 1
timeout -k 10 200 python tools/probe_units_pl.py --q4 > $O/q4.log 2>&1 || exit 1
WSU_PL_ABLATE=8 timeout -k 10 200 python tools/probe_units_pl.py --q4 > $O/q4_noderive.log 2>&1 || exit 1
WSU_PL_ABLATE=2 timeout -k 10 200 python tools/probe_units_pl.py --q4 > $O/q4_noepi.log 2>&1 || exit 1
WSU_PL_ABLATE=2 timeout -k 10 200 python tools/probe_units_pl.py > $O/product_noepi.log 2>&1 || exit 1
for f in product q4 q4_noderive q4_noepi product_noepi; do echo "== $f"; grep -o "cin=.*us" $O/$f.log | tr '\n' ';'; echo; done
