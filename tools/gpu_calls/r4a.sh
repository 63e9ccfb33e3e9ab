# is conv3x3_pl power-managed?  The product kernel on random data and on all-zero data (same instructions, no switching), interleaved
O=gpurun_out/r4a; mkdir -p $O; rm -f $O/*.log
for r in 1 2; do
  timeout -k 10 200 python tools/probe_units_pl.py --q4 > $O/data_$r.log 2>&1 || exit 1
  timeout -k 10 200 python tools/probe_units_pl.py --q4 --zeros > $O/zeros_$r.log 2>&1 || exit 1
done
timeout -k 10 200 python tools/probe_units_pl.py > $O/e4m3_data.log 2>&1 || exit 1
timeout -k 10 200 python tools/probe_units_pl.py --zeros > $O/e4m3_zeros.log 2>&1 || exit 1
for f in data_1 zeros_1 data_2 zeros_2 e4m3_data e4m3_zeros; do echo "== $f"; grep -o "cin=.*us" $O/$f.log | tr '\n' ';'; echo; done
