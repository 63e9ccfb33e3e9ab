# round-3 GPU call A: packed-f32 hazard evidence, the full GPU suite, SQ counters of the product binaries, baseline bench
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 120 ./tools/pk_hazard_probe 4000 10 > $O/probe.log 2>&1; echo "probe rc=$?"
for v in slp slpnop8 slpnop; do
  WSU_LIB=ws_unet_amd/libwsu_$v.so timeout -k 10 200 python tools/stress_pl.py 200 > $O/stress_$v.log 2>&1 || { echo "stress $v failed"; exit 1; }
  tail -1 $O/stress_$v.log
done
timeout -k 10 200 python tools/stress_pl.py 200 > $O/stress_product.log 2>&1 || exit 1
tail -1 $O/stress_product.log
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
tools/profile_sq.sh base > $O/sq.log 2>&1 || { echo "sq failed"; tail -5 $O/sq.log; }
timeout -k 10 300 python bench.py > $O/bench.log 2>&1 || { echo bench failed; exit 1; }
tail -c 400 $O/bench.log
