# round-3 final measurement set (one box): full GPU suite, smoke, kernel traces + PMC traffic + SQ counters, evaluate / attack / train tools, soak, 1024^2 per-GPU shape
O=gpurun_out/r3final; mkdir -p $O
timeout -k 10 800 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile_round rc=$?"; tail -3 $O/profile_round.log
bash tools/profile_sq.sh final > $O/sq.log 2>&1; echo "sq rc=$?"
head -18 gpurun_out/sq_final/summary.md
timeout -k 10 400 python tools/soak_train.py 300 8 256 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -3 $O/soak.log
timeout -k 10 200 python tools/time_train.py f16f8p 8 1024 > $O/train_1024_b8.json 2>/dev/null; tail -c 300 $O/train_1024_b8.json
