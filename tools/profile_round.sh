# Refreshes the numbers under profiles/: run on the GPU box from the repo root (tools/profile_round.sh), then copy gpurun_out/prof_round/*
set -e
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python3 $R/bench.py --no-other-modes --no-cpu-baseline > $O/bench_kt.log 2>&1
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py --no-other-modes --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_pmc_fetch.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py --no-other-modes --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_pmc_write.log 2>&1
echo "pmc write done"
cd $R
timeout -k 10 300 python tools/bench_evaluate.py > $O/evaluate_loop.log 2>&1
echo "evaluate done"
timeout -k 10 300 python tools/bench_ws_attack.py > $O/ws_attack.log 2>&1
timeout -k 10 300 python tools/bench_ws_attack.py --correct-bias >> $O/ws_attack.log 2>&1
echo "ws attack done"
timeout -k 10 120 tools/mfma_probe 3 > $O/mfma_probe.log 2>&1
timeout -k 10 60 tools/mfma_scale_layout_probe > $O/mfma_scale_layout_probe.log 2>&1
echo "probes done"
timeout -k 10 400 python bench.py > $O/bench_n1.log 2>&1
tail -1 $O/bench_n1.log | cut -c1-150
