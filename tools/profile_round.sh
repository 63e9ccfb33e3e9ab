# Refreshes the numbers under profiles/: run on the GPU box from the repo root (tools/profile_round.sh), then copy gpurun_out/prof_round/*
set -e   # (a failing step stops the script: nothing below it would be trustworthy)
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-other-modes --no-cpu-baseline --no-train-step --no-latency --no-trained-mae"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python3 $R/bench.py $B > $O/bench_kt.log 2>&1
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python3 $R/bench.py $B --steps 3 --warmup 1 > $O/bench_pmc_fetch.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python3 $R/bench.py $B --steps 3 --warmup 1 > $O/bench_pmc_write.log 2>&1
echo "pmc write done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_train -o train --output-format csv -- python3 $R/tools/bench_train.py --batch 64 --steps 3 > $O/train_kt.log 2>&1
echo "train kernel trace done"
cd $R
python3 tools/pmc_traffic.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) $(ls $O/pmc_write/*counter_collection.csv | head -1) $O/pmc_conv3x3_traffic.json conv3x3_q_kernel f16f4p > $O/pmc_traffic.log 2>&1 || echo "pmc summary failed"
timeout -k 10 300 python tools/bench_evaluate.py > $O/evaluate_loop.log 2>&1
echo "evaluate done"
timeout -k 10 300 python tools/bench_ws_attack.py > $O/ws_attack.log 2>&1
timeout -k 10 300 python tools/bench_ws_attack.py --correct-bias >> $O/ws_attack.log 2>&1
echo "ws attack done"
WSU_TIME_TRAIN_LAUNCHES=1 timeout -k 10 300 python tools/time_train.py f16f8p 64 512 > $O/train_step.json 2> $O/train_step_launches.log
echo "train launches done"
timeout -k 10 500 python bench.py --detail $O/bench_n1_detail.json > $O/bench_n1.log 2>&1
tail -1 $O/bench_n1.log | cut -c1-150
