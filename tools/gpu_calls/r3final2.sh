# refresh of the blob-keyed summaries after the last edit of conv3x3_pl.hip: PMC traffic + SQ counters + the bench line; RCCL world-size-1 launch check
O=gpurun_out/r3final8; mkdir -p $O
bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile_round rc=$?"
bash tools/profile_sq.sh final8 > $O/sq.log 2>&1; echo "sq rc=$?"
mkdir -p profiles/r03; cp gpurun_out/prof_round/pmc_conv3x3_traffic.json profiles/r03/; cp gpurun_out/sq_final8/sq_counters.json profiles/r03/
timeout -k 10 300 python bench.py > $O/bench_with_summaries.log 2>&1; tail -c 300 $O/bench_with_summaries.log
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-other-modes --no-train-step --no-latency > $O/bench_torchrun1.log 2>&1; echo "torchrun rc=$?"; tail -c 200 $O/bench_torchrun1.log
