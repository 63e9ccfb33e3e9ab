# Copies the summaries of the round's profile runs (tools/profile_round.sh -> gpurun_out/prof_round, tools/profile_sq.sh r04 -> gpurun_out/sq_r04,
# the WSU_Q_ROWS=4 counter passes of tools/gpu_calls/r5h.sh -> gpurun_out/sq_r04_rows4) into profiles/r04/ -- run here after the GPU calls.
P=profiles/r04; G=gpurun_out/prof_round; S=gpurun_out/sq_r04; S4=gpurun_out/sq_r04_rows4
mkdir -p $P
tail -1 $G/bench_n1.log > $P/bench_n1.json.log
cp $G/bench_n1_detail.json $P/bench_n1_detail.json
cp $G/kt/bench_kernel_stats.csv $P/bench_kernel_stats.csv
cp $G/pmc_fetch/bench_counter_collection.csv $P/pmc_bench_FETCH_SIZE.csv; cp $G/pmc_write/bench_counter_collection.csv $P/pmc_bench_WRITE_SIZE.csv
cp $G/pmc_conv3x3_traffic.json $P/
cp $S/sq_counters.json $P/; cp $S/summary.md $P/sq_counters.md
cp $S4/sq_counters.json $P/sq_counters_rows4.json; cp $S4/summary.md $P/sq_counters_rows4.md
cp $G/kt_train/train_kernel_stats.csv $P/train_b64_kernel_stats.csv; cp $G/train_step_launches.log $P/train_step_launches.log
grep -h "^{" $G/ws_attack.log > $P/ws_attack_bench.json.log
python tools/per_layer_table.py $P/bench_n1_detail.json > $P/per_layer.md
grep blob $P/pmc_conv3x3_traffic.json; git hash-object ws_unet_amd/csrc/conv3x3_q.hip
