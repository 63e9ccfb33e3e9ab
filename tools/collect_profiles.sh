# Copies the summaries of the last tools/gpu_calls/r3final2.sh run (gpurun_out/) into profiles/r03/ -- run here after the GPU call.
P=profiles/r03; G=gpurun_out/prof_round; S=gpurun_out/sq_final8; O=gpurun_out/r3final8
python - <<'PY'
l=open('gpurun_out/r3final8/bench_with_summaries.log').read().strip().split('\n')
open('profiles/r03/bench_n1.json.log','w').write(l[-1]+'\n')
l=open('gpurun_out/r3final8/bench_torchrun1.log').read().strip().split('\n')
open('profiles/r03/bench_torchrun_world1.json.log','w').write('\n'.join(x for x in l if x.startswith('{') or 'WARNING' in x or 'master' in x.lower())+'\n')
PY
cp $G/kt/bench_kernel_stats.csv $P/bench_kernel_stats.csv
cp $G/pmc_fetch/bench_counter_collection.csv $P/pmc_bench_FETCH_SIZE.csv; cp $G/pmc_write/bench_counter_collection.csv $P/pmc_bench_WRITE_SIZE.csv
cp $G/pmc_conv3x3_traffic.json $P/; cp $S/sq_counters.json $P/; cp $S/summary.md $P/sq_counters_final.md
cp $G/kt_train/train_kernel_stats.csv $P/train_b64_kernel_stats.csv; cp $G/train_step_launches.log $P/train_step_launches.log
grep -h "^{" $G/evaluate_loop.log | tail -1 > /tmp/ev.json && [ -s /tmp/ev.json ] && (grep -v "^{" $P/evaluate_loop.json.log | head -1; cat /tmp/ev.json) > /tmp/ev2 && cp /tmp/ev2 $P/evaluate_loop.json.log
grep -h "^{" $G/ws_attack.log > /tmp/wa.json && [ -s /tmp/wa.json ] && cp /tmp/wa.json $P/ws_attack_bench.json.log
python tools/per_layer_table.py $P/bench_n1.json.log > $P/per_layer.md
grep blob $P/pmc_conv3x3_traffic.json; git hash-object ws_unet_amd/csrc/conv3x3_pl.hip
