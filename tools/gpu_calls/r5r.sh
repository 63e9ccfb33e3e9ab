# round 4, call r: the accuracy margin of the inference modes on weights trained in three published-style configurations (1 500 steps each, fresh images per step)
O=gpurun_out/r5r; mkdir -p $O
timeout -k 10 1000 python tools/trained_mae_study.py 1500 2>/dev/null | grep "^{" | tee $O/trained_mae_study.json.log
