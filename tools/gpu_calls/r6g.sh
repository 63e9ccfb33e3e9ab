# fused first layer: batch-1 latency leg with and without it, one box
O=gpurun_out/r6g; mkdir -p $O
B="--no-cpu-baseline --no-train-step --no-trained-mae --no-other-modes"
for i in 1 2; do
WSU_FUSE_FIRST_Q=0 timeout -k 10 300 python bench.py $B --detail $O/sep_$i.json > $O/sep_$i.log 2>&1 || exit 1
timeout -k 10 300 python bench.py $B --detail $O/f1_$i.json > $O/f1_$i.log 2>&1 || exit 1
done
python - <<'PY'
import json
for f in ("sep_1","f1_1","sep_2","f1_2"):
    d=json.load(open(f"gpurun_out/r6g/{f}.json")); lb=d["latency_b1"]
    print(f, round(d["value"],1), {k: round(v,4) for k,v in lb.items() if isinstance(v,(int,float))})
PY
