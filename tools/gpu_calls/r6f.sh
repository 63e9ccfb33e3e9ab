# fused first layer: the computing loader waves at priority 3 (WSU_Q_F1_PRIO=1) against priority 0 and against the two-kernel path, one box
O=gpurun_out/r6f; mkdir -p $O
B="--no-cpu-baseline --no-train-step --no-latency --no-trained-mae --no-other-modes"
for i in 1 2; do
WSU_FUSE_FIRST_Q=0 timeout -k 10 300 python bench.py $B --detail $O/sep_$i.json > $O/sep_$i.log 2>&1 || exit 1
timeout -k 10 300 python bench.py $B --detail $O/f1_$i.json > $O/f1_$i.log 2>&1 || exit 1
WSU_Q_F1_PRIO=1 timeout -k 10 300 python bench.py $B --detail $O/f1prio_$i.json > $O/f1prio_$i.log 2>&1 || exit 1
done
python - <<'PY'
import json
for f in ("sep_1","f1_1","f1prio_1","sep_2","f1_2","f1prio_2"):
    d=json.load(open(f"gpurun_out/r6f/{f}.json")); pl=d["roofline"]["per_layer"]["layers"]
    print(f, round(d["value"],1), [(r["layer"], r["ms"]) for r in pl[:2]])
PY
