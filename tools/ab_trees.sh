# Same-box, interleaved A/B of the predict bench: the end-of-round-2 tree (ab_r02/, `git archive fe4a282`, built there) against this tree.
# Boxes of this pool differ by several per cent (MI355X_MICROARCH.md, DVFS give-back item 5), so round-to-round numbers from different boxes say little.
O=gpurun_out/ab_trees; mkdir -p $O
F="--no-cpu-baseline --no-other-modes --no-train-step --no-latency --steps 30 --warmup 10"
for rep in 1 2 3; do
  ( cd ab_r02 && timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-modes --no-train-step --steps 30 --warmup 10 > ../$O/r02_$rep.log 2>&1 ) || exit 1
  timeout -k 10 200 python bench.py $F > $O/r03_$rep.log 2>&1 || exit 1
done
python - <<'P'
import json
for tree in ('r02','r03'):
    vals=[]
    for rep in (1,2,3):
        d=json.loads(open(f'gpurun_out/ab_trees/{tree}_{rep}.log').read().strip().split('\n')[-1])
        vals.append((round(d['value'],1), round(d['roofline']['frac'],4), d['kernel_ms_per_step']))
    print(tree, vals)
P
