# Same-box, interleaved A/B of the predict bench: the end-of-round-2 tree (ab_r02/, `git archive fe4a282`, built there) against this tree
# in its default mode (f16f4p) and in round 2's arithmetic (--mode f16f8p).
# Boxes of this pool differ by several per cent (MI355X_MICROARCH.md, DVFS give-back item 5), so round-to-round numbers from different boxes say little.
O=gpurun_out/ab_trees; mkdir -p $O; rm -f $O/*.log
F="--no-cpu-baseline --no-other-modes --no-train-step --no-latency --steps 30 --warmup 10"
for rep in 1 2 3; do
  ( cd ab_r02 && timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-modes --no-train-step --steps 30 --warmup 10 > ../$O/r02_$rep.log 2>&1 ) || exit 1
  timeout -k 10 200 python bench.py $F > $O/r03_$rep.log 2>&1 || exit 1
  timeout -k 10 200 python bench.py $F --mode f16f8p > $O/r03f16f8p_$rep.log 2>&1 || exit 1
done
python - <<'P'
import json
for tree in ('r02','r03','r03f16f8p'):
    for rep in (1,2,3):
        d=json.loads(open(f'gpurun_out/ab_trees/{tree}_{rep}.log').read().strip().split('\n')[-1])
        k=d['kernel_ms_per_step']
        print(f"| {tree} | {rep} | {d['value']:.1f} | {d['ms_per_step']:.3f} | {d['roofline']['frac']:.4f} | {k.get('conv3x3_pl', 0):.3f} | {k.get('convt2x2_pl', 0):.3f} | {k.get('conv3x3_first_pl', 0):.3f} |")
P
