# round-3 GPU call H: fused-first-layer loader (single copy again) parity; default vs WSU_FUSE_FIRST_PL=1 on one box
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planar.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
F="--no-cpu-baseline --no-other-modes --no-train-step --no-latency --steps 30 --warmup 10"
for rep in 1 2; do
  timeout -k 10 200 python bench.py $F > $O/default_$rep.log 2>&1 || exit 1
  WSU_FUSE_FIRST_PL=1 timeout -k 10 200 python bench.py $F > $O/fused_$rep.log 2>&1 || exit 1
done
python - <<'P'
import json
for tree in ('default','fused'):
    for rep in (1,2):
        d=json.loads(open(f'gpurun_out/r3h/{tree}_{rep}.log').read().strip().split('\n')[-1])
        print(tree, rep, round(d['value'],1), d['kernel_ms_per_step'], [ (r['layer'], r['ms']) for r in d['roofline']['per_layer']['layers'][:3]])
P
