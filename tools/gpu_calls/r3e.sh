# round-3 GPU call E: inputs-first DMA order; ablations of the product kernel; bench with the batch-1 latency leg
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planar.py tests/test_gpu_planar_train.py tests/test_gpu_forward.py tests/test_gpu_evaluate.py -q -x > $O/pytest_pl.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_pl.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/ablate_pl.py > $O/ablate.log 2>&1 || { echo ablate failed; tail -5 $O/ablate.log; exit 1; }
grep ablate= $O/ablate.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-other-modes > $O/bench.log 2>&1 || { echo bench failed; tail -5 $O/bench.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3e/bench.log').read().strip().split('\n')[-1])
print('img/s',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4),'per_layer',d['roofline']['per_layer']['frac'], d['roofline']['per_layer']['frac_2B'])
print(' '.join(f"{r['layer']}:{r['ms']}" for r in d['roofline']['per_layer']['layers']))
print('train', round(d['train_step']['ms_per_step'],2), d['train_step']['kernels_ms_per_step'])
l=d['latency_b1']; print({k:v for k,v in l.items() if k!='per_layer'})
for r in l['per_layer']: print(r)
P
timeout -k 10 300 python tools/bench_evaluate.py > $O/evaluate.log 2>&1; tail -1 $O/evaluate.log
