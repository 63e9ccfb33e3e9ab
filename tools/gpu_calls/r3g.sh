# round-3 GPU call G: MSPLIT (small grids) parity + batch-1 latency; same-box A/B against the end-of-round-2 tree
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_planar.py tests/test_gpu_forward.py tests/test_gpu_planar_train.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for ms in 1 0; do WSU_PL_MSPLIT=$ms timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes --no-train-step > $O/bench_msplit$ms.log 2>&1 || { echo bench failed; tail -5 $O/bench_msplit$ms.log; exit 1; }; done
python - <<'P'
import json
for ms in (1,0):
    d=json.loads(open(f'gpurun_out/r3g/bench_msplit{ms}.log').read().strip().split('\n')[-1])
    l=d['latency_b1']
    print('msplit',ms,'img/s',round(d['value'],1),'b1 gpu ms',l['gpu_ms_per_image'],'queued',round(l['wall_ms_per_image_queued'],4), ' '.join(f"{r['layer']}:{r['ms']}" for r in l['per_layer']))
P
if [ -d ab_r02 ]; then ( cd ab_r02 && timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes > ../$O/bench_r02tree.log 2>&1 ); python - <<'P'
import json
d=json.loads(open('gpurun_out/r3g/bench_r02tree.log').read().strip().split('\n')[-1])
print('r02 tree on this box: img/s',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'train',round(d['train_step']['ms_per_step'],2))
P
fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-modes > $O/bench_full.log 2>&1
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3g/bench_full.log').read().strip().split('\n')[-1])
print('this tree: img/s',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'train',round(d['train_step']['ms_per_step'],2))
P
