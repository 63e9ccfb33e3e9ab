# round-3 GPU call F: lean wgrad loaders, fma_mix decode, pool epilogue trim -- training kernels' parity, then bench
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_planar.py tests/test_gpu_planar_train.py tests/test_gpu_forward.py tests/test_gpu_backward_large.py tests/test_gpu_dp.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline --no-other-modes > $O/bench.log 2>&1 || { echo bench failed; tail -5 $O/bench.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3f/bench.log').read().strip().split('\n')[-1])
print('img/s',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4),'per_layer',d['roofline']['per_layer']['frac'], d['roofline']['per_layer']['frac_2B'])
print(' '.join(f"{r['layer']}:{r['ms']}" for r in d['roofline']['per_layer']['layers']))
print('train', round(d['train_step']['ms_per_step'],2), d['train_step']['kernels_ms_per_step'])
print('b1', d['latency_b1']['gpu_ms_per_image'], d['latency_b1']['wall_ms_per_image_queued'])
P
WSU_TIME_TRAIN_LAUNCHES=1 timeout -k 10 300 python tools/time_train.py f16f8p 64 512 > $O/train_step.json 2> $O/train_step_launches.log; cat $O/train_step_launches.log | head -45
