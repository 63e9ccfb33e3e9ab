O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_planar.py -x -q -m gpu -k "f16f4p or q4" > $O/pytest_f4.log 2>&1; rc=$?; tail -6 $O/pytest_f4.log | cut -c1-250; [ $rc -eq 0 ] || exit 1
for md in f16f8p f16f4p f16f8p f16f4p; do
  timeout -k 10 200 python bench.py --mode $md --no-other-modes --no-cpu-baseline --no-train-step --no-latency --steps 20 --warmup 5 > $O/bench_$md.log 2>&1 || { tail -3 $O/bench_$md.log; exit 1; }
  python -c "
import json,sys
d=json.loads(open('$O/bench_$md.log').read().strip().split('\n')[-1]); print('$md', round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'frac', round(d['roofline']['frac'],4), {r['layer']: r['ms'] for r in d['roofline']['per_layer']['layers']})"
done
