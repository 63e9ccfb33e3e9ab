# finer fp4 block scales (mantissa test): fp4 tests, the whole forward suite's f16f4p cases, MAE / speed of the default mode
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_planar.py tests/test_gpu_evaluate.py -x -q -m gpu -k "f16f4p or q4 or None or default" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --no-other-modes --no-train-step --no-latency --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.log 2>&1 || { tail -3 $O/bench.log; exit 1; }
python -c "
import json
d=json.loads(open('$O/bench.log').read().strip().split('\n')[-1]); print(round(d['value'],1), 'img/s', 'mae', d.get('mae_vs_cpu_oracle'), 'frac', round(d['roofline']['frac'],4))"
