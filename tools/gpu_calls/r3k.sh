# gradient error vs the fp64 oracle for both backward `products`; soak; then the full GPU suite
O=gpurun_out/r3k; mkdir -p $O; rm -f $O/grad_error_vs_fp64_products.txt
for sz in 64 256; do for pr in f16f8 f16; do
  echo "== unet_2 ${sz}x${sz} L1WS, train_mode f16f8p, train_products $pr" >> $O/grad_error_vs_fp64_products.txt
  WSU_TRAIN_PRODUCTS=$pr timeout -k 10 600 python tools/diag_grads.py 2 $sz f16f8p >> $O/grad_error_vs_fp64_products.txt 2>&1 || { tail -5 $O/grad_error_vs_fp64_products.txt; exit 1; }
done; done
grep -E "^==|e11.weight|e31.weight|d31.weight|d42.weight" $O/grad_error_vs_fp64_products.txt | cut -c1-110
bash tools/gpu_calls/r3j.sh || exit 1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; exit $rc
