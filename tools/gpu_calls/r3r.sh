O=gpurun_out/r3r; mkdir -p $O
for q in 0 1; do for sz in 64 256; do
  echo "== train forward q4=$q size $sz" >> $O/grad_q4fwd.txt
  WSU_TRAIN_FWD_Q4=$q timeout -k 10 600 python tools/diag_grads.py 2 $sz f16f8p >> $O/grad_q4fwd.txt 2>&1 || exit 1
done; done
grep -E "^==|e11.weight|e22.weight|e32.weight|d31.weight|d42.weight" $O/grad_q4fwd.txt | cut -c1-100
for q in 0 1; do WSU_TRAIN_FWD_Q4=$q timeout -k 10 200 python tools/ab_products.py 64 512 1 2>&1 | grep -E "products=f16 " ; done
