# 300 AdamW steps in three arithmetics from the same init on the same batch: exact fp32, planar with f16f8 products, planar with f16 products
O=gpurun_out/r3j; mkdir -p $O
for cfg in "f32 f16f8" "f16f8p f16f8" "f16f8p f16"; do
  set -- $cfg
  WSU_SOAK_JSON=$O/soak_$1_$2.json timeout -k 10 300 python tools/soak_train.py 300 8 256 $1 $2 > $O/soak_$1_$2.log 2>&1 || { tail -5 $O/soak_$1_$2.log; exit 1; }
  tail -1 $O/soak_$1_$2.log | cut -c1-400
done
python tools/soak_compare.py $O/soak_f32_f16f8.json $O/soak_f16f8p_f16f8.json $O/soak_f16f8p_f16.json > $O/soak_compare.md; cat $O/soak_compare.md
