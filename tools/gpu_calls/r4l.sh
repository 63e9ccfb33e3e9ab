# 300 AdamW steps on the final kernels (sum kernels walking rows, finer fp4 scales do not touch training): the soak log of profiles/r03
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 500 python tools/soak_train.py 300 8 256 > $O/soak.log 2>&1; rc=$?; tail -6 $O/soak.log | cut -c1-200; exit $rc
