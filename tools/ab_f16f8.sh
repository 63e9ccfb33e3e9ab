# A/B of libwsu builds on single conv3x3 layers in mode f16f8 (same box, same call): tools/ab_f16f8.sh lib1.so lib2.so ...
set -e
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  for shape in "256 128 256" "128 64 512" "128 128 256" "64 64 512"; do
    echo -n "$lib: "
    WSU_LIB=$lib timeout -k 10 120 python tools/ablate_conv.py f16f8 $shape 32
  done
done
done 2>&1 | tee gpurun_out/ab_f16f8.log
