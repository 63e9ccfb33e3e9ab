# A/B of WSU_CONV_WAVES settings on single conv3x3 layers in mode f16f8 (same box, same call): tools/ab_f16f8.sh 8 4
set -e
mkdir -p gpurun_out
for rep in 1 2; do
for nw in "$@"; do
  for shape in "256 128 256" "128 64 512" "128 128 256" "64 64 512" "128 256 128" "256 256 128"; do
    echo -n "waves=$nw: "
    WSU_CONV_WAVES=$nw timeout -k 10 120 python tools/ablate_conv.py f16f8 $shape 32 2>&1 | grep us
  done
done
done 2>&1 | tee gpurun_out/ab_f16f8.log
