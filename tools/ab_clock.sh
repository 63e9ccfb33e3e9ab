set -e
mkdir -p gpurun_out
for ab in 0 1 2 3 4 5 7; do
  echo "== EXTRA_ABLATE=$ab"
  EXTRA_ABLATE=$ab WSU_CONV_WAVES=4 timeout -k 10 120 python tools/stamp_conv.py bf16x3 64 64 512 32 | grep -E "clock|lifetimes"
done 2>&1 | tee gpurun_out/ab_clock.log
for ab in 0 3 4; do
  echo "== d31 EXTRA_ABLATE=$ab"
  EXTRA_ABLATE=$ab WSU_CONV_WAVES=4 timeout -k 10 120 python tools/stamp_conv.py bf16x3 256 128 256 32 | grep -E "clock|lifetimes"
done 2>&1 | tee -a gpurun_out/ab_clock.log
