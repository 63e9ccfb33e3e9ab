set -e
mkdir -p gpurun_out
for ab in 0 16; do for nw in 8 4; do for mode in bf16x3 bf16; do
  for shp in "64 64 512 32" "128 128 256 32" "256 128 256 32" "256 256 128 32"; do
    WSU_CONV_ABLATE=$ab WSU_CONV_WAVES=$nw timeout -k 10 120 python tools/ablate_conv.py $mode $shp
  done
done; done; done | tee gpurun_out/ab_twothirds.log
