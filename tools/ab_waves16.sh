set -e
mkdir -p gpurun_out
WSU_CONV_WAVES=16 timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -3
for nw in 8 16; do for mode in bf16x3 bf16; do
  for shp in "64 64 512 32" "128 128 256 32" "256 128 256 32" "256 256 128 32" "128 64 512 32"; do
    WSU_CONV_WAVES=$nw timeout -k 10 120 python tools/ablate_conv.py $mode $shp
  done
done; done | tee gpurun_out/ab_waves16.log
