set -e
mkdir -p gpurun_out
WSU_CONV_MFMA=16 timeout -k 10 400 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -3
for mf in 32 16; do for nw in 8 4; do for mode in bf16x3 bf16; do
  for shp in "64 64 512 32" "128 128 256 32" "256 128 256 32" "256 256 128 32"; do
    WSU_CONV_MFMA=$mf WSU_CONV_WAVES=$nw timeout -k 10 120 python tools/ablate_conv.py $mode $shp | sed "s/^/mfma=$mf /"
  done
done; done; done | tee gpurun_out/ab_mfma16.log
