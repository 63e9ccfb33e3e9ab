# per-image API: images per launch 8 (default) / 16 / 4, one box
O=gpurun_out/r6n; mkdir -p $O
for mb in 8 16 4 8 16; do
WSU_PER_IMAGE_BATCH=$mb timeout -k 10 300 python tools/profile_per_image.py --images 768 2>&1 | grep "per-image API" | sed "s/^/batch $mb: /" >> $O/rates.log || exit 1
done
cat $O/rates.log
