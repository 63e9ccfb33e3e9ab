# phase stamps of the fp4 variant (stamps build, plain step loop)
O=gpurun_out/r3u; mkdir -p $O
for sh in "64 64 512 --pool" "64 128 256" "128 128 256 --pool" "256 256 128" "256 128 256 128" "128 64 512 64"; do
  timeout -k 10 120 python tools/stamp_pl.py $sh --q4 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps_q4.log || exit 1
done
