# how much of each plain 3x3-conv layer is its epilogue (planar-Q encoding + stores) and its DMA: conv3x3_q layers with WSU_PL_ABLATE = 0 / 2 (no epilogue) / 1 (no DMA after step 0)
O=gpurun_out/r6s; mkdir -p $O
for ab in 0 2 1 0; do
WSU_PL_ABLATE=$ab timeout -k 10 300 python tools/probe_q_layer.py 2>&1 | grep -v amdgpu | sed "s/^/ablate $ab: /" >> $O/probe.log || exit 1
done
cat $O/probe.log
