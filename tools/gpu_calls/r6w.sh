# experiment: cache-policy bits on the INPUT pieces of the loaders' LDS-DMA (aux = 2: nt, aux = 1: sc0; experiment builds libwsu_inaux{2,1}.so) so that the
# activation stream does not displace the weights in L2 -- single layers (plain and fused) and the forward, one box
O=gpurun_out/r6w; mkdir -p $O
for lib in libwsu.so libwsu_inaux2.so libwsu_inaux1.so libwsu.so libwsu_inaux2.so; do
WSU_LIB=$PWD/ws_unet_amd/$lib timeout -k 10 300 python tools/probe_q_layer.py 2>&1 | grep -v amdgpu >> $O/probe.log || exit 1
WSU_LIB=$PWD/ws_unet_amd/$lib timeout -k 10 300 python tools/probe_qu_layer.py --no-two 2>&1 | grep -v amdgpu | sed "s/^/$lib: /" >> $O/probe.log || exit 1
done
cut -c1-125 $O/probe.log
