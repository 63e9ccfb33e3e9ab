# fused decoder entry, low chunks first: with and without the split last step (libwsu_qunoepo.so = -DWSU_QU_EPO=0), same box
O=gpurun_out/r5z; mkdir -p $O
for i in 1 2; do
timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_LIB=$PWD/ws_unet_amd/libwsu_qunoepo.so timeout -k 10 200 python tools/probe_qu_layer.py --no-two 2>&1 | sed 's/^/noepo: /' >> $O/probe.log || { tail -5 $O/probe.log; exit 1; }
done
grep -v amdgpu.ids $O/probe.log
