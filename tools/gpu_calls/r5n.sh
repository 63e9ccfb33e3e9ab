# round 4, call n: conv3x3_q on random and on all-zero operands (power management), and the probe build without the unpaired ninth tap's fp4 instruction
O=gpurun_out/r5n; mkdir -p $O
for i in 1 2; do
  timeout -k 10 200 python tools/probe_q_layer.py 2>/dev/null | tee -a $O/probe_q_layer.log
  WSU_LIB=$PWD/ws_unet_amd/libwsu_qnox8.so timeout -k 10 200 python tools/probe_q_layer.py 2>/dev/null | tee -a $O/probe_q_layer.log
done
timeout -k 10 200 python tools/probe_q_layer.py --zeros 2>/dev/null | tee -a $O/probe_q_layer.log
