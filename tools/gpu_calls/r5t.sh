# where the fused decoder entry's time goes: the layer probe plain, without DMA after the first allocations, without the epilogue, on zeros
O=gpurun_out/r5t; mkdir -p $O
timeout -k 10 200 python tools/probe_qu_layer.py > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=1 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=2 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=3 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
timeout -k 10 200 python tools/probe_qu_layer.py --zeros >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
cat $O/probe.log
