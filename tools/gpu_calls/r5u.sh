# fused decoder entry: loader waves at priority 3 against the plain build, same box
O=gpurun_out/r5u; mkdir -p $O
timeout -k 10 200 python tools/probe_qu_layer.py --no-two > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=4 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_ABLATE=4 timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
grep -v amdgpu.ids $O/probe.log
