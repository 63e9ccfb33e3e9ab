# race screen of the optional fused first layer (conv3x3_q kernel variant F1) and once more of the fused decoder entry
O=gpurun_out/r6r; mkdir -p $O
timeout -k 10 600 python tools/stress_qu.py 100 --f1 > $O/stress.log 2>&1 || { tail -5 $O/stress.log; exit 1; }
grep -v amdgpu $O/stress.log
