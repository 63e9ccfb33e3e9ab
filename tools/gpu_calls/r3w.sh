# after the epilogue overlap of the fp4 variants: race screen of the fp4 rings, the whole GPU suite, smoke
O=gpurun_out/r3w; mkdir -p $O
timeout -k 10 400 python tools/stress_pl.py 300 --q4 > $O/stress_q4.log 2>&1; rc=$?; tail -8 $O/stress_q4.log | cut -c1-200; [ $rc -eq 0 ] || exit 1
bash tools/gpu_calls/r3p.sh
