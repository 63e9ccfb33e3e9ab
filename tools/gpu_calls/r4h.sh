O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 200 python tools/time_evaluate_chunk.py 32 2>&1 | grep -v amdgpu.ids | tee $O/chunk.log
