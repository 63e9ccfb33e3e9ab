O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 300 python tools/downstream_modes.py 128 2>&1 | grep -v amdgpu.ids | tee $O/downstream.log
