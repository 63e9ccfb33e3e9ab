# per-image API (16 images per launch): host profile by cumulative time; the evaluate tests on the new default
O=gpurun_out/r6o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 400 python tools/profile_per_image.py --images 768 > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
grep -v amdgpu $O/profile.log | grep -A 45 "Ordered by: cumulative" | cut -c1-150
