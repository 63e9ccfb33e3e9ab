# per-image API after the records iterator and one stat per row: tests, rate, host profile
O=gpurun_out/r6c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python tools/profile_per_image.py > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
grep -v amdgpu $O/profile.log | head -45
