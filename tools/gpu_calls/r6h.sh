# final call of round 4: whole GPU suite and smoke(), then the profile refresh (summaries keyed to the kernel sources' git blobs)
O=gpurun_out/r6h; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest_gpu.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 | tee $O/smoke.log || exit 1
bash tools/profile_round.sh 2>&1 | tail -9
bash tools/profile_sq.sh r04 2>&1 | grep "^|" | head -14
