# round 4, call e: whole GPU suite after the removal of round 3's Q4 variant; then the round's profiles (kernel trace, PMC traffic, full bench line)
O=gpurun_out/r5e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -15 | tee $O/pytest_gpu.log && bash tools/profile_round.sh 2>&1 | tail -12
