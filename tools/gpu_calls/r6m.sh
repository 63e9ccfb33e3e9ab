# the whole GPU suite and smoke() on the round's last tree
O=gpurun_out/r6m; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -6 | tee $O/pytest_gpu.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 | tee $O/smoke.log || exit 1
