# fused decoder entry: the matrix waves fetch the low half's weights themselves (WSU_QU_MWLOW=1) against the loaders doing it, one box
O=gpurun_out/r6k; mkdir -p $O
WSU_QU_MWLOW=1 timeout -k 10 400 python -m pytest tests/test_gpu_qu.py -x -q > $O/pytest_mwlow.log 2>&1 || { tail -30 $O/pytest_mwlow.log; exit 1; }
tail -1 $O/pytest_mwlow.log
timeout -k 10 400 python -m pytest tests/test_gpu_qu.py -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for i in 1 2; do
timeout -k 10 200 python tools/probe_qu_layer.py --no-two >> $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
WSU_QU_MWLOW=1 timeout -k 10 200 python tools/probe_qu_layer.py --no-two 2>&1 | sed 's/^/mwlow: /' >> $O/probe.log || { tail -5 $O/probe.log; exit 1; }
done
grep -v amdgpu.ids $O/probe.log
