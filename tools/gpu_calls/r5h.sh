# round 4, call h: tests of this turn's changes; SQ counter passes on the product binary for the default organisation (8 matrix waves) and for WSU_Q_ROWS=4
O=gpurun_out/r5h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_planar_train.py -x -q 2>&1 | grep -v "^$" | tail -8 | tee $O/pytest.log || exit 1
bash tools/profile_sq.sh r04 2>&1 | tail -30 | tee $O/sq_default.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
B="--no-other-modes --no-cpu-baseline --no-train-step --no-latency --no-trained-mae --steps 3 --warmup 1"
export WSU_Q_ROWS=4
mkdir -p $R/gpurun_out/sq_r04_rows4
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE -d $R/gpurun_out/sq_r04_rows4/fwd1 -o bench --output-format csv -- python3 $R/bench.py $B > $R/gpurun_out/sq_r04_rows4/fwd1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/sq_r04_rows4/fwd2 -o bench --output-format csv -- python3 $R/bench.py $B > $R/gpurun_out/sq_r04_rows4/fwd2.log 2>&1
unset WSU_Q_ROWS
cd $R
python3 tools/pmc_sq.py gpurun_out/sq_r04_rows4 --json gpurun_out/sq_r04_rows4/sq_counters.json > gpurun_out/sq_r04_rows4/summary.md 2>&1
head -12 gpurun_out/sq_r04_rows4/summary.md
