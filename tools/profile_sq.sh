# SQ counter passes on the PRODUCT binaries (VERDICT r02 missing #4): rocprofv3 --pmc only (no trace flags), the program directly after `--`.
# Run on the GPU box from the repo root: tools/profile_sq.sh [tag]; summaries land in gpurun_out/sq_<tag>/ -> copy into profiles/rNN/.
set -e
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
TAG=${1:-base}
O=$R/gpurun_out/sq_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-other-modes --no-cpu-baseline --no-train-step --no-latency --no-trained-mae --steps 3 --warmup 1"
# 8 SQ slots per pass; GRBM_GUI_ACTIVE (effective clock) rides on the GRBM block
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE -d $O/fwd1 -o bench --output-format csv -- python3 $R/bench.py $B > $O/fwd1.log 2>&1 || { tail -5 $O/fwd1.log; echo "fwd pass 1 failed (counter names?)"; }
echo "fwd pass 1 done"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -d $O/fwd2 -o bench --output-format csv -- python3 $R/bench.py $B > $O/fwd2.log 2>&1 || { tail -5 $O/fwd2.log; echo "fwd pass 2 failed"; }
echo "fwd pass 2 done"
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE -d $O/train1 -o train --output-format csv -- python3 $R/tools/bench_train.py --batch 64 --steps 2 > $O/train1.log 2>&1 || { tail -5 $O/train1.log; echo "train pass failed"; }
echo "train pass done"
cd $R
python3 tools/pmc_sq.py $O --json $O/sq_counters.json > $O/summary.md 2>&1 || echo "summary failed"
cat $O/summary.md | head -60
