# evaluate loop under the kernel trace: GPU time per chunk of 32 vs wall
O=gpurun_out/r4g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof -o ev -- python3 tools/bench_evaluate.py --images 1024 --batch 32 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep "^{" $O/run.log | cut -c1-200
f=$(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1); head -14 $f | cut -c1-160
m=$(ls $O/prof/*/*memory_copy_stats.csv $O/prof/*memory_copy_stats.csv 2>/dev/null | head -1); [ -n "$m" ] && head -5 $m | cut -c1-160
