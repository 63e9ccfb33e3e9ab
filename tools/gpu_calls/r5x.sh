# LDS counters of the fused decoder entry beside the two-kernel path (same probe script): bank conflicts, LDS-array activity, LDS issue stalls
O=$GRAFT_REPO_ROOT/gpurun_out/r5x; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/avail.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL -d $O/p1 -o probe --output-format csv -- python3 $R/tools/probe_qu_layer.py > $O/p1.log 2>&1 || { tail -5 $O/p1.log; echo "pass 1 failed"; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_BUSY_CYCLES -d $O/p2 -o probe --output-format csv -- python3 $R/tools/probe_qu_layer.py > $O/p2.log 2>&1 || { tail -5 $O/p2.log; echo "pass 2 failed"; }
cd $R
python3 tools/lds_counters.py $O/p1 $O/p2 > $O/summary.txt 2>&1
cat $O/summary.txt
rm -rf $O/p1/*/*agent_info.csv
