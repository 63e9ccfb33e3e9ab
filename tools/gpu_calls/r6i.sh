# the bench line once more on the committed summaries (traffic / mfma_busy are read from profiles/r04, keyed to the kernel sources' blobs)
O=gpurun_out/prof_round; mkdir -p $O
timeout -k 10 500 python bench.py --detail $O/bench_n1_detail.json > $O/bench_n1.log 2>&1 || { tail -5 $O/bench_n1.log; exit 1; }
tail -1 $O/bench_n1.log | cut -c1-300
