"""Per-kernel sums of the LDS counters of a rocprofv3 --pmc pass over tools/probe_qu_layer.py: python tools/lds_counters.py <dir> [<dir> ...]"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "conv3x3_q" not in k and "convt2x2_pl" not in k:
                continue
            k = "conv3x3_qu" if "conv3x3_qu" in k else ("convt2x2_pl" if "convt2x2" in k else "conv3x3_q")
            k += f" grid={r['Grid_Size']}"
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v / max(1, len(n[(k, c)])):16.0f} per launch ({len(n[(k, c)])} launches)")
