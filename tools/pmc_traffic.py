"""Per-launch HBM traffic of the dominant kernel from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE;
separate passes: the TCC block has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte read requests at 64 B -> doubled; WRITE_SIZE is exact.
Both are reported by rocprofv3 in KiB.

python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel-substring] [mode]"""
import csv
import hashlib
import json
import sys
from pathlib import Path


def git_blob_sha1(path: Path) -> str:
    """`git hash-object` of the kernel source the counters were collected on: bench.py refuses the summary when the tree differs."""
    data = path.read_bytes()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def per_launch(path, counter, kernel):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"] and "pack_" not in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]) * 1024.0)
    return vals


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "conv3x3_q_kernel"
    mode = sys.argv[5] if len(sys.argv) > 5 else "f16f4p"
    source = "conv3x3_q.hip" if "conv3x3_q" in kernel else "conv3x3_pl.hip" if "conv3x3_pl" in kernel else "conv3x3.hip"
    f = per_launch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_launch(write_csv, "WRITE_SIZE", kernel)
    assert f and w and len(f) == len(w), (len(f), len(w))
    res = {
        "kernel": kernel, "mode": mode, "launches": len(f),
        "fetch_bytes_per_launch": 2.0 * sum(f) / len(f),          # gfx950: x2
        "write_bytes_per_launch": sum(w) / len(w),
        "counters": "FETCH_SIZE (KiB, x2 on gfx950) and WRITE_SIZE (KiB), one rocprofv3 --pmc pass each",
        "kernel_source": source,
        "kernel_source_blob": git_blob_sha1(Path(__file__).resolve().parent.parent / "ws_unet_amd" / "csrc" / source),
    }
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
