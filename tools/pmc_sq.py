"""Summary of the SQ counter passes of tools/profile_sq.sh: per kernel (conv3x3_pl_kernel, wgrad_ring_kernel, ...) the matrix-pipe busy
fraction, the wave-cycle split and the effective clock, from rocprofv3 --pmc CSVs of the PRODUCT binaries (no stamps build).

    python tools/pmc_sq.py <dir with fwd1/ fwd2/ train1/ subdirectories> [--json out.json]

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the chip's 1024 SIMDs;
SQ_BUSY_CYCLES sums over the 32 shader engines (32 SIMDs each), so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)                 (clock-free)
and, with the effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time,
    mfma_busy_t = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * clock * time).
SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY count quad-cycles per wave (disjoint shares of a wave's life)."""
import csv
import glob
import hashlib
import json
import sys
from collections import defaultdict
from pathlib import Path

KERNELS = ["conv3x3_qu_kernel", "conv3x3_q_kernel", "conv3x3_pl_kernel", "convt2x2_pl_kernel", "first_pl_kernel", "wgrad_ring_kernel", "wgrad_pl_kernel", "pool_bwd_pl", "convt2x2_bwd_pl", "head_bwd_pl"]


def git_blob_sha1(path: Path) -> str:
    data = path.read_bytes()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def short(name: str):
    for k in KERNELS:
        if k in name:
            if k == "conv3x3_pl_kernel" and "<" in name:
                # template arguments <HC, POOL, XRES, F1, GRAD>: the data-gradient variant is its own row
                args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
                if len(args) >= 5 and args[4] == "true":
                    return "conv3x3_pl_kernel<GRAD>"
            return k
    return None


def load(pass_dir: Path):
    files = glob.glob(str(pass_dir / "**" / "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(dict)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if k is None:
                    continue
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for k in acc:
        acc[k]["_time_s"] = sum(disp[k].values())
        acc[k]["_launches"] = len(disp[k])
    return acc


def summarise(c):
    out = {"launches": int(c["_launches"]), "avg_launch_ms": c["_time_s"] / max(c["_launches"], 1) * 1e3}
    t = c["_time_s"]
    if c.get("GRBM_GUI_ACTIVE") and t > 0:
        out["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        out["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * c["SQ_BUSY_CYCLES"])
        if "clock_GHz" in out:
            out["mfma_busy_t"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * out["clock_GHz"] * 1e9 * t)
    w = c.get("SQ_WAVE_CYCLES")
    if w:
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if name in c:
                out[name.lower() + "_share"] = c[name] / w
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F8", "SQ_VALU_MFMA_COEXEC_CYCLES"):
        if name in c:
            out[name.lower() + "_per_launch"] = c[name] / max(c["_launches"], 1)
    return out


def main():
    root = Path(sys.argv[1])
    res = {}
    for pass_name in ("fwd1", "fwd2", "train1", "train2"):
        d = root / pass_name
        if not d.is_dir():
            continue
        for k, c in load(d).items():
            res.setdefault(k, {}).setdefault(pass_name, summarise(c))
    repo = Path(__file__).resolve().parent.parent
    meta = {"kernel_source_blobs": {f: git_blob_sha1(repo / "ws_unet_amd" / "csrc" / f) for f in ("conv3x3_q.hip", "conv3x3_qu.hip", "conv3x3_pl.hip", "wgrad.hip")},
            "counters": "rocprofv3 --pmc passes of bench.py / tools/bench_train.py on the product libwsu.so (tools/profile_sq.sh)"}
    print("| kernel | pass | launches | avg ms | clock GHz | MFMA busy (cycles) | MFMA busy (time) | wait_any | wait_inst | active_inst |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    f = lambda v, spec=".3f": "-" if v is None else format(v, spec)
    for k, passes in sorted(res.items()):
        for p, s in passes.items():
            print(f"| {k} | {p} | {s['launches']} | {f(s['avg_launch_ms'])} | {f(s.get('clock_GHz'))} | {f(s.get('mfma_busy'))} | {f(s.get('mfma_busy_t'))} | "
                  f"{f(s.get('sq_wait_any_share'))} | {f(s.get('sq_wait_inst_any_share'))} | {f(s.get('sq_active_inst_any_share'))} |")
    print()
    print(json.dumps({"meta": meta, "kernels": res}, indent=1))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump({"meta": meta, "kernels": res}, fh, indent=1)


if __name__ == "__main__":
    main()
