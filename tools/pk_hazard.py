"""The packed-f32 read-after-write sites behind round 2's intermittent fused-head failure, as code:

  scan   <file.s> ...           list every site of the pattern in gfx950 assembly (also what tests/test_isa_lint.py runs)
  build                         build two diagnostic libraries next to libwsu.so from conv3x3_pl.hip compiled WITH the SLP vectorizer:
                                  libwsu_slp.so     as the compiler schedules it (the build that failed ~25 % of the launches)
                                  libwsu_slpnop_w.so the same assembly with `s_nop 0` inserted at the sites separated only by an s_waitcnt
                                  libwsu_slpnop_a.so ... only at the ADJACENT pairs (v_pk_add_f32 op_sel_hi:[0,1] -> v_max_f32: hipcc does not pad these)
                                  libwsu_slpnop.so   ... at every site
                                (tools/stress_pl.py with WSU_LIB=... runs either; profiles/r03/pk_hazard.md has the result)

The pattern: a VOP3P packed-f32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) writes a VGPR pair and the next
vector instruction reads a register of that pair, with NOTHING in between that occupies an issue cycle -- either nothing at all or only
`s_waitcnt`.  hipcc (LLVM's GCNHazardRecognizer, the dst-sel forwarding rule: one wait state between such a producer and its consumer)
pads these pairs with `s_nop 0` or an independent instruction, but it counts an `s_waitcnt` as that wait state
(SIInstrInfo::getNumWaitStates: 1 for everything but s_nop / meta instructions).  An `s_waitcnt` whose counters are already satisfied
retires inside the wave's instruction buffer without taking an issue cycle, so whether the consumer sees the producer's result then
depends on whether the LDS reads in flight had landed -- the timing-dependent wrong logits of round 2.
"""
import re
import shlex
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "ws_unet_amd" / "csrc"
PK = re.compile(r"^\s*(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32|v_pk_mov_b32)\s+v\[(\d+):(\d+)\]\s*,(.*)$")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NO_ISSUE_CYCLE = ("s_waitcnt",)          # s_waitcnt, s_waitcnt_vscnt, ...: retire in the instruction buffer when already satisfied


def _reads(operands: str):
    regs = set()
    for m in VREG.finditer(operands):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def find_sites(asm: str):
    """[(kernel, line number (1-based) of the producer, producer text, [instructions in between], consumer text)]"""
    lines = asm.split("\n")
    sites, kern = [], None
    for i, raw in enumerate(lines):
        m = re.match(r"^([A-Za-z_]\S*):\s*(;.*)?$", raw)
        if m and not raw.startswith(".L"):
            kern = m.group(1)
        m = PK.match(raw)
        if not m:
            continue
        dst = set(range(int(m.group(2)), int(m.group(3)) + 1))
        between, j = [], i + 1
        while j < len(lines):
            t = lines[j].split(";")[0].strip()
            j += 1
            if not t or t.startswith("."):
                if t.endswith(":"):                       # a label: the fall-through successor still follows; branch targets are not followed
                    continue
                continue
            op = t.split()[0]
            if op.startswith(NO_ISSUE_CYCLE):
                between.append(t)
                continue
            if op.startswith("v_"):
                ops = t[len(op):]
                # a VALU consumer: its source operands are everything after the first (destination) operand -- for the packed
                # accumulate forms the destination is a source too, so take all registers named after the mnemonic's first comma
                srcs = ops.split(",", 1)[1] if "," in ops else ""
                if _reads(srcs) & dst:
                    sites.append((kern, i + 1, raw.strip(), between, t))
            break                                            # any other instruction (s_nop included) is an issue cycle
    return sites


def compiler_pads(producer: str) -> bool:
    """True for producers hipcc treats as dst-sel-forwarding (src0's op_sel_hi bit set -- the default of a packed instruction; LLVM reads
    it as DST_OP_SEL of a VOP3 encoding): it separates these from their consumers by one wait state, so an ADJACENT pair of this kind can
    only come from hand-written assembly.  Producers with op_sel_hi[0] = 0 (e.g. `v_pk_add_f32 ... op_sel_hi:[0,1]`) are left adjacent
    to their consumers by the compiler and run correctly (ws_stats_kernel and loss_reduce_kernel hold such pairs and are bit-exact)."""
    m = re.search(r"op_sel_hi:\[(\d)", producer)
    return m is None or m.group(1) == "1"


def patch(asm: str, which: str = "all"):
    """`s_nop 0` directly behind the producer of every site; which = 'waitcnt': only where an s_waitcnt is all that separates the pair,
    'adjacent': only the pairs with nothing in between, 'all'.  Returns (patched text, number of sites patched)."""
    lines = asm.split("\n")
    sites = [s for s in find_sites(asm) if which == "all" or (which == "waitcnt") == bool(s[3])]
    for _, ln, *_ in sorted(sites, key=lambda s: -s[1]):
        lines.insert(ln, "\ts_nop 0")
    return "\n".join(lines), len(sites)


def scan_files(paths):
    total = 0
    for p in paths:
        sites = find_sites(Path(p).read_text())
        total += len(sites)
        for kern, ln, prod, between, cons in sites:
            print(f"{p}:{ln}: {kern}\n    {prod}\n    {' ; '.join(between) or '(adjacent)'}\n    {cons}")
    print(f"{total} packed-f32 RAW site(s) without an issue cycle in between")
    return total


def hipcc_steps(src: Path, out_obj: Path, workdir: Path, flags):
    """The sub-commands `hipcc -save-temps -c` would run, as argument lists (hipcc -###)."""
    cmd = ["/opt/rocm/bin/hipcc", *flags, "-save-temps", "-c", str(src), "-o", str(out_obj), "-###"]
    r = subprocess.run(cmd, cwd=workdir, capture_output=True, text=True, check=True)
    steps = [shlex.split(ln) for ln in r.stderr.splitlines() if ln.startswith(' "')]
    assert len(steps) == 10, f"unexpected hipcc pipeline ({len(steps)} steps)"
    return steps


def build_variants(root: Path = ROOT):
    """`root`: the tree to build in (default: this one; `repro_r02/` = an export of the round-2 commit whose SLP build failed)."""
    csrc = root / "ws_unet_amd" / "csrc"
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
    srcs = re.search(r"^SRCS\s*=\s*(.*)$", (csrc / "Makefile").read_text(), re.M).group(1).split()
    objs = [str(csrc / (n[:-4] + ".o")) for n in srcs if n != "conv3x3_pl.hip"]
    for o in objs:
        assert Path(o).exists(), f"{o} missing: run make -C {csrc} first"
    for variant in ("slp", "slpnop_w", "slpnop_a", "slpnop"):
        work = Path("/tmp") / f"wsu_{root.name}_{variant}"
        work.mkdir(exist_ok=True)
        obj = work / "conv3x3_pl.o"
        steps = hipcc_steps(csrc / "conv3x3_pl.hip", obj, work, flags)
        dev_s = work / "conv3x3_pl-hip-amdgcn-amd-amdhsa-gfx950.s"
        for k, st in enumerate(steps):
            subprocess.run(st, cwd=work, check=True, stderr=subprocess.DEVNULL)
            if k == 2:                                       # the device assembly exists now
                text = dev_s.read_text()
                n_before = len(find_sites(text))
                n = 0
                if variant != "slp":
                    text, n = patch(text, {"slpnop_w": "waitcnt", "slpnop_a": "adjacent", "slpnop": "all"}[variant])
                    dev_s.write_text(text)
                left = find_sites(text)
                print(f"[{variant}] {n_before} site(s) in the device assembly, {n} patched with s_nop 0; left: {sum(1 for s_ in left if s_[3])} behind an s_waitcnt, "
                      f"{sum(1 for s_ in left if not s_[3])} adjacent")
        lib = root / "ws_unet_amd" / f"libwsu_{variant}.so"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, str(obj), "-o", str(lib)], check=True)
        print(f"[{variant}] {lib}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "scan":
        sys.exit(1 if scan_files(sys.argv[2:]) else 0)
    elif len(sys.argv) in (2, 3) and sys.argv[1] == "build":
        build_variants(Path(sys.argv[2]).resolve() if len(sys.argv) == 3 else ROOT)
    else:
        print(__doc__)
        sys.exit(2)
