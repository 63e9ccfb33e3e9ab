"""Timing-only build of the forward conv with the cross terms in the fp4 CONFIGURATION (results are wrong): one 16-byte granule per operand and
cross-term MFMA (fp4 format code), no e4m3-copy weight plane in the DMA (27 instead of 36 KB per step), no derived input plane -- the best case of
"block-scaled fp4 cross terms whose derivation costs nothing" in today's two-stage pipeline.  Patches a COPY of conv3x3_pl.hip:
    python tools/probe6_build.py   ->  ws_unet_amd/libwsu_plprobe6.so      (then: python tools/probe_units_pl.py libwsu_plprobe6.so)"""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "ws_unet_amd" / "csrc"
s = (CSRC / "conv3x3_pl.hip").read_text()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)


# 1. cross terms: one granule per operand, fp4 format (4x the bf16 rate)
rep('''                for (int q = 0; q < 2; ++q) wsu_mfma_f8x2(a0[m], a1[m], b0[q], b1[q], sc_a, sc_b, acc[m][q]);
        };''', '''                for (int q = 0; q < 2; ++q) {
                    i32x8 av = {(int)a0[m].x, (int)a0[m].y, (int)a0[m].z, (int)a0[m].w, 0, 0, 0, 0}, bv = {(int)b0[q].x, (int)b0[q].y, (int)b0[q].z, (int)b0[q].w, 0, 0, 0, 0};
                    acc[m][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[m][q], 4, 4, 0, sc_a, 0, sc_b);
                }
        };''')
rep('''                a1[m] = *reinterpret_cast<const u32x4*>(ldsA + aoff + 64 * 16 + m * 32 * 16);''', '''                a1[m] = a0[m];''')
rep('''                b1[q] = *reinterpret_cast<const u32x4*>(ldsB + boff + PLANE + q * IW * 16);''', '''                b1[q] = b0[q];''')
# 2. no weight pieces of granule plane 3
rep('''    } else {
        WSU_STATIC_FOR(W_PER_WAVE, k, {
            const int wslot = (WEIGHTS_ONLY ? lw_rt : LW) + NLOAD * k;''', '''    } else if constexpr (LW != 3) {
        WSU_STATIC_FOR(W_PER_WAVE, k, {
            const int wslot = (WEIGHTS_ONLY ? lw_rt : LW) + NLOAD * k;''')
# 3. no derivation (and no separate wait for the input pieces)
rep('''        if constexpr (!HONLY) {                                           // (f16 products only: LDS plane 3 is not used, nothing to derive)''', '''        if constexpr (false) {''')
(CSRC / "conv3x3_pl_probe6.hip").write_text(s)
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -Wno-unused-variable -Wno-unused-but-set-variable".split()
subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", "conv3x3_pl_probe6.hip", "-o", "conv3x3_pl_probe6.o"], cwd=CSRC, check=True)
import re
srcs = re.search(r"^SRCS\s*=\s*(.*)$", (CSRC / "Makefile").read_text(), re.M).group(1).split()
objs = [n[:-4] + ".o" for n in srcs if n != "conv3x3_pl.hip"]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "conv3x3_pl_probe6.o", "-o", "../libwsu_plprobe6.so"], cwd=CSRC, check=True)
(CSRC / "conv3x3_pl_probe6.hip").unlink()
print("built", ROOT / "ws_unet_amd" / "libwsu_plprobe6.so")
