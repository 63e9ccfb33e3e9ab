"""CPU-side lint of the device assembly of every translation unit of libwsu (VERDICT r02 weak #1 / ADVICE r02): the packed-f32
read-after-write sites that made the fused head of conv3x3_pl return wrong logits on ~25 % of the launches must not come back with a
future schedule.  `make -C ws_unet_amd/csrc isa` emits the gfx950 assembly with each TU's own flags (hipcc cross-compiles without a GPU);
tools/pk_hazard.py defines the pattern.  Cause and hardware evidence: profiles/r03/pk_hazard.md."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def isa_files():
    csrc = ROOT / "ws_unet_amd" / "csrc"
    r = subprocess.run(["make", "-C", str(csrc), "isa", "-j8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted((csrc / "isa").glob("*.s"))
    assert len(files) >= 11, files
    return files


def test_no_packed_f32_producer_without_an_issue_cycle_before_its_consumer(isa_files):
    """Every TU: no packed-f32 VALU result is read by the NEXT vector instruction -- neither adjacent (the form that failed on hardware:
    `v_pk_add_f32 ... op_sel_hi:[0,1]` -> `v_max_f32`, which hipcc does not pad) nor separated only by an `s_waitcnt` (which hipcc counts
    as the wait state but which may retire without an issue cycle)."""
    import pk_hazard
    bad = []
    for f in isa_files:
        for kern, ln, prod, between, cons in pk_hazard.find_sites(f.read_text()):
            bad.append(f"{f.name}:{ln} {kern}: {prod} | {' ; '.join(between) or '(adjacent)'} | {cons}")
    assert not bad, "packed-f32 RAW pair without an issue cycle in between:\n" + "\n".join(bad)


def test_every_translation_unit_is_built_without_the_slp_vectorizer():
    mk = (ROOT / "ws_unet_amd" / "csrc" / "Makefile").read_text()
    flags = [ln for ln in mk.splitlines() if ln.startswith("CXXFLAGS")]
    assert len(flags) == 1 and "-fno-slp-vectorize" in flags[0]


def test_lint_recognises_the_round2_sites():
    """The scanner on a minimal excerpt of the SLP build of conv3x3_pl_kernel<1,...> (the failing schedule) and on its padded forms."""
    import pk_hazard
    site = """
kern_a:
	ds_read_b128 v[68:71], v153
	v_pk_fma_f32 v[88:89], v[74:75], v[64:65], 0 op_sel_hi:[1,0,0]
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[88:89], v[72:73], v[68:69], v[88:89] op_sel_hi:[1,0,1]
"""
    found = pk_hazard.find_sites(site)
    assert len(found) == 1 and found[0][0] == "kern_a" and found[0][3] == ["s_waitcnt lgkmcnt(0)"]
    assert not pk_hazard.find_sites(site.replace("s_waitcnt lgkmcnt(0)", "s_nop 0\n\ts_waitcnt lgkmcnt(0)"))
    assert not pk_hazard.find_sites(site.replace("s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v90, v67"))
    assert not pk_hazard.find_sites(site.replace("v[88:89] op_sel_hi:[1,0,1]", "v[90:91] op_sel_hi:[1,0,1]"))       # no dependence
    patched, n = pk_hazard.patch(site)
    assert n == 1 and not pk_hazard.find_sites(patched) and patched.count("s_nop 0") == 1
    # the form that failed on hardware: an unpadded packed add, its low dword read by the next instruction
    adj = "kern_b:\n\tv_pk_add_f32 v[72:73], v[64:65], v[72:73] op_sel_hi:[0,1]\n\tv_max_f32_e32 v76, 0, v72\n"
    found = pk_hazard.find_sites(adj)
    assert len(found) == 1 and found[0][3] == [] and not pk_hazard.compiler_pads(found[0][2])
    assert pk_hazard.patch(adj, "waitcnt")[1] == 0 and pk_hazard.patch(adj, "adjacent")[1] == 1
