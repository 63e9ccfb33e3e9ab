"""Discrete model of the fused decoder entry's pipeline (csrc/conv3x3_qu.hip): allocations (bytes, first step, last step) issued in ring order by a loader
that streams at R GB/s per CU, landing `lat` microseconds later; a step starts when its predecessor is done and its allocations have landed.  Used in round 4
to decide whether a different step order / ring could remove the distance to the matrix-only time (answer: no -- for any R >= 50 GB/s the schedule is within
1 % of its ideal).  python tools/sim_fused_pipeline.py"""
# generic: allocation list with (size, first, last, lat_kind); slot ring with nslot; loader streams at R
def run(allocs, steps, R, latW, latI, nslot=5):
    nst = len(steps); need = [[] for _ in range(nst)]
    for k, (sz, f, l, kind) in enumerate(allocs): need[f].append(k)
    landed = [None] * len(allocs); free_time = [None] * len(allocs)
    t_loader = 0.0; k = 0; j = 0; t_mat = 0.0
    while j < nst:
        prog = False
        while k < len(allocs):
            sr = 0.0 if k < nslot else free_time[k - nslot]
            if sr is None: break
            sz, f, l, kind = allocs[k]
            start = max(t_loader, sr); t_loader = start + sz / (R * 1e3)
            landed[k] = t_loader + (latI if kind == 'I' else latW); k += 1; prog = True
        if all(landed[a] is not None for a in need[j]):
            start = max([t_mat] + [landed[a] for a in need[j]]); t_mat = start + steps[j]
            for a, al in enumerate(allocs):
                if al[2] == j: free_time[a] = t_mat
            j += 1; prog = True
        if not prog: return float("nan")
    return t_mat
IN_S, W_S, W_L, IN_L = 31936, 28672, 25600, 17376
def build(nS, nL, tS, tL, tiles, variant):
    steps = []; allocs = []
    # step indices: tile t: S steps then L half-steps
    T = nS + 2 * nL
    for t in range(tiles):
        b = t * T
        for c in range(nS):
            steps.append(tS); allocs.append((IN_S, b + c, b + c, 'I')); allocs.append((W_S, b + c, b + c, 'W'))
        for c in range(nL):
            j = b + nS + 2 * c
            steps += [tL, tL]
            if variant == 'base':
                allocs += [(W_L, j, j, 'W'), (IN_L, j, j + 1, 'I'), (W_L, j + 1, j + 1, 'W')]
            elif variant == 'ahead':      # input of chunk c+1 allocated during chunk c (the first chunk's input before the phase)
                if c == 0: allocs.append((IN_L, j, j + 1, 'I'))
                allocs.append((W_L, j, j, 'W'))
                if c + 1 < nL: allocs.append((IN_L, j + 2, j + 3, 'I'))
                allocs.append((W_L, j + 1, j + 1, 'W'))
            elif variant == 'in_first':   # all inputs of the low phase ordered earlier: IN(c+1) right after W_L0(c) AND first IN before last S weights
                if c == 0: allocs.append((IN_L, j, j + 1, 'I'))
                if c + 1 < nL: allocs.append((IN_L, j + 2, j + 3, 'I'))
                allocs.append((W_L, j, j, 'W'))
                allocs.append((W_L, j + 1, j + 1, 'W'))
    return allocs, steps
if __name__ == '__main__':
    for name, nS, nL, tS, tL, meas in (("d41", 4, 8, 3.27, 0.70, 30.4), ("d31", 8, 16, 2.98, 0.64, 55.4)):
        for (R, latW, latI) in ((35, 0.3, 0.3), (50, 0.3, 0.3), (60, 0.3, 0.3), (60, 0.3, 2.0), (60, 0.3, 3.0), (45, 0.3, 1.5), (60, 1.5, 1.5)):
            out = []
            for v in ('base', 'ahead', 'in_first'):
                for ns in (5, 6):
                    a, s = build(nS, nL, tS, tL, 6, v)
                    out.append("%s/%d %.1f" % (v, ns, run(a, s, R, latW, latI, ns) / 6))
            print(name, "ideal %.1f" % (nS * tS + 2 * nL * tL), "meas", meas, "R=%d latW=%.1f latI=%.1f:" % (R, latW, latI), " | ".join(out))
