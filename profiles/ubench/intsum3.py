"""Prototype 3: SIGNED terms.  As intsum2.py, plus: the running sum must stay inside the binade of s0 at EVERY prefix.  A lane
tracks the extremes of its own running offset (for the even-entry state; the odd-entry state differs by at most one ulp per
tie, so the bounds are widened by (ties + 1) ulp), the fold is a SCAN so every lane knows the exact sum at its start; the
chunk is accepted iff start + min >= 2^e and start + max < 2^(e+1) in every lane.  Negative s0: mirrored."""
import numpy as np, struct, math
def bits(x): return struct.unpack("<Q", struct.pack("<d", x))[0]
def frombits(b): return struct.unpack("<d", struct.pack("<Q", b))[0]
def odd(c, inv2ulp):
    z = c * inv2ulp
    return (z - math.floor(z)) != 0.0
def chunk(s0, d, lanes=64):
    neg = s0 < 0
    b = bits(abs(s0)); es = b >> 52
    if es < 60 or es > 2000: return None
    ulp = frombits((es - 52) << 52); hu = 0.5 * ulp; inv2ulp = 0.5 / ulp
    C = frombits((es << 52) | (1 << 51)); lo = frombits(es << 52); hi = 2.0 * lo; lim = 0.25 * lo
    E = len(d) // lanes
    comps = []
    for l in range(lanes):
        c0 = c1 = 0.0; mn = mx = 0.0; nt = 0; fine = True
        for x in d[l * E:(l + 1) * E]:
            x = float(x)
            if not (abs(x) <= lim): fine = False; continue
            t = C + x; q = t - C; r = x - q
            if abs(r) == hu:
                f = x - hu; nt += 1
                c0 = c0 + f + (ulp if odd(c0 + f, inv2ulp) else 0.0)
                c1 = c1 + f + (ulp if (not odd(c1 + f, inv2ulp)) else 0.0)
            else:
                c0 = c0 + q; c1 = c1 + q
            mn = min(mn, c0); mx = max(mx, c0)
        comps.append((c0, c1, mn - (nt + 1) * ulp, mx + (nt + 1) * ulp, fine))
    if not all(c[4] for c in comps): return None
    # exclusive scan of the composites in lane order
    p0 = bool(b & 1)
    start = 0.0; par = p0          # offset from s0 at the lane's start, parity of the running sum there
    for l in range(lanes):
        c0, c1, mn, mx, _ = comps[l]
        a, bb = s0 + (start + mn), s0 + (start + mx)
        if neg:
            if not (bb <= -lo and a > -hi): return None
        else:
            if not (a >= lo and bb < hi): return None
        c = c1 if par else c0
        par = par ^ odd(c, inv2ulp)
        start = start + c
    out = s0 + start
    if not (lo <= abs(out) < hi): return None
    return out
def chunk_fp(s0, d):
    s = np.float64(s0)
    for x in d: s = s + np.float64(x)
    return float(s)
rng = np.random.default_rng(3)
tot = used = bad = 0
for trial in range(2000):
    n = 64 * 64; kind = trial % 4
    if kind == 0: d = rng.standard_normal(n)
    elif kind == 1: d = rng.standard_normal(n) * rng.random(n)
    elif kind == 2: d = np.round(rng.standard_normal(n) * 32) / 32.0        # exact ties
    else: d = rng.standard_normal(n) * (rng.random(n) < 0.3)
    s0 = float(rng.uniform(30, 3000)) * (1 if trial % 3 else -1)
    if trial % 11 == 0: s0 = float(2.0 ** rng.integers(5, 11)) * (1 + 2.0 ** -20)
    r = chunk(s0, list(d)); ref = chunk_fp(s0, d); tot += 1
    if r is not None:
        used += 1
        if bits(r) != bits(ref): bad += 1; print("MISMATCH", trial, s0, r, ref)
print("chunks", tot, "used", used, "mismatches", bad)
