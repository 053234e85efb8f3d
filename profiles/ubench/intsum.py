"""Prototype: bit-exact emulation of s = fl(fl(s0 + d0) + d1) ... for POSITIVE terms by integer arithmetic within the binade
of s0, with the two-state tie transducer; falls back (returns None) on a binade crossing or a term above s's binade."""
import numpy as np, struct, random

def bits(x): return struct.unpack("<Q", struct.pack("<d", x))[0]
def frombits(b): return struct.unpack("<d", struct.pack("<Q", b))[0]

def chunk_int(s0, d):
    b = bits(s0); es = b >> 52; S0 = (b & ((1 << 52) - 1)) | (1 << 52)
    assert 0 < es < 2047
    c = [0, 0]                       # offsets for even / odd input parity
    for x in d:
        xb = bits(float(x))
        if xb == 0: continue
        ed = xb >> 52
        if ed == 0: return None       # subnormal term: fall back
        mant = (xb & ((1 << 52) - 1)) | (1 << 52)
        k = es - ed
        if k < 0: return None         # term in a higher binade than s
        kk = min(k, 63)
        q = mant >> kk
        rem = mant & ((1 << kk) - 1) if kk else 0
        half = (1 << (kk - 1)) if kk else 0
        gt = kk > 0 and rem > half
        tie = kk > 0 and rem == half
        for bpar in (0, 1):
            cur = c[bpar]
            if tie: cur = cur + q + ((bpar + cur + q) & 1)
            else: cur = cur + q + (1 if gt else 0)
            c[bpar] = cur
    S = S0 + c[S0 & 1]
    if S >= (1 << 53): return None    # crossed into the next binade: fall back
    return frombits((es << 52) | (S & ((1 << 52) - 1)))

def chunk_fp(s0, d):
    s = np.float64(s0)
    for x in d: s = s + np.float64(x)
    return float(s)

rng = np.random.default_rng(1)
tot = 0; used = 0; bad = 0
for trial in range(3000):
    n = 48 * 64
    scale = 10.0 ** rng.uniform(-3, 3)
    kind = trial % 4
    if kind == 0: d = rng.random(n) ** 2
    elif kind == 1: d = (rng.standard_normal(n) * 0.3) ** 2
    elif kind == 2: d = np.round(rng.random(n) * 64) / 64.0           # many exact ties
    else: d = rng.random(n) ** 2 * (rng.random(n) < 0.3)             # zeros
    s0 = float(rng.uniform(200, 70000) * (1 if trial % 7 else 1.0))
    if trial % 11 == 0: s0 = float(2.0 ** rng.integers(8, 16)) * (1 - 2.0 ** -30)   # just below a binade boundary
    r = chunk_int(s0, d)
    ref = chunk_fp(s0, d)
    tot += 1
    if r is not None:
        used += 1
        if bits(r) != bits(ref): bad += 1; print("MISMATCH", trial, s0, r, ref)
print("chunks", tot, "integer path used", used, "mismatches", bad)
