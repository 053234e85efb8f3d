"""Prototype 2: the same bit-exact emulation in DOUBLE arithmetic (no 64-bit shifts): quantise each term with the
add-a-big-constant trick, detect exact ties from the remainder, carry the two-state (even / odd) offsets as doubles."""
import numpy as np, struct, math
def bits(x): return struct.unpack("<Q", struct.pack("<d", x))[0]
def frombits(b): return struct.unpack("<d", struct.pack("<Q", b))[0]
def parity_of(c, inv2ulp):            # c: exact multiple of ulp, as a double
    z = c * inv2ulp
    return (z - math.floor(z)) != 0.0
def lane_fold(s0, d):
    b = bits(s0); es = b >> 52
    ulp = frombits((es - 52) << 52); hu = 0.5 * ulp; inv2ulp = 0.5 / ulp
    C = frombits((es << 52) | (1 << 51)); lim = 0.25 * frombits(es << 52)
    c0 = 0.0; c1 = 0.0; fine = True
    for x in d:
        x = float(x)
        if not (x <= lim): fine = False; continue
        t = C + x; qd = t - C; r = x - qd
        if abs(r) == hu:
            f = x - hu
            c0 = c0 + f + (ulp if parity_of(c0 + f, inv2ulp) else 0.0)
            c1 = c1 + f + (ulp if (not parity_of(c1 + f, inv2ulp)) else 0.0)
        else:
            c0 = c0 + qd; c1 = c1 + qd
    return c0, c1, fine
def chunk(s0, d, lanes=64):
    b = bits(s0); es = b >> 52
    ulp = frombits((es - 52) << 52); inv2ulp = 0.5 / ulp
    E = len(d) // lanes
    comps = [lane_fold(s0, d[l * E:(l + 1) * E]) for l in range(lanes)]
    if not all(c[2] for c in comps): return None
    # fold in order (tree, like the shuffles)
    cur = [(c[0], c[1]) for c in comps]
    off = 1
    while off < lanes:
        nxt = list(cur)
        for i in range(lanes):
            j = i + off
            if j < lanes:
                l0, l1 = cur[i]; r0, r1 = cur[j]
                n0 = l0 + (r1 if parity_of(l0, inv2ulp) else r0)
                n1 = l1 + (r1 if (not parity_of(l1, inv2ulp)) else r0)
                nxt[i] = (n0, n1)
        cur = nxt; off *= 2
    tot = cur[0][1] if (b & 1) else cur[0][0]
    out = s0 + tot
    if (bits(out) >> 52) != es: return None
    return out
def chunk_fp(s0, d):
    s = np.float64(s0)
    for x in d: s = s + np.float64(x)
    return float(s)
rng = np.random.default_rng(2)
tot = used = bad = 0
for trial in range(1500):
    n = 48 * 64; kind = trial % 4
    if kind == 0: d = rng.random(n) ** 2
    elif kind == 1: d = (rng.standard_normal(n) * 0.3).clip(-1, 1) ** 2
    elif kind == 2: d = np.round(rng.random(n) * 64) / 64.0
    else: d = rng.random(n) ** 2 * (rng.random(n) < 0.3)
    s0 = float(rng.uniform(4, 70000))
    if trial % 11 == 0: s0 = float(2.0 ** rng.integers(3, 16)) * (1 - 2.0 ** -30)
    if trial % 13 == 0: s0 = float(2.0 ** rng.integers(3, 16)) + 3 * 2.0 ** -40
    r = chunk(s0, d); ref = chunk_fp(s0, d); tot += 1
    if r is not None:
        used += 1
        if bits(r) != bits(ref): bad += 1; print("MISMATCH", trial, s0, r, ref)
print("chunks", tot, "used", used, "mismatches", bad)
