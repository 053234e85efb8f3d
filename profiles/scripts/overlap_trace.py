#!/usr/bin/env python3
"""Do the kernels of two sub-batches (private streams) really run side by side in the mid regime?  Reads a rocprofv3
--kernel-trace CSV and prints, per queue, the busy time, and the time during which kernels of more than one queue are in
flight.   python profiles/scripts/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [k for k in rows[0].keys()]
sk = next(k for k in ks if k.lower().startswith("start"))
ek = next(k for k in ks if k.lower().startswith("end"))
qk = next((k for k in ks if k.lower() in ("queue_id", "stream_id")), None)
nk = next(k for k in ks if k.lower() in ("kernel_name", "name"))
ev = []
perq = {}
names = {}
for r in rows:
    if "k_gen" in r[nk]:
        continue
    s, e, q = int(r[sk]), int(r[ek]), r[qk] if qk else "0"
    ev.append((s, 1, q)); ev.append((e, -1, q))
    perq[q] = perq.get(q, 0) + (e - s)
    key = (q, r[nk].split("(")[0].replace("void ", "")[:40])
    a = names.setdefault(key, [0, 0]); a[0] += 1; a[1] += e - s
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth = {}
last = t0
hist = {}
for t, d, q in ev:
    nq = sum(1 for v in depth.values() if v > 0)
    hist[nq] = hist.get(nq, 0) + (t - last)
    last = t
    depth[q] = depth.get(q, 0) + d
print(f"span {(t1 - t0) / 1e6:.2f} ms; columns: {ks}")
for q, b in sorted(perq.items()):
    print(f"  queue {q}: kernels busy {b / 1e6:.2f} ms")
for nq, t in sorted(hist.items()):
    print(f"  {nq} queue(s) with a kernel in flight: {t / 1e6:.2f} ms")
for (q, n), (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"    q{q} {n:40s} {c:6d} calls avg {t / c / 1e3:8.1f} us")
