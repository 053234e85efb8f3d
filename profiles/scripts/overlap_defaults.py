#!/usr/bin/env python3
"""How much of the default-options headline solve (2048 x 4096x256, automatic sub-batches on private streams) has a
trailing pass (k_qrx_pass*: the HBM-bound kernel) resident, and how much of the latency-bound kernels' time (k_qrx_pivot,
k_lmpar, k_dq_panel, the rest) lies under a pass of ANOTHER queue.  Input: a rocprofv3 --kernel-trace CSV of
profiles/scripts/defaults_one.py; the last solve of the trace is analysed (solves are separated by host gaps).

    python profiles/scripts/overlap_defaults.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap_len(s, e, un):
    # un: sorted disjoint intervals
    import bisect
    i = bisect.bisect_right(un, [s, float("inf")]) - 1
    i = max(i, 0)
    t = 0
    while i < len(un) and un[i][0] < e:
        t += max(0, min(e, un[i][1]) - max(s, un[i][0]))
        i += 1
    return t


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = list(rows[0].keys())
    sk = next(k for k in ks if k.lower().startswith("start"))
    ek = next(k for k in ks if k.lower().startswith("end"))
    qk = next((k for k in ks if k.lower() in ("queue_id", "stream_id")), None)
    nk = next(k for k in ks if k.lower() in ("kernel_name", "name"))
    ev = sorted((int(r[sk]), int(r[ek]), r[qk] if qk else "0", r[nk]) for r in rows if "k_gen" not in r[nk] and "k_dq_generate" not in r[nk])
    # the last solve: kernels after the last gap of more than 5 ms in which nothing runs
    cut, hi = 0, ev[0][1]
    for i, (s, e, q, n) in enumerate(ev):
        if s - hi > 5_000_000:
            cut = i
        hi = max(hi, e)
    ev = ev[cut:]
    t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)

    def cls(n):
        if "k_qrx_pass" in n:
            return "pass"
        if "k_qrx_pivot" in n:
            return "pivot"
        if "k_lmpar" in n:
            return "lmpar"
        if "k_dq_panel" in n:
            return "dq_panel"
        return "other"
    queues = sorted({q for _, _, q, _ in ev})
    pass_by_q = {q: union([(s, e) for s, e, qq, n in ev if qq == q and cls(n) == "pass"]) for q in queues}
    pass_all = union([(s, e) for s, e, q, n in ev if cls(n) == "pass"])
    any_all = union([(s, e) for s, e, q, n in ev])
    span = t1 - t0
    out = {"span_ms": span / 1e6, "queues": len(queues), "kernels": len(ev),
           "pass_resident_frac_of_wall": sum(e - s for s, e in pass_all) / span,
           "any_kernel_resident_frac_of_wall": sum(e - s for s, e in any_all) / span, "classes": {}}
    for c in ("pass", "pivot", "lmpar", "dq_panel", "other"):
        tot = under_other = under_any = 0
        for s, e, q, n in ev:
            if cls(n) != c:
                continue
            tot += e - s
            others = union([iv for qq in queues if qq != q for iv in pass_by_q[qq]])
            under_other += overlap_len(s, e, others)
            under_any += overlap_len(s, e, pass_all) if c != "pass" else 0
        out["classes"][c] = {"kernel_time_ms": tot / 1e6, "frac_of_wall": tot / span,
                             "frac_under_a_pass_of_another_queue": (under_other / tot) if tot else None}
    nonpass = sum(v["kernel_time_ms"] for k, v in out["classes"].items() if k != "pass")
    hidden = sum(v["kernel_time_ms"] * (v["frac_under_a_pass_of_another_queue"] or 0) for k, v in out["classes"].items() if k != "pass")
    out["non_pass_kernel_time_ms"] = nonpass
    out["non_pass_time_hidden_under_another_queues_pass_frac"] = hidden / nonpass if nonpass else None
    js = json.dumps(out, indent=1)
    print(js)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js + "\n")


if __name__ == "__main__":
    main()
