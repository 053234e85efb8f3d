#!/usr/bin/env python3
"""Reduce the rocprofv3 outputs of profiles/capture_r06.sh to the files committed under profiles/:

  r06_<cfg>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary (bench.py --steps 2 --warmup 1)
  r06_<cfg>_bench_under_rocprof.json  the JSON line bench.py printed in that same run (its roofline.avg_launch_ms is the
                                    live HIP-event figure; the stats CSV's average for the same kernel must agree)
  r06_pmc_summary.json              per config and kernel family: launches, sum of FETCH_SIZE / WRITE_SIZE over the
                                    launches of ONE bench step, algorithmic bytes of those launches, and their ratio

Counter units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced streaming read, i.e. half the bytes; that correction
(read bytes = 2 x FETCH_SIZE x 1024) is calibrated for 16-byte-per-lane loads.  k_qrx_pass reads 8 bytes per lane
(512 contiguous, 64-byte-aligned bytes per wave instruction); both readings are recorded (`hbm_read_bytes_x1`, `_x2`):
the launches cannot move fewer bytes than the algorithmic ones (the working set is far larger than every cache), which
is what decides between them.

usage: python3 profiles/summarize_r06.py <dir with the capture>   (writes into that directory)
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

CFG = {"c2": (4096, 256, 2048, 2), "c4": (2048, 128, 1024, 2), "c5": (65536, 512, 1, 2),
       "c5auto": (65536, 512, 1, 0), "c2auto": (4096, 256, 2048, 0)}


def family(name):
    name = name.strip('"')
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    fam = m.group(1) if m else name
    return "k_qrx_pass" if fam in ("k_qrx_pass_rp", "k_qrx_pass_rpw", "k_qrx_pass_col") else fam      # the forms of the trailing pass are one roofline kernel


def qr_pass_bytes(m, n):
    return 8.0 * sum((m - j) * (n - j) for j in range(n))


def main():
    out_dir = sys.argv[1]
    summary = {"generated_by": "profiles/summarize_r06.py", "configs": {}}
    # the GPU box has no git: the build container writes the commit the capture was taken at into profiles/.capture_commit
    # before the run (profiles/capture_r06.sh documents the step)
    cfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".capture_commit")
    summary["commit"] = open(cfile).read().strip() if os.path.exists(cfile) else "unknown (profiles/.capture_commit missing)"
    for cfg, (m, n, nprob, policy) in CFG.items():
        st = glob.glob(os.path.join(out_dir, f"{cfg}.stats", "**", "*kernel_stats.csv"), recursive=True)
        if st:
            rows = list(csv.reader(open(st[0])))
            keep = [rows[0]] + [r for r in rows[1:] if float(r[4]) >= 0.05]        # >= 0.05 % of GPU time
            with open(os.path.join(out_dir, f"r06_{cfg}_kernel_stats.csv"), "w", newline="") as fh:
                csv.writer(fh).writerows(keep)
        entry = {"m": m, "n": n, "problems": nprob, "policy": policy, "kernels": {}}
        bench = None
        bj = os.path.join(out_dir, f"r06_{cfg}_bench_under_rocprof.json")
        if os.path.exists(bj) and os.path.getsize(bj):
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
        per = defaultdict(lambda: defaultdict(list))
        for grp in ("FETCH_SIZE", "WRITE_SIZE"):
            cc = glob.glob(os.path.join(out_dir, f"{cfg}.{grp}", "**", "*counter_collection.csv"), recursive=True)
            if not cc:
                continue
            for r in csv.DictReader(open(cc[0])):
                per[family(r["Kernel_Name"])][grp].append(float(r["Counter_Value"]))
        for fam, d in sorted(per.items()):
            f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
            if not f or sum(f) + sum(w) < 1024.0:                # < 1 MiB in the whole step: not a data kernel
                continue
            k = {"launches": len(f), "FETCH_SIZE_KiB_sum": sum(f), "WRITE_SIZE_KiB_sum": sum(w),
                 "hbm_read_bytes_x1": sum(f) * 1024.0, "hbm_read_bytes_x2": 2.0 * sum(f) * 1024.0,
                 "hbm_write_bytes": sum(w) * 1024.0}
            entry["kernels"][fam] = k
        # algorithmic bytes of the roofline kernel over the same single step: one factorisation per counted Jacobian
        if bench is not None:
            entry["bench_line"] = {k: bench[k] for k in ("value", "ms_per_step") if k in bench}
            entry["bench_line"]["roofline"] = bench.get("roofline")
        summary["configs"][cfg] = entry
    # the bench's roofline kernel at the headline config: bytes moved per algorithmic byte.  The PMC runs are
    # `--steps 1 --warmup 0`: njac of that single step comes from the run's own JSON line.
    c2 = summary["configs"].get("c2")
    if c2 and "k_qrx_pass" in c2["kernels"]:
        m, n, nprob, policy = CFG["c2"]
        njac = None
        log = os.path.join(out_dir, "c2.FETCH_SIZE.log")
        for line in open(log):
            if line.startswith('{"metric"'):
                d = json.loads(line)
                njac = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]
        k = c2["kernels"]["k_qrx_pass"]
        if njac:
            alg = qr_pass_bytes(m, n) * njac
            k["algorithmic_bytes"] = alg
            k["problem_factorisations"] = njac
            r1 = (k["hbm_read_bytes_x1"] + k["hbm_write_bytes"]) / alg
            r2 = (k["hbm_read_bytes_x2"] + k["hbm_write_bytes"]) / alg
            k["bytes_per_algorithmic_byte_x1"] = r1
            k["bytes_per_algorithmic_byte_x2"] = r2
            # fewer bytes than the algorithmic ones cannot have moved: that picks the reading of FETCH_SIZE (the
            # calibration kernel below settles it independently: 0.500 for this access pattern -> x2)
            pick = r1 if k["hbm_read_bytes_x1"] >= 0.98 * alg else r2
            summary.update({"m": m, "n": n, "policy": policy, "problems": nprob,
                            "roofline_kernel": {"kernel": "k_qrx_pass", "hbm_bytes_per_algorithmic_byte": pick,
                                                "fetch_size_reading": "x1" if pick == r1 else "x2"}})
    # calibration run (profiles/ubench/fetch_calib.hip): FETCH_SIZE x 1024 / bytes actually read, 8 and 16 bytes per lane
    cal = glob.glob(os.path.join(out_dir, "calib", "**", "*counter_collection.csv"), recursive=True)
    if cal:
        nbytes = None
        for line in open(os.path.join(out_dir, "calib.log")):
            if line.startswith("bytes_read_per_kernel"):
                nbytes = float(line.split()[1])
        c = {}
        for r in csv.DictReader(open(cal[0])):
            if r["Kernel_Name"].startswith("k_read") and nbytes:
                c[family(r["Kernel_Name"])] = float(r["Counter_Value"]) * 1024.0 / nbytes
        summary["fetch_size_calibration"] = {"bytes_read_per_kernel": nbytes, "FETCH_SIZE_bytes_per_byte_read": c,
                                             "source": "profiles/ubench/fetch_calib.hip (k_read8: the access pattern of k_qrx_pass)"}
    with open(os.path.join(out_dir, "r06_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("wrote r06_pmc_summary.json:", json.dumps(summary.get("roofline_kernel")))


if __name__ == "__main__":
    main()
