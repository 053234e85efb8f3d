#!/usr/bin/env python3
"""Build profiles/<tag>_pmc_summary.json from the rocprofv3 counter CSVs committed next to it.

Inputs (one rocprofv3 run per counter group, all of the same command
`python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --exact-sample 0`):
    <tag>_pmc_FETCH_SIZE_counter_collection.csv        --pmc FETCH_SIZE
    <tag>_pmc_WRITE_SIZE_counter_collection.csv        --pmc WRITE_SIZE
    <tag>_pmc_GRBM_GUI_ACTIVE_counter_collection.csv   --pmc GRBM_GUI_ACTIVE
    <tag>_pmc_SQ_counter_collection.csv                --pmc SQ_* (MFMA busy, VALU issue)

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024.
The largest launch of a kernel is the one with every problem of the batch still active, so
per-problem traffic = bytes of that launch / problems per launch.

usage: python profiles/make_summary.py r01 [problems_per_launch m n]   (problems_per_launch = the bench batch, 512 by default)
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
N_SIMD = 1024          # 256 CUs x 4 SIMDs
N_XCD = 8              # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs


def short(name):
    name = name.strip('"')
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)\(", name)
    return m.group(1) if m else name


def rows(tag, group):
    path = os.path.join(HERE, f"{tag}_pmc_{group}_counter_collection.csv")
    if not os.path.exists(path):
        return []
    with open(path, newline="") as fh:
        return list(csv.DictReader(fh))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    kernels = defaultdict(dict)

    for group in ("FETCH_SIZE", "WRITE_SIZE"):
        per = defaultdict(list)
        for r in rows(tag, group):
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in per.items():
            kernels[k][f"{group}_KiB_max_per_launch"] = max(v)
            kernels[k][f"{group}_launches"] = len(v)

    # effective clock of the longest launch: GRBM_GUI_ACTIVE cycles / duration
    per = defaultdict(list)
    for r in rows(tag, "GRBM_GUI_ACTIVE"):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3     # us
        per[short(r["Kernel_Name"])].append((dur, float(r["Counter_Value"])))
    for k, v in per.items():
        dur, cyc = max(v)
        kernels[k]["longest_launch_us"] = round(dur, 3)
        kernels[k]["effective_clock_GHz_longest_launch"] = round(cyc / N_XCD / (dur * 1e3), 3)

    # SQ pass: group the counters of one dispatch, keep the longest dispatch of each kernel
    disp = defaultdict(dict)
    for r in rows(tag, "SQ"):
        key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
        disp[key][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[key]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    best = {}
    for (k, _), c in disp.items():
        if k not in best or c["_us"] > best[k]["_us"]:
            best[k] = c
    for k, c in best.items():
        d = kernels[k]
        d["sq_pass_duration_us"] = round(c["_us"], 3)
        clk = d.get("effective_clock_GHz_longest_launch")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            if clk and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
                d["mfma_utilisation"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["_us"] * 1e3 * clk * N_SIMD), 3)
        if "SQ_INSTS_VALU" in c:
            d["valu_wave_instructions"] = c["SQ_INSTS_VALU"]

    for k, d in kernels.items():
        if "FETCH_SIZE_KiB_max_per_launch" in d and "WRITE_SIZE_KiB_max_per_launch" in d:
            b = 2.0 * d["FETCH_SIZE_KiB_max_per_launch"] * 1024.0 + d["WRITE_SIZE_KiB_max_per_launch"] * 1024.0
            d["hbm_bytes_per_full_launch"] = b
            d["hbm_bytes_per_problem"] = b / nprob

    out = {
        "command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 1 "
                   "--warmup 0 --cpu-sample 0 --exact-sample 0 (one pass per counter group, "
                   f"{nprob} problems {m}x{n} per launch)",
        "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced "
                "read (MI355X_MICROARCH.md, HBM section), so read bytes = 2*FETCH_SIZE*1024.  MFMA utilisation = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (duration * effective clock * 1024 SIMDs), effective clock from the "
                "GRBM_GUI_ACTIVE pass.",
        "generated_by": "profiles/make_summary.py",
        "problems_per_launch": nprob, "m": m, "n": n,
        "kernels": dict(sorted(kernels.items())),
    }
    path = os.path.join(HERE, f"{tag}_pmc_summary.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, "with", len(kernels), "kernels")


if __name__ == "__main__":
    main()
