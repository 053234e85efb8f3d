"""Soak run (not collected by pytest; run it on a GPU box:  SOAK_SECONDS=600 SOAK_SEED=23 python tests/soak_exact_policy.py).

tests/soak_cases.py's replicated-batch sweep for as long as SOAK_SECONDS allows (its first 200 cases with seed 11 are the
collected test test_exact_policy_soak_slice).  Round 2: 551 cases found the slot-map race DESIGN.md section 2 describes;
3,159 cases after the fix (four seeds): no mismatch."""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as oracle
import soak_cases
oracle.lib()
ds = DeviceSolver(0)
sizes = [int(v) for v in os.environ["SOAK_SIZES"].split(",")] if os.environ.get("SOAK_SIZES") else soak_cases.SIZES
rng = random.Random(int(os.environ.get("SOAK_SEED", "11")))
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "300"))
case = 0; bad = 0
while time.time() < t_end:
    what, miss = soak_cases.run_case(ds, oracle, rng, sizes)
    if miss:
        bad += 1
        print("MISMATCH", dict(miss, case=case), flush=True)
    case += 1
print("soak:", case, "cases,", bad, "mismatches", flush=True)
