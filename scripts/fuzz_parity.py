"""Randomised parity campaign (tests/fuzzcase.py) on the GPU box: random small stacks and parameters, HIP path against the oracle
stage by stage; every comparison is for equality of bytes.
usage: fuzz_parity.py [seconds] [first_case]     FUZZ_DRIVER=persistent  FUZZ_BIG=1 (stacks up to 192 x 160 x 80)"""
import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import orc, fuzzcase

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
case = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = orc.load_oracle()
t_end = time.time() + budget
nbad = ncases = 0
stats = {}
while time.time() < t_end:
    desc = {}
    try:
        fuzzcase.run_case(L, case, stats, desc, driver=os.environ.get("FUZZ_DRIVER"), big=bool(os.environ.get("FUZZ_BIG")))
    except Exception as e:  # noqa
        nbad += 1
        print("MISMATCH", desc, "->", repr(e)[:300], flush=True)
        if not isinstance(e, AssertionError): traceback.print_exc()
    ncases += 1; case += 1
    if ncases % 10 == 0: print(f"[{ncases} cases, {nbad} bad] {stats}", flush=True)
print(f"done: {ncases} cases, {nbad} mismatches, {stats}, next case {case}")
sys.exit(1 if nbad else 0)
