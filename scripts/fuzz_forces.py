"""Randomised parity sweep of the whole coupled step, aimed at the link-sum table route of the hydrodynamic forces
(cases: tests/fuzz_util.py): every bit of f, obst, fhf and the grain kinematics against the CPU oracle; reports how
many grains took the table and how many the gather queue.
   python scripts/fuzz_forces.py [ncases] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as ge, fuzz_util
pkg = ge.load_package(); po = ge.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
bad = 0
for case in range(ncases):
    try:
        desc, ok, tab, gat, anomalies = fuzz_util.run_case(pkg, po, seed0 + case)
    except pkg.LbmDemError as e:
        print(f"case {case}: library error: {e}"); bad += 1; continue
    if ok is None:
        print(f"case {case}: skipped ({desc})"); continue
    print(f"case {case}: {desc}: {'bit-equal' if ok else 'DIFFERENT'}; grain-steps from table {tab}, gathered {gat}; "
          f"act anomalies {anomalies}", flush=True)
    bad += 0 if ok else 1
print(f"{'FUZZ OK' if bad == 0 else 'FUZZ FAILED: %d cases' % bad} [{time.time() - t0:.0f} s]")
sys.exit(1 if bad else 0)
