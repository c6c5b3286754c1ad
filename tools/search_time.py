"""Average time of one pair search on the bench system, binning the atoms in input order and in the previous cluster order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx  # noqa: F401,E402
from mdx import synth, engine as eng  # noqa: E402

s = synth.pseudo_complex(d=12.5)
e = eng.Engine(s, rc=1.0, dt=0.002, nstlist=20, nstcalcenergy=10, tc_groups=["Protein", "Non-Protein"], ref_t=[300.0] * 2,
               tau_t=[0.1] * 2, nsttcouple=10)
e.step(200); e.synchronize()
for rep in range(2):
    for bit, name in ((0x8000000, "input order"), (0, "previous cluster order")):
        e.debug_set(2, bit)
        print("%-24s %.1f us" % (name, 1e3 * e.debug_time_stage(1, 30)), flush=True)
e.debug_set(2, 0)
e.step(100); e.synchronize()
print("ok", e.energies()["TEMP"])
e.close()
