#!/usr/bin/env python3
"""The nonbonded kernel on a single list against the same kernel on pruned lists of the same state: units, entries, tiles
tested / evaluated (per-wave counters of the timed kernel variant) and the launch time alone.
usage: dual_list_probe.py [nstlist:nstprune[:chunk] ...]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
cfgs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(20, 0), (100, 10), (100, 20)]
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(1500)
x, v, box = e.get_state(); e.close()
s1 = copy.copy(s); s1.x = x.astype(np.float64); s1.v = v.astype(np.float64); s1.box = box.astype(np.float64)
for cfg in cfgs:
    nstlist, nstprune = cfg[0], cfg[1]
    e = eng.Engine(s1, rc=1.0, dt=0.002, nstlist=nstlist, nstprune=nstprune, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.1, 0.1], nsttcouple=10)
    if len(cfg) > 2:
        e.debug_set(14, cfg[2])
    # stop right behind a pruning step (its list is the current one) or, single list, at the same age of the list
    e.step((nstprune or 10) + 1); e.synchronize()
    st = e.list_stats()
    t = e.debug_wave_times()
    c0, c1, w0, w1, nu, ne, hw, xcc, ntest, nev = t.T
    alone = e.debug_time_nonbonded(False, 40) * 1e3
    print("nstlist %3d nstprune %2d chunk %s rlist %.3f inner %s: outer units %d entries %d tiles %d | walked: units %d entries %d tiles tested %d evaluated %d | alone %.1f us, span %.1f us"
          % (nstlist, nstprune, cfg[2] if len(cfg) > 2 else "-", e.rlist, e.rlist_inner, st["work_units"], st["cj_entries"], st["cluster_pairs"],
             nu.sum(), ne.sum(), ntest.sum(), nev.sum(), alone, (w1.max() - w0[w0 > 0].min()) / 100.0))
    e.close()
