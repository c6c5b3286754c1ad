#!/usr/bin/env python3
"""Per-wave life times of the force-only nonbonded kernel on the bench workload: how much of the launch is tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mdx
from mdx import synth, engine as eng
s = synth.pseudo_complex(d=12.5)
tc = ["Protein", "Non-Protein"]
e = eng.Engine(s, rc=1.0, dt=0.0005, nstlist=100, nstprune=10, nstcalcenergy=0, tc_groups=tc, ref_t=[300, 300], tau_t=[0.02, 0.02], nsttcouple=2)
e.step(515); e.synchronize()
print("rlist", e.rlist, e.list_stats())
for dyn, sp in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (3, 2)):
    e.debug_set(3, dyn); e.debug_set(4, sp)
    print("dyn %d split %d: alone %.1f us (force)  " % (dyn, sp, e.debug_time_nonbonded(False, 30) * 1e3), end="")
    print("%.1f us (energy)" % (e.debug_time_nonbonded(True, 30) * 1e3))
for rep, (dyn, sp) in enumerate(((1, 2), (2, 2))):
    e.debug_set(3, dyn); e.debug_set(4, sp)
    t = e.debug_wave_times()
    c0, c1, w0, w1, nu, ne, hw, xcc, ntest, nev = t.T
    print("tiles tested %d, evaluated %d (%.1f %%), entries %d" % (ntest.sum(), nev.sum(), 100.0 * nev.sum() / max(ntest.sum(), 1), ne.sum()))
    span_c = float(c1.max() - c0.min()); span_w = (w1.max() - w0[w0 > 0].min()) / 100.0
    dur = (c1 - c0)
    print("launch span %.1f us (wall clock), %d shader cycles -> %.2f GHz" % (span_w, span_c, span_c / span_w / 1e3))
    print("wave life: mean %.0f  min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f cycles; mean/span %.3f" % (
        dur.mean(), dur.min(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), dur.max(), dur.mean() / span_c))
    st = c0 - c0.min(); en = c1 - c0.min()
    print("start: p50 %.0f p90 %.0f max %.0f ; end: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (
        np.percentile(st, 50), np.percentile(st, 90), st.max(), en.min(), np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max()))
    print("units/wave min %d max %d ; entries/wave min %d mean %.1f max %d ; cycles/entry %.1f" % (nu.min(), nu.max(), ne.min(), ne.mean(), ne.max(), dur.sum() / ne.sum()))
    xcc = xcc & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ws = (w0 - w0.min()) / 100.0; we = (w1 - w0.min()) / 100.0
    print("wall: start p50 %.1f max %.1f us; end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us; mean life %.1f us" % (
        np.percentile(ws, 50), ws.max(), we.min(), np.percentile(we, 10), np.percentile(we, 50), np.percentile(we, 90), we.max(), (we - ws).mean()))
    cus = np.unique(key)
    print("distinct CUs %d, waves per CU min %d max %d" % (len(cus), min((key == k).sum() for k in cus), max((key == k).sum() for k in cus)))
    cu_end = np.array([we[key == k].max() for k in cus]); cu_first = np.array([we[key == k].min() for k in cus])
    print("per-CU last-wave end: min %.1f p50 %.1f max %.1f ; per-CU first-wave end p50 %.1f" % (cu_end.min(), np.percentile(cu_end, 50), cu_end.max(), np.percentile(cu_first, 50)))
    skey = key * 4 + simd
    sk = np.unique(skey)
    s_end = np.array([we[skey == k].max() for k in sk]); s_n = np.array([(skey == k).sum() for k in sk])
    s_ent = np.array([ne[skey == k].sum() for k in sk])
    print("SIMDs %d, waves/SIMD min %d max %d; per-SIMD end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; corr(end, entries) %.2f" % (
        len(sk), s_n.min(), s_n.max(), s_end.min(), np.percentile(s_end, 10), np.percentile(s_end, 50), np.percentile(s_end, 90), s_end.max(), np.corrcoef(s_end, s_ent)[0, 1]))
    for x in range(8):
        m = xcc == x
        if m.any(): print("  xcc %d: waves %d mean end %.1f max end %.1f mean life %.1f" % (x, m.sum(), we[m].mean(), we[m].max(), (we - ws)[m].mean()))
    if rep == 0: np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "wavetimes.npy"), t)
    # when do the waves end, in microseconds of the 100 MHz clock from the first wave's start; units and entries per wave
    en_w = (w1 - w0[w0 > 0].min()) / 100.0
    print("wave ends (us): min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; units per wave min %d max %d; entries per wave min %d p50 %d max %d" % (
        en_w.min(), np.percentile(en_w, 10), np.percentile(en_w, 50), np.percentile(en_w, 90), np.percentile(en_w, 99), en_w.max(), nu.min(), nu.max(), ne.min(), np.median(ne), ne.max()))
    # per SIMD (HW_ID: wave slot, SIMD, CU, SH, SE; XCC_ID): how far apart do the three waves of a SIMD end
    key = (hw >> 4) & 0xFFFFF
    key = key * 16 + xcc
    import collections
    groups = collections.defaultdict(list)
    for k, t in zip(key, en_w): groups[int(k)].append(float(t))
    sp = np.array([max(v) - min(v) for v in groups.values() if len(v) > 1])
    last = np.array([max(v) for v in groups.values()])
    print("SIMD groups %d: spread of wave ends inside a SIMD mean %.1f max %.1f us; last wave of a SIMD ends: min %.1f p50 %.1f max %.1f us" % (
        len(groups), sp.mean() if len(sp) else 0, sp.max() if len(sp) else 0, last.min(), np.median(last), last.max()))
