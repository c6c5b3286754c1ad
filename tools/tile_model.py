"""Offline model of the nonbonded kernel's tile efficiency (CPU only, numpy + scipy).

A TIP3P box of the bench workload's density is clustered the way csrc/pairsearch.hip does it (xy columns, z sort,
super-clusters of 64 atoms, y sort of the 32-halves, x sort of the 16-quarters) and in the alternative ways listed in
VARIANTS; for every variant the script counts what the kernel would do per step:

  listed tiles      8x8 cluster pairs with an atom pair inside the pair-list radius (distance test only: ~7 VALU instructions)
  evaluated tiles   ... with an atom pair inside the cutoff (the tile body: ~32 VALU instructions, ~23 without the LJ part)
  lanes inside rc   pairs inside the cutoff / (64 x evaluated tiles)
  LJ-free tiles     evaluated tiles one of whose clusters holds no LJ atom at all (a wave-uniform skip of the LJ arithmetic)
  entries           (i-super-cluster, j-cluster) list entries (~45 VALU instructions each: fetch, reduction, atomic)

and prices them in wave64 VALU instructions per pair inside the cutoff x 64 (lane-instructions per pair), the figure
the SQ counters give for the real kernel. Usage: python tools/tile_model.py [--side 8.0] [--rlist 1.095] [--rc 1.0]
"""
import argparse
import sys
import os
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdx  # noqa: E402,F401
from mdx import synth  # noqa: E402

C_TEST, C_EVAL, C_EVAL_NOLJ, C_ENTRY = 7.0, 32.0, 23.0, 45.0


def columns(x, L, dens):
    side = (64.0 / dens) ** (1.0 / 3.0)
    nc = max(1, int(np.floor(L / side)))
    cx = np.minimum(nc - 1, (x[:, 0] * nc / L).astype(int)); cy = np.minimum(nc - 1, (x[:, 1] * nc / L).astype(int))
    return cy * nc + cx, nc * nc


def superclusters(x, L, dens):
    """-> list of index arrays (<= 64 atoms each), the engine's z-sorted chunks of each xy column."""
    col, ncol = columns(x, L, dens)
    order = np.lexsort((x[:, 2], col))
    out = []
    start = 0
    counts = np.bincount(col, minlength=ncol)
    for c in range(ncol):
        idx = order[start:start + counts[c]]; start += counts[c]
        for k in range(0, len(idx), 64):
            out.append(idx[k:k + 64])
    return out


def split_zyx(x, idx):
    """The engine's split of a super-cluster: sorted by z already; y sort of each 32-half, x sort of each 16-quarter."""
    idx = np.asarray(idx)
    res = []
    for h in range(0, len(idx), 32):
        half = idx[h:h + 32]
        half = half[np.argsort(x[half, 1], kind="stable")]
        for q in range(0, len(half), 16):
            quarter = half[q:q + 16]
            quarter = quarter[np.argsort(x[quarter, 0], kind="stable")]
            for c in range(0, len(quarter), 8):
                res.append(quarter[c:c + 8])
    return res


def kd_order(x, idx):
    """Recursive bisection along the axis of largest extent, in multiples of 8 atoms: a compact order for any subset."""
    idx = np.asarray(idx)
    if len(idx) <= 8:
        return idx
    ext = x[idx].max(0) - x[idx].min(0)
    ax = int(np.argmax(ext))
    s = idx[np.argsort(x[idx, ax], kind="stable")]
    ncl = (len(idx) + 7) // 8
    left = (ncl // 2) * 8
    return np.concatenate([kd_order(x, s[:left]), kd_order(x, s[left:])])


def split_kd(x, idx):
    o = kd_order(x, idx)
    return [o[c:c + 8] for c in range(0, len(o), 8)]


def split_type_first(x, idx, is_lj):
    """LJ atoms first (kd order), then the rest (kd order), cut into consecutive clusters of 8."""
    idx = np.asarray(idx)
    a = idx[is_lj[idx]]; b = idx[~is_lj[idx]]
    o = np.concatenate([kd_order(x, a) if len(a) else a, kd_order(x, b) if len(b) else b])
    return [o[c:c + 8] for c in range(0, len(o), 8)]


def build(x, L, dens, how, is_lj):
    scs = superclusters(x, L, dens)
    clusters, sc_of = [], []
    for si, idx in enumerate(scs):
        if how == "zyx":
            cl = split_zyx(x, idx)
        elif how == "kd":
            cl = split_kd(x, idx)
        elif how == "type":
            cl = split_type_first(x, idx, is_lj)
        else:
            raise ValueError(how)
        for c in cl:
            clusters.append(c); sc_of.append(si)
    ncl = len(clusters)
    member = np.full((ncl, 8), -1, dtype=np.int64)
    for k, c in enumerate(clusters):
        member[k, :len(c)] = c
    return member, np.asarray(sc_of)


def analyse(x, L, member, sc_of, is_lj, rc, rlist, jhalf=False):
    """Counts over all unordered cluster pairs (a cluster with itself included)."""
    valid = member >= 0
    pos = np.where(valid[:, :, None], x[np.where(valid, member, 0)], np.nan)
    ctr = np.nanmean(pos, axis=1)
    rad = np.sqrt(np.nanmax(((pos - ctr[:, None, :]) ** 2).sum(-1), axis=1))
    ctr_w = np.mod(ctr, L)
    tree = cKDTree(ctr_w, boxsize=L)
    pairs = tree.query_pairs(rlist + 2 * rad.max(), output_type="ndarray")
    pairs = np.concatenate([pairs, np.stack([np.arange(len(member))] * 2, 1)])
    lj_c = np.where(valid, is_lj[np.where(valid, member, 0)], False)
    has_lj = lj_c.any(1)
    tot = dict(listed=0, evaluated=0, in_rc=0, in_rc_lj=0, eval_nolj=0, eval_nolj_actual=0, half_listed=0, half_eval=0)
    entries = set()
    ent_sc = []
    for s in range(0, len(pairs), 200000):
        p = pairs[s:s + 200000]
        a, b = p[:, 0], p[:, 1]
        d = pos[a][:, :, None, :] - pos[b][:, None, :, :]
        d -= L * np.round(d / L)
        r2 = (d * d).sum(-1)                                  # (np, 8, 8), nan for fillers
        same = (a == b)
        if same.any():
            tri = np.triu(np.ones((8, 8), bool), 1)
            r2[same] = np.where(tri[None], r2[same], np.nan)
        r2 = np.where(np.isnan(r2), 1e9, r2)
        listed = (r2 < rlist * rlist).any((1, 2))
        inrc = r2 < rc * rc
        ev = inrc.any((1, 2))
        ljp = inrc & lj_c[a][:, :, None] & lj_c[b][:, None, :]
        tot["listed"] += int(listed.sum()); tot["evaluated"] += int(ev.sum())
        tot["in_rc"] += int(inrc.sum()); tot["in_rc_lj"] += int(ljp.sum())
        tot["eval_nolj"] += int((ev & ~(has_lj[a] & has_lj[b])).sum())
        tot["eval_nolj_actual"] += int((ev & ~ljp.any((1, 2))).sum())
        # 8 x 4 granularity: the two j-halves of a tile tested on their own
        hl = (r2.reshape(len(p), 8, 2, 4) < rlist * rlist).any((1, 3)); he = inrc.reshape(len(p), 8, 2, 4).any((1, 3))
        tot["half_listed"] += int(hl.sum()); tot["half_eval"] += int(he.sum())
        # entries: i side = the cluster of the lower super-cluster index
        sa, sb = sc_of[a], sc_of[b]
        isc = np.where(sa <= sb, sa, sb); jc = np.where(sa <= sb, b, a)
        key = isc[listed].astype(np.int64) * (1 << 32) + jc[listed]
        ent_sc.append(np.unique(key))
    tot["entries"] = int(len(np.unique(np.concatenate(ent_sc))))
    return tot


def price(t):
    npair = t["in_rc"]
    full = (t["listed"] * C_TEST + t["evaluated"] * C_EVAL + t["entries"] * C_ENTRY) * 64.0 / npair
    skip = (t["listed"] * C_TEST + (t["evaluated"] - t["eval_nolj"]) * C_EVAL + t["eval_nolj"] * C_EVAL_NOLJ
            + t["entries"] * C_ENTRY) * 64.0 / npair
    # two 8x4 halves per wave instruction: a half costs half a tile; entries are unchanged (the halves of one j-cluster share the fetch)
    half = (t["half_listed"] * 0.5 * C_TEST + t["half_eval"] * 0.5 * C_EVAL + t["entries"] * (C_ENTRY + 8.0)) * 64.0 / npair
    return full, skip, half


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=float, default=8.0)
    ap.add_argument("--rc", type=float, default=1.0)
    ap.add_argument("--rlist", type=float, default=1.095)
    args = ap.parse_args()
    L = args.side
    s = synth.water_box(box=np.diag([L, L, L]), jitter=0.03)
    x = np.mod(s.x, L)
    # molecules are kept whole by the engine's wrap only per atom; the model wraps per atom as the engine does
    is_lj = s.eps > 0
    dens = s.natoms / L ** 3
    print(f"{s.natoms} atoms, cubic {L} nm, {dens:.1f} atoms/nm^3, rc {args.rc}, rlist {args.rlist}")
    print(f"{'variant':34s} {'listed':>9s} {'eval':>9s} {'lanes in rc':>11s} {'LJ-free(type)':>13s} {'LJ-free(actual)':>15s} {'entries':>8s}"
          f" {'instr/pair':>10s} {'with LJ skip':>12s} {'8x4 halves':>10s}")
    for how, label in (("zyx", "engine: z / y / x sort"), ("kd", "kd bisection (largest extent)"), ("type", "LJ atoms first, kd inside")):
        t0 = time.time()
        member, sc_of = build(x, L, dens, how, is_lj)
        t = analyse(x, L, member, sc_of, is_lj, args.rc, args.rlist)
        full, skip, half = price(t)
        print(f"{label:34s} {t['listed']:9d} {t['evaluated']:9d} {t['in_rc'] / (64.0 * t['evaluated']):11.3f} "
              f"{t['eval_nolj'] / t['evaluated']:13.3f} {t['eval_nolj_actual'] / t['evaluated']:15.3f} {t['entries']:8d} {full:10.1f} {skip:12.1f} {half:10.1f}"
              f"   ({time.time() - t0:.0f} s; 8x4: listed {t['half_listed']}, evaluated {t['half_eval']}, lanes in rc {t['in_rc'] / (32.0 * t['half_eval']):.3f})")


if __name__ == "__main__":
    main()


def pairing_stats(x, L, member, sc_of, rc, rlist):
    """For the engine's clustering: how many of an entry's listed tiles pair up when the eight i-clusters of a super-cluster
    are evaluated two at a time, for the three static pairings (partner = the cluster's x, y or z neighbour inside the
    super-cluster: cluster index bit 0, 1 or 2)."""
    valid = member >= 0
    pos = np.where(valid[:, :, None], x[np.where(valid, member, 0)], np.nan)
    ctr = np.nanmean(pos, axis=1)
    rad = np.sqrt(np.nanmax(((pos - ctr[:, None, :]) ** 2).sum(-1), axis=1))
    tree = cKDTree(np.mod(ctr, L), boxsize=L)
    pairs = tree.query_pairs(rlist + 2 * rad.max(), output_type="ndarray")
    first = np.zeros(sc_of.max() + 2, dtype=np.int64)
    np.add.at(first, sc_of + 1, 1); first = np.cumsum(first)
    cin = np.arange(len(sc_of)) - first[sc_of]            # cluster index inside its super-cluster
    keys, bits = [], []
    for s in range(0, len(pairs), 200000):
        p = pairs[s:s + 200000]
        a, b = p[:, 0], p[:, 1]
        d = pos[a][:, :, None, :] - pos[b][:, None, :, :]
        d -= L * np.round(d / L)
        r2 = np.where(np.isnan((d * d).sum(-1)), 1e9, (d * d).sum(-1))
        listed = (r2 < rlist * rlist).any((1, 2))
        a, b = a[listed], b[listed]
        sa, sb = sc_of[a], sc_of[b]
        ic = np.where(sa <= sb, a, b); jc = np.where(sa <= sb, b, a)
        keys.append(sc_of[ic].astype(np.int64) * (1 << 32) + jc); bits.append(1 << cin[ic])
    keys = np.concatenate(keys); bits = np.concatenate(bits)
    order = np.argsort(keys, kind="stable"); keys = keys[order]; bits = bits[order]
    uniq, start = np.unique(keys, return_index=True)
    masks = np.bitwise_or.reduceat(bits, start)
    ntiles = sum(int(((masks >> c) & 1).sum()) for c in range(8))
    out = {}
    for bit, name in ((0, "x neighbour"), (1, "y neighbour"), (2, "z neighbour")):
        paired = 0
        for c in range(8):
            if c & (1 << bit):
                continue
            both = ((masks >> c) & 1) & ((masks >> (c | (1 << bit))) & 1)
            paired += 2 * int(both.sum())
        out[name] = paired / ntiles
    return len(uniq), ntiles, out


if __name__ == "__main__" and os.environ.get("TILE_MODEL_PAIRING"):
    L = 6.0
    s = synth.water_box(box=np.diag([L, L, L]), jitter=0.03)
    x = np.mod(s.x, L)
    member, sc_of = build(x, L, s.natoms / L ** 3, "zyx", s.eps > 0)
    for rl in (1.041, 1.095):
        n, nt, o = pairing_stats(x, L, member, sc_of, 1.0, rl)
        print("rlist %.3f: %d entries, %d listed tiles (%.2f per entry); fraction of tiles that pair up:" % (rl, n, nt, nt / n),
              ", ".join("%s %.3f" % kv for kv in o.items()))
