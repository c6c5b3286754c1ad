"""Run-input and checkpoint containers of the `gmx` shim.

`gmx mdrun -deffnm PREFIX` receives nothing but the prefix
(/root/reference/scripts/run_smd_with_distance_monitoring.py:89-95): everything `grompp` was told
(`-f -c -r -t -p -n`, /root/reference/Snakefile:738) has to travel in the file `grompp -o` wrote.
The shim's run input is a self-contained, uncompressed numpy archive holding the resolved
parameters (JSON) and the fully expanded system (flat arrays), loadable without executing
anything (`allow_pickle=False`). Checkpoints (`PREFIX.cpt`) use the same container with the
state only. Neither is a GROMACS `.tpr`/`.cpt`: when a real GROMACS is installed the shim also
runs its `grompp` so that a genuine `.tpr` exists for the analysis tools, and keeps its own run
input next to it as `PREFIX.tpr.mdx`.
"""
from __future__ import annotations

import json
import os
import numpy as np

from ..system import MDSystem

MAGIC_TPR = "mdx-run-input-1"
MAGIC_CPT = "mdx-checkpoint-1"
_ARRAYS = ["x", "box", "mass", "charge", "sigma", "eps", "molid", "v", "resnr", "excl_start", "excl_list", "bonds",
           "bond_b0", "bond_kb", "angles", "angle_th0", "angle_k", "dihedrals", "dih_phi0", "dih_k", "dih_mult", "pairs14",
           "pair_c6", "pair_c12", "pair_qq", "posres_idx", "posres_ref", "posres_k", "settles", "constraints",
           "constraint_len"]


def _savez(path, **arrays):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        np.savez(f, **arrays)
    os.replace(tmp, path)


def save_run_input(path: str, sysm: MDSystem, mdp: dict, meta: dict | None = None) -> None:
    a = {}
    for k in _ARRAYS:
        val = getattr(sysm, k)
        if val is not None:
            a["sys_" + k] = np.asarray(val)
    groups = sysm.groups.as_dict() if hasattr(sysm.groups, "as_dict") else dict(sysm.groups)
    names = list(groups.keys())
    for i, n in enumerate(names):
        a["grp_%d" % i] = np.asarray(groups[n], dtype=np.int32)
    header = dict(magic=MAGIC_TPR, mdp=dict(mdp), meta=meta or {}, group_names=names, title=sysm.title,
                  settle_doh=sysm.settle_doh, settle_dhh=sysm.settle_dhh)
    a["header"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
    a["resname"] = np.array(sysm.resname if sysm.resname is not None else [], dtype="U8")
    a["atomname"] = np.array(sysm.atomname if sysm.atomname is not None else [], dtype="U8")
    _savez(path, **a)


def is_run_input(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(2) == b"PK"
    except OSError:
        return False


def load_run_input(path: str):
    """-> (MDSystem, mdp dict, meta dict)."""
    if not is_run_input(path):
        raise ValueError(f"{path} is not a run input written by this grompp (a GROMACS .tpr cannot be read; "
                         "run the shim's grompp on the same inputs)")
    with np.load(path, allow_pickle=False) as z:
        header = json.loads(bytes(z["header"]).decode())
        if header.get("magic") != MAGIC_TPR:
            raise ValueError(f"{path}: unknown run-input version {header.get('magic')}")
        kw = {}
        for k in _ARRAYS:
            if "sys_" + k in z.files:
                kw[k] = z["sys_" + k]
        from .ndx import IndexGroups
        g = IndexGroups()
        for i, n in enumerate(header["group_names"]):
            g.add(n, z["grp_%d" % i])
        resname = [str(s) for s in z["resname"]]
        atomname = [str(s) for s in z["atomname"]]
    sysm = MDSystem(**kw)
    sysm.groups = g
    sysm.title = header.get("title", "")
    sysm.settle_doh = header.get("settle_doh", sysm.settle_doh)
    sysm.settle_dhh = header.get("settle_dhh", sysm.settle_dhh)
    sysm.resname = resname or None
    sysm.atomname = atomname or None
    return sysm, header["mdp"], header["meta"]


def save_checkpoint(path: str, step: int, time: float, x, v, box, extra: dict | None = None) -> None:
    header = dict(magic=MAGIC_CPT, step=int(step), time=float(time), extra=extra or {})
    _savez(path, header=np.frombuffer(json.dumps(header).encode(), dtype=np.uint8),
           x=np.asarray(x, dtype=np.float32), v=np.asarray(v, dtype=np.float32), box=np.asarray(box, dtype=np.float64))


def load_checkpoint(path: str):
    """-> dict(step, time, x, v, box, extra)."""
    if not is_run_input(path):
        raise ValueError(f"{path} is not a checkpoint written by this mdrun (a GROMACS .cpt cannot be read: continue from the "
                         "shim's own PREFIX.cpt, or start the stage from the .gro alone)")
    with np.load(path, allow_pickle=False) as z:
        header = json.loads(bytes(z["header"]).decode())
        if header.get("magic") != MAGIC_CPT:
            raise ValueError(f"{path}: not a checkpoint ({header.get('magic')})")
        return dict(step=header["step"], time=header["time"], extra=header.get("extra", {}),
                    x=z["x"].astype(np.float64), v=z["v"].astype(np.float64), box=z["box"].astype(np.float64))
