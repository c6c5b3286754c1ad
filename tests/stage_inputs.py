"""Stage inputs for the command-line tests: a small pseudo-complex written as .gro/.top/.ndx and
the pipeline's stage parameters as .mdp files.

The parameter sets restate, key by key, what the pipeline's stage files configure
(/root/reference/mdp/em_complex.mdp:3-18, nvt_complex.mdp:2-41, npt_complex.mdp:7-72,
production.mdp:3-46, pull_short.mdp:1-46); run lengths and output cadences are shortened so that a
whole chain runs in seconds.
"""
from __future__ import annotations

import os

import mdx  # noqa: F401
from mdx import synth
from mdx.io import gro, ndx, top

COMMON = {
    "cutoff-scheme": "Verlet", "ns_type": "grid", "coulombtype": "PME", "pme_order": "4", "fourierspacing": "0.16",
    "pbc": "xyz", "constraint_algorithm": "lincs", "constraints": "h-bonds", "lincs_iter": "1", "lincs_order": "4",
}
STAGES = {
    "em": {"integrator": "steep", "emtol": "1000.0", "emstep": "0.01", "nsteps": "200", "nstlist": "1",
           "cutoff-scheme": "Verlet", "ns_type": "grid", "coulombtype": "PME", "rcoulomb": "1.0", "rvdw": "1.0", "pbc": "xyz",
           "define": "-DPOSRES"},
    "nvt": dict(COMMON, **{"define": "-DPOSRES", "integrator": "md", "nsteps": "300", "dt": "0.002", "nstxout": "100",
                           "nstvout": "100", "nstenergy": "50", "nstlog": "50", "continuation": "no", "nstlist": "10",
                           "rcoulomb": "1.0", "rvdw": "1.0", "DispCorr": "EnerPres", "tcoupl": "V-rescale",
                           "tc-grps": "Protein Non-Protein", "tau_t": "0.1 0.1", "ref_t": "300 300", "pcoupl": "no",
                           "gen_vel": "yes", "gen_temp": "300", "gen_seed": "-1"}),
    "npt": dict(COMMON, **{"define": "-DPOSRES", "integrator": "md", "nsteps": "300", "dt": "0.002", "nstxout": "0",
                           "nstvout": "0", "nstfout": "0", "nstenergy": "100", "nstlog": "100", "nstxout-compressed": "100",
                           "continuation": "yes", "nstlist": "200", "rcoulomb": "1.0", "rvdw": "1.0",
                           "verlet-buffer-tolerance": "0.005", "tcoupl": "V-rescale", "tc-grps": "Protein Non-Protein",
                           "tau_t": "1.0 1.0", "ref_t": "300 300", "nsttcouple": "100", "pcoupl": "Parrinello-Rahman",
                           "pcoupltype": "isotropic", "tau_p": "5.0", "ref_p": "1.0", "compressibility": "4.5e-5",
                           "nstpcouple": "100", "DispCorr": "EnerPres", "gen_vel": "no", "nstcomm": "1000",
                           "comm-mode": "Linear", "comm-grps": "System", "nstcalcenergy": "100", "refcoord_scaling": "all"}),
    "production": dict(COMMON, **{"integrator": "md", "nsteps": "400", "dt": "0.002", "nstxout": "200", "nstvout": "0",
                                  "nstfout": "0", "nstenergy": "10", "nstlog": "10", "nstxout-compressed": "10",
                                  "compressed-x-grps": "System", "continuation": "yes", "nstlist": "10", "rcoulomb": "1.0",
                                  "rvdw": "1.0", "tcoupl": "V-rescale", "tc-grps": "Protein Non-Protein", "tau_t": "0.1 0.1",
                                  "ref_t": "300 300", "pcoupl": "Parrinello-Rahman", "pcoupltype": "isotropic",
                                  "tau_p": "2.0", "ref_p": "1.0", "compressibility": "4.5e-5", "DispCorr": "EnerPres",
                                  "gen_vel": "no"}),
    "pull": dict(COMMON, **{"integrator": "md", "dt": "0.001", "nsteps": "250000", "nstxout": "500", "nstvout": "500",
                            "nstenergy": "100", "nstlog": "100", "nstxout-compressed": "500", "nstcheckpoint": "1000",
                            "continuation": "yes", "lincs_iter": "2", "lincs_order": "6", "rlist": "1.2", "rcoulomb": "1.2",
                            "rvdw": "1.2", "tcoupl": "V-rescale", "tc-grps": "Protein MOL SOL NA",
                            "tau_t": "0.1 0.1 0.1 0.1", "ref_t": "300 300 300 300", "pcoupl": "no", "gen_vel": "no",
                            "pull": "yes", "pull_ncoords": "1", "pull_ngroups": "2", "pull_group1_name": "ActiveSite",
                            "pull_group2_name": "MOL", "pull_coord1_type": "constant-force",
                            "pull_coord1_geometry": "distance", "pull_coord1_groups": "1 2", "pull_coord1_dim": "Y Y Y",
                            "pull_coord1_start": "yes", "pull_coord1_k": "500"}),
}


def write_mdp(path, stage, **override):
    params = dict(STAGES[stage])
    params.update({k: str(v) for k, v in override.items()})
    with open(path, "w") as f:
        f.write(f"; {stage} stage parameters\n")
        for k, v in params.items():
            f.write("%-24s = %s\n" % (k, v))
    return path


def write_system(dirpath, d=6.0, protein_radius=1.0, n_ions=3, n_active=60):
    """Small pseudo-complex as conf.gro, topol.top (+ includes) and index.ndx with an ActiveSite group."""
    os.makedirs(dirpath, exist_ok=True)
    s = synth.pseudo_complex(d=d, protein_radius=protein_radius, n_ions=n_ions, n_active=n_active)
    top.write_top_tree(dirpath, s)
    gro.write_system_gro(os.path.join(dirpath, "conf.gro"), s)
    groups = ndx.default_groups(s.resname, s.atomname)
    groups.add("ActiveSite", s.groups["ActiveSite"])
    ndx.write_ndx(os.path.join(dirpath, "index.ndx"), groups)
    return s
