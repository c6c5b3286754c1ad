"""`gmx` front-end: `grompp` and `mdrun` run on the MI355X engine, everything else is forwarded.

The pipeline invokes `gmx <tool>` as child processes (/root/reference/Snakefile:705-710,
/root/reference/scripts/run_smd_with_distance_monitoring.py:66-95). Putting `bin/gmx` of this
repository first on PATH replaces the two MD tools and leaves the Snakefile and scripts
unchanged; the preparation and analysis tools (`pdb2gmx`, `editconf`, `solvate`, `genion`,
`trjconv`, `rms`, `make_ndx`) go to an installed GROMACS when there is one; `trjconv -s P.tpr -f P.cpt -o P.gro`,
the recovery step of the SMD drivers, is served from the shim's own run input and checkpoint.
"""
from __future__ import annotations

import os
import subprocess
import sys


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    # global flags GROMACS accepts before the tool name
    while argv and argv[0] in ("-quiet", "-nobackup", "-nocopyright"):
        argv.pop(0)
    if not argv or argv[0] in ("-h", "--help", "help"):
        sys.stdout.write("usage: gmx {grompp|mdrun} [options]   (MI355X engine; other tools are forwarded to an installed GROMACS)\n")
        return 0
    if argv[0] in ("-version", "--version"):
        sys.stdout.write("gmx shim of the MI355X MD engine (grompp, mdrun)\n")
        return 0
    tool, rest = argv[0], argv[1:]
    from . import runner
    if tool == "grompp":
        return runner.grompp(rest)
    if tool == "mdrun":
        return runner.mdrun(rest)
    if tool == "make_ndx" and (runner.find_real_gmx() is None or any(
            os.path.isfile(x + ".mdx") for x in rest if not x.startswith("-"))):
        return runner.make_ndx(rest)
    if tool == "trjconv" and runner.trjconv_can_serve(rest):
        return runner.trjconv(rest)
    real = runner.find_real_gmx()
    if real is None:
        sys.stderr.write(f"gmx {tool}: this tool is not part of the MD engine and no GROMACS installation was found "
                         "to forward it to (set MDX_REAL_GMX to its gmx binary)\n")
        return 127
    return subprocess.call([real] + argv)


if __name__ == "__main__":
    sys.exit(main())
