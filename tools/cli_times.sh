#!/bin/bash
# Where the stepping thread of `gmx mdrun` spends its time on the bench workload (MDX_RUNNER_TIMES=1): the bench's
# preparation as files, grompp, then mdrun with production.mdp's keys. Usage (GPU box): bash tools/cli_times.sh [nsteps]
set -e
R=$PWD; D=$(mktemp -d /tmp/mdx_cli_times_XXXX); N=${1:-10000}
python bench.py --write-cli-inputs $D > /dev/null 2>&1
python - "$D" "$N" <<'PY'
import sys
sys.path.insert(0, ".")
import bench
d, n = sys.argv[1], int(sys.argv[2])
with open(d + "/production.mdp", "w") as f:
    for k, v in bench.PRODUCTION_MDP:
        f.write("%-24s = %s\n" % (k, n if v is None else v))
PY
cd $D
python $R/bin/gmx grompp -f production.mdp -c conf.gro -t state.cpt -p topol.top -o prod.tpr -maxwarn 1000 > grompp.out 2>&1
MDX_RUNNER_TIMES=1 OMP_NUM_THREADS=4 python $R/bin/gmx mdrun -v -deffnm prod -ntmpi 1 -ntomp 4 -resethway > mdrun.out 2> mdrun.err
grep "mdrun loop" mdrun.err; grep "^Performance" prod.log
cd $R; rm -rf $D
