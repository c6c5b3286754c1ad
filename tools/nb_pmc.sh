#!/bin/bash
# HBM traffic counters of the force-only nonbonded kernel, one counter per pass (GPU box, repository root).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/tools/nb_pmc.py > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" | tail -1)
  python3 $R/tools/pmc_summary.py $f "k_nonbonded_ft(" $c
done
# summary for bench.py (FETCH_SIZE and WRITE_SIZE are in KB; gfx950 tallies this kernel's 16- and 8-byte per-lane loads at half size)
cd $R
python3 - <<'PY'
import csv, json, statistics, glob
def med(c):
    f = sorted(glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % c))[-1]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_nonbonded_ft(" in r["Kernel_Name"] and r["Counter_Name"] == c]
    return statistics.median(v), len(v)
(fe, n), (wr, _) = med("FETCH_SIZE"), med("WRITE_SIZE")
json.dump({"kernel": "k_nonbonded_ft (force-only nonbonded, Lennard-Jones by atom type, 12-wave workgroups, 3 waves/SIMD)",
           "command": "bash tools/nb_pmc.sh  (rocprofv3 --pmc FETCH_SIZE -- python3 tools/nb_pmc.py ; rocprofv3 --pmc WRITE_SIZE -- python3 tools/nb_pmc.py: separate passes, no trace domains)",
           "system": "bench workload: pseudo-complex S2, 138792 atoms, dodecahedron d=12.5 nm, rc 1.0, dual pair list (outer radius from the verlet-buffer estimate for 100 steps, pruned every 10 steps)",
           "dispatches_sampled": n, "FETCH_SIZE_KB_median": fe, "WRITE_SIZE_KB_median": wr,
           "fetch_correction": "x2 (gfx950 tallies 128-B read requests at 64 B for wide per-lane loads; the kernel's loads are 16-B and 8-B per lane)",
           "hbm_bytes_per_launch": round((2 * fe + wr) * 1024),
           "note": "WRITE_SIZE counts the float atomics of the force accumulation (one dword per lane); far below the 8 TB/s HBM peak: the kernel is bound by its vector arithmetic and the latency of each wave's dependent chain, not by memory"},
          open("gpurun_out/nb_pmc_traffic.json", "w"), indent=1)
PY
