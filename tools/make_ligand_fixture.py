#!/usr/bin/env python3
"""Extract element symbols, coordinates (nm) and the bond table of the 44-atom ligand from
the pipeline's input structure (/root/reference/inputs/50-28-2_Estradiol.sdf:5-95) into a
compact JSON data table used by the synthetic pseudo-complex generator. Run once in the
build container; the JSON is committed, the SDF is not."""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/inputs/50-28-2_Estradiol.sdf"
out = sys.argv[2] if len(sys.argv) > 2 else "molecular-dynamics_amd/data/ligand44.json"
L = open(src).read().splitlines()
na, nb = int(L[3][0:3]), int(L[3][3:6])
el, xyz, bonds = [], [], []
for l in L[4:4 + na]:
    p = l.split()
    xyz.append([round(float(p[0]) * 0.1, 5), round(float(p[1]) * 0.1, 5), round(float(p[2]) * 0.1, 5)])
    el.append(p[3])
for l in L[4 + na:4 + na + nb]:
    bonds.append([int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])])
json.dump({"name": "estradiol", "elements": el, "xyz_nm": xyz, "bonds": bonds}, open(out, "w"))
print(na, nb, out)
