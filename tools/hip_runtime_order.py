#!/usr/bin/env python3
"""Which HIP runtime a process ends up with when the engine library is loaded before torch initialises the GPU.
Usage: python tools/hip_runtime_order.py [preload]"""
import ctypes, importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "preload":
    spec = importlib.util.find_spec("torch")
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    print("preloading", p, os.path.exists(p))
    ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
import mdx
from mdx import synth, engine as eng
e = eng.Engine(synth.water_box(d=3.4), rc=1.0)
e.step(5); e.synchronize(); e.close()
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l and "r-xp" in l])
import torch
try:
    t = torch.zeros(4, device="cuda"); print("torch on the GPU after the engine: ok", t.sum().item())
except Exception as ex:
    print("torch on the GPU after the engine: FAILED:", ex)
print(sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}))
