#!/usr/bin/env python3
"""Benchmark contract: python bench.py --gpus N --steps K --warmup W  -> one JSON line on rank 0.

Workload (N=1 and every N as independent replicas, weak scaling): the synthetic pseudo-complex
(alkane-bundle "protein" + 44-atom ligand + Na+ + TIP3P, rhombic dodecahedron d = 12.5 nm,
~1.4e5 atoms) run with the settings of /root/reference/mdp/production.mdp (dt 2 fs, rc 1.0 nm,
PME order 4 / 0.16 nm, h-bond constraints, V-rescale on Protein / Non-Protein, energies every
10 steps, DispCorr EnerPres, Parrinello-Rahman barostat; --pcoupl c-rescale selects stochastic cell rescaling instead).
A "step" is one MD step. Metric: ns/day = dt * steps * 86400 / wall / 1000 summed over replicas.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
JSON_OUT = sys.stdout

import numpy as np  # noqa: E402


def build_system(args):
    import mdx  # noqa: F401
    from mdx import synth
    if args.system == "complex":
        s = synth.pseudo_complex(d=args.box)
    else:
        s = synth.water_box(d=args.box)
    return s


def engine_for(s, args, phase, seed, device):
    from mdx import engine as eng
    tc = ["Protein", "Non-Protein"] if args.system == "complex" else ["System"]
    rc = 1.2 if args.workload == "smd" else 1.0
    common = dict(rc=rc, pme=True, fourierspacing=0.16, device=device, tc_groups=tc, ref_t=[300.0] * len(tc),
                  seed=seed, lincs_order=4, lincs_iter=1)
    if phase == "relax":      # un-timed preparation from the lattice start
        return eng.Engine(s, dt=0.0005, nstlist=10, nstcalcenergy=0, tau_t=[0.02] * len(tc), nsttcouple=2, **common)
    if phase == "equil":
        return eng.Engine(s, dt=0.002, nstlist=10, nstcalcenergy=0, tau_t=[0.1] * len(tc), nsttcouple=10, **common)
    if args.workload == "smd":  # pull_short.mdp: dt 1 fs, rc 1.2, NVT, 4 T-groups, lincs 6/2, COM pull
        common.update(tc_groups=["Protein", "MOL", "SOL", "NA"], ref_t=[300.0] * 4, lincs_order=6, lincs_iter=2)
        e = eng.Engine(s, dt=0.001, nstlist=args.nstlist, nstprune=args.nstprune, nstcalcenergy=100, tau_t=[0.1] * 4, nsttcouple=10, **common)
        e.set_pull(s.groups["ActiveSite"], s.groups["MOL"], type="constant-force", k=500.0, nstout=50)
        return e
    pc = {"c-rescale": 1, "parrinello-rahman": 2, "no": 0}[args.pcoupl]
    return eng.Engine(s, dt=0.002, nstlist=args.nstlist, nstprune=args.nstprune, nstcalcenergy=10, tau_t=[0.1] * len(tc), nsttcouple=10,
                      pcoupl=pc, tau_p=2.0, ref_p=1.0, compressibility=4.5e-5, nstpcouple=10, dispcorr=2, **common)


def carry(s, e):
    import copy
    x, v, box = e.get_state()
    t = copy.copy(s)
    t.x = x.astype(np.float64); t.v = v.astype(np.float64); t.box = box.astype(np.float64)
    return t


def cpu_baseline(s, rc, rlist, grid, dt, nsteps, threads):
    """The CPU baseline (oracle/cpu_md: single precision, OpenMP, Verlet list reused for 20 steps), timed on the host:
    a reported figure, never the product. Returns ns/day, wall seconds and the per-phase seconds."""
    from oracle import cpu_md
    import importlib
    U = importlib.import_module("molecular-dynamics_amd.units")
    c = cpu_md.CpuMD(s, rc, U.ewald_beta(rc), grid, dt, rlist=rlist, nstlist=20, threads=threads)
    c.step(1)                       # first list, first touches
    c.timers.clear()
    t0 = time.time()
    c.step(nsteps)
    wall = time.time() - t0
    timers = {k: round(v, 3) for k, v in c.timers.items()}
    c.close()
    return dt * nsteps * 86400.0 / wall / 1000.0, wall, timers


def find_gromacs():
    """A real GROMACS on this machine (not this repository's shim)? The reference's own CPU path would be timed with it."""
    import shutil
    for name in ("gmx", "gmx_mpi", "gmx_d"):
        p = shutil.which(name)
        if p and os.path.realpath(p) != os.path.realpath(os.path.join(ROOT, "bin", "gmx")):
            try:
                with open(p, "rb") as f:
                    if b"molecular-dynamics_amd" in f.read(512):
                        continue
            except OSError:
                continue
            return p
    return None


def run_dd(args, rank, world, local):
    """Tiled water box (S1 x 2x2x2, ~1.1e6 atoms) stepped by all ranks together: halo exchange of positions every
    step, the mesh transform shared by planes of the half-complex grid (or one all-reduce of the charge grid where the
    grid's planes do not fit the plane kernel: `config.mesh`), partition renewed every pair-list lifetime. NVT, dt 2 fs,
    rc 1.0 nm."""
    import torch
    import torch.distributed as dist
    from mdx import synth, engine as eng, domdec
    base = synth.water_box(d=args.box)
    # relax and equilibrate one cell on this rank's GPU (every rank computes the same state), then tile it
    e = eng.Engine(base, rc=1.0, dt=0.0005, nstlist=10, nstcalcenergy=0, tc_groups=["System"], ref_t=[300.0], tau_t=[0.02], nsttcouple=2,
                   device=local, seed=1)
    e.step(args.prep_steps); x, v, _ = e.get_state(); e.close()
    base.x = x.astype(np.float64); base.v = v.astype(np.float64)
    e = eng.Engine(base, rc=1.0, dt=0.002, nstlist=10, nstcalcenergy=0, tc_groups=["System"], ref_t=[300.0], tau_t=[0.1], nsttcouple=10,
                   device=local, seed=1)
    e.step(args.prep_steps); x, v, _ = e.get_state(); e.close()
    base.x = x.astype(np.float64); base.v = v.astype(np.float64)
    big = synth.tiled(base, (2, 2, 2))
    dd = domdec.DecomposedWater(big, device=local, rc=1.0, dt=0.002, nstlist=args.nstlist, ref_t=300.0, tau_t=0.1, nsttcouple=10)
    dd.step(args.warmup); dd.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    dd.step(args.steps); dd.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "ns/day", "value": round(0.002 * args.steps * 86400.0 / elapsed / 1000.0, 3), "unit": "ns/day", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "tiled TIP3P box S4 (%d atoms), spatial domain decomposition %s, NVT, dt 2 fs, rc 1.0 nm, PME 0.16 nm"
                                   % (big.natoms, "x".join(str(d) for d in dd.dims)),
                       "natoms": big.natoms, "home_molecules_rank0": dd.n_home, "halo_molecules_rank0": dd.n_halo,
                       "rlist_nm": round(dd.rlist, 4), "nstlist": args.nstlist, "pme_grid": list(dd.engine.grid), "mesh": dd.mesh,
                       "parallelism": "dd%d" % world}}), file=JSON_OUT, flush=True)
    dd.close()
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as children (one process per GPU,
    rendezvous on 127.0.0.1) before this process makes any GPU call, pass their output through and exit with their
    status. Counting devices does not initialise the GPU."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs, %d visible on this machine\n" % (args.gpus, args.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# production.mdp of the pipeline (/root/reference/mdp/production.mdp:3-46), key by key, with `nsteps` of the timed run
PRODUCTION_MDP = [
    ("integrator", "md"), ("nsteps", None), ("dt", "0.002"), ("nstxout", "10000"), ("nstvout", "0"), ("nstfout", "0"),
    ("nstenergy", "10"), ("nstlog", "10"), ("nstxout-compressed", "10"), ("compressed-x-grps", "System"),
    ("continuation", "yes"), ("constraint_algorithm", "lincs"), ("constraints", "h-bonds"), ("lincs_iter", "1"), ("lincs_order", "4"),
    ("cutoff-scheme", "Verlet"), ("ns_type", "grid"), ("nstlist", "10"), ("rcoulomb", "1.0"), ("rvdw", "1.0"),
    ("coulombtype", "PME"), ("pme_order", "4"), ("fourierspacing", "0.16"),
    ("tcoupl", "V-rescale"), ("tc-grps", "Protein Non-Protein"), ("tau_t", "0.1 0.1"), ("ref_t", "300 300"),
    ("pcoupl", "Parrinello-Rahman"), ("pcoupltype", "isotropic"), ("tau_p", "2.0"), ("ref_p", "1.0"), ("compressibility", "4.5e-5"),
    ("pbc", "xyz"), ("DispCorr", "EnerPres"), ("gen_vel", "no"),
]


def write_cli_inputs(args, outdir):
    """Child process of the command-line leg: the bench's own preparation (relax, equilibrate) on the GPU, then the
    prepared system as the files `gmx grompp` reads: conf.gro, topol.top with its includes, state.cpt."""
    import mdx  # noqa: F401
    from mdx import ensemble
    from mdx.io import gro, top, tpr
    s = build_system(args)
    seed = ensemble.replica_seed(20250829, 0)
    e = engine_for(s, args, "relax", seed, 0); e.step(args.prep_steps); s1 = carry(s, e); e.close()
    e = engine_for(s1, args, "equil", seed, 0); e.step(args.prep_steps); s2 = carry(s1, e); e.close()
    top.write_top_tree(outdir, s2)
    gro.write_system_gro(os.path.join(outdir, "conf.gro"), s2, x=s2.x, v=s2.v, box=s2.box)
    tpr.save_checkpoint(os.path.join(outdir, "state.cpt"), 0, 0.0, s2.x, s2.v, s2.box)
    return 0


def cli_leg(args):
    """The drop-in path, timed: the prepared system goes through `gmx grompp` and `gmx mdrun -deffnm prod -ntmpi 1 -ntomp 4`
    (this repository's bin/gmx, the way /root/reference/Snakefile:311-315 runs the production stage) as child processes,
    production.mdp's keys, 2 x --steps steps with -resethway, and the `Performance:` line of prod.log is what counts.
    Called before this process touches the GPU; every GPU call of the leg happens in a child."""
    import re
    import shutil
    import tempfile
    gmx = os.path.join(ROOT, "bin", "gmx")
    d = tempfile.mkdtemp(prefix="mdx_cli_leg_")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    res = {"command": "gmx grompp -f production.mdp -c conf.gro -t state.cpt -p topol.top -o prod.tpr -maxwarn 1000 && "
                      "gmx mdrun -v -deffnm prod -ntmpi 1 -ntomp 4 -resethway"}
    try:
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--write-cli-inputs", d, "--box", str(args.box), "--system", args.system,
                            "--prep-steps", str(args.prep_steps)], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("input preparation failed: " + r.stderr[-400:])
        nsteps = 2 * max(args.steps, 2500)
        with open(os.path.join(d, "production.mdp"), "w") as f:
            for k, v in PRODUCTION_MDP:
                f.write("%-24s = %s\n" % (k, nsteps if v is None else v))
        t1 = time.time()
        r = subprocess.run([sys.executable, gmx, "grompp", "-f", "production.mdp", "-c", "conf.gro", "-t", "state.cpt", "-p", "topol.top",
                            "-o", "prod.tpr", "-maxwarn", "1000"], cwd=d, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            raise RuntimeError("grompp failed: " + (r.stderr or r.stdout)[-400:])
        t2 = time.time()
        r = subprocess.run([sys.executable, gmx, "mdrun", "-v", "-deffnm", "prod", "-ntmpi", "1", "-ntomp", "4", "-resethway"], cwd=d,
                           capture_output=True, text=True, timeout=900, env=env)
        t3 = time.time()
        if r.returncode != 0:
            raise RuntimeError("mdrun failed: " + (r.stderr or r.stdout)[-400:])
        log = open(os.path.join(d, "prod.log")).read()
        m = re.search(r"^Performance:\s+(\S+)\s+(\S+)", log, re.M)
        if not m:
            raise RuntimeError("no Performance line in prod.log")
        files = {n: os.path.getsize(os.path.join(d, n)) for n in ("prod.xtc", "prod.edr", "prod.log", "prod.gro", "prod.cpt", "prod.trr")
                 if os.path.exists(os.path.join(d, n))}
        lst = re.search(r"pair-list radius: (\S+)\s+nstlist: (\d+)(?:\s+pruned every (\d+) steps to radius (\S+))?", log)
        res.update({"cli_ns_per_day": float(m.group(1)), "hour_per_ns": float(m.group(2)), "nsteps": nsteps,
                    "timed_steps": nsteps - nsteps // 2, "host_threads_for_output": 4,
                    "grompp_s": round(t2 - t1, 2), "mdrun_process_s": round(t3 - t2, 2), "prepare_inputs_s": round(t1 - t0, 2),
                    "outputs_MB": {n: round(v / 1e6, 2) for n, v in files.items()},
                    "pair_list": lst.group(0) if lst else None})
    except Exception as ex:          # the leg is a report beside `value`: its failure must not take the bench line with it
        res["error"] = "%s: %s" % (type(ex).__name__, ex)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


def private_stdout():
    """The JSON line must be the only thing on stdout, but RCCL prints a version banner there when its communicator is
    created. File descriptor 1 is therefore pointed at stderr for the life of the process (C libraries included) and the
    original stdout is kept for the one line."""
    sys.stdout.flush()
    keep = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--system", default="complex", choices=["complex", "water"])
    ap.add_argument("--box", type=float, default=12.5)
    ap.add_argument("--workload", default="production", choices=["production", "smd", "dd"],
                    help="dd: the tiled ~1.1 M-atom water box decomposed over the N GPUs (strong scaling); not the default")
    ap.add_argument("--pcoupl", default="parrinello-rahman", choices=["c-rescale", "parrinello-rahman", "no"])      # production.mdp:36
    ap.add_argument("--nstlist", type=int, default=0,
                    help="pair-list lifetime; the .mdp value 10 is a lower bound that mdrun raises on GPUs. Default: 100 with the "
                         "list pruned every 10 steps (dual list), 20 for the decomposed workload")
    ap.add_argument("--nstprune", type=int, default=-1,
                    help="dual pair list: the list of --nstlist steps is pruned to the pairs within the short-lived radius every "
                         "this many steps (0: one list; default 10 with the default --nstlist)")
    ap.add_argument("--nb-chunk", type=int, default=0, help="diagnostic: list entries per nonbonded work unit (0: the library's 32)")
    ap.add_argument("--no-cli", action="store_true", help="skip the leg that times the same workload through bin/gmx grompp + mdrun")
    ap.add_argument("--write-cli-inputs", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-steps", type=int, default=50, help="steps of the CPU baseline per thread count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prep-steps", type=int, default=2000)
    ap.add_argument("--no-output", action="store_true", help="diagnostic: skip the trajectory output the .mdp asks for")
    ap.add_argument("--chunk", type=int, default=0, help="diagnostic with --no-output: steps per mdx_step call")
    ap.add_argument("--ring", type=int, default=8, help="pinned trajectory-frame slots between the GPU and the encoder threads")
    ap.add_argument("--encoders", type=int, default=1, help="frames compressed at the same time")
    ap.add_argument("--encoder-threads", type=int, default=4,
                    help="host threads that share the bit packing of one frame (segments). The defaults use four host threads "
                         "for output, what the pipeline grants mdrun (-ntomp 4): one frame at a time on four threads")
    ap.add_argument("--prime-steps", type=int, default=400,
                    help="un-timed steps before the warmup that run every kernel variant, pair searches, the coupling steps and the "
                         "frame pipeline (first launches and first frames carry one-off costs of milliseconds), and that bring the GPU "
                         "back to its sustained clocks after the host-side set-up of the run: with 40 of them a 20-step window "
                         "still saw the nonbonded kernel at 0.123 ms per launch, with 400 at the 0.113 ms of a long run")
    args = ap.parse_args()
    if args.nstlist <= 0:
        args.nstlist = 20 if args.workload == "dd" else 100
        if args.nstprune < 0:
            args.nstprune = 0 if args.workload == "dd" else 10
    args.nstprune = max(args.nstprune, 0)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if args.write_cli_inputs:
        return write_cli_inputs(args, args.write_cli_inputs)
    # the command-line leg runs first, in child processes, while this process has not touched the GPU yet
    cli = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.workload == "production" and not args.no_cli
            and not args.no_output and args.system == "complex"):
        cli = cli_leg(args)
    global JSON_OUT
    JSON_OUT = private_stdout()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    args.encoder_threads = max(1, args.encoder_threads)
    use_dist = world > 1 or bool(os.environ.get("MDX_BENCH_FORCE_DIST"))     # the latter: rehearsal of the N>1 code path on one GPU
    ctl = None
    if use_dist and (args.workload == "dd" or os.environ.get("MDX_BENCH_EAGER_RCCL")):      # (the latter: diagnostic)
        if os.environ.get("MDX_BENCH_DD_LAZY"):      # diagnostic
            dist.init_process_group("nccl")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))       # collectives in every step
        # The communicator's first collective is completed before any engine handle exists: handles created between an
        # eager initialisation and that first collective run their three-stream step a third slower for the rest of the
        # process (259 against 175 us per step on the production workload, tools/rccl_queue_probe.py); handles created
        # before the initialisation or after the first collective do not.
        if not os.environ.get("MDX_BENCH_NO_FIRST_COLLECTIVE"):
            first = torch.ones(1, device="cuda"); dist.all_reduce(first); torch.cuda.synchronize()
    elif use_dist:
        # Replicas exchange nothing while they run: the SMD ensemble's one collective (the work curves, RCCL) comes after
        # the timed region, production replicas have none. The RCCL communicator is therefore left to be created by that
        # first collective (if any), and the
        # barriers and the wall-time maximum around the timed region go through a host-side (gloo) group: a process that
        # merely holds an initialised RCCL communicator runs this engine's three-stream step a third slower (662 against
        # 982 ns/day measured on one MI355X; the kernels take the same time, the hand-overs between the streams do not).
        dist.init_process_group("nccl")
        ctl = dist.new_group(backend="gloo")

    import mdx  # noqa: F401
    from mdx import engine as eng, ensemble
    if args.workload == "dd":
        return run_dd(args, rank, world, local)
    s = build_system(args)
    seed = ensemble.replica_seed(20250829, rank)
    # ---- preparation (not timed): relax the lattice start, then equilibrate at the production time step
    e = engine_for(s, args, "relax", seed, local); e.step(args.prep_steps); s1 = carry(s, e); e.close()
    e = engine_for(s1, args, "equil", seed, local); e.step(args.prep_steps); s2 = carry(s1, e); e.close()
    e = engine_for(s2, args, "run", seed, local)
    if args.nb_chunk:
        e.debug_set(14, args.nb_chunk)
    for kv in filter(None, os.environ.get("MDX_DEBUG", "").split(",")):     # diagnostic switches "key=value,..."
        e.debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
    dt = e.dt
    # ---- mdp-mandated output: compressed frames of the whole system every nstxout-compressed steps
    # (/root/reference/mdp/production.mdp:12-13, pull_short.mdp:8), written by encoder threads while the step loop
    # keeps the GPU queue filled
    nst_xtc = 0 if args.no_output else (10 if args.workload == "production" else 500)
    pipe = xtc = None
    xtc_path = None
    if nst_xtc:
        import tempfile
        from mdx.io import traj
        from mdx.output import FramePipeline
        xtc_path = os.path.join(tempfile.gettempdir(), "mdx_bench_rank%d_%d.xtc" % (rank, os.getpid()))
        xtc = traj.XtcWriter(xtc_path, s.natoms, precision=1000.0, threads=args.encoder_threads)
        e.frame_ring(args.ring)
        pipe = FramePipeline(e, args.ring, xtc, None, None, dt, 0.0, nthreads=args.encoders)

    def run_steps(nsteps):
        """The production loop: a frame whenever the step counter hits the cadence, steps in between."""
        if nst_xtc:
            done = 0
            while done < nsteps:
                if (e.get_step() % nst_xtc) == 0:
                    pipe.capture(True, False, False)
                n = min(nsteps - done, nst_xtc - e.get_step() % nst_xtc)
                e.step(n)
                done += n
        elif args.chunk > 0:      # diagnostic: the call pattern of the output loop without the output
            for _ in range(nsteps // args.chunk):
                e.step(args.chunk)
        else:
            e.step(nsteps)

    # ---- priming (not timed, before the warmup): every kernel variant of the step (plain, energy, coupling, search),
    # the frame ring's first copies and the encoder threads' first frames carry one-off costs
    pair_stats = None
    if args.prime_steps > 0:
        # the pair statistics behind the roofline figures are counted here only (mdx_debug_set key 11): the energy steps of
        # the timed region run without the counters, as a production run does
        e.debug_set(11, 1)
        run_steps(args.prime_steps); e.synchronize()
        pair_stats = e.list_stats()
        e.debug_set(11, 0)
        if pipe is not None:
            # the frames of the priming go to a file of their own: truncating and rewriting one file makes some file
            # systems write everything back synchronously when it is closed (30 ms for the 260 MB of 5000 steps)
            pipe.wait_idle()
            xtc.f.close(); os.remove(xtc_path)
            xtc_path = os.path.join(tempfile.gettempdir(), "mdx_bench_rank%d_%d_timed.xtc" % (rank, os.getpid()))
            xtc.f = open(xtc_path, "wb"); xtc.frames = 0
    # ---- warmup (the event pool of the kernel timer is created ahead of it: two thousand hipEventCreate calls take
    # the host tens of milliseconds, in which an idle GPU lowers its clocks; the second call only resets the sums)
    stride = 1 if args.steps <= 500 else 7
    e.enable_kernel_timing(True, stride=stride)
    e.step(args.warmup); e.synchronize()
    e.pop_energy_frames()
    e.enable_kernel_timing(True, stride=stride)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(group=ctl)
    t0 = time.perf_counter()
    run_steps(args.steps)
    t_enq = time.perf_counter()
    e.synchronize(); torch.cuda.synchronize()
    t_gpu = time.perf_counter()
    pending = pipe.pending() if pipe is not None else 0
    t_fin = t_gpu
    if pipe is not None:
        pipe.finish()
        t_fin = time.perf_counter()
        xtc.close()
    if use_dist:
        dist.barrier(group=ctl)
    elapsed = time.perf_counter() - t0
    # where the wall time of the timed region went on this rank: enqueueing the steps (the host runs ahead of the GPU),
    # the GPU finishing what was enqueued, the encoder threads finishing the frames still in flight
    phases = {"enqueue_ms": round((t_enq - t0) * 1e3, 3), "gpu_drain_ms": round((t_gpu - t_enq) * 1e3, 3),
              "output_drain_ms": round((time.perf_counter() - t_gpu) * 1e3, 3),
              "frames_pending_when_gpu_done": pending, "encoders_finish_ms": round((t_fin - t_gpu) * 1e3, 3)}
    if pipe is not None and pipe.stats["frames"]:
        nf = pipe.stats["frames"]
        phases["per_frame_ms"] = {k[:-2]: round(v / nf * 1e3, 3) for k, v in pipe.stats.items() if k.endswith("_s")}
        if "last" in pipe.stats:      # the last frame's way through the encoder, milliseconds from the start of the timed region
            phases["last_frame_ms"] = dict(zip(("dequeued", "copy_landed", "packed", "committed"),
                                               [round((t - t0) * 1e3, 3) for t in pipe.stats["last"]]))
            phases["gpu_done_ms"] = round((t_gpu - t0) * 1e3, 3)
    xtc_frames = xtc.frames if xtc is not None else 0
    xtc_bytes = os.path.getsize(xtc_path) if xtc_path else 0
    if pipe is not None and pipe.error is not None:
        raise pipe.error
    if xtc_path:
        os.remove(xtc_path)
    nb_ms, nb_launches = e.kernel_timing()
    e.enable_kernel_timing(False)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if (ctl is not None or not use_dist) else "cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
    elapsed_max = float(tmax.item())
    frames = e.pop_energy_frames()
    stats = e.list_stats()
    if pair_stats is not None and pair_stats["pairs_within_rc"] > 0:
        stats["pairs_within_rc"] = pair_stats["pairs_within_rc"]; stats["pairs_with_lj"] = pair_stats["pairs_with_lj"]
    else:
        e.compute(); stats = e.list_stats()       # a single evaluation counts
    smd = None
    if args.workload == "smd":
        rec = e.pop_pull_records()
        allw = ensemble.gather_work(rec[:, 3])
        dF = ensemble.jarzynski_free_energy(allw, 300.0)
        smd = dict(samples=int(len(rec)), replicas=int(allw.shape[0]), final_distance=float(rec[-1, 1]) if len(rec) else None,
                   initial_distance=float(rec[0, 1]) if len(rec) else None, k=500.0,
                   work_final_rank0=float(rec[-1, 3]) if len(rec) else None,
                   work_final_all=[float(w) for w in allw[:, -1]] if len(rec) else None,
                   jarzynski_dF_final=float(dF[-1]) if len(dF) else None)
    # (production replicas exchange nothing at all: no collective is invented for them, and the RCCL communicator of the
    # process group is never created; the SMD ensemble's one all-gather above is the path's real exchange)
    ns_day = dt * args.steps * 86400.0 / elapsed_max / 1000.0 * world
    out = None
    if rank == 0:
        n_rc, n_lj = stats["pairs_within_rc"], stats["pairs_with_lj"]
        flops = 66.0 * n_lj + 54.0 * (n_rc - n_lj)
        nb_avg_s = nb_ms / max(nb_launches, 1) * 1e-3
        achieved = flops / nb_avg_s / 1e12 if nb_avg_s > 0 else 0.0
        peak = 157.3
        # HBM bytes per launch of the nonbonded kernel from the PMC passes kept under profiles/
        # (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs; not collected live)
        traffic, traffic_src = None, None
        for rnd in ("r03", "r02", "r01"):          # the newest collection that is present
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "nb_pmc_traffic.json")) as f:
                    traffic = float(json.load(f)["hbm_bytes_per_launch"])
                traffic_src = "profiles/%s/nb_pmc_traffic.json" % rnd
                break
            except (OSError, KeyError, ValueError):
                pass
        out = {
            "metric": "ns/day", "value": round(ns_day, 3), "unit": "ns/day", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("pseudo-complex S2 (%d atoms, dodecahedron d=%.1f nm), production.mdp settings with "
                                    "pcoupl = %s, energies every 10 steps, %s" % (
                                        s.natoms, args.box, args.pcoupl,
                                        "xtc frame of the whole system every 10 steps compressed and written" if nst_xtc
                                        else "NO trajectory output (diagnostic run)"))
                       if args.workload == "production" else
                       ("pseudo-complex S2 (%d atoms), pull_short.mdp settings, constant-force COM pull, one replica per GPU, "
                        "xtc every 500 steps" % s.natoms),
                       "output": {"xtc_every": nst_xtc, "xtc_frames": xtc_frames, "xtc_MB": round(xtc_bytes / 1e6, 2),
                                  "frames_in_flight": args.encoders if nst_xtc else 0, "threads_per_frame": args.encoder_threads if nst_xtc else 0},
                       "untimed_steps_before_warmup": args.prime_steps,
                       "natoms": s.natoms, "dt_ps": dt, "rc_nm": e.rc, "rlist_nm": round(e.rlist, 4), "nstlist": args.nstlist,
                       "nstprune": e.nstprune, "rlist_inner_nm": (round(e.rlist_inner, 4) if e.rlist_inner else None),
                       "pme_grid": list(e.grid), "parallelism": "replicas x%d" % world},
            "pair_interactions_per_s": round(n_rc / nb_avg_s, 1) if nb_avg_s > 0 else None,
            "pair_interactions_per_s_end_to_end": round(n_rc * args.steps / elapsed_max * world, 1),
            "roofline": {"bound": "valu_f32", "kernel": "k_nonbonded", "achieved": round(achieved, 3), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (separate rocprofv3 --pmc passes on this workload; %s)" % traffic_src,
                         "peak_unpacked": 78.6, "frac_unpacked": round(achieved / 78.6, 4),
                         "flops_per_launch": flops, "pairs_within_rc": n_rc, "pairs_with_lj": n_lj,
                         "avg_launch_ms": round(nb_avg_s * 1e3, 5), "launches": nb_launches,
                         "evaluated_pair_lanes": stats["cluster_pairs"] * 64},
            "timed_region": phases,
            "final": {"T": float(frames[-1, 2 + eng.E["TEMP"]]) if len(frames) else None,
                      "P_bar": float(frames[-1, 2 + eng.E["PRES"]]) if len(frames) else None,
                      "Epot": float(frames[-1, 2 + eng.E["POT"]]) if len(frames) else None,
                      "volume": float(frames[-1, 2 + eng.E["VOLUME"]]) if len(frames) else None},
        }
        if smd:
            out["smd"] = smd
        if cli is not None:
            out["cli"] = cli
            out["cli_ns_per_day"] = cli.get("cli_ns_per_day")
        if not args.no_cpu_baseline and world == 1:
            s3 = carry(s2, e)
            # 4 threads as the pipeline runs gmx mdrun (-ntomp 4, /root/reference/Snakefile:232-233), and the cores of this job
            ncores = min(16, os.cpu_count() or 4)
            runs = []
            cpu_rlist = eng.verlet_buffer_rlist(s3, e.rc, e.beta, 20, dt, 300.0, 0.005)
            for threads in sorted({4, ncores}):
                # (the CPU path keeps one list for 20 steps: it gets the radius of a 20-step list, not the GPU's outer radius)
                v, wall, timers = cpu_baseline(s3, e.rc, cpu_rlist, e.grid, dt, args.cpu_steps, threads)
                runs.append({"threads": threads, "ns_per_day": round(v, 4), "ms_per_step": round(wall / args.cpu_steps * 1e3, 2),
                             "seconds": round(wall, 1), "phases_s": timers})
            gmx = find_gromacs()
            out["cpu_baseline"] = {"value": runs[0]["ns_per_day"], "unit": "ns/day", "cores": runs[0]["threads"], "kind": "port",
                                   "sample": "%d leap-frog steps of the same %d-atom system (same rc and PME grid, pair-list radius of its 20-step list) on "
                                             "the host: own CPU restatement, not GROMACS (single precision, OpenMP C, Verlet list "
                                             "reused for 20 steps, pocketfft transforms; oracle/cpu_md.*)" % (args.cpu_steps, s.natoms),
                                   "runs": runs, "gmx": gmx or "unavailable",
                                   "note": "GROMACS is not installed on this machine, so the reference's own CPU path "
                                           "(gmx mdrun -nb cpu -ntmpi 1 -ntomp 4) could not be timed" if gmx is None else
                                           "a GROMACS binary is present but the pipeline's inputs need its force-field tree; not timed"}
        # the same kernel launched back to back on the current pair list with nothing beside it (inside the step it shares
        # the SIMDs with the PME and bonded kernels, which is what `achieved` above prices)
        try:
            alone_ms = e.debug_time_nonbonded(False, 30)
            out["roofline"]["avg_launch_ms_alone"] = round(alone_ms, 5)
            out["roofline"]["achieved_alone"] = round(flops / (alone_ms * 1e-3) / 1e12, 3)
            out["roofline"]["frac_alone"] = round(flops / (alone_ms * 1e-3) / 1e12 / peak, 4)
        except Exception as ex:
            out["roofline"]["avg_launch_ms_alone"] = "failed: %s" % ex
        # The other kernels of a plain step, each timed alone on its stream with HIP events after the run (the state is
        # scrap afterwards): algorithmic bytes per launch / time. At 1.4e5 atoms and a 1.4 MB grid (resident in the L2 /
        # memory-side cache) none of them streams from HBM: what bounds them is the latency of their dependent memory
        # round trips and of their atomics, so the figure is labelled as such and the HBM fraction is only a yardstick.
        try:
            K = e.grid; k3 = 4.0 * K[0] * K[1] * K[2]; n = s.natoms
            sec = []
            for stage, name, nbytes, bound in ((6, "k_pme_spread_slab", 16.0 * n + 2 * k3, "latency (LDS and global atomics)"),
                                               (7, "k_pme_gather", k3 + 28.0 * n, "latency (grid resident in cache)"),
                                               (8, "k_update_fused", 140.0 * n, "latency (three dependent round trips per atom)")):
                ms = e.debug_time_stage(stage, 50)
                gbs = nbytes / (ms * 1e-3) / 1e9
                sec.append({"kernel": name, "bound": bound, "achieved": round(gbs, 1), "hbm_peak": 8000.0, "unit": "GB/s",
                            "frac_of_hbm_peak": round(gbs / 8000.0, 4), "bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 5)})
            out["roofline_secondary"] = sec
        except Exception as ex:      # diagnostics must not lose the bench line
            out["roofline_secondary"] = "failed: %s" % ex
        print(json.dumps(out), file=JSON_OUT, flush=True)
    e.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
