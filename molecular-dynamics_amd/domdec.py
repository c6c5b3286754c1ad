"""Spatial domain decomposition of one large rigid-water system over several GPUs.

One process per GPU (`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm), one engine handle
per process. The reference pipeline never decomposes a system (every `gmx mdrun` runs with `-ntmpi 1`,
/root/reference/Snakefile:232-233); this module serves the scaled synthetic water box only.

Scheme (see csrc/domdec.hip for the device side):

* The unit cell is cut into px x py x pz bricks in fractional coordinates. A molecule belongs to the brick
  that holds its first atom. Every pair-list lifetime (`nstlist` steps) the partition is renewed by migration
  (`migrate`): every rank sends the molecules that left its brick to their new owners, then sends every other rank
  the positions of its home molecules inside that rank's brick grown by `rlist` + molecule size (a conservative
  test per fractional dimension, so it holds in any triclinic cell): the receiver's halo. The selection doubles as
  the send list of the following steps; receive counts travel in one all-gather of `world` integers per round.
  (`migration=False` keeps the first scheme as a cross-check: all ranks all-gather every home molecule and each
  computes every rank's halo, `plan_exchange`.)
* Every other step: the positions of those atoms are packed on the device, exchanged point to point
  (one batched send/recv group: `ncclSend`/`ncclRecv` under NCCL) and scattered into the halo slots.
  Forces are not sent back: a pair across a boundary is evaluated by both owners, each keeping its own half.
* PME: every rank spreads its home charges on a full-size grid and transforms it along x (a sum over ranks commutes
  with that pass); the half-complex grids are reduce-scattered by chunks of k_x planes, every rank does the y and z
  transforms and the influence function on its own planes (each plane is a self-contained problem), the planes are
  all-gathered, and every rank transforms back along x and gathers forces for its home atoms (`mesh="slabs"`).
  `mesh="replicated"`, and any PME grid whose planes the fused plane kernel does not take, keeps the first scheme: the
  real-space grids are summed with one all-reduce and every rank solves the whole mesh problem.
* Thermostat, barostat (C-rescale or Parrinello-Rahman), dispersion correction and centre-of-mass removal act on the
  whole system: the domains sum a few scalars per coupling step and apply the same factors (`DecomposedWater`).

The partition arithmetic (`owner_of`, `halo_mask`, `plan_exchange`, `migrate`) is plain torch on whatever device the
tensors live on, so it is tested on CPU tensors with a two-rank gloo group and eight in-process ranks.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from . import units as U
from .system import MDSystem

MOL_ATOMS = 3           # rigid three-site water
MOL_RADIUS = 0.11       # nm: no atom of a molecule is farther than this from its first atom


# --------------------------------------------------------------------------------------------- partition arithmetic
def rank_grid(world: int):
    """px, py, pz with px*py*pz == world, as cubic as possible, in ascending order (x gets the fewest cuts)."""
    best = (1, 1, world)
    for a in range(1, world + 1):
        if world % a:
            continue
        for b in range(1, world // a + 1):
            if (world // a) % b:
                continue
            c = world // a // b
            cand = tuple(sorted((a, b, c)))
            if max(cand) - min(cand) < max(best) - min(best):
                best = cand
    return best  # ascending: x gets the smallest count


def fractional(x: torch.Tensor, box: torch.Tensor) -> torch.Tensor:
    """Fractional coordinates in [0, 1) of positions x (.., 3) in the cell with rows box[0..2]."""
    s = x @ torch.linalg.inv(box)
    return s - torch.floor(s)


def owner_of(s_first: torch.Tensor, dims) -> torch.Tensor:
    """Rank owning each molecule from the fractional coordinates (M, 3) of its first atom."""
    d = torch.tensor(dims, device=s_first.device, dtype=s_first.dtype)
    c = torch.minimum(torch.floor(s_first * d), d - 1).to(torch.int64)
    return (c[:, 0] * dims[1] + c[:, 1]) * dims[2] + c[:, 2]


def brick_of(rank: int, dims):
    cz = rank % dims[2]; cy = (rank // dims[2]) % dims[1]; cx = rank // (dims[1] * dims[2])
    lo = (cx / dims[0], cy / dims[1], cz / dims[2])
    hi = ((cx + 1) / dims[0], (cy + 1) / dims[1], (cz + 1) / dims[2])
    return lo, hi


def fractional_margins(box: torch.Tensor, radius: float):
    """Per dimension d, the largest change of the fractional coordinate s_d over a distance `radius`:
    |grad s_d| = norm of column d of inv(box)."""
    inv = torch.linalg.inv(box)
    return [float(radius * torch.linalg.norm(inv[:, d])) for d in range(3)]


def halo_mask(s_first: torch.Tensor, owner: torch.Tensor, rank: int, dims, margins) -> torch.Tensor:
    """Molecules not owned by `rank` whose first atom is within the margins of its brick (periodic)."""
    lo, hi = brick_of(rank, dims)
    m = owner != rank
    for d in range(3):
        if dims[d] == 1 or (hi[d] - lo[d]) + 2 * margins[d] >= 1.0:
            continue                      # the brick spans this dimension, or brick + margins wrap around: everything is near
        c = 0.5 * (lo[d] + hi[d]); half = 0.5 * (hi[d] - lo[d])
        dist_c = torch.abs(s_first[:, d] - c)
        dist_c = torch.minimum(dist_c, 1.0 - dist_c)          # periodic distance to the brick centre
        m = m & (dist_c < half + margins[d])
    return m


def plan_exchange(s_first, owner, rank, world, dims, margins):
    """For rank `rank`: indices (into the gathered molecule order) of its home molecules, of its halo molecules
    grouped by owner (ascending rank), and per destination the indices of its home molecules that the
    destination holds as halo, in the destination's slot order."""
    home = torch.nonzero(owner == rank).flatten()
    recv = []
    mine = halo_mask(s_first, owner, rank, dims, margins)
    for src in range(world):
        recv.append(torch.nonzero(mine & (owner == src)).flatten() if src != rank else home[:0])
    send = []
    for dst in range(world):
        if dst == rank:
            send.append(home[:0]); continue
        theirs = halo_mask(s_first, owner, dst, dims, margins)
        send.append(torch.nonzero(theirs & (owner == rank)).flatten())
    return home, recv, send


def migrate(tr, x, v, ids, box, dims, margins):
    """Renew the partition from what each rank holds, without gathering the system anywhere.

    x, v: (n_home, 3, 3) positions and velocities of this rank's home molecules, ids: (n_home,) their global numbers.
    1. every rank computes the new owner of its home molecules; those that left its brick (within a pair-list
       lifetime a molecule moves a fraction of a nanometre, so they go to one of the <= 26 bricks around it, but any
       owner is served) are sent to their owners: x, v and the id in one record of 19 floats;
    2. every rank then selects, per other rank, its home molecules inside that rank's brick grown by the halo margins
       and sends their positions: the receiver's halo, grouped by source rank in ascending order, in the order sent;
       the same selection is the send list of the per-step halo refresh.
    The receive counts of both rounds travel in one all-gather of `world` integers each. Traffic per rank and lifetime:
    the emigrants and the halo once, instead of every molecule of the system.
    Returns x_home, v_home, ids_home, [halo positions (m_src, 3, 3) per source rank], [home slots (k_dst,) per
    destination rank]."""
    rank, world = tr.rank, tr.world
    dev = x.device
    empty_i = torch.zeros((0,), dtype=torch.int64, device=dev)
    if world == 1:
        return x, v, ids, [x[:0]], [empty_i]
    own = owner_of(fractional(x[:, 0, :].double(), box), dims)
    rec = torch.cat([x.reshape(-1, 9).float(), v.reshape(-1, 9).float(), ids.to(torch.int32).view(torch.float32).reshape(-1, 1)], 1)
    out = [torch.nonzero(own == r).flatten() if r != rank else empty_i for r in range(world)]
    counts = torch.tensor([int(o.numel()) for o in out], dtype=torch.int64, device=dev)
    table = torch.stack(tr.all_gather(counts))               # table[src][dst]
    send = [rec[o].contiguous() for o in out]
    recv = [torch.zeros((int(table[r][rank]) if r != rank else 0, 19), dtype=torch.float32, device=dev) for r in range(world)]
    tr.exchange(send, recv)
    rec = torch.cat([rec[own == rank]] + [recv[r] for r in range(world) if r != rank])
    x_home = rec[:, :9].reshape(-1, 3, 3).to(x.dtype); v_home = rec[:, 9:18].reshape(-1, 3, 3).to(v.dtype)
    ids_home = rec[:, 18].contiguous().view(torch.int32).to(ids.dtype)
    # halo: my home molecules that lie in another rank's grown brick
    s_home = fractional(x_home[:, 0, :].double(), box)
    mine = torch.full((x_home.shape[0],), rank, dtype=torch.int64, device=dev)
    sends = [torch.nonzero(halo_mask(s_home, mine, r, dims, margins)).flatten() if r != rank else empty_i for r in range(world)]
    counts = torch.tensor([int(o.numel()) for o in sends], dtype=torch.int64, device=dev)
    table = torch.stack(tr.all_gather(counts))
    pos = x_home.reshape(-1, 9).float()
    hsend = [pos[o].contiguous() for o in sends]
    hrecv = [torch.zeros((int(table[r][rank]) if r != rank else 0, 9), dtype=torch.float32, device=dev) for r in range(world)]
    tr.exchange(hsend, hrecv)
    return x_home, v_home, ids_home, [h.reshape(-1, 3, 3).to(x.dtype) for h in hrecv], sends


# --------------------------------------------------------------------------------------------- transport
class Transport:
    """Collectives on device tensors; with a gloo group the tensors are staged through the host (tests)."""

    def __init__(self, group=None):
        import os
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # device-ordered collectives: RCCL work is ordered against the stream that is current when it is issued, so no host
        # synchronisation is needed around it. MDX_DD_FORCE_COLLECTIVES=1 issues the reductions at world size 1 as well
        # (rehearsal of that ordering on one GPU).
        # Between real peers the ordered path has not run on hardware yet (the build has one GPU): at world size > 1 the
        # host-synchronised path is the default and MDX_DD_ORDERED=1 opts in.
        self.ordered = dist.is_initialized() and not self.staged and (self.world == 1 or os.environ.get("MDX_DD_ORDERED") == "1")
        self.force = self.ordered and bool(os.environ.get("MDX_DD_FORCE_COLLECTIVES"))
        self.poison = False

    def all_gather(self, t: torch.Tensor):
        if self.world == 1:
            return [t]
        src = t.cpu() if self.staged else t
        out = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(out, src.contiguous(), group=self.group)
        return [o.to(t.device) for o in out] if self.staged else out

    def all_reduce_sum_(self, t: torch.Tensor):
        if self.world == 1 and not self.force:
            return t
        if self.staged:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    def reduce_scatter_sum_(self, t: torch.Tensor):
        """t: (world, chunk) contiguous. On return t[rank] holds the sum of every rank's t[rank]; the other rows are
        undefined (with `poison` they are filled with NaN, so that a reader of a row it does not own shows up)."""
        if self.world == 1 and not self.force:
            return t
        if self.staged:
            h = t.to("cpu", copy=True)
            dist.all_reduce(h, group=self.group)        # (gloo has no reduce-scatter)
            if self.poison:
                t.fill_(float("nan"))
            t[self.rank].copy_(h[self.rank])
        else:
            dist.reduce_scatter_tensor(t[self.rank], t, group=self.group)
        return t

    def all_gather_rows_(self, t: torch.Tensor):
        """t: (world, chunk) contiguous. On return row r of every rank's t is rank r's row r."""
        if self.world == 1 and not self.force:
            return t
        if self.staged:
            out = [torch.empty(t.shape[1:], dtype=t.dtype) for _ in range(self.world)]
            dist.all_gather(out, t[self.rank].cpu().contiguous(), group=self.group)
            t.copy_(torch.stack(out))
        else:
            dist.all_gather_into_tensor(t, t[self.rank], group=self.group)
        return t

    def exchange(self, send, recv):
        """send[r] goes to rank r, recv[r] is filled from rank r (tensors, possibly empty); one batched group."""
        if self.world == 1:
            return
        ops, stage = [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            if recv[r].numel():
                buf = torch.empty(recv[r].shape, dtype=recv[r].dtype) if self.staged else recv[r]
                if self.staged:
                    stage.append((recv[r], buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, self.group))
            if send[r].numel():
                ops.append(dist.P2POp(dist.isend, send[r].cpu() if self.staged else send[r], r, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, buf in stage:
            dst.copy_(buf)


class ThreadTransport:
    """The same collectives between `world` domains that live in one process, one thread each (rehearsal and tests
    on a single GPU). The threads run concurrently, each driving its own handle, and meet at the collectives; with
    `exclusive=True` only one thread runs at a time (each holds the hub's lock while it computes and hands it over at
    a collective). `Hub.run(fn)` starts `fn(rank, transport)` on `world` threads."""

    class Hub:
        def __init__(self, world: int, exclusive: bool = False):
            import threading
            self.world = world
            self.exclusive = exclusive
            self.barrier = threading.Barrier(world)
            self.lock = threading.Lock()
            self.slots = [None] * world

        def port(self, rank: int):
            return ThreadTransport(self, rank)

        def run(self, fn):
            """fn(rank, transport) on one thread per domain; returns the list of results, re-raises the first error."""
            import contextlib
            import threading
            res, err = [None] * self.world, []

            def body(rank):
                with (self.lock if self.exclusive else contextlib.nullcontext()):
                    try:
                        res[rank] = fn(rank, self.port(rank))
                    except BaseException as e:          # a failing domain must not leave the others at a barrier
                        err.append(e)
                        self.barrier.abort()
            th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if err:
                raise err[0]
            return res

    def __init__(self, hub, rank: int):
        self.hub, self.rank, self.world, self.staged, self.group = hub, rank, hub.world, False, None
        self.in_process = True
        self.serial = hub.exclusive     # DecomposedWater drains its handle before it hands the GPU over
        self.stream = None              # torch stream of this thread's collectives (set by DecomposedWater)
        self.poison = True              # rows a reduce-scatter leaves undefined are filled with NaN

    def _meet(self):
        """All work of this thread complete, then wait for the others (with the lock handed over, if there is one)."""
        if torch.cuda.is_available():
            if self.stream is not None and not self.hub.exclusive:
                self.stream.synchronize()
            else:
                torch.cuda.synchronize()
        if self.hub.exclusive:
            self.hub.lock.release()
        try:
            self.hub.barrier.wait()
        finally:
            if self.hub.exclusive:
                self.hub.lock.acquire()

    def all_gather(self, t):
        self.hub.slots[self.rank] = t
        self._meet()
        out = [s.clone() for s in self.hub.slots]
        self._meet()
        return out

    def all_reduce_sum_(self, t):
        self.hub.slots[self.rank] = t.clone()
        self._meet()
        total = torch.stack(list(self.hub.slots)).sum(0)
        self._meet()
        t.copy_(total)
        return t

    def reduce_scatter_sum_(self, t):
        self.hub.slots[self.rank] = t
        self._meet()
        mine = torch.stack([s[self.rank] for s in self.hub.slots]).sum(0)
        self._meet()
        if self.poison:
            t.fill_(float("nan"))
        t[self.rank].copy_(mine)
        return t

    def all_gather_rows_(self, t):
        self.hub.slots[self.rank] = t
        self._meet()
        rows = [self.hub.slots[r][r].clone() for r in range(self.world) if r != self.rank]
        self._meet()
        for r, row in zip([r for r in range(self.world) if r != self.rank], rows):
            t[r].copy_(row)
        return t

    def exchange(self, send, recv):
        self.hub.slots[self.rank] = send
        self._meet()
        for r in range(self.world):
            if r != self.rank and recv[r].numel():
                recv[r].copy_(self.hub.slots[r][self.rank])
        self._meet()


class _DevArray:
    """Wraps a raw device pointer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _template(nmol: int, box: np.ndarray) -> MDSystem:
    """`nmol` TIP3P slots (O, H, H) with exclusions and settles; positions are installed by mdx_dd_load."""
    from . import synth
    mass, q, sig, eps = synth._water_arrays(nmol)
    n = MOL_ATOMS * nmol
    idx = np.arange(n, dtype=np.int32).reshape(nmol, 3)
    o, h1, h2 = idx[:, 0], idx[:, 1], idx[:, 2]
    excl = np.stack([np.stack([h1, h2], 1), np.stack([o, h2], 1), np.stack([o, h1], 1)], 1).reshape(-1)
    s = MDSystem(x=np.zeros((n, 3)), box=np.asarray(box, dtype=np.float64), mass=mass, charge=q, sigma=sig, eps=eps,
                 molid=np.repeat(np.arange(nmol, dtype=np.int32), 3),
                 excl_start=(2 * np.arange(n + 1)).astype(np.int32), excl_list=excl.astype(np.int32))
    s.settles = idx.copy()
    s.v = np.zeros((n, 3))
    s.groups = {"System": np.arange(n, dtype=np.int32)}
    return s


# --------------------------------------------------------------------------------------------- the decomposed run
class DecomposedWater:
    """A rigid-water system (MDSystem whose molecules are consecutive O, H, H triples with settles) stepped on
    `world` GPUs. Every rank constructs this object with the same global system."""

    def __init__(self, sysm: MDSystem, device: int = 0, rc: float = 1.0, dt: float = 0.002, nstlist: int = 20,
                 ref_t: float | None = 300.0, tau_t: float = 0.1, nsttcouple: int = 10, fourierspacing: float = 0.16,
                 rlist: float | None = None, seed: int = 1, group=None, dims=None, nstcalcenergy: int = 0,
                 transport=None, global_coupling: bool = True, nstcomm: int = 0, pcoupl: str | None = None, tau_p: float = 2.0,
                 ref_p: float = 1.0, compressibility: float = 4.5e-5, nstpcouple: int = 10, dispcorr: int = 0,
                 migration: bool = True, mesh: str = "slabs"):
        """global_coupling: one V-rescale factor for the whole system from the kinetic energy summed over the domains
        (every domain draws it from the same random stream); False: every domain is its own thermostat group.
        nstcomm > 0: centre-of-mass motion of the whole system removed every nstcomm steps.
        pcoupl = "c-rescale": isotropic stochastic cell rescaling from the pressure summed over the domains, every nstpcouple
        steps (needs the shared thermostat seed: every domain draws the same factor); "parrinello-rahman"
        (/root/reference/mdp/production.mdp:36): every domain carries the same box velocity and advances it with the summed
        pressure.
        migration: the partition is renewed by sending the molecules that left a brick, and the halo, to the ranks that need
        them (`migrate`); False gathers every molecule on every rank each pair-list lifetime (the cross-check).
        mesh: "slabs" shares the mesh transform among the ranks (every rank transforms its own k_x planes of the
        half-complex grid between a reduce-scatter and an all-gather); "replicated" sums the real-space charge grids with
        one all-reduce and lets every rank solve the whole mesh problem (also taken when the PME grid's planes do not fit
        the plane kernel: `self.mesh` tells which is in use).
        dispcorr: 1 adds the analytic long-range dispersion correction of the whole system (`DispCorr = Ener`) to the
        energies, 2 (`EnerPres`, /root/reference/mdp/production.mdp:44) also to the pressure, the coupled one included: it
        depends on the atom count, the volume and <C6> of the whole system only, so every rank evaluates it itself."""
        from . import engine as eng
        self.tr = transport if transport is not None else Transport(group)
        self.rank, self.world = self.tr.rank, self.tr.world
        self.migration = bool(migration)
        self.dims = tuple(dims) if dims is not None else rank_grid(self.world)
        assert self.dims[0] * self.dims[1] * self.dims[2] == self.world
        n = sysm.natoms
        assert n % MOL_ATOMS == 0 and len(sysm.settles) == n // MOL_ATOMS, "rigid three-site molecules only"
        self.nmol = n // MOL_ATOMS
        self.dev = torch.device("cuda", device)
        self.box_np = np.asarray(sysm.box, dtype=np.float64)
        self.box = torch.tensor(self.box_np, dtype=torch.float64, device=self.dev)
        self.volume = float(abs(np.linalg.det(self.box_np)))
        self.density = n / self.volume
        self.nstlist = int(nstlist)
        beta = U.ewald_beta(rc)
        if rlist is None:
            rlist = eng.verlet_buffer_rlist(sysm, rc, beta, nstlist, dt, ref_t or 300.0, 0.005)
        self.rlist = float(rlist)
        self.margins = fractional_margins(self.box, self.rlist + 2 * MOL_RADIUS + 0.02)
        # capacities: home share with head room for density fluctuations, halo from the brick geometry
        frac = [1.0 / d for d in self.dims]
        grown = [min(1.0, f + 2 * m) if d > 1 else 1.0 for f, m, d in zip(frac, self.margins, self.dims)]
        share = frac[0] * frac[1] * frac[2]
        self.home_cap = int(self.nmol * share * 1.10) + 256
        self.halo_cap = int(self.nmol * (grown[0] * grown[1] * grown[2] - share) * 1.10) + 256 if self.world > 1 else 0
        self.cap = min(self.home_cap + self.halo_cap, self.nmol + 256)
        tmpl = _template(self.cap, self.box_np)
        grid = U.pme_grid(self.box_np, fourierspacing)
        if pcoupl not in (None, "no", "c-rescale", "parrinello-rahman"):
            raise ValueError("pcoupl: c-rescale or parrinello-rahman")
        self.pcoupl = pcoupl in ("c-rescale", "parrinello-rahman")
        pc_kind = {"c-rescale": 1, "parrinello-rahman": 2}.get(pcoupl, 0)
        if self.pcoupl and not global_coupling:
            raise ValueError("pressure coupling across domains needs global_coupling (one random stream for all domains)")
        self.nstpcouple = int(nstpcouple)
        self.dispcorr, self.rc, self.natoms = int(dispcorr), float(rc), int(n)
        self.c6avg = eng.c6_average(np.asarray(sysm.sigma), np.asarray(sysm.eps)) if self.dispcorr else 0.0
        self.rlist_margin = self.rlist + 2 * MOL_RADIUS + 0.02
        self.engine = eng.Engine(tmpl, rc=rc, rlist=self.rlist, dt=dt, nstlist=nstlist, pme=True, grid=grid, device=device,
                                 tc_groups=["System"] if ref_t is not None else None, ref_t=[ref_t or 300.0], tau_t=[tau_t],
                                 nsttcouple=nsttcouple, nstcalcenergy=nstcalcenergy, nstcomm=0,
                                 pcoupl=pc_kind, tau_p=tau_p, ref_p=ref_p, compressibility=compressibility,
                                 nstpcouple=nstpcouple, seed=seed if global_coupling else seed + self.rank)
        if self.pcoupl:
            self.pres_t = torch.as_tensor(_DevArray(self.engine.dd_pressure(), (9,), "<f8"), device=self.dev)
        # With an RCCL transport every collective is issued with the handle's own stream current (main stream; PME stream
        # for the charge grid): the reductions and exchanges then take their place in the handle's stream order and the
        # host never waits for the device inside a step.
        self.ordered = bool(getattr(self.tr, "ordered", False)) and (self.world > 1 or getattr(self.tr, "force", False))
        if self.ordered:
            sm, sp = self.engine.dd_streams()
            self.st_main = torch.cuda.ExternalStream(sm, device=self.dev)
            self.st_pme = torch.cuda.ExternalStream(sp, device=self.dev)
        self.thermostat = ref_t is not None
        self.global_coupling = bool(global_coupling)
        self.nsttcouple, self.nstcomm, self.nstcalcenergy = int(nsttcouple), int(nstcomm), int(nstcalcenergy)
        if mesh not in ("slabs", "replicated"):
            raise ValueError("mesh: slabs or replicated")
        self.mesh = "slabs" if mesh == "slabs" and self.engine.dd_mesh_slabs(self.world, self.rank) else "replicated"
        if self.mesh == "slabs":
            # every domain reports the reciprocal-space energy of its own planes
            self.engine.dd_set_roles(True, not (self.global_coupling and self.thermostat))
            cp, chunk, plane = self.engine.dd_cgrid()
            self.cgrid_t = torch.as_tensor(_DevArray(cp, (self.world, chunk * plane * 2)), device=self.dev)
        else:
            # the reciprocal-space energy is the same on every domain: the first one reports it
            self.engine.dd_set_roles(self.rank == 0, not (self.global_coupling and self.thermostat))
            gp, gn = self.engine.dd_grid()
            self.grid_t = torch.as_tensor(_DevArray(gp, (gn,)), device=self.dev)
        ep, cp = self.engine.dd_scalars()
        self.ekinh_t = torch.as_tensor(_DevArray(ep, (8,), "<f8"), device=self.dev)
        self.com_t = torch.as_tensor(_DevArray(cp, (4,), "<f8"), device=self.dev)
        # initial home set from the given coordinates
        x = torch.tensor(np.asarray(sysm.x, dtype=np.float32), device=self.dev).reshape(self.nmol, MOL_ATOMS, 3)
        v = torch.tensor(np.asarray(sysm.v if sysm.v is not None else np.zeros((n, 3)), dtype=np.float32), device=self.dev).reshape(self.nmol, MOL_ATOMS, 3)
        ids = torch.arange(self.nmol, device=self.dev, dtype=torch.int64)
        own = owner_of(fractional(x[:, 0, :].double(), self.box), self.dims)
        self._install(x, v, ids, own)
        self.steps_since_partition = 0
        self.nsteps_done = 0

    # ------------------------------------------------------------------ partition
    def _install(self, x_all, v_all, ids_all, own):
        """Install home + halo from the gathered molecule arrays (M, 3, 3), (M, 3, 3), (M,), owner (M,)."""
        s_first = fractional(x_all[:, 0, :].double(), self.box)
        home, recv, send = plan_exchange(s_first, own, self.rank, self.world, self.dims, self.margins)
        # position of each gathered molecule in my home block
        slot_of = torch.full((x_all.shape[0],), -1, device=self.dev, dtype=torch.int64)
        slot_of[home] = torch.arange(int(home.numel()), device=self.dev)
        self._install_local(x_all[home], v_all[home], ids_all[home], [x_all[rv] for rv in recv], [slot_of[sd] for sd in send])

    def _install_local(self, x_home, v_home, ids_home, halo_x, send_slots):
        """Install this rank's home molecules (n, 3, 3), (n, 3, 3), (n,), the halo positions received per source rank
        (ascending) and, per destination rank, the home slots whose positions it holds as halo (in its slot order)."""
        halo = torch.cat(halo_x) if self.world > 1 else x_home[:0]
        nh, nl = int(x_home.shape[0]), int(halo.shape[0])
        if nh + nl > self.cap:
            raise RuntimeError(f"rank {self.rank}: {nh} home + {nl} halo molecules exceed the capacity {self.cap}")
        if nh == 0:
            raise RuntimeError(f"rank {self.rank}: empty domain")
        x4 = torch.zeros((MOL_ATOMS * (nh + nl), 4), device=self.dev, dtype=torch.float32)
        v4 = torch.zeros_like(x4)
        x4[: MOL_ATOMS * nh, :3] = x_home.reshape(-1, 3)
        x4[MOL_ATOMS * nh:, :3] = halo.reshape(-1, 3)
        v4[: MOL_ATOMS * nh, :3] = v_home.reshape(-1, 3)
        self.ids_home = ids_home.clone()
        self.n_home, self.n_halo = nh, nl
        # exchange plan for the steps until the next partition, in atoms
        def atoms_of(mols_local):           # local molecule slots -> atom slots
            return (mols_local[:, None] * MOL_ATOMS + torch.arange(MOL_ATOMS, device=self.dev)[None, :]).reshape(-1)
        self.send_counts = [int(sd.numel()) * MOL_ATOMS for sd in send_slots]
        self.recv_counts = [int(hx.shape[0]) * MOL_ATOMS for hx in halo_x]
        self.send_idx = atoms_of(torch.cat(list(send_slots))).to(torch.int32).contiguous() if self.world > 1 else None
        self.send_buf = torch.zeros((sum(self.send_counts), 4), device=self.dev, dtype=torch.float32)
        self.recv_buf = torch.zeros((sum(self.recv_counts), 4), device=self.dev, dtype=torch.float32)
        self._keep = (x4, v4)               # alive until the load kernel has run
        torch.cuda.synchronize(self.dev)
        # degrees of freedom of rigid three-site molecules: the whole system's when the domains share one thermostat
        # factor (a domain's temperature read-out is then its share of the total and the shares add up), the domain's own otherwise
        self.ndf_global = 6.0 * self.nmol - (3.0 if self.nstcomm > 0 else 0.0)
        ndf = self.ndf_global if self.global_coupling else 6.0 * nh
        self.engine.dd_load(MOL_ATOMS * nh, MOL_ATOMS * (nh + nl), x4.data_ptr(), v4.data_ptr(), nh, ndf, self.density)

    def _home_state(self):
        """(n_home, 3, 3) positions and velocities of the home molecules, read from the handle."""
        self.engine.dd_stream_sync()
        xp, vp = self.engine.dd_state_pointers()
        na = MOL_ATOMS * self.n_home
        x = torch.as_tensor(_DevArray(xp, (na, 4)), device=self.dev)[:, :3].clone().reshape(self.n_home, MOL_ATOMS, 3)
        v = torch.as_tensor(_DevArray(vp, (na, 4)), device=self.dev)[:, :3].clone().reshape(self.n_home, MOL_ATOMS, 3)
        return x, v

    def _gather_all(self):
        x, v = self._home_state()
        pad = self.home_cap
        if self.n_home > pad:
            raise RuntimeError(f"rank {self.rank}: {self.n_home} home molecules exceed the gather slot {pad}")
        rec = torch.zeros((pad, 18), device=self.dev, dtype=torch.float32)
        rec[: self.n_home, :9] = x.reshape(self.n_home, 9); rec[: self.n_home, 9:] = v.reshape(self.n_home, 9)
        ids = torch.full((pad,), -1, device=self.dev, dtype=torch.int64)
        ids[: self.n_home] = self.ids_home
        recs = torch.cat(self.tr.all_gather(rec)); idss = torch.cat(self.tr.all_gather(ids))
        keep = idss >= 0
        recs, idss = recs[keep], idss[keep]
        return recs[:, :9].reshape(-1, 3, 3), recs[:, 9:].reshape(-1, 3, 3), idss

    def repartition(self):
        if self.pcoupl:
            # the cell has been rescaled since the last partition: halo margins in fractional coordinates follow it
            self.margins = fractional_margins(self.box, self.rlist_margin)
        if self.migration and self.world > 1:
            x, v = self._home_state()
            xh, vh, ih, halo_x, sends = migrate(self.tr, x, v, self.ids_home, self.box, self.dims, self.margins)
            halo_x = [h if r != self.rank else xh[:0] for r, h in enumerate(halo_x)]
            self._install_local(xh, vh, ih, halo_x, sends)
        else:
            x_all, v_all, ids_all = self._gather_all()
            own = owner_of(fractional(x_all[:, 0, :].double(), self.box), self.dims)
            self._install(x_all, v_all, ids_all, own)
        self.steps_since_partition = 0

    def _refresh_halo(self):
        if self.world == 1 or self.n_halo == 0 and sum(self.send_counts) == 0:
            return
        ns = sum(self.send_counts)
        if ns:
            self.engine.dd_pack(ns, self.send_idx.data_ptr(), self.send_buf.data_ptr())
        send = list(torch.split(self.send_buf, self.send_counts)); recv = list(torch.split(self.recv_buf, self.recv_counts))
        if self.ordered:
            with torch.cuda.stream(self.st_main):       # behind the pack, ahead of the unpack, in the main stream's order
                self.tr.exchange(send, recv)
        else:
            self.engine.dd_stream_sync()
            self.tr.exchange(send, recv)
            torch.cuda.synchronize(self.dev)
        if self.n_halo:
            self.engine.dd_unpack(MOL_ATOMS * self.n_home, MOL_ATOMS * self.n_halo, self.recv_buf.data_ptr())

    # ------------------------------------------------------------------ stepping
    def _exclusive(self):
        """With the in-process transport the handle is drained (all its streams idle) when the block ends, before the
        GPU is handed to another domain's thread; otherwise a no-op."""
        import contextlib
        if not getattr(self.tr, "serial", False):
            return contextlib.nullcontext()
        dd = self

        @contextlib.contextmanager
        def drained():
            try:
                yield
            finally:
                dd.engine.synchronize()
        return drained()

    def step(self, nsteps: int):
        import os
        dbg = bool(os.environ.get("MDX_DD_DEBUG"))
        def say(what):
            if dbg:
                print("[dd rank %d step %d] %s" % (self.rank, self.nsteps_done, what), flush=True)
        shared_tc = self.thermostat and self.global_coupling
        for _ in range(int(nsteps)):
            if shared_tc and self.nsteps_done == 0:
                # the factor of the first steps, from the kinetic energy of the whole system
                with self._exclusive():
                    self.engine.dd_kinetic(); self.engine.dd_stream_sync()
                self.tr.all_reduce_sum_(self.ekinh_t); torch.cuda.synchronize(self.dev)
                with self._exclusive():
                    self.engine.dd_tcouple(before_first_step=True)
                self.engine.dd_stream_sync()
            if self.steps_since_partition >= self.nstlist:
                say("repartition"); self.repartition()
            elif self.steps_since_partition > 0:
                say("halo refresh"); self._refresh_halo()
            slabs = self.mesh == "slabs"
            if self.ordered:
                say("begin"); self.engine.dd_step_begin()
                with torch.cuda.stream(self.st_pme):        # behind the spreading (slabs: the x pass), ahead of the transforms
                    if slabs:
                        say("plane sums"); self.tr.reduce_scatter_sum_(self.cgrid_t)
                    else:
                        say("grid sum"); self.tr.all_reduce_sum_(self.grid_t)
                if slabs:
                    say("mid"); self.engine.dd_step_mid()
                    with torch.cuda.stream(self.st_pme):    # behind this rank's planes, ahead of the x pass back
                        say("plane gather"); self.tr.all_gather_rows_(self.cgrid_t)
                say("end"); self.engine.dd_step_end()
            else:
                with self._exclusive():
                    say("begin"); self.engine.dd_step_begin()
                    if self.world > 1:
                        say("wait spread"); self.engine.dd_wait_spread()
                if self.world > 1:
                    if slabs:
                        say("plane sums"); self.tr.reduce_scatter_sum_(self.cgrid_t)
                    else:
                        say("grid sum"); self.tr.all_reduce_sum_(self.grid_t)
                    torch.cuda.synchronize(self.dev)
                if slabs:
                    with self._exclusive():
                        say("mid"); self.engine.dd_step_mid()
                        if self.world > 1:
                            self.engine.dd_wait_spread()
                    if self.world > 1:
                        say("plane gather"); self.tr.all_gather_rows_(self.cgrid_t)
                        torch.cuda.synchronize(self.dev)
                with self._exclusive():
                    say("end"); self.engine.dd_step_end()
            done = self.nsteps_done + 1
            if self.nstcalcenergy > 0 and self.nsteps_done % self.nstcalcenergy == 0:
                self._volume_at_energy = self.volume
            if self.pcoupl and self.nsteps_done % self.nstpcouple == 0:
                # the step left this domain's share of the pressure in the scalars: sum it, scale every domain alike, and
                # put the share back (energies() adds the shares up itself)
                pdisp = self.dispersion_correction()[1]
                if self.ordered:
                    with torch.cuda.stream(self.st_main):
                        share = self.pres_t.clone()
                        self.tr.all_reduce_sum_(self.pres_t)
                        if pdisp != 0.0:
                            self.pres_t[0::4] += pdisp
                    box = self.engine.dd_pcouple()          # (waits for the main stream: the new cell comes back to the host)
                    with torch.cuda.stream(self.st_main):
                        self.pres_t.copy_(share)
                else:
                    self.engine.dd_stream_sync()
                    share = self.pres_t.clone()
                    self.tr.all_reduce_sum_(self.pres_t)
                    if pdisp != 0.0:
                        self.pres_t[0::4] += pdisp
                    torch.cuda.synchronize(self.dev)
                    with self._exclusive():
                        box = self.engine.dd_pcouple()
                    self.pres_t.copy_(share); torch.cuda.synchronize(self.dev)
                self.box_np = np.asarray(box, dtype=np.float64)
                self.box = torch.tensor(self.box_np, dtype=torch.float64, device=self.dev)
                self.volume = float(abs(np.linalg.det(self.box_np)))
                self.density = 3 * self.nmol / self.volume
            if shared_tc and done % self.nsttcouple == 0:
                # the step left the home atoms' half-step kinetic energy in the scalars: sum, then one factor for all
                if self.ordered:
                    with torch.cuda.stream(self.st_main):
                        self.tr.all_reduce_sum_(self.ekinh_t)
                    self.engine.dd_tcouple()
                else:
                    self.engine.dd_stream_sync()
                    self.tr.all_reduce_sum_(self.ekinh_t); torch.cuda.synchronize(self.dev)
                    with self._exclusive():
                        self.engine.dd_tcouple()
            if self.nstcomm > 0 and done % self.nstcomm == 0:
                if self.ordered:
                    self.engine.dd_com(0)
                    with torch.cuda.stream(self.st_main):
                        self.tr.all_reduce_sum_(self.com_t)
                    self.engine.dd_com(1)
                else:
                    with self._exclusive():
                        self.engine.dd_com(0); self.engine.dd_stream_sync()
                    self.tr.all_reduce_sum_(self.com_t); torch.cuda.synchronize(self.dev)
                    with self._exclusive():
                        self.engine.dd_com(1)
            self.steps_since_partition += 1
            self.nsteps_done += 1

    def dispersion_correction(self):
        """(energy, pressure) of the long-range dispersion correction at the present volume, as the single handle's
        energy kernel evaluates them: the tail integral plus the constant the potential shift removes inside the cutoff
        (together twice the tail), and the tail's pressure for mode 2."""
        if not self.dispcorr:
            return 0.0, 0.0
        nn = float(self.natoms) ** 2
        etail = -(2.0 / 3.0) * np.pi * nn / self.volume * self.c6avg / self.rc ** 3
        pdisp = -(4.0 / 3.0) * np.pi * nn / self.volume ** 2 * self.c6avg / self.rc ** 3 * U.PRESFAC if self.dispcorr == 2 else 0.0
        return 2.0 * etail, pdisp

    def synchronize(self):
        self.engine.synchronize()

    def energies(self):
        """Energies, virial and pressure of the whole system at the last energy step (every rank gets the same numbers):
        the domains' partial sums added up. Needs nstcalcenergy > 0."""
        from . import engine as eng
        e = self.engine.energies()
        vir, pres = self.engine.virial()
        names = ["LJ", "COUL_SR", "COUL_RECIP", "COUL_EXCL", "KIN", "TEMP", "PRES"]
        vec = np.concatenate([[e[k] for k in names], vir.reshape(-1), pres.reshape(-1)])
        if not self.global_coupling:
            # a domain's temperature is over its own degrees of freedom: weight it back to a kinetic-energy share
            vec[5] *= 6.0 * self.n_home / self.ndf_global
        t = torch.tensor(vec, dtype=torch.float64, device=self.dev)
        self.tr.all_reduce_sum_(t); torch.cuda.synchronize(self.dev)
        vec = t.cpu().numpy()
        out = {k: float(vec[i]) for i, k in enumerate(names)}
        # (the frame is that of the last energy step; with pressure coupling on the same step the cell has been scaled
        # since: the correction is taken at the volume the step ran in)
        vol_now = self.volume
        self.volume = getattr(self, "_volume_at_energy", self.volume)
        edisp, pdisp = self.dispersion_correction()
        self.volume = vol_now
        out["DISPCORR"] = edisp
        out["POT"] = out["LJ"] + out["COUL_SR"] + out["COUL_RECIP"] + out["COUL_EXCL"] + edisp
        out["PRES"] += pdisp
        pres = vec[16:25].reshape(3, 3).copy()
        pres[0, 0] += pdisp; pres[1, 1] += pdisp; pres[2, 2] += pdisp
        out["VOLUME"] = self.volume
        return out, vec[7:16].reshape(3, 3), pres

    def gather_state(self):
        """Positions and velocities of the whole system, (N, 3) each, in the original molecule order (every rank)."""
        x_all, v_all, ids_all = self._gather_all()
        order = torch.argsort(ids_all)
        return x_all[order].reshape(-1, 3).cpu().numpy(), v_all[order].reshape(-1, 3).cpu().numpy()

    def close(self):
        self.engine.close()
