// Verlet cluster-pair list builder (Pall & Hess 2013 scheme, laid out for wave64).
//
// Replaces the pair search that `cutoff-scheme = Verlet`, `nstlist`, `rlist`
// (/root/reference/mdp/production.mdp:21-23, pull_short.mdp:16-18) switch on inside gmx mdrun.
//
// Layout: atoms are wrapped into the rectangular brick [0,ax) x [0,by) x [0,cz) that is a unit
// cell of every lower-triangular (GROMACS-convention) box, binned into xy columns, sorted along z
// inside a column (in LDS), cut into super-clusters of 64 atoms, and each super-cluster is split
// by a y-sort of its two z-halves and an x-sort of its four quarters into 8 clusters of 8 atoms.
// One wave builds the list of one i-super-cluster: for every lattice shift it tests bounding
// boxes, then all 8x8 atom pairs of each surviving cluster pair with one lane per pair, and emits
// (j-cluster, 8-bit i-cluster mask, exclusion-mask slot) entries plus work units of at most
// MDX_NB_CHUNK entries for the nonbonded kernel.
#include "mdx_internal.h"
#include <hip/hip_ext.h>
#include <cstring>

#define BIG_KEY 3.0e38f


// `order` (or null): the cluster order of the previous search over the same n atoms. Atoms move a fraction of a
// column during a list lifetime, so threads that walk the atoms in that order find a wave's 64 atoms in one or two
// columns and the loop of returning atomics below runs once or twice instead of once per column a wave of atoms in
// input order touches. Skipped when that search overflowed a capacity (its order may then be incomplete).
__global__ void k_bin(int n, const float4 *__restrict__ x, const DevState *__restrict__ d, int ncx, int ncy,
                      int *__restrict__ col_count, float4 *__restrict__ wrapped, int *__restrict__ atom_col,
                      int *__restrict__ atom_slot, const int *__restrict__ order)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int i = g;
    bool live = g < n;
    if (order && d->overflow == 0 && d->nsc > 0) {
        i = g < d->nsc * 64 ? order[g] : -1;
        live = i >= 0 && i < n;
        if (!live) i = 0;
    }
    const float *h = d->box.h;
    float4 p = live ? x[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int nc = (int)floorf(p.z / h[8]);
    p.x -= nc * h[6]; p.y -= nc * h[7]; p.z -= nc * h[8];
    int nb = (int)floorf(p.y / h[4]);
    p.x -= nb * h[3]; p.y -= nb * h[4];
    int na = (int)floorf(p.x / h[0]);
    p.x -= na * h[0];
    int cx = min(ncx - 1, max(0, (int)(p.x * ncx / h[0])));
    int cy = min(ncy - 1, max(0, (int)(p.y * ncy / h[4])));
    const int col = live ? cy * ncx + cx : -1;
    // neighbouring atoms mostly share a column: one returning atomic per distinct column per wave
    int slot = 0;
    unsigned long long todo = __ballot(live);
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lcol = __shfl(col, leader, 64);
        const unsigned long long same = __ballot(col == lcol) & todo;
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&col_count[lcol], __popcll(same));
        base = __shfl(base, leader, 64);
        if (col == lcol && live) slot = base + __popcll(same & below);
        todo &= ~same;
    }
    if (!live) return;
    p.w = __int_as_float(pack_shift(-na, -nb, -nc));
    wrapped[i] = p;
    atom_col[i] = col;
    atom_slot[i] = slot;
}

// One block of 1024: exclusive prefix sums of column atom counts and of column super-cluster counts (wave scans with
// lane shifts, the sixteen wave totals scanned by the first wave), the list counters reset, and the lattice shift
// vectors of the current box (MDX_MAX_SHIFTS <= 1024).
__global__ __launch_bounds__(1024) void k_colscan(int ncol, const int *__restrict__ col_count, int *__restrict__ col_astart,
                                                  int *__restrict__ col_scstart, DevState *__restrict__ d, int nsc_cap,
                                                  int nshift, const int3 *__restrict__ shift_ijk, float4 *__restrict__ shiftvec)
{
    __shared__ int wa[16], ws[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (ncol + blockDim.x - 1) / blockDim.x;
    const int b = t * per, e = min(ncol, b + per);
    int a = 0, s = 0;
    for (int c = b; c < e; c++) { a += col_count[c]; s += (col_count[c] + 63) >> 6; }
    int ia = a, is = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int va = __shfl_up(ia, o, 64), vs = __shfl_up(is, o, 64);
        if (lane >= o) { ia += va; is += vs; }
    }
    if (lane == 63) { wa[w] = ia; ws[w] = is; }
    __syncthreads();
    if (w == 0) {
        const int oa = lane < 16 ? wa[lane] : 0, os = lane < 16 ? ws[lane] : 0;
        int ta = oa, ts = os;
        for (int o = 1; o < 16; o <<= 1) {
            const int va = __shfl_up(ta, o, 64), vs = __shfl_up(ts, o, 64);
            if (lane >= o) { ta += va; ts += vs; }
        }
        if (lane < 16) { wa[lane] = ta - oa; ws[lane] = ts - os; }
        if (lane == 15) {
            col_astart[ncol] = ta; col_scstart[ncol] = ts;
            d->nsc = ts; d->ncj = 0; d->nwork = 0; d->nexcl = 1; d->ntiles = 0;
            if (ts > nsc_cap) d->overflow |= 16;
        }
    }
    if (t < 257) { d->cost_hist[t] = 0; d->cost_cursor[t] = 0; }
    if (t < nshift) {
        const float *h = d->box.h;
        const int3 sh = shift_ijk[t];
        shiftvec[t] = make_float4(sh.x * h[0] + sh.y * h[3] + sh.z * h[6], sh.y * h[4] + sh.z * h[7], sh.z * h[8], 0.0f);
    }
    __syncthreads();
    a = ia - a + wa[w]; s = is - s + ws[w];
    for (int c = b; c < e; c++) {
        col_astart[c] = a; col_scstart[c] = s;
        const int cnt = col_count[c];
        if (cnt > MDX_MAX_COL) d->overflow |= 8;
        a += cnt; s += (cnt + 63) >> 6;
    }
}

__global__ void k_fill(int n, const int *__restrict__ atom_col, const int *__restrict__ atom_slot,
                       const int *__restrict__ col_astart, int *__restrict__ col_atoms)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    col_atoms[col_astart[atom_col[i]] + atom_slot[i]] = i;
}

// Bitonic sort of aligned blocks of B (power of two) elements by key k[], ascending; carries
// three companion arrays. All threads of the block must call. The workgroup size is a multiple of 64, so element t
// and its partner t ^ j belong to the same wave whenever j < 64: a compare-exchange step with such a j, followed by
// another one, needs the wave's own LDS traffic ordered and no workgroup barrier (70 of the 80 steps of a column).
__device__ void bitonic_blocks(float *k, float *c1, float *c2, int *id, int P, int B)
{
    for (int kk = 2; kk <= B; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const int jnext = j > 1 ? (j >> 1) : kk;        // partner distance of the step that follows (if any)
            for (int t = threadIdx.x; t < P; t += blockDim.x) {
                const int u = t ^ j;
                if (u > t) {
                    const bool up = ((t & kk & (B - 1)) == 0);
                    const float a = k[t], b = k[u];
                    // ties broken by atom index for a deterministic order
                    const bool gt = (a > b) || (a == b && id[t] > id[u]);
                    if (gt == up) {
                        k[t] = b; k[u] = a;
                        float w = c1[t]; c1[t] = c1[u]; c1[u] = w;
                        w = c2[t]; c2[t] = c2[u]; c2[u] = w;
                        int q = id[t]; id[t] = id[u]; id[u] = q;
                    }
                }
            }
            if (j >= 64 || jnext >= 64) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_sortcol(int *__restrict__ col_count, const int *__restrict__ col_astart,
                                                 const int *__restrict__ col_scstart, const int *__restrict__ col_atoms,
                                                 const float4 *__restrict__ wrapped, const float4 *__restrict__ x,
                                                 const float2 *__restrict__ lj, const int *__restrict__ molid,
                                                 float4 *__restrict__ xq_s, float2 *__restrict__ lj_s, int *__restrict__ perm,
                                                 int *__restrict__ perm_src, int *__restrict__ shift_s, int *__restrict__ molid_s,
                                                 int *__restrict__ sorted_of, float4 *__restrict__ bb_lo, float4 *__restrict__ bb_hi,
                                                 float4 *__restrict__ sbb_lo, float4 *__restrict__ sbb_hi, int nsc_cap,
                                                 const float4 *__restrict__ v_home, const int *__restrict__ ljtype, int *__restrict__ ljt_s)
{
    __shared__ float kz[MDX_MAX_COL], ky[MDX_MAX_COL], kx[MDX_MAX_COL];
    __shared__ int id[MDX_MAX_COL];
    const int col = blockIdx.x;
    const int n = min(col_count[col], MDX_MAX_COL);
    if (n == 0) return;
    const int base = col_astart[col];
    const int nscol = (n + 63) >> 6;
    const int sc0 = col_scstart[col];
    // the column counters are left at zero for the binning of the next search (this kernel is their last reader)
    __syncthreads();
    if (threadIdx.x == 0) col_count[col] = 0;
    if (sc0 + nscol > nsc_cap) return;
    int P = 64;
    while (P < nscol * 64) P <<= 1;
    for (int t = threadIdx.x; t < P; t += blockDim.x) {
        if (t < n) {
            const int a = col_atoms[base + t];
            const float4 w = wrapped[a];
            kz[t] = w.z; ky[t] = w.y; kx[t] = w.x; id[t] = a;
        } else {
            kz[t] = BIG_KEY; ky[t] = BIG_KEY; kx[t] = BIG_KEY; id[t] = 0x7fffffff;
        }
    }
    __syncthreads();
    bitonic_blocks(kz, ky, kx, id, P, P);
    bitonic_blocks(ky, kz, kx, id, nscol * 64 <= P ? P : P, 32);   // each 32-half of a super-cluster by y
    bitonic_blocks(kx, ky, kz, id, P, 16);                          // each 16-quarter by x
    // emit
    const int ntot = nscol * 64;
    for (int t = threadIdx.x; t < ntot; t += blockDim.x) {
        const int g = sc0 * 64 + t;
        int a = id[t];
        int src = t;
        const bool filler = (a == 0x7fffffff);
        if (filler) {
            src = t & ~7;
            if (id[src] == 0x7fffffff) src = t & ~63;
        }
        const int as = id[src];
        const float4 w = wrapped[as];
        const float q = filler ? 0.0f : x[as].w;
        xq_s[g] = make_float4(w.x, w.y, w.z, q);
        // a filler carries no charge and no LJ well; the negative sign of its (negligible) half-sigma marks it for the pair counters
        lj_s[g] = filler ? make_float2(-1.0e-30f, 0.0f) : lj[as];
        ljt_s[g] = filler ? 0 : ljtype[as];
        perm[g] = filler ? -1 : a;
        perm_src[g] = as;
        shift_s[g] = __float_as_int(w.w);
        molid_s[g] = filler ? -1 : molid[as];
        if (!filler) sorted_of[a] = g;
    }
    __syncthreads();
    // cluster bounding boxes (fillers sit on a real atom of the same super-cluster, so a cluster
    // made only of fillers degenerates to a point inside the super-cluster)
    for (int c = threadIdx.x; c < nscol * 8; c += blockDim.x) {
        float lo[3] = { BIG_KEY, BIG_KEY, BIG_KEY }, hi[3] = { -BIG_KEY, -BIG_KEY, -BIG_KEY };
        for (int k = 0; k < 8; k++) {
            int t = c * 8 + k;
            if (id[t] == 0x7fffffff) { t = t & ~7; if (id[t] == 0x7fffffff) t = t & ~63; }
            lo[0] = fminf(lo[0], kx[t]); hi[0] = fmaxf(hi[0], kx[t]);
            lo[1] = fminf(lo[1], ky[t]); hi[1] = fmaxf(hi[1], ky[t]);
            lo[2] = fminf(lo[2], kz[t]); hi[2] = fmaxf(hi[2], kz[t]);
        }
        // domain decomposition: w = 1 marks a cluster of halo atoms (inverse mass 0) and fillers only; pairs of two
        // such clusters move no home atom and are left out of the list
        float halo = v_home ? 1.0f : 0.0f;
        if (v_home)
            for (int k = 0; k < 8; k++) {
                const int a = id[c * 8 + k];
                if (a != 0x7fffffff && v_home[a].w != 0.0f) halo = 0.0f;
            }
        bb_lo[sc0 * 8 + c] = make_float4(lo[0], lo[1], lo[2], halo);
        bb_hi[sc0 * 8 + c] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    for (int sc = threadIdx.x; sc < nscol; sc += blockDim.x) {
        float lo[3] = { BIG_KEY, BIG_KEY, BIG_KEY }, hi[3] = { -BIG_KEY, -BIG_KEY, -BIG_KEY };
        for (int k = 0; k < 64; k++) {
            const int t = sc * 64 + k;
            if (id[t] == 0x7fffffff) continue;
            lo[0] = fminf(lo[0], kx[t]); hi[0] = fmaxf(hi[0], kx[t]);
            lo[1] = fminf(lo[1], ky[t]); hi[1] = fmaxf(hi[1], ky[t]);
            lo[2] = fminf(lo[2], kz[t]); hi[2] = fmaxf(hi[2], kz[t]);
        }
        float halo = v_home ? 1.0f : 0.0f;
        if (v_home)
            for (int k = 0; k < 64; k++) {
                const int a = id[sc * 64 + k];
                if (a != 0x7fffffff && v_home[a].w != 0.0f) halo = 0.0f;
            }
        sbb_lo[sc0 + sc] = make_float4(lo[0], lo[1], lo[2], halo);
        sbb_hi[sc0 + sc] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
}

__device__ __forceinline__ float bb_dist2(float3 alo, float3 ahi, float4 blo, float4 bhi)
{
    const float dx = fmaxf(0.0f, fmaxf(alo.x - bhi.x, blo.x - ahi.x));
    const float dy = fmaxf(0.0f, fmaxf(alo.y - bhi.y, blo.y - ahi.y));
    const float dz = fmaxf(0.0f, fmaxf(alo.z - bhi.z, blo.z - ahi.z));
    return dx * dx + dy * dy + dz * dz;
}

struct ListArgs {
    const DevState *dro;
    DevState *d;
    int ncx, ncy;
    const int *col_scstart;
    const float4 *xq_s;
    const int *perm;
    const int *molid_s;
    const float4 *bb_lo, *bb_hi, *sbb_lo, *sbb_hi;
    const float4 *shiftvec;
    int nshift, central;
    const int *excl_start, *excl_list;
    NbEntry *cj;                 // MDX_CJ_PER_SC entries per i-super-cluster
    NbWork *work;                // MDX_WORK_PER_SC units per i-super-cluster
    int *sc_nwork;               // units in use per i-super-cluster
    unsigned char *exclmask;     // 64-byte masks, MDX_EXCL_PER_WAVE slots per (super-cluster, wave)
    float rlist;
    int chunk;                   // list entries per work unit (<= 64)
};

#define EBUF 768          // entries of one (super-cluster, shift) buffered in LDS

// One workgroup (4 waves) per i-super-cluster: the waves share the i-atoms in LDS, take the
// candidate j-super-clusters round-robin and append entries to one LDS buffer, which is flushed
// once per lattice shift so the entries of a (super-cluster, shift) stay contiguous.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_build_list(ListArgs A)
{
    MDX_STAMP(A.d, 17);
    __shared__ float4 xi_sh[64];
    __shared__ int orig_sh[64];
    __shared__ NbEntry ebuf[EBUF];
    __shared__ int excl_n[64], excl_lo[64], excl_hi[64];
    __shared__ unsigned excl_bits[128];                // per i-atom: 64-bit membership mask relative to excl_lo
    __shared__ unsigned excl_bloom[256];               // bit (j mod 8192) set for every partner j in those lists
    __shared__ int s_nent, s_start, s_w0, s_tiles;     // entries of the current shift; entries, units, tiles so far
    int wslots = 0;                                     // exclusion-mask slots used by this wave (wave-uniform)
    __shared__ float4 xj_sh[4][64];      // per wave: the 64 atoms of the candidate j-super-cluster
    __shared__ int origj_sh[4][64];
    __shared__ __align__(8) unsigned char ext_sh[4][64];   // per wave: exclusion bytes being transposed
    __shared__ int row_first[4][64], row_scb[4][64], row_total[4];   // per wave: flat candidate space of the grid rows
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sci = blockIdx.x;
    const int nsc = A.dro->nsc;
    if (sci >= nsc) return;                     // uniform for the whole workgroup
    const float rl2 = A.rlist * A.rlist;
    const float rbb = fmaxf(0.0f, A.rlist - 0.3f), rbb2 = rbb * rbb;
    const float *h = A.dro->box.h;
    const float wx = h[0] / A.ncx, wy = h[4] / A.ncy, cz = h[8];
    if (threadIdx.x < 64) {
        xi_sh[lane] = A.xq_s[sci * 64 + lane];
        orig_sh[lane] = A.perm[sci * 64 + lane];
    }
    if (threadIdx.x == 0) { s_nent = 0; s_start = 0; s_w0 = 0; s_tiles = 0; }
    {
        // exclusion lists of the 64 i-atoms, condensed once into LDS: index range, membership bit mask over
        // that range and a filter over all partners; the per-pair test below then reads registers and LDS only
        const int a = lane, oi = A.perm[sci * 64 + a];
        const int e0 = oi >= 0 ? A.excl_start[oi] : 0, e1 = oi >= 0 ? A.excl_start[oi + 1] : 0;
        excl_bloom[threadIdx.x] = 0u;
        if (threadIdx.x < 128) excl_bits[threadIdx.x] = 0u;
        if (wv == 0) { excl_n[a] = e1 - e0; excl_lo[a] = 0x7fffffff; excl_hi[a] = -1; }
        __syncthreads();
        int lo = 0x7fffffff, hi = -1;
        for (int k = e0 + wv; k < e1; k += 4) {
            const int j = A.excl_list[k];
            lo = min(lo, j); hi = max(hi, j);
            atomicOr(&excl_bloom[(j >> 5) & 255], 1u << (j & 31));
        }
        if (hi >= 0) { atomicMin(&excl_lo[a], lo); atomicMax(&excl_hi[a], hi); }
        __syncthreads();
        // lists spanning fewer than 64 indices (all but exotic topologies) become a bit mask over [lo, lo + 63]
        const int flo = excl_lo[a];
        if (excl_hi[a] - flo < 64)
            for (int k = e0 + wv; k < e1; k += 4) {
                const int m = A.excl_list[k] - flo;
                atomicOr(&excl_bits[a * 2 + (m >> 5)], 1u << (m & 31));
            }
    }
    const float4 slo4 = A.sbb_lo[sci], shi4 = A.sbb_hi[sci];
    const int ti = lane & 7, tj = lane >> 3;
    const float4 ilo4 = A.bb_lo[sci * 8 + ti], ihi4 = A.bb_hi[sci * 8 + ti];
    __syncthreads();
    float xi_r[8][3];
    // i-atom `lane` of the super-cluster: range, span and membership bits of its exclusion list
    // (an atom without exclusions has a range no index falls in)
    const int my_elo = excl_lo[lane];
    const unsigned my_espan = (unsigned)(excl_hi[lane] - my_elo);
    const unsigned my_eb0 = excl_bits[lane * 2], my_eb1 = excl_bits[lane * 2 + 1];
#pragma unroll
    for (int ci = 0; ci < 8; ci++) {
        const float4 xi = xi_sh[ci * 8 + ti];
        xi_r[ci][0] = xi.x; xi_r[ci][1] = xi.y; xi_r[ci][2] = xi.z;
    }

    for (int s = 0; s < A.nshift; s++) {
        const float4 S = A.shiftvec[s];
        const bool central = (s == A.central);
        const float3 slo = make_float3(slo4.x - S.x, slo4.y - S.y, slo4.z - S.z);
        const float3 shi = make_float3(shi4.x - S.x, shi4.y - S.y, shi4.z - S.z);
        if (shi.z + A.rlist < 0.0f || slo.z - A.rlist > cz) continue;
        const int cx0 = max(0, (int)floorf((slo.x - A.rlist) / wx)), cx1 = min(A.ncx - 1, (int)floorf((shi.x + A.rlist) / wx));
        const int cy0 = max(0, (int)floorf((slo.y - A.rlist) / wy)), cy1 = min(A.ncy - 1, (int)floorf((shi.y + A.rlist) / wy));
        if (cx0 > cx1 || cy0 > cy1) continue;
        const float3 ilo = make_float3(ilo4.x - S.x, ilo4.y - S.y, ilo4.z - S.z);
        const float3 ihi = make_float3(ihi4.x - S.x, ihi4.y - S.y, ihi4.z - S.z);
        // Candidate j-super-clusters. Columns are numbered x-fastest and a column's super-clusters are
        // contiguous, so the window [cx0,cx1] of one grid row is one contiguous index range: lanes fetch
        // the row ranges, a wave scan turns them into one flat candidate space, and the four waves
        // take its elements round-robin, 64 at a time (one global round trip per 256 candidates).
        const int nyw = cy1 - cy0 + 1;
        for (int r0 = 0; r0 < nyw; r0 += 64) {
            {
                const int r = r0 + lane;
                const bool lv = r < nyw;
                const int rowc = (cy0 + r) * A.ncx;
                const int rb = lv ? A.col_scstart[rowc + cx0] : 0, re = lv ? A.col_scstart[rowc + cx1 + 1] : 0;
                int incl = re - rb;
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
                __builtin_amdgcn_wave_barrier();
                row_first[wv][lane] = lv ? incl - (re - rb) : 0x7fffffff;
                row_scb[wv][lane] = rb;
                row_total[wv] = __shfl(incl, 63, 64);
                __builtin_amdgcn_wave_barrier();
            }
            const int T = row_total[wv];
            for (int it = 0; it * 256 < T; it++) {
                    const int fl = (it * 64 + lane) * 4 + wv;
                    bool ok = fl < T;
                    int l = 0;
                    for (int step = 32; step; step >>= 1) if (row_first[wv][l + step] <= fl) l += step;
                    const int scj_l = row_scb[wv][l] + (fl - row_first[wv][l]);
                    if (central) ok = ok && (scj_l >= sci);
                    if (ok) { const float4 jl = A.sbb_lo[scj_l]; ok = bb_dist2(slo, shi, jl, A.sbb_hi[scj_l]) < rl2 && !(slo4.w != 0.0f && jl.w != 0.0f); }
                    unsigned long long m = __ballot(ok);
                    while (m) {
                        const int b = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const int scj = __builtin_amdgcn_readlane(scj_l, b);
                        // cluster level: lane = (i-cluster ti, j-cluster tj); the atoms of the j-super-cluster
                        // are fetched in the same round trip as its cluster boxes
                        const float4 jlo = A.bb_lo[scj * 8 + tj], jhi = A.bb_hi[scj * 8 + tj];
                        const float4 xjv = A.xq_s[scj * 64 + lane];
                        const int ojv = A.perm[scj * 64 + lane];
                        const float bd2 = bb_dist2(ilo, ihi, jlo, jhi);
                        bool okc = bd2 < rl2 && !(ilo4.w != 0.0f && jlo.w != 0.0f);
                        if (central && scj == sci) okc = okc && (tj >= ti);
                        const unsigned long long cm = __ballot(okc);
                        // boxes this close hold a pair within rlist almost surely: listed without the atom test
                        // (a tile listed in vain costs the nonbonded kernel one skipped tile, never correctness)
                        const unsigned long long cs = __ballot(okc && bd2 < rbb2);
                        if (!cm) continue;
                        // j-atoms whose index has its bit in the filter over all exclusion partners of the i-atoms
                        const unsigned long long bh = __ballot(ojv >= 0 && ((excl_bloom[(ojv >> 5) & 255] >> (ojv & 31)) & 1u));
                        xj_sh[wv][lane] = xjv;
                        origj_sh[wv][lane] = ojv;
                        __builtin_amdgcn_wave_barrier();
                        for (int cjl = 0; cjl < 8; cjl++) {
                            const unsigned im = (unsigned)((cm >> (8 * cjl)) & 0xFFull);
                            if (!im) continue;
                            const int cjg = scj * 8 + cjl;
                            // atom level: lane = (i-atom ti, j-atom tj); the lane's eight i-atoms are in registers
                            const float4 xj = xj_sh[wv][cjl * 8 + tj];
                            const float xjs = xj.x + S.x, yjs = xj.y + S.y, zjs = xj.z + S.z;
                            const unsigned sure = (unsigned)((cs >> (8 * cjl)) & 0xFFull);
                            unsigned newim = sure, exb = 0xFF;
                            const unsigned test = im & ~sure;
                            if (test) {
#pragma unroll
                                for (int ci = 0; ci < 8; ci++) {
                                    if (!(test & (1u << ci))) continue;       // wave-uniform
                                    const float dx = xi_r[ci][0] - xjs, dy = xi_r[ci][1] - yjs, dz = xi_r[ci][2] - zjs;
                                    if (__any((dx * dx + dy * dy + dz * dz) < rl2)) newim |= (1u << ci);
                                }
                            }
                            // bit cleared in exb = topologically excluded pair (the nonbonded kernel gives those the
                            // Ewald correction and itself skips the lower half of a cluster's tile with itself).
                            // Looked up only when the filter has the bit of one of the cluster's j-atoms. For the lookup lane a is i-atom a against the eight j-atoms (registers
                            // only); the resulting bytes are transposed through LDS into the (ti, tj) lane layout.
                            if (newim && ((bh >> (8 * cjl)) & 0xFFull)) {
                                unsigned exj = 0;
#pragma unroll
                                for (int k = 0; k < 8; k++) {
                                    const int oj = origj_sh[wv][cjl * 8 + k];
                                    const unsigned dlt = (unsigned)(oj - my_elo);
                                    bool ex = dlt <= my_espan && ((((dlt & 32u) ? my_eb1 : my_eb0) >> (dlt & 31u)) & 1u);
                                    if (my_espan >= 64u && my_espan < 0x80000000u && dlt <= my_espan) {
                                        const int e0 = A.excl_start[orig_sh[lane]], ne = excl_n[lane];
                                        for (int q = 0; q < ne; q++) ex = ex || (A.excl_list[e0 + q] == oj);
                                    }
                                    if (ex && !(central && sci * 64 + lane == cjg * 8 + k)) exj |= 1u << k;
                                }
                                if (__any(exj != 0u)) {
                                    ext_sh[wv][ti * 8 + tj] = (unsigned char)exj;       // lane = i-atom tj * 8 + ti: stored as [ti][ci]
                                    __builtin_amdgcn_wave_barrier();
                                    const uint2 col = *reinterpret_cast<const uint2 *>(&ext_sh[wv][ti * 8]);
                                    unsigned cleared = 0;
#pragma unroll
                                    for (int ci = 0; ci < 8; ci++) {
                                        const unsigned byte = ((ci < 4 ? col.x : col.y) >> (8 * (ci & 3))) & 0xFFu;
                                        cleared |= ((byte >> tj) & 1u) << ci;
                                    }
                                    exb = ~cleared & 0xFFu;
                                    __builtin_amdgcn_wave_barrier();
                                }
                            }
                            if (!newim) continue;
                            const bool need = __any((exb & newim) != newim);
                            int slot = 0;
                            if (need) {
                                // mask slots come from a region owned by this wave: no returning atomic
                                if (wslots < MDX_EXCL_PER_WAVE) {
                                    slot = 1 + (sci * 4 + wv) * MDX_EXCL_PER_WAVE + wslots;
                                    wslots++;
                                    A.exclmask[(size_t)slot * 64 + lane] = (unsigned char)exb;
                                } else if (lane == 0) atomicOr(&A.d->overflow, 4);
                            }
                            if (lane == 0) {
                                const int idx = atomicAdd(&s_nent, 1);
                                if (idx < EBUF) { ebuf[idx].cj = cjg; ebuf[idx].mask = newim | ((unsigned)slot << 8); }
                                else atomicOr(&A.d->overflow, 32);
                            }
                        }
                    }
                }
            }
        __syncthreads();
        const int nent = min(s_nent, EBUF);
        if (nent) {
            const int chunk = A.chunk;
            const int nw = (nent + chunk - 1) / chunk;
            // entries and work units go to regions owned by this super-cluster: no global counters
            const int start = sci * MDX_CJ_PER_SC + s_start, w0 = sci * MDX_WORK_PER_SC + s_w0;
            if (s_start + nent <= MDX_CJ_PER_SC && s_w0 + nw <= MDX_WORK_PER_SC) {
                for (int k = threadIdx.x; k < nent; k += 256) A.cj[start + k] = ebuf[k];
                for (int k = threadIdx.x; k < nw; k += 256) {
                    NbWork w; w.sci = sci; w.cj_start = start + k * chunk;
                    w.cj_end = min(start + nent, w.cj_start + chunk);
                    // cost = number of 8x8 tiles of the unit, kept next to the shift index for the sort
                    int cost = 0;
                    for (int q = k * chunk; q < min(nent, (k + 1) * chunk); q++) cost += __popc(ebuf[q].mask & 0xFFu);
                    w.shift = s | (cost << 8);
                    A.work[w0 + k] = w;
                    atomicAdd(&s_tiles, cost);
                    atomicAdd(&A.d->cost_hist[min(256, cost)], 1);     // for the sort of the units (no value returned)
                }
            } else if (threadIdx.x == 0) atomicOr(&A.d->overflow, (s_start + nent > MDX_CJ_PER_SC) ? 1 : 2);
            __syncthreads();
            if (threadIdx.x == 0) {
                s_start += nent; s_w0 += nw; s_nent = 0;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        A.sc_nwork[sci] = min(s_w0, MDX_WORK_PER_SC);
        atomicAdd(&A.d->ncj, min(s_start, MDX_CJ_PER_SC));       // statistics only (no value returned)
        atomicAdd(&A.d->ntiles, (unsigned long long)s_tiles);
    }
    if (lane == 0 && wslots) atomicAdd(&A.d->nexcl, wslots);
}

// Rolling prune of the pair list (the dual-list scheme of `verlet-buffer-tolerance` with a long `nstlist`,
// /root/reference/mdp/npt_complex.mdp:29-32): the list above is built with the buffer of its whole lifetime; this
// kernel derives from it, from the coordinates of the current step, the list of the next few steps: a tile stays when
// one of its 64 atom pairs is within `rl2` now. One wave per work unit, lane (ti, tj) = (i-atom, j-atom) as in the
// nonbonded kernel; lane e keeps entry e and its new i-cluster mask, and the surviving entries are written compacted
// to the same place of the inner list (units keep their index, so the cost order of the outer list serves both).
__global__ void k_sort_work(DevState *__restrict__ d, const NbWork *__restrict__ work, const int *__restrict__ sc_nwork,
                            int *__restrict__ order, int which);

struct PruneArgs {
    const DevState *d;
    const NbWork *work; const int *order; const NbEntry *cj;
    const float4 *xq_s; const float4 *shiftvec;
    NbWork *work_out; NbEntry *cj_out;
    DevState *dw;
    float rl2;
};

__global__ __launch_bounds__(256) void k_prune_list(PruneArgs A)
{
    MDX_STAMP(A.d, 18);
    const int lane = threadIdx.x & 63, ti = lane & 7, tj = lane >> 3;
    const int nwork = A.d->nwork;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < nwork; k += nwaves) {
        const int u = A.order[k];
        const NbWork w = A.work[u];
        const int sci = __builtin_amdgcn_readfirstlane(w.sci);
        const int e0 = __builtin_amdgcn_readfirstlane(w.cj_start);
        const int count = __builtin_amdgcn_readfirstlane(w.cj_end) - e0;
        const float4 S = A.shiftvec[__builtin_amdgcn_readfirstlane(w.shift) & 0xFF];
        NbEntry myent; myent.cj = 0; myent.mask = 0;
        if (lane < count) myent = A.cj[e0 + lane];
        float xi[8], yi[8], zi[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float4 p = A.xq_s[(sci * 8 + c) * 8 + ti];
            xi[c] = p.x - S.x; yi[c] = p.y - S.y; zi[c] = p.z - S.z;
        }
        unsigned newmask = 0;           // of this lane's entry
        // eight entries at a time: their j-atoms are requested together (one memory round trip per eight entries)
        constexpr int NE = 8;
        for (int eb = 0; eb < count; eb += NE) {
            float4 pj[NE]; unsigned im[NE];
#pragma unroll
            for (int q = 0; q < NE; q++) {
                const int e = min(eb + q, count - 1);
                const int cj = __builtin_amdgcn_readlane(myent.cj, e);
                im[q] = eb + q < count ? ((unsigned)__builtin_amdgcn_readlane((int)myent.mask, e) & 0xFFu) : 0u;
                pj[q] = A.xq_s[cj * 8 + tj];
            }
#pragma unroll
            for (int q = 0; q < NE; q++) {
                unsigned keep = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if (!(im[q] & (1u << c))) continue;          // wave-uniform
                    const float dx = xi[c] - pj[q].x, dy = yi[c] - pj[q].y, dz = zi[c] - pj[q].z;
                    const float r2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (__builtin_amdgcn_ballot_w64(r2 < A.rl2) != 0ull) keep |= 1u << c;
                }
                if (lane == eb + q) newmask = keep;
            }
        }
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(newmask != 0u);
        const int pos = __popcll(alive & ((1ull << lane) - 1ull));
        if (newmask != 0u) {
            NbEntry o; o.cj = myent.cj; o.mask = (myent.mask & ~0xFFu) | newmask;
            A.cj_out[e0 + pos] = o;
        }
        // the unit's cost (tiles left) beside the shift index, and the histogram the sort of the units starts from
        int cost = __popc(newmask);
        for (int o = 32; o; o >>= 1) cost += __shfl_xor(cost, o, 64);
        if (lane == 0) {
            NbWork o = w; o.cj_end = e0 + __popcll(alive);
            o.shift = (w.shift & 0xFF) | (cost << 8);
            A.work_out[u] = o;
            atomicAdd(&A.dw->cost_hist_in[min(256, cost)], 1);
        }
    }
}

int mdx_launch_prune(mdx_system *s, hipStream_t st)
{
    if (s->nstprune <= 0 || !s->cj_in[0]) return MDX_OK;
    const int dst = s->inner_cur ^ 1;
    PruneArgs A;
    A.d = s->d; A.work = s->work; A.order = s->work_order; A.cj = s->cj; A.xq_s = s->xq_s; A.shiftvec = s->shiftvec;
    A.work_out = s->work_in[dst]; A.cj_out = s->cj_in[dst]; A.dw = s->d;
    A.rl2 = s->rlist_inner * s->rlist_inner;
    MDX_CHECK(s, hipMemsetAsync((char *)s->d + offsetof(DevState, cost_hist_in), 0, sizeof(int) * 2 * 258, st));
    // a few waves per SIMD beside the nonbonded kernel's: the kernel is a chain of dependent loads and short tests
    hipLaunchKernelGGL(k_prune_list, dim3(1024), dim3(256), 0, st, A);
    // units in the order of what pruning left of them; units left empty fall behind the count the nonbonded kernel reads
    hipLaunchKernelGGL(k_sort_work, dim3(32), dim3(1024), 0, st, s->d, (const NbWork *)s->work_in[dst], (const int *)s->sc_nwork,
                       s->order_in[dst], 1 + dst);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// Order the work units by descending tile count (counting sort on the cost stored next to the
// shift index). The nonbonded kernel hands unit k, k+W, k+2W, ... to wave k, so every wave gets
// one unit of each cost tier and the waves finish together.
// The list builder has already counted the units per cost (DevState::cost_hist). Every block here
// derives the first position of each cost from that histogram, counts its own slice of the sparse
// unit array (a fixed region per i-super-cluster, sc_nwork[] of it in use), reserves a range per
// cost with one returning atomic and scatters within the range through LDS.
// `which` 0: the list the builder has just written; 1, 2: the first or second pruned list (own histogram, own unit
// count, which leaves out the units that pruning has emptied: cost 0 sorts last).
__global__ __launch_bounds__(1024) void k_sort_work(DevState *__restrict__ d, const NbWork *__restrict__ work,
                                                    const int *__restrict__ sc_nwork, int *__restrict__ order, int which)
{
    __shared__ int first[257], mine[257];
    const int n = d->nsc * MDX_WORK_PER_SC;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int k0 = blockIdx.x * per, k1 = min(n, k0 + per);
    const int *hist = which ? d->cost_hist_in : d->cost_hist;
    int *cursor = which ? d->cost_cursor_in : d->cost_cursor;
    for (int c = threadIdx.x; c < 257; c += blockDim.x) { first[c] = hist[c]; mine[c] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        const int empty = first[0];
        for (int c = 256; c >= 0; c--) { const int h = first[c]; first[c] = run; run += h; }
        if (blockIdx.x == 0) { if (which) d->nwork_in[which - 1] = run - empty; else d->nwork = run; }
    }
    __syncthreads();
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x)
        if ((k % MDX_WORK_PER_SC) < sc_nwork[k / MDX_WORK_PER_SC]) atomicAdd(&mine[min(256, work[k].shift >> 8)], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < 257; c += blockDim.x) {
        const int m = mine[c];
        mine[c] = m ? first[c] + atomicAdd(&cursor[c], m) : 0;
    }
    __syncthreads();
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x)
        if ((k % MDX_WORK_PER_SC) < sc_nwork[k / MDX_WORK_PER_SC]) {
            const int pos = atomicAdd(&mine[min(256, work[k].shift >> 8)], 1);
            order[pos] = k;
        }
}

// Per step: refresh the sorted coordinates from the state and clear the sorted force buffer.
__global__ void k_prepare(int npad, DevState *__restrict__ d, const float4 *__restrict__ x,
                          const int *__restrict__ perm, const int *__restrict__ perm_src,
                          const int *__restrict__ shift_s, float4 *__restrict__ xq_s, float *__restrict__ f_s,
                          int nshift, const int3 *__restrict__ shift_ijk, float4 *__restrict__ shiftvec, int clear_mask)
{
    MDX_STAMP(d, 0);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    // (a step that accumulates energies: their accumulators are cleared here instead of by a launch of their own)
    if (clear_mask && g < 64) mdx_clear_accumulators(d, g, clear_mask & 1, clear_mask & 2, clear_mask & 4);
    const float *h = d->box.h;
    if (g < 8) d->nb_head[g * 32] = 0;       // unit queues of the nonbonded kernel
    if (g < nshift) {
        const int3 s = shift_ijk[g];
        shiftvec[g] = make_float4(s.x * h[0] + s.y * h[3] + s.z * h[6], s.y * h[4] + s.z * h[7], s.z * h[8], 0.0f);
    }
    if (g >= d->nsc * 64 || g >= npad) return;
    const int a = perm_src[g];
    const float4 p = x[a];
    int sa, sb, sc;
    unpack_shift(shift_s[g], sa, sb, sc);
    float4 o;
    o.x = p.x + sa * h[0] + sb * h[3] + sc * h[6];
    o.y = p.y + sb * h[4] + sc * h[7];
    o.z = p.z + sc * h[8];
    o.w = perm[g] >= 0 ? p.w : 0.0f;
    xq_s[g] = o;
    f_s[3 * g] = 0.0f; f_s[3 * g + 1] = 0.0f; f_s[3 * g + 2] = 0.0f;
}

// With `fork` the kernel's own completion signal is the fork event of the side streams (an event
// recorded by a separate marker costs the main stream several microseconds between two kernels).
int mdx_launch_prepare(mdx_system *s, bool fork, int clear_mask)
{
    { const int rc = mdx_wait_prune(s); if (rc) return rc; }
    const int nb = (s->npad + 255) / 256;
    if (fork) {
        hipExtLaunchKernelGGL(k_prepare, dim3(nb), dim3(256), 0, s->stream, nullptr, s->ev_fork, 0, s->npad, s->d,
                              (const float4 *)s->x, (const int *)s->perm, (const int *)s->perm_src, (const int *)s->shift_s,
                              s->xq_s, s->f_s, s->nshift, (const int3 *)s->shift_ijk, s->shiftvec, clear_mask);
        MDX_CHECK(s, hipGetLastError());
        MDX_CHECK(s, hipStreamWaitEvent(s->st_pme, s->ev_fork, 0));
        MDX_CHECK(s, hipStreamWaitEvent(s->st_bond, s->ev_fork, 0));
        return MDX_OK;
    }
    hipLaunchKernelGGL(k_prepare, dim3(nb), dim3(256), 0, s->stream, s->npad, s->d, s->x, s->perm, s->perm_src,
                       s->shift_s, s->xq_s, s->f_s, s->nshift, s->shift_ijk, s->shiftvec, clear_mask);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// Lattice shifts (i a + j b + k c) whose image of the brick can lie within rlist of the brick;
// only one of each +/- pair is kept (the central image is listed with cj >= ci).
static void build_shift_list(mdx_system *s, const double *box)
{
    s->h_shift_ijk.clear();
    const double ax = box[0], bx = box[3], by = box[4], cx = box[6], cy = box[7], cz = box[8];
    const double rl = s->rlist * 1.05 + 0.05;
    for (int k = 0; k <= 1; k++)
        for (int j = -2; j <= 2; j++)
            for (int i = -3; i <= 3; i++) {
                const bool half = (k > 0) || (k == 0 && j > 0) || (k == 0 && j == 0 && i >= 0);
                if (!half) continue;
                const double sx = i * ax + j * bx + k * cx, sy = j * by + k * cy, sz = k * cz;
                // gap between brick and shifted brick per dimension
                const double gx = fmax(0.0, fabs(sx) - ax), gy = fmax(0.0, fabs(sy) - by), gz = fmax(0.0, fabs(sz) - cz);
                if (gx * gx + gy * gy + gz * gz >= rl * rl) continue;
                if (i == 0 && j == 0 && k == 0) s->central_shift = (int)s->h_shift_ijk.size();
                s->h_shift_ijk.push_back(make_int3(i, j, k));
            }
    s->nshift = (int)s->h_shift_ijk.size();
}

int mdx_search(mdx_system *s)
{
    { const int rc = mdx_wait_prune(s); if (rc) return rc; }       // (a pruning launch still reads the lists rebuilt here)
    // host copy of the box (may lag the device by a few coupling steps; only used for grid counts)
    const double *box = s->h->boxd;
    build_shift_list(s, box);
    if (s->nshift > MDX_MAX_SHIFTS) { s->err = "too many lattice shifts (box too small for rlist)"; return MDX_ERR_UNSUPPORTED; }
    // the shift list changes only when the box shape does
    if (s->h_shift_ijk.size() != s->h_shift_sent.size() ||
        memcmp(s->h_shift_ijk.data(), s->h_shift_sent.data(), sizeof(int3) * s->h_shift_ijk.size()) != 0) {
        s->h_shift_sent = s->h_shift_ijk;
        MDX_CHECK(s, hipMemcpyAsync(s->shift_ijk, s->h_shift_sent.data(), sizeof(int3) * s->nshift, hipMemcpyHostToDevice, s->stream));
    }
    // column grid: super-cluster cells roughly cubic
    const double vol = box[0] * box[4] * box[8];
    // (a decomposed domain fills part of the box only: its columns are sized by the density of the whole system)
    const double dens = s->search_density > 0.0 ? s->search_density : s->n / vol;
    double side = cbrt(64.0 / dens);
    int ncx = (int)fmax(1.0, floor(box[0] / side)), ncy = (int)fmax(1.0, floor(box[4] / side));
    // keep the expected column population within the in-LDS sort capacity
    while (dens * vol / (ncx * ncy) > 0.55 * MDX_MAX_COL) { ncx++; ncy++; }
    while (ncx * ncy > s->ncol_cap) { if (ncx >= ncy) ncx--; else ncy--; }
    s->ncx = ncx; s->ncy = ncy;
    const int ncol = ncx * ncy;
    // k_sortcol zeroes the counters it has consumed: only the first search clears the array
    if (s->perm_n < 0) MDX_CHECK(s, hipMemsetAsync(s->col_count, 0, sizeof(int) * (s->ncol_cap + 1), s->stream));
    const int nbk = (s->n + 255) / 256;
    const bool reuse_order = s->perm_n == s->n && !(s->dbg & 0x8000000);
    hipLaunchKernelGGL(k_bin, dim3(reuse_order ? (std::max(s->n, s->npad) + 255) / 256 : nbk), dim3(256), 0, s->stream, s->n, s->x,
                       s->d, ncx, ncy, s->col_count, s->wrapped, s->atom_col, s->atom_slot,
                       (const int *)(reuse_order ? s->perm : nullptr));
    s->perm_n = s->n;
    hipLaunchKernelGGL(k_colscan, dim3(1), dim3(1024), 0, s->stream, ncol, s->col_count, s->col_astart, s->col_scstart,
                       s->d, s->npad / 64, s->nshift, s->shift_ijk, s->shiftvec);
    hipLaunchKernelGGL(k_fill, dim3(nbk), dim3(256), 0, s->stream, s->n, s->atom_col, s->atom_slot, s->col_astart, s->col_atoms);
    hipLaunchKernelGGL(k_sortcol, dim3(ncol), dim3(1024), 0, s->stream, s->col_count, s->col_astart, s->col_scstart,
                       s->col_atoms, s->wrapped, s->x, s->lj, s->molid, s->xq_s, s->lj_s, s->perm, s->perm_src, s->shift_s,
                       s->molid_s, s->sorted_of, s->bb_lo, s->bb_hi, s->sbb_lo, s->sbb_hi, s->npad / 64,
                       (const float4 *)(s->dd_on ? s->v : nullptr), (const int *)s->ljtype, s->ljt_s);
    ListArgs A;
    A.dro = s->d; A.d = s->d; A.ncx = ncx; A.ncy = ncy; A.col_scstart = s->col_scstart; A.xq_s = s->xq_s; A.perm = s->perm;
    A.molid_s = s->molid_s; A.bb_lo = s->bb_lo; A.bb_hi = s->bb_hi; A.sbb_lo = s->sbb_lo; A.sbb_hi = s->sbb_hi;
    A.shiftvec = s->shiftvec; A.nshift = s->nshift; A.central = s->central_shift; A.excl_start = s->excl_start;
    A.excl_list = s->excl_list; A.cj = s->cj; A.work = s->work; A.sc_nwork = s->sc_nwork;
    A.exclmask = s->exclmask; A.rlist = s->rlist;
    A.chunk = std::min(64, std::max(8, s->nb_chunk));
    const int nblk = s->npad / 64;
    hipLaunchKernelGGL(k_build_list, dim3(nblk), dim3(256), 0, s->stream, A);
    hipLaunchKernelGGL(k_sort_work, dim3(32), dim3(1024), 0, s->stream, s->d, (const NbWork *)s->work, (const int *)s->sc_nwork, s->work_order, 0);
    MDX_CHECK(s, hipGetLastError());
    s->steps_since_search = 0;
    s->inner_valid = false;         // the nonbonded kernel walks the new list itself until its first pruning has run
    return MDX_OK;
}
