// Fused LJ + Ewald real-space nonbonded kernel on the cluster-pair list (FP32 VALU).
//
// Replaces the short-range kernel that `coulombtype = PME`, `rcoulomb`, `rvdw`
// (/root/reference/mdp/production.mdp:24-27; pull_short.mdp:19-21) switch on in gmx mdrun.
//
// One wave = one work unit = (i-super-cluster of 8x8 atoms, lattice shift, <= 32 j-clusters).
// Lane (ti = lane & 7, tj = lane >> 3) holds i-atom ti of all 8 i-clusters in registers and, per
// j-cluster, j-atom tj: one wave instruction evaluates an 8x8 atom tile. i-forces accumulate in
// registers for the whole unit; the j-force is reduced over the 8 lanes sharing tj with DPP-style
// butterflies and added to the sorted force buffer with one 24-lane float atomic per j-cluster.
// Pair arithmetic: F/r = [4 eps (12 s6^2 - 6 s6) + qq (erfc(br) + 2/sqrt(pi) br exp(-b^2 r^2)) / r] / r^2
// with Lorentz-Berthelot sigma/eps from per-atom (sigma/2, 2 sqrt(eps)) so no type table gather.
#include "mdx_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include <cstddef>

struct NbArgs {
    const DevState *d;
    const int *nwork;            // number of work units (device memory: the list builder or the pruning kernel counts them)
    const NbWork *work;
    const int *order;            // work units by descending cost
    const NbEntry *cj;
    const unsigned char *exclmask;
    const float4 *xq_s;
    const float2 *lj_s;
    const float4 *shiftvec;
    float *f_s;
    DevState *dw;
    float rc2, beta, c_rf_shift; // c_rf_shift = erfc(beta rc)/rc
    int central;                 // index of the zero lattice shift
    int do_pairs, do_excl;       // evaluate listed pairs / the Ewald correction of excluded pairs
    const float4 *htab;          // MDX_GTAB_N entries (g, step of g, h, step of h), h(w) = erf(z)/z
    const float2 *gtab;          // MDX_GTAB_N entries (c0, c1): beta^3 g = c0 + c1 r^2 on the k-th interval of t = r^2 MDX_GTAB_N / rc^2
    float tab_scale;             // MDX_GTAB_N / rc^2
    int dyn_split;               // log2 of the parts a unit is cut into when handed out dynamically
    int dyn;                     // rounds' worth of units at the cheap end of the list handed out dynamically (0: static deal only)
    long long *wave_t;           // diagnostic: per wave (start clock, end clock, wall start, wall end, units, entries)
    const int *ljt_s;            // typed force kernel: 8 * atom type per slot of the cluster order
    const float2 *nbfp;          // ... nt x nt entries (12 C12, 6 C6)
    int nt;
};

// Element `idx` of an array of at most 4 GB through a 32-bit byte offset: the access then takes the uniform base from
// scalar registers and one lane offset register, with no 64-bit address arithmetic on the vector unit.
template <class T> __device__ __forceinline__ T *at32(T *base, unsigned idx) { return (T *)((char *)base + idx * (unsigned)sizeof(T)); }
template <class T> __device__ __forceinline__ const T *at32(const T *base, unsigned idx) { return (const T *)((const char *)base + idx * (unsigned)sizeof(T)); }
// ... with the offset in bytes: a wave-uniform part (scalar unit) plus a lane part that never changes is one vector add
template <class T> __device__ __forceinline__ T *atb(T *base, unsigned bytes) { return (T *)((char *)base + bytes); }
template <class T> __device__ __forceinline__ const T *atb(const T *base, unsigned bytes) { return (const T *)((const char *)base + bytes); }

// Cross-lane adds inside groups of 8 consecutive lanes as DPP moves (no LDS crossbar traffic).
__device__ __forceinline__ float dpp_quad_x1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_quad_x2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_row_ror8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_row_shl4(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xF, 0xF, true)); }
// Three values summed over the 8 lanes of a group in 8 instructions instead of 3 x 3 + a select: after the first exchange
// a lane keeps x or y (lane bit 0), after the second x, y or z (bit 1), so lanes 0, 1, 2 of the group end with the sums of
// x, y, z (b0, b1: the lane's bits 0 and 1).
__device__ __forceinline__ float sum8_xyz(float x, float y, float z, bool b0, bool b1)
{
    const float a = (b0 ? y : x) + dpp_quad_x1(b0 ? x : y);
    const float zz = z + dpp_quad_x1(z);
    const float r = (b1 ? zz : a) + dpp_quad_x2(b1 ? a : zz);
    return r + dpp_row_shl4(r);
}
__device__ __forceinline__ float sum8(float v)
{
    v += dpp_quad_x1(v);
    v += dpp_quad_x2(v);
    v += dpp_half_mirror(v);
    return v;
}

// g(w) = (erf(z)/z - 2/sqrt(pi) exp(-w))/w, w = z^2: rational fit of degree (5,5) on 0 <= w <= 16
// (tools/fit_pmecorr.py, max abs error 2.8e-8, 1.0e-7 evaluated in float32). With it the
// real-space Ewald force needs no exponential: F/r = qq (1/r^3 - beta^3 g(beta^2 r^2)).
__device__ __forceinline__ float mdx_pmecorr(float w)
{
    float p = -3.019193526e-07f;
    p = fmaf(p, w, 3.398092343e-05f);
    p = fmaf(p, w, 3.999912512e-04f);
    p = fmaf(p, w, 1.652836129e-02f);
    p = fmaf(p, w, -9.232622142e-03f);
    p = fmaf(p, w, 7.522528031e-01f);
    float q = 1.872059403e-04f;
    q = fmaf(q, w, 2.760296177e-03f);
    q = fmaf(q, w, 2.635980529e-02f);
    q = fmaf(q, w, 1.603129271e-01f);
    q = fmaf(q, w, 5.877281614e-01f);
    q = fmaf(q, w, 1.0f);
    return p * __builtin_amdgcn_rcpf(q);
}

// The entries of a work unit (one per lane) are fetched with one coalesced load and broadcast with v_readlane, the
// j-atoms of entry e+1 are requested before entry e is evaluated, and the force-only variant replaces exp + erfc by a
// tabulated Ewald correction.
// TAB: 0 analytic Ewald terms (no table: the fallback for a handle without Ewald tables and the cross-check of the
// tabulated kernels); 1 (force kernel) the Ewald correction g from the table `gtab` in LDS; 2 (energy kernel) g and
// h = erf(z)/z from a float4 table in LDS. With a table the LJ parameters of the i-atoms live in LDS too (`ilj`, 64
// float2 per wave) instead of 16 registers.
// NI: i-clusters of the super-cluster held in registers at a time; NI = 4 walks the unit's entries twice (36 registers
// fewer, j-atoms loaded and j-forces reduced twice): the energy kernel of a decomposed run, whose pair weights need them.
// WGT: pairs carry the weights 1, 1/2, 0 in energies and virial (a domain of a decomposed run: pairs with copies of other
// domains' atoms); a compile-time switch, because the handful of extra instructions per tile costs the energy kernel a
// third of its speed (304 against 227 us) and a single-domain run never needs them.
// TIMED: per-wave clocks and tile counters for mdx_debug_wave_times.
template <bool ENERGY, int TAB, int NI = 8, bool WGT = false, bool COUNT = true, bool TIMED = false, bool TYPED = false>
__device__ __forceinline__ void nb_body(const NbArgs &A, const float2 *gtab = nullptr, float2 *ilj = nullptr, const float2 *ljtab = nullptr)
{
    static_assert(!TYPED || (!ENERGY && TAB == 1), "Lennard-Jones by type: the tabulated force kernel only");
    MDX_STAMP_WAVES(A.dw, ENERGY ? 2 : 1);
    static_assert(TAB == 0 || TAB == (ENERGY ? 2 : 1), "force kernels take the float2 table, energy kernels the float4 table");
    constexpr bool ILDS = TAB != 0;
    constexpr bool XM = !ENERGY && TAB == 1;      // execution-masked plain tile (see the tile below)
    constexpr bool XME = ENERGY && TAB == 2 && !WGT;      // ... of the energy kernel
    // Virial of the energy kernel from position-force sums instead of nine instructions per pair: within a work unit
    // sum_pairs (x_i - x_j) (x) f_ij = sum_i (x_i - O) (x) F_i + sum_j (x_j - O) (x) F_j for any origin O (the unit's forces
    // add up to zero), and F_i (per unit) and F_j (per entry) are formed for the force accumulation anyway. Not with pair
    // weights (a decomposed run), which differ from pair to pair.
    constexpr bool PFV = ENERGY && !WGT;
    long long t_start = 0, w_start = 0; int n_units = 0, n_entries = 0, n_tested = 0, n_eval = 0;
    if (TIMED) { t_start = clock64(); w_start = wall_clock64(); }
    const int lane = threadIdx.x & 63;
    const int ti = lane & 7, tj = lane >> 3;
    const unsigned tj16 = 16u * tj, tj4 = 4u * tj, fj_off = 4u * (3 * tj + ti);       // the lane's byte offsets inside a j-cluster: coordinates, parameters, forces
    const int nwork = *A.nwork;
    const int wpb = blockDim.x >> 6;          // waves per workgroup
    const int nwaves = gridDim.x * wpb;
    float elj = 0.0f, ecoul = 0.0f, eexcl = 0.0f;
    unsigned int npair = 0, nlj = 0;           // wave-uniform: lane masks counted on the scalar unit
    const bool do_pairs = A.do_pairs != 0, do_excl = A.do_excl != 0;
    float vxx = 0.0f, vxy = 0.0f, vxz = 0.0f, vyy = 0.0f, vyz = 0.0f, vzz = 0.0f;
    float pvj[3] = { 0.0f, 0.0f, 0.0f }, pvi[3] = { 0.0f, 0.0f, 0.0f };       // PFV: row a of the tensor; the column is the force component the lane holds
    const float rc2 = A.rc2, beta = A.beta, inv_rc2 = 1.0f / A.rc2;
    float scale_v = A.tab_scale;
    if (XM || XME) asm volatile("v_mov_b32 %0, %1" : "=v"(scale_v) : "s"(A.tab_scale));    // (an operand from a scalar register halves the instruction's rate)
    const float beta2 = beta * beta, beta3 = beta2 * beta;
    const float nb2l2e = -beta2 * 1.4426950408889634f;
    // Work units are sorted by descending cost. The expensive ones are dealt statically in snake order
    // over the sorted list (balanced per wave, no queue traffic). A SIMD issues for its oldest wave first,
    // so its waves finish far apart and a purely static deal leaves the SIMD to one or two waves for
    // the last third of the launch: the cheap end of the list is therefore handed out dynamically, from
    // eight interleaved queues (wave index mod 8; heads on separate cache lines), and the ticket for the
    // next unit is requested before the loads of the current one so that its latency is covered.
    const int wave_id = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int q = wave_id & 7;
    const int nstatic = A.dyn > 0 ? max(1, nwork / nwaves - A.dyn) : (nwork + nwaves - 1) / nwaves;
    // snake order over the cost-sorted units: even rounds ascend, odd rounds descend
    auto static_unit = [&](int r) { return (r & 1) ? (r + 1) * nwaves - 1 - wave_id : r * nwaves + wave_id; };
    const int split = A.dyn_split, pmask = (1 << split) - 1;
    const int dyn_base = nstatic * nwaves + q;
    int *head = &A.dw->nb_head[q * 32];
    // Two-deep pipeline over the units of this wave: the ticket of the unit after the next and the
    // header of the next unit are requested before the current unit is evaluated, so a new unit starts
    // with its header in registers (a unit index >= nwork means "none").
    int unit = static_unit(0), part = 0, unit_n = nwork, part_n = 0;
    if (1 < nstatic) unit_n = static_unit(1);
    else if (A.dyn > 0) {
        int t0 = 0;
        if (lane == 0) t0 = atomicAdd(head, 1);
        t0 = __builtin_amdgcn_readfirstlane(t0);
        unit_n = dyn_base + 8 * (t0 >> split); part_n = t0 & pmask;
    }
    NbWork w; w.sci = 0; w.shift = 0; w.cj_start = 0; w.cj_end = 0;
    if (unit < nwork) w = A.work[A.order[unit]];
    for (int slot = 0; unit < nwork; slot++) {
        int tk = 0;
        if (A.dyn > 0 && slot + 2 >= nstatic && lane == 0) tk = atomicAdd(head, 1);
        NbWork w_n = w;
        if (unit_n < nwork) w_n = A.work[A.order[unit_n]];
        const int sci = __builtin_amdgcn_readfirstlane(w.sci);
        int e0 = __builtin_amdgcn_readfirstlane(w.cj_start);
        int count = __builtin_amdgcn_readfirstlane(w.cj_end) - e0;
        if (slot >= nstatic) {
            // a dynamic ticket is one of 2^split equal parts of a unit: the finer the grain, the closer the waves finish
            const int c0 = (count * part) >> split, c1 = (count * (part + 1)) >> split;
            e0 += c0; count = c1 - c0;
        }
        if (count > 0) {
        const int shift_idx = __builtin_amdgcn_readfirstlane(w.shift) & 0xFF;
        if (TIMED) { n_units++; n_entries += count; }
        const bool is_central = shift_idx == A.central;
        const float4 S = A.shiftvec[shift_idx];
        // origin of the unit's position-force sums: the first atom of the super-cluster, shifted (wave-uniform)
        float4 O4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (PFV) { O4 = A.xq_s[sci * 64]; O4.x -= S.x; O4.y -= S.y; O4.z -= S.z; }
        NbEntry myent; myent.cj = 0; myent.mask = 0;
        if (lane < count) myent = A.cj[e0 + lane];
        // (wave-uniform masks over the unit's entries, lane e for entry e)
        const bool general_e = (myent.mask >> 8) != 0u || !do_pairs || (is_central && (myent.cj >> 3) == sci);
        const unsigned long long general_bits = __builtin_amdgcn_ballot_w64(lane < count && general_e);
        const unsigned long long plain_bits = __builtin_amdgcn_ballot_w64(lane < count && !general_e);
#pragma unroll 1
        for (int half = 0; half < 8 / NI; half++) {
        const int c0 = half * NI;
        float xi[NI], yi[NI], zi[NI], qi[NI], hsi[ILDS ? 1 : NI], sei[ILDS ? 1 : NI];
        float2 *myilj = ILDS ? ilj + (threadIdx.x >> 6) * 64 : nullptr;
        if (ILDS) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the previous pass has read its values
        float fix[NI], fiy[NI], fiz[NI];
        // (one load and one LDS store per wave: lane l brings the pair of the l-th i-atom)
        int lio[TYPED ? NI : 1];         // TYPED: byte offset of the i-atom's row of the type-pair table
        if (!TYPED && ILDS && lane < NI * 8) myilj[lane] = *at32(A.lj_s, (unsigned)((sci * 8 + c0) * 8 + lane));
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const int ia = (sci * 8 + c0 + c) * 8 + ti;
            const float4 p = *at32(A.xq_s, (unsigned)ia);
            xi[c] = p.x - S.x; yi[c] = p.y - S.y; zi[c] = p.z - S.z;
            qi[c] = p.w * MDX_ONE_4PI_EPS0;
            if (!ILDS) { const float2 l = A.lj_s[ia]; hsi[c] = l.x; sei[c] = l.y; }
            if (TYPED) lio[c] = *at32(A.ljt_s, (unsigned)ia) * A.nt;
            fix[c] = 0.0f; fiy[c] = 0.0f; fiz[c] = 0.0f;
        }
        // (LDS operations of one wave execute in order; the fence only keeps the compiler from moving the reads of
        // other lanes' values above the writes)
        if (ILDS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // Software pipeline over the entries: j-atoms and exclusion mask are requested one entry ahead, into two register
        // sets used in turn (the loop is unrolled by two by hand, so that no set is copied into the other).
        // (unsigned element indices: a uniform base plus a zero-extended 32-bit lane offset needs no 64-bit address arithmetic)
        struct JSet { int cj; float4 pj; float2 lj; unsigned mk; unsigned ex; };
        auto fetch = [&](auto general_tag, JSet &J, const int e) {
            J.cj = __builtin_amdgcn_readlane(myent.cj, e);
            J.pj = *atb(A.xq_s, (unsigned)J.cj * 128u + tj16);
            if (TYPED) J.lj = make_float2(__int_as_float(*atb(A.ljt_s, (unsigned)J.cj * 32u + tj4)), 0.0f);      // byte offset inside a table row
            else J.lj = *atb(A.lj_s, (unsigned)J.cj * 64u + 2u * tj4);
            J.mk = (unsigned)__builtin_amdgcn_readlane((int)myent.mask, e);
            J.ex = 0xFFu;
            // (plain entries have no mask slot and never look at the exclusion bits)
            if (decltype(general_tag)::value && (J.mk >> 8)) J.ex = A.exclmask[(J.mk >> 8) * 64u + (unsigned)lane];
        };
        auto entry = [&](auto general_tag, const JSet &J, const JSet &N) {
            constexpr bool general = decltype(general_tag)::value;
            const int cj = J.cj;
            const float4 pj = J.pj;
            const float2 lj = J.lj;
            const unsigned mk = J.mk;
            const unsigned ex = J.ex;
            const unsigned imask = ((mk & 0xFFu) >> c0) & ((1u << NI) - 1u);
            const unsigned slot = mk >> 8;
            // `live` bit c: this lane's pair with i-cluster c is evaluated at all. In the tile of a
            // cluster with itself only tj > ti is (each pair once, no self pair); listed pairs need
            // do_pairs, excluded pairs need do_excl.
            unsigned live = 0xFFu, exh = 0xFFu;
            if (general) {          // plain entries never look at these
                live = do_pairs ? ex : 0u;
                if (do_excl) live |= ~ex & 0xFFu;
                if (is_central && (cj >> 3) == sci && tj <= ti) live &= ~(1u << (cj & 7));
                live >>= c0;
                exh = ex >> c0;
            }
            float fjx = 0.0f, fjy = 0.0f, fjz = 0.0f;
            bool did = false;           // wave-uniform: some tile of this entry had a pair inside the cutoff
            // One 8x8 tile: i-cluster c against the j-cluster. GENERAL tiles look at the exclusion and
            // liveness bits; the others (no mask slot, not a cluster with itself: the large majority, and
            // a wave-uniform property of the entry) are plain cutoff tests, about ten VALU instructions
            // shorter. The 1e-12 inside r2 keeps a filler atom, which sits exactly on a real one, finite.
            auto tile = [&](auto general_tag, const int c) {
                constexpr bool GENERAL = decltype(general_tag)::value;
                const float dx = xi[c] - pj.x, dy = yi[c] - pj.y, dz = zi[c] - pj.z;
                const float r2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, 1.0e-12f)));
                if (XM && !GENERAL) {
                    // Plain tile of the force-only kernel, written for the vector unit's issue costs on gfx950 (measured,
                    // tools/micro/valu_ops2.hip: add / mul / fma / and on vector registers take a SIMD ~2.3 clocks, compares,
                    // conversions, shifts, selects and anything with a scalar-register operand ~4.4, v_rsq_f32 8.2 and a select
                    // on vcc 23): the lanes inside the cutoff run the body under the execution mask (no select of the
                    // force at the end), the table offset comes out of the mantissa of r^2 scale + 2^20 (one fma and one
                    // and instead of multiply, convert, and, shift), the table holds its lines in r^2 itself, and the scale
                    // sits in a vector register.
                    const bool in_rc = r2 < rc2;
                    if (TIMED) n_tested++;
                    // the list is built with a buffer: an 8x8 tile none of whose pairs is inside the
                    // cutoff right now costs only the distance test (wave-uniform skip)
                    if (__builtin_amdgcn_ballot_w64(in_rc) == 0ull) return;
                    if (TIMED) n_eval++;
                    did = true;
                    if (in_rc) {
                        const float rinv = __frsqrt_rn(r2);
                        const float rinv2 = rinv * rinv;
                        // floor(t) * 8 for t = r^2 scale < 4096: at 2^20 a float's last bit is 1/8, so the sum's mantissa holds t in
                        // eighths; the 1/16 taken off turns round-to-nearest into rounding down (a t within rounding of an
                        // integer may land in the interval below, whose line meets the next one there)
                        const float u = fmaf(r2, scale_v, 1048576.0f - 0.0625f);
                        const float2 te = *(const float2 *)((const char *)gtab + (__float_as_uint(u) & 0x7FF8u));
                        float fs;
                        if (TYPED) {
                            // (12 C12 r^-6 - 6 C6) r^-6 from the type pair's table entry: five instructions where combining
                            // half-sigmas and well depths per pair takes nine
                            const float2 cc = *(const float2 *)((const char *)ljtab + (lio[TYPED ? c : 0] + __float_as_int(lj.x)));
                            const float rinv6 = rinv2 * rinv2 * rinv2;
                            fs = rinv6 * fmaf(cc.x, rinv6, -cc.y);
                        } else {
                            const float2 li = myilj[c * 8 + ti];
                            const float sg = li.x + lj.x;
                            const float e4 = fabsf(li.y) * fabsf(lj.y);
                            const float s2 = sg * sg * rinv2;
                            const float s6 = s2 * s2 * s2;
                            const float t6 = e4 * s6;
                            fs = t6 * fmaf(12.0f, s6, -6.0f);
                        }
                        const float qq = qi[c] * pj.w;
                        fs = fmaf(qq, rinv, fs) * rinv2;
                        const float g = fmaf(r2, te.y, te.x);
                        fs = fmaf(-qq, g, fs);
                        fix[c] = fmaf(dx, fs, fix[c]); fiy[c] = fmaf(dy, fs, fiy[c]); fiz[c] = fmaf(dz, fs, fiz[c]);
                        fjx = fmaf(-dx, fs, fjx); fjy = fmaf(-dy, fs, fjy); fjz = fmaf(-dz, fs, fjz);
                    }
                    return;
                }
                if (XME && !GENERAL) {
                    // Plain tile of the energy kernel, written like the force kernel's: under the execution mask, the table
                    // (16-byte entries: lines of beta^3 g and beta h in r^2) addressed from the mantissa of r^2 scale + 2^19.
                    const bool in_rc = r2 < rc2;
                    if (__builtin_amdgcn_ballot_w64(in_rc) == 0ull) return;
                    did = true;
                    const float2 li = myilj[c * 8 + ti];
                    if (COUNT) {
                        // statistics: pairs of real atoms (a filler carries a negative half-sigma) inside the cutoff
                        const bool counted = in_rc && li.x >= 0.0f && lj.x >= 0.0f;
                        npair += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(counted));
                        nlj += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(counted && li.y * lj.y != 0.0f));
                    }
                    if (in_rc) {
                        const float rinv = __frsqrt_rn(r2);
                        const float rinv2 = rinv * rinv;
                        const float u = fmaf(r2, scale_v, 524288.0f - 0.03125f);
                        const float4 te = *(const float4 *)((const char *)gtab + (__float_as_uint(u) & 0xFFF0u));
                        const float sg = li.x + lj.x;
                        const float e4 = li.y * lj.y;          // (no copies of other domains' atoms here: both factors >= 0)
                        const float sg2 = sg * sg;
                        const float s2 = sg2 * rinv2;
                        const float s6 = s2 * s2 * s2;
                        const float t6 = e4 * s6;
                        float fs = t6 * fmaf(12.0f, s6, -6.0f);
                        const float qq = qi[c] * pj.w;
                        fs = fmaf(qq, rinv, fs) * rinv2;
                        fs = fmaf(-qq, fmaf(r2, te.y, te.x), fs);
                        // erfc(beta r)/r - shift = 1/r - beta h - shift
                        ecoul = fmaf(qq, (rinv - fmaf(r2, te.w, te.z)) - A.c_rf_shift, ecoul);
                        const float c2 = sg2 * inv_rc2;
                        const float c6 = c2 * c2 * c2;
                        elj = fmaf(t6, s6 - 1.0f, elj);
                        elj = fmaf(-e4 * c6, c6 - 1.0f, elj);
                        fix[c] = fmaf(dx, fs, fix[c]); fiy[c] = fmaf(dy, fs, fiy[c]); fiz[c] = fmaf(dz, fs, fiz[c]);
                        fjx = fmaf(-dx, fs, fjx); fjy = fmaf(-dy, fs, fjy); fjz = fmaf(-dz, fs, fjz);
                    }
                    return;
                }
                bool within = r2 < rc2;
                if (GENERAL) within = within && ((live >> c) & 1u);
                if (TIMED) n_tested++;
                if (__builtin_amdgcn_ballot_w64(within) == 0ull) return;
                if (TIMED) n_eval++;
                did = true;
                const bool listed = GENERAL ? (((exh >> c) & 1u) != 0u) : true;   // false: excluded pair, Ewald correction only
                const float rinv = __frsqrt_rn(r2);
                const float rinv2 = rinv * rinv;
                float hs_c = 0.0f, se_c = 0.0f;
                if (TYPED) { }
                else if (ILDS) { const float2 li = myilj[c * 8 + ti]; hs_c = li.x; se_c = li.y; } else { hs_c = hsi[c]; se_c = sei[c]; }
                const float sg = hs_c + lj.x;
                // (a negative sign on 2 sqrt(eps), zero included, marks an atom that another domain owns: domain decomposition)
                const float e4 = fabsf(se_c) * fabsf(lj.y);
                const float s2 = sg * sg * rinv2;
                const float s6 = s2 * s2 * s2;
                const float t6 = e4 * s6;
                float fs = t6 * fmaf(12.0f, s6, -6.0f);
                if (TYPED) {
                    const float2 cc = *(const float2 *)((const char *)ljtab + (lio[TYPED ? c : 0] + __float_as_int(lj.x)));
                    const float rinv6 = rinv2 * rinv2 * rinv2;
                    fs = rinv6 * fmaf(cc.x, rinv6, -cc.y);
                }
                const float qq = qi[c] * pj.w;
                float erfc_br = 0.0f, e_ex = 0.0f;
                if (ENERGY && TAB == 2) {
                    // g and h(w) = erf(z)/z from one LDS table entry: erfc(beta r)/r = 1/r - beta h, no exponential
                    // (entry: the lines of beta^3 g and beta h in r^2 on the interval)
                    const float4 te = ((const float4 *)gtab)[(unsigned)(r2 * A.tab_scale) & (MDX_GTAB_N - 1)];
                    const float g = fmaf(r2, te.y, te.x), h = fmaf(r2, te.w, te.z);
                    fs = fmaf(qq, rinv, fs) * rinv2;
                    if (GENERAL && slot) fs = listed ? fs : 0.0f;
                    fs = fmaf(-qq, g, fs);
                    erfc_br = fmaf(-h, r2 * rinv, 1.0f);        // erfc(beta r) = 1 - beta r h
                    e_ex = -qq * h;
                } else if (ENERGY) {
                    const float br = beta * r2 * rinv;
                    const float ex2 = __builtin_amdgcn_exp2f(nb2l2e * r2);
                    erfc_br = ex2 * mdx_erfcx_poly(br);
                    fs = fmaf(qq * rinv, fmaf(MDX_2_SQRTPI * br, ex2, erfc_br), fs);
                    fs *= rinv2;
                    if (GENERAL && slot) {
                        // tiles that hold excluded pairs (wave-uniform test): E = -qq erf(br)/r through the
                        // cancellation-free g: erf(z)/z = g w + 2/sqrt(pi) exp(-w)
                        const float w_ = beta2 * r2;
                        const float g = mdx_pmecorr(w_);
                        fs = listed ? fs : -qq * beta3 * g;
                        e_ex = -qq * beta * fmaf(g, w_, MDX_2_SQRTPI * ex2);
                    }
                } else {
                    // (t6 (12 s6 - 6) + qq/r) / r^2 - qq beta^3 g; an excluded pair keeps only the last term
                    fs = fmaf(qq, rinv, fs) * rinv2;
                    if (GENERAL && slot) fs = listed ? fs : 0.0f;      // wave-uniform: few tiles hold an excluded pair
                    if (TAB == 1) {
                        // g(beta^2 r^2) by linear interpolation in a table uniform in r^2 (entry: the interval's line c0 + c1 r^2;
                        // the table holds beta^3 g): the look-up runs on the LDS pipe next to the LJ arithmetic and replaces a
                        // dozen VALU instructions
                        const float2 te = gtab[(unsigned)(r2 * A.tab_scale) & (MDX_GTAB_N - 1)];
                        fs = fmaf(-qq, fmaf(r2, te.y, te.x), fs);
                    } else {
                        fs = fmaf(-qq * beta3, mdx_pmecorr(beta2 * r2), fs);
                    }
                }
                fs = within ? fs : 0.0f;
                const float fx = fs * dx, fy = fs * dy, fz = fs * dz;
                fix[c] += fx; fiy[c] += fy; fiz[c] += fz;
                fjx -= fx; fjy -= fy; fjz -= fz;
                if (ENERGY) {
                    // A pair between an atom of this domain and a copy of another domain's atom is evaluated by both
                    // owners: each takes half of its energy and virial; a pair of two copies belongs to others entirely.
                    const float wgt = WGT ? 1.0f - 0.5f * (float)((__float_as_uint(se_c) >> 31) + (__float_as_uint(lj.y) >> 31)) : 1.0f;
                    if (COUNT) {
                        // statistics: pairs of real atoms (a filler carries a negative half-sigma) inside the cutoff
                        const bool counted = within && listed && hs_c >= 0.0f && lj.x >= 0.0f;
                        npair += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(counted));
                        nlj += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(counted && e4 != 0.0f));
                    }
                    if (within && listed) {
                        const float c2 = sg * sg * (1.0f / rc2);
                        const float c6 = c2 * c2 * c2;
                        if (WGT) {
                            elj = fmaf(wgt, t6 * (s6 - 1.0f) - e4 * c6 * (c6 - 1.0f), elj);
                            ecoul = fmaf(wgt, qq * (erfc_br * rinv - A.c_rf_shift), ecoul);
                        } else {
                            elj += t6 * (s6 - 1.0f) - e4 * c6 * (c6 - 1.0f);
                            ecoul += qq * (erfc_br * rinv - A.c_rf_shift);
                        }
                    } else if (within) {
                        if (WGT) eexcl = fmaf(wgt, e_ex, eexcl); else eexcl += e_ex;
                    }
                    if (WGT) {
                        const float wx = wgt * fx, wy = wgt * fy, wz = wgt * fz;
                        vxx = fmaf(dx, wx, vxx); vxy = fmaf(dx, wy, vxy); vxz = fmaf(dx, wz, vxz);
                        vyy = fmaf(dy, wy, vyy); vyz = fmaf(dy, wz, vyz); vzz = fmaf(dz, wz, vzz);
                    } else if (!PFV) {
                        vxx = fmaf(dx, fx, vxx); vxy = fmaf(dx, fy, vxy); vxz = fmaf(dx, fz, vxz);
                        vyy = fmaf(dy, fy, vyy); vyz = fmaf(dy, fz, vyz); vzz = fmaf(dz, fz, vzz);
                    }
                }
            };
#pragma unroll
            for (int c = 0; c < NI; c++)
                if (imask & (1u << c)) tile(general_tag, c);
            // The wave waits for the next entry's j-atoms here, ahead of this entry's force atomics: memory operations
            // retire in order, so a wait placed behind the atomics (where the values are first used) would also wait for
            // the atomics to reach the L2, once per entry.
            asm volatile("" :: "v"(N.pj.x), "v"(N.pj.y), "v"(N.pj.z), "v"(N.pj.w), "v"(N.lj.x), "v"(N.lj.y), "v"(N.ex));
            // an entry none of whose tiles had a pair inside the cutoff has no j-force to reduce or add
            float fj3 = 0.0f;
            if (did) fj3 = sum8_xyz(fjx, fjy, fjz, (ti & 1) != 0, (ti & 2) != 0);
            // (one 24-lane atomic per entry: pairing two entries into one 48-lane atomic costs more VALU instructions in
            // the selects than the second atomic costs now that no wait sits behind it)
            if (did && ti < 3) atomicAdd(atb(A.f_s, (unsigned)cj * 96u + fj_off), fj3);
            if (PFV && did) {
                // lanes ti < 3 hold component ti of the force on j-atom tj
                const float fm = ti < 3 ? fj3 : 0.0f;
                pvj[0] = fmaf(pj.x - O4.x, fm, pvj[0]); pvj[1] = fmaf(pj.y - O4.y, fm, pvj[1]); pvj[2] = fmaf(pj.z - O4.z, fm, pvj[2]);
            }
        };
        // The entries whose tiles look at exclusion and liveness bits (a mask slot, a cluster with itself, a launch that
        // leaves the listed pairs out) and the plain ones are walked in two loops of their own, the plain ones first: with
        // both tile variants behind one branch in one loop the compiler keeps two registers per force sum of the i-atoms
        // and copies one into the other on every tile that is skipped.
        auto walk = [&](auto general_tag, unsigned long long bits) {
            if (bits == 0ull) return;
            JSet Ja, Jb;
            fetch(general_tag, Ja, __builtin_ctzll(bits));
            bits &= bits - 1ull;
            Jb = Ja;
            for (;;) {
                if (bits) fetch(general_tag, Jb, __builtin_ctzll(bits));
                entry(general_tag, Ja, Jb);
                if (bits == 0ull) break;
                bits &= bits - 1ull;
                if (bits) fetch(general_tag, Ja, __builtin_ctzll(bits));
                entry(general_tag, Jb, Ja);
                if (bits == 0ull) break;
                bits &= bits - 1ull;
            }
        };
        walk(std::false_type{}, plain_bits);
        walk(std::true_type{}, general_bits);
#pragma unroll
        for (int c = 0; c < NI; c++) {
            const float ax = fix[c], ay = fiy[c], az = fiz[c];
            // (x, y, z over the eight j-lanes of an i-atom with the same keep-one-give-one exchanges as sum8_xyz: the lanes
            // with tj = 0, 1, 2 end with the sums of x, y, z)
            const bool t0 = (tj & 1) != 0, t1 = (tj & 2) != 0;
            const float a_ = (t0 ? ay : ax) + dpp_row_ror8(t0 ? ax : ay);
            const float z_ = az + dpp_row_ror8(az);
            float val = (t1 ? z_ : a_) + __shfl_xor(t1 ? a_ : z_, 16, 64);
            val += __shfl_xor(val, 32, 64);
            if (tj < 3) atomicAdd(at32(A.f_s, (unsigned)(3 * ((sci * 8 + c0 + c) * 8 + ti) + tj)), val);
            if (PFV) {
                // lanes tj < 3 hold component tj of the force on i-atom ti of cluster c
                const float fm = tj < 3 ? val : 0.0f;
                pvi[0] = fmaf(xi[c] - O4.x, fm, pvi[0]); pvi[1] = fmaf(yi[c] - O4.y, fm, pvi[1]); pvi[2] = fmaf(zi[c] - O4.z, fm, pvi[2]);
            }
        }
        }
        }
        unit = unit_n; part = part_n; w = w_n;
        if (slot + 2 < nstatic) { unit_n = static_unit(slot + 2); part_n = 0; }
        else if (A.dyn > 0) {
            const int t = __builtin_amdgcn_readfirstlane(tk);
            unit_n = dyn_base + 8 * (t >> split); part_n = t & pmask;
        } else unit_n = nwork;
    }
    if (TIMED && (threadIdx.x & 63) == 0) {
        long long *o = A.wave_t + 10 * (size_t)wave_id;
        o[8] = n_tested; o[9] = n_eval;
        o[0] = t_start; o[1] = clock64(); o[2] = w_start; o[3] = wall_clock64(); o[4] = n_units; o[5] = n_entries;
        o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_ID, XCC_ID
    }
    if (ENERGY) {
        elj = wave_sum(elj); ecoul = wave_sum(ecoul); eexcl = wave_sum(eexcl);
        float v9[9];
        if (PFV) {
            // element (a, b): row a of the j-side sums of the lanes that held component b (ti == b), and of the i-side
            // sums of the lanes with tj == b
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) v9[3 * a + b] = wave_sum((ti == b ? pvj[a] : 0.0f) + (tj == b ? pvi[a] : 0.0f));
        } else {
            vxx = wave_sum(vxx); vxy = wave_sum(vxy); vxz = wave_sum(vxz);
            vyy = wave_sum(vyy); vyz = wave_sum(vyz); vzz = wave_sum(vzz);
            v9[0] = vxx; v9[1] = vxy; v9[2] = vxz; v9[3] = vxy; v9[4] = vyy; v9[5] = vyz; v9[6] = vxz; v9[7] = vyz; v9[8] = vzz;
        }
        const unsigned long long np64 = npair, nl64 = nlj;
        if (lane == 0) {
            // replicated accumulators (folded by the finishing kernel): one replica per wave index mod 64
            const int rep = wave_id & (MDX_NREP - 1);
            if (COUNT) {
                atomicAdd(&A.dw->npairs_rep[rep][0], np64);
                atomicAdd(&A.dw->npairs_rep[rep][1], nl64);
            }
            atomicAdd(&A.dw->ener_rep[rep][MDX_E_LJ], (double)elj);
            atomicAdd(&A.dw->ener_rep[rep][MDX_E_COUL_SR], (double)ecoul);
            atomicAdd(&A.dw->ener_rep[rep][MDX_E_COUL_EXCL], (double)eexcl);
            double *v = A.dw->vir_rep[rep];
#pragma unroll
            for (int k = 0; k < 9; k++) atomicAdd(&v[k], -0.5 * (double)v9[k]);
        }
    }
}

// All kernels keep three waves per SIMD (one 12-wave workgroup per CU with a table, three 4-wave workgroups without), so
// that the kernels of the PME and bonded streams find wave slots, registers and LDS beside them.
//
// Force-only kernel of the step loop: the Ewald correction from a 32 KB table in LDS, the LJ parameters of the i-atoms in
// LDS (128 registers: three such waves leave a SIMD room for three waves of the co-running PME and bonded kernels).
template <bool TIMED>
__device__ __forceinline__ void nb_force_tab(const NbArgs &A)
{
    __shared__ float4 tab4[MDX_GTAB_N / 2];
    __shared__ float2 ilj[12 * 64];
    const float4 *src = (const float4 *)A.gtab;
    for (int k = threadIdx.x; k < MDX_GTAB_N / 2; k += 768) tab4[k] = src[k];
    __syncthreads();
    nb_body<false, 1, 8, false, true, TIMED>(A, (const float2 *)tab4, ilj);
}
// ... with the Lennard-Jones parameters by atom type: a table of (12 C12, 6 C6) per type pair in LDS beside the Ewald table
__device__ __forceinline__ void nb_force_tab_typed(const NbArgs &A)
{
    __shared__ float4 tab4[MDX_GTAB_N / 2];
    __shared__ float2 ilj[12 * 64];
    __shared__ float2 ljtab[MDX_NB_MAXTYPES * MDX_NB_MAXTYPES];
    const float4 *src = (const float4 *)A.gtab;
    for (int k = threadIdx.x; k < MDX_GTAB_N / 2; k += 768) tab4[k] = src[k];
    for (int k = threadIdx.x; k < A.nt * A.nt; k += 768) ljtab[k] = A.nbfp[k];
    __syncthreads();
    nb_body<false, 1, 8, false, true, false, true>(A, (const float2 *)tab4, ilj, ljtab);
}
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nonbonded_f(NbArgs A) { nb_force_tab<false>(A); }
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nonbonded_ft(NbArgs A) { nb_force_tab_typed(A); }
// ... with per-wave clocks and tile counters (mdx_debug_wave_times)
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_nonbonded_f_timed(NbArgs A) { nb_force_tab<true>(A); }
// Energy/virial kernel with g and h tabulated (64 KB per CU): no exponential, no erfc polynomial. All eight i-clusters
// of a unit in one pass over its entries (127 registers; 128 us alone against 162 for two passes over four at a time,
// which load the j-atoms and reduce the j-forces twice), except with the pair weights of a decomposed run, whose extra
// registers do not fit a single pass. COUNT adds the pair statistics (mdx_compute_forces, and steps that ask for them);
// WGT the pair weights of a decomposed run.
template <int NI, bool WGT, bool COUNT>
__device__ __forceinline__ void nb_energy_tab(const NbArgs &A)
{
    __shared__ float4 tab4[MDX_GTAB_N];
    __shared__ float2 ilj[12 * 64];
    const float4 *src = (const float4 *)A.htab;
    for (int k = threadIdx.x; k < MDX_GTAB_N; k += 768) tab4[k] = src[k];
    __syncthreads();
    nb_body<true, 2, NI, WGT, COUNT>(A, (const float2 *)tab4, ilj);
}
// (waves_per_eu(4, 4): at most 128 registers, like the force kernel. Three waves of more than 128 registers leave a SIMD too
// little for the two 64-register waves per SIMD of a PME plane workgroup, and beside such an energy kernel the whole PME chain
// waits for nonbonded workgroups to retire: with a bound of (3, 4) the one-pass kernel took 138 registers after an unrelated
// change, ran 123 instead of 152 us, and made an energy step 55 us longer. A bound of (3, 3) pads any allocation past 128.
// The two-pass kernel of a decomposed run needs fewer than 128 under either bound.)
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nonbonded_ev(NbArgs A) { nb_energy_tab<8, false, false>(A); }
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nonbonded_ev_count(NbArgs A) { nb_energy_tab<8, false, true>(A); }
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_nonbonded_ev_dd(NbArgs A) { nb_energy_tab<4, true, true>(A); }
// Analytic kernels (rational Ewald correction, exp + erfc polynomial in the energy kernel): what a handle without Ewald
// tables runs, and the cross-check of the tabulated kernels (mdx_debug_set key 2 bit 0x400000).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_nonbonded_f_analytic(NbArgs A) { nb_body<false, 0>(A); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_nonbonded_ev_analytic(NbArgs A) { nb_body<true, 0>(A); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_nonbonded_ev_analytic_dd(NbArgs A) { nb_body<true, 0, 8, true>(A); }

int mdx_launch_nonbonded(mdx_system *s, bool energy, unsigned terms)
{
    NbArgs A;
    A.central = s->central_shift;
    A.wave_t = s->nb_wave_t;
    A.dyn = s->nb_dyn; A.dyn_split = s->nb_dyn_split;
    A.htab = s->htab; A.gtab = s->gtab; A.tab_scale = (float)MDX_GTAB_N / (s->rc * s->rc);
    A.do_pairs = (terms & MDX_F_NONBONDED) ? 1 : 0;
    A.do_excl = ((terms & MDX_F_EXCL) && s->beta > 0.0f) ? 1 : 0;
    if (!A.do_pairs && !A.do_excl) return MDX_OK;
    A.d = s->d; A.dw = s->d; A.work = s->work; A.order = s->work_order; A.cj = s->cj;
    // (dual pair list: the pruned list of the current outer list, once its first pruning has run)
    A.nwork = &s->d->nwork;
    if (s->inner_valid && s->nstprune > 0) {
        A.work = s->work_in[s->inner_cur]; A.cj = s->cj_in[s->inner_cur]; A.order = s->order_in[s->inner_cur];
        A.nwork = &s->d->nwork_in[s->inner_cur];
    }
    A.exclmask = s->exclmask; A.xq_s = s->xq_s; A.lj_s = s->lj_s; A.ljt_s = s->ljt_s; A.nbfp = s->nbfp; A.nt = s->nt;
    A.shiftvec = s->shiftvec; A.f_s = s->f_s;
    A.rc2 = s->rc * s->rc; A.beta = s->beta;
    A.c_rf_shift = (float)(erfc((double)s->beta * s->rc) / s->rc);
    // kernel timing samples launches of the step loop: the start and stop events ride on the kernel's own dispatch
    // packet, so they read the kernel's begin and end time stamps (no marker packets, no launch gap inside)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    // (force-only launches only, at a stride that does not lock onto the 10- and 20-step cadences of energy and search steps)
    const bool timed = s->time_nb && (s->nb_launch_seq++ % (unsigned)s->time_stride == 0) && !energy;
    if (timed) {
        // the pool is full: read the pairs recorded so far (waits for the stream) and start over
        if (2 * (s->ev_nb.size() + 1) > s->ev_pool.size()) {
            if (hipStreamSynchronize(s->stream) != hipSuccess) { s->err = "hipStreamSynchronize failed"; return MDX_ERR_HIP; }
            for (auto &p : s->ev_nb) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, s->ev_pool[p.first], s->ev_pool[p.second]) == hipSuccess) { s->nb_ms_total += ms; s->nb_launches++; }
            }
            s->ev_nb.clear();
        }
        const int k = 2 * (int)s->ev_nb.size();
        if (k + 1 < (int)s->ev_pool.size()) {
            ev_a = s->ev_pool[k]; ev_b = s->ev_pool[k + 1];
            s->ev_nb.push_back(std::make_pair(k, k + 1));
        }
    }
    // The grid holds the resident waves only (the unit count lives in device memory, every wave walks the list until it
    // is exhausted): one 12-wave workgroup per CU for the kernels with a table, three 4-wave workgroups per CU without.
    const bool tables = s->gtab && s->htab && !(s->dbg & 0x400000);
    void (*kern)(NbArgs);
    dim3 grid(256), block(768);
    if (!tables) {
        kern = energy ? (s->dd_on ? k_nonbonded_ev_analytic_dd : k_nonbonded_ev_analytic) : k_nonbonded_f_analytic;
        grid = dim3(256 * 3); block = dim3(256);
    } else if (!energy) kern = s->nb_wave_t ? k_nonbonded_f_timed : ((s->nt > 0 && !s->dd_on && !(s->dbg & 0x1000000)) ? k_nonbonded_ft : k_nonbonded_f);
    else if (s->dd_on) kern = k_nonbonded_ev_dd;
    else kern = s->count_pairs ? k_nonbonded_ev_count : k_nonbonded_ev;
    hipExtLaunchKernelGGL(kern, grid, block, 0, s->stream, ev_a, ev_b, 0, A);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// Diagnostic: time `iters` back-to-back launches of the nonbonded kernel on the current list
// (forces accumulate garbage into the sorted buffer, which the next step clears).
extern "C" int mdx_debug_time_nonbonded(mdx_system *s, int energy, int iters, double *avg_ms)
{
    (void)hipSetDevice(s->device);
    hipEvent_t a, b;
    MDX_CHECK(s, hipEventCreate(&a)); MDX_CHECK(s, hipEventCreate(&b));
    const bool keep = s->time_nb; s->time_nb = false;
    (void)hipMemsetAsync((char *)s->d + offsetof(DevState, nb_head), 0, sizeof(int) * 8 * 32, s->stream);
    int rc = mdx_launch_nonbonded(s, energy != 0, MDX_F_NONBONDED | MDX_F_EXCL);          // warm
    MDX_CHECK(s, hipEventRecord(a, s->stream));
    for (int i = 0; i < iters && rc == MDX_OK; i++) {
        (void)hipMemsetAsync((char *)s->d + offsetof(DevState, nb_head), 0, sizeof(int) * 8 * 32, s->stream);
        rc = mdx_launch_nonbonded(s, energy != 0, MDX_F_NONBONDED | MDX_F_EXCL);
    }
    MDX_CHECK(s, hipEventRecord(b, s->stream));
    MDX_CHECK(s, hipEventSynchronize(b));
    float ms = 0.0f;
    MDX_CHECK(s, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    s->time_nb = keep;
    if (avg_ms) *avg_ms = ms / (iters > 0 ? iters : 1);
    return rc;
}

// Diagnostic: one launch of the force-only kernel with per-wave clocks; out holds 10 values per wave
// (shader clock at start and end, 100 MHz wall clock at start and end, work units, list entries, HW_ID, XCC_ID, 8x8 tiles tested, tiles evaluated).
extern "C" int mdx_debug_wave_times(mdx_system *s, long long *out, int max_waves, int *nwaves)
{
    (void)hipSetDevice(s->device);
    const int nw = 256 * 3 * 4;
    if (!out || max_waves < nw) { s->err = "mdx_debug_wave_times: buffer too small"; return MDX_ERR_ARG; }
    long long *d = nullptr;
    MDX_CHECK(s, hipMalloc((void **)&d, sizeof(long long) * 10 * nw));
    MDX_CHECK(s, hipMemsetAsync(d, 0, sizeof(long long) * 10 * nw, s->stream));
    const bool keep = s->time_nb;
    s->time_nb = false; s->nb_wave_t = d;
    (void)hipMemsetAsync((char *)s->d + offsetof(DevState, nb_head), 0, sizeof(int) * 8 * 32, s->stream);
    int rc = mdx_launch_nonbonded(s, false, MDX_F_NONBONDED | MDX_F_EXCL);
    s->nb_wave_t = nullptr; s->time_nb = keep;
    if (rc == MDX_OK && hipMemcpyAsync(out, d, sizeof(long long) * 10 * nw, hipMemcpyDeviceToHost, s->stream) != hipSuccess) rc = MDX_ERR_HIP;
    if (hipStreamSynchronize(s->stream) != hipSuccess) rc = MDX_ERR_HIP;
    (void)hipFree(d);
    if (nwaves) *nwaves = nw;
    return rc;
}

// Diagnostic (library built with -DMDX_STAMPS): reset == 1 enqueues a reset of the kernel time stamps on the main
// stream; otherwise waits for the device and copies first-start / last-end of the 24 kernel ids (100 MHz ticks).
extern "C" int mdx_debug_stamps(mdx_system *s, int reset, long long *out48)
{
    (void)hipSetDevice(s->device);
    if (reset) {
        MDX_CHECK(s, hipMemsetAsync((char *)s->d + offsetof(DevState, ts_min), 0x7f, sizeof(long long) * 24, s->stream));
        MDX_CHECK(s, hipMemsetAsync((char *)s->d + offsetof(DevState, ts_max), 0, sizeof(long long) * 24, s->stream));
        return MDX_OK;
    }
    MDX_CHECK(s, hipDeviceSynchronize());
    MDX_CHECK(s, hipMemcpy(out48, (char *)s->d + offsetof(DevState, ts_min), sizeof(long long) * 48, hipMemcpyDeviceToHost));
    return MDX_OK;
}

extern "C" int mdx_debug_set(mdx_system *s, int key, int value)
{
    if (key == 2) { s->dbg = value; return MDX_OK; }
    if (key == 3) { s->nb_dyn = value; return MDX_OK; }
    if (key == 4 && value >= 0 && value <= 3) { s->nb_dyn_split = value; return MDX_OK; }
    if (key == 11) { s->count_pairs_steps = s->count_pairs = value != 0; return MDX_OK; }     // energy steps of mdx_step count the pairs inside the cutoff
    if (key == 12) { s->fused_update = value != 0; return MDX_OK; }     // leap-frog + constraints (+ next sorted coordinates) in one launch
    if (key == 14) { s->nb_chunk = value; s->steps_since_search = 1 << 30; return MDX_OK; }   // list entries per work unit (the next search applies it)
    if (key == 10) {      // streams per handle: 1 (every kernel in one stream's order) or 3
        if (value != 1 && value != 3) { s->err = "mdx_debug_set: key 10 takes 1 or 3"; return MDX_ERR_ARG; }
        return mdx_set_stream_mode(s, value == 1);
    }
    s->err = "mdx_debug_set: unknown key"; return MDX_ERR_ARG;
}
