// Smooth particle-mesh Ewald, order 4 (Essmann et al. 1995): spread -> rocFFT R2C -> solve ->
// C2R -> gather. Replaces what `coulombtype = PME`, `pme_order = 4`, `fourierspacing = 0.16`
// (/root/reference/mdp/production.mdp:27-29) switch on inside gmx mdrun.
// Grid layout: (nx, ny, nz) with z fastest; fractional coordinates s = x . inv(box) so the
// triclinic dodecahedron needs no special casing; reciprocal vectors m = inv(box) . m'.
#include "mdx_internal.h"
#include "bonded_body.h"
#include "pull_body.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cmath>

__device__ __forceinline__ void bspline4(float t, float *w, float *dw)
{
    const float w20 = 1.0f - t, w21 = t;
    const float w30 = 0.5f * (1.0f - t) * w20;
    const float w31 = 0.5f * ((t + 1.0f) * w20 + (2.0f - t) * w21);
    const float w32 = 0.5f * t * w21;
    dw[0] = -w30; dw[1] = w30 - w31; dw[2] = w31 - w32; dw[3] = w32;
    const float third = 1.0f / 3.0f;
    w[0] = (1.0f - t) * w30 * third;
    w[1] = ((t + 2.0f) * w30 + (2.0f - t) * w31) * third;
    w[2] = ((t + 1.0f) * w31 + (3.0f - t) * w32) * third;
    w[3] = t * w32 * third;
}

__device__ __forceinline__ void frac_index(const DevBox &b, float4 p, int K0, int K1, int K2, int *i0, float *t)
{
    const int K[3] = { K0, K1, K2 };
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float s = p.x * b.inv[0 + d] + p.y * b.inv[3 + d] + p.z * b.inv[6 + d];
        s -= floorf(s);
        float u = s * K[d];
        int fl = (int)floorf(u);
        if (fl >= K[d]) { fl = K[d] - 1; }
        float tt = u - fl;
        tt = fminf(fmaxf(tt, 0.0f), 1.0f);
        i0[d] = fl; t[d] = tt;
    }
}

// Continuous (unwrapped) grid coordinates of a position that lies in or near the brick cell.
__device__ __forceinline__ void grid_coords(const DevBox &b, float4 p, int K0, int K1, int K2, float *u)
{
    u[0] = K0 * (p.x * b.inv[0] + p.y * b.inv[3] + p.z * b.inv[6]);
    u[1] = K1 * (p.y * b.inv[4] + p.z * b.inv[7]);
    u[2] = K2 * (p.z * b.inv[8]);
}
__device__ __forceinline__ int pmod(int a, int K) { a %= K; return a < 0 ? a + K : a; }
// The same for |a| < 2^22 through a float reciprocal and one correction step either way: a handful of instructions and
// two temporaries where the integer division takes a few dozen of each.
__device__ __forceinline__ int pmod_small(int a, int K, float invK)
{
    int r = a - (int)floorf((float)a * invK) * K;
    r += r < 0 ? K : 0;
    r -= r >= K ? K : 0;
    return r;
}

#define MDX_SPREAD_SCALE 67108864.0f      // 2^26

// Charge spreading over slabs of grid planes. A workgroup owns the planes [tlo, tlo + SP) (unwrapped plane numbers; the
// flush wraps them) above a block of qx x qy neighbouring search columns: it collects the super-clusters of those
// columns that can reach its planes, evaluates each of their atoms once (one thread per atom, a wave per super-cluster)
// and adds the contributions that fall on its planes to an LDS window with integer atomics (fixed point, 2^-26 e). The
// windows of different slabs share no grid point and those of neighbouring column blocks overlap by the spline
// support only, so the flush issues fewer float atomics than one window per column does (1.6 instead of 2.3 per grid
// point with 2 x 2 columns), and those atomics are what the kernel's time goes into. An atom is visited by every slab
// its four planes touch ((SP + 3) / SP times on average).
struct SpreadSlabArgs {
    int ncx, ncy, qx, qy;               // search columns; columns per workgroup in x and y
    const int *col_scstart;
    const float4 *x;
    const int *perm, *perm_src, *shift_s;
    const float4 *sbb_lo, *sbb_hi;      // super-cluster bounding boxes of the last search (brick coordinates)
    int K0, K1, K2;
    int T0, T1, SP;                     // window size in x and y; planes per slab
    int tlo0;                           // lowest unwrapped plane of the first slab
    float delta;
    float *grid;
    const float4 *v_home;
};
#define MDX_SPREAD_MAXSC 256            // candidate super-clusters of one workgroup (qx qy columns of at most 32 each)

__device__ __forceinline__ void spread_slab_block(const SpreadSlabArgs &A, const DevState *__restrict__ d, const int block_y, float *tile)
{
    int *ox = (int *)(tile + A.SP * A.T1 * A.T0);
    int *oy = ox + A.SP, *oxw = oy + A.SP, *oyw = oxw + A.SP;         // window origins per plane: as computed, and wrapped into the grid
    __shared__ int s_list[MDX_SPREAD_MAXSC];
    __shared__ int s_n;
    const int nqx = (A.ncx + A.qx - 1) / A.qx;
    const int cx0 = (blockIdx.x % nqx) * A.qx, cy0 = (blockIdx.x / nqx) * A.qy;
    const int ncolx = min(A.qx, A.ncx - cx0), ncoly = min(A.qy, A.ncy - cy0);
    const int tlo = A.tlo0 + block_y * A.SP;
    const DevBox &bx = d->box;
    const float wx = bx.h[0] / A.ncx, wy = bx.h[4] / A.ncy;
    const float x_lo = cx0 * wx - A.delta;
    const float y_lo = cy0 * wy - A.delta, y_hi = (cy0 + ncoly) * wy + A.delta;
    const float kz = A.K2 * bx.inv[8];
    if (threadIdx.x == 0) s_n = 0;
    for (int p = threadIdx.x; p < A.SP; p += blockDim.x) {
        // atoms that touch plane t have floor(u_z) in [t - 3, t]: their z range, and from it the lowest u_x, u_y inside the block
        const int t = tlo + p;
        const float z_lo = (t - 3) / kz - A.delta, z_hi = (t + 1) / kz + A.delta;
        const float ux = A.K0 * (x_lo * bx.inv[0] + fminf(y_lo * bx.inv[3], y_hi * bx.inv[3]) + fminf(z_lo * bx.inv[6], z_hi * bx.inv[6]));
        const float uy = A.K1 * (y_lo * bx.inv[4] + fminf(z_lo * bx.inv[7], z_hi * bx.inv[7]));
        ox[p] = (int)floorf(ux); oy[p] = (int)floorf(uy);
        oxw[p] = pmod(ox[p], A.K0); oyw[p] = pmod(oy[p], A.K1);
    }
    const int nt = A.SP * A.T1 * A.T0;
#pragma unroll 1
    for (int k = threadIdx.x; k < nt; k += blockDim.x) tile[k] = 0.0f;
    __syncthreads();
    // super-clusters of the block's columns whose atoms can reach the slab
    for (int cy = cy0; cy < cy0 + ncoly; cy++)
        for (int cx = cx0; cx < cx0 + ncolx; cx++) {
            const int col = cy * A.ncx + cx;
            const int s0 = A.col_scstart[col], s1 = A.col_scstart[col + 1];
            for (int sc = s0 + (int)threadIdx.x; sc < s1; sc += blockDim.x) {
                const int plo = (int)floorf(kz * (A.sbb_lo[sc].z - A.delta)), phi = (int)floorf(kz * (A.sbb_hi[sc].z + A.delta)) + 3;
                if (phi >= tlo && plo < tlo + A.SP) {
                    const int k = atomicAdd(&s_n, 1);
                    if (k < MDX_SPREAD_MAXSC) s_list[k] = sc;
                }
            }
        }
    __syncthreads();
    const int nsc = min(s_n, MDX_SPREAD_MAXSC);
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int k = threadIdx.x >> 6; k < nsc; k += 4) {
        const int gs = s_list[k] * 64 + lane;
        if (A.perm[gs] >= 0) {
            const int src = A.perm_src[gs];
            float4 p = A.x[src];
            const bool skip = p.w == 0.0f || (A.v_home && A.v_home[src].w == 0.0f);
            if (!skip) {
                int sa, sb, sc;
                unpack_shift(A.shift_s[gs], sa, sb, sc);
                p.x += sa * bx.h[0] + sb * bx.h[3] + sc * bx.h[6];
                p.y += sb * bx.h[4] + sc * bx.h[7];
                p.z += sc * bx.h[8];
                float u[3];
                grid_coords(bx, p, A.K0, A.K1, A.K2, u);
                const int px = (int)floorf(u[0]), py = (int)floorf(u[1]), pz = (int)floorf(u[2]);
                float wxs[4], wys[4], wzs[4], dws[4];
                bspline4(u[0] - px, wxs, dws); bspline4(u[1] - py, wys, dws); bspline4(u[2] - pz, wzs, dws);
                const float qs = p.w * MDX_SPREAD_SCALE;
                // (at most 40 registers, so that three of these waves fit on a SIMD beside three 128-register nonbonded waves)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int pl = pz + c - tlo;
                    if (pl >= 0 && pl < A.SP) {             // this plane of the atom belongs to the slab
                        const float qz = qs * wzs[c];
                        const int rx0 = px - ox[pl], ry0 = py - oy[pl];
                        if (rx0 >= 0 && rx0 + 3 < A.T0 && ry0 >= 0 && ry0 + 3 < A.T1) {
                            int *t = (int *)tile + (pl * A.T1 + ry0) * A.T0 + rx0;
#pragma unroll
                            for (int b = 0; b < 4; b++) {
                                const float qzy = qz * wys[b];
#pragma unroll
                                for (int a = 0; a < 4; a++) atomicAdd(&t[b * A.T0 + a], __float2int_rn(qzy * wxs[a]));
                            }
                        } else {
                            // outside the window (an atom that drifted far): straight to the grid
                            const int gz = pmod_small(pz + c, A.K2, 1.0f / A.K2);
                            const float qzf = p.w * wzs[c];
#pragma unroll 1
                            for (int ba = 0; ba < 16; ba++) {       // rare path: rolled, registers matter more than speed here
                                const int b = ba >> 2, a = ba & 3;
                                const float wyb = (b == 0) ? wys[0] : ((b == 1) ? wys[1] : ((b == 2) ? wys[2] : wys[3]));
                                const float wxa = (a == 0) ? wxs[0] : ((a == 1) ? wxs[1] : ((a == 2) ? wxs[2] : wxs[3]));
                                const int gy = pmod_small(py + b, A.K1, 1.0f / A.K1), gx = pmod_small(px + a, A.K0, 1.0f / A.K0);
                                atomicAdd(&A.grid[((size_t)gz * A.K1 + gy) * A.K0 + gx], qzf * wyb * wxa);
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // Flush of the non-zero window points: a thread keeps its (rx, ry) inside the plane window and walks the planes
    const int t01 = A.T0 * A.T1;
    const int gz0 = pmod(tlo, A.K2);
#pragma unroll 1
    for (int i = threadIdx.x; i < t01; i += blockDim.x) {
        const int ry = i / A.T0, rx = i - ry * A.T0;
        int gz = gz0;
#pragma unroll 1
        for (int pl = 0; pl < A.SP; pl++) {
            const int vi = ((const int *)tile)[pl * t01 + i];
            if (vi != 0) {
                int gx = oxw[pl] + rx, gy = oyw[pl] + ry;
                gx -= gx >= A.K0 ? A.K0 : 0; gy -= gy >= A.K1 ? A.K1 : 0;
                atomicAdd(&A.grid[((size_t)gz * A.K1 + gy) * A.K0 + gx], (float)vi * (1.0f / MDX_SPREAD_SCALE));
            }
            gz++; gz -= gz >= A.K2 ? A.K2 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_pme_spread_slab(SpreadSlabArgs A, const DevState *__restrict__ d)
{
    MDX_STAMP_ALL(d, 3);
    extern __shared__ float tile[];
    spread_slab_block(A, d, blockIdx.y, tile);
}
// The spreading with the step's bonded work in the same launch: the first `brows` rows of the grid (dispatched first) are
// workgroups of the bonded launch, which keep their force window in the dynamic LDS the spreading windows use. On a step
// without energies the bonded stream then carries nothing, and the update waits for one other stream instead of two
// (a wait for an event of another stream costs the main stream 2-3 us even when the event has long fired).
__global__ __launch_bounds__(256) void k_pme_spread_slab_bonded(SpreadSlabArgs A, const DevState *__restrict__ d, BondedArgs B, int brows)
{
    MDX_STAMP_ALL(d, 3);      // (the launch as a whole, its bonded rows included)
    extern __shared__ float tile[];
    if ((int)blockIdx.y < brows) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if (b < B.nblocks) bonded_block<false>(B, b, tile, (int *)(tile + 3 * MDX_BWIN));
        return;
    }
    spread_slab_block(A, d, (int)blockIdx.y - brows, tile);
}

// ... and with the centre-of-mass pull of the step as one more workgroup behind the bonded ones (a run with `pull = yes`).
__global__ __launch_bounds__(256) void k_pme_spread_slab_bonded_pull(SpreadSlabArgs A, const DevState *__restrict__ d, BondedArgs B, int brows,
                                                                     PullArgs P, const float4 *__restrict__ v, float *__restrict__ f, DevState *dw)
{
    MDX_STAMP_ALL(d, 3);
    extern __shared__ float tile[];
    if ((int)blockIdx.y < brows) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        if (b < B.nblocks) bonded_block<false>(B, b, tile, (int *)(tile + 3 * MDX_BWIN));
        else if (b == B.nblocks) pull_block(P, A.x, v, f, dw);
        return;
    }
    spread_slab_block(A, d, (int)blockIdx.y - brows, tile);
}

// Multiply the transformed grid by the Ewald influence function; optionally accumulate the
// reciprocal energy and virial. cgrid is (K2, K1, K0/2+1) complex (x fastest).
template <bool ENERGY>
__global__ __launch_bounds__(256) void k_pme_solve(float2 *__restrict__ cgrid, const DevState *__restrict__ d, DevState *__restrict__ dw,
                                                   int K0, int K1, int K2, const float *__restrict__ bm0,
                                                   const float *__restrict__ bm1, const float *__restrict__ bm2, float beta)
{
    __shared__ double red[16 * 7];
    const int K0c = K0 / 2 + 1;
    const size_t total = (size_t)K2 * K1 * K0c;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0, v[6] = { 0, 0, 0, 0, 0, 0 };
    if (g < total) {
        const int k0 = (int)(g % K0c);
        const int k1 = (int)((g / K0c) % K1);
        const int k2 = (int)(g / ((size_t)K0c * K1));
        // x is the halved dimension: m0 = k0 >= 0 (the Nyquist term counts as +K0/2); y and z are signed
        // with the Nyquist index of an even dimension counted as -K/2 (the enumeration GROMACS uses)
        const int m0 = k0;
        const int m1 = (k1 < (K1 + 1) / 2) ? k1 : k1 - K1;
        const int m2 = (k2 < (K2 + 1) / 2) ? k2 : k2 - K2;
        if (m0 != 0 || m1 != 0 || m2 != 0) {
            const float *iv = d->box.inv;
            const float mx = iv[0] * m0 + iv[1] * m1 + iv[2] * m2;
            const float my = iv[3] * m0 + iv[4] * m1 + iv[5] * m2;
            const float mz = iv[6] * m0 + iv[7] * m1 + iv[8] * m2;
            const float msq = mx * mx + my * my + mz * mz;
            const float pi = 3.14159265358979f;
            const float fac = pi * pi / (beta * beta);
            const float denom = msq * bm0[k0] * bm1[k1] * bm2[k2] * pi * d->box.vol;
            const float eterm = MDX_ONE_4PI_EPS0 * __expf(-fac * msq) / denom;
            float2 c = cgrid[g];
            if (ENERGY) {
                // half-complex storage along x: 0 < k0 < K0/2 (K0 odd: all k0 > 0) stands for two terms
                const float mult = (k0 == 0 || (K0 % 2 == 0 && k0 == K0 / 2)) ? 1.0f : 2.0f;
                const double em = 0.5 * mult * (double)eterm * ((double)c.x * c.x + (double)c.y * c.y);
                e = em;
                const double vf = 2.0 * (1.0 + fac * msq) / msq;
                v[0] = em * (1.0 - vf * mx * mx); v[1] = em * (-vf * mx * my); v[2] = em * (-vf * mx * mz);
                v[3] = em * (1.0 - vf * my * my); v[4] = em * (-vf * my * mz); v[5] = em * (1.0 - vf * mz * mz);
            }
            c.x *= eterm; c.y *= eterm;
            cgrid[g] = c;
        } else {
            cgrid[g] = make_float2(0.0f, 0.0f);
        }
    }
    if (ENERGY) {
        const double ev[7] = { e, v[0], v[1], v[2], v[3], v[4], v[5] };
        const double r = block_sum_n<7>(ev, red);
        const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
        if (threadIdx.x == 0) atomicAdd(&MDX_ENER(dw, MDX_E_COUL_RECIP), r);
        else if (threadIdx.x < 7) {
            const int k = threadIdx.x - 1;
            atomicAdd(&MDX_VIR(dw, map[k][0]), -0.5 * r);
            if (map[k][1] != map[k][0]) atomicAdd(&MDX_VIR(dw, map[k][1]), -0.5 * r);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Own 3D transform for PME-sized grids (a few MB): five launches of small workgroups that fit into
// the registers and LDS the nonbonded kernel leaves free, where the library's 3D plan uses 400-thread
// workgroups that wait for a CU to drain. A line of length N = N1 N2 is transformed in LDS by two
// direct DFT passes (Cooley-Tukey split, N1 + N2 complex multiply-adds per point: at N <= 160 the
// arithmetic is negligible against the launch). Passes: x (real to half-complex), y, z forward +
// influence function + z inverse in one kernel, y inverse, x inverse (half-complex to real).
// Unnormalised in both directions, layout of cgrid (K2, K1, K0/2+1) as hipFFT's R2C plan.
// Element-wise copy loops of the transform kernels with B loads of a thread in flight at a time: a plain loop issues one
// load, waits for it (a trip to the L2 or beyond: the data was written by another kernel) and stores, once per trip.
template <int B, class T, class Load, class Store>
__device__ __forceinline__ void copy_batched(const int n, Load &&ld, Store &&st)
{
    const int nt = blockDim.x;
    for (int base = threadIdx.x; base < n; base += B * nt) {
        T v[B];
#pragma unroll
        for (int u = 0; u < B; u++) { const int i = base + u * nt; if (i < n) v[u] = ld(i); }
#pragma unroll
        for (int u = 0; u < B; u++) { const int i = base + u * nt; if (i < n) st(i, v[u]); }
    }
}
struct FftArgs {
    float *grid_in; float *grid_out; float2 *cgrid;   // x forward reads grid_in and leaves it zeroed for the next spreading
    const float2 *W;            // exp(-2 pi i j / N), j < N, of the transformed dimension
    int K0, K1, K2, K0c;
    int N, N1, N2;
    int nl;                     // lines per workgroup
    int tr;                     // cgrid stored as (K0/2+1, K2, K1), y fastest: every k_x plane contiguous (the fused y/z kernel's layout)
    // influence function (z pass)
    const DevState *d; DevState *dw; const float *bm0, *bm1, *bm2; float beta;
};

// nl lines of length N at a[l * (N + 1)]; result in a again (b is scratch of the same size)
__device__ __forceinline__ void dft_lines(float2 *a, float2 *b, const float2 *W, int nl, int N, int N1, int N2, bool inverse)
{
    const int stride = N + 1;
    const float sgn = inverse ? -1.0f : 1.0f;
    // pass 1: b[k1 N2 + n2] = W_N^(n2 k1) sum_n1 a[N2 n1 + n2] W_N^(N2 n1 k1)
    for (int idx = threadIdx.x; idx < nl * N; idx += blockDim.x) {
        const int line = idx / N, r = idx - line * N;
        const int k1 = r / N2, n2 = r - k1 * N2;
        const float2 *x = a + line * stride + n2;
        float re = 0.0f, im = 0.0f;
        int j = 0; const int step = k1 * N2;
        for (int n1 = 0; n1 < N1; n1++) {
            const float2 w = W[j]; const float2 v = x[n1 * N2];
            const float wy = sgn * w.y;
            re = fmaf(v.x, w.x, fmaf(-v.y, wy, re));
            im = fmaf(v.x, wy, fmaf(v.y, w.x, im));
            j += step; if (j >= N) j -= N;
        }
        const float2 t = W[n2 * k1]; const float ty = sgn * t.y;
        b[line * stride + r] = make_float2(re * t.x - im * ty, re * ty + im * t.x);
    }
    __syncthreads();
    // pass 2: a[k1 + N1 k2] = sum_n2 b[k1 N2 + n2] W_N^(N1 n2 k2)
    for (int idx = threadIdx.x; idx < nl * N; idx += blockDim.x) {
        const int line = idx / N, r = idx - line * N;
        const int k1 = r / N2, k2 = r - k1 * N2;
        const float2 *x = b + line * stride + k1 * N2;
        float re = 0.0f, im = 0.0f;
        int j = 0; const int step = k2 * N1;
        for (int n2 = 0; n2 < N2; n2++) {
            const float2 w = W[j]; const float2 v = x[n2];
            const float wy = sgn * w.y;
            re = fmaf(v.x, w.x, fmaf(-v.y, wy, re));
            im = fmaf(v.x, wy, fmaf(v.y, w.x, im));
            j += step; if (j >= N) j -= N;
        }
        a[line * stride + k1 + N1 * k2] = make_float2(re, im);
    }
    __syncthreads();
}

// The same transform with the small DFTs of a pass done by one thread each: a thread loads the R inputs of one small DFT
// into registers once and forms its R outputs from them, the R-th roots of unity in scalar registers (they are the same
// for every lane) and every index known at compile time. That is about R multiply-adds per output point (dft_small) and one
// LDS read per input point, where the one-output-per-thread form above spends about ten instructions and two LDS reads per
// multiply-add on loads and index arithmetic - and the transforms run beside the nonbonded kernel, which leaves them few
// issue slots. Radices up to 10 (every factor pair of the PME grids up to 100 points); other lines take the form above.
#define MDX_DFT_MAXR 10
template <int R>
__device__ __forceinline__ void dft_roots(const float2 *W, int N, float sgn, float (&wr)[R], float (&wi)[R])
{
    // exp(-+ 2 pi i m / R), m < R, as wave-uniform values
#pragma unroll
    for (int m = 0; m < R; m++) {
        const float2 w = W[m * (N / R)];
        wr[m] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w.x)));
        wi[m] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sgn * w.y)));
    }
}
// R-point DFT of register values, X_k = sum_j x_j w^(jk) with w^m = wr[m] + i wi[m]. The inputs j and R - j meet the same
// cosine and opposite sines, for the outputs k and R - k alike: with s_j = x_j + x_(R-j), d_j = x_j - x_(R-j),
// A_k = x_0 (+- x_(R/2)) + sum_j s_j wr[jk] and B_k = sum_j d_j wi[jk] (j = 1 .. (R-1)/2) the outputs are
// X_k = A_k + i B_k and X_(R-k) = A_k - i B_k: about R^2 multiply-adds instead of 4 R^2. emit(k, re, im) takes the outputs.
template <int R, class F>
__device__ __forceinline__ void dft_small(const float (&xr)[R], const float (&xi)[R], const float (&wr)[R], const float (&wi)[R], F &&emit)
{
    constexpr int H = (R - 1) / 2;
    constexpr bool EVEN = (R % 2) == 0;
    float sr[H > 0 ? H : 1], si[H > 0 ? H : 1], dr[H > 0 ? H : 1], di[H > 0 ? H : 1];
#pragma unroll
    for (int j = 1; j <= H; j++) {
        sr[j - 1] = xr[j] + xr[R - j]; si[j - 1] = xi[j] + xi[R - j];
        dr[j - 1] = xr[j] - xr[R - j]; di[j - 1] = xi[j] - xi[R - j];
    }
    {
        float re = xr[0], im = xi[0];
        if (EVEN) { re += xr[R / 2]; im += xi[R / 2]; }
#pragma unroll
        for (int j = 0; j < H; j++) { re += sr[j]; im += si[j]; }
        emit(0, re, im);
    }
#pragma unroll
    for (int k = 1; k <= H; k++) {
        float ar = xr[0], ai = xi[0], br = 0.0f, bi = 0.0f;
        if (EVEN) { if (k & 1) { ar -= xr[R / 2]; ai -= xi[R / 2]; } else { ar += xr[R / 2]; ai += xi[R / 2]; } }
#pragma unroll
        for (int j = 1; j <= H; j++) {
            const int m = (j * k) % R;
            ar = fmaf(sr[j - 1], wr[m], ar); ai = fmaf(si[j - 1], wr[m], ai);
            br = fmaf(dr[j - 1], wi[m], br); bi = fmaf(di[j - 1], wi[m], bi);
        }
        emit(k, ar - bi, ai + br);
        emit(R - k, ar + bi, ai - br);
    }
    if (EVEN) {
        float re = xr[0], im = xi[0];
        if ((R / 2) & 1) { re -= xr[R / 2]; im -= xi[R / 2]; } else { re += xr[R / 2]; im += xi[R / 2]; }
#pragma unroll
        for (int j = 1; j <= H; j++) { if (j & 1) { re -= sr[j - 1]; im -= si[j - 1]; } else { re += sr[j - 1]; im += si[j - 1]; } }
        emit(R / 2, re, im);
    }
}
// pass 1 with N1 = R: b[k1 N2 + n2] = W_N^(n2 k1) sum_n1 a[N2 n1 + n2] W_R^(n1 k1)
template <int R>
__device__ __forceinline__ void dft_pass1_r(const float2 *a, float2 *b, const float2 *W, int nl, int N, int N2, float sgn)
{
    float wr[R], wi[R];
    dft_roots<R>(W, N, sgn, wr, wi);
    const int stride = N + 1;
    for (int idx = threadIdx.x; idx < nl * N2; idx += blockDim.x) {
        const int line = idx / N2, n2 = idx - line * N2;
        const float2 *x = a + line * stride + n2;
        float xr[R], xi[R];
#pragma unroll
        for (int n1 = 0; n1 < R; n1++) { const float2 v = x[n1 * N2]; xr[n1] = v.x; xi[n1] = v.y; }
        float2 *o = b + line * stride + n2;
        dft_small<R>(xr, xi, wr, wi, [&](const int k1, const float re, const float im) {
            const float2 t = W[n2 * k1]; const float ty = sgn * t.y;
            o[k1 * N2] = make_float2(re * t.x - im * ty, re * ty + im * t.x);
        });
    }
}
// pass 2 with N2 = R: a[k1 + N1 k2] = sum_n2 b[k1 N2 + n2] W_R^(n2 k2)
template <int R>
__device__ __forceinline__ void dft_pass2_r(float2 *a, const float2 *b, const float2 *W, int nl, int N, int N1, float sgn)
{
    float wr[R], wi[R];
    dft_roots<R>(W, N, sgn, wr, wi);
    const int stride = N + 1;
    for (int idx = threadIdx.x; idx < nl * N1; idx += blockDim.x) {
        const int line = idx / N1, k1 = idx - line * N1;
        const float2 *x = b + line * stride + k1 * R;
        float xr[R], xi[R];
#pragma unroll
        for (int n2 = 0; n2 < R; n2++) { const float2 v = x[n2]; xr[n2] = v.x; xi[n2] = v.y; }
        float2 *o = a + line * stride + k1;
        dft_small<R>(xr, xi, wr, wi, [&](const int k2, const float re, const float im) { o[N1 * k2] = make_float2(re, im); });
    }
}
#define MDX_DFT_CASES(F, ...) \
    case 2: F<2>(__VA_ARGS__); break; case 3: F<3>(__VA_ARGS__); break; case 4: F<4>(__VA_ARGS__); break; \
    case 5: F<5>(__VA_ARGS__); break; case 6: F<6>(__VA_ARGS__); break; case 7: F<7>(__VA_ARGS__); break; \
    case 8: F<8>(__VA_ARGS__); break; case 9: F<9>(__VA_ARGS__); break; case 10: F<10>(__VA_ARGS__); break;
// radices 11 .. 16 for the plane kernel of large grids (160 = 10 x 16, 112 = 8 x 14, ...): only where BIG is set
#define MDX_DFT_MAXR_BIG 16
#define MDX_DFT_CASES_BIG(BIG, F, ...) \
    case 11: if (BIG) F<11>(__VA_ARGS__); break; case 12: if (BIG) F<12>(__VA_ARGS__); break; case 13: if (BIG) F<13>(__VA_ARGS__); break; \
    case 14: if (BIG) F<14>(__VA_ARGS__); break; case 15: if (BIG) F<15>(__VA_ARGS__); break; case 16: if (BIG) F<16>(__VA_ARGS__); break;
__device__ __forceinline__ void dft_lines_fast(float2 *a, float2 *b, const float2 *W, int nl, int N, int N1, int N2, bool inverse)
{
    if (N1 < 2 || N1 > MDX_DFT_MAXR || N2 < 2 || N2 > MDX_DFT_MAXR) { dft_lines(a, b, W, nl, N, N1, N2, inverse); return; }
    const float sgn = inverse ? -1.0f : 1.0f;
    switch (N1) { MDX_DFT_CASES(dft_pass1_r, a, b, W, nl, N, N2, sgn) }
    __syncthreads();
    switch (N2) { MDX_DFT_CASES(dft_pass2_r, a, b, W, nl, N, N1, sgn) }
    __syncthreads();
}

// MODE 0: x forward (real lines -> half-complex), 4: x inverse; 1 / 3: y forward / inverse; 2: z forward, solve, z inverse
template <int MODE, bool ENERGY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pme_fft(FftArgs A)
{
    MDX_STAMP(A.d, 4 + MODE);
    extern __shared__ float2 fsm[];
    const int N = A.N, stride = N + 1, nl = A.nl;
    float2 *Wl = fsm, *a = fsm + N, *b = a + nl * stride;
    for (int j = threadIdx.x; j < N; j += blockDim.x) Wl[j] = A.W[j];
    const int K0c = A.K0c;
    if (MODE == 0 || MODE == 4) {
        const int nlines = A.K1 * A.K2, L0 = blockIdx.x * nl;
        if (MODE == 0) {
            // (the nl lines are contiguous in the real grid)
            float *src = A.grid_in + (size_t)L0 * N;
            const int nvalid = min(nl, nlines - L0) * N;
            copy_batched<6, float>(nl * N, [&](const int idx) { return idx < nvalid ? src[idx] : 0.0f; },
                                   [&](const int idx, const float q) {
                                       const int l = idx / N, n = idx - l * N;
                                       if (idx < nvalid) src[idx] = 0.0f;
                                       a[l * stride + n] = make_float2(q, 0.0f);
                                   });
        } else {
            // (transposed layout: consecutive threads take consecutive lines, nl x 8 bytes contiguous per k_x)
            copy_batched<4, float2>(nl * K0c, [&](const int idx) {
                                        const int l = A.tr ? idx % nl : idx / K0c, k = A.tr ? idx / nl : idx - l * K0c;
                                        if (L0 + l >= nlines) return make_float2(0.0f, 0.0f);
                                        return A.tr ? A.cgrid[(size_t)k * nlines + L0 + l] : A.cgrid[(size_t)(L0 + l) * K0c + k];
                                    },
                                    [&](const int idx, const float2 v) {
                                        const int l = A.tr ? idx % nl : idx / K0c, k = A.tr ? idx / nl : idx - l * K0c;
                                        a[l * stride + k] = v;
                                        if (k > 0 && N - k >= K0c) a[l * stride + N - k] = make_float2(v.x, -v.y);
                                    });
        }
        __syncthreads();
        dft_lines_fast(a, b, Wl, nl, N, A.N1, A.N2, MODE == 4);
        if (MODE == 0) {
            for (int idx = threadIdx.x; idx < nl * K0c; idx += blockDim.x) {
                const int l = A.tr ? idx % nl : idx / K0c, k = A.tr ? idx / nl : idx - l * K0c;
                if (L0 + l < nlines) {
                    if (A.tr) A.cgrid[(size_t)k * nlines + L0 + l] = a[l * stride + k];
                    else A.cgrid[(size_t)(L0 + l) * K0c + k] = a[l * stride + k];
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < nl * N; idx += blockDim.x) {
                const int l = idx / N, n = idx - l * N;
                if (L0 + l < nlines) A.grid_out[(size_t)(L0 + l) * N + n] = a[l * stride + n].x;
            }
        }
        return;
    }
    // y and z passes: workgroup = (fixed other index, chunk of nl consecutive kx), lines run along the transformed dimension
    const int nchunk = (K0c + nl - 1) / nl;
    const int other = blockIdx.x / nchunk, kx0 = (blockIdx.x - other * nchunk) * nl;
    const size_t base = (MODE == 2) ? (size_t)other * K0c : (size_t)other * A.K1 * K0c;       // z pass: other = y; y pass: other = z
    const size_t estride = (MODE == 2) ? (size_t)A.K1 * K0c : (size_t)K0c;
    copy_batched<3, float2>(nl * N, [&](const int idx) {
                                const int n = idx / nl, c = idx - n * nl;
                                return (kx0 + c < K0c) ? A.cgrid[base + (size_t)n * estride + kx0 + c] : make_float2(0.0f, 0.0f);
                            },
                            [&](const int idx, const float2 v) { const int n = idx / nl; a[(idx - n * nl) * stride + n] = v; });
    __syncthreads();
    dft_lines_fast(a, b, Wl, nl, N, A.N1, A.N2, MODE == 3);
    if (MODE == 2) {
        // influence function on (kx0 + c, y = other, kz = n); energy and virial as in the stand-alone solve kernel
        double e = 0.0, v[6] = { 0, 0, 0, 0, 0, 0 };
        const float *iv = A.d->box.inv;
        const float pi = 3.14159265358979f;
        const float fac = pi * pi / (A.beta * A.beta);
        const float pivol = pi * A.d->box.vol;
        const int K0 = A.K0, K1 = A.K1, K2 = A.K2;
        for (int idx = threadIdx.x; idx < nl * N; idx += blockDim.x) {
            const int k2 = idx / nl, c = idx - k2 * nl;
            const int k0 = kx0 + c, k1 = other;
            if (k0 >= K0c) continue;
            const int m0 = k0;
            const int m1 = (k1 < (K1 + 1) / 2) ? k1 : k1 - K1;
            const int m2 = (k2 < (K2 + 1) / 2) ? k2 : k2 - K2;
            float2 cv = a[c * stride + k2];
            if (m0 != 0 || m1 != 0 || m2 != 0) {
                const float mx = iv[0] * m0 + iv[1] * m1 + iv[2] * m2;
                const float my = iv[3] * m0 + iv[4] * m1 + iv[5] * m2;
                const float mz = iv[6] * m0 + iv[7] * m1 + iv[8] * m2;
                const float msq = mx * mx + my * my + mz * mz;
                const float denom = msq * A.bm0[k0] * A.bm1[k1] * A.bm2[k2] * pivol;
                const float eterm = MDX_ONE_4PI_EPS0 * __expf(-fac * msq) / denom;
                if (ENERGY) {
                    const float mult = (k0 == 0 || (K0 % 2 == 0 && k0 == K0 / 2)) ? 1.0f : 2.0f;
                    const double em = 0.5 * mult * (double)eterm * ((double)cv.x * cv.x + (double)cv.y * cv.y);
                    e += em;
                    const double vf = 2.0 * (1.0 + fac * msq) / msq;
                    v[0] += em * (1.0 - vf * mx * mx); v[1] += em * (-vf * mx * my); v[2] += em * (-vf * mx * mz);
                    v[3] += em * (1.0 - vf * my * my); v[4] += em * (-vf * my * mz); v[5] += em * (1.0 - vf * mz * mz);
                }
                cv.x *= eterm; cv.y *= eterm;
            } else cv = make_float2(0.0f, 0.0f);
            a[c * stride + k2] = cv;
        }
        if (ENERGY) {
            // energy and the six virial sums in one reduction; thread k adds the k-th of them
            __shared__ double red[16 * 7];
            const double ev[7] = { e, v[0], v[1], v[2], v[3], v[4], v[5] };
            const double r = block_sum_n<7>(ev, red);
            const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
            if (threadIdx.x == 0) atomicAdd(&MDX_ENER(A.dw, MDX_E_COUL_RECIP), r);
            else if (threadIdx.x < 7) {
                const int k = threadIdx.x - 1;
                atomicAdd(&MDX_VIR(A.dw, map[k][0]), -0.5 * r);
                if (map[k][1] != map[k][0]) atomicAdd(&MDX_VIR(A.dw, map[k][1]), -0.5 * r);
            }
        }
        __syncthreads();
        dft_lines_fast(a, b, Wl, nl, N, A.N1, A.N2, true);
    }
    for (int idx = threadIdx.x; idx < nl * N; idx += blockDim.x) {
        const int n = idx / nl, c = idx - n * nl;
        if (kx0 + c < K0c) A.cgrid[base + (size_t)n * estride + kx0 + c] = a[c * stride + n];
    }
}

// ---------------------------------------------------------------------------------------------
// y and z transforms, influence function and the two inverse transforms of one k_x plane in one workgroup: with the
// half-complex grid stored as (k_x, z, y) a plane is one contiguous block (36 KB for 80 x 56 points), it is read
// once, transformed in LDS and written once, and the chain x -> (y, z, solve, z, y) -> x takes three launches instead
// of five (each launch beside the nonbonded kernel costs 8-11 us whatever it computes, plus 2-4 us to the next).
// The small DFTs are done in registers and IN PLACE: a line of N = N1 N2 points goes through
//   pass 1  y[k1, n2] = W_N^(n2 k1) sum_n1 x[N2 n1 + n2] W_N1^(n1 k1)      on the positions {n2 + N2 j}
//   pass 2  X[k1 + N1 k2] = sum_n2 y[k1, n2] W_N2^(n2 k2)                   on the positions {N2 k1 + j}
// and the result stays where pass 2 computed it: position p holds the frequency p / N2 + N1 (p % N2). The influence
// function is applied in that order, and the inverse runs the same passes backwards (pass 2, conjugate twiddle, pass 1),
// which takes the permuted order as its input and ends in natural order. No scratch plane, no transposition.
struct FftYzArgs {
    float2 *cgrid;                      // (K0c, K2, K1), y fastest
    const float2 *Wy, *Wz;              // exp(-2 pi i j / N) of the y and z dimensions
    int K0, K1, K2, K0c;
    int plane0;                         // first k_x plane of this launch (one workgroup per plane)
    int N1y, N2y, N1z, N2z;
    const DevState *d; DevState *dw; const float *bm0, *bm1, *bm2; float beta;
};
// One R-point DFT per task on the elements p[j * rs], j < R, of `nlines` lines (line l at P + l * ls; consecutive threads
// take consecutive lines, which keeps their LDS accesses on different banks for both directions of the plane) and `nsub`
// sub-sequences per line (sub-sequence s at + s * ss). TW 1: output k times W^(+- s k) (forward pass 1); TW 2: input j
// times W^(+- s j) (inverse pass 1); TW 0: none.
template <int R, int TW>
__device__ __forceinline__ void yz_pass(float2 *P, const float2 *W, int N, int nlines, int ls, int nsub, int ss, int rs, float sgn)
{
    float wr[R], wi[R];
    dft_roots<R>(W, N, sgn, wr, wi);
    for (int idx = threadIdx.x; idx < nlines * nsub; idx += blockDim.x) {
        const int sub = idx / nlines, line = idx - sub * nlines;
        float2 *p = P + line * ls + sub * ss;
        float xr[R], xi[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const float2 v = p[j * rs];
            if (TW == 2 && j > 0) {
                const float2 t = W[sub * j]; const float ty = sgn * t.y;
                xr[j] = v.x * t.x - v.y * ty; xi[j] = v.x * ty + v.y * t.x;
            } else { xr[j] = v.x; xi[j] = v.y; }
        }
        dft_small<R>(xr, xi, wr, wi, [&](const int k, const float re, const float im) {
            if (TW == 1 && k > 0) {
                const float2 t = W[sub * k]; const float ty = sgn * t.y;
                p[k * rs] = make_float2(re * t.x - im * ty, re * ty + im * t.x);
            } else p[k * rs] = make_float2(re, im);
        });
    }
}
template <int R> __device__ __forceinline__ void yz_pass_tw_out(float2 *P, const float2 *W, int N, int nl, int ls, int nsub, int ss, int rs, float sgn) { yz_pass<R, 1>(P, W, N, nl, ls, nsub, ss, rs, sgn); }
template <int R> __device__ __forceinline__ void yz_pass_tw_in(float2 *P, const float2 *W, int N, int nl, int ls, int nsub, int ss, int rs, float sgn) { yz_pass<R, 2>(P, W, N, nl, ls, nsub, ss, rs, sgn); }
template <int R> __device__ __forceinline__ void yz_pass_plain(float2 *P, const float2 *W, int N, int nl, int ls, int nsub, int ss, int rs, float sgn) { yz_pass<R, 0>(P, W, N, nl, ls, nsub, ss, rs, sgn); }
// all lines of one direction of the plane: element n of line l at P[l * ls + n * es]
template <bool INVERSE, bool BIG>
__device__ __forceinline__ void yz_lines(float2 *P, const float2 *W, int N, int N1, int N2, int nlines, int ls, int es)
{
    if (!INVERSE) {
        switch (N1) { MDX_DFT_CASES(yz_pass_tw_out, P, W, N, nlines, ls, N2, es, N2 * es, 1.0f) MDX_DFT_CASES_BIG(BIG, yz_pass_tw_out, P, W, N, nlines, ls, N2, es, N2 * es, 1.0f) }
        __syncthreads();
        switch (N2) { MDX_DFT_CASES(yz_pass_plain, P, W, N, nlines, ls, N1, N2 * es, es, 1.0f) MDX_DFT_CASES_BIG(BIG, yz_pass_plain, P, W, N, nlines, ls, N1, N2 * es, es, 1.0f) }
    } else {
        switch (N2) { MDX_DFT_CASES(yz_pass_plain, P, W, N, nlines, ls, N1, N2 * es, es, -1.0f) MDX_DFT_CASES_BIG(BIG, yz_pass_plain, P, W, N, nlines, ls, N1, N2 * es, es, -1.0f) }
        __syncthreads();
        switch (N1) { MDX_DFT_CASES(yz_pass_tw_in, P, W, N, nlines, ls, N2, es, N2 * es, -1.0f) MDX_DFT_CASES_BIG(BIG, yz_pass_tw_in, P, W, N, nlines, ls, N2, es, N2 * es, -1.0f) }
    }
    __syncthreads();
}
#define MDX_YZ_THREADS 512
template <bool ENERGY, bool BIG>
__device__ __forceinline__ void yz_plane(const FftYzArgs &A)
{
    MDX_STAMP(A.d, 5);
    // One workgroup per k_x plane: a few dozen workgroups, each a serial chain of passes over its plane, on as many CUs
    // beside the nonbonded kernel, and the longest link but one of the PME chain. Through the forward passes and the
    // influence function its waves issue ahead of the other kernels' waves on their CU (20 instead of 29 us beside the
    // nonbonded kernel, which loses 3 us on those CUs: +1.2 % on the step). The same in the PME kernels that cover all
    // CUs holds the nonbonded kernel back everywhere (96 -> 115 us) and costs 8 %.
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ float2 fsm[];
    const int K1 = A.K1, K2 = A.K2, ps = K1 + 1;
    float2 *Wy = fsm, *Wz = Wy + K1, *P = Wz + K2;         // P[z * ps + y]
    for (int j = threadIdx.x; j < K1; j += blockDim.x) Wy[j] = A.Wy[j];
    for (int j = threadIdx.x; j < K2; j += blockDim.x) Wz[j] = A.Wz[j];
    const int k0 = A.plane0 + blockIdx.x;
    float2 *plane = A.cgrid + (size_t)k0 * K1 * K2;
    copy_batched<9, float2>(K1 * K2, [&](const int idx) { return plane[idx]; },
                            [&](const int idx, const float2 v) { const int z = idx / K1; P[z * ps + idx - z * K1] = v; });
    __syncthreads();
    yz_lines<false, BIG>(P, Wy, K1, A.N1y, A.N2y, K2, ps, 1);
    yz_lines<false, BIG>(P, Wz, K2, A.N1z, A.N2z, K1, 1, ps);
    {
        // influence function on position (pz, py), which holds the frequency (k1, k2) of the permuted order; energy and
        // virial as in the stand-alone solve kernel
        double e = 0.0, v[6] = { 0, 0, 0, 0, 0, 0 };
        const float *iv = A.d->box.inv;
        const float pi = 3.14159265358979f;
        const float fac = pi * pi / (A.beta * A.beta);
        const float pivol = pi * A.d->box.vol;
        const int K0 = A.K0;
        const float bm0 = A.bm0[k0];
        for (int idx = threadIdx.x; idx < K1 * K2; idx += blockDim.x) {
            const int pz = idx / K1, py = idx - pz * K1;
            const int qy = py / A.N2y, qz = pz / A.N2z;
            const int k1 = qy + A.N1y * (py - qy * A.N2y), k2 = qz + A.N1z * (pz - qz * A.N2z);
            const int m0 = k0;
            const int m1 = (k1 < (K1 + 1) / 2) ? k1 : k1 - K1;
            const int m2 = (k2 < (K2 + 1) / 2) ? k2 : k2 - K2;
            float2 cv = P[pz * ps + py];
            if (m0 != 0 || m1 != 0 || m2 != 0) {
                const float mx = iv[0] * m0 + iv[1] * m1 + iv[2] * m2;
                const float my = iv[3] * m0 + iv[4] * m1 + iv[5] * m2;
                const float mz = iv[6] * m0 + iv[7] * m1 + iv[8] * m2;
                const float msq = mx * mx + my * my + mz * mz;
                const float denom = msq * bm0 * A.bm1[k1] * A.bm2[k2] * pivol;
                const float eterm = MDX_ONE_4PI_EPS0 * __expf(-fac * msq) / denom;
                if (ENERGY) {
                    const float mult = (k0 == 0 || (K0 % 2 == 0 && k0 == K0 / 2)) ? 1.0f : 2.0f;
                    const double em = 0.5 * mult * (double)eterm * ((double)cv.x * cv.x + (double)cv.y * cv.y);
                    e += em;
                    const double vf = 2.0 * (1.0 + fac * msq) / msq;
                    v[0] += em * (1.0 - vf * mx * mx); v[1] += em * (-vf * mx * my); v[2] += em * (-vf * mx * mz);
                    v[3] += em * (1.0 - vf * my * my); v[4] += em * (-vf * my * mz); v[5] += em * (1.0 - vf * mz * mz);
                }
                cv.x *= eterm; cv.y *= eterm;
            } else cv = make_float2(0.0f, 0.0f);
            P[pz * ps + py] = cv;
        }
        if (ENERGY) {
            __shared__ double red[16 * 7];
            const double ev[7] = { e, v[0], v[1], v[2], v[3], v[4], v[5] };
            const double r = block_sum_n<7>(ev, red);
            const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
            if (threadIdx.x == 0) atomicAdd(&MDX_ENER(A.dw, MDX_E_COUL_RECIP), r);
            else if (threadIdx.x < 7) {
                const int k = threadIdx.x - 1;
                atomicAdd(&MDX_VIR(A.dw, map[k][0]), -0.5 * r);
                if (map[k][1] != map[k][0]) atomicAdd(&MDX_VIR(A.dw, map[k][1]), -0.5 * r);
            }
        }
        __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    yz_lines<true, BIG>(P, Wz, K2, A.N1z, A.N2z, K1, 1, ps);
    yz_lines<true, BIG>(P, Wy, K1, A.N1y, A.N2y, K2, ps, 1);
    for (int idx = threadIdx.x; idx < K1 * K2; idx += blockDim.x) {
        const int z = idx / K1, y = idx - z * K1;
        plane[idx] = P[z * ps + y];
    }
}
// Planes up to 96 KB with radices up to 10: eight waves per SIMD (64 registers), two or more workgroups per CU beside the
// nonbonded kernel. Larger planes (one workgroup per CU whatever its registers) and radices up to 16 take the second form.
template <bool ENERGY>
__global__ __launch_bounds__(MDX_YZ_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pme_fft_yz(FftYzArgs A) { yz_plane<ENERGY, false>(A); }
template <bool ENERGY>
__global__ __launch_bounds__(MDX_YZ_THREADS) void k_pme_fft_yz_big(FftYzArgs A) { yz_plane<ENERGY, true>(A); }

// N = N1 N2 with the smallest N1 + N2; false when a factor would exceed 32 (the library plan serves those grids)
static bool fft_split(int N, int &N1, int &N2)
{
    N1 = 1; N2 = N;
    for (int a = 1; a * a <= N; a++) if (N % a == 0) { N1 = a; N2 = N / a; }
    return N2 <= 32 && N <= 512;
}

// 16 lanes per atom gather phi and spline derivatives; reduced with shuffles, one lane adds the force.
__global__ __launch_bounds__(256) void k_pme_gather(int n, const float4 *__restrict__ x, const DevState *__restrict__ d,
                                                    int K0, int K1, int K2, const float *__restrict__ phi, float *__restrict__ f)
{
    MDX_STAMP(d, 9);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int atom = g >> 4, sub = g & 15;
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    float q = 0.0f;
    if (atom < n) {
        const float4 p = x[atom];
        q = p.w;
        if (q != 0.0f) {
            int i0[3]; float t[3];
            frac_index(d->box, p, K0, K1, K2, i0, t);
            float wx[4], wy[4], wz[4], dx[4], dy[4], dz[4];
            bspline4(t[0], wx, dx); bspline4(t[1], wy, dy); bspline4(t[2], wz, dz);
            const int c = sub >> 2, a = sub & 3;
            int ia = i0[0] + a; if (ia >= K0) ia -= K0;
            int ic = i0[2] + c; if (ic >= K2) ic -= K2;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int ib = i0[1] + b; if (ib >= K1) ib -= K1;
                const float ph = phi[((size_t)ic * K1 + ib) * K0 + ia];
                gx = fmaf(dx[a] * wy[b] * wz[c], ph, gx);
                gy = fmaf(wx[a] * dy[b] * wz[c], ph, gy);
                gz = fmaf(wx[a] * wy[b] * dz[c], ph, gz);
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        gx += __shfl_xor(gx, o, 64); gy += __shfl_xor(gy, o, 64); gz += __shfl_xor(gz, o, 64);
    }
    if (atom < n && sub == 0) {
        const float *iv = d->box.inv;
        const float ux = gx * K0, uy = gy * K1, uz = gz * K2;
        // F_e = -q sum_d dE/du_d K_d inv[e][d]; every atom is written (zero for uncharged ones)
        f[3 * atom + 0] = -q * (ux * iv[0] + uy * iv[1] + uz * iv[2]);
        f[3 * atom + 1] = -q * (ux * iv[3] + uy * iv[4] + uz * iv[5]);
        f[3 * atom + 2] = -q * (ux * iv[6] + uy * iv[7] + uz * iv[8]);
    }
}

static void bspline_moduli(int K, std::vector<float> &out)
{
    // |sum_{k=0}^{2} M4(k+1) exp(2 pi i m k / K)|^2 with M4(1)=1/6, M4(2)=2/3, M4(3)=1/6
    const double w[3] = { 1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0 };
    out.resize(K);
    for (int m = 0; m < K; m++) {
        double re = 0, im = 0;
        for (int k = 0; k < 3; k++) {
            const double a = 2.0 * M_PI * m * k / K;
            re += w[k] * cos(a); im += w[k] * sin(a);
        }
        double v = re * re + im * im;
        if (v < 1e-10) v = 1e-10;
        out[m] = (float)v;
    }
}

int mdx_pme_init(mdx_system *s)
{
    mdx_pme_free(s);
    if (s->K[0] <= 0) { s->have_pme = false; return MDX_OK; }
    const size_t nr = (size_t)s->K[0] * s->K[1] * s->K[2];
    // (domains that share the transform exchange the half-complex grid in equal chunks of planes: unused planes at the end)
    const int planes = s->dd_mesh_world > 0 ? s->dd_plane_chunk * s->dd_mesh_world : s->K[0] / 2 + 1;
    const size_t ncx = (size_t)s->K[2] * s->K[1] * planes;
    MDX_CHECK(s, hipMalloc(&s->grid, sizeof(float) * nr));
    MDX_CHECK(s, hipMalloc(&s->phi, sizeof(float) * nr));
    s->grid_clean = false;
    MDX_CHECK(s, hipMalloc(&s->cgrid, sizeof(float2) * ncx));
    MDX_CHECK(s, hipMemset(s->cgrid, 0, sizeof(float2) * ncx));
    for (int d = 0; d < 3; d++) {
        std::vector<float> bm;
        bspline_moduli(s->K[d], bm);
        MDX_CHECK(s, hipMalloc(&s->bmod[d], sizeof(float) * s->K[d]));
        MDX_CHECK(s, hipMemcpy(s->bmod[d], bm.data(), sizeof(float) * s->K[d], hipMemcpyHostToDevice));
    }
    if (hipfftPlan3d(&s->plan_r2c, s->K[2], s->K[1], s->K[0], HIPFFT_R2C) != HIPFFT_SUCCESS ||
        hipfftPlan3d(&s->plan_c2r, s->K[2], s->K[1], s->K[0], HIPFFT_C2R) != HIPFFT_SUCCESS) {
        s->err = "hipfftPlan3d failed";
        return MDX_ERR_HIP;
    }
    hipfftSetStream(s->plan_r2c, s->st_pme);
    hipfftSetStream(s->plan_c2r, s->st_pme);
    s->own_fft = true;
    for (int d = 0; d < 3; d++) {
        int n1, n2;
        if (!fft_split(s->K[d], n1, n2)) s->own_fft = false;
        std::vector<float2> w(s->K[d]);
        for (int j = 0; j < s->K[d]; j++) w[j] = make_float2((float)cos(2.0 * M_PI * j / s->K[d]), (float)-sin(2.0 * M_PI * j / s->K[d]));
        MDX_CHECK(s, hipMalloc((void **)&s->fftW[d], sizeof(float2) * s->K[d]));
        MDX_CHECK(s, hipMemcpy(s->fftW[d], w.data(), sizeof(float2) * s->K[d], hipMemcpyHostToDevice));
    }
    {
        int a1, a2, b1, b2;
        const size_t ldsy = sizeof(float2) * ((size_t)s->K[1] + s->K[2] + (size_t)s->K[2] * (s->K[1] + 1));
        const bool split = s->own_fft && fft_split(s->K[1], a1, a2) && fft_split(s->K[2], b1, b2) && a1 >= 2 && b1 >= 2;
        const int rmax = split ? std::max(std::max(a1, a2), std::max(b1, b2)) : 0;
        s->fft_yz = split && rmax <= MDX_DFT_MAXR_BIG && ldsy <= (size_t)s->yz_lds_limit;
        s->fft_yz_big = s->fft_yz && (rmax > MDX_DFT_MAXR || ldsy > 96 * 1024);
        if (s->fft_yz && ldsy > 48 * 1024) {
            if (s->fft_yz_big) {
                MDX_CHECK(s, hipFuncSetAttribute((const void *)k_pme_fft_yz_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsy));
                MDX_CHECK(s, hipFuncSetAttribute((const void *)k_pme_fft_yz_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsy));
            } else {
                MDX_CHECK(s, hipFuncSetAttribute((const void *)k_pme_fft_yz<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsy));
                MDX_CHECK(s, hipFuncSetAttribute((const void *)k_pme_fft_yz<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsy));
            }
        }
    }
    s->have_pme = true;
    return MDX_OK;
}

void mdx_pme_free(mdx_system *s)
{
    if (s->grid) { (void)hipFree(s->grid); s->grid = nullptr; }
    if (s->phi) { (void)hipFree(s->phi); s->phi = nullptr; }
    if (s->cgrid) { (void)hipFree(s->cgrid); s->cgrid = nullptr; }
    for (int d = 0; d < 3; d++) if (s->bmod[d]) { (void)hipFree(s->bmod[d]); s->bmod[d] = nullptr; }
    for (int d = 0; d < 3; d++) if (s->fftW[d]) { (void)hipFree(s->fftW[d]); s->fftW[d] = nullptr; }
    if (s->plan_r2c) { hipfftDestroy(s->plan_r2c); s->plan_r2c = 0; }
    if (s->plan_c2r) { hipfftDestroy(s->plan_c2r); s->plan_c2r = 0; }
    s->have_pme = false;
}

// Zero the charge grid (unless the last transform's x pass left it zeroed) and spread onto it, on stream `st`.
// bonded_flags (MDX_F_BONDED | MDX_F_POSRES [| MDX_F_PULL], or 0): the bonded terms (and the pull) of a step without energies ride in the same launch.
int mdx_launch_pme_spread(mdx_system *s, hipStream_t st, unsigned bonded_flags)
{
    if (!s->have_pme) return MDX_OK;
    const int K0 = s->K[0], K1 = s->K[1], K2 = s->K[2];
    const size_t nr = (size_t)K0 * K1 * K2;
    if (!s->grid_clean) MDX_CHECK(s, hipMemsetAsync(s->grid, 0, sizeof(float) * nr, st));
    s->grid_clean = false;
    // window sizes from the column geometry (host copy of the box; a drift margin covers the rest)
    const double *bxd = s->h->boxd;
    const double inv0 = 1.0 / bxd[0], inv4 = 1.0 / bxd[4], inv8 = 1.0 / bxd[8];
    const double inv3 = -bxd[3] * inv0 * inv4, inv6 = (bxd[3] * bxd[7] - bxd[4] * bxd[6]) * inv0 * inv4 * inv8, inv7 = -bxd[7] * inv4 * inv8;
    const double delta = 0.12;
    const double wz = 4.0 / (K2 * inv8) + 2 * delta;
    SpreadSlabArgs B;
    // 2 x 2 search columns and 8 planes per workgroup (measured best on the bench workload among 1-3 columns, 4-12 planes)
    B.ncx = s->ncx; B.ncy = s->ncy; B.qx = 2; B.qy = 2;
    while (B.qx * B.qy * (MDX_MAX_COL / 64) > MDX_SPREAD_MAXSC) { if (B.qx >= B.qy) B.qx--; else B.qy--; }
    B.col_scstart = s->col_scstart; B.x = s->x; B.perm = s->perm; B.perm_src = s->perm_src; B.shift_s = s->shift_s;
    B.sbb_lo = s->sbb_lo; B.sbb_hi = s->sbb_hi; B.K0 = K0; B.K1 = K1; B.K2 = K2; B.SP = 8;
    const double wxq = B.qx * bxd[0] / s->ncx + 2 * delta, wyq = B.qy * bxd[4] / s->ncy + 2 * delta;
    B.T0 = std::min(K0, (int)ceil(K0 * (inv0 * wxq + fabs(inv3) * wyq + fabs(inv6) * wz)) + 5);
    B.T1 = std::min(K1, (int)ceil(K1 * (inv4 * wyq + fabs(inv7) * wz)) + 5);
    while ((size_t)B.SP * B.T0 * B.T1 * 4 > 60 * 1024 && B.SP > 2) B.SP--;
    // Unwrapped planes an atom of the brick can touch: -1 .. K2 + 3 with a drift of less than a plane (the
    // margin of 0.12 nm); the slabs cover -4 .. K2 + 7 at least, three planes (0.47 nm) more on either side.
    B.tlo0 = -4;
    const int nslab = (K2 + 12 + B.SP - 1) / B.SP;
    B.delta = (float)delta; B.grid = s->grid; B.v_home = s->dd_on ? s->v : nullptr;
    size_t ldss = (size_t)B.SP * B.T0 * B.T1 * 4 + (size_t)B.SP * 16;
    const int nq = ((s->ncx + B.qx - 1) / B.qx) * ((s->ncy + B.qy - 1) / B.qy);
    BondedArgs G; G.nblocks = 0;
    if (bonded_flags) mdx_bonded_args(s, bonded_flags, G);
    const bool with_pull = (bonded_flags & MDX_F_PULL) && s->pull_on;
    if (with_pull) {
        // (the record of a logging step is counted here, as mdx_launch_pull does)
        const bool log_now = s->step % s->pull_nstout == 0;
        PullArgs P; mdx_pull_args(s, log_now, false, P);
        const int brows = (G.nblocks + 1 + nq - 1) / nq;
        ldss = std::max(ldss, sizeof(float) * 3 * MDX_BWIN + sizeof(int));
        hipLaunchKernelGGL(k_pme_spread_slab_bonded_pull, dim3(nq, nslab + brows), dim3(256), ldss, st, B, (const DevState *)s->d, G, brows,
                           P, (const float4 *)s->v, s->f, s->d);
        if (log_now) s->pull_log_count++;
    } else if (G.nblocks > 0) {
        const int brows = (G.nblocks + nq - 1) / nq;
        ldss = std::max(ldss, sizeof(float) * 3 * MDX_BWIN + sizeof(int));
        hipLaunchKernelGGL(k_pme_spread_slab_bonded, dim3(nq, nslab + brows), dim3(256), ldss, st, B, (const DevState *)s->d, G, brows);
    } else hipLaunchKernelGGL(k_pme_spread_slab, dim3(nq, nslab), dim3(256), ldss, st, B, (const DevState *)s->d);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}

// The PME chain on the PME stream: spreading (unless the caller has done it), transforms with the influence function,
// gather; the gather's completion signal is the event the main stream waits for. `s->pme_part` (stage timing only)
// restricts the chain to the spreading (1) or the gather (2). `stage` cuts the chain where domains that share the
// transform exchange the half-complex grid: 1 ends behind the x forward pass, 2 is the fused y/z kernel on this domain's
// planes, 3 the x inverse pass and the gather (0: everything).
int mdx_launch_pme(mdx_system *s, bool energy, bool spread_done, int stage, unsigned bonded_flags)
{
    if (!s->have_pme) return MDX_OK;
    const int K0 = s->K[0], K1 = s->K[1], K2 = s->K[2];
    const size_t ncx = (size_t)K2 * K1 * (K0 / 2 + 1);
    const int nb = (int)(((size_t)s->n * 16 + 255) / 256);
    if (stage != 0 && !(s->own_fft && s->fft_yz)) { s->err = "mdx_launch_pme: this grid has no plane-wise transform to share among domains"; return MDX_ERR_UNSUPPORTED; }
    if (s->pme_part == 2) {
        hipLaunchKernelGGL(k_pme_gather, dim3(nb), dim3(256), 0, s->st_pme, s->n, (const float4 *)s->x, (const DevState *)s->d, K0, K1, K2,
                           (const float *)(s->own_fft ? s->phi : s->grid), s->f_pme);
        MDX_CHECK(s, hipGetLastError());
        return MDX_OK;
    }
    if (stage <= 1 && !spread_done) { int rc = mdx_launch_pme_spread(s, s->st_pme, energy ? 0u : bonded_flags); if (rc) return rc; }
    if (s->pme_part == 1) return MDX_OK;
    const float *potential = s->grid;       // the library path transforms in place
    if (s->own_fft && (stage != 0 || !(s->dbg & 512))) {
        FftArgs F;
        F.grid_in = s->grid; F.grid_out = s->phi; F.cgrid = s->cgrid; F.K0 = K0; F.K1 = K1; F.K2 = K2; F.K0c = K0 / 2 + 1;
        F.d = s->d; F.dw = s->d; F.bm0 = s->bmod[0]; F.bm1 = s->bmod[1]; F.bm2 = s->bmod[2]; F.beta = s->beta;
        auto lds = [](const FftArgs &f) { return sizeof(float2) * ((size_t)f.N + 2 * (size_t)f.nl * (f.N + 1)); };
        const int ynl = 8, xnl = 16;        // lines per workgroup: 64 and 128 bytes contiguous per access of the strided side
        const int nchunk = (F.K0c + ynl - 1) / ynl;
        // one workgroup per k_x plane for the y and z passes where a plane fits into LDS and both dimensions split into
        // two factors <= 10 (mdx_debug_set key 2 bit 0x2000000 keeps the pass-per-launch form: cross-check and A/B)
        const bool fused_yz = stage != 0 || (s->fft_yz && !(s->dbg & 0x2000000));
        F.tr = fused_yz ? 1 : 0;
        if (stage <= 1) {
            // x forward
            F.N = K0; fft_split(K0, F.N1, F.N2); F.W = s->fftW[0]; F.nl = xnl;
            hipLaunchKernelGGL((k_pme_fft<0, false>), dim3((K1 * K2 + F.nl - 1) / F.nl), dim3(256), lds(F), s->st_pme, F);
            s->grid_clean = true;
            if (stage == 1) { MDX_CHECK(s, hipGetLastError()); return MDX_OK; }
        }
        if (fused_yz && stage <= 2) {
            FftYzArgs Y;
            Y.cgrid = s->cgrid; Y.Wy = s->fftW[1]; Y.Wz = s->fftW[2]; Y.K0 = K0; Y.K1 = K1; Y.K2 = K2; Y.K0c = F.K0c;
            Y.plane0 = stage == 2 ? s->dd_plane_first : 0;
            const int nplanes = stage == 2 ? s->dd_plane_count : F.K0c;
            fft_split(K1, Y.N1y, Y.N2y); fft_split(K2, Y.N1z, Y.N2z);
            Y.d = s->d; Y.dw = s->d; Y.bm0 = s->bmod[0]; Y.bm1 = s->bmod[1]; Y.bm2 = s->bmod[2]; Y.beta = s->beta;
            const size_t ldsy = sizeof(float2) * ((size_t)K1 + K2 + (size_t)K2 * (K1 + 1));
            // (512 threads: on energy steps the workgroups start only when the energy kernel's waves retire; with 256 threads
            // they run beside it, twice as long, and the bench is the same within its noise: 1062 against 1059 ns/day)
            if (nplanes > 0) {
                if (s->fft_yz_big) {
                    if (energy) hipLaunchKernelGGL(k_pme_fft_yz_big<true>, dim3(nplanes), dim3(MDX_YZ_THREADS), ldsy, s->st_pme, Y);
                    else hipLaunchKernelGGL(k_pme_fft_yz_big<false>, dim3(nplanes), dim3(MDX_YZ_THREADS), ldsy, s->st_pme, Y);
                } else if (energy) hipLaunchKernelGGL(k_pme_fft_yz<true>, dim3(nplanes), dim3(MDX_YZ_THREADS), ldsy, s->st_pme, Y);
                else hipLaunchKernelGGL(k_pme_fft_yz<false>, dim3(nplanes), dim3(MDX_YZ_THREADS), ldsy, s->st_pme, Y);
            }
            if (stage == 2) { MDX_CHECK(s, hipGetLastError()); return MDX_OK; }
        } else if (stage == 0) {
            // y forward
            F.N = K1; fft_split(K1, F.N1, F.N2); F.W = s->fftW[1]; F.nl = ynl;
            hipLaunchKernelGGL((k_pme_fft<1, false>), dim3(K2 * nchunk), dim3(256), lds(F), s->st_pme, F);
            // z forward, influence function, z inverse
            F.N = K2; fft_split(K2, F.N1, F.N2); F.W = s->fftW[2]; F.nl = ynl;
            if (energy) hipLaunchKernelGGL((k_pme_fft<2, true>), dim3(K1 * nchunk), dim3(256), lds(F), s->st_pme, F);
            else hipLaunchKernelGGL((k_pme_fft<2, false>), dim3(K1 * nchunk), dim3(256), lds(F), s->st_pme, F);
            // y inverse
            F.N = K1; fft_split(K1, F.N1, F.N2); F.W = s->fftW[1]; F.nl = ynl;
            hipLaunchKernelGGL((k_pme_fft<3, false>), dim3(K2 * nchunk), dim3(256), lds(F), s->st_pme, F);
        }
        // x inverse
        F.N = K0; fft_split(K0, F.N1, F.N2); F.W = s->fftW[0]; F.nl = xnl;
        hipLaunchKernelGGL((k_pme_fft<4, false>), dim3((K1 * K2 + F.nl - 1) / F.nl), dim3(256), lds(F), s->st_pme, F);
        potential = s->phi;
    } else {
        hipfftSetStream(s->plan_r2c, s->st_pme); hipfftSetStream(s->plan_c2r, s->st_pme);
        if (hipfftExecR2C(s->plan_r2c, s->grid, (hipfftComplex *)s->cgrid) != HIPFFT_SUCCESS) { s->err = "hipfftExecR2C failed"; return MDX_ERR_HIP; }
        const int nbs = (int)((ncx + 255) / 256);
        if (energy)
            hipLaunchKernelGGL(k_pme_solve<true>, dim3(nbs), dim3(256), 0, s->st_pme, s->cgrid, s->d, s->d, K0, K1, K2, s->bmod[0], s->bmod[1], s->bmod[2], s->beta);
        else
            hipLaunchKernelGGL(k_pme_solve<false>, dim3(nbs), dim3(256), 0, s->st_pme, s->cgrid, s->d, s->d, K0, K1, K2, s->bmod[0], s->bmod[1], s->bmod[2], s->beta);
        if (hipfftExecC2R(s->plan_c2r, (hipfftComplex *)s->cgrid, s->grid) != HIPFFT_SUCCESS) { s->err = "hipfftExecC2R failed"; return MDX_ERR_HIP; }
    }
    // the completion signal of the last kernel of the chain is the event the main stream waits for
    hipExtLaunchKernelGGL(k_pme_gather, dim3(nb), dim3(256), 0, s->st_pme, nullptr, s->ev_pme, 0, s->n, (const float4 *)s->x,
                          (const DevState *)s->d, K0, K1, K2, potential, s->f_pme);
    s->pme_signalled = true;
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
