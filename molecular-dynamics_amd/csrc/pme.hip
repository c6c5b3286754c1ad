// Smooth particle-mesh Ewald, order 4 (Essmann et al. 1995): spread -> rocFFT R2C -> solve ->
// C2R -> gather. Replaces what `coulombtype = PME`, `pme_order = 4`, `fourierspacing = 0.16`
// (/root/reference/mdp/production.mdp:27-29) switch on inside gmx mdrun.
// Grid layout: (nx, ny, nz) with z fastest; fractional coordinates s = x . inv(box) so the
// triclinic dodecahedron needs no special casing; reciprocal vectors m = inv(box) . m'.
#include "mdx_internal.h"

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

// 16 lanes per atom: lane (a = sub>>2 over x, c = sub&3 over z) loops over the 4 y offsets, so the
// four lanes of a quad add to 4 consecutive floats of one z-row.
__global__ __launch_bounds__(256) void k_pme_spread(int n, const float4 *__restrict__ x, const DevState *__restrict__ d,
                                                    int K0, int K1, int K2, float *__restrict__ grid)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int atom = g >> 4, sub = g & 15;
    if (atom >= n) return;
    const float4 p = x[atom];
    if (p.w == 0.0f) return;
    int i0[3]; float t[3];
    frac_index(d->box, p, K0, K1, K2, i0, t);
    float wx[4], wy[4], wz[4], dws[4];
    bspline4(t[0], wx, dws); bspline4(t[1], wy, dws); bspline4(t[2], wz, dws);
    const int a = sub >> 2, c = sub & 3;
    int ia = i0[0] + a; if (ia >= K0) ia -= K0;
    int ic = i0[2] + c; if (ic >= K2) ic -= K2;
    const float qa = p.w * wx[a] * wz[c];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int ib = i0[1] + b; if (ib >= K1) ib -= K1;
        atomicAdd(&grid[((size_t)ia * K1 + ib) * K2 + ic], qa * wy[b]);
    }
}

// Multiply the transformed grid by the Ewald influence function; optionally accumulate the
// reciprocal energy and virial. cgrid is (K0, K1, K2/2+1) complex.
template <bool ENERGY>
__global__ __launch_bounds__(256) void k_pme_solve(float2 *__restrict__ cgrid, const DevState *__restrict__ d, DevState *__restrict__ dw,
                                                   int K0, int K1, int K2, const float *__restrict__ bm0,
                                                   const float *__restrict__ bm1, const float *__restrict__ bm2, float beta)
{
    __shared__ double red[16];
    const int K2c = K2 / 2 + 1;
    const size_t total = (size_t)K0 * K1 * K2c;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0, v[6] = { 0, 0, 0, 0, 0, 0 };
    if (g < total) {
        const int k2 = (int)(g % K2c);
        const int k1 = (int)((g / K2c) % K1);
        const int k0 = (int)(g / ((size_t)K2c * K1));
        // signed frequencies; the Nyquist index of an even dimension counts as -K/2
        const int m0 = (k0 < (K0 + 1) / 2) ? k0 : k0 - K0;
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
                // half-complex storage: planes 0 < k2 < K2/2 (and K2 odd: all k2>0) stand for two terms
                const float mult = (k2 == 0 || (K2 % 2 == 0 && k2 == K2 / 2)) ? 1.0f : 2.0f;
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
        double r = block_sum_d(e, red);
        if (threadIdx.x == 0) atomicAdd(&dw->ener[MDX_E_COUL_RECIP], r);
        const int map[6][2] = { { 0, 0 }, { 1, 3 }, { 2, 6 }, { 4, 4 }, { 5, 7 }, { 8, 8 } };
        for (int k = 0; k < 6; k++) {
            r = block_sum_d(v[k], red);
            if (threadIdx.x == 0) {
                atomicAdd(&dw->vir[map[k][0]], -0.5 * r);
                if (map[k][1] != map[k][0]) atomicAdd(&dw->vir[map[k][1]], -0.5 * r);
            }
        }
    }
}

// 16 lanes per atom gather phi and spline derivatives; reduced with shuffles, one lane adds the force.
__global__ __launch_bounds__(256) void k_pme_gather(int n, const float4 *__restrict__ x, const DevState *__restrict__ d,
                                                    int K0, int K1, int K2, const float *__restrict__ phi, float *__restrict__ f)
{
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
            const int a = sub >> 2, c = sub & 3;
            int ia = i0[0] + a; if (ia >= K0) ia -= K0;
            int ic = i0[2] + c; if (ic >= K2) ic -= K2;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int ib = i0[1] + b; if (ib >= K1) ib -= K1;
                const float ph = phi[((size_t)ia * K1 + ib) * K2 + ic];
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
    if (atom < n && sub == 0 && q != 0.0f) {
        const float *iv = d->box.inv;
        const float ux = gx * K0, uy = gy * K1, uz = gz * K2;
        // F_e = -q sum_d dE/du_d K_d inv[e][d]
        f[3 * atom + 0] += -q * (ux * iv[0] + uy * iv[1] + uz * iv[2]);
        f[3 * atom + 1] += -q * (ux * iv[3] + uy * iv[4] + uz * iv[5]);
        f[3 * atom + 2] += -q * (ux * iv[6] + uy * iv[7] + uz * iv[8]);
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
    const size_t ncx = (size_t)s->K[0] * s->K[1] * (s->K[2] / 2 + 1);
    MDX_CHECK(s, hipMalloc(&s->grid, sizeof(float) * nr));
    MDX_CHECK(s, hipMalloc(&s->cgrid, sizeof(float2) * ncx));
    for (int d = 0; d < 3; d++) {
        std::vector<float> bm;
        bspline_moduli(s->K[d], bm);
        MDX_CHECK(s, hipMalloc(&s->bmod[d], sizeof(float) * s->K[d]));
        MDX_CHECK(s, hipMemcpy(s->bmod[d], bm.data(), sizeof(float) * s->K[d], hipMemcpyHostToDevice));
    }
    if (hipfftPlan3d(&s->plan_r2c, s->K[0], s->K[1], s->K[2], HIPFFT_R2C) != HIPFFT_SUCCESS ||
        hipfftPlan3d(&s->plan_c2r, s->K[0], s->K[1], s->K[2], HIPFFT_C2R) != HIPFFT_SUCCESS) {
        s->err = "hipfftPlan3d failed";
        return MDX_ERR_HIP;
    }
    hipfftSetStream(s->plan_r2c, s->stream);
    hipfftSetStream(s->plan_c2r, s->stream);
    s->have_pme = true;
    return MDX_OK;
}

void mdx_pme_free(mdx_system *s)
{
    if (s->grid) { (void)hipFree(s->grid); s->grid = nullptr; }
    if (s->cgrid) { (void)hipFree(s->cgrid); s->cgrid = nullptr; }
    for (int d = 0; d < 3; d++) if (s->bmod[d]) { (void)hipFree(s->bmod[d]); s->bmod[d] = nullptr; }
    if (s->plan_r2c) { hipfftDestroy(s->plan_r2c); s->plan_r2c = 0; }
    if (s->plan_c2r) { hipfftDestroy(s->plan_c2r); s->plan_c2r = 0; }
    s->have_pme = false;
}

int mdx_launch_pme(mdx_system *s, bool energy)
{
    if (!s->have_pme) return MDX_OK;
    const int K0 = s->K[0], K1 = s->K[1], K2 = s->K[2];
    const size_t nr = (size_t)K0 * K1 * K2;
    const size_t ncx = (size_t)K0 * K1 * (K2 / 2 + 1);
    MDX_CHECK(s, hipMemsetAsync(s->grid, 0, sizeof(float) * nr, s->stream));
    const int nb = (int)(((size_t)s->n * 16 + 255) / 256);
    hipLaunchKernelGGL(k_pme_spread, dim3(nb), dim3(256), 0, s->stream, s->n, s->x, s->d, K0, K1, K2, s->grid);
    if (hipfftExecR2C(s->plan_r2c, s->grid, (hipfftComplex *)s->cgrid) != HIPFFT_SUCCESS) { s->err = "hipfftExecR2C failed"; return MDX_ERR_HIP; }
    const int nbs = (int)((ncx + 255) / 256);
    if (energy)
        hipLaunchKernelGGL(k_pme_solve<true>, dim3(nbs), dim3(256), 0, s->stream, s->cgrid, s->d, s->d, K0, K1, K2, s->bmod[0], s->bmod[1], s->bmod[2], s->beta);
    else
        hipLaunchKernelGGL(k_pme_solve<false>, dim3(nbs), dim3(256), 0, s->stream, s->cgrid, s->d, s->d, K0, K1, K2, s->bmod[0], s->bmod[1], s->bmod[2], s->beta);
    if (hipfftExecC2R(s->plan_c2r, (hipfftComplex *)s->cgrid, s->grid) != HIPFFT_SUCCESS) { s->err = "hipfftExecC2R failed"; return MDX_ERR_HIP; }
    hipLaunchKernelGGL(k_pme_gather, dim3(nb), dim3(256), 0, s->stream, s->n, s->x, s->d, K0, K1, K2, s->grid, s->f);
    MDX_CHECK(s, hipGetLastError());
    return MDX_OK;
}
