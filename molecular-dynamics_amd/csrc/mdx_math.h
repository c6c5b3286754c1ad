// Device helpers shared by the mdx HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MDX_ONE_4PI_EPS0 138.935458f
#define MDX_BOLTZ 0.0083144626
#define MDX_PRESFAC 16.6054
#define MDX_2_SQRTPI 1.1283791670955126f

// erfc(z) = exp(-z^2) * t * P(t), t = 1/(1 + 0.41 z); fit by tools/fit_erfc.py
// (max relative error 3.5e-7 in float32 on 0 <= z <= 4.4).
__device__ __forceinline__ float mdx_erfcx_poly(float z)
{
    const float t = __builtin_amdgcn_rcpf(fmaf(0.41f, z, 1.0f));
    float p = 7.959749251e-02f;
    p = fmaf(p, t, -3.893341305e-01f);
    p = fmaf(p, t, 6.172475282e-01f);
    p = fmaf(p, t, -3.034404266e-01f);
    p = fmaf(p, t, 3.679784521e-01f);
    p = fmaf(p, t, 1.573482378e-01f);
    p = fmaf(p, t, 2.398621772e-01f);
    p = fmaf(p, t, 2.307406958e-01f);
    return p * t;
}

__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float3 xyz(float4 a) { return make_float3(a.x, a.y, a.z); }

// Sum over the 64 lanes of a wave; every lane receives the total.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in thread 0.
__device__ __forceinline__ double block_sum_d(double v, double *smem /* >= 16 doubles */)
{
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) r += smem[i];
    }
    return r;
}

// Block-wide sums of N values at once (blockDim.x <= 1024, a multiple of 64, and >= N): thread k < N returns the sum of
// v[k], the other threads 0. One pair of barriers for all N values; N calls of block_sum_d cost N pairs, and the kernels
// of the energy steps sum seven to fourteen values per workgroup.
template <int N>
__device__ __forceinline__ double block_sum_n(const double (&v)[N], double *smem /* >= 16 * N doubles */)
{
    double w[N];
#pragma unroll
    for (int k = 0; k < N; k++) w[k] = wave_sum_d(v[k]);
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) smem[wv * N + k] = w[k];
    }
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < N) for (int i = 0; i < nw; i++) r += smem[i * N + threadIdx.x];
    return r;
}

// Triclinic box in device memory: rows a=(h0,0,0) b=(h3,h4,0) c=(h6,h7,h8); inv = h^-1 with
// fractional s = x . inv (s_d = sum_e x_e inv[3e+d]).
struct DevBox {
    float h[9];
    float inv[9];
    float vol;
    float pad[5];
};

// Minimum-image displacement for vectors already close to the minimum image (bonded terms,
// restraints, pull): successive reduction along c, b, a followed by a 27-neighbour polish.
__device__ __forceinline__ float3 mdx_min_image(const DevBox &b, float3 d)
{
    float s = rintf(d.z / b.h[8]);
    d.x -= s * b.h[6]; d.y -= s * b.h[7]; d.z -= s * b.h[8];
    s = rintf(d.y / b.h[4]);
    d.x -= s * b.h[3]; d.y -= s * b.h[4];
    s = rintf(d.x / b.h[0]);
    d.x -= s * b.h[0];
    float best = dot3(d, d);
    float3 r = d;
    for (int k = -1; k <= 1; k++)
        for (int j = -1; j <= 1; j++)
            for (int i = -1; i <= 1; i++) {
                float3 c = make_float3(d.x + i * b.h[0] + j * b.h[3] + k * b.h[6],
                                       d.y + j * b.h[4] + k * b.h[7], d.z + k * b.h[8]);
                float r2 = dot3(c, c);
                if (r2 < best) { best = r2; r = c; }
            }
    return r;
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011), used for the stochastic terms of
// the V-rescale thermostat and C-rescale barostat so no generator state lives on the host.
struct Philox {
    uint32_t c[4];
    uint32_t k[2];
};
__host__ __device__ inline void philox_round(uint32_t *c, const uint32_t *k)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__host__ __device__ inline void philox4x32(uint32_t *out, uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key)
{
    uint32_t c[4] = { (uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32) };
    uint32_t k[2] = { (uint32_t)key, (uint32_t)(key >> 32) };
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// Stream of uniform doubles in (0,1) from Philox blocks.
struct RngStream {
    uint64_t key, hi, lo;
    uint32_t buf[4];
    int have;
    __host__ __device__ RngStream(uint64_t key_, uint64_t stream_id, uint64_t step) : key(key_), hi(stream_id), lo(step << 16), have(0) {}
    __host__ __device__ double uniform()
    {
        if (have < 2) { philox4x32(buf, lo++, hi, key); have = 4; }
        const uint64_t a = buf[4 - have], b = buf[5 - have];
        have -= 2;
        const uint64_t m = ((a << 32) | b) >> 11; // 53 bits
        return ((double)m + 0.5) * (1.0 / 9007199254740992.0);
    }
    __host__ __device__ double gauss()
    {
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    // Gamma(shape a >= 1, scale 1): Marsaglia-Tsang squeeze.
    __host__ __device__ double gamma(double a)
    {
        const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (int it = 0; it < 64; it++) {
            double xg, v;
            do { xg = gauss(); v = 1.0 + c * xg; } while (v <= 0.0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * xg * xg * xg * xg) return d * v;
            if (log(u) < 0.5 * xg * xg + d * (1.0 - v + log(v))) return d * v;
        }
        return d;
    }
    // Sum of n squared unit Gaussians.
    __host__ __device__ double chi2(double n)
    {
        if (n <= 0.0) return 0.0;
        if (n < 2.0) { double s = 0; for (int i = 0; i < (int)n; i++) { double g = gauss(); s += g * g; } return s; }
        return 2.0 * gamma(0.5 * n);
    }
};

// Lattice shift (multiples of the three box vectors) applied to an atom when it was wrapped into
// the brick-shaped search cell, packed into one int (each component in -512..511).
__device__ __forceinline__ int pack_shift(int a, int b, int c) { return (a + 512) | ((b + 512) << 10) | ((c + 512) << 20); }
__device__ __forceinline__ void unpack_shift(int p, int &a, int &b, int &c)
{
    a = (p & 1023) - 512; b = ((p >> 10) & 1023) - 512; c = ((p >> 20) & 1023) - 512;
}
