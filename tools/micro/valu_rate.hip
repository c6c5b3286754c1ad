// VALU issue-rate micro-benchmark (wave64, gfx950): how many v_fma_f32 per clock a SIMD retires as a function of the
// resident waves per SIMD and of the independent dependency chains per wave, and what a nonbonded-like mix
// (v_rsq_f32, LDS table look-up, compare + select) does to it.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(256) void k_fma(float *out, float a, float b, int iters)
{
    float x[CH];
    for (int c = 0; c < CH; c++) x[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; i += 8)         // eight rounds per loop iteration: the loop's own instructions are a few per cent
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) x[c] = __builtin_fmaf(x[c], a, b);
    float s = 0; for (int c = 0; c < CH; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 32 VALU instructions per iteration in a mix like a pair tile: 1 rsq, 1 LDS read, 1 compare, 1 select, 28 fma/mul
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b, int iters)
{
    __shared__ float tab[4096];
    for (int k = threadIdx.x; k < 4096; k += 256) tab[k] = k * 1e-4f;
    __syncthreads();
    float x = threadIdx.x * 0.001f + 1.0f, y = 0.5f, acc0 = 0, acc1 = 0, acc2 = 0;
    for (int i = 0; i < iters; i++) {
        const float r2 = __builtin_fmaf(x, x, __builtin_fmaf(y, y, 1e-3f));
        const float ri = __builtin_amdgcn_rsqf(r2);
        const float t = tab[(unsigned)(r2 * 37.0f) & 4095];
        float f = ri * ri;
        f = f * f * f; f = __builtin_fmaf(f, a, b); f = __builtin_fmaf(f, ri, t); f = f * ri; f = __builtin_fmaf(f, a, b);
        f = __builtin_fmaf(f, f, a); f = __builtin_fmaf(f, b, a); f = __builtin_fmaf(f, a, b); f = __builtin_fmaf(f, a, b);
        f = r2 < 2.0f ? f : 0.0f;
        acc0 = __builtin_fmaf(f, x, acc0); acc1 = __builtin_fmaf(f, y, acc1); acc2 = __builtin_fmaf(f, a, acc2);
        x = __builtin_fmaf(x, 0.9999f, 1e-4f); y = __builtin_fmaf(y, 0.9998f, 2e-4f);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2;
}
template <typename K> double run(K kern, float *d, int blocks, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    float *d; hipMalloc(&d, sizeof(float) * 256 * 8192);
    const int iters = 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("clock %.2f GHz, %d CUs\n", ghz, p.multiProcessorCount);
    for (int wpe = 1; wpe <= 6; wpe++) {
        const int blocks = 256 * wpe;   // wpe blocks of 4 waves per CU = wpe waves per SIMD
        const double n1 = (double)iters * blocks * 4;      // wave-iterations
        double m1 = run(k_fma<1>, d, blocks, iters), m2 = run(k_fma<2>, d, blocks, iters), m4 = run(k_fma<4>, d, blocks, iters), m8 = run(k_fma<8>, d, blocks, iters);
        double mm = run(k_mix, d, blocks, iters);
        // cycles per wave-instruction per SIMD = time * clock / (instructions per SIMD)
        auto cyc = [&](double ms, int per_iter) { return ms * 1e-3 * ghz * 1e9 / (n1 * per_iter / 1024.0); };
        printf("waves/SIMD %d: cycles per VALU instruction per SIMD: 1 chain %.2f  2 chains %.2f  4 chains %.2f  8 chains %.2f   pair-like mix (~24 VALU + rsq + LDS read per iteration) %.1f cycles per iteration\n",
               wpe, cyc(m1, 1), cyc(m2, 2), cyc(m4, 4), cyc(m8, 8), mm * 1e-3 * ghz * 1e9 / (n1 / 1024.0));
    }
    return 0;
}
