// VALU issue rate of a gfx950 SIMD in real shader clocks: plain v_fma_f32 written as inline asm (nothing the compiler can
// pack or fuse), 1-8 independent chains per wave, 1-8 waves per SIMD, timed inside the kernel with s_memtime (shader
// clock) and s_memrealtime (100 MHz), so that the result does not depend on a nominal frequency.
// Question: does a SIMD retire one wave64 VALU instruction every 4 clocks (16 lanes wide) or every 2?
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate2 valu_rate2.hip ; run: ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a2), "v"(b2))

template <int CH, bool PK>
__global__ __launch_bounds__(256) void k_rate(float *out, unsigned long long *clk, float a, float b, int iters)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    float x[8]; v2f x2[8];
    const v2f a2 = { a, a }, b2 = { b, b };
    for (int c = 0; c < 8; c++) { x[c] = threadIdx.x * 0.001f + c; x2[c] = (v2f){ x[c], x[c] + 1.0f }; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CH; c++) { if (PK) PKFMA(x2[c]); else FMA(x[c]); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0; for (int c = 0; c < CH; c++) s += PK ? x2[c].x + x2[c].y : x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0;
    }
}

template <int CH, bool PK> void run(float *d, unsigned long long *dc, int wpe, int iters)
{
    const int blocks = 256 * wpe;          // wpe workgroups of four waves per CU: wpe waves per SIMD
    std::vector<unsigned long long> h(2 * blocks * 4);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_rate<CH, PK>), dim3(blocks), dim3(256), 0, 0, d, dc, 0.999f, 0.001f, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), dc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::vector<double> cyc, mhz;
    for (int w = 0; w < blocks * 4; w++) { cyc.push_back((double)h[2 * w]); mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
    const double c = cyc[cyc.size() / 2];
    const double per_wave = c / ((double)iters * 16 * CH);       // clocks between two instructions of one wave
    printf("  %s chains %d: %6.2f clocks per instruction per wave, %5.2f per SIMD (%d waves), clock %.0f MHz\n", PK ? "v_pk_fma_f32" : "v_fma_f32   ",
           CH, per_wave, per_wave / wpe, wpe, mhz[mhz.size() / 2]);
}

int main()
{
    float *d; unsigned long long *dc;
    hipMalloc(&d, sizeof(float) * 256 * 8 * 256); hipMalloc(&dc, sizeof(unsigned long long) * 2 * 256 * 8 * 4);
    const int iters = 4000;
    // keep the GPU busy for a while first: the shader clock ramps with load
    for (int k = 0; k < 200; k++) hipLaunchKernelGGL((k_rate<4, false>), dim3(1024), dim3(256), 0, 0, d, dc, 0.999f, 0.001f, iters);
    hipDeviceSynchronize();
    for (int wpe = 1; wpe <= 8; wpe = wpe < 4 ? wpe + 1 : wpe + 2) {
        printf("%d waves per SIMD\n", wpe);
        run<1, false>(d, dc, wpe, iters); run<2, false>(d, dc, wpe, iters); run<4, false>(d, dc, wpe, iters); run<8, false>(d, dc, wpe, iters);
        run<1, true>(d, dc, wpe, iters); run<4, true>(d, dc, wpe, iters);
    }
    return 0;
}
