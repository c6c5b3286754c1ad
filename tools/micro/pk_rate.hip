// Issue-rate micro-benchmark: scalar v_fma_f32 against packed v_pk_fma_f32 (wave64, gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CH>
__global__ __launch_bounds__(256) void k_scalar(float *out, float a, float b, int iters)
{
    float x[CH];
    for (int c = 0; c < CH; c++) x[c] = threadIdx.x * 0.001f + c;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = __builtin_fmaf(x[c], a, b);
    float s = 0; for (int c = 0; c < CH; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k_packed(float *out, float a, float b, int iters)
{
    v2f x[CH];
    const v2f va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
    for (int c = 0; c < CH; c++) { x[c].x = threadIdx.x * 0.001f + c; x[c].y = threadIdx.x * 0.002f + c; }
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = x[c] * va + vb;
    float s = 0; for (int c = 0; c < CH; c++) s += x[c].x + x[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    float *d; hipMalloc(&d, sizeof(float) * 256 * 4096);
    const int iters = 20000;
    for (int wpe = 1; wpe <= 4; wpe *= 2) {
        const int blocks = 256 * wpe;   // wpe blocks of 4 waves per CU = wpe waves per SIMD
        double ms1 = run(k_scalar<8>, d, blocks, iters);
        double ms2 = run(k_packed<8>, d, blocks, iters);
        const double f1 = 2.0 * 8 * iters * 256.0 * blocks, f2 = 2.0 * f1;
        printf("waves/SIMD %d: scalar fma %.1f TFLOP/s (%.3f ms)   packed fma %.1f TFLOP/s (%.3f ms)\n", wpe, f1 / ms1 * 1e-9, ms1, f2 / ms2 * 1e-9, ms2);
    }
    return 0;
}
