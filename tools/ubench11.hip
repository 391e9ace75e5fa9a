// Cycles per MFMA: v_mfma_f32_16x16x32_f16 (8 halves per lane and operand) against
// v_mfma_f32_16x16x16_f16 (4 halves), one wave per SIMD issuing them back to back on four
// independent accumulators.  Would a half slice (16 window positions) cost half a slice?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ __launch_bounds__(256) void loop(float *out, const float *in, int iters, long long *cycles)
{
    f16x8 a8, b8;
    for (int i = 0; i < 8; i++) { a8[i] = (_Float16)in[threadIdx.x + i]; b8[i] = (_Float16)in[threadIdx.x + 8 + i]; }
    f16x4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
    f32x4 acc[4] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (K == 32) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[u & 3], 0, 0, 0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    asm volatile("" : "+v"(s));
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main()
{
    float *in, *out;
    long long *cyc;
    CK(hipMalloc(&in, 4096)); CK(hipMemset(in, 0, 4096)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    for (int k : {32, 16}) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            if (k == 32) hipLaunchKernelGGL(loop<32>, dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
            else hipLaunchKernelGGL(loop<16>, dim3(256), dim3(256), 0, 0, out, in, iters, cyc);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double n = (double)iters * 16;
        printf("K=%2d: %.3f ms for %d MFMAs per wave, one wave per SIMD: %.1f ns each, %.1f counter ticks each; %.0f TFLOP/s chip-wide\n", k, ms,
               iters * 16, ms * 1e6 / n, (double)c / n, 256.0 * 4 * n * 16 * 16 * k * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
