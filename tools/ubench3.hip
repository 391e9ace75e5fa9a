// f32 MFMA issue-rate calibration: register-only operands, 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma16(float* out, const float* in, int iters)
{
    float a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[threadIdx.x + 256 * (i + 4)]; }
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], b[q], acc[q], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// same MFMA stream, operands re-read from LDS every iteration (the FIR kernel's pattern),
// ping-pong register sets
__global__ __launch_bounds__(256) void mfma16_lds(float* out, const float* in, int iters)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ap = lds + 3000 + (lane >> 4) + 15 - (lane & 15);
    const float* bp = lds + wave * 700 + 18 * (lane & 15) + (lane >> 4);
    f32x4 acc[4] = {};
    float a0[4], b0[4][4], a1[4], b1[4][4];
    auto load = [&](float (&a)[4], float (&b)[4][4], int u) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
            a[v] = ap[(u & 7) * 16 + 4 * v];
#pragma unroll
            for (int q = 0; q < 4; q++) b[q][v] = bp[(u & 7) * 18 + q * 72 + 4 * v];
        }
    };
    auto mac16 = [&](const float (&a)[4], const float (&b)[4][4]) {
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[v], b[q][v], acc[q], 0, 0, 0);
    };
    load(a0, b0, 0);
    for (int it = 0; it < iters; it += 2) {
        load(a1, b1, it + 1);
        __builtin_amdgcn_sched_barrier(0);
        mac16(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load(a0, b0, it + 2);
        __builtin_amdgcn_sched_barrier(0);
        mac16(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

__global__ __launch_bounds__(256) void mfma32(float* out, const float* in, int iters)
{
    float a[4], b[2];
    for (int i = 0; i < 4; i++) a[i] = in[threadIdx.x + 256 * i];
    for (int i = 0; i < 2; i++) b[i] = in[threadIdx.x + 256 * (i + 4)];
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int q = 0; q < 2; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v], b[q], acc[q], 0, 0, 0);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    f32x16 s = acc[0] + acc[1];
    float r = 0;
    for (int i = 0; i < 16; i++) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    float *in, *out;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 4096 * 4));
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; i++) h[i] = (float)(rand() % 20001) / 10000.f - 1.f;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc;
        for (int iters : {20000}) {
            float ms = time_ms([&] { mfma16<<<blocks, 256>>>(out, in, iters); }, 3);
            double fl = 2.0 * 16 * 16 * 4 * 16.0 * iters * blocks * 4;
            printf("mfma 16x16x4 f32  blocks/CU %d iters %6d: %8.3f ms %7.1f TFLOP/s\n", bpc, iters, ms, fl / ms * 1e-9);
            ms = time_ms([&] { mfma16_lds<<<blocks, 256>>>(out, in, iters); }, 3);
            printf("mfma 16x16x4 +LDS blocks/CU %d iters %6d: %8.3f ms %7.1f TFLOP/s\n", bpc, iters, ms, fl / ms * 1e-9);
            ms = time_ms([&] { mfma32<<<blocks, 256>>>(out, in, iters); }, 3);
            fl = 2.0 * 32 * 32 * 2 * 8.0 * iters * blocks * 4;
            printf("mfma 32x32x2 f32  blocks/CU %d iters %6d: %8.3f ms %7.1f TFLOP/s\n", bpc, iters, ms, fl / ms * 1e-9);
        }
    }
    return 0;
}
