// Do the matrix pipe and the vector pipe run f32 FMAs side by side?  One wave issues
// 16 independent MFMAs (16x16x4 f32) per iteration and NV packed FMAs between them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NV = packed FMAs per MFMA (0..8)
// PRIO 1: the second, third... resident workgroup of a CU runs at a higher wave priority, so
// the waves sharing a SIMD take turns in long bursts instead of alternating per instruction
template <int NV, bool MF, int PRIO = 0>
__global__ __launch_bounds__(256) void mix(float* out, const float* in, int iters)
{
    if (PRIO == 1) {
        const int level = (blockIdx.x / 256) & 3;
        if (level == 1) __builtin_amdgcn_s_setprio(1);
        if (level == 2) __builtin_amdgcn_s_setprio(2);
        if (level == 3) __builtin_amdgcn_s_setprio(3);
    }
    float a[4], b[4];
    for (int i = 0; i < 4; i++) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[threadIdx.x + 256 * (i + 4)]; }
    f32x4 acc[4] = {};
    constexpr int NACC = 16;  // independent packed accumulators
    f32x2 v[NACC];
    for (int i = 0; i < NACC; i++) v[i] = {in[threadIdx.x + i], in[threadIdx.x + i + 1]};
    const f32x2 m = {in[threadIdx.x + 100], in[threadIdx.x + 101]};
    const f32x2 c = {in[threadIdx.x + 200], in[threadIdx.x + 201]};
    for (int it = 0; it < iters; it++) {
        int nv = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (MF) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[q], acc[q], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NV; u++) {
                    f32x2& r = v[nv % NACC];
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(m), "v"(c));
                    nv++;
                }
            }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float t = 0;
    for (int i = 0; i < NACC; i++) t += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + t;
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

template <int NV, bool MF, int PRIO = 0>
static void run(float* out, const float* in, int bpc)
{
    const int blocks = 256 * bpc, iters = 10000;
    float ms = time_ms([&] { mix<NV, MF, PRIO><<<blocks, 256>>>(out, in, iters); }, 3);
    if (PRIO) printf("[prio] ");
    const double waves = (double)blocks * 4;
    const double mf = MF ? 2.0 * 16 * 16 * 4 * 16.0 * iters * waves : 0;
    const double vf = 2.0 * 2 * 64 * NV * 16.0 * iters * waves;
    printf("blocks/CU %d  mfma %d  pk_fma per mfma %d: %8.3f ms  matrix %6.1f + vector %6.1f = %6.1f TFLOP/s\n", bpc, (int)MF, NV, ms,
           mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9);
}

int main()
{
    float *in, *out;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 4096 * 4));
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; i++) h[i] = (float)(rand() % 20001) / 10000.f - 1.f;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    for (int bpc : {1, 2, 4}) {
        run<0, true>(out, in, bpc);
        run<1, true>(out, in, bpc);
        run<2, true>(out, in, bpc);
        run<4, true>(out, in, bpc);
        run<6, true>(out, in, bpc);
        run<8, true>(out, in, bpc);
        run<0, true, 1>(out, in, bpc);
        run<4, false>(out, in, bpc);
        run<8, false>(out, in, bpc);
    }
    return 0;
}
