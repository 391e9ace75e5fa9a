// Issue rate of the vector instructions the FFT butterflies are made of: plain and packed
// adds / multiplies / FMAs, with and without half-swapping operand selects, and register moves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(float* out, const float* in, int iters)       \
    {                                                                                          \
        f32x2 v[16];                                                                           \
        for (int i = 0; i < 16; i++) v[i] = {in[threadIdx.x + i], in[threadIdx.x + i + 1]};   \
        const f32x2 m = {in[threadIdx.x + 100], in[threadIdx.x + 101]};                       \
        const f32x2 c = {in[threadIdx.x + 200], in[threadIdx.x + 201]};                       \
        for (int it = 0; it < iters; it++) {                                                   \
            _Pragma("unroll") for (int u = 0; u < 16; u++) { f32x2& r = v[u]; asm volatile(ASM : "+v"(r) : "v"(m), "v"(c)); } \
        }                                                                                      \
        float t = 0;                                                                           \
        for (int i = 0; i < 16; i++) t += v[i].x + v[i].y;                                     \
        out[blockIdx.x * 256 + threadIdx.x] = t;                                               \
    }

KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL(k_pk_fma_sel, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]")
KERNEL(k_pk_fma_neg, "v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]")
KERNEL(k_pk_add, "v_pk_add_f32 %0, %1, %0")
KERNEL(k_pk_add_sel, "v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1]")
KERNEL(k_pk_add_neg, "v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]")
KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")
KERNEL(k_pk_mul_sel, "v_pk_mul_f32 %0, %1, %0 op_sel:[0,1]")
KERNEL(k_pk_mov, "v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]")
// the same work as two single-lane-pair instructions
#define KERNEL_S(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, const float* in, int iters)       \
    {                                                                                          \
        float v[32];                                                                           \
        for (int i = 0; i < 32; i++) v[i] = in[threadIdx.x + i];                               \
        const float m = in[threadIdx.x + 100], c = in[threadIdx.x + 200];                      \
        for (int it = 0; it < iters; it++) {                                                   \
            _Pragma("unroll") for (int u = 0; u < 32; u++) { float& r = v[u]; asm volatile(ASM : "+v"(r) : "v"(m), "v"(c)); } \
        }                                                                                      \
        float t = 0;                                                                           \
        for (int i = 0; i < 32; i++) t += v[i];                                                \
        out[blockIdx.x * 256 + threadIdx.x] = t;                                               \
    }
// cross-lane moves: the register-only alternatives to an LDS exchange between FFT passes
#define KERNEL_X(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float* out, const float* in, int iters)       \
    {                                                                                          \
        float v[32];                                                                           \
        for (int i = 0; i < 32; i++) v[i] = in[threadIdx.x + i];                               \
        for (int it = 0; it < iters; it++) {                                                   \
            _Pragma("unroll") for (int u = 0; u < 32; u += 2) { asm volatile(ASM : "+v"(v[u]), "+v"(v[u + 1])); } \
        }                                                                                      \
        float t = 0;                                                                           \
        for (int i = 0; i < 32; i++) t += v[i];                                                \
        out[blockIdx.x * 256 + threadIdx.x] = t;                                               \
    }
KERNEL_X(k_permlane32_swap, "v_permlane32_swap_b32 %0, %1")
KERNEL_X(k_permlane16_swap, "v_permlane16_swap_b32 %0, %1")
KERNEL_X(k_dpp_row_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_X(k_dpp_quad_perm, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_S(k_add2, "v_add_f32 %0, %1, %0")
KERNEL_S(k_fma2, "v_fma_f32 %0, %1, %2, %0")
KERNEL_S(k_mov2, "v_mov_b32 %0, %1")

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
    for (int i = 0; i < 4096; i++) h[i] = (float)(rand() % 20001) / 10000.f - 1.f;
    CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    const int iters = 20000, blocks = 256 * 4;
#define RUN(K, NI)                                                                                     \
    {                                                                                                  \
        float ms = time_ms([&] { K<<<blocks, 256>>>(out, in, iters); }, 3);                            \
        double instr = 16.0 * NI * iters * blocks * 4;  /* wave instructions */                        \
        double cyc = ms * 1e-3 * 2.4e9 * 1024 / instr;                                                 \
        printf("%-14s %8.3f ms  %6.2f cycles per wave instruction per SIMD (at 2.4 GHz)\n", #K, ms, cyc); \
    }
    RUN(k_pk_fma, 1) RUN(k_pk_fma_sel, 1) RUN(k_pk_fma_neg, 1) RUN(k_pk_add, 1) RUN(k_pk_add_sel, 1) RUN(k_pk_add_neg, 1)
    RUN(k_pk_mul, 1) RUN(k_pk_mul_sel, 1) RUN(k_pk_mov, 1) RUN(k_add2, 2) RUN(k_fma2, 2) RUN(k_mov2, 2)
    RUN(k_permlane32_swap, 1) RUN(k_permlane16_swap, 1) RUN(k_dpp_row_shr, 1) RUN(k_dpp_quad_perm, 1)
    return 0;
}
