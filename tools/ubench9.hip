// A copy with the transform kernel's memory shape: a wave moves one 8 KB frame as 16 loads of
// 512 B (stride 64 elements per lane, as the first pass reads), then 16 stores; 4 frames per
// workgroup, workgroups per CU limited through an LDS request, ITERS frame-blocks per workgroup.
// Tells how much of the gap to the plain copy is the access shape and how much the occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void frame_copy(const f2* __restrict__ in, f2* __restrict__ out, size_t nfb)
{
    extern __shared__ char smem[];
    const unsigned slot = threadIdx.x >> 6, tl = threadIdx.x & 63;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * 4 + slot;
        const f2* src = in + frame * 1024;
        f2* dst = out + frame * 1024;
        f2 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = __builtin_nontemporal_load(&src[tl + 64 * r]);
#pragma unroll
        for (int r = 0; r < 16; r++) __builtin_nontemporal_store(v[r], &dst[tl + 64 * r]);
    }
    if (smem[0] == 123 && tl == 9999) out[0] = in[1];
}

typedef float f4 __attribute__((ext_vector_type(4)));

// the same frames moved as 8 accesses of 16 B per lane (1 KB per wave instruction)
__global__ __launch_bounds__(256) void frame_copy16(const f4* __restrict__ in, f4* __restrict__ out, size_t nfb)
{
    extern __shared__ char smem[];
    const unsigned slot = threadIdx.x >> 6, tl = threadIdx.x & 63;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * 4 + slot;
        const f4* src = in + frame * 512;
        f4* dst = out + frame * 512;
        f4 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = __builtin_nontemporal_load(&src[tl + 64 * r]);
#pragma unroll
        for (int r = 0; r < 8; r++) __builtin_nontemporal_store(v[r], &dst[tl + 64 * r]);
    }
    if (smem[0] == 123 && tl == 9999) out[0] = in[1];
}

int main()
{
    const size_t bytes = (size_t)8 << 30, nfb = bytes / 8192 / 4;
    void *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    CK(hipFuncSetAttribute((const void*)frame_copy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bpc : {2, 3, 4, 8}) {
        const size_t lds = bpc >= 8 ? 0 : (160 * 1024) / bpc - 512;
        for (double iters : {1.0, 2.67, 8.0, 0.0}) {
            size_t grid = iters > 0 ? (size_t)(nfb / iters) : (size_t)256 * bpc;
            if (grid > nfb) grid = nfb;
            auto run = [&] { hipLaunchKernelGGL(frame_copy, dim3((unsigned)grid), dim3(256), lds, 0, (const f2*)in, (f2*)out, nfb); };
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; i++) run();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
            printf("workgroups/CU %d, frame-blocks per workgroup %5s: %7.3f ms  %7.1f GB/s\n", bpc,
                   iters > 0 ? (iters == 1.0 ? "1" : iters == 8.0 ? "8" : "2.67") : "all", ms, 2.0 * bytes / ms / 1e6);
        }
    }
    CK(hipFuncSetAttribute((const void*)frame_copy16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int bpc : {3, 8}) {
        const size_t lds = bpc >= 8 ? 0 : (160 * 1024) / bpc - 512;
        auto run = [&] { hipLaunchKernelGGL(frame_copy16, dim3((unsigned)nfb), dim3(256), lds, 0, (const f4*)in, (f4*)out, nfb); };
        run(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; i++) run();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
        printf("16-byte accesses, workgroups/CU %d, one frame-block per workgroup: %7.3f ms  %7.1f GB/s\n", bpc, ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
