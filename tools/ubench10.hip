// The frame-shaped copy of ubench9 with the loads going straight into LDS (global_load_lds, 16 B
// per lane, 1 KB per wave instruction, no registers in between), the frame then read back from
// LDS and stored 16 B per lane.  DEPTH frames per wave are in flight: the loads of frame f+DEPTH-1
// are issued before frame f is stored.  Tells whether a transform that takes its input this way
// could stream faster than one that loads into registers (5.86-5.89 TB/s in ubench9).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma_frame(const f4* src, char* lds_frame, unsigned lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane + 64 * r),
                                         (__attribute__((address_space(3))) void*)(lds_frame + 1024 * r), 16, 0, 0);
}

template <int DEPTH>
__global__ __launch_bounds__(256) void frame_copy_dma(const f4* __restrict__ in, f4* __restrict__ out, size_t nframes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* mine = smem + (size_t)wave * DEPTH * 8192;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t f = (size_t)blockIdx.x * 4 + wave;
    // prologue: DEPTH-1 frames on their way
    for (int d = 0; d < DEPTH - 1; d++)
        if (f + d * stride < nframes) dma_frame(in + (f + d * stride) * 512, mine + d * 8192, lane);
    int slot = 0;
    bool first = true;
    for (; f < nframes; f += stride) {
        const size_t ahead = f + (DEPTH - 1) * stride;
        if (ahead < nframes) dma_frame(in + ahead * 512, mine + ((slot + DEPTH - 1) % DEPTH) * 8192, lane);
        // wait until this frame's loads have landed: vmcnt counts loads and stores alike, in
        // order, so what may stay outstanding is whatever was issued after them (the stores of the
        // frames before, the loads of the frames ahead)
        const int newer = (ahead < nframes ? 8 * (DEPTH - 1) : 0) + (first ? 0 : 8 * (DEPTH - 1));
        if (newer >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (newer >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (newer >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (newer >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        first = false;
        const f4* fr = reinterpret_cast<const f4*>(mine + slot * 8192);
        f4* dst = out + f * 512;
        f4 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = fr[lane + 64 * r];
#pragma unroll
        for (int r = 0; r < 8; r++) __builtin_nontemporal_store(v[r], &dst[lane + 64 * r]);
        // the stores must not be counted in the next iteration's wait: vmcnt counts loads only on gfx950? (stores use vscnt-less vmcnt)
        slot = (slot + 1) % DEPTH;
    }
}

template <int DEPTH>
void run_all(const void* in, void* out, size_t bytes)
{
    const size_t nframes = bytes / 8192;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)frame_copy_dma<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (size_t)4 * DEPTH * 8192;
    for (int bpc : {1, 2, 3, 4}) {
        if (lds * bpc > 160 * 1024) continue;
        for (double iters : {1.0, 4.0, 0.0}) {
            size_t grid = iters > 0 ? (size_t)(nframes / 4 / iters) : (size_t)256 * bpc;
            // limit residency through the LDS request
            const size_t req = bpc == 4 && lds * 4 <= 160 * 1024 ? lds : (160 * 1024) / bpc - 512 > lds ? (160 * 1024) / bpc - 512 : lds;
            auto run = [&] { hipLaunchKernelGGL(frame_copy_dma<DEPTH>, dim3((unsigned)grid), dim3(256), req, 0, (const f4*)in, (f4*)out, nframes); };
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; i++) run();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
            printf("depth %d, workgroups/CU <= %d, frame-blocks per workgroup %4s: %7.3f ms  %7.1f GB/s\n", DEPTH, bpc,
                   iters == 1.0 ? "1" : iters == 4.0 ? "4" : "all", ms, 2.0 * bytes / ms / 1e6);
        }
    }
}

int main()
{
    const size_t bytes = (size_t)8 << 30;
    void *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    run_all<1>(in, out, bytes);
    run_all<2>(in, out, bytes);
    run_all<3>(in, out, bytes);
    // check
    unsigned char probe[64];
    CK(hipMemcpy(probe, (char*)out + bytes - 64, 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; i++) if (probe[i] != 1) { printf("COPY WRONG at tail byte %d\n", i); return 1; }
    printf("copy verified\n");
    return 0;
}
