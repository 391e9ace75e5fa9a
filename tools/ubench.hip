// Micro-benchmarks that calibrate the two roofs this library is measured
// against on the box it runs on: streaming-copy bandwidth (8 vs 16 bytes per
// lane, grid-stride vs one-shot) and f32 FMA issue rate (scalar vs packed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

template <typename V>
__global__ __launch_bounds__(256) void copy_k(const V* __restrict__ in, V* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// each thread moves U vectors per iteration (more loads in flight per lane)
template <typename V, int U>
__global__ __launch_bounds__(256) void copy_u(const V* __restrict__ in, V* __restrict__ out, size_t n)
{
    size_t base = ((size_t)blockIdx.x * U) * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (; base < n; base += stride) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = in[base + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; u++) out[base + (size_t)u * blockDim.x] = v[u];
    }
}

// "frame per wave" shape: one wave moves 8 KiB frames, 16 loads of 8 B per lane
// (the access shape of a radix-16 first pass) or 8 loads of 16 B per lane.
template <typename V, int U>
__global__ __launch_bounds__(256) void copy_frame(const V* __restrict__ in, V* __restrict__ out, size_t nframes)
{
    const int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t f = wave; f < nframes; f += nwaves) {
        const V* p = in + f * (64 * U);
        V* q = out + f * (64 * U);
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[lane + 64 * u];
#pragma unroll
        for (int u = 0; u < U; u++) q[lane + 64 * u] = v[u];
    }
}

__global__ __launch_bounds__(256) void fma_k(float* out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pkfma_k(float* out, int iters, float a, float b)
{
    f2 x[8];
    f2 av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
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

int main(int argc, char** argv)
{
    size_t gib = argc > 1 ? atol(argv[1]) : 4;
    size_t bytes = gib << 30;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz mem %zu GiB\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem >> 30);
    char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    auto report = [&](const char* name, float ms) {
        printf("%-34s %8.3f ms  %7.1f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms * 1e-6);
    };
    for (int blocks : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy float2 gridstride %d", blocks);
        report(nm, time_ms([&] { copy_k<float2><<<blocks, 256>>>((float2*)in, (float2*)out, bytes / 8); }, 5));
        snprintf(nm, sizeof nm, "copy float4 gridstride %d", blocks);
        report(nm, time_ms([&] { copy_k<float4><<<blocks, 256>>>((float4*)in, (float4*)out, bytes / 16); }, 5));
        snprintf(nm, sizeof nm, "copy float4 x4 %d", blocks);
        report(nm, time_ms([&] { copy_u<float4, 4><<<blocks, 256>>>((float4*)in, (float4*)out, bytes / 16); }, 5));
        snprintf(nm, sizeof nm, "copy float2 x8 %d", blocks);
        report(nm, time_ms([&] { copy_u<float2, 8><<<blocks, 256>>>((float2*)in, (float2*)out, bytes / 8); }, 5));
        snprintf(nm, sizeof nm, "frame float2 x16 %d", blocks);
        report(nm, time_ms([&] { copy_frame<float2, 16><<<blocks, 256>>>((float2*)in, (float2*)out, bytes / 8192); }, 5));
        snprintf(nm, sizeof nm, "frame float4 x8 %d", blocks);
        report(nm, time_ms([&] { copy_frame<float4, 8><<<blocks, 256>>>((float4*)in, (float4*)out, bytes / 8192); }, 5));
    }
    {   // one block per 4 frames, no loop
        size_t nblk = bytes / 8192 / 4;
        report("frame float2 x16 oneshot", time_ms([&] { copy_frame<float2, 16><<<nblk, 256>>>((float2*)in, (float2*)out, bytes / 8192); }, 5));
        report("frame float4 x8 oneshot", time_ms([&] { copy_frame<float4, 8><<<nblk, 256>>>((float4*)in, (float4*)out, bytes / 8192); }, 5));
        report("hipMemcpyAsync d2d", time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 5));
    }
    float* o; CK(hipMalloc(&o, 256 * 8192 * 4));
    for (int wpb : {256, 512, 1024}) {
        int blocks = 256 * 8; int iters = 20000;
        float ms = time_ms([&] { fma_k<<<blocks, wpb>>>(o, iters, 1.0001f, 0.5f); }, 3);
        double fl = 2.0 * 16 * iters * (double)blocks * wpb;
        printf("v_fma_f32    block %4d: %8.3f ms %8.1f TFLOP/s\n", wpb, ms, fl / ms * 1e-9);
        ms = time_ms([&] { pkfma_k<<<blocks, wpb>>>(o, iters, 1.0001f, 0.5f); }, 3);
        printf("v_pk_fma_f32 block %4d: %8.3f ms %8.1f TFLOP/s\n", wpb, ms, fl / ms * 1e-9);
    }
    return 0;
}
