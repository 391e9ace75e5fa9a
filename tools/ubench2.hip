// Streaming-bandwidth calibration, part 2: cache-policy and shape variants of a
// read-once / write-once stream (what the FFT and short-FIR kernels are).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename V, int U, bool NTL, bool NTS>
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
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(&p[lane + 64 * u]) : p[lane + 64 * u];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NTS) __builtin_nontemporal_store(v[u], &q[lane + 64 * u]); else q[lane + 64 * u] = v[u]; }
    }
}

template <typename V, int U>
__global__ __launch_bounds__(256) void read_frame(const V* __restrict__ in, float* sink, size_t nframes)
{
    const int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0;
    for (size_t f = wave; f < nframes; f += nwaves) {
        const V* p = in + f * (64 * U);
#pragma unroll
        for (int u = 0; u < U; u++) { V v = p[lane + 64 * u]; acc += v.x; }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename V, int U>
__global__ __launch_bounds__(256) void write_frame(V* __restrict__ out, size_t nframes)
{
    const int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    V v; v.x = lane; v.y = 1;
    for (size_t f = wave; f < nframes; f += nwaves) {
        V* q = out + f * (64 * U);
#pragma unroll
        for (int u = 0; u < U; u++) q[lane + 64 * u] = v;
    }
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
    size_t gib = argc > 1 ? atol(argv[1]) : 8;
    size_t bytes = gib << 30;
    char *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    float* sink; CK(hipMalloc(&sink, 4));
    size_t nf = bytes / 8192;
    auto rep = [&](const char* nm, int blocks, float ms, double mult) {
        printf("%-28s grid %6d %8.3f ms %7.1f GB/s\n", nm, blocks, ms, mult * bytes / ms * 1e-6);
    };
    for (int blocks : {1024, 2048, 3072, 4096, 8192, 16384}) {
        rep("copy f2x16", blocks, time_ms([&] { copy_frame<f2, 16, false, false><<<blocks, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
        rep("copy f2x16 ntload", blocks, time_ms([&] { copy_frame<f2, 16, true, false><<<blocks, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
        rep("copy f2x16 ntstore", blocks, time_ms([&] { copy_frame<f2, 16, false, true><<<blocks, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
        rep("copy f2x16 nt both", blocks, time_ms([&] { copy_frame<f2, 16, true, true><<<blocks, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
        rep("copy f4x8 nt both", blocks, time_ms([&] { copy_frame<f4, 8, true, true><<<blocks, 256>>>((f4*)in, (f4*)out, nf); }, 5), 2);
        rep("read f2x16", blocks, time_ms([&] { read_frame<f2, 16><<<blocks, 256>>>((f2*)in, sink, nf); }, 5), 1);
        rep("read f4x8", blocks, time_ms([&] { read_frame<f4, 8><<<blocks, 256>>>((f4*)in, sink, nf); }, 5), 1);
        rep("write f2x16", blocks, time_ms([&] { write_frame<f2, 16><<<blocks, 256>>>((f2*)out, nf); }, 5), 1);
        rep("write f4x8", blocks, time_ms([&] { write_frame<f4, 8><<<blocks, 256>>>((f4*)out, nf); }, 5), 1);
    }
    rep("copy f2x16 oneshot", (int)(nf / 4), time_ms([&] { copy_frame<f2, 16, false, false><<<nf / 4, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
    rep("copy f2x16 nt oneshot", (int)(nf / 4), time_ms([&] { copy_frame<f2, 16, true, true><<<nf / 4, 256>>>((f2*)in, (f2*)out, nf); }, 5), 2);
    // in place: read and write the same buffer (half the footprint)
    rep("inplace f2x16", 4096, time_ms([&] { copy_frame<f2, 16, false, false><<<4096, 256>>>((f2*)in, (f2*)in, nf); }, 5), 2);
    return 0;
}
