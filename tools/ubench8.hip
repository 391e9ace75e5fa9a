// Streaming copy, simplest shapes: one element per thread in a grid that covers the array
// (BabelStream's copy), for three element widths and two workgroup sizes, and hipMemcpy
// device-to-device, 8 GiB -> 8 GiB.  Reference points for the ceilings of tools/ubench6.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename V, bool NT>
__global__ void copy1(const V* __restrict__ in, V* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
    else out[i] = in[i];
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

template <typename V, bool NT>
static void run(const char* name, const void* in, void* out, size_t bytes, int bs)
{
    const size_t n = bytes / sizeof(V);
    const float ms = time_ms([&] { hipLaunchKernelGGL((copy1<V, NT>), dim3((unsigned)(n / bs)), dim3(bs), 0, 0, (const V*)in, (V*)out); }, 10);
    printf("%-8s nt %d block %4d: %7.3f ms  %7.1f GB/s\n", name, (int)NT, bs, ms, 2.0 * bytes / ms / 1e6);
}

int main()
{
    const size_t bytes = (size_t)8 << 30;
    void *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    for (int bs : {256, 1024}) {
        run<float, false>("float", in, out, bytes, bs);
        run<f2, false>("float2", in, out, bytes, bs);
        run<f4, false>("float4", in, out, bytes, bs);
        run<f2, true>("float2", in, out, bytes, bs);
        run<f4, true>("float4", in, out, bytes, bs);
    }
    const float ms = time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 5);
    printf("hipMemcpy device-to-device: %7.3f ms  %7.1f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
