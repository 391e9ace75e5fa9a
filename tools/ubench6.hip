// Streaming-copy shapes, part 3: which walk / occupancy / prefetch combination moves
// 8 GiB -> 8 GiB fastest?  (The short-FIR kernel streams faster than the plain copies of
// ubench2; this looks for the reason.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// A workgroup moves "tiles" of 256*U elements of V.  CONTIG: the workgroup owns a contiguous
// run of tiles; otherwise tiles are dealt round-robin.  PF: the next tile's loads are issued
// before the current tile is stored.
template <typename V, int U, bool CONTIG, bool PF, bool NT>
__global__ __launch_bounds__(256) void copy_tiles(const V* __restrict__ in, V* __restrict__ out, size_t ntiles)
{
    extern __shared__ char smem[];
    const unsigned t = threadIdx.x;
    size_t first, step, count;
    if (CONTIG) {
        const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
        first = (size_t)blockIdx.x * per; step = 1;
        count = first >= ntiles ? 0 : (first + per > ntiles ? ntiles - first : per);
    } else {
        first = blockIdx.x; step = gridDim.x;
        count = first >= ntiles ? 0 : (ntiles - first + step - 1) / step;
    }
    auto ld = [&](V (&v)[U], size_t tile) {
        const V* p = in + tile * (256 * U);
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&p[t + 256 * u]) : p[t + 256 * u];
    };
    auto st = [&](const V (&v)[U], size_t tile) {
        V* q = out + tile * (256 * U);
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &q[t + 256 * u]); else q[t + 256 * u] = v[u]; }
    };
    if (count == 0) return;
    if (PF) {
        V a[U], b[U];
        ld(a, first);
        for (size_t i = 0; i < count; i++) {
            if (i + 1 < count) ld(b, first + (i + 1) * step);
            st(a, first + i * step);
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = b[u];
        }
    } else {
        V a[U];
        for (size_t i = 0; i < count; i++) { ld(a, first + i * step); st(a, first + i * step); }
    }
    if (smem[0] == 123 && t == 9999) out[0] = in[1];
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

static size_t g_bytes;
template <typename V, int U, bool CONTIG, bool PF, bool NT>
static void run(const char* nm, const void* in, void* out, int bpc, int mult)
{
    auto k = copy_tiles<V, U, CONTIG, PF, NT>;
    const size_t lds = (160 * 1024) / bpc - 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t ntiles = g_bytes / (sizeof(V) * 256 * U);
    const int blocks = 256 * bpc * mult;
    float ms = time_ms([&] { k<<<blocks, 256, lds>>>((const V*)in, (V*)out, ntiles); }, 5);
    printf("%-10s U %2d contig %d pf %d nt %d  blocks/CU %d x%d: %7.3f ms %7.1f GB/s\n", nm, U, (int)CONTIG, (int)PF, (int)NT, bpc, mult, ms,
           2.0 * g_bytes / ms * 1e-6);
}

int main(int argc, char** argv)
{
    g_bytes = (size_t)(argc > 1 ? atol(argv[1]) : 8) << 30;
    char *in, *out; CK(hipMalloc(&in, g_bytes)); CK(hipMalloc(&out, g_bytes));
    CK(hipMemset(in, 1, g_bytes)); CK(hipMemset(out, 0, g_bytes));
    for (int bpc : {1, 2, 3, 4, 8}) {
        for (int mult : {1, 16}) {
            run<float, 16, false, true, true>("f1", in, out, bpc, mult);
            run<float, 16, true, true, true>("f1", in, out, bpc, mult);
            run<f2, 16, false, false, true>("f2", in, out, bpc, mult);
            run<f2, 16, false, true, true>("f2", in, out, bpc, mult);
            run<f2, 16, true, true, true>("f2", in, out, bpc, mult);
            run<f4, 4, false, true, true>("f4", in, out, bpc, mult);
            run<f4, 4, true, true, true>("f4", in, out, bpc, mult);
            run<f4, 8, true, true, true>("f4", in, out, bpc, mult);
            run<f4, 8, true, true, false>("f4", in, out, bpc, mult);
        }
    }
    return 0;
}
