// Direct (time-domain) batched FIR / convolution for gfx950.
//
// One workgroup of 256 lanes produces a tile of 256*R*G consecutive outputs of
// one channel.  The tile's input window (tile + tap halo) is staged once in
// LDS; each lane then keeps R consecutive outputs per group in registers and
// slides over the taps in blocks of NHB: per block it reads R+NHB-1 inputs
// from LDS with 16-byte reads and issues R*NHB FMAs whose tap operand is a
// scalar register (taps are wave-uniform, fetched through the scalar cache).
//
//   y[c][i] = sum_{j<nh, 0<=i-j<nx} h[j] * x[c][i-j]
//
// restates `convolve` (fir/src/main.c:258-280, fft/src/main.c:180-205) and
// conv0/conv1 (convolution/main.c:15-53) of the reference; see dspmi.h.
#include "common.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace {

template <typename T>
struct FirParams {
    T *y;
    const T *x;
    const T *hist;
    const T *taps;  // nhp values, zero-padded past nh
    size_t y_stride, x_stride, hist_stride;
    unsigned nxx, nx, nh, nhp;
    unsigned tiles_per_chan;
    unsigned flags;
};

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4))); };

// Logical LDS index -> physical index.  With R > 4 consecutive outputs per
// lane the lanes' 16-byte reads are R floats apart; one 4-element pad per R
// elements makes the lane stride R+4 elements, which spreads any 16 lanes
// over all 16 slots of the 256-byte bank row.
template <int R>
__device__ __forceinline__ unsigned lds_phys(unsigned l)
{
    if constexpr (R > 4) return l + (l / R) * 4;
    return l;
}

template <int R>
constexpr unsigned lds_elems(unsigned logical)
{
    return R > 4 ? logical + (logical / R + 1) * 4 : logical;
}

// Multiply-accumulate.  EXACT keeps the product and the sum as two roundings
// in the reference's order, so the double-precision wrappers reproduce the
// reference's results bit for bit; otherwise one fused rounding.
template <bool EXACT, typename T>
__device__ __forceinline__ T mac(T acc, T h, T v)
{
    if constexpr (EXACT) {
        if constexpr (sizeof(T) == 8) return __dadd_rn(acc, __dmul_rn(h, v));
        else return __fadd_rn(acc, __fmul_rn(h, v));
    } else {
        if constexpr (sizeof(T) == 8) return __builtin_fma(h, v, acc);
        else return __builtin_fmaf(h, v, acc);
    }
}

// SEG: taps handled per LDS staging round (bounds the LDS footprint for any nh).
template <typename T, int R, int G, int NHB, int SEG, bool EXACT, bool DESC>
__global__ __launch_bounds__(256) void fir_direct_kernel(const FirParams<T> p)
{
    constexpr int TILE = 256 * R * G;
    constexpr int W = R + NHB;  // window elements read per tap block (R+NHB-1, rounded up to 4)
    static_assert(R % 4 == 0 && NHB % 4 == 0 && SEG % NHB == 0, "alignment of the 16-byte LDS reads");
    static_assert(R <= 4 || NHB % R == 0 || R % NHB == 0, "pad arithmetic");
    typedef typename Vec4<T>::type V4;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw);

    const unsigned t = threadIdx.x;
    const unsigned chan = blockIdx.x / p.tiles_per_chan;
    const unsigned tile = blockIdx.x - chan * p.tiles_per_chan;
    const long tile_start = (long)tile * TILE;

    const T *__restrict__ xrow = p.x + (size_t)chan * p.x_stride;
    const T *__restrict__ hrow = p.hist ? p.hist + (size_t)chan * p.hist_stride : nullptr;
    const bool drop_first = (p.flags & DSPMI_FIR_DROP_FIRST) != 0;
    const long nx = p.nx, nh = p.nh;

    T acc[G][R];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[g][r] = T(0);

    const int nseg = (p.nhp + SEG - 1) / SEG;
    for (int si = 0; si < nseg; si++) {
        const int seg = DESC ? nseg - 1 - si : si;
        const int seg0 = seg * SEG;
        const int seglen = min((int)p.nhp - seg0, SEG);

        // ---- stage samples [tile_start-seg0-seglen+1, tile_start-seg0+TILE-1] ----
        const long sbase = tile_start - seg0 - seglen + 1;
        const int nstage = TILE + seglen - 1;
        if (si) __syncthreads();
        for (int l = t; l < nstage; l += 256) {
            const long s = sbase + l;
            T v = T(0);
            if (s >= 0) {
                if (s < nx && !(drop_first && s == 0)) v = xrow[s];
            } else if (hrow && s >= 1 - nh) {
                v = hrow[nh - 1 + s];
            }
            xs[lds_phys<R>(l)] = v;
        }
        if (t < 4) xs[lds_phys<R>(nstage + t)] = T(0);  // tail of the last 16-byte read
        __syncthreads();

        // ---- sliding-window FMA over the segment's taps ----
        const int nblk = seglen / NHB;
        for (int bi = 0; bi < nblk; bi++) {
            const int jb = (DESC ? nblk - 1 - bi : bi) * NHB;
            T hb[NHB];
#pragma unroll
            for (int jj = 0; jj < NHB; jj++) hb[jj] = p.taps[seg0 + jb + jj];
#pragma unroll
            for (int g = 0; g < G; g++) {
                // logical index of the oldest sample this block of taps touches
                const unsigned l0 = g * (256 * R) + t * R + (seglen - NHB - jb);
                T w[W];
#pragma unroll
                for (int m = 0; m < W / 4; m++) {
                    V4 q = *reinterpret_cast<const V4 *>(&xs[lds_phys<R>(l0 + 4 * m)]);
                    w[4 * m + 0] = q.x; w[4 * m + 1] = q.y; w[4 * m + 2] = q.z; w[4 * m + 3] = q.w;
                }
#pragma unroll
                for (int q = 0; q < NHB; q++) {
                    const int jj = DESC ? NHB - 1 - q : q;
#pragma unroll
                    for (int r = 0; r < R; r++)
                        acc[g][r] = mac<EXACT>(acc[g][r], hb[jj], w[r - jj + NHB - 1]);
                }
            }
        }
    }

    // ---- store ----
    T *__restrict__ yrow = p.y + (size_t)chan * p.y_stride;
    const bool y_vec = (reinterpret_cast<uintptr_t>(yrow) % (4 * sizeof(T))) == 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long o0 = tile_start + g * (256 * R) + t * R;
        if (y_vec && o0 + R <= (long)p.nxx) {
#pragma unroll
            for (int m = 0; m < R / 4; m++) {
                V4 q;
                q.x = acc[g][4 * m]; q.y = acc[g][4 * m + 1]; q.z = acc[g][4 * m + 2]; q.w = acc[g][4 * m + 3];
                *reinterpret_cast<V4 *>(&yrow[o0 + 4 * m]) = q;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (o0 + r < (long)p.nxx) yrow[o0 + r] = acc[g][r];
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// plan + dispatch
// ---------------------------------------------------------------------------

struct dspmi_fir_plan {
    int device;
    int precision;  // DSPMI_F32 / DSPMI_F64
    unsigned nh, nhp;
    void *taps_dev;
    int variant;  // index into the table below
    std::vector<double> taps_host;
};

namespace {

enum {
    V_F32_SHORT = 0,  // R=4 G=4 NHB=16: HBM-bound regime, lane-contiguous 16-byte traffic
    V_F32_LONG,       // R=8 G=2 NHB=16: FMA-bound regime
    V_F64_EXACT,      // R=4 G=2 NHB=8, separate mul/add, ascending taps
};

const char *const kVariantName[] = {"fir_f32_r4g4_b16", "fir_f32_r8g2_b16", "fir_f64_exact_r4g2_b8"};

constexpr int kTapPad = 16;   // taps are padded to a multiple of this (>= every NHB)
constexpr int kSeg = 2048;    // taps per LDS staging round

template <typename T, int R, int G, int NHB, bool EXACT, bool DESC>
int launch(const FirParams<T> &p, unsigned nchan, hipStream_t stream)
{
    constexpr int TILE = 256 * R * G;
    const unsigned seg = p.nhp < (unsigned)kSeg ? p.nhp : (unsigned)kSeg;
    const size_t lds = lds_elems<R>(TILE + seg + 8) * sizeof(T);
    auto kern = fir_direct_kernel<T, R, G, NHB, kSeg, EXACT, DESC>;
    if (lds > 48 * 1024)
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned long long blocks = (unsigned long long)p.tiles_per_chan * nchan;
    if (blocks == 0) return DSPMI_OK;
    if (blocks > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu workgroups exceed the grid limit", blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T>
int create(dspmi_fir_t *out, const T *taps, unsigned nh, int device, int precision)
{
    if (!out || !taps || nh == 0) return dspmi::fail(DSPMI_EINVAL, "fir_create: null argument or nh == 0");
    if (device < 0) DSPMI_HIP(hipGetDevice(&device));
    int prev = 0;
    DSPMI_HIP(hipGetDevice(&prev));
    DSPMI_HIP(hipSetDevice(device));
    dspmi_fir_plan *pl = new dspmi_fir_plan();
    pl->device = device;
    pl->precision = precision;
    pl->nh = nh;
    pl->nhp = (nh + kTapPad - 1) / kTapPad * kTapPad;
    pl->taps_host.assign(taps, taps + nh);
    std::vector<T> padded(pl->nhp, T(0));
    std::memcpy(padded.data(), taps, nh * sizeof(T));
    hipError_t e = hipMalloc(&pl->taps_dev, pl->nhp * sizeof(T));
    if (e == hipSuccess) e = hipMemcpy(pl->taps_dev, padded.data(), pl->nhp * sizeof(T), hipMemcpyHostToDevice);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        delete pl;
        return dspmi::fail(DSPMI_EIO, "fir_create: %s", hipGetErrorString(e));
    }
    if (precision == DSPMI_F64) pl->variant = V_F64_EXACT;
    else pl->variant = nh <= 32 ? V_F32_SHORT : V_F32_LONG;
    *out = pl;
    return DSPMI_OK;
}

template <typename T>
int exec(dspmi_fir_t pl, T *y, size_t y_stride, unsigned nxx, const T *x, size_t x_stride, unsigned nx,
         unsigned nchan, const T *hist, size_t hist_stride, unsigned flags, void *stream, int precision, bool desc)
{
    if (!pl) return dspmi::fail(DSPMI_EINVAL, "fir_exec: null plan");
    if (pl->precision != precision) return dspmi::fail(DSPMI_EINVAL, "fir_exec: plan precision mismatch");
    if (nchan == 0 || nxx == 0) return DSPMI_OK;
    if (!y || (!x && nx)) return dspmi::fail(DSPMI_EINVAL, "fir_exec: null buffer");
    if (y_stride < nxx || x_stride < nx) return dspmi::fail(DSPMI_EINVAL, "fir_exec: stride smaller than the row length");
    if (hist && hist_stride + 1 < pl->nh) return dspmi::fail(DSPMI_EINVAL, "fir_exec: hist_stride < nh-1");
    FirParams<T> p;
    p.y = y; p.x = x; p.hist = hist; p.taps = static_cast<const T *>(pl->taps_dev);
    p.y_stride = y_stride; p.x_stride = x_stride; p.hist_stride = hist_stride;
    p.nxx = nxx; p.nx = nx; p.nh = pl->nh; p.nhp = pl->nhp; p.flags = flags;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if constexpr (sizeof(T) == 4) {
        if (pl->variant == V_F32_SHORT) {
            p.tiles_per_chan = (nxx + 4095) / 4096;
            return launch<float, 4, 4, 16, false, false>(p, nchan, s);
        }
        p.tiles_per_chan = (nxx + 4095) / 4096;
        return launch<float, 8, 2, 16, false, false>(p, nchan, s);
    } else {
        p.tiles_per_chan = (nxx + 2047) / 2048;
        if (desc) return launch<double, 4, 2, 8, true, true>(p, nchan, s);
        return launch<double, 4, 2, 8, true, false>(p, nchan, s);
    }
}

}  // namespace

// Internal entry used by the conv0 wrapper: double precision, taps visited in
// descending order so each output sums its terms in conv0's order
// (convolution/main.c:27-32: ascending input index).
int dspmi_fir_exec_f64_desc(dspmi_fir_t plan, double *y, size_t y_stride, unsigned nxx, const double *x,
                            size_t x_stride, unsigned nx, unsigned nchan, unsigned flags, void *stream)
{
    return exec<double>(plan, y, y_stride, nxx, x, x_stride, nx, nchan, nullptr, 0, flags, stream, DSPMI_F64, true);
}

extern "C" {

int dspmi_fir_create_f32(dspmi_fir_t *plan, const float *taps, unsigned nh, int device)
{
    return create<float>(plan, taps, nh, device, DSPMI_F32);
}

int dspmi_fir_create_f64(dspmi_fir_t *plan, const double *taps, unsigned nh, int device)
{
    return create<double>(plan, taps, nh, device, DSPMI_F64);
}

int dspmi_fir_destroy(dspmi_fir_t plan)
{
    if (!plan) return DSPMI_OK;
    hipError_t e = hipFree(plan->taps_dev);
    delete plan;
    if (e != hipSuccess) return dspmi::fail(DSPMI_EIO, "fir_destroy: %s", hipGetErrorString(e));
    return DSPMI_OK;
}

int dspmi_fir_exec_f32(dspmi_fir_t plan, float *y, size_t y_stride, unsigned nxx, const float *x, size_t x_stride,
                       unsigned nx, unsigned nchan, const float *hist, size_t hist_stride, unsigned flags,
                       void *stream)
{
    return exec<float>(plan, y, y_stride, nxx, x, x_stride, nx, nchan, hist, hist_stride, flags, stream, DSPMI_F32,
                       false);
}

int dspmi_fir_exec_f64(dspmi_fir_t plan, double *y, size_t y_stride, unsigned nxx, const double *x, size_t x_stride,
                       unsigned nx, unsigned nchan, const double *hist, size_t hist_stride, unsigned flags,
                       void *stream)
{
    return exec<double>(plan, y, y_stride, nxx, x, x_stride, nx, nchan, hist, hist_stride, flags, stream, DSPMI_F64,
                        false);
}

const char *dspmi_fir_variant(dspmi_fir_t plan)
{
    return plan ? kVariantName[plan->variant] : "";
}

}  // extern "C"
