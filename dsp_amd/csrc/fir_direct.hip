// Direct (time-domain) batched FIR / convolution for gfx950.
//
// Two kernels share one staging scheme: a workgroup of 256 lanes produces a
// tile of consecutive outputs of one channel, and the tile's input window (tile
// + tap halo, zero / history filled at the edges) is staged once in LDS.
//
//  * fir_mfma_kernel (f32, the measured path): the convolution of a 16-output
//    block is a 16 x K Toeplitz matrix of taps times a K x 16 matrix whose
//    column n is the input window of block n, so 16 blocks (256 outputs) are
//    one chain of v_mfma_f32_16x16x4_f32 instructions over K = nh + 15 window
//    positions.  That instruction is an exact f32 FMA chain at the full f32
//    vector rate; nh of every nh + 15 products are useful.  Each wave keeps four
//    such tiles (four independent accumulators) in flight, operands come from
//    LDS as one 4-byte read per MFMA, and a tile's 256 results leave as one
//    fully coalesced 16-byte-per-lane store.
//  * fir_direct_kernel (f64 host wrappers): register-blocked vector MACs, each
//    lane keeping R consecutive outputs per group and sliding over the taps in
//    blocks of NHB with scalar-register tap operands.  Products and sums are
//    rounded separately in the reference's order, so results are bit-exact.
//
//   y[c][i] = sum_{j<nh, 0<=i-j<nx} h[j] * x[c][i-j]
//
// restates `convolve` (fir/src/main.c:258-280, fft/src/main.c:180-205) and
// conv0/conv1 (convolution/main.c:15-53) of the reference; see dspmi.h.
#include "common.h"
#include "fir_half.h"

#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

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
// PCM: x and y hold interleaved 16-bit stereo frames (one 32-bit word per frame) instead of
// T: a frame enters as (left + right) / 2 and a result leaves as (int16_t)value in both
// channels, the conversions of alsa/main.c:527-528,533-539.
template <typename T, int R, int G, int NHB, int SEG, bool EXACT, bool DESC, bool PCM = false>
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
    const int *__restrict__ xpcm = reinterpret_cast<const int *>(p.x) + (size_t)chan * p.x_stride;
    const T *__restrict__ hrow = (!PCM && p.hist) ? p.hist + (size_t)chan * p.hist_stride : nullptr;
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
                if (s < nx && !(drop_first && s == 0)) {
                    if constexpr (PCM) {
                        const int f = xpcm[s];  // left in the low half, right in the high half
                        v = ((T)(short)(f & 0xffff) + (T)(short)(f >> 16)) / T(2);
                    } else {
                        v = xrow[s];
                    }
                }
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
    if constexpr (PCM) {
        int *__restrict__ ypcm = reinterpret_cast<int *>(p.y) + (size_t)chan * p.y_stride;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const long o0 = tile_start + g * (256 * R) + t * R;
#pragma unroll
            for (int r = 0; r < R; r++) {
                // (int16_t)value as the reference's x86 build performs it: truncate toward zero
                // to 32 bits (anything outside that range, or NaN, becomes 0x80000000), keep
                // the low 16 bits
                const T a = acc[g][r];
                const unsigned v = (a > T(-2147483649.0) && a < T(2147483648.0)) ? ((unsigned)(int)a & 0xffffu) : 0u;
                if (o0 + r < (long)p.nxx) ypcm[o0 + r] = (int)(v | (v << 16));
            }
        }
        return;
    }
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


// ---------------------------------------------------------------------------
// f32 Toeplitz-MFMA kernel
// ---------------------------------------------------------------------------


constexpr int mfma_tile(int TPW) { return 4 * TPW * 256; }  // outputs per workgroup: 4 waves x TPW tiles x 256
constexpr int kMfmaSeg = 2048;   // taps per LDS staging round

// 16 consecutive floats, then two pads.  One 4-byte LDS read serves 32 lanes: the 16
// columns of an MFMA B operand (16 floats apart) at two adjacent window positions.
// With rows 18 floats apart the columns fall on the 16 even banks and the second
// position on the 16 odd ones, so the read is conflict-free.
constexpr unsigned kRow = 18;
constexpr int kLoadAux = 0;  // default policy: nontemporal loads measured 3 % slower on the 16-tap stream
__device__ __forceinline__ unsigned xpad(unsigned l) { return l + 2 * (l >> 4); }
constexpr unsigned mfma_window_floats(unsigned tile, unsigned seglen) { return (tile + seglen + 16) / 16 * kRow + kRow; }
constexpr unsigned mfma_lds_floats(unsigned tile, unsigned seglen)
{
    // padded input window + reversed tap table with 15 leading and >= 33 trailing zeros
    return mfma_window_floats(tile, seglen) + seglen + 48;
}

// Each workgroup walks `tiles_per_block` consecutive tiles.  A "unit" is one
// (tile, tap segment) pair; while the MFMA loop of unit u runs out of LDS, the
// input window of unit u+1 is already on its way from HBM into registers, and
// is written to LDS once the loop is done (one LDS buffer, two barriers).
// NIT = staging registers per lane = ceil((tile + segment - 1) / 256).
// BLOCKSUM: an f32 chain of n products carries a rounding error that grows like sqrt(n)
// (measured against the double sum of fft/src/main.c:193-204: 1.8e-7 at 127 taps, 7e-7 at
// 2049, 1.5e-6 at 8191).  Long filters therefore sum in blocks: every kBlockGroups * 16 taps
// the chain's accumulators are added to a second set and start again from zero, which holds
// every tap count to the error of a 500-tap chain (about 4e-7).
constexpr int kBlockGroups = 31;
template <int NIT, int TPW, bool BLOCKSUM>
__global__ __launch_bounds__(256) void fir_mfma_kernel(const FirParams<float> p)
{
    constexpr int kMfmaTile = mfma_tile(TPW);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *lds = reinterpret_cast<float *>(smem_raw);

    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63, wave = t >> 6;
    const bool drop_first = (p.flags & DSPMI_FIR_DROP_FIRST) != 0;
    const long nx = p.nx, nh = p.nh;

    const unsigned seg_cap = p.nhp < (unsigned)kMfmaSeg ? p.nhp : (unsigned)kMfmaSeg;
    float *xs = lds;                                              // padded window
    float *hr = lds + mfma_window_floats(kMfmaTile, seg_cap);                // reversed taps

    const int nseg = (p.nhp + kMfmaSeg - 1) / kMfmaSeg;
    const long tile_lo = (long)blockIdx.x * p.tiles_per_block;
    const long tile_hi = min(tile_lo + (long)p.tiles_per_block, (long)p.ntiles_total);
    const long nunits = (tile_hi - tile_lo) * nseg;

    // A operand: lane holds row i = lane&15, window position kk = lane>>4 of each 4-step
    const unsigned a_base = (lane >> 4) + 15 - (lane & 15);
    // B operand: lane holds column n = lane&15 (outputs 16n..16n+15 of the tile), position kk
    const unsigned b_base = wave * (TPW * 256 / 16 * kRow) + kRow * (lane & 15) + (lane >> 4);

    float stage[NIT];

    struct Unit {
        const float *xrow, *hrow;
        long sbase, tile_start;
        unsigned chan;
        int seg0, seglen, nstage;
        bool plain;
    };
    // (channel, tile in channel, segment) of the unit being described, advanced one unit at
    // a time: the walk never divides (a 64-bit division per unit per lane costs hundreds of
    // VALU instructions that would compete with the MFMA stream)
    unsigned cur_chan = (unsigned)(tile_lo / p.tiles_per_chan);
    unsigned cur_tile = (unsigned)(tile_lo - (long)cur_chan * p.tiles_per_chan);
    int cur_seg = 0;
    auto describe = [&]() {
        Unit d;
        const int seg = cur_seg;
        const unsigned chan = cur_chan;
        const long tile_start = (long)cur_tile * kMfmaTile;
        if (++cur_seg == nseg) {
            cur_seg = 0;
            if (++cur_tile == p.tiles_per_chan) {
                cur_tile = 0;
                ++cur_chan;
            }
        }
        d.chan = chan;
        d.tile_start = tile_start;
        d.seg0 = seg * kMfmaSeg;
        d.seglen = min((int)p.nhp - d.seg0, kMfmaSeg);
        d.sbase = tile_start - d.seg0 - d.seglen + 1;
        d.nstage = kMfmaTile + d.seglen - 1;
        d.xrow = p.x + (size_t)chan * p.x_stride;
        d.hrow = p.hist ? p.hist + (size_t)chan * p.hist_stride : nullptr;
        // A buffer load returns 0 for any offset outside [0, nx*4): exactly the zero extension
        // on both sides of the row.  History, the conv1 quirk and windows whose byte offsets do
        // not fit 32 bits (a long row, or a zero tail far past a short one) are staged by the
        // checked loop in commit() instead.
        d.plain = nx <= (1l << 29) && d.sbase + d.nstage <= (1l << 29) && !(d.hrow && d.sbase < 0) &&
                  !(drop_first && d.sbase <= 0);
        return d;
    };

    // global -> registers for one unit (asynchronous until the registers are read)
    auto fetch = [&](const Unit &d) {
        if (d.plain) {
            // the descriptor must be provably wave-uniform or every load is wrapped in a
            // serialising loop: rebuild the row pointer from scalar halves
            const uintptr_t row = reinterpret_cast<uintptr_t>(d.xrow);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)row);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(row >> 32));
            float *base = reinterpret_cast<float *>(((uintptr_t)hi << 32) | lo);
            const int bytes = __builtin_amdgcn_readfirstlane((int)(nx * 4));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
            // the whole byte offset goes in the per-lane operand: only that operand (plus the
            // instruction's immediate) takes part in the range check
            const int sb = __builtin_amdgcn_readfirstlane((int)d.sbase);
            if (sb >= 0) {
                const int voff = (sb + (int)t) * 4;
#pragma unroll
                for (int it = 0; it < NIT; it++)
                    stage[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + it * 1024, 0, kLoadAux));
            } else {
                // left edge: the offset is unsigned to the hardware, so samples before the
                // row are sent to an offset that is out of range for every row length
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int s = sb + (int)t + 256 * it;
                    stage[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, s < 0 ? 0x7ffffff0 : s * 4, 0, kLoadAux));
                }
            }
        }
    };

    // registers -> LDS, plus the segment's reversed tap table when it changes
    auto commit = [&](const Unit &d, bool first) {
        const unsigned l0 = xpad(t);  // xpad(t + 256*it) = xpad(t) + 16*kRow*it
        if (d.plain && d.nstage >= 256 * (NIT - 1)) {
            // every staging round but the last is full
#pragma unroll
            for (int it = 0; it < NIT - 1; it++) xs[l0 + 16 * kRow * it] = stage[it];
            if ((int)(t + 256 * (NIT - 1)) < d.nstage) xs[l0 + 16 * kRow * (NIT - 1)] = stage[NIT - 1];
        } else if (d.plain) {
#pragma unroll
            for (int it = 0; it < NIT; it++)
                if ((int)(t + 256 * it) < d.nstage) xs[l0 + 16 * kRow * it] = stage[it];
        } else {
            for (int l = t; l < d.nstage; l += 256) {
                const long s = d.sbase + l;
                float v = 0.f;
                if (s >= 0) {
                    if (s < nx && !(drop_first && s == 0)) v = d.xrow[s];
                } else if (d.hrow && s >= 1 - nh) {
                    v = d.hrow[nh - 1 + s];
                }
                xs[xpad(l)] = v;
            }
        }
        if (t < kRow) xs[xpad(d.nstage) + t] = 0.f;  // positions only zero taps multiply
        if (first || nseg > 1) {
            // hr[m + 15] = h[seg0 + seglen - 1 - m]; zeros on both sides
            for (int m = t; m < d.seglen + 48; m += 256) {
                const int k = m - 15;
                hr[m] = (k >= 0 && k < d.seglen) ? p.taps[d.seg0 + d.seglen - 1 - k] : 0.f;
            }
        }
    };

    if (nunits <= 0) return;
    Unit next = describe();
    fetch(next);
    commit(next, true);
    __syncthreads();

    f32x4 acc[TPW], tot[BLOCKSUM ? TPW : 1];
    for (long unit = 0; unit < nunits; unit++) {
        const Unit cur = next;
        const bool more = unit + 1 < nunits;
        if (more) {
            next = describe();
            fetch(next);
        }

        const int seglen = cur.seglen;  // multiple of 16
        const bool last_seg = cur.seg0 + cur.seglen >= (int)p.nhp;

        const float *ap = hr + a_base;
        const float *bp = xs + b_base;
        // 16 window positions (4 MFMA steps per tile) per group; the vector ALU shares its
        // pipe with the f32 MFMAs, so every address update costs MFMA issue slots: groups are
        // unrolled three at a time, and a tile's first segment starts its accumulators from
        // the instruction's constant-zero operand instead of clearing registers
        auto group = [&](const float *a_ptr, const float *b_ptr, bool zero) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float a = a_ptr[4 * v];
#pragma unroll
                for (int q = 0; q < TPW; q++) {
                    const float b = b_ptr[q * (256 / 16 * kRow) + 4 * v];
                    const f32x4 c = (zero && v == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[q];
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
                }
            }
        };
        int nu = (seglen + 16) / 16;  // >= 2
        if constexpr (BLOCKSUM) {
            if (cur.seg0 == 0) {
#pragma unroll
                for (int q = 0; q < TPW; q++) tot[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            while (nu > 0) {
                int take = nu < kBlockGroups ? nu : kBlockGroups;
                nu -= take;
                group(ap, bp, true);
                ap += 16;
                bp += kRow;
                take--;
                for (; take >= 3; take -= 3) {
                    group(ap, bp, false);
                    group(ap + 16, bp + kRow, false);
                    group(ap + 32, bp + 2 * kRow, false);
                    ap += 48;
                    bp += 3 * kRow;
                }
                for (; take > 0; take--) {
                    group(ap, bp, false);
                    ap += 16;
                    bp += kRow;
                }
#pragma unroll
                for (int q = 0; q < TPW; q++) tot[q] += acc[q];
            }
        } else {
            if (cur.seg0 == 0) group(ap, bp, true);
            else group(ap, bp, false);
            ap += 16;
            bp += kRow;
            nu--;
            for (; nu >= 3; nu -= 3) {
                group(ap, bp, false);
                group(ap + 16, bp + kRow, false);
                group(ap + 32, bp + 2 * kRow, false);
                ap += 48;
                bp += 3 * kRow;
            }
            for (; nu > 0; nu--) {
                group(ap, bp, false);
                ap += 16;
                bp += kRow;
            }
        }

        if (last_seg) {
            // D layout: column n = lane&15, rows 4*(lane>>4) .. +3 -> 4 consecutive outputs per
            // lane; the wave's 64 lanes cover 1 KiB contiguously.
            const long tile_start = cur.tile_start;
            float *__restrict__ yrow = p.y + (size_t)cur.chan * p.y_stride;
            const bool y_vec = (reinterpret_cast<uintptr_t>(yrow) % 16) == 0;
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                const long o0 = tile_start + wave * (TPW * 256) + q * 256 + 16 * (lane & 15) + 4 * (lane >> 4);
                const f32x4 res = BLOCKSUM ? tot[q] : acc[q];
                if (y_vec && o0 + 4 <= (long)p.nxx) {
                    __builtin_nontemporal_store(res, reinterpret_cast<f32x4 *>(&yrow[o0]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (o0 + r < (long)p.nxx) yrow[o0 + r] = res[r];
                }
            }
        }

        if (more) {
            __syncthreads();  // every wave is done reading this unit's window
            commit(next, false);
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------
// f32 FIR on the bf16 matrix pipe: operands split three ways
// ---------------------------------------------------------------------------
//
// A float is the exact sum of three bf16 values (8 significant bits each: hi = the float rounded
// to bf16, mid = the remainder rounded, lo = what is left).  With samples and taps both split,
// a product is the sum of nine bf16 products, of which the six largest are kept:
//   hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi       (the dropped ones are below 2^-24 of it)
// each exact in the f32 accumulator of v_mfma_f32_16x16x32_bf16.  That instruction retires
// sixteen times the products per cycle of the f32 one, so six of them per 32 window positions
// cost 3/8 of the f32 MFMA chain over the same window.
//
// The tap matrix is the A operand (row b = output within its block of 16, k = window position
// m, entry h[P + b - m]) and stays in registers; the B operand's column a is the window of
// block a, read from LDS as eight consecutive bf16 samples per lane and plane: one 16-byte
// read, conflict-free on unpadded planes (lane groups of that read land on distinct 16-byte
// slots for a 32-byte lane stride).  NS = 32-position slices of the window: NS*32 >= nh + 15.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kSplitMaxSlices = 10;
constexpr int kHalfMaxSlices = 17;  // the f16 kernel alone: its tap images are 8 registers a slice
constexpr int kHalfLdsMaxSlices = 32;  // ... or 2 KB of LDS a slice (fir_half_lds_kernel)
constexpr size_t split_lds_bytes(int ns) { return (size_t)2 * 3 * (kSplitTile + 32 * ns - 16) * 2; }  // two sets of three planes

// bf16(lo) in the low half, bf16(hi) in the high half, both rounded to nearest even
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// two floats -> their three bf16 parts, packed pairwise; every subtraction is exact
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo)
{
    hi = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    r0 -= __builtin_bit_cast(float, mid << 16);
    r1 -= __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = cvt_pk_bf16(r0, r1);
}

// A workgroup walks consecutive tiles of 4096 outputs.  The three planes are double-buffered:
// while the MFMAs of tile i read one set, the same waves split tile i+1 (whose samples arrived
// in registers during tile i-1) into the other set, piece by piece between the MFMA steps, and
// reissue each register's load for tile i+2 as soon as it is free.  Loads are therefore in
// flight for a whole tile time, the split's vector instructions sit in the issue slots the
// 16-cycle MFMAs leave open, and a tile costs one barrier.
//
// Every tile issues the same memory instructions, none under a branch: samples come through a
// buffer descriptor over the part of the row the window covers (a 16-byte access is range-checked
// dword by dword, so whatever lies outside the row reads as zero), the caller's history
// through a second descriptor that is empty except for a row's first tile, results leave through
// a third.  With a fixed count per tile the compiler's wait for a staging register can leave all
// younger operations outstanding; one conditional load or store in the loop and it waits for
// nearly everything, the loads issued moments before included.
template <int NS>
__global__ __launch_bounds__(256) void fir_split_kernel(const FirParams<float> p, const u32x4 *__restrict__ htab)
{
    constexpr int P = 32 * NS - 16;          // window positions ahead of a block's first output
    constexpr int NSTAGE = kSplitTile + P;   // samples staged per tile (a multiple of 16)
    constexpr int NQ = NSTAGE / 4;           // ... in groups of four
    constexpr int NIT = (NQ + 255) / 256;    // groups per lane (the last one on a few lanes only)
    constexpr int NSTEP = 4 * NS;            // MFMA steps per wave and tile
    static_assert(P / 4 <= 256, "the history falls into every lane's first group");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned short *lds = reinterpret_cast<unsigned short *>(smem_raw);  // [2][3][NSTAGE]

    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63, wave = t >> 6;
    const long nx = p.nx, nh = p.nh;

    bf16x8 hf[NS][3];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) hf[s][pl] = __builtin_bit_cast(bf16x8, htab[(s * 3 + pl) * 64 + lane]);

    const long tile_lo = (long)blockIdx.x * p.tiles_per_block;
    const long tile_hi = min(tile_lo + (long)p.tiles_per_block, (long)p.ntiles_total);
    if (tile_hi <= tile_lo) return;
    const long nunits = tile_hi - tile_lo;
    unsigned cur_chan = (unsigned)(tile_lo / p.tiles_per_chan);
    unsigned cur_tile = (unsigned)(tile_lo - (long)cur_chan * p.tiles_per_chan);

    // a descriptor must be provably wave-uniform, or every access through it is wrapped in a
    // serialising loop: rebuild the pointer from scalar halves
    auto descriptor = [](const void *ptr, int bytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
        // (the builtin returns int: without the unsigned casts the low half would sign-extend over the high one)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        void *base = reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo);
        return __builtin_amdgcn_make_buffer_rsrc(base, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };

    struct Unit {
        const float *xbase, *hbase;  // first sample / first history sample the descriptors cover
        int xbytes, hbytes;          // ... and how much (0: nothing to read)
        int xoff, hoff;              // byte offset of the window's first sample from those
        long tile_start;
        unsigned chan;
    };
    long described = 0;
    auto describe = [&]() {
        Unit d;
        const bool valid = described++ < nunits;  // a tile past this workgroup's last reads nothing
        d.chan = valid ? cur_chan : 0;
        d.tile_start = (long)cur_tile * kSplitTile;
        if (++cur_tile == p.tiles_per_chan) {
            cur_tile = 0;
            ++cur_chan;
        }
        const long sbase = d.tile_start - P;          // the window is samples [sbase, sbase + NSTAGE)
        const long lo = sbase < 0 ? 0 : sbase;
        long n = min(nx, sbase + NSTAGE) - lo;
        if (n < 0 || !valid) n = 0;
        d.xbase = p.x + (size_t)d.chan * p.x_stride + lo;
        d.xbytes = (int)n * 4;
        d.xoff = (int)(sbase - lo) * 4;               // <= 0: an offset before the row is out of range
        // history sample j stands for sample j - (nh - 1) of the row
        const bool hist = valid && p.hist && sbase < 0;
        d.hbase = hist ? p.hist + (size_t)d.chan * p.hist_stride : p.x;
        d.hbytes = hist ? (int)(nh - 1) * 4 : 0;
        d.hoff = hist ? (int)(sbase + nh - 1) * 4 : 0;
        return d;
    };

    f32x4 stage[NIT];
    u32x4 hstage;  // the history's share of the first group (zero except in a row's first tile)
    auto load_piece = [&](const Unit &d, int it) {
        // lanes past the window's end repeat its last group (and drop it in put_piece)
        const unsigned u = t + 256 * it;
        const int g = (int)(it < NIT - 1 || u < (unsigned)NQ ? u : (unsigned)NQ - 1) * 16;
        // Offsets before the row: the offset is unsigned to the hardware, so they go to one that is
        // out of range for every length.  A group of samples starts a multiple of four samples from
        // the row's start, so it lies before the row entirely or not at all; the history does not
        // line up with the groups and is read sample by sample.
        const int xo = d.xoff + g;
        stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(descriptor(d.xbase, d.xbytes), xo < 0 ? 0x7ffffff0 : xo, 0, 0));
        if (it == 0) {
            const __amdgpu_buffer_rsrc_t hrsrc = descriptor(d.hbase, d.hbytes);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ho = d.hoff + g + 4 * e;
                hstage[e] = __builtin_amdgcn_raw_buffer_load_b32(hrsrc, ho < 0 ? 0x7ffffff0 : ho, 0, 0);
            }
        }
    };
    auto put_piece = [&](unsigned short *planes, int it) {
        const unsigned u = t + 256 * it;
        f32x4 v = stage[it];
        if (it == 0) v = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, v) | hstage);
        if (it < NIT - 1 || u < (unsigned)NQ) {
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v[0], v[1], h0, m0, l0);
            split_pair(v[2], v[3], h1, m1, l1);
            *reinterpret_cast<u32x2 *>(planes + 4 * u) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2 *>(planes + NSTAGE + 4 * u) = (u32x2){m0, m1};
            *reinterpret_cast<u32x2 *>(planes + 2 * NSTAGE + 4 * u) = (u32x2){l0, l1};
        }
    };

    // B operand: lane holds column a = lane&15 (block a of 16 outputs), window positions
    // 8*(lane>>4) .. +7 of each slice
    const unsigned b_base = wave * 1024 + 16 * (lane & 15) + 8 * (lane >> 4);

    Unit cur = describe();
#pragma unroll
    for (int it = 0; it < NIT; it++) load_piece(cur, it);
    // The tap images must have landed before the loop: left pending, the compiler's wait for
    // them inside the loop would also wait for each iteration's prefetch (the counter is in order).
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) asm volatile("" : "+v"(hf[s][pl]));
#pragma unroll
    for (int it = 0; it < NIT; it++) put_piece(lds, it);
    Unit nxt = describe();
#pragma unroll
    for (int it = 0; it < NIT; it++) load_piece(nxt, it);
    __syncthreads();

    // One tile: its MFMAs out of one set of planes and, step by step, the next tile's split into
    // the other set and the loads of the tile after it.  A step = one slice of one block of 256
    // outputs: three 16-byte reads, six MFMAs (96 cycles); reads run two steps ahead of the MFMAs
    // that use them.  The leading products go to one accumulator and the five corrections to
    // another: the small terms then round against each other, not against the running sum.
    auto iteration = [&](long unit) {
        const unsigned short *rd_planes = lds + (unit & 1) * (3 * NSTAGE);
        unsigned short *wr_planes = lds + ((unit & 1) ^ 1) * (3 * NSTAGE);
        const Unit nxt2 = describe();
        // results leave through a descriptor over this tile's part of the row: the row's ragged
        // end needs no branch, and only 4-byte alignment
        const long left = (long)p.nxx - cur.tile_start;
        const __amdgpu_buffer_rsrc_t yrsrc = descriptor(p.y + (size_t)cur.chan * p.y_stride + cur.tile_start,
                                                        (int)(left < kSplitTile ? left : kSplitTile) * 4);
        bf16x8 xf[3][3];
        f32x4 big[4], small[4];
        const unsigned short *xp = rd_planes + b_base;
        auto rd = [&](int step) {
            const int off = 256 * (step / NS) + 32 * (step % NS);
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
                xf[step % 3][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(xp + pl * NSTAGE + off));
        };
        auto store = [&](int q) {
            // D layout: column a = lane&15, rows 4*(lane>>4) .. +3 -> four consecutive outputs per
            // lane, the wave's 64 lanes cover 1 KiB contiguously
            u32x4 res;
#pragma unroll
            for (int r = 0; r < 4; r++) res[r] = __builtin_bit_cast(unsigned, big[q][r] + small[q][r]);
            const int off = (int)(wave * 1024 + q * 256 + 16 * (lane & 15) + 4 * (lane >> 4)) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(res, yrsrc, off, 0, 2);  // nontemporal
        };
        rd(0);
        if (NSTEP > 1) rd(1);
        int piece = 0;
#pragma unroll
        for (int step = 0; step < NSTEP; step++) {
            const int q = step / NS, s = step % NS, slot = step % 3;
            if (step + 2 < NSTEP) rd(step + 2);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 sm = s ? small[q] : zero, bg = s ? big[q] : zero;
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][0], xf[slot][2], sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][2], xf[slot][0], sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][1], xf[slot][1], sm, 0, 0, 0);
            bg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][0], xf[slot][0], bg, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][0], xf[slot][1], sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[s][1], xf[slot][0], sm, 0, 0, 0);
            small[q] = sm;
            big[q] = bg;
#pragma unroll
            for (; piece < NIT && piece * NSTEP < (step + 1) * NIT; piece++) {
                put_piece(wr_planes, piece);
                load_piece(nxt2, piece);
            }
            __builtin_amdgcn_sched_barrier(0);
            // a block's results leave one step after its last MFMA, when the pipe has drained
            if (s == 0 && q > 0) store(q - 1);
        }
        store(3);
        if (unit + 1 < nunits) __syncthreads();  // the other planes are complete, and every wave has left these
        cur = nxt;
        nxt = nxt2;
    };
    // The first tile runs from a copy of the loop body.  Where two paths meet, the compiler keeps
    // the smaller count of operations that may stay outstanding: entering the loop straight from
    // the prologue (its loads back to back) would make every iteration wait as if it were the first.
    iteration(0);
    for (long unit = 1; unit < nunits; unit++) iteration(unit);
}

// ---------------------------------------------------------------------------
// f32 FIR on the f16 matrix pipe (fir_half.h), results stored
// ---------------------------------------------------------------------------

// Results leave through a descriptor over the tile's part of the row: the row's ragged end
// needs no branch, and only 4-byte alignment.
template <int TILE>
struct StoreSink {
    static constexpr bool kTileHook = false;
    static constexpr bool kUnscaled = false;
    static constexpr bool kRedo = true;
    const FirParams<float> &p;
    __amdgpu_buffer_rsrc_t yrsrc;
    __device__ __forceinline__ explicit StoreSink(const FirParams<float> &params) : p(params) {}
    __device__ __forceinline__ void begin_tile(unsigned chan, long tile_start, float)
    {
        const long left = (long)p.nxx - tile_start;
        yrsrc = uniform_descriptor(p.y + (size_t)chan * p.y_stride + tile_start, (int)(left < TILE ? left : TILE) * 4);
    }
    __device__ __forceinline__ void emit(int q, f32x4 res)
    {
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int off = (int)(wave * (TILE / 4) + q * 256 + 16 * (lane & 15) + 4 * (lane >> 4)) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, res), yrsrc, off, 0, 2);  // nontemporal
    }
    __device__ __forceinline__ void end_tile() {}
    // A tile the walker had no scale for (an infinity or a NaN in its window), summed again tap by
    // tap in f32 in the reference's order (fft/src/main.c:197-203): exactly the outputs whose nh
    // samples include the non-finite one then are not finite, as in the reference.
    __device__ void redo_tile(unsigned chan, long tile_start)
    {
        const long nx = p.nx, nh = p.nh;
        const float *xr = p.x + (size_t)chan * p.x_stride;
        const float *hr = p.hist ? p.hist + (size_t)chan * p.hist_stride + (nh - 1) : nullptr;
        float *yr = p.y + (size_t)chan * p.y_stride;
        const long end = tile_start + TILE < (long)p.nxx ? tile_start + TILE : (long)p.nxx;
        for (long i = tile_start + threadIdx.x; i < end; i += 256) {
            float acc = 0.f;
            for (long j = 0; j < nh; j++) {
                const long k = i - j;
                float xv = 0.f;
                if (k >= 0 && k < nx) xv = xr[k];
                else if (k < 0 && hr) xv = hr[k];
                acc = __builtin_fmaf(p.taps[j], xv, acc);
            }
            yr[i] = acc;
        }
    }
};

// (Four workgroups per CU up to five slices: at five the walker needs three registers fewer than
// the compiler takes unasked, and finds them without spilling.)
template <int NS, int TILE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NS <= 5 ? 4 : NS <= 10 ? 3 : 2))) void fir_half_kernel(const FirParams<float> p, const u32x4 *__restrict__ htab, float tap_unscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    StoreSink<TILE> sink(p);
    fir_half_tiles<NS, StoreSink<TILE>, MemorySource, TILE>(p, htab, tap_unscale, smem_raw, sink, MemorySource{});
}

// 18 .. 32 slices (530 .. 1009 taps): the tap images in LDS, tiles of 1024 outputs (the window of a
// tile is up to 1008 samples longer than the tile, and two workgroups still fit a CU).
template <int NS>
__global__ __launch_bounds__(256) void fir_half_lds_kernel(const FirParams<float> p, const u32x4 *__restrict__ htab, float tap_unscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    StoreSink<1024> sink(p);
    fir_half_tiles<NS, StoreSink<1024>, MemorySource, 1024, true>(p, htab, tap_unscale, smem_raw, sink, MemorySource{});
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
    bool pinned = false;  // DSPMI_FIR_VARIANT named the kernel: no other is substituted
    std::vector<double> taps_host;
    // f32 plans beyond the 16-bit kernels' tap counts also hold an overlap-save plan: the same
    // outputs through FFTs cost fewer flops than 2*nh per sample (doc/notes:81-82) and beat the f32
    // kernel from about 64 taps, the f16 kernel from about 310
    dspmi_ols_t ols = nullptr;
    unsigned ols_n = 0;
    dspmi_ols_t ols_16k = nullptr;   // 16384-sample blocks, one per transform: long filters on rows shorter than that
    dspmi_ols_t ols_small = nullptr; // the shortest transform that holds the taps twice: rows of a few blocks
    unsigned ols_small_n = 0;
    // f32 plans of up to 32 * kSplitMaxSlices - 15 taps: the tap matrix of the bf16 kernel, as
    // [slice][part][lane] register images, and its slice count
    void *split_tab = nullptr;
    int split_ns = 0;
    // ... and of the f16 kernel, with the factor that undoes the scale the taps were split under
    void *half_tab = nullptr;
    bool half_exec = false;  // dspmi_fir_exec_f32 runs the f16 kernel (shorter plans hold its tables for the fused filter + spectrum entry only)
    unsigned half_rows_below = 0;  // overlap-save plans of 306 .. 529 taps: rows shorter than this run the f16 kernel
    float half_unscale = 1.f;
    char name[48] = "";
};

namespace {

enum {
    V_F32_MFMA = 0,   // Toeplitz MFMA 16x16x4, every tap count
    V_F32_SHORT,      // vector FMA, R=4 G=4 NHB=16 (kept for A/B runs: DSPMI_FIR_VARIANT=valu)
    V_F32_LONG,       // vector FMA, R=8 G=2 NHB=16
    V_F64_EXACT,      // R=4 G=2 NHB=8, separate mul/add, ascending taps
};

const char *const kVariantName[] = {"fir_f32_mfma16x16x4", "fir_f32_valu_r4g4_b16", "fir_f32_valu_r8g2_b16",
                                    "fir_f64_exact_r4g2_b8"};

// TPW: 256-output MFMA tiles per wave (4 in use: 4096 outputs per workgroup).
template <int TPW>
int launch_mfma_t(FirParams<float> p, unsigned nchan, int bpc_default, hipStream_t stream)
{
    constexpr int TILE = mfma_tile(TPW);
    const unsigned seg = p.nhp < (unsigned)kMfmaSeg ? p.nhp : (unsigned)kMfmaSeg;
    size_t lds = mfma_lds_floats(TILE, seg) * sizeof(float);
    {
        // Resident workgroups per CU can be capped by inflating the LDS request
        // (DSPMI_FIR_BPC, for experiments); both regimes measured fastest with all eight.
        const char *env = getenv("DSPMI_FIR_BPC");
        const int bpc = env && atoi(env) > 0 ? atoi(env) : bpc_default;
        const size_t want = (160 * 1024) / bpc - 512;
        if (want > lds) lds = want;
    }
    p.tiles_per_chan = (unsigned)(((unsigned long long)p.nxx + TILE - 1) / TILE);
    const unsigned long long tiles = (unsigned long long)p.tiles_per_chan * nchan;
    if (tiles == 0) return DSPMI_OK;
    if (tiles > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu tiles exceed the grid limit", tiles);
    p.ntiles_total = (unsigned)tiles;
    // a few thousand workgroups keep all 256 CUs busy for several rounds; beyond that,
    // longer walks amortise the pipeline fill of each workgroup.  A short filter is an HBM-bound
    // stream, which runs fastest with many short-lived workgroups: two tiles each (16 taps,
    // 512 x 2^20 samples: 5.82 TB/s against 5.50 with eight tiles and two workgroups per CU).
    unsigned tpb = (unsigned)(tiles / 16384);
    p.tiles_per_block = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
    const unsigned short_cap = p.nhp <= 32 ? 2 : 4;  // 33..64 taps: four tiles measured best
    if (p.nhp <= 64 && p.tiles_per_block > short_cap) p.tiles_per_block = short_cap;
    {
        const char *env = getenv("DSPMI_FIR_TPB");  // experiment knob
        if (env && atoi(env) > 0) p.tiles_per_block = (unsigned)atoi(env);
    }
    const unsigned blocks = (unsigned)((tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    auto k_small = fir_mfma_kernel<(TILE + 256 + 255) / 256, TPW, false>;      // segments up to 256 taps
    auto k_large = fir_mfma_kernel<(TILE + kMfmaSeg + 255) / 256, TPW, true>;  // up to kMfmaSeg taps, summed in blocks
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_small), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_large), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL(seg <= 256 ? k_small : k_large, dim3(blocks), dim3(256), lds, stream, p);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

// ---------------------------------------------------------------------------
// up to 16 taps as a plain stream: no tiles, no LDS
// ---------------------------------------------------------------------------

// A filter this short is a copy with a few multiply-adds on the way.  The tile kernels above
// load a tile, then store a tile; the fastest copy on this chip is the simplest one -- one
// 16-byte element per thread, a grid that covers the array.  Here a thread takes four consecutive
// outputs: it loads its own four samples and the NT before them (NT = the tap count rounded up
// to a multiple of four) as NT/4 + 1 aligned 16-byte loads (all but one are what lower-numbered
// lanes of the same wave load as well: cache hits), sums NT products per output with the taps in
// scalar registers, and stores 16 bytes.  The first NT/4 threads of a row (history or zeros before the row) and the last one of a row
// whose length is not a multiple of four go element by element.
// Requires nxx == nx, no flags, rows that start on 16-byte boundaries.  512 x 2^20 samples, 16 taps:
// 0.732 ms against 0.752 for the tile kernel on the same box (5.88 against 5.72 TB/s); eight or
// sixteen outputs per thread are slower (0.806, 1.50 ms: a lane's accesses are then 32 or 64 bytes apart).
// T = double: two outputs per lane, and -- the double path being bit-exact against the reference's
// order of operations -- every product and every sum rounded on its own, taps in ascending order
// (convolve, fft/src/main.c:193-204) or, DESC, descending (conv0's ascending input index,
// convolution/main.c:27-32).
template <typename T, int NT, bool DESC>  // NT: taps, rounded up to a multiple of four
__global__ __launch_bounds__(256) void fir_stream_kernel(const FirParams<T> p, unsigned rows_per_y)
{
    constexpr int V = 16 / sizeof(T);  // outputs per lane: one 16-byte group
    typedef T vec __attribute__((ext_vector_type(V)));
    const size_t row = (size_t)blockIdx.y + (size_t)rows_per_y * blockIdx.z;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;  // which group of the row
    const size_t i0 = (size_t)q * V;
    // the ends of a row belong to fir_stream_edge_kernel: kept in here, the one wave per row that
    // meets them runs a chain of dependent loads and its workgroup outlives twenty others (rows
    // of 16384 samples: 593 Gsamples/s against 730 on rows of 2^20)
    if (i0 < NT || i0 + V > p.nx) return;
    T h[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) h[j] = p.taps[j];  // uniform: scalar loads (zero-padded past nh)
    const vec *src = reinterpret_cast<const vec *>(p.x + row * p.x_stride + i0 - NT);
    T w[NT + V];
#pragma unroll
    for (int k = 0; k < NT / V + 1; k++) {
        const vec v = src[k];
#pragma unroll
        for (int e = 0; e < V; e++) w[V * k + e] = v[e];
    }
    vec acc;
#pragma unroll
    for (int o = 0; o < V; o++) {
        T a = 0;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int j = NT - 1; j >= 0; j--) a = __builtin_fmaf(h[j], w[NT + o - j], a);  // descending tap index, as the tile kernel sums
        } else if constexpr (DESC) {
#pragma unroll
            for (int j = NT - 1; j >= 0; j--) a = a + h[j] * w[NT + o - j];
        } else {
#pragma unroll
            for (int j = 0; j < NT; j++) a = a + h[j] * w[NT + o - j];
        }
        acc[o] = a;
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<vec *>(p.y + row * p.y_stride + i0));
}

// The groups of outputs the kernel above leaves out: the first NT/V of every row (history or
// zeros before the row) and the last one of a row whose length is not a multiple of V.  One
// lane each, element by element, every load unconditional from a clamped index.
template <typename T, int NT, bool DESC>
__global__ __launch_bounds__(64) void fir_stream_edge_kernel(const FirParams<T> p, unsigned nchan)
{
    constexpr int V = 16 / sizeof(T);
    constexpr unsigned PER_ROW = NT / V + 1;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t row = t / PER_ROW;
    const unsigned e = (unsigned)(t - row * PER_ROW);
    if (row >= nchan) return;
    const long nx = p.nx, nh = p.nh;
    const long qlast = (nx - 1) / V;
    long q = e;
    if (e == NT / V) {
        if (!(nx % V) || qlast < NT / V) return;  // the row ends on a whole group, or its last group is one of the first
        q = qlast;
    }
    const long i0 = V * q;
    if (i0 >= nx) return;
    const T *x = p.x + row * p.x_stride;
    const T *hist = (p.hist && nh > 1) ? p.hist + row * p.hist_stride : nullptr;
    T w[NT + V];
#pragma unroll
    for (int k = 0; k < NT + V; k++) {
        const long s = i0 - NT + k, u = nh - 1 + s;  // position in the row, in the history
        const T vx = x[s < 0 ? 0 : (s >= nx ? nx - 1 : s)];
        const T vh = hist ? hist[u < 0 ? 0 : (u > nh - 2 ? nh - 2 : u)] : (T)0;
        w[k] = s >= 0 ? (s < nx ? vx : (T)0) : (s >= 1 - nh ? vh : (T)0);
    }
    T *y = p.y + row * p.y_stride;
#pragma unroll
    for (int o = 0; o < V; o++) {
        T a = 0;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int j = NT - 1; j >= 0; j--) a = __builtin_fmaf(p.taps[j], w[NT + o - j], a);
        } else if constexpr (DESC) {
#pragma unroll
            for (int j = NT - 1; j >= 0; j--) a = a + p.taps[j] * w[NT + o - j];
        } else {
#pragma unroll
            for (int j = 0; j < NT; j++) a = a + p.taps[j] * w[NT + o - j];
        }
        if (i0 + o < nx) y[i0 + o] = a;
    }
}

// Beyond 16 taps the multiply-adds of four outputs per lane cost more than the stream (512 x 2^20
// samples: 20 taps 0.805 ms against 0.735 for the tile kernel, 32 taps 1.06 against 0.735); below,
// the stream kernel is ahead, the more the fewer taps (8 taps: 0.674 against 0.771 ms = 6.4 TB/s).
constexpr unsigned kStreamMaxTaps = 16;
// ... except on rows shorter than the tile kernel's 4096-output tiles, which that kernel serves
// badly: there the stream kernel takes filters of up to 32 taps (2^27 samples as rows of 1024, 21
// taps: 246 -> 488 Gsamples/s; rows of 2048 are level from 25 taps on).
constexpr unsigned kStreamMaxTapsShortRows = 32;

// nullptr-safe test of what the stream kernel needs; the caller falls back to the tile kernel otherwise
template <typename T>
bool stream_applies(const FirParams<T> &p)
{
    constexpr unsigned V = 16 / sizeof(T);
    const char *env = getenv("DSPMI_FIR_STREAM");  // 0: keep the tile kernel (A/B)
    if (env && !atoi(env)) return false;
    // In double every tap costs a multiplication and an addition per output (rounded separately),
    // so the stream kernel's range is shorter: 256 x 2^20 samples, 4 taps 0.673 ms against 0.881
    // for the tile kernel (6.4 against 4.9 TB/s), 8 taps 0.724 / 0.885, 12 taps 0.853 / 0.890, 16
    // taps 1.03 / 0.895; on rows of 1024 it is ahead up to 24 taps, on rows of 256 (2048-output
    // tiles seven eighths empty) three to four times faster for every count up to 32.
    const bool taps_ok = sizeof(T) == 4 ? p.nh <= kStreamMaxTaps || (p.nh <= kStreamMaxTapsShortRows && p.nx < (p.nh <= 24 ? 4096u : 2048u))
                                        : p.nh <= 12 || (p.nh <= 24 && p.nx < 2048u) || (p.nh <= kStreamMaxTapsShortRows && p.nx < 512u);
    return taps_ok && p.nhp >= ((p.nh + 3) & ~3u) && p.flags == 0 && p.nxx == p.nx && p.nx >= 64 && !(p.x_stride & (V - 1)) &&
           !(p.y_stride & (V - 1)) && !((reinterpret_cast<size_t>(p.x) | reinterpret_cast<size_t>(p.y)) & 15);
}

template <typename T, int NT, bool DESC>
int launch_stream_t(const FirParams<T> &p, unsigned nchan, hipStream_t stream)
{
    constexpr unsigned V = 16 / sizeof(T);
    const unsigned groups = (unsigned)(((unsigned long long)p.nx + V - 1) / V);
    const unsigned gx = (groups + 255) / 256;
    const unsigned gy = nchan < 65535u ? nchan : 65535u;
    // rows beyond 65535 go into the grid's third dimension; the remainder is a second launch
    const unsigned gz = nchan / gy;
    hipLaunchKernelGGL((fir_stream_kernel<T, NT, DESC>), dim3(gx, gy, gz), dim3(256), 0, stream, p, gy);
    DSPMI_HIP(hipGetLastError());
    const unsigned done = gy * gz;
    if (done < nchan) {
        FirParams<T> r = p;
        r.x += (size_t)done * p.x_stride;
        r.y += (size_t)done * p.y_stride;
        if (r.hist) r.hist += (size_t)done * p.hist_stride;
        hipLaunchKernelGGL((fir_stream_kernel<T, NT, DESC>), dim3(gx, nchan - done, 1), dim3(256), 0, stream, r, nchan - done);
        DSPMI_HIP(hipGetLastError());
    }
    const unsigned long long edges = (unsigned long long)nchan * (NT / V + 1);
    hipLaunchKernelGGL((fir_stream_edge_kernel<T, NT, DESC>), dim3((unsigned)((edges + 63) / 64)), dim3(64), 0, stream, p, nchan);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T, bool DESC>
int launch_stream(const FirParams<T> &p, unsigned nchan, hipStream_t stream)
{
    static_assert(kStreamMaxTapsShortRows == 32, "one instance per multiple of four taps");
    switch ((p.nh + 3) / 4) {
    case 0: case 1: return launch_stream_t<T, 4, DESC>(p, nchan, stream);
    case 2: return launch_stream_t<T, 8, DESC>(p, nchan, stream);
    case 3: return launch_stream_t<T, 12, DESC>(p, nchan, stream);
    case 4: return launch_stream_t<T, 16, DESC>(p, nchan, stream);
    case 5: return launch_stream_t<T, 20, DESC>(p, nchan, stream);
    case 6: return launch_stream_t<T, 24, DESC>(p, nchan, stream);
    case 7: return launch_stream_t<T, 28, DESC>(p, nchan, stream);
    default: return launch_stream_t<T, 32, DESC>(p, nchan, stream);
    }
}

int launch_mfma(const FirParams<float> &p, unsigned nchan, hipStream_t stream)
{
    // four 256-output tiles per wave (eight measured no faster at 127 taps)
    return launch_mfma_t<4>(p, nchan, 8, stream);
}

// float -> bf16, round to nearest even (finite inputs)
inline unsigned short to_bf16(float v)
{
    unsigned u;
    std::memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline float from_bf16(unsigned short b)
{
    const unsigned u = (unsigned)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

// The A-operand images of fir_split_kernel: lane (b = lane & 15, g = lane >> 4) of slice s holds
// tap-matrix entries [b][32 s + 8 g + j], j = 0..7, entry [b][m] = h[P + b - m] (0 outside the taps).
std::vector<unsigned short> split_tap_images(const float *taps, unsigned nh, int ns)
{
    const int P = 32 * ns - 16;
    std::vector<unsigned short> part[3];
    for (auto &v : part) v.assign(nh, 0);
    for (unsigned k = 0; k < nh; k++) {
        float r = taps[k];
        for (int pl = 0; pl < 3; pl++) {
            part[pl][k] = to_bf16(r);
            r -= from_bf16(part[pl][k]);
        }
    }
    std::vector<unsigned short> img((size_t)ns * 3 * 64 * 8, 0);
    for (int s = 0; s < ns; s++)
        for (int pl = 0; pl < 3; pl++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int m = 32 * s + 8 * (lane >> 4) + j, k = P + (lane & 15) - m;
                    if (k >= 0 && k < (int)nh) img[(((size_t)s * 3 + pl) * 64 + lane) * 8 + j] = part[pl][k];
                }
    return img;
}

// float -> f16 bits, round to nearest even (|v| < 65520)
inline unsigned short to_f16(float v)
{
    unsigned u;
    std::memcpy(&u, &v, 4);
    const unsigned short sign = (unsigned short)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x47800000u) return sign | 0x7c00;
    if (u < 0x38800000u) {  // below 2^-14: a multiple of 2^-24
        float f;
        std::memcpy(&f, &u, 4);
        return sign | (unsigned short)__builtin_rintf(f * 16777216.0f);
    }
    unsigned m = u - 0x38000000u;  // exponent bias 127 -> 15
    m += 0xfffu + ((m >> 13) & 1u);
    return sign | (unsigned short)(m >> 13);
}
inline float from_f16(unsigned short b)
{
    const int e = (b >> 10) & 31, m = b & 1023;
    const float mag = e ? __builtin_ldexpf(1024.f + m, e - 25) : __builtin_ldexpf((float)m, -24);
    return (b & 0x8000) ? -mag : mag;
}

// The A-operand images of fir_half_kernel, laid out like split_tap_images but with two f16 parts
// of the taps times 2^k, k chosen so that the largest tap lies in [2^14, 2^15); *unscale = 2^-k.
std::vector<unsigned short> half_tap_images(const float *taps, unsigned nh, int ns, float *unscale)
{
    const int P = 32 * ns - 16;
    float peak = 0.f;
    for (unsigned k = 0; k < nh; k++) peak = __builtin_fmaxf(peak, __builtin_fabsf(taps[k]));
    int ex = 0;
    if (peak > 0.f && peak < __builtin_inff()) (void)__builtin_frexpf(peak, &ex);  // peak = f * 2^ex, f in [0.5, 1)
    int k2 = peak > 0.f ? 15 - ex : 0;
    k2 = k2 < -100 ? -100 : (k2 > 100 ? 100 : k2);
    *unscale = __builtin_ldexpf(1.f, -k2);
    std::vector<unsigned short> part[2];
    for (auto &v : part) v.assign(nh, 0);
    for (unsigned k = 0; k < nh; k++) {
        float r = __builtin_ldexpf(taps[k], k2);
        for (int pl = 0; pl < 2; pl++) {
            part[pl][k] = to_f16(r);
            r -= from_f16(part[pl][k]);
        }
    }
    std::vector<unsigned short> img((size_t)ns * 2 * 64 * 8, 0);
    for (int s = 0; s < ns; s++)
        for (int pl = 0; pl < 2; pl++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int m = 32 * s + 8 * (lane >> 4) + j, k = P + (lane & 15) - m;
                    if (k >= 0 && k < (int)nh) img[(((size_t)s * 2 + pl) * 64 + lane) * 8 + j] = part[pl][k];
                }
    return img;
}

// Which tile a row length gets.  Measured (2^27 samples as rows of 256 ... 16384, 33 / 127 / 305
// taps): a row costs a fixed amount per tile it touches -- in the ratio 1 : 0.58 : 0.42 for tiles
// of 4096 : 2048 : 1024 outputs, whatever the tap count -- plus the same amount per sample for
// every tile size (a partly filled tile moves only the samples that exist), so the tile with the
// smallest tiles-times-weight wins; ties go to the longer tile.
int half_tile_for_rows(unsigned nxx, int)
{
    const unsigned long long n = nxx;
    const unsigned long long c4 = (n + 4095) / 4096 * 100, c2 = (n + 2047) / 2048 * 58, c1 = (n + 1023) / 1024 * 42;
    if (c4 <= c2 && c4 <= c1) return 4096;
    return c2 <= c1 ? 2048 : 1024;
}

template <int NS, int TILE>
int launch_half_t(const FirParams<float> &p, const void *tab, float unscale, unsigned blocks, hipStream_t stream)
{
    size_t lds = half_lds_bytes(NS, TILE);
    {
        const char *env = getenv("DSPMI_FIR_BPC");  // experiments: cap the resident workgroups per CU
        if (env && atoi(env) > 0) {
            const size_t want = (160 * 1024) / atoi(env) - 512;
            if (want > lds) lds = want;
        }
    }
    auto kern = fir_half_kernel<NS, TILE>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, p, static_cast<const u32x4 *>(tab), unscale);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <int TILE>
int launch_half_tile(FirParams<float> p, const void *tab, float unscale, int ns, unsigned nchan, hipStream_t stream)
{
    p.tiles_per_chan = (unsigned)(((unsigned long long)p.nxx + TILE - 1) / TILE);
    const unsigned long long tiles = (unsigned long long)p.tiles_per_chan * nchan;
    if (tiles == 0) return DSPMI_OK;
    if (tiles > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu tiles exceed the grid limit", tiles);
    p.ntiles_total = (unsigned)tiles;
    // A workgroup's first two tiles are loaded with nothing to overlap them, which asks for long
    // walks; an HBM-bound stream runs fastest as many short-lived workgroups.  Measured on 131072
    // tiles of 4096 outputs (one box): up to four slices (113 taps) 4 tiles per workgroup (33 taps
    // 0.745 ms, 113 taps 0.789; with 8 tiles 0.79, with 16 0.805 whatever the tap count), five
    // slices with four workgroups per CU 8 (0.787 ms; 4 tiles 0.789, 16 tiles 0.800), six 16, and
    // 32 beyond (305 taps: 1.06 ms against 1.07 with 16).  Handing a workgroup tiles a grid apart
    // instead of consecutive ones changes nothing at these settings (16 taps, 16 tiles: 0.805 ->
    // 0.762, level with 4 consecutive tiles' 0.748; 127 taps: 0.861 -> 0.866).  Short tiles walk
    // as many outputs per workgroup as long ones.
    const unsigned per = kSplitTile / TILE;
    unsigned tpb = (unsigned)(tiles / (8192 * per));
    const unsigned cap = ns > 6 ? 32 : ns > 5 ? 16 : ns > 4 ? 8 : 4;
    p.tiles_per_block = (tpb < 2 ? (tiles >= 2048 * per ? 2 : 1) : (tpb > cap ? cap : tpb)) * per;
    {
        const char *env = getenv("DSPMI_FIR_TPB");  // experiment knob
        if (env && atoi(env) > 0) p.tiles_per_block = (unsigned)atoi(env);
    }
    if (p.tiles_per_block > 64) p.tiles_per_block = 64;  // the walker's list of tiles to do over holds 64
    const unsigned blocks = (unsigned)((tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    switch (ns) {
    case 1: return launch_half_t<1, TILE>(p, tab, unscale, blocks, stream);
    case 2: return launch_half_t<2, TILE>(p, tab, unscale, blocks, stream);
    case 3: return launch_half_t<3, TILE>(p, tab, unscale, blocks, stream);
    case 4: return launch_half_t<4, TILE>(p, tab, unscale, blocks, stream);
    case 5: return launch_half_t<5, TILE>(p, tab, unscale, blocks, stream);
    case 6: return launch_half_t<6, TILE>(p, tab, unscale, blocks, stream);
    case 7: return launch_half_t<7, TILE>(p, tab, unscale, blocks, stream);
    case 8: return launch_half_t<8, TILE>(p, tab, unscale, blocks, stream);
    case 9: return launch_half_t<9, TILE>(p, tab, unscale, blocks, stream);
    case 10: return launch_half_t<10, TILE>(p, tab, unscale, blocks, stream);
    case 11: return launch_half_t<11, TILE>(p, tab, unscale, blocks, stream);
    case 12: return launch_half_t<12, TILE>(p, tab, unscale, blocks, stream);
    case 13: return launch_half_t<13, TILE>(p, tab, unscale, blocks, stream);
    case 14: return launch_half_t<14, TILE>(p, tab, unscale, blocks, stream);
    case 15: return launch_half_t<15, TILE>(p, tab, unscale, blocks, stream);
    case 16: return launch_half_t<16, TILE>(p, tab, unscale, blocks, stream);
    case 17: return launch_half_t<17, TILE>(p, tab, unscale, blocks, stream);
    }
    return dspmi::fail(DSPMI_EINVAL, "fir: no f16 kernel of %d slices", ns);
}

// Tiles of 4096 outputs, or of 1024 / 2048 where the rows would leave much of a long tile empty.
int launch_half(const FirParams<float> &p, const void *tab, float unscale, int ns, unsigned nchan, hipStream_t stream)
{
    int tile = kSplitTile;
    if (const char *env = getenv("DSPMI_FIR_TILE")) {  // experiment knob: 1024, 2048 or 4096
        if (atoi(env) == 1024 || atoi(env) == 2048) tile = atoi(env);
    } else {
        tile = half_tile_for_rows(p.nxx, ns);
    }
    if (tile == 1024) return launch_half_tile<1024>(p, tab, unscale, ns, nchan, stream);
    if (tile == 2048) return launch_half_tile<2048>(p, tab, unscale, ns, nchan, stream);
    return launch_half_tile<kSplitTile>(p, tab, unscale, ns, nchan, stream);
}

template <int NS>
int launch_half_lds_t(const FirParams<float> &p, const void *tab, float unscale, unsigned blocks, hipStream_t stream)
{
    const size_t lds = half_lds_bytes(NS, 1024) + half_taps_lds_bytes(NS);
    auto kern = fir_half_lds_kernel<NS>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, p, static_cast<const u32x4 *>(tab), unscale);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

int launch_half_lds(FirParams<float> p, const void *tab, float unscale, int ns, unsigned nchan, hipStream_t stream)
{
    p.tiles_per_chan = (unsigned)(((unsigned long long)p.nxx + 1023) / 1024);
    const unsigned long long tiles = (unsigned long long)p.tiles_per_chan * nchan;
    if (tiles == 0) return DSPMI_OK;
    if (tiles > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu tiles exceed the grid limit", tiles);
    p.ntiles_total = (unsigned)tiles;
    // a workgroup copies the tap images (2 KB a slice) into LDS before its first tile: walks of 4 .. 64 tiles
    const unsigned long long tpb = tiles / 512;
    p.tiles_per_block = (unsigned)(tpb < 4 ? 4 : tpb > 64 ? 64 : tpb);
    const unsigned blocks = (unsigned)((tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    switch (ns) {
#define DSPMI_CASE(N) case N: return launch_half_lds_t<N>(p, tab, unscale, blocks, stream);
        DSPMI_CASE(18) DSPMI_CASE(19) DSPMI_CASE(20) DSPMI_CASE(21) DSPMI_CASE(22) DSPMI_CASE(23) DSPMI_CASE(24) DSPMI_CASE(25)
        DSPMI_CASE(26) DSPMI_CASE(27) DSPMI_CASE(28) DSPMI_CASE(29) DSPMI_CASE(30) DSPMI_CASE(31) DSPMI_CASE(32)
#undef DSPMI_CASE
    }
    return dspmi::fail(DSPMI_EINVAL, "fir: no f16 kernel of %d slices", ns);
}

template <int NS>
int launch_split_t(const FirParams<float> &p, const void *tab, unsigned blocks, hipStream_t stream)
{
    size_t lds = split_lds_bytes(NS);
    {
        const char *env = getenv("DSPMI_FIR_BPC");  // experiments: cap the resident workgroups per CU
        if (env && atoi(env) > 0) {
            const size_t want = (160 * 1024) / atoi(env) - 512;
            if (want > lds) lds = want;
        }
    }
    auto kern = fir_split_kernel<NS>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, p, static_cast<const u32x4 *>(tab));
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

int launch_split(FirParams<float> p, const void *tab, int ns, unsigned nchan, hipStream_t stream)
{
    p.tiles_per_chan = (unsigned)(((unsigned long long)p.nxx + kSplitTile - 1) / kSplitTile);
    const unsigned long long tiles = (unsigned long long)p.tiles_per_chan * nchan;
    if (tiles == 0) return DSPMI_OK;
    if (tiles > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu tiles exceed the grid limit", tiles);
    p.ntiles_total = (unsigned)tiles;
    // every workgroup loads the tap images once (12 registers per slice) and then walks its tiles
    unsigned tpb = (unsigned)(tiles / 8192);
    p.tiles_per_block = tpb < 2 ? (tiles >= 2048 ? 2 : 1) : (tpb > 16 ? 16 : tpb);
    {
        const char *env = getenv("DSPMI_FIR_TPB");  // experiment knob
        if (env && atoi(env) > 0) p.tiles_per_block = (unsigned)atoi(env);
    }
    const unsigned blocks = (unsigned)((tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    switch (ns) {
    case 1: return launch_split_t<1>(p, tab, blocks, stream);
    case 2: return launch_split_t<2>(p, tab, blocks, stream);
    case 3: return launch_split_t<3>(p, tab, blocks, stream);
    case 4: return launch_split_t<4>(p, tab, blocks, stream);
    case 5: return launch_split_t<5>(p, tab, blocks, stream);
    case 6: return launch_split_t<6>(p, tab, blocks, stream);
    case 7: return launch_split_t<7>(p, tab, blocks, stream);
    case 8: return launch_split_t<8>(p, tab, blocks, stream);
    case 9: return launch_split_t<9>(p, tab, blocks, stream);
    case 10: return launch_split_t<10>(p, tab, blocks, stream);
    }
    return dspmi::fail(DSPMI_EINVAL, "fir: no bf16 kernel of %d slices", ns);
}

constexpr int kTapPad = 16;   // taps are padded to a multiple of this (>= every NHB)
// The f16 kernel takes every tap count its ten slices hold, from where the f32 matrix kernel stops
// being as fast (measured, 512 x 2^20 samples on one box: 17 and 24 taps 0.747 ms against 0.737 and
// 0.745 for the f32 kernel; 33 taps 0.745 against 0.805; 63 taps 0.758 against 0.93; 127 taps 0.806
// against 1.03 for overlap-save and 0.98 for the bf16 kernel; 305 taps 1.09 against 1.16 for
// overlap-save).
constexpr unsigned kSplitAutoMinTaps = 33;
constexpr unsigned kOlsLongMinTaps = 900;
constexpr unsigned kOlsMinTaps = 64;  // measured crossover (direct 1.12 ms vs 1.08 at 64 taps; direct wins below)
constexpr int kSeg = 2048;    // taps per LDS staging round

template <typename T, int R, int G, int NHB, bool EXACT, bool DESC, bool PCM = false>
int launch(const FirParams<T> &p, unsigned nchan, hipStream_t stream)
{
    constexpr int TILE = 256 * R * G;
    const unsigned seg = p.nhp < (unsigned)kSeg ? p.nhp : (unsigned)kSeg;
    const size_t lds = lds_elems<R>(TILE + seg + 8) * sizeof(T);
    auto kern = fir_direct_kernel<T, R, G, NHB, kSeg, EXACT, DESC, PCM>;
    static dspmi::PerDeviceOnce attr;  // one per instantiation; the limit covers every tap count
    constexpr size_t lds_max = lds_elems<R>(TILE + kSeg + 8) * sizeof(T);
    if (lds_max > 48 * 1024 && attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr.mark();
    }
    const unsigned long long blocks = (unsigned long long)p.tiles_per_chan * nchan;
    if (blocks == 0) return DSPMI_OK;
    if (blocks > 0x7fffffffull) return dspmi::fail(DSPMI_EINVAL, "fir: %llu workgroups exceed the grid limit", blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, p);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T>
int create(dspmi_fir_t *out, const T *taps, unsigned nh, int device, int precision, unsigned opts = 0)
{
    if (!out || !taps || nh == 0) return dspmi::fail(DSPMI_EINVAL, "fir_create: null argument or nh == 0");
    if (opts & ~(DSPMI_FIR_FULL_RANGE | DSPMI_FIR_F32_PRODUCTS)) return dspmi::fail(DSPMI_EINVAL, "fir_create: unknown option bits %#x", opts);
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
    if (precision == DSPMI_F64) {
        pl->variant = V_F64_EXACT;
    } else {
        // A/B knob: "valu", "mfma" (the f32 matrix kernel), "f16" / "bf16" (the two split kernels,
        // wherever they apply) or "ols" (overlap-save from kOlsMinTaps up) pin that path; anything
        // else is automatic
        // (the creation options select the same paths from the API: DSPMI_FIR_F32_PRODUCTS = "mfma",
        // DSPMI_FIR_FULL_RANGE = "bf16" where the f16 kernel would run)
        const char *env = getenv("DSPMI_FIR_VARIANT");
        if (opts & DSPMI_FIR_F32_PRODUCTS) env = "mfma";
        else if ((opts & DSPMI_FIR_FULL_RANGE) && !env && nh >= kSplitAutoMinTaps && nh <= 32 * kSplitMaxSlices - 15) env = "bf16";
        const bool valu = env && !strcmp(env, "valu"), mfma = env && !strcmp(env, "mfma");
        const bool want_bf16 = env && !strcmp(env, "bf16"), want_ols = env && !strcmp(env, "ols");
        const bool want_half = !want_bf16, want_split = want_bf16 || (env && !strcmp(env, "f16"));
        pl->variant = valu ? (nh <= 32 ? V_F32_SHORT : V_F32_LONG) : V_F32_MFMA;
        pl->pinned = valu || mfma;
        if constexpr (sizeof(T) == 4) {
            const unsigned split_max = 32 * ((env && !strcmp(env, "f16")) ? kHalfLdsMaxSlices : kSplitMaxSlices) - 15;
            // (plans below the f16 kernel's own range hold its tap images too, for the fused filter +
            // spectrum entry point; their FIR calls keep the f32 kernels)
            const bool tables_only = !env && nh < kSplitAutoMinTaps;
            // 306 .. 529 taps (11 .. 17 slices; 8 registers of tap images a slice, 226 in all at 17):
            // the f16 kernel is bound by the matrix pipe there -- 440 Gsamples/s at 306 taps, 395 at
            // 400, 343 at 511, whatever the row length from 4096 samples on -- where overlap-save runs
            // 210-330 on rows of up to 65536 samples and 420-470 on rows of 2^20.  Such plans hold
            // both, and a call takes the f16 kernel for rows below the measured crossover.
            const bool mid_rows = !env && !(opts & DSPMI_FIR_FULL_RANGE) && nh > 32 * kSplitMaxSlices - 15 && nh <= 32 * kHalfLdsMaxSlices - 15;
            if (!valu && !mfma && !want_ols && (nh >= (want_split ? 1u : kSplitAutoMinTaps) || tables_only) && (nh <= split_max || mid_rows)) {
                pl->half_exec = !tables_only && !mid_rows;
                if (mid_rows) {
                    const unsigned slices = (nh + 15 + 31) / 32;
                    pl->half_rows_below = slices <= 13 ? (1u << 19) : slices <= kHalfMaxSlices ? (1u << 17) : slices <= 20 ? 32768u : slices <= 26 ? 16384u : nh < kOlsLongMinTaps ? 12288u : 8192u;
                }
                pl->split_ns = (int)((nh + 15 + 31) / 32);
                const std::vector<unsigned short> img = want_half ? half_tap_images(taps, nh, pl->split_ns, &pl->half_unscale)
                                                                  : split_tap_images(taps, nh, pl->split_ns);
                void **tab = want_half ? &pl->half_tab : &pl->split_tab;
                e = hipSetDevice(device);
                if (e == hipSuccess) e = hipMalloc(tab, img.size() * 2);
                if (e == hipSuccess) e = hipMemcpy(*tab, img.data(), img.size() * 2, hipMemcpyHostToDevice);
                (void)hipSetDevice(prev);
                if (e != hipSuccess) {
                    (void)hipFree(pl->split_tab);
                    (void)hipFree(pl->half_tab);
                    (void)hipFree(pl->taps_dev);
                    delete pl;
                    return dspmi::fail(DSPMI_EIO, "fir_create: %s", hipGetErrorString(e));
                }
            }
            if (!valu && !mfma && !pl->split_tab && !(pl->half_tab && pl->half_exec) && nh >= kOlsMinTaps && nh <= 16384) {
                const char *real = getenv("DSPMI_OLS_REAL");
                const bool real_blocks = !(real && !atoi(real));  // 0: the kernels that run block pairs (A/B)
                // Up to 899 taps: the smallest power of two >= 4*nh, at least 1024 (measured: 127 and
                // 255 taps are fastest at 1024, 511 at 2048).  From 900 taps the kernels that run
                // one real block per transform: 32768-sample blocks for rows of 65536 samples and
                // more (512 x 2^20 samples: 1024 taps 1.31 ms against 1.36 with 4096-point pairs,
                // 2048 taps 1.34 against 1.73), and 16384-sample blocks below that, where the direct
                // kernel's cost per 4096-output tile grows with the tap count whatever the row length
                // while a block costs the same for any filter (512 rows of 4096 samples, 8191 taps:
                // 0.260 ms direct, 0.023 ms as one block each; 900 taps: 0.036 against 0.023).
                unsigned n = 1024;
                while (n < 4 * nh && n < 16384) n <<= 1;
                if (nh > 2048 || (nh >= kOlsLongMinTaps && real_blocks)) n = 32768;
                if ((real_blocks || nh <= 8192) && dspmi_ols_create_f32(&pl->ols, taps, nh, n, device) == DSPMI_OK)
                    pl->ols_n = dspmi_ols_fft_length(pl->ols);
                else pl->ols = nullptr;
                if (pl->ols && n == 32768 && nh <= 8192 && real_blocks &&
                    dspmi_ols_create_f32(&pl->ols_16k, taps, nh, 16384, device) != DSPMI_OK)
                    pl->ols_16k = nullptr;
                // Rows of a few blocks: what counts is how many transforms a row costs, not how
                // much of each block is overlap, and the direct kernel's 4096-output tiles are
                // mostly empty (2^26 samples as rows of 1024, 511 taps: 2.03 ms direct, 0.32 ms with
                // 1024-point transforms).  Plans below 900 taps also hold the shortest transform
                // that fits, and a call takes whichever costs less for its row length.
                unsigned ns = 1024;
                while (ns < 2 * nh) ns <<= 1;
                if (pl->ols && n < 32768 && ns < n && dspmi_ols_create_f32(&pl->ols_small, taps, nh, ns, device) == DSPMI_OK)
                    pl->ols_small_n = ns;
                else pl->ols_small = nullptr;
            }
        }
    }
    if (pl->half_tab && pl->half_exec) snprintf(pl->name, sizeof pl->name, "fir_f32_f16x2_mfma16x16x32_s%d", pl->split_ns);
    else if (pl->split_tab) snprintf(pl->name, sizeof pl->name, "fir_f32_bf16x3_mfma16x16x32_s%d", pl->split_ns);
    else if (pl->ols) snprintf(pl->name, sizeof pl->name, "fir_f32_overlap_save_n%u", pl->ols_n);
    else if (precision == DSPMI_F32 && pl->variant == V_F32_MFMA && !pl->pinned && nh <= kStreamMaxTaps)
        snprintf(pl->name, sizeof pl->name, "fir_f32_stream%u", (nh + 3) & ~3u);  // (calls it does not cover run the matrix kernel)
    else snprintf(pl->name, sizeof pl->name, "%s", kVariantName[pl->variant]);
    *out = pl;
    return DSPMI_OK;
}

template <typename T>
int exec(dspmi_fir_t pl, T *y, size_t y_stride, unsigned nxx, const T *x, size_t x_stride, unsigned nx,
         unsigned nchan, const T *hist, size_t hist_stride, unsigned flags, void *stream, int precision, bool desc)
{
    if (!pl) return dspmi::fail(DSPMI_EINVAL, "fir_exec: null plan");
    if (int rc = dspmi::check_device(pl->device, "fir_exec")) return rc;
    if (pl->precision != precision) return dspmi::fail(DSPMI_EINVAL, "fir_exec: plan precision mismatch");
    if (nchan == 0 || nxx == 0) return DSPMI_OK;
    if (!y || (!x && nx)) return dspmi::fail(DSPMI_EINVAL, "fir_exec: null buffer");
    if (y_stride < nxx || x_stride < nx) return dspmi::fail(DSPMI_EINVAL, "fir_exec: stride smaller than the row length");
    if (hist && hist_stride + 1 < pl->nh) return dspmi::fail(DSPMI_EINVAL, "fir_exec: hist_stride < nh-1");
    if (dspmi::rows_overlap(x, x_stride, nx, y, y_stride, nxx, nchan, sizeof(T)))
        return dspmi::fail(DSPMI_EINVAL, "fir_exec: input and output rows overlap (filtering in place is not supported)");
    FirParams<T> p{};
    p.y = y; p.x = x; p.hist = hist; p.taps = static_cast<const T *>(pl->taps_dev);
    p.y_stride = y_stride; p.x_stride = x_stride; p.hist_stride = hist_stride;
    p.nxx = nxx; p.nx = nx; p.nh = pl->nh; p.nhp = pl->nhp; p.flags = flags;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if constexpr (sizeof(T) == 4) {
        // overlap-save covers the plain case: as many outputs as inputs, rows long enough to
        // fill a few transforms; tails past the input, the conv1 form and short rows stay direct
        auto run_half = [&]() {
            return pl->split_ns <= kHalfMaxSlices ? launch_half(p, pl->half_tab, pl->half_unscale, pl->split_ns, nchan, s)
                                                  : launch_half_lds(p, pl->half_tab, pl->half_unscale, pl->split_ns, nchan, s);
        };
        if (pl->half_rows_below && flags == 0 && nx < pl->half_rows_below) return run_half();
        if (pl->ols && flags == 0 && nxx == nx) {
            const unsigned long long work = (unsigned long long)nx * pl->nh;
            if (nx >= 2 * pl->ols_n && !(pl->ols_small && nx < 8 * pl->ols_n))
                return dspmi_ols_exec_f32(pl->ols, y, y_stride, x, x_stride, nx, nchan, hist, hist_stride, stream);
            // Below that only rows that are mostly steady state: a transform's rounding error is
            // relative to the whole block that goes through it, so the first nh-1 outputs of a row
            // that starts from zero state -- sums over the filter's small outer taps -- come out
            // with a larger error relative to themselves; a row that is little else keeps the
            // direct kernel.
            const bool settled = hist != nullptr || nx >= 2 * pl->nh;
            if (!settled) {
                // (falls through to the direct kernels below)
            } else if (pl->ols_n < 32768) {
                // transforms per row (two blocks ride one) times the transform's size, for the plan's
                // own length and the shortest one
                auto cost = [&](unsigned n) {
                    const unsigned long long blocks = ((unsigned long long)nx + (n - pl->nh)) / (n - pl->nh + 1);
                    return ((blocks + 1) / 2) * n;
                };
                dspmi_ols_t pick = pl->ols_small && cost(pl->ols_small_n) <= cost(pl->ols_n) ? pl->ols_small : pl->ols;
                return dspmi_ols_exec_f32(pick, y, y_stride, x, x_stride, nx, nchan, hist, hist_stride, stream);
            } else if (pl->ols_16k && work >= 400000ull) {
                return dspmi_ols_exec_f32(pl->ols_16k, y, y_stride, x, x_stride, nx, nchan, hist, hist_stride, stream);
            } else if (!pl->ols_16k && pl->nh > 8192 && work >= 4000000ull) {  // beyond 8192 taps only the 32768-sample blocks fit
                return dspmi_ols_exec_f32(pl->ols, y, y_stride, x, x_stride, nx, nchan, hist, hist_stride, stream);
            }
        }
        if (pl->half_tab && (pl->half_exec || pl->half_rows_below) && flags == 0) return run_half();  // (also what overlap-save leaves: tails past the input, rows that start from zero state)
        if (pl->split_tab && flags == 0) return launch_split(p, pl->split_tab, pl->split_ns, nchan, s);
        if (pl->variant == V_F32_MFMA) {
            if (!pl->pinned && stream_applies(p)) return launch_stream<float, false>(p, nchan, s);
            return launch_mfma(p, nchan, s);
        }
        if (pl->variant == V_F32_SHORT) {
            p.tiles_per_chan = (unsigned)(((unsigned long long)nxx + 4095) / 4096);
            return launch<float, 4, 4, 16, false, false>(p, nchan, s);
        }
        p.tiles_per_chan = (unsigned)(((unsigned long long)nxx + 4095) / 4096);
        return launch<float, 8, 2, 16, false, false>(p, nchan, s);
    } else {
        if (stream_applies(p)) return desc ? launch_stream<double, true>(p, nchan, s) : launch_stream<double, false>(p, nchan, s);
        p.tiles_per_chan = (unsigned)(((unsigned long long)nxx + 2047) / 2048);
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

namespace dspmi {

int fir_half_images(dspmi_fir_t plan, const void **images, float *unscale, int *slices, unsigned *nh, int *device,
                    const float **taps)
{
    if (!plan) return fail(DSPMI_EINVAL, "null FIR plan");
    if (!plan->half_tab || plan->split_ns > kSplitMaxSlices)
        return fail(DSPMI_ENOSYS, "the FIR plan runs %s and holds no tap images of the f16 matrix kernel (plans of up to 305 taps do, unless DSPMI_FIR_VARIANT pins another kernel)", plan->name);
    *images = plan->half_tab;
    *unscale = plan->half_unscale;
    *slices = plan->split_ns;
    *nh = plan->nh;
    *device = plan->device;
    *taps = static_cast<const float *>(plan->taps_dev);
    return DSPMI_OK;
}

}  // namespace dspmi

extern "C" {

int dspmi_fir_create_f32(dspmi_fir_t *plan, const float *taps, unsigned nh, int device)
{
    return create<float>(plan, taps, nh, device, DSPMI_F32);
}

int dspmi_fir_create_f32_opts(dspmi_fir_t *plan, const float *taps, unsigned nh, int device, unsigned options)
{
    return create<float>(plan, taps, nh, device, DSPMI_F32, options);
}

int dspmi_fir_create_f64(dspmi_fir_t *plan, const double *taps, unsigned nh, int device)
{
    return create<double>(plan, taps, nh, device, DSPMI_F64);
}

int dspmi_fir_destroy(dspmi_fir_t plan)
{
    if (!plan) return DSPMI_OK;
    if (plan->ols) (void)dspmi_ols_destroy(plan->ols);
    if (plan->ols_16k) (void)dspmi_ols_destroy(plan->ols_16k);
    if (plan->ols_small) (void)dspmi_ols_destroy(plan->ols_small);
    (void)hipFree(plan->split_tab);
    (void)hipFree(plan->half_tab);
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

int dspmi_fir_exec_s16x2(dspmi_fir_t plan, int16_t *out, size_t out_stride, const int16_t *in, size_t in_stride,
                         unsigned nsampl, unsigned nblock, void *stream)
{
    if (!plan) return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: null plan");
    if (int rc = dspmi::check_device(plan->device, "fir_exec_s16x2")) return rc;
    if (plan->precision != DSPMI_F64)
        return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: needs a double-precision plan (results are exact integers)");
    if (nsampl == 0 || nblock == 0) return DSPMI_OK;
    if (!out || !in) return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: null buffer");
    if (out_stride < nsampl || in_stride < nsampl)
        return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: stride smaller than the block length");
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 3)
        return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: buffers must be 4-byte aligned (one stereo frame)");
    {
        // a workgroup reads the nh-1 frames before its tile after another may have overwritten them
        const char *a = reinterpret_cast<const char *>(in), *b = reinterpret_cast<const char *>(out);
        const size_t span_in = ((size_t)(nblock - 1) * in_stride + nsampl) * 4;
        const size_t span_out = ((size_t)(nblock - 1) * out_stride + nsampl) * 4;
        if (a < b + span_out && b < a + span_in) return dspmi::fail(DSPMI_EINVAL, "fir_exec_s16x2: in and out overlap");
    }
    FirParams<double> p{};
    p.y = reinterpret_cast<double *>(out); p.x = reinterpret_cast<const double *>(in); p.hist = nullptr;
    p.taps = static_cast<const double *>(plan->taps_dev);
    p.y_stride = out_stride; p.x_stride = in_stride; p.hist_stride = 0;
    p.nxx = nsampl; p.nx = nsampl; p.nh = plan->nh; p.nhp = plan->nhp; p.flags = 0;
    p.tiles_per_chan = (unsigned)(((unsigned long long)nsampl + 2047) / 2048);
    return launch<double, 4, 2, 8, true, false, true>(p, nblock, static_cast<hipStream_t>(stream));
}

const char *dspmi_fir_variant(dspmi_fir_t plan)
{
    return plan ? plan->name : "";
}

}  // extern "C"
