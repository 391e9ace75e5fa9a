// The f32 FIR through the f16 matrix pipe, as a tile walker that hands every block of results to
// a sink: fir_direct.hip stores them (fir_half_kernel), fft_stockham.hip transforms each tile of
// 4096 outputs into a spectrum without the round trip through HBM (fir_spectrum_kernel).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

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
    unsigned ntiles_total, tiles_per_block;  // MFMA kernels: tiles over all channels / per workgroup
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kSplitTile = 4096;  // outputs per workgroup step: 4 waves x 4 blocks of 256

// A buffer descriptor over `bytes` bytes at ptr.  It must be provably wave-uniform, or every
// access through it is wrapped in a serialising loop: the pointer is rebuilt from scalar halves
// (the builtin returns int: without the unsigned casts the low half would sign-extend over the
// high one).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_descriptor(const void *ptr, int bytes)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *base = reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// ---------------------------------------------------------------------------
// f32 FIR on the f16 matrix pipe: operands split two ways, one scale per tile
// ---------------------------------------------------------------------------
//
// The Toeplitz tiling of fir_split_kernel (fir_direct.hip: tap matrix as the A operand in
// registers, windows of 16-output blocks as B operand columns read from LDS planes, tiles of 4096
// outputs per workgroup step, next tile split while this one multiplies) with half-precision
// parts.  A float scaled into [2^14, 2^15) is hi + lo + e
// with hi = the value rounded to f16 (11 significant bits), lo = the remainder rounded, and
// |e| <= 2^-23 of it; f16 reaches 2^-24 in absolute terms, so under one scale per tile -- the
// power of two that brings the window's largest sample into that range -- a sample keeps this
// relative accuracy down to about 105 dB below the tile's peak, and below that the error stays
// 2^-40 of the peak.  Taps are scaled and split the same way once, at plan creation.  Three
// products are kept of the four (hi*hi, hi*lo, lo*hi; lo*lo is 2^-22 of the product), so a
// slice costs three MFMAs instead of the six of the three-way bf16 split, two LDS planes
// instead of three, and results are multiplied by the inverse scales on their way out.
//
// The scale of tile i+2 is found while tile i runs: its samples sit in the staging registers
// (reissued during the first steps of the iteration, so that they have landed at its end),
// every wave reduces its share to a maximum and leaves it in LDS ahead of the iteration's
// barrier; the next iteration reads the four maxima, and splits the tile with that scale.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr size_t half_taps_lds_bytes(int ns) { return (size_t)ns * 2 * 64 * 16; }
constexpr size_t half_lds_bytes(int ns, int tile = kSplitTile) { return (size_t)2 * 2 * (tile + 32 * ns - 16) * 2 + 64; }  // two sets of two planes + the maxima (a multiple of 64)

// f16(lo) in the low half, f16(hi) in the high half, both rounded to nearest even
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// two floats, scaled -> their two f16 parts, packed pairwise; the subtractions are exact
__device__ __forceinline__ void half_pair(float x0, float x1, float scale, unsigned &hi, unsigned &lo)
{
    const float s0 = x0 * scale, s1 = x1 * scale;
    hi = cvt_pk_f16(s0, s1);
    const f16x2 h = __builtin_bit_cast(f16x2, hi);
    lo = cvt_pk_f16(s0 - (float)h[0], s1 - (float)h[1]);
}

// The largest of a wave's values, returned in every lane.  The values are magnitudes of floats as
// bit patterns (sign bit cleared), which order like the floats they stand for: integer maxima
// need none of the canonicalising instructions a float maximum brings with it.
__device__ __forceinline__ unsigned wave_max(unsigned v)
{
#define DSPMI_DPP_MAX(ctrl, row_mask)                                                                        \
    {                                                                                                        \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xf, false);     \
        v = v > o ? v : o;                                                                                   \
    }
    DSPMI_DPP_MAX(0xb1, 0xf);    // quad_perm [1,0,3,2]
    DSPMI_DPP_MAX(0x4e, 0xf);    // quad_perm [2,3,0,1]
    DSPMI_DPP_MAX(0x141, 0xf);   // row_half_mirror
    DSPMI_DPP_MAX(0x140, 0xf);   // row_mirror: every lane of a row of 16 holds the row's maximum
    DSPMI_DPP_MAX(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    DSPMI_DPP_MAX(0x143, 0xc);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's maximum
#undef DSPMI_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Source: where a tile's samples come from.  MemorySource: from the rows at p.x (the walker's
// double-buffered prefetch).  ToneSource<MAXT>: the rows are never in memory -- they are the sum of
// up to MAXT tones per channel (the generator of tonegen/tonegen.c:25-40), and a tile's window is
// computed straight into the staging registers from tables a small kernel prepares per call
// (spectrum.hip, tone_chain_*):
//   lane_tab[chan][j][256][2] cos, sin of 4 t dw for lane t (dw = the tone's increment)
//   tile_tab[chan][tile][j][12] p, q = the tone's value and its quadrature (amplitude included) at
//                             the window's first position, from the exactly advanced running
//                             angle; delta = the tile's exact increment minus dw (the rounded
//                             additions of one binade advance by a fixed multiple of the angle's
//                             unit, which differs from dw by up to half a unit); cos, sin of
//                             1024 increments; a spare slot; cos, sin of 1, 2, 3 increments
//   hard[chan][tile]          1: the window is not one run of fixed increments (an angle crosses
//                             a power of two inside it, a decreasing angle, ...): its samples were
//                             generated into the row at p.x by the stand-alone generator and are
//                             loaded like any other row
// Lane t holds window positions 4 t + 1024 it + e (e < 4): the tone's (p, q) pair is turned to
// 4 t by the lane table, corrected to second order for the drift 4 t delta, stepped by 1024
// positions between pieces, and the three neighbours of a group come from the tile's constants
// of 1, 2, 3 increments: 2.5 double-precision multiply-adds per sample and tone.  The sum is rounded to f32
// once, like the generator's own output.  hist_out receives the last nh-1 samples of every row.
struct MemorySource {
    static constexpr int kTones = 0;
};
template <int MAXT>
struct ToneSource {
    static constexpr int kTones = MAXT;
    const double *lane_tab, *tile_tab;
    const unsigned *hard;
    float *hist_out;
    size_t hist_out_stride;
};

// Sink: what becomes of the results.  begin_tile(channel, first output of the tile, the factor
// that undoes the tile's and the taps' scales) is called by every lane before a tile's first
// block, emit(q, res) with a lane's four consecutive results (already multiplied by that factor
// unless Sink::kUnscaled)
// of block q (outputs tile + 1024 wave + 256 q + 16 (lane & 15) + 4 (lane >> 4) .. +3), and, where
// Sink::kTileHook is set, end_tile() after a workgroup barrier that follows the tile's last emit
// (the hook may use barriers itself).  Memory instructions in a sink must be unconditional: the
// compiler's waits for the staging registers count the operations issued after each load, and
// one load or store under a branch makes it wait for nearly everything (see fir_split_kernel).
// LDS behind half_lds_bytes(NS) belongs to the sink.  A tile whose window holds an infinity or a
// NaN has no scale: its results are emitted as NaN, and where Sink::kRedo is set
// redo_tile(channel, first output) is called for it after the workgroup's last tile.
// TILE: outputs per workgroup step, 1024 BLK with BLK = 4 (the default), 2 or 1 blocks of 256
// outputs per wave: rows much shorter than 4096 samples fill short tiles where they would leave
// most of a long one empty (the sink numbers its outputs by the same TILE).
// TAPS_LDS: the tap images (8 registers a slice) stay in LDS behind the planes and the sink's share
// -- half_taps_lds_bytes(NS) more -- and are read two steps ahead like the window: what filters of
// 18 .. 32 slices (530 .. 1009 taps) need, whose images no register file holds.
template <int NS, class Sink, class Source, int TILE = kSplitTile, bool TAPS_LDS = false>
__device__ __forceinline__ void fir_half_tiles(const FirParams<float> &p, const u32x4 *__restrict__ htab, float tap_unscale,
                                               char *smem_raw, Sink &sink, const Source &src)
{
    constexpr int MAXT = Source::kTones;  // 0: samples from memory
    constexpr int P = 32 * NS - 16;          // window positions ahead of a block's first output
    constexpr int BLK = TILE / 1024;          // blocks of 256 outputs per wave and tile
    static_assert(TILE == 1024 * BLK && (BLK == 1 || BLK == 2 || BLK == 4), "tiles of 1024, 2048 or 4096 outputs");
    static_assert(MAXT == 0 || TILE == kSplitTile, "the tone source fills whole 4096-output tiles");
    constexpr int NSTAGE = TILE + P;          // samples staged per tile (a multiple of 16)
    constexpr int NQ = NSTAGE / 4;           // ... in groups of four
    constexpr int NIT = (NQ + 255) / 256;    // groups per lane (the last one on a few lanes only)
    constexpr int NSTEP = BLK * NS;          // MFMA steps per wave and tile
    static_assert(P / 4 <= 256, "the history falls into every lane's first group");
    unsigned short *lds = reinterpret_cast<unsigned short *>(smem_raw);  // [2][2][NSTAGE]
    unsigned *maxima = reinterpret_cast<unsigned *>(lds + 4 * NSTAGE);    // [2][4]: per tile parity, per wave; magnitudes as bit patterns

    const unsigned t = threadIdx.x;
    const unsigned lane = t & 63, wave = t >> 6;
    const long nx = p.nx, nh = p.nh;

    f16x8 hf[TAPS_LDS ? 1 : NS][2];
    u32x4 *taps_lds = reinterpret_cast<u32x4 *>(smem_raw + half_lds_bytes(NS, TILE));  // TAPS_LDS: [NS][2][64]
    if constexpr (TAPS_LDS) {
        for (unsigned i = t; i < (unsigned)NS * 128; i += 256) taps_lds[i] = htab[i];  // (the prologue's barrier follows)
    } else {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) hf[s][pl] = __builtin_bit_cast(f16x8, htab[(s * 2 + pl) * 64 + lane]);
    }

    const long tile_lo = (long)blockIdx.x * p.tiles_per_block;
    const long tile_hi = min(tile_lo + (long)p.tiles_per_block, (long)p.ntiles_total);
    if (tile_hi <= tile_lo) return;
    const long nunits = tile_hi - tile_lo;
    unsigned cur_chan = (unsigned)(tile_lo / p.tiles_per_chan);
    unsigned cur_tile = (unsigned)(tile_lo - (long)cur_chan * p.tiles_per_chan);

    auto descriptor = [](const void *ptr, int bytes) { return uniform_descriptor(ptr, bytes); };

    struct Unit {
        const float *xbase, *hbase;  // first sample / first history sample the descriptors cover
        int xbytes, hbytes;          // ... and how much (0: nothing to read)
        int xoff, hoff;              // byte offset of the window's first sample from those
        long tile_start;
        unsigned chan;
        // tone source only
        unsigned tile;               // index of the tile in its row
        bool hard;                   // samples from memory (the generator's own output), not computed here
        float *obase;                // where the row's last nh-1 samples go (the row's last tile only)
        int obytes;
    };
    long described = 0;
    auto describe = [&]() {
        Unit d;
        const bool valid = described++ < nunits;  // a tile past this workgroup's last reads nothing
        d.chan = valid ? cur_chan : 0;
        d.tile = valid ? cur_tile : 0;
        d.tile_start = (long)cur_tile * TILE;
        d.hard = true;
        d.obase = nullptr;
        d.obytes = 0;
        if constexpr (MAXT > 0) {
            d.hard = !valid || __builtin_amdgcn_readfirstlane((int)src.hard[(size_t)d.chan * p.tiles_per_chan + d.tile]) != 0;
            const bool last = valid && src.hist_out && cur_tile + 1 == p.tiles_per_chan;
            d.obase = last ? src.hist_out + (size_t)d.chan * src.hist_out_stride : nullptr;
            d.obytes = last ? (int)(nh - 1) * 4 : 0;
        }
        if (++cur_tile == p.tiles_per_chan) {
            cur_tile = 0;
            ++cur_chan;
        }
        const long sbase = d.tile_start - P;          // the window is samples [sbase, sbase + NSTAGE)
        const long lo = sbase < 0 ? 0 : sbase;
        long n = min(nx, sbase + NSTAGE) - lo;
        if (n < 0 || !valid || (MAXT > 0 && !d.hard)) n = 0;
        d.xbase = p.x + (size_t)d.chan * p.x_stride + lo;
        d.xbytes = (int)n * 4;
        d.xoff = (int)(sbase - lo) * 4;               // <= 0: an offset before the row is out of range
        const bool hist = valid && p.hist && sbase < 0;
        d.hbase = hist ? p.hist + (size_t)d.chan * p.hist_stride : p.x;
        d.hbytes = hist ? (int)(nh - 1) * 4 : 0;
        d.hoff = hist ? (int)(sbase + nh - 1) * 4 : 0;
        return d;
    };

    f32x4 stage[NIT];
    u32x4 hstage;  // the history's share of the first group (zero except in a row's first tile)
    auto load_piece = [&](const Unit &d, int it) {
        const unsigned u = t + 256 * it;
        const int g = (int)(it < NIT - 1 || u < (unsigned)NQ ? u : (unsigned)NQ - 1) * 16;
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
    auto put_piece = [&](unsigned short *planes, int it, float scale) {
        const unsigned u = t + 256 * it;
        if (it < NIT - 1 || u < (unsigned)NQ) {
            unsigned h0, l0, h1, l1;
            half_pair(stage[it][0], stage[it][1], scale, h0, l0);
            half_pair(stage[it][2], stage[it][3], scale, h1, l1);
            *reinterpret_cast<u32x2 *>(planes + 4 * u) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2 *>(planes + NSTAGE + 4 * u) = (u32x2){l0, l1};
        }
    };
    // once the staging registers have landed: the history joins the first group, and the wave's
    // largest magnitude goes to LDS for the tile of this parity
    auto publish_max = [&](long tile_index) {
        stage[0] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, stage[0]) | hstage);
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const u32x4 b = __builtin_bit_cast(u32x4, stage[it]) & 0x7fffffffu;
            const unsigned m01 = b[0] > b[1] ? b[0] : b[1], m23 = b[2] > b[3] ? b[2] : b[3];
            const unsigned m4 = m01 > m23 ? m01 : m23;
            m = m > m4 ? m : m4;
        }
        m = wave_max(m);
        if (lane == 0) maxima[(tile_index & 1) * 4 + wave] = m;
    };
    // Tone source: a tile's whole window into the staging registers.  Only the history (first
    // tile of a row) and the rare hard tiles touch memory on the way in.
    auto fill = [&](const Unit &d) {
        if constexpr (MAXT > 0) {
            {
                const __amdgpu_buffer_rsrc_t hrsrc = descriptor(d.hbase, d.hbytes);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int ho = d.hoff + (int)t * 16 + 4 * e;
                    hstage[e] = __builtin_amdgcn_raw_buffer_load_b32(hrsrc, ho < 0 ? 0x7ffffff0 : ho, 0, 0);
                }
            }
            if (d.hard) {
                const __amdgpu_buffer_rsrc_t xrsrc = descriptor(d.xbase, d.xbytes);
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const unsigned u = t + 256 * it;
                    const int g = (int)(it < NIT - 1 || u < (unsigned)NQ ? u : (unsigned)NQ - 1) * 16;
                    const int xo = d.xoff + g;
                    stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo < 0 ? 0x7ffffff0 : xo, 0, 0));
                }
            } else {
                const double *tt = src.tile_tab + ((size_t)d.chan * p.tiles_per_chan + d.tile) * (MAXT * 12);
                const double *lt = src.lane_tab + ((size_t)d.chan * MAXT * 256 + t) * 2;
                const double t4 = (double)(4 * t);
                double tp[MAXT], tq[MAXT];
#pragma unroll
                for (int j = 0; j < MAXT; j++) {
                    const double p0 = tt[12 * j], q0 = tt[12 * j + 1], dl = tt[12 * j + 2];
                    const double cl = lt[512 * j], sl = lt[512 * j + 1];
                    const double pl = fma(q0, sl, p0 * cl), ql = fma(-p0, sl, q0 * cl);  // turned by 4 t dw
                    const double eps = t4 * dl;                                          // ... and by the drift, to second order
                    const double hc = fma(-0.5 * eps, eps, 1.0);
                    tp[j] = fma(eps, ql, pl * hc);
                    tq[j] = fma(-eps, pl, ql * hc);
                }
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    double v[4];
#pragma unroll
                    for (int j = 0; j < MAXT; j++) {
                        v[0] = j ? v[0] + tp[j] : tp[j];
#pragma unroll
                        for (int e = 1; e < 4; e++) {
                            const double ce = tt[12 * j + 4 + 2 * e], se = tt[12 * j + 5 + 2 * e];
                            v[e] = fma(tq[j], se, j ? fma(tp[j], ce, v[e]) : tp[j] * ce);
                        }
                        if (it + 1 < NIT) {
                            const double cg = tt[12 * j + 3], sg = tt[12 * j + 4];
                            const double pn = fma(tq[j], sg, tp[j] * cg);
                            tq[j] = fma(-tp[j], sg, tq[j] * cg);
                            tp[j] = pn;
                        }
                    }
                    // positions before the row's first sample (first tile: the history's, or zeros)
                    // and, in the last piece, past the window hold nothing
                    const unsigned u = t + 256 * it;
                    const bool live = (it > 0 || d.tile_start + 4 * (long)u >= P) && (it < NIT - 1 || u < (unsigned)NQ);
#pragma unroll
                    for (int e = 0; e < 4; e++) stage[it][e] = live ? (float)v[e] : 0.f;
                }
            }
            // the row's last nh-1 samples are the next call's history
            const __amdgpu_buffer_rsrc_t orsrc = descriptor(d.obase, d.obytes);
            const u32x4 tail = __builtin_bit_cast(u32x4, stage[NIT - 1]);
            const int o0 = ((int)(4 * (t + 256 * (NIT - 1))) - NSTAGE) * 4 + d.obytes;
            __builtin_amdgcn_raw_buffer_store_b32(tail[0], orsrc, o0 < 0 ? 0x7ffffff0 : o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(tail[1], orsrc, o0 + 4 < 0 ? 0x7ffffff0 : o0 + 4, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(tail[2], orsrc, o0 + 8 < 0 ? 0x7ffffff0 : o0 + 8, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(tail[3], orsrc, o0 + 12 < 0 ? 0x7ffffff0 : o0 + 12, 0, 0);
        }
    };
    // the tile's scale, a power of two that brings its largest sample into [2^14, 2^15) (kept
    // within 2^-100 .. 2^100), and the factor that takes a sum of scaled products back
    unsigned long long redo = 0;  // tiles of this workgroup (at most 64: the launchers see to it) whose windows held non-finite samples
    auto scales = [&](long tile_index, float &scale, float &unscale) {
        const u32x4 m4 = *reinterpret_cast<const u32x4 *>(maxima + (tile_index & 1) * 4);
        const unsigned m01 = m4[0] > m4[1] ? m4[0] : m4[1], m23 = m4[2] > m4[3] ? m4[2] : m4[3];
        const unsigned m = m01 > m23 ? m01 : m23;
        const int mu = __builtin_amdgcn_readfirstlane((int)m);
        int e = 268 - (int)((mu >> 23) & 0xff);
        e = e < 27 ? 27 : (e > 227 ? 227 : e);
        scale = __builtin_bit_cast(float, e << 23);
        unscale = __builtin_bit_cast(float, (254 - e) << 23) * tap_unscale;
        // A window with an infinity or a NaN in it has no scale, and what the matrix unit makes of
        // it means nothing: every result of the tile leaves as NaN (the factor is one), and the
        // tile is remembered for the sink to do over once the walk is finished.
        if ((unsigned)mu >= 0x7f800000u) {
            unscale = __builtin_bit_cast(float, 0x7fc00000);
            if (tile_index < 64) redo |= 1ull << tile_index;
        }
    };

    // B operand: lane holds column a = lane&15 (block a of 16 outputs), window positions
    // 8*(lane>>4) .. +7 of each slice
    const unsigned b_base = wave * (256 * BLK) + 16 * (lane & 15) + 8 * (lane >> 4);

    // prologue: the first tile is split with nothing to overlap it, the second is on its way
    Unit cur = describe();
    if constexpr (MAXT > 0) {
        fill(cur);
    } else {
#pragma unroll
        for (int it = 0; it < NIT; it++) load_piece(cur, it);
    }
    if constexpr (!TAPS_LDS) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) asm volatile("" : "+v"(hf[s][pl]));  // landed before the loop (see fir_split_kernel)
    }
    publish_max(0);
    __syncthreads();
    float scale_nxt, unscale_cur, unscale_nxt;
    scales(0, scale_nxt, unscale_cur);
#pragma unroll
    for (int it = 0; it < NIT; it++) put_piece(lds, it, scale_nxt);
    Unit nxt = describe();
    if constexpr (MAXT > 0) {
        fill(nxt);
    } else {
#pragma unroll
        for (int it = 0; it < NIT; it++) load_piece(nxt, it);
    }
    publish_max(1);
    __syncthreads();

    // One tile: its MFMAs out of one set of planes; in its first steps the next tile's split into
    // the other set and the loads of the tile after it; at its end that tile's maximum.  A step =
    // one slice of one block of 256 outputs: two 16-byte reads, three MFMAs (48 cycles); reads run
    // two steps ahead.  The leading products go to one accumulator and the two corrections to
    // another: the small terms then round against each other, not against the running sum.
    auto iteration = [&](long unit) {
        const unsigned short *rd_planes = lds + (unit & 1) * (2 * NSTAGE);
        unsigned short *wr_planes = lds + ((unit & 1) ^ 1) * (2 * NSTAGE);
        const Unit nxt2 = describe();
        scales(unit + 1, scale_nxt, unscale_nxt);
        sink.begin_tile(cur.chan, cur.tile_start, unscale_cur);
        f16x8 xf[3][2], af[TAPS_LDS ? 3 : 1][2];
        f32x4 big[BLK], small[BLK];
        const unsigned short *xp = rd_planes + b_base;
        auto rd = [&](int step) {
            const int off = 256 * (step / NS) + 32 * (step % NS);
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                xf[step % 3][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(xp + pl * NSTAGE + off));
            if constexpr (TAPS_LDS) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++) af[step % 3][pl] = __builtin_bit_cast(f16x8, taps_lds[((step % NS) * 2 + pl) * 64 + lane]);
            }
        };
        auto store = [&](int q) {
            f32x4 res;
#pragma unroll
            for (int r = 0; r < 4; r++) res[r] = Sink::kUnscaled ? big[q][r] + small[q][r] : (big[q][r] + small[q][r]) * unscale_cur;
            sink.emit(q, res);
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
            const f16x8 h0 = TAPS_LDS ? af[slot][0] : hf[TAPS_LDS ? 0 : s][0], h1 = TAPS_LDS ? af[slot][1] : hf[TAPS_LDS ? 0 : s][1];
            sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xf[slot][1], sm, 0, 0, 0);
            bg = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xf[slot][0], bg, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xf[slot][0], sm, 0, 0, 0);
            small[q] = sm;
            big[q] = bg;
            // one piece per step from the first step on (all of them at once where a tile has
            // fewer steps than pieces): the earlier a register is reloaded, the surer it has
            // landed when the iteration ends
#pragma unroll
            for (; piece < NIT && (piece <= step || step == NSTEP - 1); piece++) {
                put_piece(wr_planes, piece, scale_nxt);
                if constexpr (MAXT == 0) load_piece(nxt2, piece);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0 && q > 0) store(q - 1);
        }
        store(BLK - 1);
        if constexpr (MAXT > 0) fill(nxt2);
        if (unit + 1 < nunits) publish_max(unit + 2);
        if (Sink::kTileHook || unit + 1 < nunits) __syncthreads();  // the other planes and the maxima are complete, and every wave has left these planes
        if constexpr (Sink::kTileHook) sink.end_tile();
        cur = nxt;
        nxt = nxt2;
        unscale_cur = unscale_nxt;
    };
    iteration(0);  // a copy of the loop body, so that the loop is entered with its own order of memory operations
    for (long unit = 1; unit < nunits; unit++) iteration(unit);

    // Here, with the walk over and its registers free, the rare path costs the walker nothing.
    if constexpr (Sink::kRedo) {
        if (redo) {
            __builtin_amdgcn_s_waitcnt(0);  // the NaNs written above have landed: what follows overwrites them
            __syncthreads();
            for (long unit = 0; unit < nunits && unit < 64; unit++) {
                if (!((redo >> unit) & 1)) continue;
                const long tile = tile_lo + unit;
                const unsigned chan = (unsigned)(tile / p.tiles_per_chan);
                sink.redo_tile(chan, (tile - (long)chan * p.tiles_per_chan) * TILE);
            }
        }
    }
}

}  // namespace
