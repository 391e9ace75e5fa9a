// Batched complex FFT for gfx950: Stockham autosort passes, one frame per
// group of NT lanes, intermediate passes exchanged through a padded LDS image.
//
// A frame of N points is owned by NT = N/P lanes (P points per lane held in
// registers).  Each pass p of radix R:
//
//   j  = butterfly index, lane-contiguous (lane + NT*b)
//   v[r]  = src[j + r*N/R]                       (global for the first pass)
//   v[r] *= W_{Ns*R}^{(j mod Ns)*r}               (Ns = product of earlier radices)
//   v     = DFT_R(v)                              (in registers)
//   dst[(j div Ns)*Ns*R + (j mod Ns) + r*Ns] = v[r]   (global for the last pass)
//
// Input and output are in natural order, so no bit-reversal pass exists.  The
// twiddles come from a table built in double precision on the host; the f32
// kernels copy the entries their passes need into a per-workgroup LDS table
// (registers stay free for the butterflies; long frames keep two short tables
// there and multiply a twiddle together from them), the f64 kernels read the
// global table.  When NT <= 64 a frame lives inside one wave and the passes need
// no workgroup barrier.  Besides the plain c2c kernel the file holds the fused
// real-input kernel (split and spectrum epilogues), the fused real-output kernel
// and the fused overlap-save kernel (forward transform, multiply, backward
// transform on chip).
// It is compiled without SLP vectorisation: packed f32 instructions issue at half
// rate on gfx950, so pairing scalars only costs register moves.
//
// Semantics are those of fftw_plan_dft_1d as the reference calls it
// (fftc/main.c:31-43): unnormalised in both directions, sign -1 forward.
// The backward transform is conj(forward(conj(x))).
#include "common.h"
#include "fft_internal.h"
#include "fir_half.h"
#include "tone_math.h"

#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace {

template <typename T>
struct alignas(2 * sizeof(T)) C {
    T re, im;
};

template <typename T> __device__ __forceinline__ C<T> operator+(C<T> a, C<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ __forceinline__ C<T> operator-(C<T> a, C<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T> __device__ __forceinline__ C<T> cmul(C<T> a, C<T> b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
// multiply by -i (forward quarter turn)
template <typename T> __device__ __forceinline__ C<T> mul_mi(C<T> a) { return {a.im, -a.re}; }

// nontemporal (streaming) global access of one complex point
template <typename T>
__device__ __forceinline__ C<T> nt_load(const C<T> *p)
{
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 v = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(p));
    return {v.x, v.y};
}
template <typename T>
__device__ __forceinline__ void nt_store(C<T> q, C<T> *p)
{
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 v = {q.re, q.im};
    __builtin_nontemporal_store(v, reinterpret_cast<V2 *>(p));
}

// cos/sin of 2*pi*k/32, k = 0..31
__device__ constexpr double kCos32[32] = {
    1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254523567,
    0.70710678118654757274, 0.55557023301960228867, 0.38268343236508983729, 0.19509032201612833135,
    0.00000000000000006123, -0.19509032201612819257, -0.38268343236508972627, -0.55557023301960195560,
    -0.70710678118654746172, -0.83146961230254534669, -0.92387953251128673848, -0.98078528040323043058,
    -1.00000000000000000000, -0.98078528040323043058, -0.92387953251128684951, -0.83146961230254545772,
    -0.70710678118654768376, -0.55557023301960217765, -0.38268343236509033689, -0.19509032201612866442,
    -0.00000000000000018370, 0.19509032201612830359, 0.38268343236509000382, 0.55557023301960184458,
    0.70710678118654735069, 0.83146961230254523567, 0.92387953251128651644, 0.98078528040323031956,
};
__device__ constexpr double kSin32[32] = {
    0.00000000000000000000, 0.19509032201612824808, 0.38268343236508978178, 0.55557023301960217765,
    0.70710678118654746172, 0.83146961230254523567, 0.92387953251128673848, 0.98078528040323043058,
    1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254545772,
    0.70710678118654757274, 0.55557023301960217765, 0.38268343236508989280, 0.19509032201612860891,
    0.00000000000000012246, -0.19509032201612835911, -0.38268343236508967076, -0.55557023301960195560,
    -0.70710678118654746172, -0.83146961230254523567, -0.92387953251128651644, -0.98078528040323031956,
    -1.00000000000000000000, -0.98078528040323043058, -0.92387953251128662746, -0.83146961230254545772,
    -0.70710678118654768376, -0.55557023301960217765, -0.38268343236509039240, -0.19509032201612871993,
};

// a * exp(-2*pi*i*K/32), K a compile-time constant; trivial factors cost no multiply
template <int K, typename T>
__device__ __forceinline__ C<T> mul_w32(C<T> a)
{
    constexpr int k = K & 31;
    if constexpr (k == 0) return a;
    else if constexpr (k == 8) return mul_mi(a);
    else if constexpr (k == 16) return {-a.re, -a.im};
    else if constexpr (k == 24) return {-a.im, a.re};
    else {
        constexpr T c = (T)kCos32[k], s = (T)(-kSin32[k]);
        return {a.re * c - a.im * s, a.re * s + a.im * c};
    }
}

// t[a*RB + c] *= W_R^{a*c} for every element of the RA x RB intermediate
template <int R, int RB, typename T, int... I>
__device__ __forceinline__ void inner_twiddle(C<T> *t, std::integer_sequence<int, I...>)
{
    ((t[I] = mul_w32<(I / RB) * (I % RB) * (32 / R)>(t[I])), ...);
}

// Forward DFT of R points held in v[0], v[S], v[2S], ... (natural order in and out).
template <int R, int S, typename T>
__device__ __forceinline__ void dft_reg(C<T> *v)
{
    if constexpr (R == 1) {
    } else if constexpr (R == 2) {
        C<T> a = v[0], b = v[S];
        v[0] = a + b;
        v[S] = a - b;
    } else if constexpr (R == 4) {
        C<T> t0 = v[0] + v[2 * S], t1 = v[0] - v[2 * S];
        C<T> t2 = v[S] + v[3 * S], t3 = mul_mi(v[S] - v[3 * S]);
        v[0] = t0 + t2;
        v[S] = t1 + t3;
        v[2 * S] = t0 - t2;
        v[3 * S] = t1 - t3;
    } else {
        // R = RA*RB, n = a + RA*b, k = c + RB*d
        constexpr int RA = (R == 8) ? 2 : 4, RB = R / RA;
        C<T> t[R];
#pragma unroll
        for (int a = 0; a < RA; a++) {
#pragma unroll
            for (int b = 0; b < RB; b++) t[a * RB + b] = v[(a + RA * b) * S];
            dft_reg<RB, 1>(&t[a * RB]);  // t[a][c]
        }
        // twiddle t[a][c] by W_R^{a*c}, then RA-point DFTs over a
        inner_twiddle<R, RB>(t, std::make_integer_sequence<int, R>{});
#pragma unroll
        for (int c = 0; c < RB; c++) {
            dft_reg<RA, RB>(&t[c]);  // over a, stride RB -> index d
#pragma unroll
            for (int d = 0; d < RA; d++) v[(c + RB * d) * S] = t[d * RB + c];
        }
    }
}

__device__ __forceinline__ unsigned pad(unsigned i) { return i + (i >> 4); }
constexpr unsigned padded_len(unsigned n) { return n + (n >> 4) + 1; }

template <int NT>
__device__ __forceinline__ void frame_sync()
{
    if constexpr (NT > 64) {
        __syncthreads();
    } else {
        // the frame's lanes are in one wave: LDS operations of a wave complete in
        // issue order, so only the compiler has to be kept from reordering them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The lane index as the transform passes see it.  With twiddles read from the global table
// (TW == 0) every twiddle load has a loop-invariant address, and the compiler hoists all of them
// out of the frame loop -- hundreds of registers, spilled to scratch.  Passing the index through
// an empty asm once per frame makes the loads per-frame again (they hit L1 / L2).  The split
// tables (TW == 3) have the same problem with their index arithmetic.
template <int TW>
__device__ __forceinline__ unsigned pass_lane(unsigned tl)
{
    if constexpr (TW == 0 || TW == 3) asm volatile("" : "+v"(tl));
    return tl;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Number of twiddle registers (complex) pass `R` with P points per lane needs.
constexpr int tw_count(int P, int R, int NS) { return (R > 1 && NS > 1) ? (P / R) * (R - 1) : 0; }

template <typename T, int N, int NT, int R, int NS>
__device__ __forceinline__ void load_twiddles(C<T> *twr, const C<T> *__restrict__ tw, unsigned tl)
{
    constexpr int P = N / NT;
    if constexpr (R > 1 && NS > 1) {
#pragma unroll
        for (int b = 0; b < P / R; b++) {
            const unsigned j = tl + NT * b;
            const unsigned k = j % NS;
#pragma unroll
            for (int r = 1; r < R; r++) twr[b * (R - 1) + r - 1] = tw[k * r * (N / (NS * R))];
        }
    }
}

// One pass over the P register-resident points of this lane.
//   FIRST: sources are v as loaded from global by the caller; otherwise read from LDS
//   LAST : results stay in v (caller stores to global); otherwise written to LDS
// Twiddle source (TW): 0 = the global table (through L1), 1 = registers filled once per
// lane (twr), 2 = a per-workgroup LDS table laid out [r-1][j mod NS] (twr points at it),
// 3 = that table for passes with NS <= 16 and, for the others, two short LDS tables
// A[a] = W^(64a), B[b] = W^b whose product is the twiddle (twr points at A, B follows).
// Twiddles and butterflies of one pass over the P register-resident points of this lane.
template <typename T, int N, int NT, int R, int NS, int TW>
__device__ __forceinline__ void pass_butterflies(C<T> *v, const C<T> *twr, const C<T> *__restrict__ tw, unsigned tl)
{
    constexpr int P = N / NT, NB = P / R;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if constexpr (NS > 1) {
            const unsigned k = (tl + NT * b) % NS;
#pragma unroll
            for (int r = 1; r < R; r++) {
                C<T> w;
                if constexpr (TW == 1) w = twr[b * (R - 1) + r - 1];
                else if constexpr (TW == 2 || (TW == 3 && NS <= 16)) w = twr[(r - 1) * NS + k];
                else if constexpr (TW == 3) {
                    // W^e, e = 64*a + b, as the product of two short tables
                    const unsigned e = k * (r * (N / (NS * R)));
                    w = cmul(twr[e >> 6], twr[N / 64 + (e & 63)]);
                } else w = tw[k * r * (N / (NS * R))];
                v[b * R + r] = cmul(v[b * R + r], w);
            }
        }
        dft_reg<R, 1>(&v[b * R]);
    }
}

template <typename T, int N, int NT, int R, int NS, bool FIRST, bool LAST, int TW>
__device__ __forceinline__ void fft_pass(C<T> *v, C<T> *lds, const C<T> *twr, const C<T> *__restrict__ tw, unsigned tl)
{
    constexpr int P = N / NT, NB = P / R;
    if constexpr (R == 1) return;
    if constexpr (!FIRST) {
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < R; r++) v[b * R + r] = lds[pad(tl + NT * b + r * (N / R))];
        frame_sync<NT>();
    }
    pass_butterflies<T, N, NT, R, NS, TW>(v, twr, tw, tl);
    if constexpr (!LAST) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned j = tl + NT * b;
            const unsigned j0 = (j / NS) * (NS * R) + (j % NS);
#pragma unroll
            for (int r = 0; r < R; r++) lds[pad(j0 + r * NS)] = v[b * R + r];
        }
        frame_sync<NT>();
    }
}

// Per-workgroup twiddle tables behind the FPB frame buffers: pass p holds W^{k*r},
// r = 1..R-1, k < NS, as [r-1][k] so a wave reads contiguous entries.  Ends with a barrier.
// TWS: the table `tw` holds exp(-2 pi i k / (TWS * N)): every TWS-th entry is read (a longer
// transform's table serving its N-point rows).
template <typename T, int N, int BT, int FPB, int R0, int R1, int R2, int R3, int TWS = 1>
__device__ __forceinline__ void setup_lds_twiddles(char *smem_raw, const C<T> *__restrict__ tw, const C<T> *&tw1,
                                                   const C<T> *&tw2, const C<T> *&tw3)
{
    constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2;
    C<T> *t1 = reinterpret_cast<C<T> *>(smem_raw) + (size_t)FPB * padded_len(N);
    C<T> *t2 = t1 + (R1 > 1 ? (R1 - 1) * NS1 : 0);
    C<T> *t3 = t2 + (R2 > 1 ? (R2 - 1) * NS2 : 0);
    if constexpr (R1 > 1)
        for (unsigned i = threadIdx.x; i < (R1 - 1) * NS1; i += BT) t1[i] = tw[TWS * ((i % NS1) * (i / NS1 + 1) * (N / (NS1 * R1)))];
    if constexpr (R2 > 1)
        for (unsigned i = threadIdx.x; i < (R2 - 1) * NS2; i += BT) t2[i] = tw[TWS * ((i % NS2) * (i / NS2 + 1) * (N / (NS2 * R2)))];
    if constexpr (R3 > 1)
        for (unsigned i = threadIdx.x; i < (R3 - 1) * NS3; i += BT) t3[i] = tw[TWS * ((i % NS3) * (i / NS3 + 1) * (N / (NS3 * R3)))];
    __syncthreads();
    tw1 = t1;
    tw2 = t2;
    tw3 = t3;
}

// Long frames: full tables hold about N entries whatever the radix order, as much LDS as the
// frame itself.  Only the first twiddled pass (NS = R0 <= 16) keeps its table; the other passes
// build W^e from A[e >> 6] * B[e & 63], N/64 + 64 entries in all, at the price of one complex
// multiplication per twiddle.  Ends with a barrier.
template <typename T, int N, int BT, int FPB, int R0, int R1, size_t IMAGE_BYTES = (size_t)FPB * padded_len(N) * sizeof(C<T>)>
__device__ __forceinline__ void setup_lds_twiddles_split(char *smem_raw, const C<T> *__restrict__ tw, const C<T> *&tw1,
                                                         const C<T> *&tw2, const C<T> *&tw3)
{
    static_assert(R0 <= 16 && N % 64 == 0, "first pass table only for NS <= 16");
    static_assert(IMAGE_BYTES % sizeof(C<T>) == 0, "tables start on a complex element");
    C<T> *t1 = reinterpret_cast<C<T> *>(smem_raw + IMAGE_BYTES);
    C<T> *ta = t1 + (R1 > 1 ? (R1 - 1) * R0 : 0);
    if constexpr (R1 > 1)
        for (unsigned i = threadIdx.x; i < (R1 - 1) * R0; i += BT) t1[i] = tw[(i % R0) * (i / R0 + 1) * (N / (R0 * R1))];
    for (unsigned i = threadIdx.x; i < N / 64; i += BT) ta[i] = tw[64 * i];
    for (unsigned i = threadIdx.x; i < 64; i += BT) ta[N / 64 + i] = tw[i];
    __syncthreads();
    tw1 = t1;
    tw2 = ta;
    tw3 = ta;
}

constexpr size_t lds_twiddle_entries(int TW, int R0, int R1, int R2, int R3, int N = 0)
{
    return TW == 2   ? (size_t)(R1 > 1 ? (R1 - 1) * R0 : 0) + (R2 > 1 ? (R2 - 1) * R0 * R1 : 0) +
                         (R3 > 1 ? (R3 - 1) * R0 * R1 * R2 : 0)
           : TW == 3 ? (size_t)(R1 > 1 ? (R1 - 1) * R0 : 0) + N / 64 + 64
                     : 0;
}

// All passes of one frame: v holds the first pass's inputs on entry and the last pass's
// outputs (butterfly j = tl + NT*b, output r at position j + r*N/RLAST) on return.
template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
__device__ __forceinline__ void run_passes(C<T> *v, C<T> *lds, const C<T> *tw1, const C<T> *tw2, const C<T> *tw3,
                                           const C<T> *__restrict__ tw, unsigned tl)
{
    constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    fft_pass<T, N, NT, R0, 1, true, NPASS == 1, TW>(v, lds, nullptr, tw, tl);
    if constexpr (R1 > 1) fft_pass<T, N, NT, R1, NS1, false, NPASS == 2, TW>(v, lds, tw1, tw, tl);
    if constexpr (R2 > 1) fft_pass<T, N, NT, R2, NS2, false, NPASS == 3, TW>(v, lds, tw2, tw, tl);
    if constexpr (R3 > 1) fft_pass<T, N, NT, R3, NS3, false, NPASS == 4, TW>(v, lds, tw3, tw, tl);
}

// run_passes<float, 1024, 64, 16, 4, 16, 1, 2> -- a 1024-point frame inside one wave, sixteen
// points per lane, full twiddle tables in LDS -- with the LDS addresses written as a few lane
// bases plus compile-time constants (pad(i + c) = pad(i) + 17 c / 16 for c a multiple of 16),
// so every access is a base register and an immediate offset.  The generic passes form each of
// the 64 frame addresses and 18 table addresses with three or four integer instructions; here
// the transform costs about ten.  Same arithmetic in the same order, so the same bits.
__device__ __forceinline__ void row_passes_1024(C<float> *v, C<float> *lds, const C<float> *tw1, const C<float> *tw2,
                                                unsigned tl)
{
    const unsigned rb = tl + (tl >> 4);                // pad(tl)
    const unsigned w1 = 17 * tl;                       // pad(16 tl)
    const unsigned w2 = 68 * (tl >> 4) + (tl & 15);    // pad(64 (tl / 16) + tl % 16)
    // pass 1: radix 16 across the sixteen points 64 apart; butterfly tl writes 16 tl + r
    dft_reg<16, 1>(v);
#pragma unroll
    for (int r = 0; r < 16; r++) lds[w1 + r] = v[r];
    frame_sync<64>();
    // pass 2: radix 4 over sixteen interleaved sub-transforms; butterflies tl + 64 b, all with
    // the same twiddles (64 b is a multiple of 16)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) v[4 * b + r] = lds[rb + 68 * b + 272 * r];
    frame_sync<64>();
    {
        const C<float> *t = tw1 + (tl & 15);
        const C<float> wa = t[0], wb = t[16], wc = t[32];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            v[4 * b + 1] = cmul(v[4 * b + 1], wa);
            v[4 * b + 2] = cmul(v[4 * b + 2], wb);
            v[4 * b + 3] = cmul(v[4 * b + 3], wc);
            dft_reg<4, 1>(&v[4 * b]);
        }
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) lds[w2 + 272 * b + 17 * r] = v[4 * b + r];
    frame_sync<64>();
    // pass 3: radix 16 over 64 interleaved sub-transforms; results stay in registers, output r
    // of butterfly tl at position tl + 64 r
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[rb + 68 * r];
    frame_sync<64>();
    {
        const C<float> *t = tw2 + tl;
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = cmul(v[r], t[64 * (r - 1)]);
    }
    dft_reg<16, 1>(v);
}

// run_passes<float, 512, 64, 8, 8, 8, 1, 2> -- a 512-point frame inside one wave, eight points
// per lane -- written the same way: pad(8 tl + r) = 8 tl + tl / 2 + r, pad(tl + 64 r) = pad(tl) + 68 r,
// pad(64 (tl / 8) + tl % 8 + 8 r) = 68 (tl / 8) + tl % 8 + 8 r + r / 2.
__device__ __forceinline__ void row_passes_512(C<float> *v, C<float> *lds, const C<float> *tw1, const C<float> *tw2, unsigned tl)
{
    const unsigned rb = tl + (tl >> 4);
    const unsigned w1 = 8 * tl + (tl >> 1);
    const unsigned w2 = 68 * (tl >> 3) + (tl & 7);
    dft_reg<8, 1>(v);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[w1 + r] = v[r];
    frame_sync<64>();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = lds[rb + 68 * r];
    frame_sync<64>();
    {
        const C<float> *t = tw1 + (tl & 7);
#pragma unroll
        for (int r = 1; r < 8; r++) v[r] = cmul(v[r], t[8 * (r - 1)]);
    }
    dft_reg<8, 1>(v);
#pragma unroll
    for (int r = 0; r < 8; r++) lds[w2 + 8 * r + (r >> 1)] = v[r];
    frame_sync<64>();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = lds[rb + 68 * r];
    frame_sync<64>();
    {
        const C<float> *t = tw2 + tl;
#pragma unroll
        for (int r = 1; r < 8; r++) v[r] = cmul(v[r], t[64 * (r - 1)]);
    }
    dft_reg<8, 1>(v);
}

template <int NT> constexpr int block_threads() { return NT > 256 ? NT : 256; }

// Workgroups to launch for `nfb` frame-blocks when `resident` fit the chip at once.  A grid of
// just the resident set, each workgroup looping over many frame-blocks, streams 5-8 % slower than
// a grid the dispatcher keeps refilling: the memory-bound kernels run fastest when a workgroup
// handles only as many frame-blocks as it takes to amortise its set-up (`iters`; measured optimum
// ~2.7 for the 8 KB twiddle table of 1024 points, 1 where the tables are a few hundred bytes),
// the compute-bound ones with a few times the resident set.  DSPMI_GRIDX=<k> forces
// resident * k for experiments.
inline size_t grid_for(size_t nfb, size_t resident, double iters)
{
    size_t grid = resident;
    const char *env = getenv("DSPMI_GRIDX");
    if (env && atoi(env) > 0) {
        grid *= (size_t)atoi(env);
    } else if (iters > 0) {
        const size_t g = (size_t)((double)nfb / iters + 0.999);
        if (g > grid) grid = g;
    }
    if (grid > nfb) grid = nfb;
    return grid > 0x7fffffffu ? 0x7fffffffu : grid;  // the kernels stride by the grid: any size covers every frame
}

// Frame-blocks per workgroup that amortise the LDS twiddle table set-up (see grid_for).
constexpr double table_iters(size_t tw_entries, size_t esz)
{
    return tw_entries * esz > 3072 ? (double)(tw_entries * esz) / 3072.0 : 1.0;
}

// The fused overlap-save kernel is latency-bound and its occupancy LDS-bound (frame images plus
// one twiddle table per workgroup): eight 1024-point frames per workgroup share the table, which
// fits 16 waves on a CU where four-frame workgroups fit 12.
template <int N, int NT> constexpr int ols_block_threads() { return N == 1024 ? 512 : block_threads<NT>(); }
// The complex kernel at 1024 points (f32): eight frames per workgroup share one 8 KB twiddle table,
// so two workgroups = sixteen waves fit a CU where three four-frame workgroups are twelve; with one
// frame-block per workgroup 2.89 ms for 2^20 frames against 2.98 (5.94 against 5.77 TB/s; sixteen
// frames per workgroup, one workgroup per CU: 3.6-3.7 ms).
template <typename T, int N, int NT> constexpr int c2c_block_threads() { return (N == 1024 && sizeof(T) == 4) ? 512 : block_threads<NT>(); }

// flag bits of the c2c kernel
constexpr int kFftBackward = 1;   // conj . forward . conj
// bits 1-2: log2 of the number of fronts the frame-blocks are visited from (see the frame loop)

template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW, bool PREFETCH, int NTMEM = 0>
__global__ __launch_bounds__((c2c_block_threads<T, N, NT>())) void fft_c2c_kernel(const C<T> *in, C<T> *out,
                                                                      const C<T> *__restrict__ tw, size_t batch,
                                                                      int flags)
{
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    constexpr int P = N / NT;
    constexpr int BT = c2c_block_threads<T, N, NT>();
    constexpr int FPB = BT / NT;  // frames per workgroup per iteration
    constexpr int NS1 = R0, NS2 = R0 * R1, NS3 = R0 * R1 * R2;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    constexpr int RLAST = NPASS == 1 ? R0 : NPASS == 2 ? R1 : NPASS == 3 ? R2 : R3;
    constexpr int NSLAST = N / RLAST;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned slot = threadIdx.x / NT;
    const unsigned tl = threadIdx.x % NT;
    C<T> *lds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)slot * padded_len(N);

    constexpr bool TWREG = TW == 1;
    C<T> tw1r[cmax(1, TWREG ? tw_count(P, R1, NS1) : 0)];
    C<T> tw2r[cmax(1, TWREG ? tw_count(P, R2, NS2) : 0)];
    C<T> tw3r[cmax(1, TWREG ? tw_count(P, R3, NS3) : 0)];
    const C<T> *tw1 = tw1r, *tw2 = tw2r, *tw3 = tw3r;
    if constexpr (TWREG) {
        load_twiddles<T, N, NT, R1, NS1>(tw1r, tw, tl);
        load_twiddles<T, N, NT, R2, NS2>(tw2r, tw, tl);
        load_twiddles<T, N, NT, R3, NS3>(tw3r, tw, tl);
    }
    if constexpr (TW == 2) setup_lds_twiddles<T, N, BT, FPB, R0, R1, R2, R3>(smem_raw, tw, tw1, tw2, tw3);
    if constexpr (TW == 3) setup_lds_twiddles_split<T, N, BT, FPB, R0, R1>(smem_raw, tw, tw1, tw2, tw3);
    const T sgn = (flags & kFftBackward) ? T(-1) : T(1);

    const size_t nfb = (batch + FPB - 1) / FPB;

    // Short frames (fewer than 16 lanes each): a lane's own points are N/R0 apart and the lanes of
    // a frame cover only 16*NT contiguous bytes, so direct global access would touch memory in
    // 16-128 byte pieces.  The wave's 64/NT frames are contiguous in memory instead: they are
    // moved as one 64*P-element run, 512 contiguous bytes per instruction, through the frames'
    // LDS images.
    constexpr bool STAGED = NT < 16;
    constexpr int FPW = 64 / (NT < 64 ? NT : 64);  // frames per wave
    const unsigned wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    C<T> *wlds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)wv * FPW * padded_len(N);

    // first pass reads global memory: lane-contiguous, R0 loads N/R0 apart
    auto load_frame = [&](C<T> (&v)[P], size_t fb) {
        if constexpr (STAGED) {
            const size_t f0 = fb * FPB + (size_t)wv * FPW;
            const C<T> *chunk = in + f0 * N;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const unsigned e = ln + 64 * i, fr = e / N, idx = e % N;
                C<T> q = {T(0), T(0)};
                if (f0 + fr < batch) q = (NTMEM & 1) ? nt_load(&chunk[e]) : chunk[e];
                wlds[fr * padded_len(N) + pad(idx)] = q;
            }
            frame_sync<NT>();
#pragma unroll
            for (int b = 0; b < P / R0; b++)
#pragma unroll
                for (int r = 0; r < R0; r++) v[b * R0 + r] = lds[pad(tl + NT * b + r * (N / R0))];
            frame_sync<NT>();
            return;
        }
        const size_t frame = fb * FPB + slot;
        const C<T> *src = in + frame * N;
        const bool active = frame < batch;
#pragma unroll
        for (int b = 0; b < P / R0; b++)
#pragma unroll
            for (int r = 0; r < R0; r++) {
                C<T> q = {T(0), T(0)};
                if (active) q = (NTMEM & 1) ? nt_load(&src[tl + NT * b + r * (N / R0)]) : src[tl + NT * b + r * (N / R0)];
                v[b * R0 + r] = q;
            }
    };
    auto process = [&](C<T> (&v)[P], size_t fb) {
        const size_t frame = fb * FPB + slot;
        const bool active = frame < batch;
        C<T> *dst = out + frame * N;
#pragma unroll
        for (int i = 0; i < P; i++) v[i].im *= sgn;
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));
        if constexpr (STAGED) {
            // outputs to their natural places in the frame's image, then out as one contiguous run
#pragma unroll
            for (int b = 0; b < P / RLAST; b++)
#pragma unroll
                for (int r = 0; r < RLAST; r++) {
                    C<T> q = v[b * RLAST + r];
                    q.im *= sgn;
                    lds[pad(tl + NT * b + r * NSLAST)] = q;
                }
            frame_sync<NT>();
            const size_t f0 = fb * FPB + (size_t)wv * FPW;
            C<T> *chunk = out + f0 * N;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const unsigned e = ln + 64 * i, fr = e / N, idx = e % N;
                if (f0 + fr < batch) {
                    const C<T> q = wlds[fr * padded_len(N) + pad(idx)];
                    if constexpr ((NTMEM & 2) != 0) nt_store(q, &chunk[e]);
                    else chunk[e] = q;
                }
            }
            frame_sync<NT>();
            return;
        }
        // last pass writes global memory: butterfly j, output r goes to j + r*N/RLAST
        if (active) {
#pragma unroll
            for (int b = 0; b < P / RLAST; b++)
#pragma unroll
                for (int r = 0; r < RLAST; r++) {
                    C<T> q = v[b * RLAST + r];
                    q.im *= sgn;
                    if constexpr ((NTMEM & 2) != 0) nt_store(q, &dst[tl + NT * b + r * NSLAST]);
                    else dst[tl + NT * b + r * NSLAST] = q;
                }
        }
    };

    if constexpr (PREFETCH) {
        // two register sets: the next frame's loads are in flight while this one is computed
        const size_t step = gridDim.x;
        size_t fb = blockIdx.x;
        if (fb >= nfb) return;
        C<T> va[P], vb[P];
        load_frame(va, fb);
        while (true) {
            const bool more1 = fb + step < nfb;
            if (more1) load_frame(vb, fb + step);
            process(va, fb);
            if (!more1) break;
            fb += step;
            const bool more2 = fb + step < nfb;
            if (more2) load_frame(va, fb + step);
            process(vb, fb);
            if (!more2) break;
            fb += step;
        }
    } else {
        for (size_t fb0 = blockIdx.x; fb0 < nfb; fb0 += gridDim.x) {
            size_t fb = fb0;
            if (const unsigned lf = (flags >> 1) & 3) {
                // Launched as a moving front, the kernel's rate depends on the distance between
                // the two buffers modulo the 32 MiB period of the memory-channel mapping (5.35 to
                // 5.9 TB/s at N = 1024).  Workgroups therefore alternate between 2^lf fronts spread
                // evenly over every full 32 MiB group of frame-blocks: reads and writes then meet
                // at several phases at once and the rate no longer depends on where the buffers are.
                constexpr size_t G = ((size_t)32 << 20) / (FPB * N * sizeof(C<T>));
                const size_t g0 = fb0 / G * G;
                if (g0 + G <= nfb) {
                    const size_t i = fb0 - g0;
                    fb = g0 + (i & ((1u << lf) - 1)) * (G >> lf) + (i >> lf);
                }
            }
            C<T> v[P];
            load_frame(v, fb);
            process(v, fb);
        }
    }
}

// ---------------------------------------------------------------------------
// real input: n = 2N real samples per frame through one N-point complex transform
// ---------------------------------------------------------------------------

// Sum of `val` over the NT lanes that own one frame.
template <int NT>
__device__ __forceinline__ float frame_sum(float val, float *scratch, unsigned slot, unsigned tl)
{
    constexpr int W = NT < 64 ? NT : 64;
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
    if constexpr (NT > 64) {
        constexpr int NW = NT / 64;
        __syncthreads();
        if ((tl & 63) == 0) scratch[slot * NW + tl / 64] = val;
        __syncthreads();
        val = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) val += scratch[slot * NW + w];
    }
    return val;
}

// EPI 1: out = bins 0..N of the 2N-point real transform (the fftw_plan_dft_r2c_1d contract,
//        fft/src/main.c:155-158), N+1 complex per frame.
// EPI 2: out = magnitude spectrum over nbin bins (N+1 or N), normalised as `mode` says
//        (0: divide by the sum when it exceeds 1e-5, fft/src/main.c:120-141;
//         1: always divide, alsa/main.c:473-487; 2: 2|X|/scale, fftc/main.c:80-105).
// The frame's packed transform Z stays on chip: it is parked in the frame's LDS image and
// bin k is formed from Z[k] and Z[N-k] there (X[k] = E[k] + W_2N^k O[k]).
template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW, int EPI>
__global__ __launch_bounds__(block_threads<NT>()) void fft_r2c_kernel(const C<T> *in, void *out_raw,
                                                                      const C<T> *__restrict__ tw,
                                                                      const C<T> *__restrict__ tw2n, size_t batch,
                                                                      unsigned nbin, int mode, float scale, int pcm)
{
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    constexpr int P = N / NT;
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    constexpr int RLAST = NPASS == 1 ? R0 : NPASS == 2 ? R1 : NPASS == 3 ? R2 : R3;
    constexpr int NSLAST = N / RLAST;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned slot = threadIdx.x / NT;
    const unsigned tl = threadIdx.x % NT;
    C<T> *lds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)slot * padded_len(N);
    float *scratch = reinterpret_cast<float *>(reinterpret_cast<C<T> *>(smem_raw) + (size_t)FPB * padded_len(N) +
                                               lds_twiddle_entries(TW, R0, R1, R2, R3, N));

    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    if constexpr (TW == 2) setup_lds_twiddles<T, N, BT, FPB, R0, R1, R2, R3>(smem_raw, tw, tw1, tw2, tw3);
    if constexpr (TW == 3) setup_lds_twiddles_split<T, N, BT, FPB, R0, R1>(smem_raw, tw, tw1, tw2, tw3);

    const size_t nfb = (batch + FPB - 1) / FPB;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * FPB + slot;
        const bool active = frame < batch;
        const C<T> *src = in + frame * N;  // 2N reals read as N complex points
        C<T> v[P];
#pragma unroll
        for (int b = 0; b < P / R0; b++)
#pragma unroll
            for (int r = 0; r < R0; r++) {
                C<T> q = {T(0), T(0)};
                if (active) q = nt_load(&src[tl + NT * b + r * (N / R0)]);
                if constexpr (sizeof(T) == 4) {
                    // pcm: the same 8 bytes are two interleaved 16-bit stereo frames, each
                    // entering as (left + right) / 2 (alsa/main.c:464-468)
                    if (pcm) {
                        const int f0 = __float_as_int(q.re), f1 = __float_as_int(q.im);
                        q.re = (float)((short)(f0 & 0xffff) + (f0 >> 16)) * 0.5f;
                        q.im = (float)((short)(f1 & 0xffff) + (f1 >> 16)) * 0.5f;
                    }
                }
                v[b * R0 + r] = q;
            }
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));

        // park Z in natural order (the last pass's LDS reads are behind a frame_sync already)
#pragma unroll
        for (int b = 0; b < P / RLAST; b++)
#pragma unroll
            for (int r = 0; r < RLAST; r++) lds[pad(tl + NT * b + r * NSLAST)] = v[b * RLAST + r];
        frame_sync<NT>();

        // Bins come in pairs: with E, O the even / odd half-transforms,
        //   X[k] = E[k] + W^k O[k],   X[N-k] = conj(E[k] - W^k O[k]),   W = exp(-2 pi i / 2N),
        // so one pair of LDS reads and one twiddle give two bins.  A lane forms the pairs
        // k = tl + NT*i, i < P/2 (k = 0 pairs bin 0 with the Nyquist bin N); lane 0 adds the
        // self-paired bin N/2 = conj(Z[N/2]).
        constexpr int H = P / 2;
        // The lane index is laundered through an empty asm: otherwise every address, offset and
        // twiddle below is loop-invariant, gets hoisted out of the frame loop and occupies ~50
        // registers through the transform passes (2 waves per SIMD instead of 3-4).
        unsigned te = tl;
        asm volatile("" : "+v"(te));
        C<T> xlo[H], xhi[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            const unsigned k = te + NT * i;
            const C<T> a = lds[pad(k)];
            const C<T> b0 = lds[pad((N - k) & (N - 1))];
            const C<T> e = {(a.re + b0.re) * T(0.5), (a.im - b0.im) * T(0.5)};
            const C<T> d = {(a.re - b0.re) * T(0.5), (a.im + b0.im) * T(0.5)};
            const C<T> o = {d.im, -d.re};  // -i * d
            const C<T> wo = cmul(tw2n[k], o);
            xlo[i] = {e.re + wo.re, e.im + wo.im};
            xhi[i] = {e.re - wo.re, wo.im - e.im};
        }
        const C<T> zmid = lds[pad(N / 2)];
        const C<T> xmid = {zmid.re, -zmid.im};
        frame_sync<NT>();  // reads done before the next frame's passes overwrite the image

        if constexpr (EPI == 1) {
            C<T> *dst = static_cast<C<T> *>(out_raw) + frame * (N + 1);
            if (active) {
#pragma unroll
                for (int i = 0; i < H; i++) {
                    nt_store(xlo[i], &dst[te + NT * i]);
                    nt_store(xhi[i], &dst[N - (te + NT * i)]);
                }
                if (tl == 0) dst[N / 2] = xmid;
            }
        } else {
            float mlo[H], mhi[H];
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < H; i++) {
                // the hardware square root (1 ulp): the library routine's extra steps would make
                // this epilogue cost more vector instructions than the transform itself
                mlo[i] = __builtin_amdgcn_sqrtf((float)(xlo[i].re * xlo[i].re + xlo[i].im * xlo[i].im));
                mhi[i] = __builtin_amdgcn_sqrtf((float)(xhi[i].re * xhi[i].re + xhi[i].im * xhi[i].im));
                // bin N (the partner of bin 0) belongs to the spectrum only when nbin = N + 1
                part += mlo[i] + ((i == 0 && tl == 0 && nbin <= N) ? 0.f : mhi[i]);
            }
            const float mmid = __builtin_amdgcn_sqrtf((float)(xmid.re * xmid.re + xmid.im * xmid.im));
            if (tl == 0) part += mmid;
            float div = 1.f, mul = 1.f;
            if (mode == 2) {
                mul = 2.f;
                div = scale;
            } else {
                const float sum = frame_sum<NT>(part, scratch, slot, tl);
                if (mode == 1 || sum > 0.00001f) div = sum;
            }
            // one reciprocal per frame instead of a division per bin (2-3 ulp in all, against a
            // tolerance of 1e-5); a zero sum in mode 1 still yields inf / nan as a division would
            const float gain = mul * __builtin_amdgcn_rcpf(div);
            float *dst = static_cast<float *>(out_raw) + frame * nbin;
            if (active) {
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const unsigned k = te + NT * i;
                    __builtin_nontemporal_store(mlo[i] * gain, &dst[k]);
                    if (N - k < nbin) __builtin_nontemporal_store(mhi[i] * gain, &dst[N - k]);
                }
                if (tl == 0) dst[N / 2] = mmid * gain;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// FIR -> 4096-point real transform -> normalised magnitude spectrum, in one kernel
// ---------------------------------------------------------------------------

// The chain of alsa/main.c:542-568 after the tone generator: do_fir, then do_power_spectrum on
// frames of 4096 filtered samples.  A tile of the f16 FIR walker (fir_half.h) is 4096 outputs,
// i.e. one frame, so its results go into an LDS frame image instead of HBM, and once the tile's
// MFMAs are done the workgroup's 256 lanes run the frame's packed 2048-point complex transform
// (8 points per lane, radices 8.8.8.4, split twiddle tables), form the bins from Z[k] and
// Z[N-k], take magnitudes, sum, normalise and store: 4 B read and 2 B written per sample where
// the two kernels move 14.  Every memory instruction of the epilogue is unconditional (a store
// that must not happen goes to an out-of-range offset of its descriptor), so the walker's
// prefetch keeps its exact wait counts.
template <int NS>
struct SpectrumSink {
    static constexpr bool kTileHook = true;
    static constexpr bool kRedo = false;     // a frame with a non-finite sample in its window: a spectrum of NaNs
    static constexpr bool kUnscaled = true;  // emit() takes the sums as they leave the matrix unit; the tile's factor goes into the spectrum's gain
    static constexpr int N = 2048, NT = 256, R0 = 8, R1 = 8, R2 = 8, R3 = 4, TW = 3, P = N / NT;
    C<float> *img;                    // the frame image, padded_len(N) complex
    const C<float> *tw1, *tw2, *tw3;  // LDS twiddle tables
    const C<float> *tw, *tw2n;        // exp(-2 pi i k / N), exp(-2 pi i k / 2N) in device memory
    float *scratch;                   // the waves' partial sums
    float *ps;
    unsigned nbin, frames_per_chan;
    int mode;
    float scale;
    size_t frame;
    float tile_unscale;               // what the tile's results still have to be multiplied by (applied to the spectrum)
    // A lane meets the same twiddles in every tile: those of the two long passes stay in registers
    // for the whole launch (pass 3: W^(4 k r), k = lane mod 64, r = 1..7; pass 4: W^(j r), j = lane
    // and lane + 256, r = 1..3) where the 26 registers do not cost a workgroup per CU; longer
    // filters, whose tap images fill the registers, form them per tile as powers of one table product.
    static constexpr bool kResident = NS <= 5;
    C<float> rot3[kResident ? 7 : 1], rot4[2][kResident ? 3 : 1];

    __device__ __forceinline__ void load_twiddles()
    {
        if constexpr (!kResident) return;
        const unsigned tl = threadIdx.x;
#pragma unroll
        for (int r = 1; r < 8; r++) rot3[r - 1] = tw[(4 * (tl & 63) * r) & (N - 1)];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 1; r < 4; r++) rot4[b][r - 1] = tw[((tl + 256 * b) * r) & (N - 1)];
    }

    __device__ __forceinline__ void begin_tile(unsigned chan, long tile_start, float unscale)
    {
        frame = (size_t)chan * frames_per_chan + (size_t)(tile_start / kSplitTile);
        tile_unscale = unscale;
    }
    // four consecutive real results = two consecutive complex points of the packed frame
    __device__ __forceinline__ void emit(int q, f32x4 res)
    {
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned j = (wave * 1024 + q * 256 + 16 * (lane & 15) + 4 * (lane >> 4)) / 2;  // even: j and j+1 share a padding group
        img[pad(j)] = {res[0], res[1]};
        img[pad(j) + 1] = {res[2], res[3]};
    }
    // The transform is run_passes<float, 2048, 256, 8, 8, 8, 4> written out: with N and the
    // radices fixed, every LDS address of a pass is one lane-dependent base plus a constant
    // (pad(i + c) = pad(i) + 17 c / 16 for c a multiple of 16, and the pass-1 / pass-2 write
    // patterns by inspection), which the generic index arithmetic -- four vector instructions per
    // address, 260 per tile -- does not reveal to the compiler; and the twiddles of the two long
    // passes are powers of one table product instead of seven.
    __device__ __forceinline__ void end_tile()
    {
        unsigned tl = threadIdx.x;
        asm volatile("" : "+v"(tl));  // keeps the bases below out of the walker's loop-invariant registers
        const unsigned rb = tl + (tl >> 4);                       // pad(tl)
        const unsigned w1 = 8 * tl + (tl >> 1);                   // pad(8 tl)
        const unsigned j2 = ((tl >> 3) << 6) + (tl & 7), w2 = j2 + (j2 >> 4);
        const unsigned j3 = ((tl >> 6) << 9) + (tl & 63), w3 = j3 + (j3 >> 4);
        const unsigned pb = (256 - tl) + ((256 - tl) >> 4);       // pad(256 - tl)
        C<float> v[P];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = img[rb + 272 * r];    // points tl + 256 r
        __syncthreads();  // every lane holds its inputs before the first pass writes the image

        // pass 1: radix 8 over stride 256; butterfly tl writes 8 tl + r
        dft_reg<8, 1>(v);
#pragma unroll
        for (int r = 0; r < 8; r++) img[w1 + r] = v[r];
        __syncthreads();

        // pass 2: radix 8, 8 sub-transforms interleaved; twiddles W^(32 k r), k = tl mod 8, from their table
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = img[rb + 272 * r];
        __syncthreads();
        {
            const C<float> *t = tw1 + (tl & 7);
#pragma unroll
            for (int r = 1; r < 8; r++) v[r] = cmul(v[r], t[(r - 1) * 8]);
        }
        dft_reg<8, 1>(v);
#pragma unroll
        for (int r = 0; r < 8; r++) img[w2 + 8 * r + (r >> 1)] = v[r];  // pad(j2 + 8 r)
        __syncthreads();

        // pass 3: radix 8, 64 interleaved; twiddles W^(4 k r), k = tl mod 64, as powers of W^(4k)
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = img[rb + 272 * r];
        __syncthreads();
        if constexpr (kResident) {
#pragma unroll
            for (int r = 1; r < 8; r++) v[r] = cmul(v[r], rot3[r - 1]);
        } else {
            const C<float> *ta = tw2, *tb = tw2 + N / 64;          // W^(64 a), W^b
            const unsigned e = 4 * (tl & 63);
            const C<float> p1 = cmul(ta[e >> 6], tb[e & 63]);
            const C<float> p2 = cmul(p1, p1), p3 = cmul(p2, p1), p4 = cmul(p2, p2);
            v[1] = cmul(v[1], p1);
            v[2] = cmul(v[2], p2);
            v[3] = cmul(v[3], p3);
            v[4] = cmul(v[4], p4);
            v[5] = cmul(v[5], cmul(p4, p1));
            v[6] = cmul(v[6], cmul(p3, p3));
            v[7] = cmul(v[7], cmul(p4, p3));
        }
        dft_reg<8, 1>(v);
#pragma unroll
        for (int r = 0; r < 8; r++) img[w3 + 68 * r] = v[r];     // pad(j3 + 64 r)
        __syncthreads();

        // pass 4: radix 4, two butterflies per lane (j = tl, tl + 256); twiddles W^(j r)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[4 * b + r] = img[rb + 272 * b + 544 * r];  // points j + 512 r
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if constexpr (kResident) {
#pragma unroll
                for (int r = 1; r < 4; r++) v[4 * b + r] = cmul(v[4 * b + r], rot4[b][r - 1]);
            } else {
                const C<float> *ta = tw2, *tb = tw2 + N / 64;
                const unsigned e = tl + 256 * b;
                const C<float> p1 = cmul(ta[e >> 6], tb[e & 63]);
                const C<float> p2 = cmul(p1, p1);
                v[4 * b + 1] = cmul(v[4 * b + 1], p1);
                v[4 * b + 2] = cmul(v[4 * b + 2], p2);
                v[4 * b + 3] = cmul(v[4 * b + 3], cmul(p2, p1));
            }
            dft_reg<4, 1>(&v[4 * b]);
        }
        // park Z in natural order; Z[0] once more in the spare slot, where bin 0 looks for Z[N]
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) img[rb + 272 * b + 544 * r] = v[4 * b + r];
        if (tl == 0) img[pad(N)] = v[0];
        __syncthreads();

        // bins in pairs, as in fft_r2c_kernel: X[k] = E[k] + W^k O[k], X[N-k] = conj(E[k] - W^k O[k])
        constexpr int H = P / 2;
        float mlo[H], mhi[H];
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < H; i++) {
            const unsigned k = tl + NT * i;
            const C<float> a = img[rb + 272 * i];
            const C<float> b0 = img[pb + 1904 - 272 * i];         // pad(N - k); the spare slot for k = 0
            // twice E and twice D: the halving, like the tile's scale, goes into the gain below
            const C<float> e = {a.re + b0.re, a.im - b0.im};
            const C<float> d = {a.re - b0.re, a.im + b0.im};
            const C<float> o = {d.im, -d.re};
            const C<float> wo = cmul(tw2n[k], o);
            const C<float> xlo = {e.re + wo.re, e.im + wo.im}, xhi = {e.re - wo.re, wo.im - e.im};
            mlo[i] = __builtin_amdgcn_sqrtf(xlo.re * xlo.re + xlo.im * xlo.im);
            mhi[i] = __builtin_amdgcn_sqrtf(xhi.re * xhi.re + xhi.im * xhi.im);
            // bin N (the partner of bin 0) belongs to the spectrum only when nbin = N + 1
            part += mlo[i] + ((i == 0 && tl == 0 && nbin <= N) ? 0.f : mhi[i]);
        }
        const C<float> zmid = img[pad(N / 2)];
        const float mmid = 2.f * __builtin_amdgcn_sqrtf(zmid.re * zmid.re + zmid.im * zmid.im);
        if (tl == 0) part += mmid;
        float div = 1.f, mul = 1.f;
        if (mode == 2) {
            mul = 2.f;
            div = scale;
        }
        // the two barriers of the sum also keep the next tile's results out of the image until
        // every lane has read its bins
        // magnitudes so far: 2 / tile_unscale times the true ones
        const float back = 0.5f * tile_unscale;
        const float sum = frame_sum<NT>(part, scratch, 0, tl) * back;
        if (mode == 1 || (mode == 0 && sum > 0.00001f)) div = sum;
        const float gain = mul * back * __builtin_amdgcn_rcpf(div);
        const __amdgpu_buffer_rsrc_t dst = uniform_descriptor(ps + frame * nbin, (int)nbin * 4);
#pragma unroll
        for (int i = 0; i < H; i++) {
            const unsigned k = tl + NT * i;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mlo[i] * gain), dst, (int)k * 4, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mhi[i] * gain), dst, (int)(N - k) * 4, 0, 2);  // bin N: out of range when nbin = N
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mmid * gain), dst, tl == 0 ? (N / 2) * 4 : 0x7ffffff0, 0, 2);
    }
};

constexpr size_t kSpectrumImageBytes = padded_len(2048) * sizeof(C<float>);
constexpr size_t fir_spectrum_lds_bytes(int ns)
{
    return half_lds_bytes(ns) + kSpectrumImageBytes + lds_twiddle_entries(3, 8, 8, 8, 4, 2048) * sizeof(C<float>) + 4 * sizeof(float);
}

template <int NS>
__global__ __launch_bounds__(256) void fir_spectrum_kernel(const FirParams<float> p, const u32x4 *__restrict__ htab, float tap_unscale,
                                                           const C<float> *__restrict__ tw, const C<float> *__restrict__ tw2n,
                                                           float *ps, unsigned nbin, int mode, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr size_t kPlanes = half_lds_bytes(NS);
    SpectrumSink<NS> sink;
    sink.img = reinterpret_cast<C<float> *>(smem_raw + kPlanes);
    setup_lds_twiddles_split<float, 2048, 256, 1, 8, 8, kPlanes + kSpectrumImageBytes>(smem_raw, tw, sink.tw1, sink.tw2, sink.tw3);
    sink.tw = tw;
    sink.tw2n = tw2n;
    sink.scratch = reinterpret_cast<float *>(smem_raw + kPlanes + kSpectrumImageBytes +
                                             lds_twiddle_entries(3, 8, 8, 8, 4, 2048) * sizeof(C<float>));
    sink.ps = ps;
    sink.nbin = nbin;
    sink.frames_per_chan = p.tiles_per_chan;
    sink.mode = mode;
    sink.scale = scale;
    sink.frame = 0;
    sink.load_twiddles();
    fir_half_tiles<NS>(p, htab, tap_unscale, smem_raw, sink, MemorySource{});
}

// The same kernel with the tone generator in front (tonegen/tonegen.c:25-40 -> do_fir ->
// do_power_spectrum): the samples are computed into the walker's staging registers
// (ToneSource, fir_half.h) instead of being read, so a chunk of the chain costs the 2 bytes
// of spectrum per sample and nothing else in HBM.
template <int NS, int MAXT>
__global__ __launch_bounds__(256) void tone_fir_spectrum_kernel(const FirParams<float> p, const u32x4 *__restrict__ htab,
                                                                float tap_unscale, const C<float> *__restrict__ tw,
                                                                const C<float> *__restrict__ tw2n, float *ps, unsigned nbin,
                                                                int mode, float scale, const double *__restrict__ lane_tab,
                                                                const double *__restrict__ tile_tab,
                                                                const unsigned *__restrict__ hard, float *__restrict__ hist_out,
                                                                size_t hist_out_stride, double *__restrict__ w,
                                                                const double *__restrict__ dw, unsigned nseq)
{
    ToneSource<MAXT> src;
    src.lane_tab = lane_tab;
    src.tile_tab = tile_tab;
    src.hard = hard;
    src.hist_out = hist_out;
    src.hist_out_stride = hist_out_stride;
    // The tables were made from the running angles by the launch before this one, and nothing
    // here reads them: they move on by the row length now, every addition rounded as the
    // reference rounds it.
    for (unsigned s = blockIdx.x * 256 + threadIdx.x; s < nseq; s += gridDim.x * 256) w[s] = advance(w[s], dw[s], p.nx);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr size_t kPlanes = half_lds_bytes(NS);
    SpectrumSink<NS> sink;
    sink.img = reinterpret_cast<C<float> *>(smem_raw + kPlanes);
    setup_lds_twiddles_split<float, 2048, 256, 1, 8, 8, kPlanes + kSpectrumImageBytes>(smem_raw, tw, sink.tw1, sink.tw2, sink.tw3);
    sink.tw = tw;
    sink.tw2n = tw2n;
    sink.scratch = reinterpret_cast<float *>(smem_raw + kPlanes + kSpectrumImageBytes +
                                             lds_twiddle_entries(3, 8, 8, 8, 4, 2048) * sizeof(C<float>));
    sink.ps = ps;
    sink.nbin = nbin;
    sink.frames_per_chan = p.tiles_per_chan;
    sink.mode = mode;
    sink.scale = scale;
    sink.frame = 0;
    sink.load_twiddles();
    fir_half_tiles<NS>(p, htab, tap_unscale, smem_raw, sink, src);
}

template <int NS, int MAXT>
int launch_fir_spectrum_t(const FirParams<float> &p, const dspmi::FirSpectrumLaunch &a, unsigned blocks, hipStream_t s)
{
    static dspmi::PerDeviceOnce attr;
    if constexpr (MAXT == 0) {
        auto kern = fir_spectrum_kernel<NS>;
        if (attr.pending()) {
            DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr.mark();
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), fir_spectrum_lds_bytes(NS), s, p, static_cast<const u32x4 *>(a.tap_images),
                           a.tap_unscale, static_cast<const C<float> *>(a.tw_half), static_cast<const C<float> *>(a.tw_full), a.ps,
                           a.nbin, a.mode, a.scale);
    } else {
        auto kern = tone_fir_spectrum_kernel<NS, MAXT>;
        if (attr.pending()) {
            DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr.mark();
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), fir_spectrum_lds_bytes(NS), s, p, static_cast<const u32x4 *>(a.tap_images),
                           a.tap_unscale, static_cast<const C<float> *>(a.tw_half), static_cast<const C<float> *>(a.tw_full), a.ps,
                           a.nbin, a.mode, a.scale, a.lane_tab, a.tile_tab, a.hard, a.hist_out, a.hist_out_stride, a.w, a.dw, a.nseq);
    }
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <int NS>
int launch_fir_spectrum_n(const FirParams<float> &p, const dspmi::FirSpectrumLaunch &a, unsigned blocks, hipStream_t s)
{
    switch (a.tone_slots) {
    case 0: return launch_fir_spectrum_t<NS, 0>(p, a, blocks, s);
    case 2: return launch_fir_spectrum_t<NS, 2>(p, a, blocks, s);
    case 4: return launch_fir_spectrum_t<NS, 4>(p, a, blocks, s);
    }
    return dspmi::fail(DSPMI_EINVAL, "fir_spectrum: no kernel for %d tone slots", a.tone_slots);
}

// ---------------------------------------------------------------------------
// real output: bins 0..N of a 2N-point real frame back to the frame
// ---------------------------------------------------------------------------

// The fftw_plan_dft_c2r_1d contract (util/filter/filter.c:98-103): unnormalised, the imaginary
// parts of bins 0 and N ignored.  The 2N reals, read as N complex points, are the N-point
// backward transform of
//   Z[m] = (X[m] + conj X[N-m]) + i W^-m (X[m] - conj X[N-m]),   W = exp(-2 pi i / 2N).
// A lane loads the bins its first pass wants (the c2c load pattern), the frame's LDS image
// supplies each one's partner X[N-m] (bin N sits in the image's spare slot), and Z goes
// straight into the passes: the merge costs no trip through HBM.
template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
__global__ __launch_bounds__(block_threads<NT>()) void fft_c2r_kernel(const C<T> *in, C<T> *out,
                                                                      const C<T> *__restrict__ tw,
                                                                      const C<T> *__restrict__ tw2n, size_t batch)
{
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    constexpr int P = N / NT;
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    constexpr int RLAST = NPASS == 1 ? R0 : NPASS == 2 ? R1 : NPASS == 3 ? R2 : R3;
    constexpr int NSLAST = N / RLAST;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned slot = threadIdx.x / NT;
    const unsigned tl = threadIdx.x % NT;
    C<T> *lds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)slot * padded_len(N);
    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    if constexpr (TW == 2) setup_lds_twiddles<T, N, BT, FPB, R0, R1, R2, R3>(smem_raw, tw, tw1, tw2, tw3);
    if constexpr (TW == 3) setup_lds_twiddles_split<T, N, BT, FPB, R0, R1>(smem_raw, tw, tw1, tw2, tw3);

    const size_t nfb = (batch + FPB - 1) / FPB;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * FPB + slot;
        const bool active = frame < batch;
        const C<T> *src = in + frame * (N + 1);
        // laundered lane index: keeps the twiddle loads and partner addresses per-frame instead
        // of hoisted out of the frame loop into registers that would live through the passes
        unsigned te = tl;
        asm volatile("" : "+v"(te));
        C<T> v[P], w[P];
#pragma unroll
        for (int b = 0; b < P / R0; b++)
#pragma unroll
            for (int r = 0; r < R0; r++) {
                const unsigned m = te + NT * b + r * (N / R0);
                C<T> q = {T(0), T(0)};
                if (active) q = nt_load(&src[m]);
                v[b * R0 + r] = q;
            }
        // the merge twiddles are in flight together with the bins
#pragma unroll
        for (int i = 0; i < P; i++) w[i] = tw2n[te + NT * (i / R0) + (i % R0) * (N / R0)];
#pragma unroll
        for (int i = 0; i < P; i++) lds[pad(te + NT * (i / R0) + (i % R0) * (N / R0))] = v[i];
        if (tl == 0) lds[pad(N)] = active ? src[N] : C<T>{T(0), T(0)};
        frame_sync<NT>();
#pragma unroll
        for (int i = 0; i < P; i++) {
            const unsigned m = te + NT * (i / R0) + (i % R0) * (N / R0);
            C<T> a = v[i];
            C<T> b0 = lds[pad(N - m)];
            if (i == 0 && te == 0) a.im = b0.im = T(0);  // m == 0: bins 0 and N
            const C<T> sm = {a.re + b0.re, a.im - b0.im};
            const C<T> d = {a.re - b0.re, a.im + b0.im};
            const C<T> wd = cmul(C<T>{w[i].re, -w[i].im}, d);
            // Z = sm + i*wd, conjugated for the backward transform (conj . forward . conj)
            v[i] = {sm.re - wd.im, -(sm.im + wd.re)};
        }
        frame_sync<NT>();  // partners read before the passes overwrite the image
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));
        if (active) {
            C<T> *dst = out + frame * N;
#pragma unroll
            for (int b = 0; b < P / RLAST; b++)
#pragma unroll
                for (int r = 0; r < RLAST; r++) {
                    const C<T> q = v[b * RLAST + r];
                    nt_store(C<T>{q.re, -q.im}, &dst[tl + NT * b + r * NSLAST]);
                }
        }
    }
}

// ---------------------------------------------------------------------------
// overlap-save: load two real blocks, transform, multiply by H, transform back, keep
// the part that does not wrap -- one kernel, the block never leaves the CU
// ---------------------------------------------------------------------------

struct OlsArgs {
    const float *x, *hist;
    float *y;
    const C<float> *H;  // FFT_N(taps) / N
    size_t x_stride, y_stride, hist_stride;
    unsigned nx, nh, L, nblk, npair, nchan;
    unsigned stagger = 0;  // ols_wave_kernel: spread of the workgroups' start times, in clock ticks
    unsigned stagger_all = 0;  // ... applied to every workgroup (many rounds), not only to those with a round less
    unsigned prefetch = 0;     // ols_wave_kernel: the next block pair's input is requested ahead of this pair's stores
};

// The radix list is symmetric (R0 == last radix), so the positions a lane holds after the
// forward transform's last pass (j + r*N/R) are exactly the ones the backward transform's
// first pass wants: the spectrum is multiplied in registers and goes straight back in.
template <int N, int NT, int R0, int R1, int R2, int R3, int TW>
__global__ __launch_bounds__((ols_block_threads<N, NT>())) __attribute__((amdgpu_waves_per_eu((N / NT > 16) ? 2 : (N == 1024 ? 4 : 3), 8))) void ols_fused_kernel(const OlsArgs a, const C<float> *__restrict__ tw)
{
    typedef float T;
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    constexpr int P = N / NT;
    constexpr int BT = ols_block_threads<N, NT>();
    constexpr int FPB = BT / NT;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    constexpr int RLAST = NPASS == 1 ? R0 : NPASS == 2 ? R1 : NPASS == 3 ? R2 : R3;
    static_assert(RLAST == R0, "first and last radix must match");
    constexpr int STEP = N / R0;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned slot = threadIdx.x / NT;
    const unsigned tl = threadIdx.x % NT;
    C<T> *lds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)slot * padded_len(N);
    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    if constexpr (TW == 2) setup_lds_twiddles<T, N, BT, FPB, R0, R1, R2, R3>(smem_raw, tw, tw1, tw2, tw3);
    if constexpr (TW == 3) setup_lds_twiddles_split<T, N, BT, FPB, R0, R1>(smem_raw, tw, tw1, tw2, tw3);

    const long nx = a.nx, nh = a.nh, L = a.L;
    const size_t total = (size_t)a.nchan * a.npair;
    const size_t nfb = (total + FPB - 1) / FPB;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * FPB + slot;
        const bool active = frame < total;
        const size_t c = active ? frame / a.npair : 0;
        const unsigned p = active ? (unsigned)(frame - c * a.npair) : 0;
        const float *xrow = a.x + c * a.x_stride;
        const float *hrow = a.hist ? a.hist + c * a.hist_stride : nullptr;
        const long base = (long)(2 * p) * L - (nh - 1);  // sample index of element 0 of the even block
        const bool second = 2 * p + 1 < a.nblk;
        // the edge path synchronises, so the choice has to be the same for every lane that
        // takes part in that synchronisation: the wave (frames within one wave) or the workgroup
        const bool mine = active && base >= 0 && base + L + N <= nx && second;
        bool interior;
        if constexpr (NT > 64) interior = __syncthreads_and(mine);
        else interior = __all(mine);

        // element m = tl + NT*b + r*STEP of the pair: re = x[base + m], im = x[base + L + m].
        // One per-lane pointer per block of the pair; every other address is a compile-time
        // offset from it.
        C<T> v[P];
        if (interior) {
            const float *p0 = xrow + base + tl;
            const float *p1 = p0 + L;
#pragma unroll
            for (int b = 0; b < P / R0; b++)
#pragma unroll
                for (int r = 0; r < R0; r++) {
                    v[b * R0 + r].re = __builtin_nontemporal_load(p0 + (NT * b + r * STEP));
                    v[b * R0 + r].im = __builtin_nontemporal_load(p1 + (NT * b + r * STEP));
                }
        } else {
            // edge of the signal (or an idle frame): a plain checked loop fills the frame's LDS
            // image, so the unrolled code above stays free of per-element conditions
#pragma unroll 1
            for (int m = tl; m < N; m += NT) {
                const long s = base + m;
                C<T> q = {0.f, 0.f};
                if (active) {
                    if (s >= 0) q.re = s < nx ? xrow[s] : 0.f;
                    else if (hrow && s >= 1 - nh) q.re = hrow[nh - 1 + s];
                    const long s2 = s + L;
                    if (second) {
                        if (s2 >= 0) q.im = s2 < nx ? xrow[s2] : 0.f;
                        else if (hrow && s2 >= 1 - nh) q.im = hrow[nh - 1 + s2];
                    }
                }
                lds[pad(m)] = q;
            }
            frame_sync<NT>();
#pragma unroll
            for (int b = 0; b < P / R0; b++)
#pragma unroll
                for (int r = 0; r < R0; r++) v[b * R0 + r] = lds[pad(tl + NT * b + r * STEP)];
            frame_sync<NT>();
        }

        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));
        // spectrum times H, then the backward transform as swap . forward . swap (swap exchanges
        // the real and imaginary parts: swap(z) = i conj(z), so forward(swap(Y)) = i conj(backward(Y))
        // = swap(backward(Y)); unlike conj . forward . conj it costs no instruction, only the names
        // of two registers)
        {
            // laundered lane index: H[...] is the same for every frame, and hoisting its 2*P
            // values out of the frame loop would cost their registers through both transforms
            unsigned th = tl;
            asm volatile("" : "+v"(th));
            const C<T> *hp = a.H + th;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const C<T> hq = hp[NT * (i / R0) + (i % R0) * STEP];
                const C<T> m = cmul(v[i], hq);
                v[i] = {m.im, m.re};
            }
        }
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));

        // outputs j = m - (nh-1) >= 0 of each block of the pair
        if (interior) {
            float *q0 = a.y + c * a.y_stride + ((long)(2 * p) * L - (nh - 1)) + tl;
            float *q1 = q0 + L;
            const int skip = (int)nh - 1 - (int)tl;  // element offsets below this are overlap
#pragma unroll
            for (int i = 0; i < P; i++) {
                const int off = NT * (i / R0) + (i % R0) * STEP;
                if (off >= skip) {
                    __builtin_nontemporal_store(v[i].im, q0 + off);  // swapped back: im holds the even block
                    __builtin_nontemporal_store(v[i].re, q1 + off);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < P; i++) lds[pad(tl + NT * (i / R0) + (i % R0) * STEP)] = v[i];
            frame_sync<NT>();
            if (active) {
                float *yrow = a.y + c * a.y_stride;
#pragma unroll 1
                for (int m = tl; m < N; m += NT) {
                    const long j = m - (nh - 1);
                    if (j < 0) continue;
                    const C<T> q = lds[pad(m)];
                    const long o0 = (long)(2 * p) * L + j;
                    if (o0 < nx) yrow[o0] = q.im;
                    if (second && o0 + L < nx) yrow[o0 + L] = q.re;
                }
            }
            frame_sync<NT>();
        }
    }
}

// ---------------------------------------------------------------------------
// overlap-save, long blocks: a radix-K1 stage across the workgroup, then one 1024-point
// transform per wave
// ---------------------------------------------------------------------------

// N = K1 * 1024 points per block pair (K1 = 32: 32768).  With n = n2 + 1024 n1 and
// k = k1 + K1 k2 the transform factors as
//   X[k1 + K1 k2] = sum_n2 W_1024^(n2 k2) . W_N^(n2 k1) . sum_n1 x[n2 + 1024 n1] W_K1^(n1 k1):
// lane n2 of a 1024-lane workgroup holds x[n2 + 1024 n1] for every n1 (coalesced loads) and does
// the radix-K1 sum and the W_N^(n2 k1) factors in registers; one exchange across the workgroup
// then hands wave w the K1/16 rows k1 it owns, and each row's 1024-point transform runs inside
// the wave (passes exchanged through the wave's own LDS region, no workgroup barrier).  The
// spectrum is multiplied by H there, and the second transform is the mirror image (the same
// factorisation read the other way: rows first, then W_N^(k1 n2) and the radix-K1 sum), so it
// ends with lane n2 holding y[n2 + 1024 n1] -- the layout it was loaded in.
// A block costs two workgroup-wide exchanges instead of six, and between them the sixteen waves
// run independently, so one wave's LDS traffic overlaps another's butterflies.  The frame
// (256 KB as complex points) exceeds the LDS: the workgroup-wide exchanges move the real parts,
// then the imaginary parts, through one N-float image; the waves' regions alias it.
// The stage factors W^(k1 n2), k1 = 1 .. K1-1, of one lane.  Only the rows k1 = 1, 2, 3 and
// 4, 8, .. are loaded (ten values instead of thirty-one, so they can be requested at the top of
// the frame and sit in registers until the radix-K1 sum is done); the others are applied as
// two factors, W^(4a n2) and W^(b n2): one more rounding on those points (the block error stays
// at 2.1e-7).
template <int K1, int NS = 1024>
struct StageFactors {
    C<float> lo[3], hi[K1 / 4 - 1];
    __device__ __forceinline__ void load(const C<float> *tp)
    {
#pragma unroll
        for (int b = 1; b < 4; b++) lo[b - 1] = tp[NS * b];
#pragma unroll
        for (int a = 1; a < K1 / 4; a++) hi[a - 1] = tp[NS * 4 * a];
    }
    __device__ __forceinline__ void apply(C<float> *v) const
    {
#pragma unroll
        for (int k1 = 1; k1 < K1; k1++) {
            if (k1 / 4) v[k1] = cmul(v[k1], hi[k1 / 4 - 1]);
            if (k1 % 4) v[k1] = cmul(v[k1], lo[k1 % 4 - 1]);
        }
    }
};

template <int K1>
__global__ __launch_bounds__(1024) void ols_wave_kernel(const OlsArgs a, const C<float> *__restrict__ tw_sub,
                                                        const C<float> *__restrict__ tw_stage)
{
    typedef float T;
    constexpr int NS = 1024, NW = 16, SPW = K1 / NW;  // points per row, waves, rows per wave
    constexpr int R0 = 16, R1 = 4, R2 = 16;           // the row transform: symmetric radix list
    static_assert(K1 % NW == 0 && K1 <= 32, "rows are dealt out evenly; radix up to 32");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned tl = threadIdx.x, lane = tl & 63, wave = tl >> 6;
    float *img = reinterpret_cast<float *>(smem_raw);  // [K1][1024], one component at a time
    C<T> *wreg = reinterpret_cast<C<T> *>(smem_raw) + (size_t)wave * padded_len(NS);
    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    setup_lds_twiddles<T, NS, 1024, NW, R0, R1, R2, 1>(smem_raw, tw_sub, tw1, tw2, tw3);

    const long nx = a.nx, nh = a.nh, L = a.L;
    const size_t total = (size_t)a.nchan * a.npair;
    if (a.stagger) {
        // Workgroups that start together stay in step: every CU stores its 192 KB and loads its
        // 256 KB at the same moment -- a burst that HBM takes ~20 us to move while every CU waits --
        // and then every CU computes for 30 us while HBM idles (measured with per-phase
        // timestamps: 12 us from the top of a block pair until its input has arrived).  Spreading
        // the start times over about a period keeps the phases apart for the first several pairs
        // (they drift back into step after about ten).  With few rounds, the workgroups that stride
        // over one pair more than the others start at once and the others share the later
        // starts, which costs nothing at the end (their last round is the one they do not have);
        // from eight rounds on every workgroup takes part (the shorter period repays the late start:
        // 2736 pairs 0.562 -> 0.534 ms, 2560 pairs 0.515 -> 0.489 ms).
        const unsigned extra = (unsigned)(total % gridDim.x);
        unsigned long long wait;
        if (a.stagger_all)  // many rounds: the shorter period repays a late start for everyone
            wait = (unsigned long long)a.stagger * blockIdx.x / gridDim.x;
        else
            wait = blockIdx.x < extra ? 0 : (unsigned long long)a.stagger * (blockIdx.x - extra + (extra ? 1 : 0)) / (gridDim.x - extra);
        const unsigned long long t0 = __builtin_readcyclecounter();
        // bounded by the clock, and by a count of sleeps in case the counter ticks at another rate
        for (unsigned naps = 0; naps < 64 && __builtin_readcyclecounter() - t0 < wait; naps++) __builtin_amdgcn_s_sleep(32);
    }
    // Where a block pair lies: channel, pair within the channel, sample index of element 0 of the
    // even block, whether the odd block exists, and whether every sample of both blocks does
    // (the same answers for the whole workgroup).
    struct Place {
        size_t c;
        unsigned p;
        long base;
        bool second, interior;
    };
    auto locate = [&](size_t frame) {
        Place f;
        f.c = frame / a.npair;
        f.p = (unsigned)(frame - f.c * a.npair);
        f.base = (long)(2 * f.p) * L - (nh - 1);
        f.second = 2 * f.p + 1 < a.nblk;
        f.interior = f.base >= 0 && f.base + L + K1 * NS <= nx && f.second;
        return f;
    };
    // A pair's input requested while the pair before it is still being finished (a.prefetch,
    // interior pairs only): the even block goes straight into the exchange image, which is idle
    // from the last exchange of one pair to the first of the next (buffer loads with the LDS
    // bit: no registers), the odd block into `ahead` once the last radix-K1 sum has freed the
    // registers of the row phase -- both ahead of the pair's stores.  The wait for a pair's input
    // is then a wait for operations older than those stores (the counter is in order); issued the
    // other way round, every load waits behind 192 KB of stores per CU.  The stores themselves
    // need not have completed until the stage factors, loaded at the top of the pair, are used
    // after its first radix-K1 sum.  Measured on 2^26 samples, 8191 taps, alternating on one box:
    // 0.279 -> 0.265 ms, 0.272 -> 0.266, 0.2685 -> 0.264.  (What-if builds: with the stores left out
    // the launch takes 0.221 ms, with loads that hit the cache 0.242.  Requesting the stage
    // factors ahead of the stores as well, or keeping them in registers across the pair boundary,
    // costs more in registers than it hides: 0.272 - 0.283 ms.)
    bool ready = false;
    float ahead[K1];
    for (size_t frame = blockIdx.x; frame < total; frame += gridDim.x) {
        const Place here = locate(frame);
        const size_t c = here.c;
        const unsigned p = here.p;
        const float *xrow = a.x + c * a.x_stride;
        const float *hrow = a.hist ? a.hist + c * a.hist_stride : nullptr;
        const long base = here.base;
        const bool second = here.second;
        const bool interior = here.interior;
        // laundered lane index: the stage factors and H are the same for every frame; hoisted out
        // of the frame loop they would occupy registers through both transforms
        unsigned tt = tl;
        asm volatile("" : "+v"(tt));

        // element tl + 1024 n1 of the pair: re = x[base + ...], im = x[base + L + ...]
        C<T> v[K1];
        if (ready) {
            // Everything but the youngest 48 memory operations -- the stores that followed the
            // requests -- has completed: vmcnt(48).  Stated here, and the image read back in asm:
            // before an LDS read the compiler can see it waits for every memory operation once
            // a load into LDS is outstanding, stores included.  Each wave reads back the columns
            // its own lanes requested (lane address + 4096 n1, two bases: the offset field ends at 65535).
            __builtin_amdgcn_s_waitcnt(0xcf70);
            const unsigned a0 = tl * 4u, a1 = tl * 4u + 65536u;
#define DSPMI_READ16(base, v0)                                                                                         \
    asm volatile("ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:4096\n ds_read_b32 %2, %16 offset:8192\n"           \
                 "ds_read_b32 %3, %16 offset:12288\n ds_read_b32 %4, %16 offset:16384\n ds_read_b32 %5, %16 offset:20480\n" \
                 "ds_read_b32 %6, %16 offset:24576\n ds_read_b32 %7, %16 offset:28672\n ds_read_b32 %8, %16 offset:32768\n" \
                 "ds_read_b32 %9, %16 offset:36864\n ds_read_b32 %10, %16 offset:40960\n ds_read_b32 %11, %16 offset:45056\n" \
                 "ds_read_b32 %12, %16 offset:49152\n ds_read_b32 %13, %16 offset:53248\n ds_read_b32 %14, %16 offset:57344\n" \
                 "ds_read_b32 %15, %16 offset:61440\n s_waitcnt lgkmcnt(0)"                                             \
                 : "=&v"(v[v0].re), "=&v"(v[v0 + 1].re), "=&v"(v[v0 + 2].re), "=&v"(v[v0 + 3].re), "=&v"(v[v0 + 4].re),   \
                   "=&v"(v[v0 + 5].re), "=&v"(v[v0 + 6].re), "=&v"(v[v0 + 7].re), "=&v"(v[v0 + 8].re), "=&v"(v[v0 + 9].re), \
                   "=&v"(v[v0 + 10].re), "=&v"(v[v0 + 11].re), "=&v"(v[v0 + 12].re), "=&v"(v[v0 + 13].re),                \
                   "=&v"(v[v0 + 14].re), "=&v"(v[v0 + 15].re)                                                           \
                 : "v"(base)                                                                                           \
                 : "memory")
            static_assert(K1 == 32 && NS == 1024, "two groups of sixteen rows, 4096 bytes apart");
            DSPMI_READ16(a0, 0);
            DSPMI_READ16(a1, 16);
#undef DSPMI_READ16
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                v[n1].im = ahead[n1];
                asm volatile("" : "+v"(v[n1].im));  // the wait for it belongs here, not where the three paths meet
            }
        } else if (interior) {
            const float *p0 = xrow + base + tt;
            const float *p1 = p0 + L;
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                v[n1].re = __builtin_nontemporal_load(p0 + NS * n1);
                v[n1].im = __builtin_nontemporal_load(p1 + NS * n1);
            }
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) asm volatile("" : "+v"(v[n1].re), "+v"(v[n1].im));  // (the wait for these, likewise)
        } else {
            auto sample = [&](long s) -> float {
                if (s >= 0) return s < nx ? xrow[s] : 0.f;
                return (hrow && s >= 1 - nh) ? hrow[nh - 1 + s] : 0.f;
            };
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                const long s = base + tt + NS * n1;
                v[n1].re = sample(s);
                v[n1].im = second ? sample(s + L) : 0.f;
            }
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) asm volatile("" : "+v"(v[n1].re), "+v"(v[n1].im));
        }

        // ---- first transform: radix-K1 sum, stage factors, rows ----
        StageFactors<K1> sf;
        sf.load(tw_stage + tt);
        dft_reg<K1, 1>(v);
        sf.apply(v);
        C<T> u[SPW][16];
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
#pragma unroll
            for (int k1 = 0; k1 < K1; k1++) img[NS * k1 + tl] = comp ? v[k1].im : v[k1].re;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SPW; s++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float q = img[NS * (wave * SPW + s) + lane + 64 * r];
                    if (comp) u[s][r].im = q;
                    else u[s][r].re = q;
                }
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < SPW; s++) {
            // laundered lane index, once per use: every LDS and table address below is the same
            // for every row and frame, and would otherwise be computed once and kept (spilled)
            unsigned la = lane, lh = lane, lb = lane;
            asm volatile("" : "+v"(la));
            row_passes_1024(u[s], wreg, tw1, tw2, la);
            // spectrum times H (stored row by row: H[k1 + K1 k2] at [k1][k2]), real and imaginary
            // parts swapped for the second transform (backward = swap . forward . swap).
            // Requesting H before the transform would hide its latency but does not fit the 128
            // registers next to the other row.
            asm volatile("" : "+v"(lh));
            const C<T> *hp = a.H + (size_t)NS * (wave * SPW + s) + lh;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const C<T> m = cmul(u[s][r], hp[64 * r]);
                u[s][r] = {m.im, m.re};
            }
            // ---- second transform: rows, stage factors, radix-K1 sum ----
            asm volatile("" : "+v"(lb));
            row_passes_1024(u[s], wreg, tw1, tw2, lb);
        }
        {
            unsigned t2 = tl;
            asm volatile("" : "+v"(t2));
            sf.load(tw_stage + t2);  // on their way during the exchange
        }
        __syncthreads();  // the waves' regions alias the image: every wave is done with its rows
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
#pragma unroll
            for (int s = 0; s < SPW; s++)
#pragma unroll
                for (int r = 0; r < 16; r++) img[NS * (wave * SPW + s) + lane + 64 * r] = comp ? u[s][r].im : u[s][r].re;
            __syncthreads();
#pragma unroll
            for (int k1 = 0; k1 < K1; k1++) {
                const float q = img[NS * k1 + tl];
                if (comp) v[k1].im = q;
                else v[k1].re = q;
            }
            __syncthreads();
        }
        sf.apply(v);  // (ahead of the requests below: its wait for the stage factors would be a wait for them too)
        // The image is idle from here on: the next pair's even block lands in it.
        const size_t after = frame + gridDim.x;
        Place next{};
        if (a.prefetch && interior && after < total) next = locate(after);
        const bool fetch = next.interior;  // false unless located
        const float *nrow = a.x + next.c * a.x_stride + next.base;
        if (fetch) {
            // Buffer loads with the LDS bit (a uniform descriptor over the pair's input, the lane's
            // offset in one register, the row's in a scalar); the global_load_lds form would do as
            // well, but the compiler then treats every later wait as "wait for everything".  The
            // lane offset is tied to the last of the stage-factor products, so that the wait for
            // the factors comes before these requests, not after them.
            unsigned td = tl * 4u;
            asm volatile("" : "+v"(td) : "v"(v[K1 - 1].re), "v"(v[K1 - 1].im));
            const int nbytes = (int)((L + K1 * NS) * 4);
            const unsigned col = 64u * (unsigned)__builtin_amdgcn_readfirstlane((int)wave);
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(uniform_descriptor(nrow, nbytes),
                                                     (__attribute__((address_space(3))) void *)(img + NS * n1 + col), 4, (int)td,
                                                     NS * n1 * 4, 0, 0);
        }
        dft_reg<K1, 1>(v);

        // outputs j = m - (nh-1) >= 0 of each block of the pair; the result comes out swapped:
        // im belongs to the even block, re to the odd one
        unsigned to = tl;
        asm volatile("" : "+v"(to) : "v"(v[0].re));  // keeps what follows from being formed (and kept) early
        if (interior) {
            // Stores through a descriptor over the pair's outputs (built at each use: a descriptor
            // held in a variable that a lambda captures is kept in scratch memory): a lane whose
            // element is overlap gets an offset that is out of range for any size.  The rows that
            // can be overlap for every lane and are then skipped (up to 8192 taps: the first
            // eight) go first; the others are all issued, after the loads of `ahead`: a store under
            // a condition there would make the wait for `ahead` a wait for everything.
            const float *ybase = a.y + c * a.y_stride + ((long)(2 * p) * L - (nh - 1));
            const int ybytes = (int)((L + K1 * NS) * 4);
#define DSPMI_YRSRC uniform_descriptor(ybase, ybytes)
            const int skip = (int)nh - 1 - (int)to;  // element offsets below this are overlap
            auto store_row = [&](int n1) __attribute__((always_inline)) {
                const int off = NS * n1 >= skip ? (int)(to + NS * n1) * 4 : 0x7ffffff0;
                const int off1 = NS * n1 >= skip ? off + (int)L * 4 : 0x7ffffff0;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[n1].im), DSPMI_YRSRC, off, 0, 2);  // nontemporal
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[n1].re), DSPMI_YRSRC, off1, 0, 2);
                __builtin_amdgcn_sched_barrier(0);  // offsets formed row by row, not all at once
            };
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++)
                if (NS * n1 + NS > (int)nh - 1) store_row(n1);
            if (fetch) {
                const char *g = reinterpret_cast<const char *>(nrow + L);
                const unsigned lo = to * 4u;
#pragma unroll
                for (int n1 = 0; n1 < K1; n1++)
                    ahead[n1] = __builtin_nontemporal_load(reinterpret_cast<const float *>(g + (size_t)NS * n1 * 4 + (size_t)lo));
            } else {
                // (a value on every path: left alone, last round's values would stay live -- 32
                // registers -- through the whole of the next pair)
#pragma unroll
                for (int n1 = 0; n1 < K1; n1++) ahead[n1] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n1 = 8; n1 < 16; n1++) store_row(n1);
            // from row 16 on nothing is overlap (the transform is at least twice the tap count):
            // one lane offset per block, the row's offset in a scalar
            static_assert(K1 == 32, "rows 16 .. K1-1 are past any overlap");
            const int lane_off = (int)to * 4, lane_off1 = lane_off + (int)L * 4;
#pragma unroll
            for (int n1 = 16; n1 < K1; n1++) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[n1].im), DSPMI_YRSRC, lane_off, NS * n1 * 4, 2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[n1].re), DSPMI_YRSRC, lane_off1, NS * n1 * 4, 2);
            }
#undef DSPMI_YRSRC
            ready = fetch;
        } else {
            ready = false;
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) ahead[n1] = 0.f;
            float *yrow = a.y + c * a.y_stride;
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                const long j = (long)to + NS * n1 - (nh - 1);
                const long o0 = (long)(2 * p) * L + j;
                if (j >= 0 && o0 < nx) yrow[o0] = v[n1].im;
                if (j >= 0 && second && o0 + L < nx) yrow[o0 + L] = v[n1].re;
            }
            // vmcnt(0): nothing is outstanding where this path joins the other, so the compiler's
            // count of what may still be in flight at the top of the loop is the other path's
            // (these stores are conditional, and "perhaps none" would turn into "wait for all")
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
    }
}

template <int K1>
int launch_ols_wave(const dspmi::OlsLaunch &o, const void *tables, int num_cu, hipStream_t s)
{
    constexpr int NS = 1024, N = K1 * NS;
    constexpr size_t img = (size_t)N * sizeof(float), regions = (size_t)16 * padded_len(NS) * sizeof(C<float>);
    constexpr size_t lds = (img > regions ? img : regions) + lds_twiddle_entries(2, 16, 4, 16, 1) * sizeof(C<float>);
    static_assert(img <= regions, "the row twiddle tables sit behind the waves' regions");
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    auto kern = ols_wave_kernel<K1>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    OlsArgs a;
    a.x = o.x; a.hist = o.hist; a.y = o.y; a.H = static_cast<const C<float> *>(o.H);
    a.x_stride = o.x_stride; a.y_stride = o.y_stride; a.hist_stride = o.hist_stride;
    a.nx = o.nx; a.nh = o.nh; a.L = N - (o.nh - 1);
    a.nblk = (unsigned)(((unsigned long long)o.nx + a.L - 1) / a.L);
    a.npair = (a.nblk + 1) / 2;
    a.nchan = o.nchan;
    const size_t total = (size_t)a.nchan * a.npair;
    if (total == 0) return DSPMI_OK;
    // one workgroup fits a CU: a resident set that strides over the block pairs (0.286 ms against
    // 0.294 with one workgroup per pair on 2^26 samples -- the twiddle tables are set up once)
    size_t grid = total < (size_t)num_cu ? total : (size_t)num_cu;
    if (const char *g = getenv("DSPMI_OLS_GRID"))  // tests: few workgroups, so that short signals make each walk several pairs
        if (atoi(g) > 0 && (size_t)atoi(g) < grid) grid = (size_t)atoi(g);
    {
        // start-time spread: 60000 ticks (~25 us, about 0.6 of a block pair's time) measured best
        // on 2^26 samples (0.289 -> 0.258 ms); with every workgroup striding over the same number
        // of pairs the spread is paid at the end, which it earns back from four rounds on
        // (1280 pairs: 0.259 -> 0.236 ms).  DSPMI_OLS_STAGGER=<ticks> overrides, 0 turns it off.
        const size_t rounds = total / grid;
        const char *env = getenv("DSPMI_OLS_STAGGER");
        a.stagger = env ? (unsigned)atoi(env) : 60000u;
        if (rounds < 2 || (total % grid == 0 && rounds < 4)) a.stagger = 0;
        a.stagger_all = rounds >= 8;  // 5.33 rounds: 0.265-0.273 ms with only the shorter walks delayed, 0.270-0.280 with all
        if (const char *all = getenv("DSPMI_OLS_STAGGER_ALL")) a.stagger_all = (unsigned)atoi(all);  // experiment knob
        const char *pf = getenv("DSPMI_OLS_PREFETCH");  // 0 turns the early requests off (A/B)
        a.prefetch = pf ? (unsigned)atoi(pf) : 1u;
    }
    // tables: exp(-2 pi i k / 1024), k < 1024, then the stage factors exp(-2 pi i k1 n2 / N) as [k1][n2]
    const C<float> *tw_sub = static_cast<const C<float> *>(tables);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), lds, s, a, tw_sub, tw_sub + NS);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

// ---------------------------------------------------------------------------
// overlap-save, long blocks, one real block per transform: 32768 samples as 16384 complex
// points, two workgroups of 512 lanes per CU
// ---------------------------------------------------------------------------

// ols_wave_kernel fits one workgroup on a CU, and everything between its row phases is done
// by all sixteen waves in step: while they exchange through LDS, wait at a barrier or wait for
// memory, nothing else is there to run.  Here a block of M = 2 N2 real samples rides one N2-point
// complex transform as z[n] = x[2n] + i x[2n+1] (N2 = K1 * 512 = 16384), which halves the frame:
// 32 points per lane of a 512-lane workgroup, a 64 KB exchange image, two workgroups per CU that
// run out of step and fill each other's gaps.  The factorisation is ols_wave_kernel's
// (n = n2 + 512 n1, k = k1 + K1 k2; radix-K1 sum and stage factors in registers, one exchange,
// 512-point rows inside a wave).
// What replaces the multiplication by H: with h split into its even and odd samples,
// Hev = FFT_N2(h[0::2]), Hod = FFT_N2(h[1::2]), the packed spectrum of the filtered block is
//   Zy[k] = Hev[k] Z[k] + (i/2) Hod[k] (S - w D),   S = Z[k] + conj Z[N2-k], D = Z[k] - conj Z[N2-k],
//   w = exp(-2 pi i k / N2)
// (S/2 and -iD/2 are the spectra of the even and odd input samples; the odd taps delay the odd
// samples by one even sample: the factor w), and for the partner bin
//   Zy[N2-k] = Hev[N2-k] Z[N2-k] + (i/2) Hod[N2-k] conj(S + w D).
// Bin k = k1 + K1 k2 and its partner N2-k sit in rows k1 and (K1 - k1) mod K1, at positions k2 and
// 511 - k2 (row 0: 512 - k2): a wave is given both rows of a pair -- (1,31), (2,30) .. (15,17), and
// rows 0 and 16, which pair with themselves -- so the partner of what lane l holds in register r
// is in lane 63-l, register 7-r of the other row, one cross-lane read away.  A lane of a pair of
// rows works out both bins and hands the partner's result back the same way; the two rows that
// pair with themselves work out every bin from its own side.
// Tables (H): one 16-byte entry per bin, {Hev / N2, (i/2) Hod / N2}, row by row ([k1][k2]).  (tw): exp(-2 pi i k / 512),
// k < 512; the stage factors exp(-2 pi i k1 n2 / N2) as [k1][n2]; exp(-2 pi i (k1 + K1 l) / N2) as
// [k1][l], l < 64 (the factor w of a lane's register 0; register r has w . exp(-2 pi i r / 8)).

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ C<float> lane_read(unsigned byte_addr, C<float> q)
{
    return {__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)byte_addr, __builtin_bit_cast(int, q.re))),
            __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)byte_addr, __builtin_bit_cast(int, q.im)))};
}

// multiply-accumulate on complex numbers: acc + a b
__device__ __forceinline__ C<float> cmla(C<float> acc, C<float> a, C<float> b)
{
    return {acc.re + a.re * b.re - a.im * b.im, acc.im + a.re * b.im + a.im * b.re};
}

// Rows a and b = K1 - a of one wave, spectra in, swapped filtered spectra out (the second
// transform is swap . forward . swap).  hg_tab points at this lane's entry of row a; the partner
// bin's factors need no tables: the taps are real, so Hev[N2-k] = conj Hev[k] and
// (i/2) Hod[N2-k] = -conj((i/2) Hod[k]).
template <int r>
__device__ __forceinline__ void pair_step(C<float> *ua, C<float> *ub, const f32x4_t *hg_tab, C<float> w0, unsigned rev)
{
    const C<float> A = ua[r], B = lane_read(rev, ub[7 - r]);
    const f32x4_t hg = hg_tab[64 * r];  // both factors of a bin in one 16-byte load
    const C<float> hk = {hg.x, hg.y}, gk = {hg.z, hg.w};
    const C<float> S = {A.re + B.re, A.im - B.im}, D = {A.re - B.re, A.im + B.im};
    const C<float> t = cmul(mul_w32<4 * r>(w0), D);
    const C<float> U = {S.re - t.re, S.im - t.im}, V = {S.re + t.re, S.im + t.im};
    const C<float> zk = cmla(cmul(hk, A), gk, U);
    // conj(hk) B - conj(gk) conj(V) = conj(hk conj(B) - gk V)
    const C<float> q = cmla(cmul(hk, C<float>{B.re, -B.im}), C<float>{-gk.re, -gk.im}, V);
    ua[r] = {zk.im, zk.re};
    ub[7 - r] = lane_read(rev, C<float>{-q.im, q.re});  // conj(q), swapped
}
template <int... R>
__device__ __forceinline__ void pair_rows(C<float> *ua, C<float> *ub, const f32x4_t *hg_tab, C<float> w0, unsigned rev,
                                          std::integer_sequence<int, R...>)
{
    (pair_step<R>(ua, ub, hg_tab, w0, rev), ...);
}

// A row that pairs with itself: the partner of register r is register 7-r of lane `from` / 4
// (row 16: 63 - lane; row 0: (64 - lane) mod 64, and lane 0 pairs its own register r with
// register (8 - r) mod 8).
template <int r>
__device__ __forceinline__ void self_step(C<float> *u, const C<float> *b, const f32x4_t *hg_tab, C<float> w0)
{
    const C<float> A = u[r], B = b[r];
    const C<float> S = {A.re + B.re, A.im - B.im}, D = {A.re - B.re, A.im + B.im};
    const C<float> t = cmul(mul_w32<4 * r>(w0), D);
    const C<float> U = {S.re - t.re, S.im - t.im};
    const f32x4_t hg = hg_tab[64 * r];
    const C<float> zk = cmla(cmul(C<float>{hg.x, hg.y}, A), C<float>{hg.z, hg.w}, U);
    u[r] = {zk.im, zk.re};
}
template <bool ROW0, int... R>
__device__ __forceinline__ void pair_self(C<float> *u, const f32x4_t *hg_tab, C<float> w0, unsigned from, bool lane0,
                                          std::integer_sequence<int, R...>)
{
    C<float> b[8];
    ((b[R] = lane_read(from, u[7 - R])), ...);
    if constexpr (ROW0) {
        ((b[R] = lane0 ? u[(8 - R) & 7] : b[R]), ...);
    }
    (self_step<R>(u, b, hg_tab, w0), ...);
}

template <int K1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, K1 == 32 ? 4 : 6))) void ols_real_kernel(const OlsArgs a,
                                                                                                  const C<float> *__restrict__ tw_sub,
                                                                                                  const C<float> *__restrict__ tw_stage,
                                                                                                  const C<float> *__restrict__ tw_bin)
{
    typedef float T;
    constexpr int NS = 512, NW = 8, N2 = K1 * NS, M = 2 * N2;
    static_assert(K1 == 32 || K1 == 16, "K1 / 2 pairs of rows over eight waves");
    constexpr int UPW = K1 / 16, RPW = 2 * UPW;  // pairs of rows, rows per wave
    constexpr std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7> regs{};

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned tl = threadIdx.x, lane = tl & 63, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tl >> 6));
    float *img = reinterpret_cast<float *>(smem_raw);  // [K1][512], one component at a time
    C<T> *wreg = reinterpret_cast<C<T> *>(smem_raw) + (size_t)wave * padded_len(NS);
    // the waves' regions alias the image; the row tables sit behind the larger of the two
    constexpr size_t REGIONS = (size_t)NW * padded_len(NS) * sizeof(C<T>), IMG = (size_t)N2 * sizeof(float);
    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    setup_lds_twiddles<T, NS, 512, 0, 8, 8, 8, 1>(smem_raw + (IMG > REGIONS ? IMG : REGIONS), tw_sub, tw1, tw2, tw3);

    // the rows of this wave: s = 0, 1 one pair, s = 2, 3 the other
    // (K1 = 32: pairs 2 wave and 2 wave + 1; K1 = 16: pair wave.  Pair 0 is rows 0 and K1 / 2.)
    const unsigned row[4] = {UPW * wave, wave ? K1 - UPW * wave : K1 / 2, 2 * wave + 1, K1 - 1 - 2 * wave};

    const long nx = a.nx, nh = a.nh, L = a.L;
    const size_t total = (size_t)a.nchan * a.nblk;
    // Blocks overlap by an even number of samples (nh-1, or nh when that is odd), so that every
    // block starts on an even sample: 8-byte accesses when the rows start on 8-byte boundaries
    const long ov = M - L;
    const bool even = !(a.x_stride & 1) && !(a.y_stride & 1) && !((reinterpret_cast<size_t>(a.x) | reinterpret_cast<size_t>(a.y)) & 7);
    for (size_t frame = blockIdx.x; frame < total; frame += gridDim.x) {
        const size_t c = frame / a.nblk;
        const unsigned blk = (unsigned)(frame - c * a.nblk);
        const float *xrow = a.x + c * a.x_stride;
        const float *hrow = a.hist ? a.hist + c * a.hist_stride : nullptr;
        const long base = (long)blk * L - ov;  // sample index of element 0 of the block
        const bool interior = even && base >= 0 && base + M <= nx;
        unsigned tt = tl;
        asm volatile("" : "+v"(tt));  // (see ols_wave_kernel: keeps per-frame addresses from being formed once and kept)

        // element n = tl + 512 n1 of the block: re = x[base + 2n], im = x[base + 2n + 1]
        C<T> v[K1];
        if (interior) {
            const C<T> *p0 = reinterpret_cast<const C<T> *>(xrow + base) + tt;
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) v[n1] = nt_load(p0 + NS * n1);
        } else {
            auto sample = [&](long s) -> float {
                if (s >= 0) return s < nx ? xrow[s] : 0.f;
                return (hrow && s >= 1 - nh) ? hrow[nh - 1 + s] : 0.f;
            };
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                const long s = base + 2 * (long)(tt + NS * n1);
                v[n1].re = sample(s);
                v[n1].im = sample(s + 1);
            }
        }
#pragma unroll
        for (int n1 = 0; n1 < K1; n1++) asm volatile("" : "+v"(v[n1].re), "+v"(v[n1].im));

        // ---- first transform: radix-K1 sum, stage factors, rows ----
        StageFactors<K1, NS> sf;
        sf.load(tw_stage + tt);
        dft_reg<K1, 1>(v);
        sf.apply(v);
        C<T> u[RPW][8];
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
#pragma unroll
            for (int k1 = 0; k1 < K1; k1++) img[NS * k1 + tl] = comp ? v[k1].im : v[k1].re;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < RPW; s++)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float q = img[NS * row[s] + lane + 64 * r];
                    if (comp) u[s][r].im = q;
                    else u[s][r].re = q;
                }
            __syncthreads();
        }
#pragma unroll
        for (int pr = 0; pr < UPW; pr++) {
            C<T> *ua = u[2 * pr], *ub = u[2 * pr + 1];
            const unsigned ra = row[2 * pr], rb = row[2 * pr + 1];
            unsigned la = lane, lb = lane, lh = lane;
            asm volatile("" : "+v"(la));
            row_passes_512(ua, wreg, tw1, tw2, la);
            asm volatile("" : "+v"(lb));
            row_passes_512(ub, wreg, tw1, tw2, lb);
            // the spectrum of the filtered block, from each bin and its partner
            asm volatile("" : "+v"(lh));
            const f32x4_t *hg_all = reinterpret_cast<const f32x4_t *>(a.H);
            const f32x4_t *he = hg_all + (size_t)NS * ra + lh, *heb = hg_all + (size_t)NS * rb + lh;
            if (pr == 0 && wave == 0) {
                pair_self<true>(ua, he, tw_bin[64 * ra + lh], ((64u - lh) & 63u) * 4u, lh == 0, regs);
                pair_self<false>(ub, heb, tw_bin[64 * rb + lh], (63u - lh) * 4u, false, regs);
            } else {
                pair_rows(ua, ub, he, tw_bin[64 * ra + lh], (63u - lh) * 4u, regs);
            }
            // ---- second transform: rows, stage factors, radix-K1 sum ----
            asm volatile("" : "+v"(la));
            row_passes_512(ua, wreg, tw1, tw2, la);
            asm volatile("" : "+v"(lb));
            row_passes_512(ub, wreg, tw1, tw2, lb);
        }
        {
            unsigned t2 = tl;
            asm volatile("" : "+v"(t2));
            sf.load(tw_stage + t2);  // on their way during the exchange
        }
        __syncthreads();  // the waves' regions alias the image: every wave is done with its rows
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
#pragma unroll
            for (int s = 0; s < RPW; s++)
#pragma unroll
                for (int r = 0; r < 8; r++) img[NS * row[s] + lane + 64 * r] = comp ? u[s][r].im : u[s][r].re;
            __syncthreads();
#pragma unroll
            for (int k1 = 0; k1 < K1; k1++) {
                const float q = img[NS * k1 + tl];
                if (comp) v[k1].im = q;
                else v[k1].re = q;
            }
            __syncthreads();
        }
        sf.apply(v);
        dft_reg<K1, 1>(v);

        // sample m = 2n (+1) of the block is output m - ov of it; the result comes out
        // swapped: im is the even sample, re the odd one
        unsigned to = tl;
        asm volatile("" : "+v"(to) : "v"(v[0].re));
        float *yrow = a.y + c * a.y_stride;
        if (interior) {
            C<T> *q0 = reinterpret_cast<C<T> *>(yrow + base) + to;
            const int skip = (int)ov / 2 - (int)to;  // element offsets below this are overlap
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++)
                if (NS * n1 >= skip) nt_store(C<T>{v[n1].im, v[n1].re}, q0 + NS * n1);
        } else {
#pragma unroll
            for (int n1 = 0; n1 < K1; n1++) {
                const long j = 2 * (long)(to + NS * n1) - ov;
                const long o0 = (long)blk * L + j;
                if (j >= 0 && o0 < nx) yrow[o0] = v[n1].im;
                if (j + 1 >= 0 && o0 + 1 < nx) yrow[o0 + 1] = v[n1].re;
            }
        }
    }
}

template <int K1>
int launch_ols_real(const dspmi::OlsLaunch &o, const void *tables, int num_cu, hipStream_t s)
{
    constexpr int NS = 512, N2 = K1 * NS, M = 2 * N2;
    constexpr size_t regions = (size_t)8 * padded_len(NS) * sizeof(C<float>), img = (size_t)N2 * sizeof(float);
    constexpr size_t lds = (img > regions ? img : regions) + lds_twiddle_entries(2, 8, 8, 8, 1) * sizeof(C<float>);
    static_assert((K1 == 32 ? 2 : 3) * lds <= 160 * 1024, "two workgroups per CU (three with 16 points per lane)");
    auto kern = ols_real_kernel<K1>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    OlsArgs a;
    a.x = o.x; a.hist = o.hist; a.y = o.y; a.H = static_cast<const C<float> *>(o.H);
    a.x_stride = o.x_stride; a.y_stride = o.y_stride; a.hist_stride = o.hist_stride;
    a.nx = o.nx; a.nh = o.nh; a.L = M - (o.nh & ~1u);  // an even overlap: nh-1, or nh
    a.nblk = (unsigned)(((unsigned long long)o.nx + a.L - 1) / a.L);
    a.npair = a.nblk;
    a.nchan = o.nchan;
    const size_t total = (size_t)a.nchan * a.nblk;
    if (total == 0) return DSPMI_OK;
    // One block per workgroup, two workgroups per CU at a time: the dispatcher hands a CU its next
    // block the moment one is finished, which evens out what a fixed walk leaves to the last round
    // (2731 blocks of 2^26 samples with rotating buffers: 0.224 ms with a resident set of 512
    // workgroups walking the blocks, 0.218 with four times that, 0.209 with one block each).
    size_t grid = total < 0x7fffffffu ? total : 0x7fffffffu;
    if (const char *g = getenv("DSPMI_OLS_GRID"))  // tests: few workgroups, so that short signals make each walk several blocks
        if (atoi(g) > 0 && (size_t)atoi(g) < grid) grid = (size_t)atoi(g);
    // tables: exp(-2 pi i k / 512), k < 512; the stage factors [K1][512]; the bin factors [K1][64]
    const C<float> *tw_sub = static_cast<const C<float> *>(tables);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, a, tw_sub, tw_sub + NS, tw_sub + NS + N2);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <int N, int NT, int R0, int R1, int R2, int R3, int TW>
int launch_ols(const dspmi::OlsLaunch &o, const void *tw, int num_cu, hipStream_t s)
{
    constexpr int BT = ols_block_threads<N, NT>();
    constexpr int FPB = BT / NT;
    constexpr size_t lds = ((size_t)FPB * padded_len(N) + lds_twiddle_entries(TW, R0, R1, R2, R3, N)) * sizeof(C<float>);
    auto kern = ols_fused_kernel<N, NT, R0, R1, R2, R3, TW>;
    static dspmi::PerDeviceOnce attr;
    if (lds > 48 * 1024 && attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    OlsArgs a;
    a.x = o.x; a.hist = o.hist; a.y = o.y; a.H = static_cast<const C<float> *>(o.H);
    a.x_stride = o.x_stride; a.y_stride = o.y_stride; a.hist_stride = o.hist_stride;
    a.nx = o.nx; a.nh = o.nh; a.L = N - (o.nh - 1);
    a.nblk = (unsigned)(((unsigned long long)o.nx + a.L - 1) / a.L);
    a.npair = (a.nblk + 1) / 2;
    a.nchan = o.nchan;
    const size_t total = (size_t)a.nchan * a.npair;
    if (total == 0) return DSPMI_OK;
    const size_t nfb = (total + FPB - 1) / FPB;
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 2048 / BT) per_cu = 2048 / BT;
    if (per_cu < 1) per_cu = 1;
    // compute-bound: four times the resident set (N = 1024: 1.11 -> 1.08 ms, 8192: 1.94 -> 1.82; larger grids are slower)
    const size_t grid = grid_for(nfb, (size_t)num_cu * per_cu, (double)nfb / (4.0 * num_cu * per_cu));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BT), lds, s, a, static_cast<const C<float> *>(tw));
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

struct OlsEntry {
    unsigned n;
    dspmi::ols_launch_fn fn;
};
#define OLS(N, NT, A, B, C_, D, TW) {N, launch_ols<N, NT, A, B, C_, D, TW>}
const OlsEntry kOls[] = {
    OLS(16, 1, 16, 1, 1, 1, 0),        OLS(32, 2, 4, 2, 4, 1, 2),          OLS(64, 4, 4, 4, 4, 1, 2),
    OLS(128, 8, 4, 8, 4, 1, 2),        OLS(256, 16, 16, 16, 1, 1, 2),      OLS(512, 32, 16, 2, 16, 1, 2),
    OLS(1024, 64, 16, 4, 16, 1, 2),    OLS(2048, 128, 16, 8, 16, 1, 2),    OLS(4096, 256, 16, 16, 16, 1, 3),
    OLS(8192, 512, 16, 16, 2, 16, 3),  OLS(16384, 1024, 16, 16, 4, 16, 3),
    {32768, launch_ols_wave<32>},
};

// ---------------------------------------------------------------------------
// any length n through a power-of-two convolution (Bluestein), one kernel
// ---------------------------------------------------------------------------

// X[k] = c[k] * sum_m (x[m] c[m]) conj(c)[k-m],  c[m] = exp(-i pi m^2 / n): the frame times the
// chirp, zero-extended to M >= 2n-1 points, is convolved with conj(c) through two M-point
// transforms (filt = FFT_M(conj c laid out circularly) / M) and multiplied by the chirp again.
// Same shape as the overlap-save kernel: with a symmetric radix list the spectrum is multiplied
// in registers and goes straight back, so a frame costs one read and one write of n points
// instead of the five passes over M points of the separate kernels.
template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
__global__ __launch_bounds__(block_threads<NT>()) void bluestein_fused_kernel(const C<T> *in, C<T> *out,
                                                                              const C<T> *__restrict__ chirp,
                                                                              const C<T> *__restrict__ filt,
                                                                              const C<T> *__restrict__ tw, unsigned n,
                                                                              size_t batch, int backward)
{
    static_assert(R0 * R1 * R2 * R3 == N, "radices must multiply to N");
    constexpr int P = N / NT;
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr int NPASS = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    constexpr int RLAST = NPASS == 1 ? R0 : NPASS == 2 ? R1 : NPASS == 3 ? R2 : R3;
    static_assert(RLAST == R0, "first and last radix must match");
    constexpr int STEP = N / R0;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned slot = threadIdx.x / NT;
    const unsigned tl = threadIdx.x % NT;
    C<T> *lds = reinterpret_cast<C<T> *>(smem_raw) + (size_t)slot * padded_len(N);
    const C<T> *tw1 = nullptr, *tw2 = nullptr, *tw3 = nullptr;
    if constexpr (TW == 2) setup_lds_twiddles<T, N, BT, FPB, R0, R1, R2, R3>(smem_raw, tw, tw1, tw2, tw3);
    if constexpr (TW == 3) setup_lds_twiddles_split<T, N, BT, FPB, R0, R1>(smem_raw, tw, tw1, tw2, tw3);
    const T sgn = backward ? T(-1) : T(1);  // backward = conj . forward . conj

    const size_t nfb = (batch + FPB - 1) / FPB;
    for (size_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const size_t frame = fb * FPB + slot;
        const bool active = frame < batch;
        const C<T> *src = in + frame * n;
        // laundered lane index: the chirp and filter addresses are the same for every frame, and
        // hoisting those loads out of the frame loop would cost their registers through both transforms
        unsigned te = tl;
        asm volatile("" : "+v"(te));
        C<T> v[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const unsigned m = te + NT * (i / R0) + (i % R0) * STEP;
            C<T> q = {T(0), T(0)};
            if (active && m < n) {
                q = src[m];
                q.im *= sgn;
                q = cmul(q, chirp[m]);
            }
            v[i] = q;
        }
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));
        {
            unsigned th = tl;
            asm volatile("" : "+v"(th));
            const C<T> *fp = filt + th;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const C<T> m = cmul(v[i], fp[NT * (i / R0) + (i % R0) * STEP]);
                v[i] = {m.re, -m.im};
            }
        }
        run_passes<T, N, NT, R0, R1, R2, R3, TW>(v, lds, tw1, tw2, tw3, tw, pass_lane<TW>(tl));
        if (active) {
            C<T> *dst = out + frame * n;
            unsigned to = tl;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (int i = 0; i < P; i++) {
                const unsigned k = to + NT * (i / R0) + (i % R0) * STEP;
                if (k < n) {
                    C<T> q = cmul(C<T>{v[i].re, -v[i].im}, chirp[k]);
                    q.im *= sgn;
                    dst[k] = q;
                }
            }
        }
    }
}

template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
int launch_bluestein(const void *in, void *out, const void *chirp, const void *filt, const void *tw, unsigned n, size_t batch,
                     int backward, int num_cu, hipStream_t s)
{
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr size_t lds = ((size_t)FPB * padded_len(N) + lds_twiddle_entries(TW, R0, R1, R2, R3, N)) * sizeof(C<T>);
    auto kern = bluestein_fused_kernel<T, N, NT, R0, R1, R2, R3, TW>;
    static dspmi::PerDeviceOnce attr;
    if (lds > 48 * 1024 && attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    if (batch == 0) return DSPMI_OK;
    const size_t nfb = (batch + FPB - 1) / FPB;
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 2048 / BT) per_cu = 2048 / BT;
    if (per_cu < 1) per_cu = 1;
    const size_t grid = grid_for(nfb, (size_t)num_cu * per_cu, 0.0);  // compute-bound: the resident set measured fastest
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BT), lds, s, static_cast<const C<T> *>(in), static_cast<C<T> *>(out),
                       static_cast<const C<T> *>(chirp), static_cast<const C<T> *>(filt), static_cast<const C<T> *>(tw), n, batch,
                       backward);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

struct BluesteinEntry {
    unsigned m;
    dspmi::bluestein_launch_fn fn;
};
#define BLU(N, NT, A, B, C_, D, TW) {N, launch_bluestein<float, N, NT, A, B, C_, D, TW>}
const BluesteinEntry kBluesteinF32[] = {
    BLU(16, 1, 16, 1, 1, 1, 0),       BLU(32, 2, 4, 2, 4, 1, 2),         BLU(64, 4, 4, 4, 4, 1, 2),
    BLU(128, 8, 4, 8, 4, 1, 2),       BLU(256, 16, 16, 16, 1, 1, 2),     BLU(512, 32, 16, 2, 16, 1, 2),
    BLU(1024, 64, 16, 4, 16, 1, 2),   BLU(2048, 128, 16, 8, 16, 1, 2),   BLU(4096, 256, 16, 16, 16, 1, 3),
    BLU(8192, 512, 16, 16, 2, 16, 3), BLU(16384, 1024, 16, 16, 4, 16, 3),
};

// ---------------------------------------------------------------------------
// c2c with the whole frame in registers: 1024 .. 32768 points
// ---------------------------------------------------------------------------

// A whole frame in the registers of one workgroup: N = P . P . RC points, L = P RC lanes with P
// points each.  n = n0 + L n1 (n1 < P), n0 = m0 + RC m1; k = k1 + P q1 + P^2 q0:
//   X[k] = sum_m0 W_RC^(m0 q0) . W_L^(m0 q1) sum_m1 W_P^(m1 q1) . W_N^(n0 k1) sum_n1 W_P^(n1 k1) x[n]
// Lane n0 loads x[n0 + L n1] straight into registers (a wave reads 512 contiguous bytes) and
// transforms over n1; an exchange through LDS -- real parts, then imaginary parts, through an image
// of about N scalars -- hands lane (k1, m0) its P values of m1; after the second radix a second
// exchange hands lane t the RC values of m0 for k1 + P q1 = t + L i, whose transforms
// X[t + L i + P^2 q0] go out in natural order (contiguous over the lanes again).  Both images are
// read and written without bank conflicts (rows of L + 32 / P scalars in the first, unit stride
// over k1 and padded planes in the second).  Half the LDS of a frame image of complex points, so
// two workgroups fit a CU at 16384 points in single precision and one loads while the other
// computes; 32768 points fit at all.
// The forward transform of the points a workgroup's lanes hold: in, v[j] = x[t + L j]; out,
// v[i RC + q0] = X[t + L (i + TC q0)] -- the same set of indices t + L j per lane, so a second
// transform can follow in registers (after a permutation of register names).  SYNC_FIRST: the image
// may still be read by other lanes (a transform went before).
template <typename T, int P, int RC, bool SYNC_FIRST>
__device__ __forceinline__ void reg3_forward(C<T> (&v)[P], T *img, const C<T> *__restrict__ tw, unsigned t)
{
    constexpr int L = P * RC, TC = P / RC, ROW = L + 32 / P, PLANE = P * P + (P < 32 ? 16 : 0);
    static_assert(RC * TC == P && (P == 16 || P == 32), "whole last-radix transforms per lane");
    const unsigned k1 = t % P, m0 = t / P;
    // two factors for every W^(g k), k = 1 .. P - 1: k = 4 a + b
    auto twiddle = [&](unsigned g) {
        C<T> l3[3], h4[P / 4 - 1];
#pragma unroll
        for (int b = 1; b < 4; b++) l3[b - 1] = tw[g * b];
#pragma unroll
        for (int a = 1; a < P / 4; a++) h4[a - 1] = tw[g * 4 * a];
#pragma unroll
        for (int k = 1; k < P; k++) {
            if (k / 4) v[k] = cmul(v[k], h4[k / 4 - 1]);
            if (k % 4) v[k] = cmul(v[k], l3[k % 4 - 1]);
        }
    };
    dft_reg<P, 1>(v);  // over n1 -> k1
    twiddle(t);        // W_N^(n0 k1)
    {
        T *wp = img + t;                      // [k1][n0]
        const T *rp = img + k1 * ROW + m0;    // n0 = m0 + RC m1
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            if (comp || SYNC_FIRST) __syncthreads();
#pragma unroll
            for (int j = 0; j < P; j++) wp[j * ROW] = comp ? v[j].im : v[j].re;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P; j++) {
                const T q = rp[RC * j];
                if (comp) v[j].im = q;
                else v[j].re = q;
            }
        }
    }
    dft_reg<P, 1>(v);   // over m1 -> q1
    twiddle(P * m0);    // W_L^(m0 q1)
    {
        T *wp = img + k1 + PLANE * m0;  // [m0][q1][k1]
        const T *rp = img + t;          // k1 + P q1 = t + L i
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P; j++) wp[P * j] = comp ? v[j].im : v[j].re;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TC; i++)
#pragma unroll
                for (int j = 0; j < RC; j++) {
                    const T q = rp[L * i + PLANE * j];
                    if (comp) v[i * RC + j].im = q;
                    else v[i * RC + j].re = q;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TC; i++) dft_reg<RC, 1>(&v[i * RC]);  // over m0 -> q0
}

template <typename T, int P, int RC, int NTMEM>
__global__ __launch_bounds__(P * RC) void fft_c2c_reg3_kernel(const C<T> *in, C<T> *out, const C<T> *__restrict__ tw, size_t batch,
                                                              int flags)
{
    constexpr int L = P * RC, N = P * L, TC = P / RC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *img = reinterpret_cast<T *>(smem_raw);
    const unsigned t = threadIdx.x;
    const T sgn = (flags & kFftBackward) ? T(-1) : T(1);  // conjugate . forward . conjugate
    const size_t f = blockIdx.x;
    if (f >= batch) return;

    C<T> v[P];
    {
        const C<T> *src = in + f * N + t;
#pragma unroll
        for (int j = 0; j < P; j++) {
            C<T> q = (NTMEM & 1) ? nt_load(src + L * j) : src[L * j];
            q.im *= sgn;
            v[j] = q;
        }
    }
    reg3_forward<T, P, RC, false>(v, img, tw, t);
    C<T> *dst = out + f * N + t;
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int q0 = 0; q0 < RC; q0++) {
            C<T> q = v[i * RC + q0];
            q.im *= sgn;
            if constexpr ((NTMEM & 2) != 0) nt_store(q, dst + L * (i + TC * q0));
            else dst[L * (i + TC * q0)] = q;
        }
}

// Any length n through an M-point convolution (bluestein_fused_kernel's arithmetic) with the frame in
// registers: frame times chirp, zero-extended -> transform -> times the filter's spectrum ->
// transform back -> times chirp, one read and one write of n points per frame.  The first
// transform leaves every lane the bins t + L j it started with as samples, so the second follows in
// registers.  Double precision, where the frame-image kernel does not fit: M = 512 .. 16384.
template <typename T, int P, int RC>
__global__ __launch_bounds__(P * RC) void bluestein_reg3_kernel(const C<T> *in, C<T> *out, const C<T> *__restrict__ chirp,
                                                                const C<T> *__restrict__ filt, const C<T> *__restrict__ tw, unsigned n,
                                                                size_t batch, int backward)
{
    constexpr int L = P * RC, TC = P / RC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *img = reinterpret_cast<T *>(smem_raw);
    const unsigned t = threadIdx.x;
    const T sgn = backward ? T(-1) : T(1);  // backward = conj . forward . conj
    const size_t f = blockIdx.x;
    if (f >= batch) return;

    C<T> v[P];
    {
        const C<T> *src = in + f * n;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const unsigned m = t + L * j;
            C<T> q = {T(0), T(0)};
            if (m < n) {
                q = src[m];
                q.im *= sgn;
                q = cmul(q, chirp[m]);
            }
            v[j] = q;
        }
    }
    reg3_forward<T, P, RC, false>(v, img, tw, t);
    C<T> w[P];
    {
        const C<T> *fp = filt + t;
#pragma unroll
        for (int i = 0; i < TC; i++)
#pragma unroll
            for (int q0 = 0; q0 < RC; q0++) {
                const int j = i + TC * q0;  // bin t + L j
                const C<T> m = cmul(v[i * RC + q0], fp[L * j]);
                w[j] = {m.re, -m.im};
            }
    }
    reg3_forward<T, P, RC, true>(w, img, tw, t);
    C<T> *dst = out + f * n;
#pragma unroll
    for (int i = 0; i < TC; i++)
#pragma unroll
        for (int q0 = 0; q0 < RC; q0++) {
            const unsigned k = t + L * (i + TC * q0);
            if (k < n) {
                C<T> q = cmul(C<T>{w[i * RC + q0].re, -w[i * RC + q0].im}, chirp[k]);
                q.im *= sgn;
                dst[k] = q;
            }
        }
}

template <typename T, int P, int RC>
int launch_bluestein_reg3(const void *in, void *out, const void *chirp, const void *filt, const void *tw, unsigned n, size_t batch,
                          int backward, int num_cu, hipStream_t s)
{
    constexpr int L = P * RC, ROW = L + 32 / P, PLANE = P * P + (P < 32 ? 16 : 0);
    constexpr size_t lds = (size_t)(P * ROW > RC * PLANE ? P * ROW : RC * PLANE) * sizeof(T);
    auto kern = bluestein_reg3_kernel<T, P, RC>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    (void)num_cu;
    for (size_t done = 0; done < batch;) {
        const size_t g = batch - done < 0x40000000u ? batch - done : 0x40000000u;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(L), lds, s, static_cast<const C<T> *>(in) + done * n,
                           static_cast<C<T> *>(out) + done * n, static_cast<const C<T> *>(chirp), static_cast<const C<T> *>(filt),
                           static_cast<const C<T> *>(tw), n, g, backward);
        DSPMI_HIP(hipGetLastError());
        done += g;
    }
    return DSPMI_OK;
}

template <typename T, int P, int RC, int NTMEM>
int launch_c2c_reg3(const void *in, void *out, const void *tw, size_t batch, int backward, int num_cu, hipStream_t s)
{
    constexpr int L = P * RC, N = P * L, ROW = L + 32 / P, PLANE = P * P + (P < 32 ? 16 : 0);
    constexpr size_t lds = (size_t)(P * ROW > RC * PLANE ? P * ROW : RC * PLANE) * sizeof(T);
    auto kern = fft_c2c_reg3_kernel<T, P, RC, NTMEM>;
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    (void)num_cu;
    // one frame per workgroup: the dispatcher keeps every CU filled
    for (size_t done = 0; done < batch;) {
        const size_t g = batch - done < 0x40000000u ? batch - done : 0x40000000u;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(L), lds, s, static_cast<const C<T> *>(in) + done * N,
                           static_cast<C<T> *>(out) + done * N, static_cast<const C<T> *>(tw), g, backward ? kFftBackward : 0);
        DSPMI_HIP(hipGetLastError());
        done += g;
    }
    return DSPMI_OK;
}

// ---------------------------------------------------------------------------
// long transforms in two passes: the transposes folded into the transforms' loads and stores
// ---------------------------------------------------------------------------

// n = n1 n2 points, x[a + n1 b] -> X[n2 k1 + k2]:
//   X[n2 k1 + k2] = sum_a W_n1^(a k1) . W_n^(a k2) . sum_b x[a + n1 b] W_n2^(b k2)
// Pass 1 (MODE 1, N = n2, ncol = n1) transforms COLUMNS of the frame seen as [n2][n1] and
// multiplies by W_n^(a k2); pass 2 (MODE 2, N = n1, ncol = n2) transforms the n1 points of each k2
// and stores them as a column of the output seen as [n1][n2].  Two passes over memory instead of
// the five of transposing between batched row transforms.  The backward transform conjugates on
// the way in (pass 1) and on the way out (pass 2).  W_n^e = twa[e >> shift] . twb[e & mask], as in
// transpose_kernel.
//
// A frame's points are held in registers: N = RA . RB, a workgroup of 16 RB
// lanes takes 16 adjacent columns, lane (row % RB, column) loads its RA points straight from memory
// (rows of 16 adjacent columns: 128 contiguous bytes), transforms them, and ONE exchange through
// LDS -- real parts, then imaginary parts, through an image of 16 N floats -- hands lane
// (k % RB.., column) the RB points of the second radix.  No transposition through LDS on the way
// in or out, and 16 N points in flight per workgroup.  Pass 1 leaves its result in tiles
// [k2 / 16][a / 16][k2 % 16][a % 16], so that a wave stores 512 contiguous bytes and pass 2 reads
// its 16 rows as one contiguous run of 128 N bytes; pass 2 stores rows of 16 adjacent columns of
// the output.
//
// Three passes for n = t1 t2 t3 beyond 2^20 points, x[a + t1 (b + t2 c)] -> X[t2 t3 k1 + t3 kb + kc]
// (the same identity applied twice; every pass leaves its result where it found its input, so one
// work frame per frame serves all three):
//   MODE 3  N = t3, ncol = t1 t2, p0 = t1: columns (b, a) over c, times W_(t2 t3)^(b kc) = W_n^(t1 b kc)
//   MODE 4  N = t2, ncol = t1, p0 = t3, one frame per kc: columns a over b, times W_n^(a (t3 kb + kc))
//   MODE 5  N = t1, ncol = t2 t3, p0 = t2, p1 = t3: the rows (kc, kb) of 16 adjacent kc over a, stored as
//           16 adjacent columns t3 kb + kc of the output seen as [t1][t2 t3]
template <typename T, int RA, int RB, int MODE>
__global__ __launch_bounds__(16 * RB) void fft_c2c_colreg_kernel(const C<T> *in, C<T> *out, const C<T> *__restrict__ tw, unsigned bpf,
                                                                 unsigned ncol, const C<T> *__restrict__ twa,
                                                                 const C<T> *__restrict__ twb, unsigned shift, int backward,
                                                                 unsigned p0, unsigned p1)
{
    // RA >= RB: lane (kq, column) of the second radix takes M = RA / RB transforms, k % RA = kq + RB m;
    // RA < RB: one transform in the lanes with kq < RA, the others only keep the barriers company
    constexpr int N = RA * RB, M = RA >= RB ? RA / RB : 1, P = RA >= RB ? RA : RB, SB = RA + 1,
                  SC = RB * SB + (34 - (RB * SB) % 32) % 32;
    static_assert((RA >= RB ? M * RB == RA : RB % RA == 0) && RB % 16 == 0, "whole transforms per lane, whole tiles of 16 rows");
    constexpr bool kColumnsIn = MODE == 1 || MODE == 3 || MODE == 4;   // natural [N][ncol] input, 16 adjacent columns
    constexpr bool kConjIn = MODE == 1 || MODE == 3, kConjOut = MODE == 2 || MODE == 5;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *cimg = reinterpret_cast<T *>(smem_raw);
    const unsigned tid = threadIdx.x;
    const size_t F = blockIdx.x / bpf;
    const unsigned cb = (unsigned)(blockIdx.x - F * bpf);  // the bundle of 16 columns within the frame
    const size_t base = F * (size_t)N * ncol;
    const T sgn = backward ? T(-1) : T(1);

    C<T> v[P];
    unsigned lo, col;
    if constexpr (kColumnsIn) {
        col = tid & 15;
        lo = tid >> 4;
        const C<T> *src = in + base + (size_t)lo * ncol + cb * 16 + col;
#pragma unroll
        for (int j = 0; j < RA; j++) {
            C<T> q = MODE == 4 ? src[(size_t)(RB * j) * ncol] : nt_load(src + (size_t)(RB * j) * ncol);
            if constexpr (kConjIn) q.im *= sgn;
            v[j] = q;
        }
    } else if constexpr (MODE == 2) {
        col = (tid >> 4) & 15;
        lo = (tid & 15) + 16 * (tid >> 8);
        const C<T> *src = in + base + (size_t)cb * (16 * N) + tid;
#pragma unroll
        for (int j = 0; j < RA; j++) v[j] = src[RB * 16 * j];
    } else {
        // F = (frame, kb); row `col` of the bundle is row (16 cb + col, kb) of the frame's [t3][t2][N] image
        col = tid / RB;
        lo = tid % RB;
        const size_t frame = F / p0;
        const unsigned kb = (unsigned)(F - frame * p0);
        const C<T> *src = in + frame * ((size_t)N * ncol) + ((size_t)(cb * 16 + col) * p0 + kb) * N + lo;
#pragma unroll
        for (int j = 0; j < RA; j++) v[j] = src[RB * j];
    }
    dft_reg<RA, 1>(v);
    {
        // W_N^(lo ka) for ka = 1, 2, 3 and 4, 8, ..; the others as two factors
        C<T> l3[3], h4[RA / 4 - 1];
#pragma unroll
        for (int b = 1; b < 4; b++) l3[b - 1] = tw[lo * b];
#pragma unroll
        for (int a = 1; a < RA / 4; a++) h4[a - 1] = tw[lo * 4 * a];
#pragma unroll
        for (int ka = 1; ka < RA; ka++) {
            if (ka / 4) v[ka] = cmul(v[ka], h4[ka / 4 - 1]);
            if (ka % 4) v[ka] = cmul(v[ka], l3[ka % 4 - 1]);
        }
    }
    const unsigned kq = tid >> 4, colr = tid & 15;
    const bool active = RA >= RB || kq < RA;
    {
        T *wp = cimg + col * SC + lo * SB;
        const T *rp = cimg + colr * SC + kq;
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            if (comp) __syncthreads();
#pragma unroll
            for (int j = 0; j < RA; j++) wp[j] = comp ? v[j].im : v[j].re;
            __syncthreads();
            if (active)
#pragma unroll
                for (int m = 0; m < M; m++)
#pragma unroll
                    for (int j = 0; j < RB; j++) {
                        const T q = rp[j * SB + RB * m];
                        if (comp) v[m * RB + j].im = q;
                        else v[m * RB + j].re = q;
                    }
        }
    }
    if (!active) return;  // no barrier from here on
#pragma unroll
    for (int m = 0; m < M; m++) dft_reg<RB, 1>(&v[m * RB]);
    // v[m RB + kb] = X[kq + RB m + RA kb] of column colr
    if constexpr (kColumnsIn) {
        // times W_n^(a (kmul k + kadd)): one look-up per transform, step factors for the rest
        const unsigned a = MODE == 3 ? (cb * 16 / p0) * p0 : cb * 16 + colr, mask = (1u << shift) - 1;
        const unsigned kmul = MODE == 4 ? p0 : 1u, kadd = MODE == 4 ? (unsigned)(F % p0) : 0u;
        auto W = [&](unsigned e) { return cmul(twa[e >> shift], twb[e & mask]); };  // W_n^e, e < n
        C<T> l[4], h[RB / 4];
        const unsigned gstep = a * kmul * RA;
#pragma unroll
        for (int i = 1; i < 4; i++) l[i] = W(gstep * i);
#pragma unroll
        for (int i = 1; i < RB / 4; i++) h[i] = W(gstep * 4 * i);
        // MODE 1: tiles [k / 16][column / 16][k % 16][column % 16]; MODE 3, 4: back in place
        C<T> *dst = out + base + (MODE == 1 ? (size_t)cb * 256 : (size_t)cb * 16) + colr;
        const size_t tile_pitch = (size_t)ncol * 16;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const C<T> b0 = W(a * (kmul * (kq + RB * m) + kadd));
            const C<T> lb[4] = {b0, cmul(b0, l[1]), cmul(b0, l[2]), cmul(b0, l[3])};
#pragma unroll
            for (int kb = 0; kb < RB; kb++) {
                C<T> f = lb[kb % 4];
                if (kb / 4) f = cmul(f, h[kb / 4]);
                const unsigned k = kq + RB * m + RA * kb;
                if constexpr (MODE == 1) dst[(size_t)(k >> 4) * tile_pitch + (k & 15) * 16] = cmul(v[m * RB + kb], f);
                else dst[(size_t)k * ncol] = cmul(v[m * RB + kb], f);
            }
        }
    } else {
        C<T> *dst;
        if constexpr (MODE == 2) {
            dst = out + base + cb * 16 + colr;
        } else {
            const size_t frame = F / p0;
            const unsigned kb = (unsigned)(F - frame * p0);
            dst = out + frame * ((size_t)N * ncol) + (size_t)p1 * kb + cb * 16 + colr;
        }
#pragma unroll
        for (int m = 0; m < M; m++)
#pragma unroll
            for (int kb = 0; kb < RB; kb++) {
                const unsigned k = kq + RB * m + RA * kb;
                C<T> q = v[m * RB + kb];
                if constexpr (kConjOut) q.im *= sgn;
                nt_store(q, dst + (size_t)k * ncol);
            }
    }
}

template <typename T, int RA, int RB>
int launch_c2c_colreg(const void *in, void *out, const void *tw, size_t nframe, unsigned ncol, const void *twa, const void *twb,
                      unsigned shift, int mode, int backward, unsigned p0, unsigned p1, hipStream_t s)
{
    constexpr int SB = RA + 1, SC = RB * SB + (34 - (RB * SB) % 32) % 32;
    constexpr size_t lds = (size_t)16 * SC * sizeof(T);
    typedef void (*kern_t)(const C<T> *, C<T> *, const C<T> *, unsigned, unsigned, const C<T> *, const C<T> *, unsigned, int, unsigned,
                           unsigned);
    static const kern_t kerns[5] = {fft_c2c_colreg_kernel<T, RA, RB, 1>, fft_c2c_colreg_kernel<T, RA, RB, 2>,
                                    fft_c2c_colreg_kernel<T, RA, RB, 3>, fft_c2c_colreg_kernel<T, RA, RB, 4>,
                                    fft_c2c_colreg_kernel<T, RA, RB, 5>};
    static dspmi::PerDeviceOnce attr;
    if (attr.pending()) {
        for (kern_t k : kerns)
            DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    if (mode < 1 || mode > 5) return dspmi::fail(DSPMI_EINVAL, "fft: column pass mode %d", mode);
    // bundles of 16 columns per frame; mode 5: 16 of the p1 rows (kc) for each of the p0 rows kb
    const unsigned bundled = mode == 5 ? p1 : ncol;
    if (bundled % 16) return dspmi::fail(DSPMI_EINVAL, "fft: %u columns do not divide into bundles of 16", bundled);
    const size_t nbundle = nframe * (mode == 5 ? (size_t)p0 : 1) * (bundled / 16);
    if (nbundle == 0) return DSPMI_OK;
    if (nbundle > 0x7fffffffu) return dspmi::fail(DSPMI_EINVAL, "fft: %zu column bundles in one launch", nbundle);
    hipLaunchKernelGGL(kerns[mode - 1], dim3((unsigned)nbundle), dim3(16 * RB), lds, s, static_cast<const C<T> *>(in),
                       static_cast<C<T> *>(out), static_cast<const C<T> *>(tw), bundled / 16, ncol, static_cast<const C<T> *>(twa),
                       static_cast<const C<T> *>(twb), shift, backward, p0, p1);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW, bool PREFETCH, int NTMEM = 0>
int launch_c2c(const void *in, void *out, const void *tw, size_t batch, int backward, int num_cu, hipStream_t s)
{
    constexpr int BT = c2c_block_threads<T, N, NT>();
    constexpr int FPB = BT / NT;
    constexpr bool single_pass = R1 == 1 && R2 == 1 && R3 == 1;  // global -> registers -> global
    constexpr size_t tw_entries = lds_twiddle_entries(TW, R0, R1, R2, R3, N);
    constexpr bool staged = NT < 16;  // short frames move through their LDS images (see the kernel)
    constexpr size_t lds = (single_pass && !staged) ? 0 : ((size_t)FPB * padded_len(N) + tw_entries) * sizeof(C<T>);
    auto kern = fft_c2c_kernel<T, N, NT, R0, R1, R2, R3, TW, PREFETCH, NTMEM>;
    static dspmi::PerDeviceOnce attr;
    if (lds && attr.pending()) {
        // the full 160 KB: the request below may be inflated to cap the resident workgroups
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
        attr.mark();
    }
    if (batch == 0) return DSPMI_OK;
    const size_t nfb = (batch + FPB - 1) / FPB;
    // workgroups that fit a CU at once
    size_t per_cu = lds ? (160 * 1024) / lds : 8;
    if (per_cu > 2048 / BT) per_cu = 2048 / BT;
    if (per_cu < 1) per_cu = 1;
    // The prefetching variant (persistent grid, next frame's loads in flight) streams fastest with
    // two workgroups per CU (5.47 TB/s against 5.21 with three); DSPMI_FFT_BPC overrides.
    size_t cap = (PREFETCH && NT <= 64) ? 2 : per_cu;
    {
        const char *env = getenv("DSPMI_FFT_BPC");
        if (env && atoi(env) > 0) cap = (size_t)atoi(env);
    }
    size_t lds_req = lds;
    if (lds && cap < per_cu) {
        per_cu = cap;
        lds_req = (160 * 1024) / per_cu - 512;  // inflated request: no more than `cap` fit a CU
        if (lds_req < lds) lds_req = lds;
    }
    // short frames staged through LDS and the prefetching variant keep a persistent grid; the rest
    // launch one workgroup per few frame-blocks (see grid_for)
    double iters = (staged || PREFETCH || !lds) ? 0.0 : table_iters(tw_entries, sizeof(C<T>));
    if (iters > 0 && BT > block_threads<NT>()) iters = 1.0;  // eight frames per table already (2.89 ms against 2.93 with two frame-blocks, 3.20 with four)
    {
        const char *env = getenv("DSPMI_FFT_ITERS");  // experiment: frame-blocks per workgroup
        if (env && atof(env) > 0) iters = atof(env);
    }
    const size_t grid = grid_for(nfb, (size_t)num_cu * per_cu, iters);
    int flags = backward ? kFftBackward : 0;
    {
        const char *env = getenv("DSPMI_FFT_FRONTS");  // A/B knob: 1, 2, 4 or 8 fronts
        const int fronts = env && atoi(env) > 0 ? atoi(env) : 2;
        if (iters > 0) flags |= (fronts >= 8 ? 3 : fronts >= 4 ? 2 : fronts >= 2 ? 1 : 0) << 1;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BT), lds_req, s, static_cast<const C<T> *>(in),
                       static_cast<C<T> *>(out), static_cast<const C<T> *>(tw), batch, flags);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
int launch_r2c(const void *in, void *out, const void *tw, const void *tw2n, size_t batch, int epilogue, unsigned nbin,
               int mode, float scale, int num_cu, hipStream_t s)
{
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr size_t lds = ((size_t)FPB * padded_len(N) + lds_twiddle_entries(TW, R0, R1, R2, R3, N)) * sizeof(C<T>) +
                           (size_t)FPB * (NT > 64 ? NT / 64 : 1) * sizeof(float);
    auto k1 = fft_r2c_kernel<T, N, NT, R0, R1, R2, R3, TW, 1>;
    auto k2 = fft_r2c_kernel<T, N, NT, R0, R1, R2, R3, TW, 2>;
    static dspmi::PerDeviceOnce attr;
    if (lds > 48 * 1024 && attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    if (batch == 0) return DSPMI_OK;
    const int pcm = (epilogue & dspmi::kR2cSourceS16x2) != 0;
    epilogue &= 0xff;
    const size_t nfb = (batch + FPB - 1) / FPB;
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 2048 / BT) per_cu = 2048 / BT;
    if (per_cu < 1) per_cu = 1;
    // frames inside one wave: as the complex kernel (n = 1024: 0.50 -> 0.43 ms per 2^28 samples); longer frames measured flat
    const size_t grid = grid_for(nfb, (size_t)num_cu * per_cu, NT <= 64 ? table_iters(lds_twiddle_entries(TW, R0, R1, R2, R3, N), sizeof(C<T>)) : 0.0);
    hipLaunchKernelGGL(epilogue == 1 ? k1 : k2, dim3((unsigned)grid), dim3(BT), lds, s, static_cast<const C<T> *>(in), out,
                       static_cast<const C<T> *>(tw), static_cast<const C<T> *>(tw2n), batch, nbin, mode, scale, pcm);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

template <typename T, int N, int NT, int R0, int R1, int R2, int R3, int TW>
int launch_c2r(const void *in, void *out, const void *tw, const void *tw2n, size_t batch, int num_cu, hipStream_t s)
{
    constexpr int BT = block_threads<NT>();
    constexpr int FPB = BT / NT;
    constexpr size_t lds = ((size_t)FPB * padded_len(N) + lds_twiddle_entries(TW, R0, R1, R2, R3, N)) * sizeof(C<T>);
    auto kern = fft_c2r_kernel<T, N, NT, R0, R1, R2, R3, TW>;
    static dspmi::PerDeviceOnce attr;
    if (lds > 48 * 1024 && attr.pending()) {
        DSPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    if (batch == 0) return DSPMI_OK;
    const size_t nfb = (batch + FPB - 1) / FPB;
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 2048 / BT) per_cu = 2048 / BT;
    if (per_cu < 1) per_cu = 1;
    // a few times the resident set for frames inside one wave (0.445 -> 0.42 ms); longer frames are fastest persistent
    const size_t grid = grid_for(nfb, (size_t)num_cu * per_cu, NT <= 64 ? (double)nfb / (4.0 * num_cu * per_cu) : 0.0);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BT), lds, s, static_cast<const C<T> *>(in), static_cast<C<T> *>(out),
                       static_cast<const C<T> *>(tw), static_cast<const C<T> *>(tw2n), batch);
    DSPMI_HIP(hipGetLastError());
    return DSPMI_OK;
}

typedef dspmi::FftKernel FftEntry;

// f32: twiddles in a per-workgroup LDS table (registers stay free for the butterflies: 4 waves
// per SIMD instead of 2); f64: twiddles read from the global table through L1.
#define F32(N, NT, A, B, C_, D)                                                                             \
    {N, launch_c2c<float, N, NT, A, B, C_, D, 2, false, 0>, launch_c2c<float, N, NT, A, B, C_, D, 2, false, 3>, \
     "fft_f32_n" #N "_r" #A "x" #B "x" #C_ "x" #D, launch_r2c<float, N, NT, A, B, C_, D, 2>, \
     launch_c2r<float, N, NT, A, B, C_, D, 2>}
// short frames (fewer than 8 lanes each): no fused real-input / real-output kernels -- those
// touch memory in pieces of 64 bytes or less there, and the complex kernel (which moves short
// frames as contiguous runs) followed by a split / merge kernel is several times faster
#define F32S(N, NT, A, B, C_, D)                                                                            \
    {N, launch_c2c<float, N, NT, A, B, C_, D, 2, false, 0>, launch_c2c<float, N, NT, A, B, C_, D, 2, false, 3>, \
     "fft_f32_n" #N "_r" #A "x" #B "x" #C_ "x" #D, nullptr, nullptr}
// long frames: full twiddle tables hold about N entries whatever the radix order -- as much LDS
// as the frame itself -- so the complex kernel builds its twiddles from two short tables and
// twice as many workgroups fit a CU (4096: 4.7 -> 5.2 TB/s, 8192: 3.6 -> 5.1, 16384: 2.8 -> 3.8).
// TWR is the table kind of the real-input / real-output kernels (split pays there from 8192 on).
#define F32T(N, NT, A, B, C_, D, TWR)                                                                       \
    {N, launch_c2c<float, N, NT, A, B, C_, D, 3, false, 0>, launch_c2c<float, N, NT, A, B, C_, D, 3, false, 3>, \
     "fft_f32_n" #N "_r" #A "x" #B "x" #C_ "x" #D "_splittw", launch_r2c<float, N, NT, A, B, C_, D, TWR>,     \
     launch_c2r<float, N, NT, A, B, C_, D, TWR>}
#define F64(N, NT, A, B, C_, D)                                                                               \
    {N, launch_c2c<double, N, NT, A, B, C_, D, 0, false, 0>, launch_c2c<double, N, NT, A, B, C_, D, 0, false, 0>, \
     "fft_f64_n" #N "_r" #A "x" #B "x" #C_ "x" #D, nullptr, nullptr}

// the whole frame in registers (fft_c2c_reg3_kernel): the long single-precision frames, and double
// precision from 1024 points on (4.4-4.8 -> 5.7-5.8 TB/s; 8192 points 3.2 -> 5.7)
#define REG3(T, TN, P, RC) \
    {P * P * RC, launch_c2c_reg3<T, P, RC, 0>, launch_c2c_reg3<T, P, RC, 3>, "fft_" TN "_n" REG3_STR(P, RC) "_reg" #P "x" #P "x" #RC, nullptr, nullptr}
#define REG3_STR(P, RC) REG3_N_##P##_##RC
#define REG3_N_16_4 "1024"
#define REG3_N_16_8 "2048"
#define REG3_N_16_16 "4096"
#define REG3_N_32_8 "8192"
#define REG3_N_32_16 "16384"
#define REG3_N_32_32 "32768"

const FftEntry kF32[] = {
    F32S(2, 1, 2, 1, 1, 1),         F32S(4, 1, 4, 1, 1, 1),         F32S(8, 1, 8, 1, 1, 1),
    F32S(16, 1, 16, 1, 1, 1),       F32S(32, 2, 16, 2, 1, 1),       F32S(64, 4, 16, 4, 1, 1),
    F32(128, 8, 16, 8, 1, 1),       F32(256, 16, 16, 16, 1, 1),     F32(512, 32, 16, 16, 2, 1),
    F32(1024, 64, 16, 16, 4, 1),    F32T(2048, 128, 16, 16, 8, 1, 2), F32T(4096, 256, 16, 16, 16, 1, 2),
    F32T(8192, 512, 16, 16, 16, 2, 3),
    REG3(float, "f32", 32, 16),     REG3(float, "f32", 32, 32),
};
// the 16384-point transform as one frame image exchanged three times (A/B: DSPMI_FFT_VARIANT=1 at N = 16384)
const FftEntry kF32Alt16384 = {16384, launch_c2c<float, 16384, 1024, 16, 16, 16, 4, 3, false, 0>,
                               launch_c2c<float, 16384, 1024, 16, 16, 16, 4, 3, false, 3>, "fft_f32_n16384_r16x16x16x4_splittw", nullptr, nullptr};
const FftEntry kF64[] = {
    F64(2, 1, 2, 1, 1, 1),        F64(4, 1, 4, 1, 1, 1),         F64(8, 1, 8, 1, 1, 1),
    F64(16, 2, 8, 2, 1, 1),       F64(32, 4, 8, 4, 1, 1),        F64(64, 8, 8, 8, 1, 1),
    F64(128, 16, 8, 8, 2, 1),     F64(256, 32, 8, 8, 4, 1),      F64(512, 64, 8, 8, 8, 1),
    REG3(double, "f64", 16, 4),   REG3(double, "f64", 16, 8),    REG3(double, "f64", 16, 16),
    REG3(double, "f64", 32, 8),   REG3(double, "f64", 32, 16),
};
// the same lengths with the frame in an LDS image (A/B: DSPMI_FFT_VARIANT=1)
const FftEntry kF64Alt[] = {
    F64(1024, 128, 8, 8, 8, 2),   F64(2048, 256, 8, 8, 8, 4),    F64(4096, 512, 8, 8, 8, 8),
    F64(8192, 512, 16, 16, 16, 2),
};

// alternative kernels for A/B runs: DSPMI_FFT_VARIANT=<index>
const FftEntry kF32Alt1024[] = {
    {1024, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 1, false>, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 1, false>, "fft_f32_n1024_r16x16x4_twreg", nullptr, nullptr},
    {1024, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 2, false>, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 2, false, 3>, "fft_f32_n1024_r16x16x4_twlds_nopf", nullptr, nullptr},
    {1024, launch_c2c<float, 1024, 64, 16, 8, 8, 1, 2, false>, launch_c2c<float, 1024, 64, 16, 8, 8, 1, 2, false, 3>, "fft_f32_n1024_r16x8x8_twlds", nullptr, nullptr},
    {1024, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 0, false>, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 0, false, 3>, "fft_f32_n1024_r16x16x4_twglobal", nullptr, nullptr},
    // persistent grid, next frame's loads issued before the current frame is computed, two workgroups per CU
    {1024, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 2, true>, launch_c2c<float, 1024, 64, 16, 16, 4, 1, 2, true, 3>, "fft_f32_n1024_r16x16x4_pf", nullptr, nullptr},
};

const FftEntry *find_entry(unsigned n, int precision)
{
    if (n == 1024 && precision == DSPMI_F32) {
        const char *env = getenv("DSPMI_FFT_VARIANT");
        if (env && atoi(env) >= 1 && atoi(env) <= 5) return &kF32Alt1024[atoi(env) - 1];
    }
    if (n == 16384 && precision == DSPMI_F32) {
        const char *env = getenv("DSPMI_FFT_VARIANT");
        if (env && atoi(env) == 1) return &kF32Alt16384;
    }
    if (precision == DSPMI_F64) {
        const char *env = getenv("DSPMI_FFT_VARIANT");
        if (env && atoi(env) == 1)
            for (const FftEntry &e : kF64Alt)
                if (e.n == n) return &e;
    }
    const FftEntry *tab = precision == DSPMI_F64 ? kF64 : kF32;
    const size_t cnt = precision == DSPMI_F64 ? sizeof(kF64) / sizeof(kF64[0]) : sizeof(kF32) / sizeof(kF32[0]);
    for (size_t i = 0; i < cnt; i++)
        if (tab[i].n == n) return &tab[i];
    return nullptr;
}

}  // namespace

namespace dspmi {

int fir_spectrum_launch(const FirSpectrumLaunch &a, hipStream_t s)
{
    FirParams<float> p{};
    p.x = a.x;
    p.hist = a.hist;
    p.taps = a.taps;
    p.x_stride = a.x_stride;
    p.hist_stride = a.hist_stride;
    p.nxx = p.nx = a.nx;
    p.nh = a.nh;
    p.tiles_per_chan = a.nx / kSplitTile;
    const unsigned long long tiles = (unsigned long long)p.tiles_per_chan * a.nchan;
    if (tiles == 0) return DSPMI_OK;
    if (tiles > 0x7fffffffull) return fail(DSPMI_EINVAL, "fir_spectrum: %llu frames exceed the grid limit", tiles);
    p.ntiles_total = (unsigned)tiles;
    unsigned tpb = (unsigned)(tiles / 8192);
    p.tiles_per_block = tpb < 2 ? (tiles >= 2048 ? 2 : 1) : (tpb > 16 ? 16 : tpb);
    if (const char *env = getenv("DSPMI_FIR_TPB"))
        if (atoi(env) > 0) p.tiles_per_block = (unsigned)atoi(env);
    const unsigned blocks = (unsigned)((tiles + p.tiles_per_block - 1) / p.tiles_per_block);
    switch (a.slices) {
    case 1: return launch_fir_spectrum_n<1>(p, a, blocks, s);
    case 2: return launch_fir_spectrum_n<2>(p, a, blocks, s);
    case 3: return launch_fir_spectrum_n<3>(p, a, blocks, s);
    case 4: return launch_fir_spectrum_n<4>(p, a, blocks, s);
    case 5: return launch_fir_spectrum_n<5>(p, a, blocks, s);
    case 6: return launch_fir_spectrum_n<6>(p, a, blocks, s);
    case 7: return launch_fir_spectrum_n<7>(p, a, blocks, s);
    case 8: return launch_fir_spectrum_n<8>(p, a, blocks, s);
    case 9: return launch_fir_spectrum_n<9>(p, a, blocks, s);
    case 10: return launch_fir_spectrum_n<10>(p, a, blocks, s);
    }
    return fail(DSPMI_EINVAL, "fir_spectrum: no kernel of %d slices", a.slices);
}


const FftKernel *fft_find(unsigned n, int precision) { return find_entry(n, precision); }

fft_bundle_fn fft_bundle_find(unsigned n, int precision)
{
    if (precision == DSPMI_F64) switch (n) {
        case 128: return launch_c2c_colreg<double, 8, 16>;
        case 256: return launch_c2c_colreg<double, 16, 16>;
        case 512: return launch_c2c_colreg<double, 32, 16>;
        case 1024: return launch_c2c_colreg<double, 32, 32>;
        default: return nullptr;
        }
    switch (n) {
    case 128: return launch_c2c_colreg<float, 8, 16>;
    case 256: return launch_c2c_colreg<float, 16, 16>;
    case 512: return launch_c2c_colreg<float, 32, 16>;
    case 1024: return launch_c2c_colreg<float, 32, 32>;
    }
    return nullptr;
}

bluestein_launch_fn bluestein_find(unsigned m, int precision)
{
    if (precision == DSPMI_F64) switch (m) {
        case 512: return launch_bluestein_reg3<double, 16, 2>;
        case 1024: return launch_bluestein_reg3<double, 16, 4>;
        case 2048: return launch_bluestein_reg3<double, 16, 8>;
        case 4096: return launch_bluestein_reg3<double, 16, 16>;
        case 8192: return launch_bluestein_reg3<double, 32, 8>;
        case 16384: return launch_bluestein_reg3<double, 32, 16>;
        default: return nullptr;
        }
    if (precision != DSPMI_F32) return nullptr;
    if (m == 32768) return launch_bluestein_reg3<float, 32, 32>;  // (no frame image of 32768 complex points fits the LDS)
    for (const BluesteinEntry &e : kBluesteinF32)
        if (e.m == m) return e.fn;
    return nullptr;
}

ols_launch_fn ols_find_real(unsigned block)
{
    return block == 32768 ? launch_ols_real<32> : block == 16384 ? launch_ols_real<16> : nullptr;
}

ols_launch_fn ols_find(unsigned n)
{
    for (const OlsEntry &e : kOls)
        if (e.n == n) return e.fn;
    return nullptr;
}

}  // namespace dspmi
