// Arithmetic of the tone generator shared by its kernels (spectrum.hip) and by the fused
// generator + FIR + spectrum kernel (fft_stockham.hip): sin / cos of a running angle, and the
// running angle of tonegen/tonegen.c:25-40 after any number of its rounded additions.
#pragma once

#include <hip/hip_runtime.h>

namespace {

// sin / cos of a running angle.  The reference evaluates libm's sin/cos on the accumulated,
// never-reduced angle (tonegen/tonegen.c:36-37; it reaches ~1e6 rad after a million samples),
// where a generic device sin falls into its slow large-argument path.  Here the angle is
// reduced by multiples of pi/2 with a three-constant Cody-Waite step (each fused step leaves a
// multiple of 2^-52 below 2, so it is exact; what remains is the ~1e-16 of the constants), and one
// minimax polynomial in t^2 -- its coefficients chosen by the quadrant's parity -- gives
// sin(t) or cos(t) on [-pi/4, pi/4].  The value agrees with libm's to about an ulp of double,
// far below the f32 (or 16-bit) sample it ends up in.
template <bool COS>
__device__ __forceinline__ double tone_trig(double w)
{
    const double two_over_pi = 0.63661977236758134308;
    const double c1 = 1.57079632679489655800e+00;   // pi/2, leading 53 bits
    const double c2 = 6.12323399573676603587e-17;   // next bits
    const double c3 = -1.49738490485916983145e-33;  // and the next
    const double k = rint(w * two_over_pi);
    double t = fma(-k, c1, w);
    t = fma(-k, c2, t);
    t = fma(-k, c3, t);
    int q = (int)(long long)k;
    if (COS) q += 1;  // cos(w) = sin(w + pi/2)
    const bool even = (q & 1) != 0;  // odd quadrant count: the even function (cosine) of t
    const double z = t * t;
    // sin(t) = t + t*z*S(z), cos(t) = 1 - z/2 + z*z*C(z)   (coefficients of fdlibm's kernels)
    double p = even ? -1.13596475577881948265e-11 : 1.58969099521155010221e-10;
    p = fma(p, z, even ? 2.08757232129817482790e-09 : -2.50507602534068634195e-08);
    p = fma(p, z, even ? -2.75573143513906633035e-07 : 2.75573137070700676789e-06);
    p = fma(p, z, even ? 2.48015872894767294178e-05 : -1.98412698298579493134e-04);
    p = fma(p, z, even ? -1.38888888888741095749e-03 : 8.33333333332248946124e-03);
    p = fma(p, z, even ? 4.16666666666666019037e-02 : -1.66666666666666324348e-01);
    const double odd_val = fma(t * z, p, t);
    const double even_val = fma(z * z, p, fma(-0.5, z, 1.0));
    const double v = even ? even_val : odd_val;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ double tone_sin(double w) { return tone_trig<false>(w); }
__device__ __forceinline__ double tone_cos(double w) { return tone_trig<true>(w); }

// ---- the same generator, parallel in time ---------------------------------------------
// The recurrence w = fl(w + dw) is sequential, but its rounding is regular: while w stays in one
// binade [2^e, 2^(e+1)) it is a multiple M*u of that binade's unit u = 2^(e-52), and adding dw
// then rounds to M + D with one fixed integer D (dw/u rounded to nearest; on an exact tie the
// result is the even neighbour, which is again a fixed step once M is even).  So the additions
// inside a binade collapse into one integer multiply-add, and only the additions that cross into
// the next binade are performed for real.  advance() returns bit for bit what `steps` sequential
// additions produce, in a number of iterations proportional to the binades crossed, and every
// range of the signal can start from its own exactly known angle.
// The run of additions that stays on one binade's grid: starting from a (>= 0) with increment
// d (> 0), a_k = a + k*step exactly for every k <= n.  n == 0 means the next addition has to be
// performed for real (it changes the binade, or starts from zero / a subnormal / an odd multiple
// on an exact tie); step == 0 with n = all ones means the angle no longer moves.
__device__ __forceinline__ unsigned long long fixed_run(double a, double d, double &step)
{
    constexpr unsigned long long kMant = (1ull << 52) - 1, kOne = 1ull << 52;
    step = 0.0;
    const unsigned long long id = (unsigned long long)__double_as_longlong(d);
    const int ed = (int)(id >> 52);  // d > 0: sign bit clear
    const unsigned long long md = ed ? ((id & kMant) | kOne) : (id & kMant);
    const unsigned long long ia = (unsigned long long)__double_as_longlong(a);
    const int ea = (int)(ia >> 52);
    const int sh = ea - (ed ? ed : 1);  // d / u = md / 2^sh, u = the unit of a's binade
    if (ea == 0 || ea == 0x7ff || sh <= 0) return 0;
    if (sh >= 54) return ~0ull;  // d < u/2
    const unsigned long long M = (ia & kMant) | kOne;
    const unsigned long long q = md >> sh, r = md & ((1ull << sh) - 1), half = 1ull << (sh - 1);
    unsigned long long D;
    if (r > half) D = q + 1;
    else if (r < half) D = q;
    else if (M & 1) return 0;  // tie from an odd multiple: one real addition makes it even
    else D = q + (q & 1);
    if (D == 0) return ~0ull;
    // additions whose exact sum stays below 2^(e+1) round on this binade's grid
    const unsigned long long lim = (1ull << 53) - 1 - q;
    if (M > lim) return 0;
    step = ldexp((double)D, ea - 1075);  // D * u, exact
    return (lim - M) / D + 1;
}

__device__ __forceinline__ double advance(double a, double d, unsigned long long steps)
{
    if (!(a >= 0.0) || !(d > 0.0)) {
        // negative or NaN operands: no shortcut, walk the additions
        for (; steps; steps--) a += d;
        return a;
    }
    while (steps) {
        double step;
        unsigned long long n = fixed_run(a, d, step);
        if (n == 0) {
            a += d;
            steps--;
            continue;
        }
        if (step == 0.0) return a;
        if (n > steps) n = steps;
        a = fma((double)n, step, a);  // the exact value is representable (at most 2^53 units): no rounding
        steps -= n;
    }
    return a;
}

}  // namespace
