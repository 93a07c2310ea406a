// rng_dev.h - Philox4x32-10 counter-based generator and the short branch-free FP64 Box-Muller
// kernels shared by the a_lm draw (draw.hip) and the flat-sky field generator (flatsky.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// ------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator -> N(0,1) by Box-Muller, stream order
// ------------------------------------------------------------------------------------
// xor of three words in one instruction (v_bitop3_b32, truth table 0x96)
__device__ static inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

__device__ static inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint64_t p0 = (uint64_t)M0 * c[0];
    const uint64_t p1 = (uint64_t)M1 * c[2];
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c[1], k0);
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c[3], k1);
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}

__device__ static inline void philox4x32_10(uint64_t counter, uint64_t key, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}

// Box-Muller in FP64 with short, branch-free kernels (the generic libm log/sincospi cost ~130 DP instructions
// per pair and made the RNG, not the MFMAs, the bound of K3): every step is accurate to ~1e-16.
//   ln x   : x = 2^e m, m in [1/sqrt2, sqrt2); c = round(64 m)/64; r = m (1/c) - 1 (one fma, |r| <= 0.0111);
//            ln x = e ln2 - ln(1/c) + log1p(r), log1p by its degree-8 Taylor polynomial (remainder 3e-17 r).
//            c = 1 is a centre, so x -> 1 keeps full RELATIVE accuracy (the radius there is sqrt(-2 ln x)).
//   sqrt t : v_rsq_f64 seed + two Goldschmidt steps + one residual correction.
//   sin/cos(2 pi u): octant from the top three bits of u (exact), argument in [0, pi/4], the classic
//            degree-13/14 minimax kernels (Sun fdlibm k_sin/k_cos coefficients; max error 1.1e-16).
__device__ static const double2 LOG_TAB[47] = {
#include "log_tab.inc"
};

// `tab`: the 47-entry table, LOG_TAB itself or a copy of it in LDS (K3: a look-up in global memory is a dependent
// vmcnt-ordered load on the critical path of every normal pair - and waits for the LDS-DMA stage issued before it)
__device__ static inline double fast_log01(double x, const double2 *tab = LOG_TAB) {  // x in (0, 1]
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                  // [1/sqrt2, sqrt2)
    e = lo ? e - 1 : e;
    const int i = (int)__builtin_rint(m * 64.0);         // 45 .. 91
    const double2 tc = tab[i - 45];
    const double r = fma(m, tc.x, -1.0);
    double p = -1.0 / 8.0;
    p = fma(p, r, 1.0 / 7.0);
    p = fma(p, r, -1.0 / 6.0);
    p = fma(p, r, 1.0 / 5.0);
    p = fma(p, r, -1.0 / 4.0);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -1.0 / 2.0);
    p = fma(p * r, r, r);                                // log1p(r)
    return fma((double)e, 0.69314718055994530942, tc.y + p);
}

__device__ static inline double fast_sqrt_pos(double t) {  // t in [0, ~80]
    const double y = __builtin_amdgcn_rsq(t);
    double g = t * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double dd = fma(-g, g, t);
    g = fma(dd, h, g);
    return t > 0.0 ? g : 0.0;
}

// cos(2 pi u), sin(2 pi u) for u in (0, 1)
__device__ static inline void fast_sincos2pi(double u, double &sn, double &cs) {
    const double a = 8.0 * u;                 // exact
    const int q = (int)a;                     // octant 0..7 (a < 8 always: u <= 1 - 2^-54 rounds to at most 1.0 ...)
    const double f = a - (double)q;           // exact, [0, 1)
    const double g = (q & 1) ? 1.0 - f : f;   // reflect odd octants
    const double x = g * 0.78539816339744830962;
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const bool swap = ((q + 1) >> 1) & 1;     // octants 1,2,5,6
    const double cc = swap ? s : c, ss = swap ? c : s;
    cs = (((q + 2) >> 2) & 1) ? -cc : cc;     // octants 2..5
    sn = (q & 4) ? -ss : ss;                  // octants 4..7
}

// ------------------------------------------------------------------------------------
// The device normal stream (specification: oracle/philox.py).  One Philox block (r0, r1, r2, r3) gives two N(0,1):
//     k = r0 2^20 + (r1 >> 12)                     (52 bits)      u1 = (k + 1/2) 2^-52  in (0, 1), exact in a double
//     j = r2 >> 24,  w = (r2 & 0xffffff) 2^28 + (r3 >> 4)  (52 bits)  theta = 2 pi (j + (w + 1/2) 2^-52) / 256
//     (sqrt(-2 ln u1) cos theta, sqrt(-2 ln u1) sin theta)
// Every step works on the BITS of the uniforms instead of converting, splitting and reducing doubles (the round-2
// form of this chain - u = (k53 + 0.5) 2^-53 by cvt/ldexp/add, frexp + rint for the log table, octant reduction and
// two degree-13 kernels for the angle - was ~170 VALU instructions per pair and, next to FP64 MFMAs that share the
// issue slots, 40 % of K3; this one is ~85):
//   u1    : 1 + k 2^-52 by OR-ing the exponent of 1.0 over the bits, minus (1 - 2^-53): exact (Sterbenz).
//   ln u1 : top 8 mantissa bits ROUNDED select c = 1 + i/256 (i = 0..256); r = m (1/c) - 1 by one fma with the
//           tabulated rounded reciprocal, |r| < 2^-9; ln m = ln c + log1p(r), degree-5 Taylor (remainder r^6/6 < 1e-17).
//           Mantissas >= 1 + 107/256 (~sqrt2) are read as m/2 with the exponent one up: the table holds ln(c/2) for
//           them, entry 256 = (1/2, 0), so u1 -> 1 keeps full RELATIVE accuracy (the radius there is sqrt(2 (1 - u1))).
//           The table carries -2 ln, the polynomial the factor -2: the result is -2 ln u1 directly.
//   sqrt  : v_rsq_f64 seed (~2^-23), one Goldschmidt step, one residual correction (< 1 ulp).
//   theta : (cos, sin) of the centre of sector j from a 256-entry table, rotated by x = (w + 1/2 - 2^51) 2^-52 2 pi / 256,
//           |x| <= pi/256: sin x to x^5, cos x to x^6 (remainders 8e-18, 1e-20).
// `lg`, `sc`: RNG_LOG_TAB / RNG_SC_TAB or copies of them in LDS (K3: a table look-up in global memory is a dependent
// vmcnt-ordered load on the critical path of every pair - and waits for the LDS-DMA stage issued before it).
// ------------------------------------------------------------------------------------
#include "rng_tab.inc"

__device__ static inline double rng_bits_to_double(uint32_t hi, uint32_t lo) {
    return __hiloint2double((int)hi, (int)lo);
}

__device__ static inline double2 rng_boxmuller_bits(const uint32_t (&r)[4], const double2 *lg = RNG_LOG_TAB,
                                                    const double2 *sc = RNG_SC_TAB) {
    // ---- u1 and -2 ln u1
    const uint32_t k_lo = __builtin_amdgcn_alignbit(r[0], r[1], 12);             // (r0 << 20) | (r1 >> 12)
    const uint32_t k_hi = __builtin_amdgcn_alignbit(0x3FFu, r[0], 12);            // exponent of 1.0 | (r0 >> 12)
    const double u1 = rng_bits_to_double(k_hi, k_lo) - (1.0 - 0x1p-53);           // (k + 1/2) 2^-52, exact
    const uint32_t uh = (uint32_t)__double2hiint(u1);
    const uint32_t mh = uh & 0xFFFFFu;
    const uint32_t idx = (mh + 0x800u) >> 12;                                     // 0 .. 256
    const double m = rng_bits_to_double(mh | 0x3FF00000u, (uint32_t)__double2loint(u1));   // mantissa in [1, 2)
    const int e = (int)(uh >> 20) - 1023 + (idx > RNG_LOG_SPLIT ? 1 : 0);
    const double2 tc = lg[idx];
    const double rr = fma(m, tc.x, -1.0);
    double q = fma(rr, -2.0 / 5.0, 0.5);
    q = fma(q, rr, -2.0 / 3.0);
    q = fma(q, rr, 1.0);                                                          // -2 (log1p(r) - r) / r^2
    const double t = fma((double)e, -2.0 * 0.69314718055994530942, fma(rr * rr, q, fma(rr, -2.0, tc.y)));   // -2 ln u1 > 0
    // ---- radius
    const double y = __builtin_amdgcn_rsq(t);
    double g = t * y, h = 0.5 * y;
    const double e0 = fma(-h, g, 0.5);
    g = fma(g, e0, g);
    h = fma(h, e0, h);
    const double rad = fma(fma(-g, g, t), h, g);
    // ---- angle
    const double2 cs0 = sc[r[2] >> 24];
    const uint32_t w_lo = __builtin_amdgcn_alignbit(r[2], r[3], 4);               // (r2 << 28) | (r3 >> 4)
    const uint32_t w_hi = ((r[2] >> 4) & 0xFFFFFu) | 0x3FF00000u;
    const double tw = rng_bits_to_double(w_hi, w_lo) - 1.5;                       // w 2^-52 - 1/2, exact
    const double x = fma(tw, 6.283185307179586476925 / 256.0, 0x1p-53 * (6.283185307179586476925 / 256.0));
    const double z = x * x;
    const double sx = fma(x * z, fma(z, 1.0 / 120.0, -1.0 / 6.0), x);
    const double cx = fma(z, fma(fma(z, -1.0 / 720.0, 1.0 / 24.0), z, -0.5), 1.0);
    const double a = rad * cs0.x, b = rad * cs0.y;
    return make_double2(fma(a, cx, -(b * sx)), fma(a, sx, b * cx));
}

// two independent N(0,1) from one Philox block
__device__ static inline double2 philox_boxmuller(uint64_t counter, uint64_t key, const double2 *lg = RNG_LOG_TAB,
                                                  const double2 *sc = RNG_SC_TAB) {
    uint32_t r[4];
    philox4x32_10(counter, key, r);
    return rng_boxmuller_bits(r, lg, sc);
}
