// fft_ct.h - the compile-time FFT passes shared by the ring transforms (sht_ringfft_ct.hip) and the flat-sky line
// transforms (flatsky_ct.hip): LDS padding, pass schedules, register butterflies, twiddle application, one in-LDS pass.
#pragma once
#include "sht_internal.h"

#ifndef CT_ABLATE_TW
#define CT_ABLATE_TW 0
#endif

// LDS padding of a channel buffer, per kernel family (PK, a template parameter of everything below that touches a
// buffer): 0 = one spare 16-byte slot per 8 elements plus 8 per 128 (rounds 1-3), 1 = one spare slot per 16 elements.
// In the lane-group simulation of ds_read_b128 / ds_write_b128 (tools/lds_bank_sim.py) padding 0 makes every read of
// a pass 2-way conflicted (a padded unit-stride run is no longer aligned to the bank rows the read groups assume);
// padding 1 leaves only the strided first / last passes so.  Measured: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the
// Bluestein classes 0.32 -> 0.12 (tools/pmc_lds.sh), their time -1 .. -5 %; the belt (direct class: radix-8 last pass fused
// with the pixel store) reads 0.18 -> 0.27 and +3 % with it and keeps padding 0.
template <int PK>
__host__ __device__ constexpr int fpk(int i) {
    return PK ? i + (i >> 4) : i + (i >> 3) + ((i >> 7) << 3);
}
#define fpad(i) fpk<PK>(i)
#define fpc(i) fpk<PK>(i)
#ifndef K5_PK_BLU
#define K5_PK_BLU 1
#endif
#ifndef K5_PK_DIRECT
#define K5_PK_DIRECT 0
#endif

template <int N>
struct Sch;   // DIF radices of a three-pass transform, largest stride first
template <>
struct Sch<1024> {
    static constexpr int R0 = 16, R1 = 16, R2 = 4;
};
template <>
struct Sch<2048> {
    static constexpr int R0 = 16, R1 = 16, R2 = 8;
};
template <>
struct Sch<4096> {
    static constexpr int R0 = 16, R1 = 16, R2 = 16;
};
// 256 and 512 (flat-sky lines only: linec2r_ct below): 16 x 16 and 16 x 16 x 2
template <>
struct Sch<256> {
    static constexpr int R0 = 16, R1 = 16, R2 = 1;
};
template <>
struct Sch<512> {
    static constexpr int R0 = 16, R1 = 16, R2 = 2;
};
// 3 * 2^k flat-sky lengths (384^3, 768^3 cubes): radix 12 in the first pass, as Sch<1536> / Sch<3072>
template <>
struct Sch<192> {
    static constexpr int R0 = 12, R1 = 16, R2 = 1;
};
template <>
struct Sch<384> {
    static constexpr int R0 = 12, R1 = 16, R2 = 2;
};
template <>
struct Sch<768> {
    static constexpr int R0 = 12, R1 = 16, R2 = 4;
};
// 5 * 2^k (flat-sky Bluestein lengths: 2 n - 1 <= 320 / 640 / 1280): radix 10 in the first pass, as Sch<2560>
template <>
struct Sch<320> {
    static constexpr int R0 = 10, R1 = 16, R2 = 2;
};
template <>
struct Sch<640> {
    static constexpr int R0 = 10, R1 = 16, R2 = 4;
};
template <>
struct Sch<1280> {
    static constexpr int R0 = 10, R1 = 16, R2 = 8;
};
template <int SIGN>
struct DftR<1, SIGN> {
    __device__ __forceinline__ static void run(double2 (&)[1]) {}
};
// 3 * 2^k: the factor 3 sits in the first pass (radix 12), whose stride P / 12 is a power of two, so every
// butterfly address is still base + immediates; the Bluestein length of a ring is then at most 1.5 (not 2) times
// 2 h - 1
template <>
struct Sch<3072> {
    static constexpr int R0 = 12, R1 = 16, R2 = 16;
};
template <>
struct Sch<1536> {
    static constexpr int R0 = 12, R1 = 16, R2 = 8;
};

// 5 * 2^9 and 7 * 2^9 (round 4): of the rings whose 2 h - 1 needs more than 2048, those up to 2560 / between 3072 and
// 3584 get a length 17 % / 12.5 % shorter than 3072 / 4096.  Radix 10 / 14 in the first, strided pass (stride 256)
template <>
struct Sch<2560> {
    static constexpr int R0 = 10, R1 = 16, R2 = 16;
};
template <>
struct Sch<3584> {
    static constexpr int R0 = 14, R1 = 16, R2 = 16;
};

// cfg-5 geometry (nside 2048): the cap rings 1025 .. 2047 need Bluestein lengths above 4096, one channel per workgroup
// (152 KB).  Three passes with the large radix (32, 24 = 3 * 8) in the FIRST, strided pass - half of whose inputs are
// the zero padding, as half of the last inverse pass's outputs do not exist - and the register-fused middle stage at
// radix 16 like the shorter lengths (a radix-32 middle stage holds 32 filter values on top of its 32 points: spills)
template <>
struct Sch<8192> {
    static constexpr int R0 = 32, R1 = 16, R2 = 16;
};
template <>
struct Sch<6144> {
    static constexpr int R0 = 24, R1 = 16, R2 = 16;
};

// 12-point DFT, natural order in and out: n = 3 a + c, k = k1 + 4 k2: DFT4 over a, twiddle w12^{c k1}, DFT3 over c
template <int SIGN>
struct DftR<12, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[12]) {
        const double h3 = 0.86602540378443864676;   // sqrt(3) / 2
#pragma unroll
        for (int c = 0; c < 3; c++) dft4<SIGN>(x[c], x[3 + c], x[6 + c], x[9 + c]);
        // now x[3 k1 + c] = t_c[k1]; twiddles w12^{c k1}: c = 1: w1, w2, w3 = SIGN i;  c = 2: w2, w4, w6 = -1
        const double2 w1 = make_double2(h3, SIGN * 0.5), w2 = make_double2(0.5, SIGN * h3), w4 = make_double2(-0.5, SIGN * h3);
        x[3 + 1] = cmul(x[3 + 1], w1);
        x[6 + 1] = cmul(x[6 + 1], w2);
        x[9 + 1] = cmuli<SIGN>(x[9 + 1]);
        x[3 + 2] = cmul(x[3 + 2], w2);
        x[6 + 2] = cmul(x[6 + 2], w4);
        x[9 + 2] = make_double2(-x[9 + 2].x, -x[9 + 2].y);
        double2 y[12];
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) {
            const double2 a = x[3 * k1], b = x[3 * k1 + 1], c = x[3 * k1 + 2];
            const double2 sm = cadd(b, c), d = csub(b, c);
            const double2 m = make_double2(a.x - 0.5 * sm.x, a.y - 0.5 * sm.y);
            const double2 n = cmuli<SIGN>(make_double2(h3 * d.x, h3 * d.y));
            y[k1] = cadd(a, sm);
            y[k1 + 4] = cadd(m, n);
            y[k1 + 8] = csub(m, n);
        }
#pragma unroll
        for (int k = 0; k < 12; k++) x[k] = y[k];
    }
};

// 5- and 7-point DFTs (w = e^{SIGN 2 pi i / R}) through the sums / differences of the pairs (j, R - j): X_k = a_k + SIGN i b_k,
// X_{R-k} = a_k - SIGN i b_k with a_k = x_0 + sum_j cos(2 pi j k / R) (x_j + x_{R-j}), b_k = sum_j sin(2 pi j k / R) (x_j - x_{R-j})
template <int SIGN>
__device__ __forceinline__ static void dft5(double2 (&x)[5]) {
    constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi / 5), sin(4 pi / 5)
    const double2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    const double2 a1 = make_double2(fma(c2, t2.x, fma(c1, t1.x, x[0].x)), fma(c2, t2.y, fma(c1, t1.y, x[0].y)));
    const double2 a2 = make_double2(fma(c1, t2.x, fma(c2, t1.x, x[0].x)), fma(c1, t2.y, fma(c2, t1.y, x[0].y)));
    const double2 b1 = make_double2(fma(s2, t4.x, s1 * t3.x), fma(s2, t4.y, s1 * t3.y));
    const double2 b2 = make_double2(fma(-s1, t4.x, s2 * t3.x), fma(-s1, t4.y, s2 * t3.y));
    const double2 ib1 = cmuli<SIGN>(b1), ib2 = cmuli<SIGN>(b2);
    x[0] = cadd(x[0], cadd(t1, t2));
    x[1] = cadd(a1, ib1);
    x[4] = csub(a1, ib1);
    x[2] = cadd(a2, ib2);
    x[3] = csub(a2, ib2);
}
template <int SIGN>
__device__ __forceinline__ static void dft7(double2 (&x)[7]) {
    constexpr double c[4] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624};   // cos(2 pi k / 7)
    constexpr double sn[4] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048};  // sin(2 pi k / 7)
    const double2 t[4] = {x[0], cadd(x[1], x[6]), cadd(x[2], x[5]), cadd(x[3], x[4])};
    const double2 d[4] = {x[0], csub(x[1], x[6]), csub(x[2], x[5]), csub(x[3], x[4])};
    double2 y[7];
    y[0] = cadd(cadd(x[0], t[1]), cadd(t[2], t[3]));
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        double2 a = x[0], b = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            const int q = (j * k) % 7;                       // cos(2 pi q / 7) = c[min(q, 7 - q)], sin = +- sn[min(q, 7 - q)]
            const double cc = c[q <= 3 ? q : 7 - q], ss = q <= 3 ? sn[q] : -sn[7 - q];
            a = make_double2(fma(cc, t[j].x, a.x), fma(cc, t[j].y, a.y));
            b = make_double2(fma(ss, d[j].x, b.x), fma(ss, d[j].y, b.y));
        }
        const double2 ib = cmuli<SIGN>(b);
        y[k] = cadd(a, ib);
        y[7 - k] = csub(a, ib);
    }
#pragma unroll
    for (int k = 0; k < 7; k++) x[k] = y[k];
}
// 10- and 14-point DFTs, natural order in and out, by the prime-factor map (no twiddles between the two stages):
// n = (R n1 + 2 n2) mod 2R, k = (R k1 + 2 (2^{-1} mod R) k2) mod 2R with R = 5 / 7 - DFT2 over n1, DFT_R over n2
template <int SIGN>
struct DftR<10, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[10]) {
        double2 u0[5], u1[5];
#pragma unroll
        for (int n2 = 0; n2 < 5; n2++) {
            const double2 a = x[(2 * n2) % 10], b = x[(5 + 2 * n2) % 10];
            u0[n2] = cadd(a, b);
            u1[n2] = csub(a, b);
        }
        dft5<SIGN>(u0);
        dft5<SIGN>(u1);
#pragma unroll
        for (int k2 = 0; k2 < 5; k2++) {
            x[(6 * k2) % 10] = u0[k2];
            x[(5 + 6 * k2) % 10] = u1[k2];
        }
    }
};
template <int SIGN>
struct DftR<14, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[14]) {
        double2 u0[7], u1[7];
#pragma unroll
        for (int n2 = 0; n2 < 7; n2++) {
            const double2 a = x[(2 * n2) % 14], b = x[(7 + 2 * n2) % 14];
            u0[n2] = cadd(a, b);
            u1[n2] = csub(a, b);
        }
        dft7<SIGN>(u0);
        dft7<SIGN>(u1);
#pragma unroll
        for (int k2 = 0; k2 < 7; k2++) {
            x[(8 * k2) % 14] = u0[k2];
            x[(7 + 8 * k2) % 14] = u1[k2];
        }
    }
};

// e^{i pi r / 16}, r < 16 (indices are compile-time after unrolling: these fold into immediates)
__device__ constexpr double kCos16[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                          0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785,
                                          0.0, -0.19509032201612826785, -0.38268343236508977173, -0.55557023301960222474,
                                          -0.70710678118654752440, -0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913};
__device__ constexpr double kSin16[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                          0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613, 0.98078528040323044913,
                                          1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                          0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};

// 32-point DFT, natural order in and out: DFT16 of the even and of the odd inputs, combined with w32^k = e^{SIGN i pi k / 16}
__device__ constexpr double kCos12[12] = {1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5,
                                          0.25881904510252076235, 0.0, -0.25881904510252076235, -0.5, -0.70710678118654752440,
                                          -0.86602540378443864676, -0.96592582628906828675};
__device__ constexpr double kSin12[12] = {0.0, 0.25881904510252076235, 0.5, 0.70710678118654752440, 0.86602540378443864676,
                                          0.96592582628906828675, 1.0, 0.96592582628906828675, 0.86602540378443864676,
                                          0.70710678118654752440, 0.5, 0.25881904510252076235};
template <int SIGN>
struct DftR<32, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[32]) {
        double2 e[16], o[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            e[k] = x[2 * k];
            o[k] = x[2 * k + 1];
        }
        DftR<16, SIGN>::run(e);
        DftR<16, SIGN>::run(o);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double2 t = k == 0 ? o[0] : cmul(o[k], make_double2(kCos16[k], SIGN * kSin16[k]));
            x[k] = cadd(e[k], t);
            x[k + 16] = csub(e[k], t);
        }
    }
};

// 24-point DFT, natural order in and out: n = 3 a + c, k = k1 + 8 k2: DFT8 over a, twiddle w24^{c k1}, DFT3 over c
template <int SIGN>
struct DftR<24, SIGN> {
    __device__ __forceinline__ static void run(double2 (&x)[24]) {
        const double h3 = 0.86602540378443864676;   // sqrt(3) / 2
        // cos, sin of pi k / 12, k = 0 .. 14 (w24^j = e^{SIGN i pi j / 12}; c k1 <= 14)
        constexpr double c12[15] = {1.0, 0.96592582628906828675, 0.86602540378443864676, 0.70710678118654752440, 0.5,
                                    0.25881904510252076235, 0.0, -0.25881904510252076235, -0.5, -0.70710678118654752440,
                                    -0.86602540378443864676, -0.96592582628906828675, -1.0, -0.96592582628906828675,
                                    -0.86602540378443864676};
        constexpr double s12[15] = {0.0, 0.25881904510252076235, 0.5, 0.70710678118654752440, 0.86602540378443864676,
                                    0.96592582628906828675, 1.0, 0.96592582628906828675, 0.86602540378443864676,
                                    0.70710678118654752440, 0.5, 0.25881904510252076235, 0.0, -0.25881904510252076235, -0.5};
        double2 t[3][8];
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int a = 0; a < 8; a++) t[c][a] = x[3 * a + c];
            DftR<8, SIGN>::run(t[c]);                  // t[c][k1]
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) {
            const double2 a = t[0][k1];
            const double2 b = k1 == 0 ? t[1][0] : cmul(t[1][k1], make_double2(c12[k1], SIGN * s12[k1]));
            const double2 c = k1 == 0 ? t[2][0] : cmul(t[2][k1], make_double2(c12[2 * k1], SIGN * s12[2 * k1]));
            const double2 sm = cadd(b, c), d = csub(b, c);
            const double2 m = make_double2(a.x - 0.5 * sm.x, a.y - 0.5 * sm.y);
            const double2 n = cmuli<SIGN>(make_double2(h3 * d.x, h3 * d.y));
            x[k1] = cadd(a, sm);
            x[k1 + 8] = cadd(m, n);
            x[k1 + 16] = csub(m, n);
        }
    }
};

__device__ __forceinline__ static double2 csqr(double2 a) {
    return make_double2(fma(a.x, a.x, -(a.y * a.y)), 2.0 * a.x * a.y);
}
__device__ __forceinline__ static double2 cconj(double2 a) { return make_double2(a.x, -a.y); }

// x[r] *= w^r, r = 1 .. R-1.  Powers of two by squaring (kept), every other power as w^(r - lowbit) * w^lowbit and
// applied at once, so that besides w, w^2, w^4, w^8 only one or two products are live at a time (a table of all
// powers cost 60 VGPRs and made the kernels spill).
// The base twiddle is loop-invariant in the persistent item loop; without the empty asm the compiler hoists all 15
// powers of every pass out of that loop and, having no registers for ~200 values, keeps them in scratch memory
// (reloaded with vmcnt-ordered loads behind the prefetch).  Recomputing them costs 11 complex multiplies per butterfly.
template <int R>
__device__ __forceinline__ static void tw_apply(double2 (&x)[R], double2 w1) {
    asm volatile("" : "+v"(w1.x), "+v"(w1.y));
    double2 w[R];
    w[1] = w1;
    x[1] = cmul(x[1], w1);
#pragma unroll
    for (int r = 2; r < R; r++) {
        const int lb = r & (-r);
#if CT_ABLATE_TW     // diagnostic (wrong results): no twiddle powers - what would a table of them be worth?
        w[r] = make_double2(w1.x + (double)r, w1.y);
#else
        w[r] = (lb == r) ? csqr(w[r >> 1]) : cmul(w[r - lb], w[lb]);
#endif
        x[r] = cmul(x[r], w[r]);
    }
}

// one in-LDS pass of a length-N transform on NCH channel buffers (channel c at sm + c BS), sub-length Ls, radix R.
// DIT = false: DFT then twiddle (decimation in frequency); true: twiddle then DFT.  w1 = e^{+2 pi i j / Ls} of this
// thread's j = tid mod (Ls / R) (the same for every butterfly the thread ever gets in this pass).
template <int PK, int N, int NCH, int BS, int Ls, int R, int SIGN, bool DIT, int T>
__device__ __forceinline__ static void ct_pass(double2 *sm, const double2 w1, const int tid) {
    constexpr int NB = N / R, Q = Ls / R, TOT = NCH * NB;
    constexpr int IT = (TOT + T - 1) / T;
    const double2 w = make_double2(w1.x, SIGN > 0 ? w1.y : -w1.y);
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int idx = tid + it * T;
        if ((TOT % T) != 0 && idx >= TOT) break;
        const int ch = idx / NB, t = idx - ch * NB;
        const int b = t / Q, j = t - b * Q;
        double2 *p = sm + ch * BS + fpad(b * Ls + j);
        double2 x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = p[fpc(r * Q)];
        if (DIT && Q > 1) tw_apply<R>(x, w);
        DftR<R, SIGN>::run(x);
        if (!DIT && Q > 1) tw_apply<R>(x, w);
#pragma unroll
        for (int r = 0; r < R; r++) p[fpc(r * Q)] = x[r];
    }
}

