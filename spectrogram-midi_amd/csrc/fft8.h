// 2048-point complex float64 FFT for 256 cooperating threads: Stockham autosort, radix 8 x 8 x 8 x 4, butterflies
// in registers, ONE 32 KB buffer in LDS (every pass reads all its inputs, the workgroup synchronises, then writes its
// outputs in place).  Each function below is one thread's share of one phase, so the same code runs on the host with
// the 256 threads emulated in a loop (tools/fft8_host_check.cpp checks it against a direct DFT without a GPU).
//
// Pass with stride NS and radix R (T = 2048 / R butterflies): butterfly j takes in[j + q T], q < R, multiplies input q
// by exp(-2 pi i q k / (NS R)), k = j mod NS, transforms, and writes out[(j - k) R + k + r NS], r < R.
// The buffer is addressed through zsw(): an XOR swizzle of the low three index bits with bits 3..5, which keeps the
// strided writes of the first two passes (8 j + r, 64 a + k + 8 r) off a single bank group at no cost in space.
#pragma once
#include <hip/hip_runtime.h>

namespace aegis {

#define AEGIS_HD __host__ __device__ __forceinline__

AEGIS_HD int zsw(int i) { return i ^ ((i >> 3) & 7); }

AEGIS_HD double2 c_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
AEGIS_HD double2 c_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
AEGIS_HD double2 c_mul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// the same product with two fused multiply-adds (two roundings fewer, four instructions instead of six): the twiddle
// multiplications of the FFT passes, a quarter of their float64 work
AEGIS_HD double2 c_mul_f(double2 a, double2 b) {
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
AEGIS_HD double2 c_mul_mi(double2 a) { return make_double2(a.y, -a.x); }   // a * (-i)

// forward 4-point DFT, natural order in and out
AEGIS_HD void dft4(double2 &a0, double2 &a1, double2 &a2, double2 &a3) {
    const double2 s0 = c_add(a0, a2), s1 = c_sub(a0, a2), s2 = c_add(a1, a3), s3 = c_mul_mi(c_sub(a1, a3));
    a0 = c_add(s0, s2); a1 = c_add(s1, s3); a2 = c_sub(s0, s2); a3 = c_sub(s1, s3);
}

// forward 8-point DFT, natural order in and out: X[r] = sum_q a[q] w^(q r), w = exp(-2 pi i / 8)
AEGIS_HD void dft8(double2 (&a)[8]) {
    constexpr double c = 0.70710678118654752440;     // sqrt(1/2)
    double2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
    double2 o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    // w^1 = (c, -c), w^2 = -i, w^3 = (-c, -c)
    o1 = make_double2(c * (o1.x + o1.y), c * (o1.y - o1.x));
    o2 = c_mul_mi(o2);
    o3 = make_double2(c * (o3.y - o3.x), -(c * (o3.x + o3.y)));
    a[0] = c_add(e0, o0); a[4] = c_sub(e0, o0);
    a[1] = c_add(e1, o1); a[5] = c_sub(e1, o1);
    a[2] = c_add(e2, o2); a[6] = c_sub(e2, o2);
    a[3] = c_add(e3, o3); a[7] = c_sub(e3, o3);
}

// Twiddles of thread j (they depend on the thread only, never on the data): read once into registers.
struct Fft8Tw {
    double2 p2[7];      // pass NS = 8:   tw[q * 32 (j & 7)]
    double2 p3[7];      // pass NS = 64:  tw[q * 4 (j & 63)]
    double2 p4[2][3];   // pass NS = 512: butterflies j and j + 256, tw[q * (j + 256 b)]
};
// tw[m] = exp(-2 pi i m / 2048)
AEGIS_HD void fft8_load_twiddles(Fft8Tw &w, const double2 *tw, int j) {
#pragma unroll
    for (int q = 1; q < 8; ++q) { w.p2[q - 1] = tw[q * 32 * (j & 7)]; w.p3[q - 1] = tw[q * 4 * (j & 63)]; }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 1; q < 4; ++q) w.p4[b][q - 1] = tw[q * (j + 256 * b)];
}

// ---- one thread's share of each phase -------------------------------------------------------------------------------
// Pass 1 takes its inputs from registers (the caller builds v[q] = x[j + 256 q]) and needs no twiddles.
AEGIS_HD void fft8_pass1_write(double2 *z, int j, double2 (&v)[8]) {
    dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) z[zsw(8 * j + r)] = v[r];
}
AEGIS_HD void fft8_read8(const double2 *z, int j, double2 (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = z[zsw(j + 256 * q)];
}
template <int NS>
AEGIS_HD void fft8_pass_write(double2 *z, int j, double2 (&v)[8], const double2 (&w)[7]) {
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = c_mul_f(v[q], w[q - 1]);
    dft8(v);
    const int k = j & (NS - 1);
    const int j0 = ((j - k) << 3) + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) z[zsw(j0 + NS * r)] = v[r];
}
// Last pass (radix 4, NS = 512): butterflies j and j + 256 read and write the same four slots each, so no barrier
// separates its reads from its writes.
AEGIS_HD void fft8_pass4(double2 *z, int j, const Fft8Tw &w) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int jj = j + 256 * b;
        double2 a0 = z[zsw(jj)], a1 = z[zsw(jj + 512)], a2 = z[zsw(jj + 1024)], a3 = z[zsw(jj + 1536)];
        a1 = c_mul_f(a1, w.p4[b][0]); a2 = c_mul_f(a2, w.p4[b][1]); a3 = c_mul_f(a3, w.p4[b][2]);
        dft4(a0, a1, a2, a3);
        z[zsw(jj)] = a0; z[zsw(jj + 512)] = a1; z[zsw(jj + 1024)] = a2; z[zsw(jj + 1536)] = a3;
    }
}

// The last pass of the INVERSE transform of the frame kernel: only the outputs 1024 .. 1024 + max_lag are ever read
// (the autocorrelation at lags 0..max_lag of the two packed frames), i.e. output 2 of every butterfly and output 3 of the
// butterflies jj <= max_lag - 512.  Same values as fft8_pass4 at those positions; the other three quarters of the
// outputs are neither computed nor stored.
AEGIS_HD void fft8_pass4_lags(double2 *z, int j, const Fft8Tw &w, int max_lag) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int jj = j + 256 * b;
        double2 a0 = z[zsw(jj)], a1 = z[zsw(jj + 512)], a2 = z[zsw(jj + 1024)], a3 = z[zsw(jj + 1536)];
        a1 = c_mul_f(a1, w.p4[b][0]); a2 = c_mul_f(a2, w.p4[b][1]); a3 = c_mul_f(a3, w.p4[b][2]);
        const double2 s0 = c_add(a0, a2), s2 = c_add(a1, a3);
        z[zsw(jj + 1024)] = c_sub(s0, s2);
        if (jj + 512 <= max_lag) {
            const double2 s1 = c_sub(a0, a2), s3 = c_mul_mi(c_sub(a1, a3));
            z[zsw(jj + 1536)] = c_sub(s1, s3);
        }
    }
}

}  // namespace aegis
