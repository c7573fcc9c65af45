// Host emulation of csrc/fft8.h: the 256 threads of the workgroup run in a loop, phase by phase (the loops' ends are the
// barriers).  Checks the 2048-point transform against a direct long-double DFT and the inverse-by-conjugation use.
//   hipcc -x hip --cuda-host-only -O2 tools/fft8_host_check.cpp -o tools/_build/fft8_host_check && tools/_build/fft8_host_check
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../spectrogram-midi_amd/csrc/fft8.h"
using namespace aegis;

static void fft2048_emulated(std::vector<double2> &x, const std::vector<double2> &tw) {
    std::vector<double2> z(2048);
    std::vector<Fft8Tw> w(256);
    for (int j = 0; j < 256; ++j) fft8_load_twiddles(w[j], tw.data(), j);
    static double2 v[256][8];
    for (int j = 0; j < 256; ++j) { for (int q = 0; q < 8; ++q) v[j][q] = x[j + 256 * q]; fft8_pass1_write(z.data(), j, v[j]); }
    for (int j = 0; j < 256; ++j) fft8_read8(z.data(), j, v[j]);
    for (int j = 0; j < 256; ++j) fft8_pass_write<8>(z.data(), j, v[j], w[j].p2);
    for (int j = 0; j < 256; ++j) fft8_read8(z.data(), j, v[j]);
    for (int j = 0; j < 256; ++j) fft8_pass_write<64>(z.data(), j, v[j], w[j].p3);
    for (int j = 0; j < 256; ++j) fft8_pass4(z.data(), j, w[j]);
    for (int i = 0; i < 2048; ++i) x[i] = z[zsw(i)];
}

int main() {
    std::vector<double2> tw(2048);
    for (int m = 0; m < 2048; ++m) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * m / 2048;
        tw[m] = make_double2((double)cosl(a), (double)sinl(a));
    }
    srand(7);
    std::vector<double2> x(2048), y;
    for (auto &c : x) c = make_double2(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    y = x;
    fft2048_emulated(y, tw);
    long double worst = 0, scale = 0;
    for (int k = 0; k < 2048; k += 1) {
        long double re = 0, im = 0;
        for (int n = 0; n < 2048; ++n) {
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * ((long long)k * n % 2048) / 2048;
            re += x[n].x * cosl(a) - x[n].y * sinl(a);
            im += x[n].x * sinl(a) + x[n].y * cosl(a);
        }
        worst = fmaxl(worst, fmaxl(fabsl(re - y[k].x), fabsl(im - y[k].y)));
        scale = fmaxl(scale, fmaxl(fabsl(re), fabsl(im)));
    }
    printf("forward: max abs error %.3Le, max |X| %.3Le\n", worst, scale);
    // inverse by conjugation: FFT(conj(FFT(x))) = N conj(x)
    std::vector<double2> c = y;
    for (auto &e : c) e.y = -e.y;
    fft2048_emulated(c, tw);
    long double w2 = 0;
    for (int i = 0; i < 2048; ++i) w2 = fmaxl(w2, fmaxl(fabsl(c[i].x / 2048 - x[i].x), fabsl(-c[i].y / 2048 - x[i].y)));
    printf("round trip: max abs error %.3Le\n", w2);
    // swizzle is a permutation of 0..2047
    std::vector<int> seen(2048, 0);
    for (int i = 0; i < 2048; ++i) seen[zsw(i)]++;
    for (int i = 0; i < 2048; ++i) if (seen[i] != 1) { printf("zsw is not a permutation\n"); return 1; }
    const bool ok = worst < 2e-12L * scale && w2 < 1e-14L;
    printf(ok ? "OK\n" : "FAIL\n");
    return ok ? 0 : 1;
}
