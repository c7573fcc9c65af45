// Trend-filter kernels for the v2 "financial" pitch analysis:
// /root/reference/aegis_engine_core_v2/financial_analysis.py (SMA, EMA, Bollinger, MACD, RSI,
// articulation / slide state machines) and financial_filters.py (Savitzky-Golay on NaN-compacted
// samples, scalar Kalman, Holt, nan-median consensus), over ragged batches of float64 series.
//
// The reference's recurrences (EMA, Wilder RSI, Kalman, Holt, the state machines) are strictly
// sequential Python loops: each series is walked by ONE lane in the same order with the same
// float64 operations (-ffp-contract=off), so those outputs are bit-identical; the windowed
// operators (SMA, rolling std, Savitzky-Golay FIR, consensus) are one element per thread and
// reproduce NumPy's summation orders (pairwise sums, ndimage's symmetric correlate).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "trend.h"

namespace aegis {

__device__ __forceinline__ int series_of(const int64_t *__restrict__ off, int n, int64_t i) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// numpy pairwise_sum for n <= 128 (umath loops), reading a[i] through `get`
template <typename F>
__device__ __forceinline__ double np_sum_small(F get, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += get(i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = get(k);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += get(i + k);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += get(i);
    return res;
}

// One lane walks x[lo..hi) in order, `body(i, x[i])` per element, with the samples fetched 64 elements ahead:
// a lane that loads each sample right before using it pays a full memory round trip per element (0.2 us; a
// 3-minute pitch track took 1.6 ms per filter, the ghost-note filter's 77 k-element density track 17 ms).
template <typename Body>
__device__ __forceinline__ void walk_ahead(const double *__restrict__ x, int64_t lo, int64_t hi, Body body) {
    // K values per block (one 64-byte line), D blocks in flight.  A body costs ~30 cycles and a line that misses the caches
    // ~0.8 us, so one block of look-ahead left the lane waiting for memory seven eighths of the time (122 ns per element on a
    // single series); eight lines in flight cover it.  The loads are unconditional (index clamped to the last element): a
    // load under an in-range test ends at a control-flow join, where the compiler waits for EVERYTHING in flight.
    constexpr int K = 8, D = 8;
    if (hi <= lo) return;
    const int64_t last = hi - 1;
    double q[D][K];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < K; ++k) q[d][k] = x[min(lo + d * K + k, last)];
    for (int64_t i0 = lo; i0 < hi; i0 += D * K) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t b0 = i0 + d * K;
            double cur[K];
#pragma unroll
            for (int k = 0; k < K; ++k) cur[k] = q[d][k];
#pragma unroll
            for (int k = 0; k < K; ++k) q[d][k] = x[min(b0 + D * K + k, last)];   // in flight under D blocks of bodies
            if (b0 + K <= hi) {               // whole block: no test per element
#pragma unroll
                for (int k = 0; k < K; ++k) body(b0 + k, cur[k]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (b0 + k < hi) body(b0 + k, cur[k]);
            }
        }
    }
}
template <typename Body>
__device__ __forceinline__ void walk_ahead3(const double *__restrict__ x, const double *__restrict__ y,
                                            const double *__restrict__ z, int64_t lo, int64_t hi, Body body) {
    constexpr int K = 4, D = 8;
    if (hi <= lo) return;
    const int64_t last = hi - 1;
    double qx[D][K], qy[D][K], qz[D][K];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < K; ++k) { const int64_t i = min(lo + d * K + k, last); qx[d][k] = x[i]; qy[d][k] = y[i]; qz[d][k] = z[i]; }
    for (int64_t i0 = lo; i0 < hi; i0 += D * K) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t b0 = i0 + d * K;
            double cx[K], cy[K], cz[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { cx[k] = qx[d][k]; cy[k] = qy[d][k]; cz[k] = qz[d][k]; }
#pragma unroll
            for (int k = 0; k < K; ++k) { const int64_t i = min(b0 + D * K + k, last); qx[d][k] = x[i]; qy[d][k] = y[i]; qz[d][k] = z[i]; }
            if (b0 + K <= hi) {               // whole block: no test per element
#pragma unroll
                for (int k = 0; k < K; ++k) body(b0 + k, cx[k], cy[k], cz[k]);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (b0 + k < hi) body(b0 + k, cx[k], cy[k], cz[k]);
            }
        }
    }
}



// ---- a13: simple_moving_average (financial_analysis.py:45-69) -------------------------------
// np.convolve(nan->0, ones(w)/w, 'same') then NaN restored.
__global__ void sma_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series,
                           int64_t total, int w, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = series_of(off, n_series, i);
    const int64_t lo = off[s], hi = off[s + 1];
    const double xi = x[i];
    if (xi != xi) { out[i] = NAN; return; }
    const double kv = 1.0 / (double)w;
    const int64_t m1 = i + (w - 1) / 2, m0 = m1 - (w - 1);
    double acc = 0.0;
    for (int64_t m = m0 > lo ? m0 : lo; m <= m1 && m < hi; ++m) {
        const double v = x[m];
        acc += (v != v ? 0.0 : v) * kv;
    }
    out[i] = acc;
}

// ---- a14: exponential_moving_average (financial_analysis.py:71-107), one lane per series ----
__device__ void ema_series(const double *__restrict__ x, int64_t n, int span, double *__restrict__ out) {
    const double alpha = 2.0 / (double)(span + 1);
    double prev = NAN;
    bool started = false;
    const double beta = 1 - alpha;
    walk_ahead(x, 0, n, [&](int64_t i, double v) {
        // branch-free (selects): a data-dependent branch per element costs more than the recurrence itself on one lane
        const bool ok = v == v;
        const double rec = alpha * v + beta * prev;
        double e = (started && prev == prev) ? rec : v;      // the first valid sample, or the first after a NaN gap, restarts
        e = ok ? e : NAN;
        started = started || ok;
        out[i] = e;
        prev = e;
    });
}

__global__ __launch_bounds__(64) void ema_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series, int span,
                           double *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    ema_series(x + off[s], off[s + 1] - off[s], span, out + off[s]);
}

// ---- a15: bollinger_bands (financial_analysis.py:113-146): needs ma (SMA) computed before ----
__global__ void bollinger_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series,
                                 int64_t total, int w, double num_std, const double *__restrict__ ma,
                                 double *__restrict__ upper, double *__restrict__ lower) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = series_of(off, n_series, i);
    const int64_t lo = off[s];
    int64_t a = i - w + 1;
    if (a < lo) a = lo;
    // np.std of the non-NaN values of x[a..i] (population), only when there are at least two
    double buf[128];
    int cnt = 0;
    for (int64_t m = a; m <= i; ++m) { const double v = x[m]; if (v == v && cnt < 128) buf[cnt++] = v; }
    double sd = NAN;
    if (cnt > 1) {
        const double mean = np_sum_small([&](int k) { return buf[k]; }, cnt) / (double)cnt;
        const double var = np_sum_small([&](int k) { const double d = buf[k] - mean; return d * d; }, cnt) / (double)cnt;
        sd = sqrt(var);
    }
    upper[i] = ma[i] + (num_std * sd);
    lower[i] = ma[i] - (num_std * sd);
}

// ---- detect_articulation_bollinger state machine (financial_analysis.py:148-197) -------------
// codes: 0 None, 1 normal, 2 bend, 3 vibrato, 4 noise
__global__ __launch_bounds__(64) void articulation_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series,
                                    const double *__restrict__ upper, const double *__restrict__ lower,
                                    int8_t *__restrict__ codes) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    int prev = 0, vib = 0;   // state: 0 normal, 1 above, 2 below
    walk_ahead3(x, upper, lower, off[s], off[s + 1], [&](int64_t i, double v, double up, double lw) {
        const bool ok = v == v;                  // a NaN sample writes 0 and leaves the state alone (selects, no branch)
        const int st = v > up ? 1 : (v < lw ? 2 : 0);
        const int vib_new = (prev != st && prev != 0) ? vib + 1 : 0;
        const int code = vib_new >= 2 ? 3 : (st == 1 ? 2 : (st == 2 ? 4 : 1));
        codes[i] = (int8_t)(ok ? code : 0);
        vib = ok ? vib_new : vib;
        prev = ok ? st : prev;
    });
}

// ---- a16: macd (financial_analysis.py:203-226) ------------------------------------------------
// One walk per series: the fast and the slow EMA advance together (same input, same NaN / restart state), their difference
// feeds the signal EMA of the same step.  Every value is produced by the operations, in the order, of three separate
// exponential_moving_average calls (the reference's form), so the outputs are bit-identical to them; the three separate
// walks plus two element-wise passes of the first version cost 11 ms per call, one lane waiting on global memory.
__global__ __launch_bounds__(64) void macd_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series, int fast,
                            int slow, int sig, double *__restrict__ macd, double *__restrict__ signal,
                            double *__restrict__ hist) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    const int64_t o = off[s], n = off[s + 1] - o;
    const double af = 2.0 / (double)(fast + 1), as = 2.0 / (double)(slow + 1), ag = 2.0 / (double)(sig + 1);
    double pf = NAN, ps = NAN, pg = NAN;
    bool started = false, gstarted = false;
    const double bf = 1 - af, bs = 1 - as, bg = 1 - ag;
    walk_ahead(x + o, 0, n, [&](int64_t i, double v) {
        // the three EMAs of ema_series, branch-free
        const bool ok = v == v;
        const double rf = af * v + bf * pf, rs = as * v + bs * ps;
        double ef = (started && pf == pf) ? rf : v;
        double es = (started && ps == ps) ? rs : v;
        ef = ok ? ef : NAN;
        es = ok ? es : NAN;
        started = started || ok;
        pf = ef; ps = es;
        const double m = ef - es;
        const bool mok = m == m;
        const double rg = ag * m + bg * pg;
        double g = (gstarted && pg == pg) ? rg : m;
        g = mok ? g : NAN;
        gstarted = gstarted || mok;
        pg = g;
        macd[o + i] = m;
        signal[o + i] = g;
        hist[o + i] = m - g;
    });
}

// Hz -> MIDI semitones for detect_slides_macd (financial_analysis.py:242-248; librosa.hz_to_midi)
__global__ void semitone_kernel(const double *__restrict__ x, int64_t total, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double v = x[i];
    out[i] = v != v ? NAN : 12 * (log2(v) - log2(440.0)) + 69;
}

// codes: 0 None, 1 normal, 2 slide_up, 3 slide_down (financial_analysis.py:254-268)
__global__ void slides_kernel(const double *__restrict__ macd, const double *__restrict__ hist, int64_t total,
                              double thr, int8_t *__restrict__ codes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double m = macd[i], h = hist[i];
    int8_t c;
    if (m != m) c = 0;
    else if (m > thr && h > 0) c = 2;
    else if (m < -thr && h < 0) c = 3;
    else c = 1;
    codes[i] = c;
}

// ---- rsi (financial_analysis.py:274-320), Wilder smoothing, one lane per series ---------------
// AVERAGES: write the two Wilder averages (avg_gain to `out`, avg_loss to `out2`; NaN where the RSI is the constant 50)
// instead of the RSI: the ghost-note filter reads the RSI at a hundred positions of a 77 k-element track, and the two
// divisions that turn the averages into an RSI value are then done for those positions only (by the caller, the same
// IEEE operations) instead of for every element on the one lane that walks the series.
template <bool AVERAGES>
__global__ __launch_bounds__(64) void rsi_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series, int period,
                           double *__restrict__ out, double *__restrict__ out2) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    const double *d = x + off[s];
    double *o = out + off[s];
    double *o2 = AVERAGES ? out2 + off[s] : nullptr;
    const int64_t n = off[s + 1] - off[s];
    const double flat = AVERAGES ? (double)NAN : 50.0;
    if (n - 1 < period || period < 1) {
        for (int64_t i = 0; i < n; ++i) { o[i] = flat; if (AVERAGES) o2[i] = flat; }
        return;
    }
    for (int64_t i = 0; i < period; ++i) { o[i] = flat; if (AVERAGES) o2[i] = flat; }       // the rest is written by the recurrence below
    auto gain = [&](int64_t i) { const double dl = d[i + 1] - d[i]; return dl > 0 ? dl : 0.0; };
    auto loss = [&](int64_t i) { const double dl = d[i + 1] - d[i]; return dl < 0 ? -dl : 0.0; };
    double ag, al;
    if (period <= 128) {
        ag = np_sum_small([&](int k) { return gain(k); }, period) / (double)period;
        al = np_sum_small([&](int k) { return loss(k); }, period) / (double)period;
    } else {
        ag = al = NAN;   // host rejects period > 128
    }
    // One lane walks the series.  Per element the loop-carried work is one division deep (the two Wilder averages);
    // the RSI value itself (two more dependent divisions) hangs off it.  Blocks of eight elements without a branch
    // inside let the scheduler run the output chains of earlier elements under the recurrence of later ones, and the
    // samples of the next block are fetched while this one is computed (the ghost-note filter calls this on a
    // 10-per-frame density track: 77 k elements for 3 minutes).  Same operations per element as the plain loop.
    constexpr int K = 8;
    const double pm1 = (double)(period - 1), pd = (double)period;
    auto step = [&](double dl, int64_t i) {
        ag = (ag * pm1 + (dl > 0 ? dl : 0.0)) / pd;
        al = (al * pm1 + (dl < 0 ? -dl : 0.0)) / pd;
        if (AVERAGES) { o[i] = ag; o2[i] = al; return; }
        const double rs = ag / al;
        const double val = 100 - (100 / (1 + rs));
        o[i] = al == 0 ? 100.0 : val;
    };
    {   // i = period: the seed averages themselves
        if (AVERAGES) { o[period] = ag; o2[period] = al; }
        else if (al == 0) o[period] = 100;
        else { const double rs = ag / al; o[period] = 100 - (100 / (1 + rs)); }
    }
    int64_t i = (int64_t)period + 1;                // element i uses d[i] - d[i-1]
    double nxt[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) nxt[k] = d[min(i - 1 + k, n - 1)];      // (clamped, not tested: see walk_ahead)
    for (; i + K <= n; i += K) {
        double cur[K + 1];
#pragma unroll
        for (int k = 0; k <= K; ++k) cur[k] = nxt[k];
#pragma unroll
        for (int k = 0; k <= K; ++k) nxt[k] = d[min(i + K - 1 + k, n - 1)];
#pragma unroll
        for (int k = 0; k < K; ++k) step(cur[k + 1] - cur[k], i + k);
    }
    for (; i < n; ++i) step(d[i] - d[i - 1], i);
}

// ---- a17: Savitzky-Golay on NaN-compacted samples (financial_filters.py:25-59) ----------------
// pass 1 (one lane per series): compact valid samples, remember their positions
__global__ __launch_bounds__(64) void compact_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series,
                               double *__restrict__ cx, int64_t *__restrict__ cpos, int64_t *__restrict__ ccount) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    int64_t k = off[s];
    // every sample is written at the next free slot and the slot is kept only if the sample is valid (no branch; the slot
    // after the last valid sample may end up holding an invalid one: slots >= count are never read)
    walk_ahead(x, off[s], off[s + 1], [&](int64_t i, double v) {
        cx[k] = v;
        cpos[k] = i;
        k += (v == v) ? 1 : 0;
    });
    ccount[s] = k - off[s];
}

// pass 2: scipy.signal.savgol_filter(valid, window, polyorder, mode='nearest') =
// ndimage.correlate1d with the reversed coefficients; `coef` arrives already reversed, centre at
// coef[half].  Sum order follows NI_Correlate1D: symmetric kernels fold the two sides.  Two launches:
// initialise the output (NaN, or the input when a series has no valid sample), then filter + scatter.
__global__ void savgol_init_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series,
                                   int64_t total, const int64_t *__restrict__ ccount, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = series_of(off, n_series, i);
    out[i] = ccount[s] == 0 ? x[i] : NAN;
}

__global__ void savgol_apply_kernel(const int64_t *__restrict__ off, int n_series, int64_t total,
                                    const double *__restrict__ cx, const int64_t *__restrict__ cpos,
                                    const int64_t *__restrict__ ccount, const double *__restrict__ coef, int window,
                                    int symmetric, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int s = series_of(off, n_series, i);
    const int64_t o = off[s], nv = ccount[s];
    const int64_t k = i - o;
    if (nv <= window || k >= nv) return;
    const int half = window / 2;
    auto at = [&](int64_t q) { q = q < 0 ? 0 : (q >= nv ? nv - 1 : q); return cx[o + q]; };
    const double *fw = coef + half;
    double tmp;
    if (symmetric) {
        tmp = at(k) * fw[0];
        for (int jj = -half; jj < 0; ++jj) tmp += (at(k + jj) + at(k - jj)) * fw[jj];
    } else {
        tmp = at(k + half) * fw[half];
        for (int jj = -half; jj < half; ++jj) tmp += at(k + jj) * fw[jj];
    }
    out[cpos[o + k]] = tmp;
}

// ---- Kalman (financial_filters.py:62-99) and Holt (:102-141), one lane per series -------------
__global__ __launch_bounds__(64) void kalman_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series, double q,
                              double r, double *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    double x_est = NAN, p_est = 1.0;
    bool started = false;
    walk_ahead(x, off[s], off[s + 1], [&](int64_t i, double v) {
        const bool ok = v == v;                       // a NaN sample leaves the state alone (selects, no branch)
        const double x_pred = started ? x_est : v, p_pred = p_est + q;
        const double k = p_pred / (p_pred + r);
        const double x_new = x_pred + k * (v - x_pred);
        const double p_new = (1 - k) * p_pred;
        out[i] = ok ? x_new : NAN;
        x_est = ok ? x_new : x_est;
        p_est = ok ? p_new : p_est;
        started = started || ok;
    });
}

__global__ __launch_bounds__(64) void holt_kernel(const double *__restrict__ x, const int64_t *__restrict__ off, int n_series, double alpha,
                            double beta, double *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    const int64_t lo = off[s], hi = off[s + 1];
    double first = NAN, second = NAN;
    int found = 0;
    for (int64_t i = lo; i < hi && found < 2; ++i) {
        const double v = x[i];
        if (v == v) { if (found == 0) first = v; else second = v; ++found; }
    }
    if (found < 2) {                 // fewer than two valid samples: the input comes back unchanged
        for (int64_t i = lo; i < hi; ++i) out[i] = x[i];
        return;
    }
    double level = first, trend = second - first;
    const double ca = 1 - alpha, cb = 1 - beta;
    walk_ahead(x, lo, hi, [&](int64_t i, double v) {
        const bool ok = v == v;                       // a NaN sample leaves the state alone (selects, no branch)
        const double forecast = level + trend;
        const double level_new = alpha * v + ca * forecast;
        const double trend_new = beta * (level_new - level) + cb * trend;
        out[i] = ok ? level_new : NAN;
        level = ok ? level_new : level;
        trend = ok ? trend_new : trend;
    });
}

// ---- multi_filter_consensus (financial_filters.py:256-298): nanmedian / 1/(1+nanstd) ----------
__global__ void consensus_kernel(const double *__restrict__ stacked, int k, int64_t len, double *__restrict__ med,
                                 double *__restrict__ conf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double v[8];
    int cnt = 0;
    double sum = 0.0;                       // np.sum over axis 0 of the NaN->0 copy: ((0 + r0) + r1) + ...
    for (int r = 0; r < k; ++r) {
        const double a = stacked[(int64_t)r * len + i];
        const bool ok = a == a;
        sum += ok ? a : 0.0;
        if (ok && cnt < 8) v[cnt++] = a;
    }
    if (cnt == 0) { med[i] = NAN; conf[i] = NAN; return; }
    // nanstd
    const double avg = sum / (double)cnt;
    double sq = 0.0;
    for (int r = 0; r < k; ++r) {
        const double a = stacked[(int64_t)r * len + i];
        const double d = (a == a) ? a - avg : 0.0;
        sq += d * d;
    }
    const double sd = sqrt(sq / (double)cnt);
    conf[i] = 1.0 / (1.0 + sd);
    // nanmedian: insertion sort of <= 8 values
    for (int a = 1; a < cnt; ++a) {
        const double key = v[a];
        int b = a - 1;
        while (b >= 0 && v[b] > key) { v[b + 1] = v[b]; --b; }
        v[b + 1] = key;
    }
    med[i] = (cnt & 1) ? v[cnt / 2] : (v[cnt / 2 - 1] == v[cnt / 2] ? v[cnt / 2] : (v[cnt / 2 - 1] + v[cnt / 2]) / 2.0);
}

// analyze_pitch_financial's confidence (financial_analysis.py:409-417): 1 / (1 + band width) where the sample and the
// width are numbers (1 when the width is not positive), 0 elsewhere
__global__ void band_confidence_kernel(const double *__restrict__ x, const double *__restrict__ upper, const double *__restrict__ lower,
                                       int64_t total, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double v = x[i], w = upper[i] - lower[i];
    double c = 0.0;
    if (v == v && w == w) c = w > 0 ? 1.0 / (1.0 + w) : 1.0;
    out[i] = c;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline unsigned blocks(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

void trend_sma(const TrendArgs &a, int w, double *out, hipStream_t s) {
    if (a.total) hipLaunchKernelGGL(sma_kernel, dim3(blocks(a.total, 256)), dim3(256), 0, s, a.x, a.off, a.n_series, a.total, w, out);
}
void trend_ema(const TrendArgs &a, int span, double *out, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(ema_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, span, out);
}
void trend_bollinger(const TrendArgs &a, int w, double k, double *ma, double *up, double *lo, hipStream_t s) {
    if (!a.total) return;
    trend_sma(a, w, ma, s);
    hipLaunchKernelGGL(bollinger_kernel, dim3(blocks(a.total, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, a.total, w, k, ma, up, lo);
}
void trend_articulation(const TrendArgs &a, const double *up, const double *lo, int8_t *codes, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(articulation_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, up, lo, codes);
}
void trend_macd(const TrendArgs &a, int fast, int slow, int sig, double *m, double *sg, double *h, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(macd_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, fast, slow, sig, m, sg, h);
}
void trend_semitones(const double *x, int64_t total, double *out, hipStream_t s) {
    if (total) hipLaunchKernelGGL(semitone_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, x, total, out);
}
void trend_slides(const double *macd, const double *hist, int64_t total, double thr, int8_t *codes, hipStream_t s) {
    if (total) hipLaunchKernelGGL(slides_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, macd, hist, total, thr, codes);
}
void trend_rsi(const TrendArgs &a, int period, double *out, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(rsi_kernel<false>, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, period, out, (double *)nullptr);
}
void trend_rsi_averages(const TrendArgs &a, int period, double *avg_gain, double *avg_loss, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(rsi_kernel<true>, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, period, avg_gain, avg_loss);
}
// ---- ghost-note density tracks (financial_analysis.py:333-347) built on the device -----------------------------------------
// The filter adds 1 over [start*10, end*10) per note to a track of int(max_end * 10) elements (77 k for a three-minute
// clip) and reads the RSI of that track at the notes' start positions.  The counts are small integers, exact in float64
// whatever the order of the additions, so the track is a difference array (two atomic adds per note) and a parallel
// prefix sum; only the Wilder averages at the notes' own positions go back to the host.
__global__ __launch_bounds__(256) void ghost_diff_kernel(const int64_t *__restrict__ ev_a, const int64_t *__restrict__ ev_b,
                                                         const int32_t *__restrict__ ev_series, int64_t n_events,
                                                         const int64_t *__restrict__ off, double *__restrict__ d) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    const int s = ev_series[e];
    const int64_t base = off[s], n = off[s + 1] - base;
    const int64_t a = ev_a[e], b = min(ev_b[e], n);
    if (a < 0 || a >= n || b <= a) return;
    atomicAdd(d + base + a, 1.0);
    if (b < n) atomicAdd(d + base + b, -1.0);
}
// in-place inclusive prefix sum of every series, one workgroup per series (exact: integer values)
__global__ __launch_bounds__(256) void ghost_scan_kernel(const int64_t *__restrict__ off, double *__restrict__ d) {
    constexpr int PER = 8, TILE = 256 * PER;
    __shared__ double wsum[4];
    __shared__ double carry_s;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double *x = d + off[s];
    const int64_t n = off[s + 1] - off[s];
    if (tid == 0) carry_s = 0.0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < n; t0 += TILE) {
        double v[PER];
        double run = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = t0 + (int64_t)tid * PER + k;
            run += i < n ? x[i] : 0.0;
            v[k] = run;
        }
        double incl = run;                     // inclusive scan of the threads' sums over the wave, then over the 4 waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double nb = __shfl_up(incl, o); if (lane >= o) incl += nb; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        double before = carry_s + (incl - run);
        for (int w = 0; w < wid; ++w) before += wsum[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = t0 + (int64_t)tid * PER + k;
            if (i < n) x[i] = before + v[k];
        }
        __syncthreads();
        if (tid == 255) carry_s = before + run;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void ghost_gather_kernel(const int64_t *__restrict__ ev_a, const int32_t *__restrict__ ev_series,
                                                           int64_t n_events, const int64_t *__restrict__ off,
                                                           const double *__restrict__ ag, const double *__restrict__ al,
                                                           double *__restrict__ out_g, double *__restrict__ out_l) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    const int s = ev_series[e];
    const int64_t base = off[s], n = off[s + 1] - base, a = ev_a[e];
    const bool in = a >= 0 && a < n;
    out_g[e] = in ? ag[base + a] : (double)NAN;
    out_l[e] = in ? al[base + a] : (double)NAN;
}
void trend_ghost_rsi(const int64_t *ev_a, const int64_t *ev_b, const int32_t *ev_series, int64_t n_events, const int64_t *off,
                     int n_series, int64_t total, int period, double *density, double *ag, double *al, double *out_g, double *out_l,
                     hipStream_t s) {
    if (!n_events || !n_series) return;
    (void)hipMemsetAsync(density, 0, (size_t)total * 8, s);
    hipLaunchKernelGGL(ghost_diff_kernel, dim3(blocks(n_events, 256)), dim3(256), 0, s, ev_a, ev_b, ev_series, n_events, off, density);
    hipLaunchKernelGGL(ghost_scan_kernel, dim3(n_series), dim3(256), 0, s, off, density);
    hipLaunchKernelGGL(rsi_kernel<true>, dim3(blocks(n_series, 64)), dim3(64), 0, s, density, off, n_series, period, ag, al);
    hipLaunchKernelGGL(ghost_gather_kernel, dim3(blocks(n_events, 256)), dim3(256), 0, s, ev_a, ev_series, n_events, off, ag, al, out_g, out_l);
}
void trend_savgol(const TrendArgs &a, const double *coef_rev, int window, int symmetric, double *cx, int64_t *cpos,
                  int64_t *ccount, double *out, hipStream_t s) {
    if (!a.total) return;
    hipLaunchKernelGGL(compact_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, cx, cpos, ccount);
    hipLaunchKernelGGL(savgol_init_kernel, dim3(blocks(a.total, 256)), dim3(256), 0, s, a.x, a.off, a.n_series, a.total, ccount, out);
    hipLaunchKernelGGL(savgol_apply_kernel, dim3(blocks(a.total, 256)), dim3(256), 0, s, a.off, a.n_series, a.total, cx, cpos, ccount,
                       coef_rev, window, symmetric, out);
}
void trend_kalman(const TrendArgs &a, double q, double r, double *out, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(kalman_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, q, r, out);
}
void trend_holt(const TrendArgs &a, double alpha, double beta, double *out, hipStream_t s) {
    if (a.n_series) hipLaunchKernelGGL(holt_kernel, dim3(blocks(a.n_series, 64)), dim3(64), 0, s, a.x, a.off, a.n_series, alpha, beta, out);
}
void trend_band_confidence(const double *x, const double *upper, const double *lower, int64_t total, double *out, hipStream_t s) {
    if (total) hipLaunchKernelGGL(band_confidence_kernel, dim3(blocks(total, 256)), dim3(256), 0, s, x, upper, lower, total, out);
}
void trend_consensus(const double *stacked, int k, int64_t len, double *med, double *conf, hipStream_t s) {
    if (len) hipLaunchKernelGGL(consensus_kernel, dim3(blocks(len, 256)), dim3(256), 0, s, stacked, k, len, med, conf);
}

}  // namespace aegis
