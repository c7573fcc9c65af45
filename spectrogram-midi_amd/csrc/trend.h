// Launchers of the v2 trend-filter kernels (trend.hip).  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace aegis {

struct TrendArgs {
    const double *x;        // concatenated series
    const int64_t *off;     // [n_series+1]
    int n_series;
    int64_t total;
};

void trend_sma(const TrendArgs &a, int w, double *out, hipStream_t s);
void trend_ema(const TrendArgs &a, int span, double *out, hipStream_t s);
void trend_bollinger(const TrendArgs &a, int w, double k, double *ma, double *up, double *lo, hipStream_t s);
void trend_articulation(const TrendArgs &a, const double *up, const double *lo, int8_t *codes, hipStream_t s);
void trend_macd(const TrendArgs &a, int fast, int slow, int sig, double *m, double *sg, double *h, hipStream_t s);
void trend_semitones(const double *x, int64_t total, double *out, hipStream_t s);
void trend_slides(const double *macd, const double *hist, int64_t total, double thr, int8_t *codes, hipStream_t s);
void trend_rsi(const TrendArgs &a, int period, double *out, hipStream_t s);
void trend_rsi_averages(const TrendArgs &a, int period, double *avg_gain, double *avg_loss, hipStream_t s);
// density tracks of the ghost-note filter from the notes' [a, b) intervals, their Wilder averages (rsi_kernel<true>) and the
// averages at the notes' own positions; density / ag / al: scratch of `total` doubles, off: [n_series + 1] track offsets
void trend_ghost_rsi(const int64_t *ev_a, const int64_t *ev_b, const int32_t *ev_series, int64_t n_events, const int64_t *off,
                     int n_series, int64_t total, int period, double *density, double *ag, double *al, double *out_g, double *out_l,
                     hipStream_t s);
void trend_savgol(const TrendArgs &a, const double *coef_rev, int window, int symmetric, double *cx, int64_t *cpos,
                  int64_t *ccount, double *out, hipStream_t s);
void trend_kalman(const TrendArgs &a, double q, double r, double *out, hipStream_t s);
void trend_holt(const TrendArgs &a, double alpha, double beta, double *out, hipStream_t s);
void trend_band_confidence(const double *x, const double *upper, const double *lower, int64_t total, double *out, hipStream_t s);
void trend_consensus(const double *stacked, int k, int64_t len, double *med, double *conf, hipStream_t s);

}  // namespace aegis
