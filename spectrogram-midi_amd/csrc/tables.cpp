// Host table construction.  Each block cites the librosa 0.10 routine whose
// output it reproduces (the reference calls them through
// /root/reference/aegis_engine.py:25-26,63,67).
#include "tables.h"

#include <cfloat>
#include <cmath>

namespace aegis {

double np_pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

static std::vector<double> np_linspace(double start, double stop, int num) {
    std::vector<double> y(num);
    const double step = (stop - start) / (num - 1);
    for (int i = 0; i < num; ++i) y[i] = i * step + start;
    y[num - 1] = stop;
    return y;
}

// filters.py: Slaney mel scale
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa core/pitch.py::pyin: `p_init = np.zeros(2 * n_pitch_bins); p_init[n_pitch_bins:] = 1 / n_pitch_bins`, then
// sequence.viterbi takes log(p_init + tiny).  Mode 1 is the uniform start SURVEY.md P11 describes.
bool Tables::set_pyin_init(int mode) {
    if (mode != 0 && mode != 1) return false;
    pyin_init = mode;
    if (mode == 0) {
        log_pinit[0] = std::log(0.0 + DBL_MIN);
        log_pinit[1] = std::log(1.0 / n_bins + DBL_MIN);
    } else {
        log_pinit[0] = log_pinit[1] = std::log(1.0 / (2 * n_bins) + DBL_MIN);
    }
    return true;
}

std::string Tables::build(int sr_, int hop_, int n_fft_, int n_mels_, double fmin_, double fmax_) {
    sr = sr_; hop = hop_; n_fft = n_fft_; n_mels = n_mels_; fmin = fmin_; fmax = fmax_;
    if (n_fft != kFrameLength) return "only n_fft=2048 is built (the reference's value, aegis_engine.py:17)";
    if (sr <= 0 || hop <= 0 || n_mels <= 0 || n_mels > 128) return "bad sample_rate/hop_length/n_mels";
    if (!(fmin > 0) || !(fmax > fmin) || fmax > sr / 2.0) return "bad fmin/fmax";

    // ---- pYIN geometry: core/pitch.py::pyin -------------------------------------------------
    min_period = (int)std::floor(sr / fmax);
    max_period = std::min((int)std::ceil(sr / fmin), kFrameLength - kWinLength - 1);
    n_lags = max_period - min_period + 1;
    if (min_period < 1 || n_lags < 3) return "fmin/fmax leave fewer than 3 lags";
    const int bps = (int)std::ceil(1.0 / 0.1);
    n_bins = (int)std::floor(12 * bps * std::log2(fmax / fmin)) + 1;
    if (n_bins < 2 || 2 * n_bins > 1024) return "pitch grid needs 2..512 bins (one thread per HMM state)";
    const int max_semitones = (int)std::nearbyint(35.92 * 12 * hop / sr);  // Python round(): half-to-even
    width = max_semitones * bps + 1;
    half_width = width / 2;
    n_cls = width;
    if (width < 3 || n_bins <= 2 * half_width + 1) return "transition width does not fit the pitch grid";
    log_tiny = std::log(0.0 + DBL_MIN);
    set_pyin_init(0);

    // ---- Hann: scipy.signal.get_window('hann', n, fftbins=True) -------------------------------
    {
        std::vector<double> fac = np_linspace(-M_PI, M_PI, n_fft + 1);
        hann.resize(n_fft);
        for (int i = 0; i < n_fft; ++i) {
            double w = 0.0;
            w += 0.5 * std::cos(0 * fac[i]);
            w += 0.5 * std::cos(1 * fac[i]);
            hann[i] = w;
        }
    }

    // ---- mel filterbank: filters.py::mel(htk=False, norm='slaney', dtype=float32) -------------
    {
        const int n_bins_fft = 1 + n_fft / 2;
        const double d = 1.0 / sr, val = 1.0 / (n_fft * d);
        std::vector<double> fftfreqs(n_bins_fft);
        for (int k = 0; k < n_bins_fft; ++k) fftfreqs[k] = k * val;
        std::vector<double> mels = np_linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2);
        std::vector<double> mel_f(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(mels[i]);
        mel_dense.assign((size_t)n_mels * n_bins_fft, 0.0f);
        for (int i = 0; i < n_mels; ++i) {
            const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
            const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
            for (int k = 0; k < n_bins_fft; ++k) {
                const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
                const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
                const double w = std::fmax(0.0, std::fmin(lower, upper));
                float w32 = (float)w;
                w32 = (float)((double)w32 * enorm);
                mel_dense[(size_t)i * n_bins_fft + k] = w32;
            }
        }
        mel_start.assign(n_mels, 0); mel_len.assign(n_mels, 0); mel_off.assign(n_mels, 0);
        mel_w.clear();
        for (int i = 0; i < n_mels; ++i) {
            int lo = -1, hi = -1;
            for (int k = 0; k < n_bins_fft; ++k)
                if (mel_dense[(size_t)i * n_bins_fft + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
            mel_off[i] = (int32_t)mel_w.size();
            if (lo >= 0) {
                mel_start[i] = lo; mel_len[i] = hi - lo + 1;
                for (int k = lo; k <= hi; ++k) mel_w.push_back(mel_dense[(size_t)i * n_bins_fft + k]);
            }
        }
    }

    // chunks of <= 16 bins: the frame kernel gives every chunk to one thread (balanced: the widest band has ~85 bins)
    {
        mel_chunk_bin.clear(); mel_chunk_w.clear(); mel_band_chunk.assign(n_mels + 1, 0);
        for (int i = 0; i < n_mels; ++i) {
            mel_band_chunk[i] = (int32_t)mel_chunk_bin.size();
            for (int o = 0; o < mel_len[i]; o += 16) {
                mel_chunk_bin.push_back(mel_start[i] + o);
                for (int q = 0; q < 16; ++q) mel_chunk_w.push_back(o + q < mel_len[i] ? mel_w[mel_off[i] + o + q] : 0.0f);
            }
        }
        mel_band_chunk[n_mels] = (int32_t)mel_chunk_bin.size();
        if (mel_chunk_bin.size() > 256) return "mel filterbank needs more than 256 chunks of 16 bins";
    }

    // ---- pYIN priors: thresholds, Beta(2,18) mass per threshold, Boltzmann(2) pieces ----------
    {
        thresholds = np_linspace(0.0, 1.0, kNThresholds + 1);
        std::vector<double> cdf(kNThresholds + 1);
        for (int i = 0; i <= kNThresholds; ++i) {
            // I_x(2,18) = 1 - (1-x)^18 (1+18x), evaluated in extended precision
            const long double x = (long double)thresholds[i];
            cdf[i] = (double)(1.0L - powl(1.0L - x, 18.0L) * (1.0L + 18.0L * x));
        }
        cdf[0] = 0.0; cdf[kNThresholds] = 1.0;
        beta_probs.resize(kNThresholds);
        for (int i = 0; i < kNThresholds; ++i) beta_probs[i] = cdf[i + 1] - cdf[i];
        beta_cumsum.resize(kNThresholds + 1);
        for (int n = 0; n <= kNThresholds; ++n) beta_cumsum[n] = np_pairwise_sum(beta_probs.data(), n);
        beta_suffix.assign(kNThresholds + 1, 0.0);
        for (int n = kNThresholds - 1; n >= 0; --n) beta_suffix[n] = beta_suffix[n + 1] + beta_probs[n];
        // scipy.stats.boltzmann._pmf: fact = (1-exp(-l))/(1-exp(-l*N)); fact*exp(-l*k)
        const int nb = n_lags / 2 + 2;
        boltz_fact.assign(nb + 1, 0.0); boltz_exp.assign(nb + 1, 0.0);
        const double lam = 2.0;
        for (int N = 0; N <= nb; ++N) {
            boltz_fact[N] = N == 0 ? 0.0 : (1 - std::exp(-lam)) / (1 - std::exp(-lam * N));
            boltz_exp[N] = std::exp(-lam * N);
        }
    }

    // ---- HMM transitions: sequence.py::transition_local('triangle') x transition_loop(2,.99) --
    {
        const int B = n_bins, W = width, H = half_width;
        std::vector<double> tri(W);   // scipy.signal.windows.triang(W, sym=True), W odd
        for (int n = 1; n <= (W + 1) / 2; ++n) {
            const double w = 2.0 * n / (W + 1.0);
            tri[n - 1] = w; tri[W - n] = w;
        }
        const double p_stay = 1.0 - 0.01;
        const double sw[2][2] = {{p_stay, (1.0 - p_stay) / 1}, {(1.0 - p_stay) / 1, p_stay}};
        log_trans_band.assign((size_t)4 * n_cls * W, log_tiny);
        std::vector<double> row(B);
        for (int c = 0; c < n_cls; ++c) {
            // representative source row of the class
            const int b = c < H ? c : (c == H ? B / 2 : B - 1 - 2 * H + c);
            for (int j = 0; j < B; ++j) {
                const int dd = j - b + H;
                row[j] = (dd >= 0 && dd < W) ? tri[dd] : 0.0;
            }
            const double Z = 0.0 + np_pairwise_sum(row.data(), B);
            for (int dd = 0; dd < W; ++dd) {
                const int j = b + dd - H;
                if (j < 0 || j >= B) continue;
                const double loc = row[j] / Z;
                for (int v = 0; v < 2; ++v)
                    for (int v2 = 0; v2 < 2; ++v2)
                        log_trans_band[((size_t)(v * 2 + v2) * n_cls + c) * W + dd] =
                            std::log(sw[v][v2] * loc + DBL_MIN);
            }
        }
        // packed copy for the LDS of the band-specialised Viterbi: blocks (0,0) == (1,1) and (0,1) == (1,0)
        log_trans_pack.clear();
        if (n_cls == W) {
            bool sym = true;
            const size_t blk = (size_t)n_cls * W;
            for (size_t i = 0; i < blk && sym; ++i)
                sym = log_trans_band[i] == log_trans_band[3 * blk + i] && log_trans_band[blk + i] == log_trans_band[2 * blk + i];
            if (sym) {
                const int NP = 3 * H * H + 3 * H + 2;
                log_trans_pack.assign((size_t)2 * NP, log_tiny);
                for (int q = 0; q < 2; ++q) {
                    const double *src = log_trans_band.data() + (size_t)q * blk;     // block 0 = stay, block 1 = switch
                    double *dst = log_trans_pack.data() + (size_t)q * NP;
                    int o = 1;                                                       // slot 0: sentinel
                    for (int e = 0; e < H; ++e)
                        for (int dd = H - e; dd < W; ++dd) dst[o++] = src[(size_t)e * W + dd];
                    for (int dd = 0; dd < W; ++dd) dst[o++] = src[(size_t)H * W + dd];
                    for (int e = 0; e < H; ++e)
                        for (int dd = 0; dd <= 2 * H - 1 - e; ++dd) dst[o++] = src[(size_t)(H + 1 + e) * W + dd];
                    if (o != NP) { log_trans_pack.clear(); break; }
                }
            }
        }
        freqs.resize(B);
        for (int i = 0; i < B; ++i) freqs[i] = fmin * std::pow(2.0, (double)i / (12 * bps));
    }

    // ---- FFT twiddles exp(-2*pi*i*m/n), rounded from extended precision -----------------------
    twiddle.resize((size_t)2 * n_fft);
    for (int m = 0; m < n_fft; ++m) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / n_fft;
        twiddle[2 * m] = (double)cosl(ang);
        twiddle[2 * m + 1] = (double)sinl(ang);
    }
    return "";
}

}  // namespace aegis
