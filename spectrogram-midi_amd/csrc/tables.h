// Host-side constant tables for the analyze path (window, mel filterbank, pYIN
// priors, HMM log-transitions, FFT twiddles).  librosa builds these in Python
// (filters.py::mel/get_window, core/pitch.py::pyin, sequence.py::transition_*)
// every time the reference calls it (/root/reference/aegis_engine.py:25,63);
// here they are built once per handle, in float64 on the host, and uploaded.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace aegis {

constexpr int kFrameLength = 2048;   // librosa.pyin / feature.rms default frame_length
constexpr int kWinLength = 1024;     // pyin win_length = frame_length // 2
constexpr int kNThresholds = 100;    // pyin n_thresholds

struct Tables {
    // configuration
    int sr = 44100, hop = 512, n_fft = 2048, n_mels = 128;
    double fmin = 0, fmax = 0;
    // derived pYIN geometry
    int min_period = 0, max_period = 0, n_lags = 0;
    int n_bins = 0;          // n_pitch_bins
    int half_width = 0;      // transition half width in bins
    int width = 0;           // 2*half_width+1
    int n_cls = 0;           // source-row classes of the banded transition table (= width)
    double log_tiny = 0;     // log(0 + tiny)
    double log_pinit[2] = {0, 0};   // log(p_init + tiny) of a voiced / an unvoiced state (set_pyin_init)
    int pyin_init = 0;       // AEGIS_PYIN_INIT_*
    // tables
    std::vector<double> hann;          // [n_fft]
    std::vector<float> mel_dense;      // [n_mels][1+n_fft/2]
    std::vector<int32_t> mel_start;    // [n_mels] first non-zero bin
    std::vector<int32_t> mel_len;      // [n_mels] run length of non-zero bins
    std::vector<int32_t> mel_off;      // [n_mels] offset into mel_w
    std::vector<float> mel_w;          // packed non-zero weights
    // the same triangles cut into chunks of <= 16 consecutive bins (one thread of the frame kernel each, <= 256 chunks):
    std::vector<int32_t> mel_chunk_bin;    // [n_chunks] first FFT bin
    std::vector<float> mel_chunk_w;        // [n_chunks][16] weights, zero padded
    std::vector<int32_t> mel_band_chunk;   // [n_mels + 1] first chunk of each band (prefix)
    std::vector<double> thresholds;    // [101]
    std::vector<double> beta_probs;    // [100]
    std::vector<double> beta_cumsum;   // [101]  np.sum(beta_probs[:n])
    std::vector<double> beta_suffix;   // [101]  sum(beta_probs[n:]) accumulated from the small end
    std::vector<double> boltz_fact;    // [n]    (1-e^-2)/(1-e^-2N)
    std::vector<double> boltz_exp;     // [n]    e^-2k
    std::vector<double> log_trans_band; // [4][n_cls][width]  (v*2+v') major
    std::vector<double> log_trans_pack; // [2 (stay, switch)][3H^2+3H+2]: the band table without duplicate blocks and
                                        // without the unreachable parts of the edge rows (viterbi.hip pk_*); empty if
                                        // the (v,v') blocks are not pairwise identical
    std::vector<double> freqs;         // [n_bins]
    std::vector<double> twiddle;       // [n_fft][2]  cos, sin of -2*pi*m/n_fft

    std::string build(int sr, int hop, int n_fft, int n_mels, double fmin, double fmax);
    // core/pitch.py::pyin's p_init: 0 = zeros(2B) with p_init[B:] = 1/B (librosa's code), 1 = uniform 1/(2B)
    bool set_pyin_init(int mode);
};

// numpy's pairwise summation (umath loops: pairwise_sum_DOUBLE), used where the
// reference's tables are defined through np.sum.
double np_pairwise_sum(const double *a, int64_t n);

}  // namespace aegis
