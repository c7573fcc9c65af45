// Constant-Q filter bank on the matrix cores (SURVEY.md 8a row a19 / BASELINE.json configs[2]).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace aegis {

constexpr int kCqtMaxTiles = 32;     // 16 filter rows (8 bins x re/im) per tile -> up to 256 bins (chroma_cqt: 252)
constexpr int kCqtChunk = 512;       // samples of every frame staged in LDS per pass
constexpr int kCqtFrames = 64;       // frames per workgroup (4 MFMA column tiles)

struct CqtBank {
    int n_bins = 0, n_tiles = 0;
    int sr = 0;
    double fmin = 0, filter_scale = 0;
    int bins_per_octave = 0;
    int half[kCqtMaxTiles] = {};          // half support of the tile, multiple of kCqtChunk, descending
    int64_t offset[kCqtMaxTiles] = {};    // float offset of the tile's fragments in `data`
    std::vector<float> data;              // [tile][wave][pass][group][64 lanes][4 k-steps], see cqt_bank_index()
    float *dev = nullptr;
};

// librosa 0.10 filters.wavelet(norm=1, window='hann') atoms, time-reversed and scaled by sqrt(N_k)
// (cqt(scale=True)), laid out as MFMA fragments.  Returns "" or an error message.
const char *build_cqt_bank(CqtBank &b, int sr, int n_bins, double fmin, int bins_per_octave, double filter_scale);

struct CqtArgs {
    const float *pcm;
    const int64_t *sample_off;   // [n_clips+1]
    const int64_t *frame_off;    // [n_clips+1]
    int n_clips;
    int64_t n_frames;
    int hop;
    float *out;                  // per clip [n_bins][F_clip], clip after clip
};

// tile_off: device [n_clips+1] prefix of ceil(F_clip / kCqtSlideFrames) (sliding-window kernel), or nullptr
constexpr int kCqtSlideFrames = 48;
void launch_cqt(const CqtArgs &a, const CqtBank &b, const int64_t *tile_off, int64_t n_slide_tiles, hipStream_t s);
// chroma folding + per-frame max normalisation of per-clip [n_bins, F] magnitudes -> per-clip [n_chroma, F]; n_bins <= 256, n_chroma <= 24
void launch_chroma_fold(const float *mag, const int64_t *frame_off, int n_clips, int64_t n_frames, int n_bins, int n_chroma,
                        const int32_t *bin_class, float *out, hipStream_t s);
hipError_t cqt_configure();
hipError_t cqt_debug_fetch(long long *dst);   // cycle counters of one workgroup (zeros unless built with CQT_ABLATE&8)

}  // namespace aegis
