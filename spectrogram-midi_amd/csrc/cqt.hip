// Constant-Q magnitudes as a block-sparse float32 GEMM on the MFMA units.
//
//   C[2k+c, t] = sum_j G[2k+c, j] * y[t*hop + j],   G = (re, im) rows of sqrt(N_k) * atom_k(-j)
//
// Rows are grouped in tiles of 16 (8 bins x re/im), bins ascending = supports descending; tile T only
// spans |j| < half[T].  One workgroup = 64 consecutive frames (4 column tiles of 16) x all row tiles.
// The signal is staged through LDS in passes of 256 samples per frame, double buffered ([2][64][256+2]
// floats: the +2 makes the 16 columns x 4 k-lanes of an MFMA B fragment hit 64 distinct banks); the next
// pass is fetched to registers while the current one feeds the MFMAs.  Every wave keeps
// accumulators for ALL row tiles (<= 11 tiles x 4 column tiles x 4 VGPRs) and takes every 4th k-step of
// a pass, so a B fragment is read from LDS once and reused for every active row tile from registers, the
// four waves are perfectly balanced however short the outer tiles are, and the split-K partial sums are
// reduced once at the end through LDS.  v_mfma_f32_16x16x4_f32: exact float32 products, float32 accumulate.
#include "cqt.h"

#include <algorithm>
#include <cmath>

namespace aegis {

using f32x4 = __attribute__((ext_vector_type(4))) float;

const char *build_cqt_bank(CqtBank &b, int sr, int n_bins, double fmin, int bins_per_octave, double filter_scale) {
    if (n_bins < 1 || n_bins > 8 * kCqtMaxTiles) return "cqt: n_bins must be 1..128";
    if (!(fmin > 0) || bins_per_octave < 1 || !(filter_scale > 0) || sr <= 0) return "cqt: bad parameters";
    b.n_bins = n_bins; b.sr = sr; b.fmin = fmin; b.bins_per_octave = bins_per_octave; b.filter_scale = filter_scale;
    b.n_tiles = (n_bins + 7) / 8;
    const double r = std::pow(2.0, 1.0 / bins_per_octave);
    const double alpha = (r * r - 1) / (r * r + 1);
    std::vector<double> freq(n_bins), ilen(n_bins);
    std::vector<int> lo(n_bins), len(n_bins);
    for (int k = 0; k < n_bins; ++k) {
        freq[k] = fmin * std::pow(2.0, (double)k / bins_per_octave);
        if (freq[k] >= sr / 2.0) return "cqt: a bin lies above Nyquist";
        ilen[k] = (filter_scale / alpha) * sr / freq[k];
        lo[k] = (int)std::floor(-ilen[k] / 2);                  // np.arange(-ilen // 2, ilen // 2)
        len[k] = (int)std::floor(ilen[k] / 2) - lo[k];
        if (len[k] < 1) return "cqt: empty atom";
    }
    int64_t total = 0;
    for (int T = 0; T < b.n_tiles; ++T) {
        const int k0 = 8 * T;
        const int reach = -lo[k0] + 1;                           // |j| <= -lo for the longest atom of the tile
        b.half[T] = (reach + kCqtChunk - 1) / kCqtChunk * kCqtChunk;
        b.offset[T] = total;
        total += (int64_t)2 * b.half[T] * 16;
    }
    if (b.half[0] > 64 * kCqtChunk) return "cqt: lowest bin needs more than 65536 taps";
    b.data.assign((size_t)total, 0.0f);
    for (int k = 0; k < n_bins; ++k) {
        const int T = k / 8, L = len[k];
        double wsum = 0;
        std::vector<double> w(L);
        for (int i = 0; i < L; ++i) { w[i] = 0.5 - 0.5 * std::cos(2 * M_PI * i / L); wsum += w[i]; }
        const double sc = std::sqrt(ilen[k]) / wsum;
        for (int i = 0; i < L; ++i) {
            const int m = lo[k] + i, j = -m;                     // coefficient of y[t*hop + j] is atom[m = -j]
            const double ang = 2 * M_PI * freq[k] * m / sr;
            const double re = sc * w[i] * std::cos(ang), im = sc * w[i] * std::sin(ang);
            const int kidx = j + b.half[T];
            const int step = kidx >> 2, q = kidx & 3;
            for (int c = 0; c < 2; ++c) {
                const int row = 2 * (k & 7) + c;
                b.data[(size_t)b.offset[T] + (size_t)step * 64 + q * 16 + row] = (float)(c ? im : re);
            }
        }
    }
    return "";
}

struct CqtMeta {
    int n_bins, n_tiles;
    int half[kCqtMaxTiles];
    int64_t offset[kCqtMaxTiles];
};

__device__ __forceinline__ int clip_of(const int64_t *__restrict__ off, int n, int64_t f) {
    int lo = 0, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= f) lo = mid; else hi = mid; }
    return lo;
}

constexpr int kPass = 256;              // samples of every frame staged per pass (one per thread and column)
constexpr int kBst = kPass + 2;         // padded row of the staging tile: B fragments hit 64 distinct banks
constexpr int kCqtRowTiles = 11;        // register accumulators are sized for 84 bins; more bins loop in groups
constexpr int kGroup = 4;               // k-steps whose A fragments are fetched ahead together

template <int NT>
__global__ __launch_bounds__(256) void cqt_kernel(CqtArgs a, CqtMeta m, const float *__restrict__ bank, int tile0) {
    extern __shared__ __align__(16) float sm[];
    float *Bst = sm;                                     // [2 buffers][64][kBst]
    __shared__ int64_t col_src[kCqtFrames];              // pcm offset of sample (t*hop) of the column's frame
    __shared__ int64_t col_lo[kCqtFrames];               // valid sample range of the clip, relative to t*hop
    __shared__ int64_t col_hi[kCqtFrames];
    __shared__ int64_t col_out[kCqtFrames];              // output offset of (bin 0, t); -1 for padding columns
    __shared__ int64_t col_F[kCqtFrames];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t f0 = (int64_t)blockIdx.x * kCqtFrames;
    if (tid < kCqtFrames) {
        const int64_t f = f0 + tid;
        if (f < a.n_frames) {
            const int c = clip_of(a.frame_off, a.n_clips, f);
            const int64_t t = f - a.frame_off[c], n = a.sample_off[c + 1] - a.sample_off[c];
            col_src[tid] = a.sample_off[c] + t * a.hop;
            col_lo[tid] = -t * a.hop;
            col_hi[tid] = n - t * a.hop;
            col_F[tid] = a.frame_off[c + 1] - a.frame_off[c];
            col_out[tid] = (int64_t)m.n_bins * a.frame_off[c] + t;
        } else {
            col_src[tid] = 0; col_lo[tid] = 0; col_hi[tid] = 0; col_F[tid] = 0; col_out[tid] = -1;
        }
    }
    f32x4 acc[NT][4];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[T][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // pass p covers j in [p*kPass, (p+1)*kPass) relative to the frame centre; thread `tid` owns sample p*kPass+tid
    // of every column (coalesced across the workgroup, per-column base and bounds are uniform)
    const int pmax = m.half[tile0] / kPass;
    float stage[kCqtFrames];
    auto fetch = [&](int p) {
        const int64_t j = (int64_t)p * kPass + tid;
#pragma unroll
        for (int col = 0; col < kCqtFrames; ++col)
            stage[col] = (j >= col_lo[col] && j < col_hi[col]) ? a.pcm[col_src[col] + j] : 0.0f;
    };
    auto commit = [&](int buf) {
        float *dst = Bst + buf * kCqtFrames * kBst + tid;
#pragma unroll
        for (int col = 0; col < kCqtFrames; ++col) dst[col * kBst] = stage[col];
    };
    fetch(-pmax);
    commit(0);
    __syncthreads();
    for (int p = -pmax; p < pmax; ++p) {
        const int buf = (p + pmax) & 1;
        if (p + 1 < pmax) fetch(p + 1);                  // global loads in flight under the MFMAs below
        const float *B = Bst + buf * kCqtFrames * kBst;
        // tiles whose support reaches this pass (supports descend with the tile index)
        const int reach_j = p >= 0 ? p * kPass : -(p + 1) * kPass;
        int na = 0;
#pragma unroll
        for (int T = 0; T < NT; ++T) na += (tile0 + T < m.n_tiles && m.half[tile0 + T] > reach_j) ? 1 : 0;
        // this wave's k-steps of the pass: s = w, w+4, ...; A fragments fetched one group of steps ahead
        constexpr int kSteps = kPass / 4 / 4;            // steps per wave and pass
        float an[NT][kGroup];
        auto load_a = [&](int g, float (&dst)[NT][kGroup]) {
#pragma unroll
            for (int T = 0; T < NT; ++T)
                if (T < na) {
                    const int64_t base = m.offset[tile0 + T] + ((int64_t)(p * kPass + m.half[tile0 + T]) / 4) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) dst[T][i] = bank[base + (int64_t)(w + 4 * (g * kGroup + i)) * 64];
                }
        };
        load_a(0, an);
        for (int g = 0; g < kSteps / kGroup; ++g) {
            float ac[NT][kGroup];
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int i = 0; i < kGroup; ++i) ac[T][i] = an[T][i];
            if (g + 1 < kSteps / kGroup) load_a(g + 1, an);
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int s = w + 4 * (g * kGroup + i);
                float b[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) b[ct] = B[(ct * 16 + (lane & 15)) * kBst + 4 * s + (lane >> 4)];
#pragma unroll
                for (int T = 0; T < NT; ++T)
                    if (T < na) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[T][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[T][i], b[ct], acc[T][ct], 0, 0, 0);
                    }
            }
        }
        if (p + 1 < pmax) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- split-K reduction over the 4 waves and magnitudes, one row tile at a time ----------------
    float *red = sm;                                     // [4 waves][4 ct][4 regs][64 lanes]
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (tile0 + T >= m.n_tiles) break;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w * 4 + ct) * 4 + r) * 64 + lane] = acc[T][ct][r];
        __syncthreads();
        for (int o = tid; o < 512; o += 256) {           // (ct, lane, pair)
            const int pr = o & 1, ln = (o >> 1) & 63, ct = o >> 7;
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                re += red[((ww * 4 + ct) * 4 + 2 * pr) * 64 + ln];
                im += red[((ww * 4 + ct) * 4 + 2 * pr + 1) * 64 + ln];
            }
            const int bin = 8 * (tile0 + T) + (ln >> 4) * 2 + pr;    // C/D layout: row = (lane>>4)*4 + reg
            const int col = ct * 16 + (ln & 15);
            if (bin < m.n_bins && col_out[col] >= 0)
                a.out[col_out[col] + (int64_t)bin * col_F[col]] = sqrtf(re * re + im * im);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Sliding-window variant (hop <= 680): the 48 frames of a workgroup are hop-shifted views of ONE
// stretch of signal, so LDS holds that stretch as a ring in sample space (32 768 floats + bank
// padding) and a pass of 256 taps needs only 256 NEW samples -- one per thread -- instead of
// re-staging 256 samples for every frame.  Workgroups are cut per clip so a tile never spans two
// signals.  B fragment (col, k) = ring[col*hop + pass*256 + k]; with hop = 512 the +2 floats per
// 512 keep the 16 columns x 4 k-lanes on 64 distinct banks.
// ------------------------------------------------------------------------------------------
constexpr int kRing = 32768;
constexpr int kSlideFrames = 48;
__device__ __forceinline__ int ring_idx(int u) {
    const int x = u & (kRing - 1);
    return x + 2 * (x >> 9);
}

template <int NT>
__global__ __launch_bounds__(256) void cqt_slide_kernel(CqtArgs a, CqtMeta m, const float *__restrict__ bank, int tile0,
                                                        const int64_t *__restrict__ tile_off) {
    extern __shared__ __align__(16) float ring[];        // kRing + 2*64 floats
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = clip_of(tile_off, a.n_clips, (int64_t)blockIdx.x);
    const int64_t t0 = ((int64_t)blockIdx.x - tile_off[c]) * kSlideFrames;
    const int64_t Fc = a.frame_off[c + 1] - a.frame_off[c];
    const int64_t n = a.sample_off[c + 1] - a.sample_off[c];
    const float *__restrict__ y = a.pcm + a.sample_off[c];
    const int hop = a.hop;
    const int half0 = m.half[tile0];
    const int64_t base = t0 * hop - half0;               // absolute sample of ring coordinate u = 0
    const int span = (kSlideFrames - 1) * hop + kPass;   // samples one pass touches

    f32x4 acc[NT][3];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[T][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    int colb[3];                                         // ring coordinate of (column, k-lane) at pass 0, step 0
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) colb[ct] = (ct * 16 + (lane & 15)) * hop + (lane >> 4);

    for (int u = tid; u < span; u += 256) {              // window of the first pass
        const int64_t sidx = base + u;
        ring[ring_idx(u)] = (sidx >= 0 && sidx < n) ? y[sidx] : 0.0f;
    }
    __syncthreads();
    const int npass = 2 * half0 / kPass;
    for (int q = 0; q < npass; ++q) {
        // the 256 samples that enter the window with the next pass (in flight under the MFMAs)
        const int un = span + q * kPass + tid;
        const int64_t sn = base + un;
        const float fresh = (q + 1 < npass && sn >= 0 && sn < n) ? y[sn] : 0.0f;
        // tiles whose support reaches this pass: j in [q*256 - half0, +256)
        const int j0 = q * kPass - half0;
        const int reach_j = j0 >= 0 ? j0 : -(j0 + kPass);
        int na = 0;
#pragma unroll
        for (int T = 0; T < NT; ++T) na += (tile0 + T < m.n_tiles && m.half[tile0 + T] > reach_j) ? 1 : 0;
        constexpr int kSteps = kPass / 4 / 4;
        float an[NT][kGroup];
        auto load_a = [&](int g, float (&dst)[NT][kGroup]) {
#pragma unroll
            for (int T = 0; T < NT; ++T)
                if (T < na) {
                    const int64_t b0 = m.offset[tile0 + T] + ((int64_t)(j0 + m.half[tile0 + T]) / 4) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) dst[T][i] = bank[b0 + (int64_t)(w + 4 * (g * kGroup + i)) * 64];
                }
        };
        load_a(0, an);
        for (int g = 0; g < kSteps / kGroup; ++g) {
            float ac[NT][kGroup];
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int i = 0; i < kGroup; ++i) ac[T][i] = an[T][i];
            if (g + 1 < kSteps / kGroup) load_a(g + 1, an);
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int s = w + 4 * (g * kGroup + i);
                float b[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    b[ct] = ring[ring_idx(colb[ct] + q * kPass + 4 * s)];
#pragma unroll
                for (int T = 0; T < NT; ++T)
                    if (T < na) {
#pragma unroll
                        for (int ct = 0; ct < 3; ++ct)
                            acc[T][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[T][i], b[ct], acc[T][ct], 0, 0, 0);
                    }
            }
        }
        ring[ring_idx(un)] = fresh;      // lands >= 8 192 slots away from anything the current pass reads
        __syncthreads();
    }

    // ---- split-K reduction over the 4 waves and magnitudes, one row tile at a time ----------------
    float *red = ring;                                   // [4 waves][3 ct][4 regs][64 lanes]
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (tile0 + T >= m.n_tiles) break;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w * 3 + ct) * 4 + r) * 64 + lane] = acc[T][ct][r];
        __syncthreads();
        for (int o = tid; o < 384; o += 256) {           // (ct, lane, pair)
            const int pr = o & 1, ln = (o >> 1) & 63, ct = o >> 7;
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                re += red[((ww * 3 + ct) * 4 + 2 * pr) * 64 + ln];
                im += red[((ww * 3 + ct) * 4 + 2 * pr + 1) * 64 + ln];
            }
            const int bin = 8 * (tile0 + T) + (ln >> 4) * 2 + pr;
            const int64_t t = t0 + ct * 16 + (ln & 15);
            if (bin < m.n_bins && t < Fc)
                a.out[(int64_t)m.n_bins * a.frame_off[c] + (int64_t)bin * Fc + t] = sqrtf(re * re + im * im);
        }
        __syncthreads();
    }
}

hipError_t cqt_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cqt_kernel<kCqtRowTiles>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kCqtFrames * kBst * sizeof(float)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(cqt_slide_kernel<kCqtRowTiles>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kRing + 128) * sizeof(float)));
}

void launch_cqt(const CqtArgs &a, const CqtBank &b, const int64_t *tile_off, int64_t n_slide_tiles, hipStream_t s) {
    if (a.n_frames == 0) return;
    CqtMeta m{};
    m.n_bins = b.n_bins; m.n_tiles = b.n_tiles;
    for (int T = 0; T < b.n_tiles; ++T) { m.half[T] = b.half[T]; m.offset[T] = b.offset[T]; }
    if (tile_off != nullptr && n_slide_tiles > 0 && (kSlideFrames - 1) * a.hop + 2 * kPass <= kRing - 8192) {
        for (int tile0 = 0; tile0 < b.n_tiles; tile0 += kCqtRowTiles)
            hipLaunchKernelGGL(cqt_slide_kernel<kCqtRowTiles>, dim3((unsigned)n_slide_tiles), dim3(256),
                               (kRing + 128) * sizeof(float), s, a, m, b.dev, tile0, tile_off);
        return;
    }
    const size_t lds = (size_t)2 * kCqtFrames * kBst * sizeof(float);
    const unsigned grid = (unsigned)((a.n_frames + kCqtFrames - 1) / kCqtFrames);
    for (int tile0 = 0; tile0 < b.n_tiles; tile0 += kCqtRowTiles)
        hipLaunchKernelGGL(cqt_kernel<kCqtRowTiles>, dim3(grid), dim3(256), lds, s, a, m, b.dev, tile0);
}

}  // namespace aegis
