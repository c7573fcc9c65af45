// Constant-Q magnitudes as a block-sparse float32 GEMM on the MFMA units.
//
//   C[2k+c, t] = sum_j G[2k+c, j] * y[t*hop + j],   G = (re, im) rows of sqrt(N_k) * atom_k(-j)
//
// Rows are grouped in tiles of 16 (8 bins x re/im), bins ascending = supports descending; tile T only
// spans |j| < half[T].  Two kernels share one filter bank (layout: cqt_bank_index):
//   * cqt_slide_kernel (the one that runs for hop <= 512, hop % 16 == 0): 48 frames per workgroup as hop-shifted
//     views of ONE stretch of signal held as a ring in LDS; see the comment above it;
//   * cqt_kernel (any other hop): 64 frames (4 column tiles) per workgroup, the signal staged per frame through
//     LDS in passes of 256 samples, double buffered ([2][64][256+1] floats: the +1 makes the 16 columns x 4
//     k-lanes, 16 taps apart, of an MFMA B fragment hit 64 distinct banks); the next pass is fetched to
//     registers while the current one feeds the MFMAs.
// In both, every wave keeps accumulators for ALL row tiles of the launch (<= 11 tiles x 3 or 4 column tiles x
// 4 registers) and owns a quarter of the taps of every pass, so a B fragment is read from LDS once and reused
// for every active row tile from registers, the four waves are balanced however short the outer tiles are, and
// the split-K partial sums are reduced once at the end through LDS.
// v_mfma_f32_16x16x4_f32: exact float32 products, float32 accumulate.
#include "cqt.h"
#include <type_traits>

#include <algorithm>
#include <cmath>

namespace aegis {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kPassTaps = 256;          // taps per pass (see cqt_bank_index)
size_t cqt_bank_index(int half, int kidx, int row);

const char *build_cqt_bank(CqtBank &b, int sr, int n_bins, double fmin, int bins_per_octave, double filter_scale) {
    if (n_bins < 1 || n_bins > 8 * kCqtMaxTiles) return "cqt: n_bins must be 1..256";
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
    if (b.half[0] > 128 * kCqtChunk) return "cqt: lowest bin needs more than 131072 taps";
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
            for (int c = 0; c < 2; ++c) {
                const int row = 2 * (k & 7) + c;
                b.data[(size_t)b.offset[T] + cqt_bank_index(b.half[T], j + b.half[T], row)] = (float)(c ? im : re);
            }
        }
    }
    return "";
}

// Fragment order of one tile: [wave w][pass p][group g][lane = kl*16 + row][e].  Tap kidx = j + half lies in pass
// p = kidx / 256; inside the pass wave w owns taps [64w, 64w + 64); MFMA k-step i = 4g + e of that wave takes, on
// k-lane kl, tap 64w + 16kl + i.  So (a) one lane's four k-steps of a group are ONE 16-byte load, (b) a wave's
// fragments are one seamless 1-KiB-per-group stream across passes, and (c) the matching B operand -- 16
// consecutive samples per lane and pass -- is four ds_read_b128.  Any assignment of taps to (step, k-lane) is a
// valid GEMM as long as A and B agree; this one only fixes the float32 summation order.
size_t cqt_bank_index(int half, int kidx, int row) {
    const int p = kidx / kPassTaps, local = kidx % kPassTaps;
    const int w = local / 64, kl = (local % 64) / 16, i = local % 16;
    const size_t npass = (size_t)(2 * half / kPassTaps);
    return (((size_t)w * npass + p) * 4 + i / 4) * 256 + (size_t)(kl * 16 + row) * 4 + i % 4;
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

constexpr int kPass = kPassTaps;        // samples of every frame staged per pass (one per thread and column)
constexpr int kBst = kPass + 1;         // padded row of the staging tile: 16 columns x 4 k-lanes (16 apart) hit 64 distinct banks
constexpr int kCqtRowTiles = 11;        // register accumulators are sized for 84 bins; more bins loop in groups
#ifndef CQT_ABLATE
#define CQT_ABLATE 0
#endif
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
        // this wave's taps of the pass are [64w, 64w+64): k-step i on k-lane kl takes tap 64w + 16kl + i
        // (cqt_bank_index); the four A fragments of a group are one 16-byte load, fetched one group ahead
        constexpr int kSteps = kPass / 4 / 4;            // steps per wave and pass
        f32x4 an[NT];
        auto load_a = [&](int g, f32x4 (&dst)[NT]) {
#pragma unroll
            for (int T = 0; T < NT; ++T)
                if (T < na) {
                    const int hT = m.half[tile0 + T];
                    const int64_t npT = 2 * hT / kPass, pT = (p * kPass + hT) / kPass;
                    const int64_t base = m.offset[tile0 + T] + ((w * npT + pT) * 4 + g) * 256 + lane * 4;
                    dst[T] = *reinterpret_cast<const f32x4 *>(bank + base);
                }
        };
        load_a(0, an);
        for (int g = 0; g < kSteps / kGroup; ++g) {
            f32x4 ac[NT];
#pragma unroll
            for (int T = 0; T < NT; ++T) ac[T] = an[T];
            if (g + 1 < kSteps / kGroup) load_a(g + 1, an);
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                float b[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    b[ct] = B[(ct * 16 + (lane & 15)) * kBst + 64 * w + 16 * (lane >> 4) + g * kGroup + i];
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
// Sliding-window variant (hop <= 512, hop % 16 == 0): the 48 frames of a workgroup are hop-shifted
// views of ONE stretch of signal, so LDS holds that stretch as a ring in sample space (32 768 floats
// + bank padding) and a pass of 256 taps needs only 256 NEW samples -- one per thread -- instead of
// re-staging 256 samples for every frame.  Workgroups are cut per clip so a tile never spans two
// signals.  The B operand of lane (col, kl) in pass q is the 16 consecutive samples
// ring[col*hop + q*256 + 64w + 16kl + (0..15)]: four ds_read_b128; with hop = 512 the 4 floats of
// padding per 512 samples put the 16 columns on 16 distinct 16-byte bank groups.
//
// What bounds it (profiles/README.md): every workgroup streams the whole bank (4.4 MB for CQT-84) from
// L2 through ONE wave per SIMD.  With one dword per lane per k-step the vector-memory pipe (~1 load per
// 20-40 cycles per CU) and 250 cycles of address/branch work per k-step, not the MFMAs, set the pace
// (27 % of the f32 MFMA peak); hence 16-byte A loads, 16-byte B reads, per-tile register queues whose
// depth follows the time a k-step lasts, and one taken branch per group.
// ------------------------------------------------------------------------------------------
constexpr int kRing = 32768;
constexpr int kRingPad = 4;                                     // floats of padding per 512 samples
constexpr int kRingFloats = kRing + kRingPad * (kRing / 512);
constexpr int kSlideFrames = kCqtSlideFrames;
__device__ __forceinline__ int ring_idx(int u) {
    const int x = u & (kRing - 1);
    return x + kRingPad * (x >> 9);
}
// Groups (of 4 k-steps) that tile T's A fragments are fetched ahead of their MFMAs.  Tile T is only ever active
// together with tiles 0..T-1, so a group lasts >= (T+1) x 12 MFMAs = (T+1) x 384 cycles: the longest filters run
// alone in the outer passes and need the deepest queue.  Must divide the 4 groups of a pass.
// Round 4: queues of 8 / 8 / 4 groups (tile 0, tiles 1..3, the others) instead of 4 / 2 / 1, and refills that are
// UNCONDITIONAL.  The compiler places the waits for these loads; vector loads return in order, so the wait before a
// group's MFMAs is vmcnt(number of loads certainly issued since the fragment it needs).  While a refill sat behind
// `if (tile continues into the next pair)`, no refill counted as certain: the count fell by one per group and the last
// group of every pair waited with vmcnt(0) -- for the refill issued 400 cycles earlier, a full memory latency, once per
// pair of passes.  With unconditional refills (and tile 0 known active) every group waits vmcnt(depth - 1):
// 4.05 -> 3.5 ms for 64 x 30 s, and the depth no longer matters between 4 and 8.
#ifndef CQT_LA0
#define CQT_LA0 8
#endif
#ifndef CQT_LA1
#define CQT_LA1 8
#endif
#ifndef CQT_LA2
#define CQT_LA2 4
#endif
__host__ __device__ constexpr int slide_lookahead(int T) { return T == 0 ? CQT_LA0 : T <= 3 ? CQT_LA1 : CQT_LA2; }
constexpr int kSlotDepth = 8;           // groups of a pair of passes: the deepest queue (slot indices stay compile-time constants)
static_assert(kSlotDepth % CQT_LA0 == 0 && kSlotDepth % CQT_LA1 == 0 && kSlotDepth % CQT_LA2 == 0, "queue depths must divide 8");

#if CQT_ABLATE & 8
__device__ long long g_cqt_dbg[16];
#endif

template <int NT>
__global__ __launch_bounds__(256) void cqt_slide_kernel(CqtArgs a, CqtMeta m, const float *__restrict__ bank, int tile0,
                                                        const int64_t *__restrict__ tile_off) {
    extern __shared__ __align__(16) float ring[];        // kRingFloats
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the bank offsets in SGPRs
    const int c = clip_of(tile_off, a.n_clips, (int64_t)blockIdx.x);
    const int64_t t0 = ((int64_t)blockIdx.x - tile_off[c]) * kSlideFrames;
    const int64_t Fc = a.frame_off[c + 1] - a.frame_off[c];
    const int64_t n = a.sample_off[c + 1] - a.sample_off[c];
    const float *__restrict__ y = a.pcm + a.sample_off[c];
    const int hop = a.hop;
    const int half0 = m.half[tile0];
    const int64_t base = t0 * hop - half0;               // absolute sample of ring coordinate u = 0
    const int span = (kSlideFrames - 1) * hop + 2 * kPass;   // samples a pair of passes touches

#if CQT_ABLATE & 8
    long long tk0 = clock64(), tk_bar = 0; long long tk_na[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    f32x4 acc[NT][3];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[T][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    int colb[3];                                         // ring coordinate of (column, k-lane, wave) at pass 0, step 0
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) colb[ct] = (ct * 16 + (lane & 15)) * hop + 16 * (lane >> 4) + 64 * w;

    if (n > 0) {                                         // window of the first pair; clamped, branch-free loads batch
#pragma unroll 8
        for (int u = tid; u < span; u += 256) {
            const int64_t sidx = base + u;
            const int64_t cl = sidx < 0 ? 0 : sidx >= n ? n - 1 : sidx;
            const float v = y[cl];
            ring[ring_idx(u)] = cl == sidx ? v : 0.0f;
        }
    } else {
        for (int u = tid; u < span; u += 256) ring[ring_idx(u)] = 0.0f;
    }
    __syncthreads();
    const int npass = 2 * half0 / kPass;
    constexpr int kGroups = kPass / 4 / 16;              // groups of 4 k-steps (16 taps) per wave and pass (= 4)
    // Row tiles are ordered longest filter first, so the tiles whose support reaches pass q are a prefix
    // [0, na(q)): the per-tile work is a chain of nested ifs that is left at the first inactive tile (ONE
    // taken branch per group).  The A fragments are register queues refilled IN PLACE: the slot a group has
    // just consumed is loaded with the group slide_lookahead(T) further down the wave's stream, which
    // continues seamlessly into the next pass.
    // tile T is active in passes [first[T], npass - first[T]) (supports are centred); kept in SGPRs so that the
    // per-pass count is a dozen scalar compares instead of a chain of kernarg loads
    int first[NT];
    // byte offset of (tile T, this wave, group 0 of pass 0) in the bank; below the tile's data until it is active
    int tb[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const bool present = tile0 + T < m.n_tiles;
        const int hT = present ? m.half[tile0 + T] : half0;
        first[T] = present ? (half0 - hT) / kPass : 0x3fffffff;
        tb[T] = present ? (int)(m.offset[tile0 + T] * 4) + (w * (2 * hT / kPass) - (half0 - hT) / kPass) * (kGroups * 1024) : 0;
    }
    auto active_tiles = [&](int q) {
        const int d = q < npass - 1 - q ? q : npass - 1 - q;
        int na = 0;
#pragma unroll
        for (int T = 0; T < NT; ++T) na += first[T] <= d ? 1 : 0;
        return na;
    };
    const char *__restrict__ bank_b = reinterpret_cast<const char *>(bank);
    auto frag = [&](int T, int G) {                      // G = pass * 4 + group, counted from pass 0 of tile tile0
#if CQT_ABLATE & 1
        const float v = __int_as_float((lane + T + G) | 0x3f000000);
        return f32x4{v, v, v, v};
#else
        const uint32_t off = (uint32_t)(tb[T] + G * 1024) + (uint32_t)lane * 16u;
        return *reinterpret_cast<const f32x4 *>(bank_b + off);
#endif
    };
    f32x4 slot[NT][kSlotDepth];                          // tile T uses the first slide_lookahead(T) entries
    int na_next = active_tiles(0);
    {                                                    // queues of the tiles active in pass 0
        auto fill0 = [&](auto self, auto tc) -> void {
            constexpr int T = decltype(tc)::value;
            if constexpr (T < NT) {
                if (T < na_next) {
#pragma unroll
                    for (int gi = 0; gi < slide_lookahead(T); ++gi) slot[T][gi] = frag(T, gi);
                    self(self, std::integral_constant<int, T + 1>{});
                }
            }
        };
        fill0(fill0, std::integral_constant<int, 0>{});
    }
    // Passes run in PAIRS between barriers: supports are multiples of 512 taps, so both passes of a pair see the
    // same active tiles; one tile count, one barrier and one 512-sample refill of the ring per 512 taps.
    float fresh_next[2];                                 // the 512 samples that enter the window with the next pair
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t sn = base + span + k * 256 + tid;
        fresh_next[k] = (2 < npass && sn >= 0 && sn < n) ? y[sn] : 0.0f;
    }
#if CQT_ABLATE & 8
    long long tk1 = clock64();
#endif
    for (int q2 = 0; q2 < npass; q2 += 2) {
#if CQT_ABLATE & 8
        long long tp0 = clock64();
#endif
        const int un = span + q2 * kPass + tid;
        const float fresh[2] = {fresh_next[0], fresh_next[1]};      // fetched during the previous pair
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t sn = base + un + 2 * kPass + k * 256;
            fresh_next[k] = (q2 + 4 < npass && sn >= 0 && sn < n) ? y[sn] : 0.0f;
        }
        const int na = na_next;
        __builtin_assume(na >= 1);                       // tile 0 spans every pass: its block is not behind a branch
        na_next = q2 + 2 < npass ? active_tiles(q2 + 2) : 0;
#ifdef CQT_COND_REFILL
        const int ncont = na_next < na ? na_next : na;   // tiles whose queue keeps running into the next pair
#endif
        {                                                // tiles that join in the next pair: start their queues now
            auto fill = [&](auto self, auto tc) -> void {
                constexpr int T = decltype(tc)::value;
                if constexpr (T < NT) {
                    if (T < na_next) {
                        if (T >= na) {
#pragma unroll
                            for (int gi = 0; gi < slide_lookahead(T); ++gi) slot[T][gi] = frag(T, (q2 + 2) * kGroups + gi);
                        }
                        self(self, std::integral_constant<int, T + 1>{});
                    }
                }
            };
            if (na_next > na) fill(fill, std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int q = q2 + hp;
            int rb[3];                                   // opaque per pass: stops LICM from hoisting per-pass addresses
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                rb[ct] = ring_idx(colb[ct] + q * kPass) >> 2;    // in 16-byte units: 16 samples never straddle a pad
                asm volatile("" : "+v"(rb[ct]));
            }
            auto read_b = [&](int ct, int g) {
#if CQT_ABLATE & 2
                const float v = __int_as_float(((rb[ct] + g) & 0xffff) | 0x3f000000);
                return f32x4{v, v, v, v};
#else
                return reinterpret_cast<const f32x4 *>(ring)[rb[ct] + g];
#endif
            };
            f32x4 bn[3];                                 // B fragments are read one group ahead as well
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) bn[ct] = read_b(ct, 0);
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                f32x4 b[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) b[ct] = bn[ct];
                if (g + 1 < kGroups) {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) bn[ct] = read_b(ct, g + 1);
                }
                auto tiles = [&](auto self, auto tc) -> void {
                    constexpr int T = decltype(tc)::value;
                    if constexpr (T < NT) {
                        if (T < na) {
                            constexpr int LA = slide_lookahead(T);
                            const int si = (hp * kGroups + g) % LA;      // compile-time: hp and g are unrolled, q2 is even
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int ct = 0; ct < 3; ++ct)
                                    acc[T][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(slot[T][si][e], b[ct][e], acc[T][ct], 0, 0, 0);
                            // the slot just consumed takes the group LA further down the stream: inside this pair of passes, or in
                            // the next pair if the tile is still active there
#ifdef CQT_COND_REFILL
                            if (hp * kGroups + g + LA < 2 * kGroups || T < ncont) slot[T][si] = frag(T, q * kGroups + g + LA);
#else
                            // unconditional (a tile in its last pair requests up to LA KiB past its stream -- the next wave's or
                            // tile's fragments, or the padding behind the bank -- and never uses them): every group of an active
                            // tile then issues exactly one load, so the compiler's s_waitcnt vmcnt counts stay at the queue depth
                            // instead of falling to 0 by the last group of a pair
                            slot[T][si] = frag(T, q * kGroups + g + LA);
#endif
                            self(self, std::integral_constant<int, T + 1>{});
                        }
                    }
                };
                tiles(tiles, std::integral_constant<int, 0>{});
            }
        }
#if CQT_ABLATE & 8
        long long tp1 = clock64();
#endif
#if !(CQT_ABLATE & 4)
        // lands >= 7 680 slots away from anything the current pair reads (launch_cqt checks the window size)
        ring[ring_idx(un)] = fresh[0];
        ring[ring_idx(un + 256)] = fresh[1];
        __syncthreads();
#endif
#if CQT_ABLATE & 8
        { long long tp2 = clock64(); tk_bar += tp2 - tp1;
#pragma unroll
          for (int k = 1; k < 12; ++k) if (na == k) tk_na[k] += tp1 - tp0; }
#endif
    }

#if CQT_ABLATE & 8
    long long tk2 = clock64();
#endif
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
#if CQT_ABLATE & 8
    if (blockIdx.x == 1000 && tid == 0) {
        long long tk3 = clock64();
        g_cqt_dbg[0] = tk1 - tk0;
        for (int k = 1; k < 12; ++k) g_cqt_dbg[k] = tk_na[k];
        g_cqt_dbg[12] = tk_bar; g_cqt_dbg[13] = tk3 - tk2; g_cqt_dbg[14] = tk3 - tk0;
    }
#endif
}

#if CQT_ABLATE & 8
hipError_t cqt_debug_fetch(long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_cqt_dbg), sizeof(long long) * 16); }
#else
hipError_t cqt_debug_fetch(long long *dst) { for (int i = 0; i < 16; ++i) dst[i] = 0; return hipSuccess; }
#endif

// ------------------------------------------------------------------------------------------
// chroma_cqt's two steps behind the magnitudes (librosa.feature.chroma_cqt, auto_matcher.py:68-69): the 0/1 folding
// matrix filters.cq_to_chroma (every CQT bin belongs to exactly one chroma class: bin_class) and util.normalize(norm=inf)
// per frame (a frame whose maximum is below float tiny is left as it is).  One thread per frame: the magnitude rows are
// read with consecutive threads on consecutive frames, the n_chroma sums stay in registers (bins added in ascending
// order, float32 -- a BLAS product's order is unspecified and differs from it in the last bit at most).
// ------------------------------------------------------------------------------------------
constexpr int kMaxChroma = 24;
__global__ __launch_bounds__(256) void chroma_fold_kernel(const float *__restrict__ mag, const int64_t *__restrict__ frame_off, int n_clips,
                                                          int64_t n_frames, int n_bins, int n_chroma, const int32_t *__restrict__ bin_class,
                                                          float *__restrict__ out) {
    __shared__ int cls[256];
    for (int i = threadIdx.x; i < n_bins; i += blockDim.x) cls[i] = bin_class[i];
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const int c = clip_of(frame_off, n_clips, f);
    const int64_t F = frame_off[c + 1] - frame_off[c], t = f - frame_off[c];
    const float *__restrict__ m = mag + (int64_t)n_bins * frame_off[c] + t;
    float acc[kMaxChroma];
#pragma unroll
    for (int k = 0; k < kMaxChroma; ++k) acc[k] = 0.0f;
    for (int b = 0; b < n_bins; ++b) {
        const float v = m[(int64_t)b * F];
        const int k = cls[b];
#pragma unroll
        for (int q = 0; q < kMaxChroma; ++q) if (q == k) acc[q] = acc[q] + v;       // registers stay registers
    }
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxChroma; ++k) if (k < n_chroma) mx = fmaxf(mx, fabsf(acc[k]));
    const float div = mx < 1.17549435e-38f ? 1.0f : mx;
    float *__restrict__ o = out + (int64_t)n_chroma * frame_off[c] + t;
#pragma unroll
    for (int k = 0; k < kMaxChroma; ++k) if (k < n_chroma) o[(int64_t)k * F] = acc[k] / div;
}

void launch_chroma_fold(const float *mag, const int64_t *frame_off, int n_clips, int64_t n_frames, int n_bins, int n_chroma,
                        const int32_t *bin_class, float *out, hipStream_t s) {
    if (n_frames == 0) return;
    hipLaunchKernelGGL(chroma_fold_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, s, mag, frame_off, n_clips, n_frames,
                       n_bins, n_chroma, bin_class, out);
}

hipError_t cqt_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cqt_kernel<kCqtRowTiles>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kCqtFrames * kBst * sizeof(float)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(cqt_slide_kernel<kCqtRowTiles>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRingFloats * sizeof(float)));
}

void launch_cqt(const CqtArgs &a, const CqtBank &b, const int64_t *tile_off, int64_t n_slide_tiles, hipStream_t s) {
    if (a.n_frames == 0) return;
    CqtMeta m{};
    m.n_bins = b.n_bins; m.n_tiles = b.n_tiles;
    for (int T = 0; T < b.n_tiles; ++T) { m.half[T] = b.half[T]; m.offset[T] = b.offset[T]; }
    if (tile_off != nullptr && n_slide_tiles > 0 && a.hop % 16 == 0 && (kSlideFrames - 1) * a.hop + 4 * kPass <= kRing - 7168) {
        for (int tile0 = 0; tile0 < b.n_tiles; tile0 += kCqtRowTiles)
            hipLaunchKernelGGL(cqt_slide_kernel<kCqtRowTiles>, dim3((unsigned)n_slide_tiles), dim3(256),
                               kRingFloats * sizeof(float), s, a, m, b.dev, tile0, tile_off);
        return;
    }
    const size_t lds = (size_t)2 * kCqtFrames * kBst * sizeof(float);
    const unsigned grid = (unsigned)((a.n_frames + kCqtFrames - 1) / kCqtFrames);
    for (int tile0 = 0; tile0 < b.n_tiles; tile0 += kCqtRowTiles)
        hipLaunchKernelGGL(cqt_kernel<kCqtRowTiles>, dim3(grid), dim3(256), lds, s, a, m, b.dev, tile0);
}

}  // namespace aegis
