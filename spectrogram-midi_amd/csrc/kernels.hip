// HIP kernels for the Aegis analyze path on gfx950 (MI355X / CDNA4, wave64).
//
// Reference behaviour being reproduced (the arithmetic lives in librosa, which
// /root/reference/aegis_engine.py:25-26,63,67,70 calls):
//   frame_yin_kernel   melspectrogram power + feature.rms + pyin's difference function
//                      (FFT autocorrelation + running energy)  
//   pyin_obs_kernel    CMND, troughs, threshold prior, pitch-bin observation (P4-P10)
//   (viterbi.hip)      882-state log-Viterbi with chunked back-tracking (P11, P12)
//   finalize kernels   power_to_db(ref=max), rake mask (vision.py:3-38), f0 decode
//
// Built with -ffp-contract=off: every multiply/add below rounds exactly where
// NumPy rounds; fused operations are written as fma() where they are wanted.
#include "kernels.h"
#include "fft8.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace aegis {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_clip(const int64_t *__restrict__ frame_off, int n_clips, int64_t f) {
    int lo = 0, hi = n_clips;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// selection / step range of this launch: by-value fields, or the device control block of a graph replay
__device__ __forceinline__ int64_t geo_n_sel(const PassParams &p) { return p.ctl ? p.ctl->n_sel : p.n_sel; }
__device__ __forceinline__ int64_t geo_t_begin(const PassParams &p) { return p.ctl ? p.ctl->t_begin : p.t_begin; }
__device__ __forceinline__ int64_t geo_vt_begin(const PassParams &p) { return p.ctl ? p.ctl->vt_begin : p.vt_begin; }
__device__ __forceinline__ int64_t geo_vt_end(const PassParams &p) { return p.ctl ? p.ctl->vt_end : p.vt_end; }

// selected-frame index of this launch -> (clip, frame within clip, frame within pass = workspace row)
__device__ __forceinline__ void map_frame(const PassParams &p, int64_t fs, int &c, int64_t &t, int64_t &f) {
    c = find_clip(p.sel_off, p.n_clips, fs);
    t = (p.clip_t0 ? p.clip_t0[c] : geo_t_begin(p)) + (fs - p.sel_off[c]);
    f = p.frame_off[c] + t;
}
// frame t of clip c in the output arrays
__device__ __forceinline__ int64_t out_index(const PassParams &p, int c, int64_t t) { return p.out_off[c] + t; }

// wave maximum of unsigned values, one v_max_u32_dpp per level (0, the identity, stands in for lanes without a source)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
    return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {   // uniform result
    v = dpp_umax<0x111, 0xf>(v);   // row_shr:1
    v = dpp_umax<0x112, 0xf>(v);   // row_shr:2
    v = dpp_umax<0x114, 0xf>(v);   // row_shr:4
    v = dpp_umax<0x118, 0xf>(v);   // row_shr:8
    v = dpp_umax<0x142, 0xa>(v);   // row_bcast:15
    v = dpp_umax<0x143, 0xc>(v);   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}


// ------------------------------------------------------------------------------------------
// Kernel 1: frame stage.  A 256-thread workgroup owns `frames_per_wg` consecutive selected frames (16 in batch
// launches, 2 for streaming pushes) and takes them two at a time; 81 KB of LDS, so two workgroups share a CU.
//
//   prologue  float32 running energy of pyin's difference function for ALL the workgroup's frames at once, one frame
//             per lane of wave 0: e = np.cumsum(frame**2) is strictly sequential, so the lanes walk their own frame
//             (samples straight from global memory, 16-byte loads, next block in flight) and leave
//             en[tau] = e[1024 + tau] - e[tau] in an LDS row per frame.  Same operations in the same order as NumPy.
//   per frame centred frame -> LDS (zero padded at the clip edges; the next frame's samples are already in flight),
//             feature.rms in NumPy's float32 pairwise order (bit-exact), ONE packed forward FFT
//             Z = FFT(x + i*b), b = reversed first half: A = FFT(x) and B = FFT(b) separate by symmetry, P = A*B is the
//             spectrum of pyin's autocorrelation, and the Hann-windowed spectrum librosa.stft needs is
//             XW[k] = A[k]/2 - (A[k-1] + A[k+1])/4 (periodic Hann in the frequency domain): the mel path costs no FFT.
//             |XW|^2 rounded through complex64 like librosa.stft -> sparse Slaney mel -> clip maximum.
//   per pair  ONE inverse FFT for both frames (Q = P0 + i*P1 by Hermitian extension) -> acf0 + i*acf1, then
//             d[tau] = (en[0] + en[tau]) - 2 acf[tau] with librosa's |.| < 1e-6 clamps, written to HBM (the only
//             pYIN intermediate this kernel materialises; the CMND cumsum and quotient open pyin_obs_kernel).
// FFT: csrc/fft8.h (radix 8 x 8 x 8 x 4 in registers, one in-place LDS buffer).  1.5 FFTs per frame.
// ------------------------------------------------------------------------------------------
constexpr int kFramesPerWg = 16;

// 8 consecutive samples x[idx..idx+7] with zero fill outside [0, n)
__device__ __forceinline__ void load8(const float *__restrict__ x, int64_t n, int64_t idx, bool vec_ok, float (&v)[8]) {
    if (vec_ok && idx >= 0 && idx + 7 < n) {
        const float4 a = *reinterpret_cast<const float4 *>(x + idx);
        const float4 b = *reinterpret_cast<const float4 *>(x + idx + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int64_t q = idx + i; v[i] = (q >= 0 && q < n) ? x[q] : 0.0f; }
    }
}

// floats per running-energy row: lags 0..max_period rounded up to whole groups of 32 (the walk stores whole groups),
// then to a multiple of 4 whose quarter is odd, so that the 16-byte accesses of 16 lanes (one row each) fall on 16
// distinct bank groups
__host__ __device__ inline int frame_en_stride(int max_period) {
    int s = ((max_period + 32) >> 5) << 5;          // whole groups of 32 lags (energy_walk)
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}

// One lane's walk of np.cumsum(frame**2) (float32, strictly sequential) in straight-line groups of 32 samples:
//   A  j <  32 nG         e[j] stored                      (nG groups cover lags 0..max_period)
//   B  32 nG <= j < 1024  chain only
//   C  j = 1024 + tau     row[tau] = e[1024 + tau] - e[tau]
// `fetch(j0, a, b)` returns samples j0..j0+7 of the lane's frame.  A group's eight fetches are issued while the group
// before it is chained, into the other of two register sets (no copies), and little but the 32 dependent adds and the
// group's own row traffic sits between two groups (tools/ubench_walk.hip times the variants on an idle CU).  The row is over-written in whole groups (entries past
// max_period are never read; frame_en_stride leaves room for them).
__host__ __device__ inline int energy_groups(int max_period) { return (max_period + 32) >> 5; }
// The same walk as one plain loop, one sample at a time (rare workgroups: frames of two clips, or a hop the staging
// area cannot hold): small code, no concern for speed.  `sample(j)` returns sample j of the lane's frame.
template <typename Sample>
__device__ __forceinline__ void energy_walk_plain(Sample sample, float *__restrict__ row, int mp) {
    const int n = 32 * energy_groups(mp);
    float e = 0.0f;
    for (int j = 0; j < 1024 + n; ++j) {
        const float x = sample(j);
        e = e + x * x;
        if (j < n) row[j] = e;
        else if (j >= 1024) row[j - 1024] = e - row[j - 1024];
    }
}
template <bool SQUARED, typename Fetch>
__device__ __forceinline__ void energy_walk(Fetch fetch, float *__restrict__ row, int mp) {
    const int nG = energy_groups(mp);
    float4 ra[8], rb[8];            // two register sets: one group is chained while the next one's samples arrive
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(8 * u, ra[2 * u], ra[2 * u + 1]);
    float e = 0.0f;                 // 0 + x*x == x*x exactly: the first add reproduces np.cumsum's first element
#define AEGIS_SQ(x) (SQUARED ? (x) : (x) * (x))
#define AEGIS_CHAIN4(q, o)                                                                       \
    { float sq;                                                                                  \
      sq = AEGIS_SQ(q.x); e = e + sq; o.x = e;  sq = AEGIS_SQ(q.y); e = e + sq; o.y = e;         \
      sq = AEGIS_SQ(q.z); e = e + sq; o.z = e;  sq = AEGIS_SQ(q.w); e = e + sq; o.w = e; }
    // group g from `cur`, group g + 1 requested into `nxt` first (past the last group: inside the staging area or
    // bounds-checked, never used).  Each phase is its own straight-line loop, so the wait before a chain counts exactly
    // the eight requests behind the samples it needs.
    auto group = [&](auto phase, float4 (&cur)[8], float4 (&nxt)[8], int g) {
        constexpr int PH = decltype(phase)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(32 * (g + 1) + 8 * u, nxt[2 * u], nxt[2 * u + 1]);
        float4 o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) AEGIS_CHAIN4(cur[u], o[u])
        if (PH == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<float4 *>(row + 32 * g + 4 * u) = o[u];
        }
        if (PH == 2) {              // the row reads wait here, 17 times per walk; ahead of the chain they would put
            float *r = row + 32 * (g - 32);      // sixteen requests in flight, more than a wait can count
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 lo = *reinterpret_cast<const float4 *>(r + 4 * u);
                *reinterpret_cast<float4 *>(r + 4 * u) = make_float4(o[u].x - lo.x, o[u].y - lo.y, o[u].z - lo.z, o[u].w - lo.w);
            }
        }
    };
    // groups [g0, g1) of one phase; returns with the current samples in `a` again (an odd count is finished through
    // the mirrored instance)
    auto run = [&](auto phase, float4 (&a)[8], float4 (&b2)[8], int g0, int g1) {
        int g = g0;
        for (; g + 1 < g1; g += 2) { group(phase, a, b2, g); group(phase, b2, a, g + 1); }
        return g;
    };
    using PA = std::integral_constant<int, 0>;
    using PB = std::integral_constant<int, 1>;
    using PC = std::integral_constant<int, 2>;
    // three phases, register roles alternating with every group: a phase of odd length hands over in the other set
    int g = run(PA{}, ra, rb, 0, nG);
    bool in_a = true;
    if (g < nG) { group(PA{}, ra, rb, g); ++g; in_a = false; }
    if (in_a) { g = run(PB{}, ra, rb, g, 32); if (g < 32) { group(PB{}, ra, rb, g); ++g; in_a = false; } }
    else      { g = run(PB{}, rb, ra, g, 32); if (g < 32) { group(PB{}, rb, ra, g); ++g; in_a = true; } }
    if (in_a) { g = run(PC{}, ra, rb, g, 32 + nG); if (g < 32 + nG) group(PC{}, ra, rb, g); }
    else      { g = run(PC{}, rb, ra, g, 32 + nG); if (g < 32 + nG) group(PC{}, rb, ra, g); }
#undef AEGIS_CHAIN4
#undef AEGIS_SQ
}
// doubles per CMND row of the frame kernel's epilogue: lags 0..max_period, even, half of it odd, so that the 16-byte
// accesses of 16 lanes (one row each) fall on 16 distinct bank groups
__host__ __device__ inline int frame_cmnd_stride(int max_period) {
    int s = (max_period + 2) & ~1;
    if (((s >> 1) & 1) == 0) s += 2;
    return s;
}
// One lane's walk of np.cumsum(d[1:]) (float64, strictly sequential) over its LDS row, IN PLACE: r[tau] becomes
// cs[tau] = d[1] + ... + d[tau].  Straight-line groups of 8 lags over two register sets, a group's 16-byte reads requested
// while the group before it is chained; the look-ahead may read up to 64 bytes past the row (the next row, or the slack
// behind the last one), never writes there.
__device__ __forceinline__ void cmnd_walk(double *__restrict__ r, int mp) {
    constexpr int GL = 8;
    double cs = r[1];
    double2 ra[GL / 2], rb[GL / 2];
    auto request = [&](double2 (&q)[GL / 2], int t0) {
#pragma unroll
        for (int i = 0; i < GL / 2; ++i) q[i] = *reinterpret_cast<const double2 *>(r + t0 + 2 * i);
    };
    auto group = [&](double2 (&cur)[GL / 2], double2 (&nxt)[GL / 2], int t0) {
        request(nxt, t0 + GL);
#pragma unroll
        for (int i = 0; i < GL / 2; ++i) {
            double2 o;
            cs = cs + cur[i].x; o.x = cs;
            cs = cs + cur[i].y; o.y = cs;
            *reinterpret_cast<double2 *>(r + t0 + 2 * i) = o;
        }
    };
    int tau = 2;
    request(ra, tau);
    for (; tau + 2 * GL - 1 <= mp; tau += 2 * GL) { group(ra, rb, tau); group(rb, ra, tau + GL); }
    if (tau + GL - 1 <= mp) { group(ra, rb, tau); tau += GL; }
    for (; tau <= mp; ++tau) { cs = cs + r[tau]; r[tau] = cs; }
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_up(v, o);
        if (lane >= o) v += nb;
    }
    return v;
}

// Layout of a frame's trough list in its dfn row (PassParams::troughs): [0] the count K (as an integer bit pattern), then
// from double 8 on th[KM] (CMND value of each trough, ascending lag), tsh[KM] (its parabolic shift) and ti[KM] (int16 lag
// index); KM = n_lags / 2 + 2 bounds the number of local minima.
__host__ __device__ inline int trough_km(int n_lags) { return n_lags / 2 + 2; }
__host__ __device__ inline int trough_row_doubles(int n_lags) { const int km = trough_km(n_lags); return 8 + 2 * km + (km + 3) / 4; }

constexpr size_t kFrameLdsFixed = (size_t)2048 * 16 + 2048 * 4 + 1040 * 4 + 128 * 4 + 16 * 4 + 256 * 4;

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 128)
__device__ long long g_frm_dbg[24];
#define FRM_TICK(k) { __builtin_amdgcn_s_waitcnt(0); const long long now__ = clock64(); facc[k] += now__ - flast; flast = now__; }
#else
#define FRM_TICK(k)
#endif
__global__ __launch_bounds__(256, 2) void frame_yin_kernel(PassParams p, DevTables tb, int frames_per_wg, int en_stride) {
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 128)
    long long facc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, flast = clock64();
#endif
    extern __shared__ __align__(16) unsigned char fsm[];
    double2 *z = reinterpret_cast<double2 *>(fsm);               // [2048] FFT buffer (swizzled index, fft8.h)
    float *xs = reinterpret_cast<float *>(z + 2048);             // [2048] the frame
    float *pw = xs + 2048;                                       // [1040] windowed power spectrum (1025 bins, zero tail)
    float *red = pw + 1040;                                      // [128]  rms partial sums
    float *blk = red + 128;                                      // [16]
    float *part = blk + 16;                                      // [256]  mel partial sums, one per 16-bin chunk of a triangle
    float *en = part + 256;                                      // [frames_per_wg][en_stride] running energies
    int64_t *frow = reinterpret_cast<int64_t *>(en + (size_t)frames_per_wg * en_stride);   // [frames_per_wg] workspace row of each frame, -1: not live (epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t n_sel = geo_n_sel(p);
    const int64_t fs0 = (int64_t)blockIdx.x * frames_per_wg;
    if (fs0 >= n_sel) return;
    const int nfr = (int)min((int64_t)frames_per_wg, n_sel - fs0);
    const bool want_pyin = (p.stages & 0x4u) != 0, want_mel = (p.stages & 0x3u) != 0;
    const bool want_rms = (p.stages & 0x8u) && p.out_rms != nullptr;
    const bool want_fft = (p.stages & 0x7u) != 0;
    const int mp = p.max_period;

    struct Geo { bool live; int c; int64_t base, n, start, f, o; };
    auto locate = [&](int i) {
        Geo g{false, 0, 0, 0, 0, 0, 0};
        g.live = i < nfr;
        if (g.live) {
            int64_t t;
            map_frame(p, fs0 + i, g.c, t, g.f);
            g.o = out_index(p, g.c, t);
            g.base = p.sample_off[g.c];
            g.n = p.sample_len[g.c];
            g.start = t * p.hop - 1024;
        }
        return g;
    };
    auto fetch = [&](const Geo &g, float (&v)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t idx = g.start + tid + r * 256;
            v[r] = (g.live && idx >= 0 && idx < g.n) ? p.pcm[g.base + idx] : 0.0f;
        }
    };
    // Frame i + 1 of the workgroup is normally the next frame of frame i's clip: one comparison against the end of the
    // clip's selection then replaces locate()'s binary search, a chain of dependent loads each frame had to wait for.
    auto locate_next = [&](const Geo &g, int i, int64_t sel_end) {
        if (i < nfr && g.live && fs0 + i < sel_end) {
            Geo n = g;
            n.start += p.hop; n.f += 1; n.o += 1;
            return n;
        }
        return locate(i);
    };
    Geo geo = locate(0);
    int64_t geo_sel_end = geo.live ? p.sel_off[geo.c + 1] : 0;
    float nx[8];
    fetch(geo, nx);
    Fft8Tw twr;

    // mel: thread t owns chunk t of the filterbank (<= 16 consecutive bins of one triangle) and, for t < n_mels, band t
    const bool has_chunk = want_mel && tid < tb.mel_chunks;
    const int mel_bin = has_chunk ? tb.mel_chunk_bin[tid] : 0;
    int band_c0 = 0, band_c1 = 0;             // band (tid mod 128): threads 0..127 sum it for the pair's first frame, 128..255 for the second
    if (want_mel && (tid & 127) < p.n_mels) { band_c0 = tb.mel_band_chunk[tid & 127]; band_c1 = tb.mel_band_chunk[(tid & 127) + 1]; }
    float *pw1 = xs;                          // the SECOND frame's power spectrum lives in the frame buffer, dead once that frame's FFT has read it
    float *part1 = xs + 1040;                 // ... and its chunk sums behind it
    if (tid < 15) pw[1025 + tid] = 0.0f;         // the last chunks read (zero-weighted) bins past 1024

    // ---- prologue: running energy of every frame of the workgroup, one frame per lane ------------------------
    if (want_pyin) {
        // The workgroup's frames are normally consecutive frames of one clip: their samples (one contiguous stretch,
        // <= 15 hops + 2048 samples) are staged with coalesced loads in the FFT buffer + frame area, which nothing
        // uses yet, so the serial walk reads LDS instead of waiting for a global load per eight samples.
        const Geo g0 = geo;
        Geo gl = g0;
        if (fs0 + nfr - 1 < geo_sel_end) { gl.start += (int64_t)(nfr - 1) * p.hop; } else gl = locate(nfr - 1);
        const int64_t span = (int64_t)(nfr - 1) * p.hop + 2048;
        const bool staged = g0.c == gl.c && p.hop == 512 && span + 4 * (span >> 9) <= 10240;
        float *stage = reinterpret_cast<float *>(z);
        if (staged) {                                // sample i sits at i + 4 (i / 512): the lanes' frames start one hop apart,
            // the skew puts their 16-byte reads on distinct bank groups.  Four samples per thread and request, all of a
            // thread's (<= 10) requests in flight together: one sample per request and iteration made the staging, not the
            // walk, two thirds of the prologue (a memory latency per iteration, 38 iterations).
            const float *__restrict__ x0 = p.pcm + g0.base;
            const int n4 = (int)(span >> 2);                 // span is a multiple of 4 (hop = 512)
            float4 sv[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int q = tid + 256 * u;
                const int64_t idx = g0.start + 4 * (int64_t)q;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (q < n4) {
                    if (idx >= 0 && idx + 3 < g0.n) {
                        // clip offsets are arbitrary, so the address is only 4-byte aligned: the vector type says so (the
                        // load is still one global_load_dwordx4, but nothing may assume 16-byte alignment)
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        const f4u t = *reinterpret_cast<const f4u *>(x0 + idx);
                        v = make_float4(t.x, t.y, t.z, t.w);
                    } else {                                 // clip edge: librosa's centre padding is zeros
                        if (idx >= 0 && idx < g0.n) v.x = x0[idx];
                        if (idx + 1 >= 0 && idx + 1 < g0.n) v.y = x0[idx + 1];
                        if (idx + 2 >= 0 && idx + 2 < g0.n) v.z = x0[idx + 2];
                        if (idx + 3 >= 0 && idx + 3 < g0.n) v.w = x0[idx + 3];
                    }
                }
                sv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int q = tid + 256 * u;
                if (q < n4) {                                // squared here, by all threads, not inside the serial walk
                    const int i = 4 * q;
                    *reinterpret_cast<float4 *>(stage + i + 4 * (i >> 9)) =
                        make_float4(sv[u].x * sv[u].x, sv[u].y * sv[u].y, sv[u].z * sv[u].z, sv[u].w * sv[u].w);
                }
            }
            __syncthreads();
        }
        // The walk is one dependent add after another: beside another workgroup's FFT waves on the same SIMD it would
        // get an issue slot only when they stall, while this workgroup's other three waves wait for it -- raise it.
        if (wid == 0) __builtin_amdgcn_s_setprio(3);
        if (wid == 0 && lane < nfr) {
            float *row = en + lane * en_stride;
            if (staged) {
                const float *srow = stage + lane * (512 + 4);
                energy_walk<true>([&](int j, float4 &a, float4 &b) {
                    const float *q = srow + j + 4 * (j >> 9);                // j + 8 <= 1024 + 8 nA + 8 < 2048: inside the frame
                    a = *reinterpret_cast<const float4 *>(q);
                    b = *reinterpret_cast<const float4 *>(q + 4);
                }, row, mp);
            } else {                                         // frames of two clips, or a hop the staging area cannot hold
                const Geo g = locate(lane);
                const float *__restrict__ x = p.pcm + g.base;
                energy_walk_plain([&](int j) {
                    const int64_t q = g.start + j;
                    return (q >= 0 && q < g.n) ? x[q] : 0.0f;
                }, row, mp);
            }
        }
        if (wid == 0) __builtin_amdgcn_s_setprio(0);
        if (staged) __syncthreads();                     // the staging area becomes the FFT buffer and the frame again
    }
    // (the first barrier of the frame loop publishes the rows)
    if (want_fft) fft8_load_twiddles(twr, tb.twiddle, tid);      // 40 registers: loaded after the walk, which wants its own
    FRM_TICK(0)

    for (int pr = 0; pr < nfr; pr += 2) {
        double2 P[2][5];
        int64_t fidx[2] = {0, 0};
        bool flive[2] = {false, false};
        int fclip[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 5; ++r) P[h][r] = make_double2(0.0, 0.0);
            if (pr + h >= nfr) continue;                      // odd tail: the pair's second frame does not exist (uniform)
            const bool live = geo.live;
            const int c = geo.c;
            const int64_t f = geo.f, fo = geo.o;
            fidx[h] = f; flive[h] = live; fclip[h] = c;
            if (tid == 0) frow[pr + h] = live ? f : -1;
#pragma unroll
            for (int r = 0; r < 8; ++r) xs[tid + r * 256] = nx[r];
            {
                const int cprev = geo.c;
                geo = locate_next(geo, pr + h + 1, geo_sel_end);
                if (geo.live && geo.c != cprev) geo_sel_end = p.sel_off[geo.c + 1];
            }
            fetch(geo, nx);                                   // in flight under everything below
            __syncthreads();
            FRM_TICK(1)

            // ---- feature.rms: np.mean(np.square(x), axis=-2) then sqrt, float32, NumPy's pairwise order -------
            if (want_rms && tid < 128) {
                const int bb = tid >> 3, a = tid & 7;
                const float *xb = xs + bb * 128 + a;
                float r = xb[0] * xb[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) { const float v = xb[8 * i]; r = r + v * v; }
                red[tid] = r;
            }
            if (!want_fft) {
                __syncthreads();
                if (want_rms && tid < 16) {
                    const float *r = red + tid * 8;
                    blk[tid] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                }
                __syncthreads();
            } else {
                // ---- packed forward FFT of (frame, reversed first half); pass 1 straight from the frame ----
                double2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = tid + 256 * q;
                    v[q] = make_double2((double)xs[i], i < 1024 ? (double)xs[1024 - i] : 0.0);
                }
                fft8_pass1_write(z, tid, v);
                __syncthreads();
                if (want_rms && tid < 16) {
                    const float *r = red + tid * 8;
                    blk[tid] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                }
                fft8_read8(z, tid, v);
                __syncthreads();
                fft8_pass_write<8>(z, tid, v, twr.p2);
                __syncthreads();
                fft8_read8(z, tid, v);
                __syncthreads();
                fft8_pass_write<64>(z, tid, v, twr.p3);
                __syncthreads();
                fft8_pass4(z, tid, twr);
                __syncthreads();
            }
            FRM_TICK(2)
            if (want_rms && tid == 0 && live) {
                float b0 = (blk[0] + blk[1]) + (blk[2] + blk[3]);
                float b1 = (blk[4] + blk[5]) + (blk[6] + blk[7]);
                float b2 = (blk[8] + blk[9]) + (blk[10] + blk[11]);
                float b3 = (blk[12] + blk[13]) + (blk[14] + blk[15]);
                const float total = 0.0f + ((b0 + b1) + (b2 + b3));
                p.out_rms[fo] = sqrtf(total / 2048.0f);
            }
            if (!want_fft) continue;

            // ---- A[k] = FFT(x)[k], B[k] = FFT(b)[k] from Z by symmetry; P = A*B; windowed power from A ----------
            // A thread owns FIVE CONSECUTIVE bins k = 5 tid .. 5 tid + 4 (205 threads cover 0..1024): the windowed spectrum
            // needs A[k - 1] and A[k + 1], which are then the thread's own neighbours -- seven (z[k], z[2048 - k]) pairs per
            // thread and frame instead of fifteen with bins 256 apart.  (Lanes 80 bytes apart: 16 lanes of a ds_read_b128
            // fall on 16 distinct bank groups.)  k is taken modulo 2048: A[-1] = conj(A[1]), A[1025] = conj(A[1023]).
            if (tid < 205) {
                auto zpair = [&](int k, double2 &zk, double2 &zn) { zk = z[zsw(k & 2047)]; zn = z[zsw((2048 - k) & 2047)]; };
                // A2 = 2 A and B2 = 2 B: the halves are folded into the powers of two applied later (P = A B arrives as 4 P in the
                // inverse transform, whose outputs are scaled by 1/8192 instead of 1/2048; the windowed spectrum is
                // A/2 - (A- + A+)/4 = (2 A2 - (A2- + A2+)) / 8) -- exact scalings, every value bit-identical
                auto Afrom = [](const double2 &zk, const double2 &zn) { return make_double2(zk.x + zn.x, zk.y - zn.y); };
                const int k0 = 5 * tid;
                double2 zk, zn, zk1, zn1;
                zpair(k0 - 1, zk, zn);
                double2 Am = Afrom(zk, zn);
                zpair(k0, zk, zn);
                double2 A = Afrom(zk, zn);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int k = k0 + r;
                    zpair(k + 1, zk1, zn1);
                    const double2 Ap = Afrom(zk1, zn1);
                    if (want_pyin) {
                        const double2 Bv = make_double2(zk.y + zn.y, zn.x - zk.x);
                        P[h][r] = c_mul(A, Bv);
                    }
                    if (want_mel) {
                        const float re = (float)(0.125 * (2.0 * A.x - (Am.x + Ap.x)));
                        const float im = (float)(0.125 * (2.0 * A.y - (Am.y + Ap.y)));
                        const float mag = (float)sqrt((double)re * (double)re + (double)im * (double)im);  // npy_hypotf
                        (h == 0 ? pw : pw1)[k] = mag * mag;
                    }
                    Am = A; A = Ap; zk = zk1; zn = zn1;
                }
            }
            if (want_mel && h == 1 && tid < 15) pw1[1025 + tid] = 0.0f;      // the last chunks read (zero-weighted) bins past 1024
            __syncthreads();
            FRM_TICK(3)
        }
        // ---- mel projection of BOTH frames of the pair: sparse Slaney triangles.  Every <= 16-bin chunk of a triangle is
        // one thread's float32 fma chain per frame (the 16 weights requested once per pair); a band then adds its
        // chunks' sums in order (the widest band has six), threads 0..127 for the first frame and 128..255 for the
        // second.  Clip maximum.  One section per pair instead of one per frame: half the barriers, half the weight
        // traffic, all four waves busy in the band phase.
        if (want_mel) {
            if (has_chunk) {
                const float4 *wp = reinterpret_cast<const float4 *>(tb.mel_chunk_w) + tid * 4;
                const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float *pp = (h == 0 ? pw : pw1) + mel_bin;
                    float acc = 0.0f;
                    acc = fmaf(w0.x, pp[0], acc); acc = fmaf(w0.y, pp[1], acc); acc = fmaf(w0.z, pp[2], acc); acc = fmaf(w0.w, pp[3], acc);
                    acc = fmaf(w1.x, pp[4], acc); acc = fmaf(w1.y, pp[5], acc); acc = fmaf(w1.z, pp[6], acc); acc = fmaf(w1.w, pp[7], acc);
                    acc = fmaf(w2.x, pp[8], acc); acc = fmaf(w2.y, pp[9], acc); acc = fmaf(w2.z, pp[10], acc); acc = fmaf(w2.w, pp[11], acc);
                    acc = fmaf(w3.x, pp[12], acc); acc = fmaf(w3.y, pp[13], acc); acc = fmaf(w3.z, pp[14], acc); acc = fmaf(w3.w, pp[15], acc);
                    (h == 0 ? part : part1)[tid] = acc;
                }
            }
            __syncthreads();
            {
                const int h = wid >> 1, bt = tid & 127;           // waves 0, 1: first frame; waves 2, 3: second
                const bool live = flive[h];
                const float *pt = h == 0 ? part : part1;
                float acc = 0.0f;
                if (bt < p.n_mels && live) {
                    // the band's chunk sums added in order; the first six (every band of the default bank has at most
                    // six) are requested together instead of one LDS round trip per chunk
                    const int c0 = band_c0, c1 = band_c1;
                    float pv[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) pv[k] = pt[min(c0 + k, 255)];
                    acc = c0 < c1 ? pv[0] : 0.0f;
#pragma unroll
                    for (int k = 1; k < 6; ++k) if (c0 + k < c1) acc = acc + pv[k];
                    for (int cidx = c0 + 6; cidx < c1; ++cidx) acc = acc + pt[cidx];
                    p.melpow[fidx[h] * p.n_mels + bt] = acc;
                }
                // powers are >= 0: float order == unsigned order of the bits
                const unsigned m = wave_umax(__float_as_uint(acc));
                if (lane == 0 && live) atomicMax(&p.clipmax[fclip[h]], m);
            }
            // pw, pw1 and the chunk sums are next written behind the barriers of the inverse FFT; without it the next pair's
            // first frame would land in the frame buffer while the second frame's chunk sums are still being read
            if (!want_pyin) __syncthreads();
        }
        FRM_TICK(4)
        if (!want_pyin) continue;

        // ---- one inverse FFT for both frames: conj(Q), Q = Hermitian extension of P0 + i*P1 ---------------------
        if (tid < 205) {                         // the thread's own five consecutive bins (see above)
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int k = 5 * tid + r;
                const double2 u = P[0][r], v = P[1][r];
                z[zsw(k)] = make_double2(u.x - v.y, -u.y - v.x);                          // conj(P0) - i conj(P1)
                if (k > 0 && k < 1024) z[zsw(2048 - k)] = make_double2(u.x + v.y, u.y - v.x);   // P0 - i P1
            }
        }
        __syncthreads();
        {
            double2 v[8];
            fft8_read8(z, tid, v);
            __syncthreads();
            fft8_pass1_write(z, tid, v);
            __syncthreads();
            fft8_read8(z, tid, v);
            __syncthreads();
            fft8_pass_write<8>(z, tid, v, twr.p2);
            __syncthreads();
            fft8_read8(z, tid, v);
            __syncthreads();
            fft8_pass_write<64>(z, tid, v, twr.p3);
            __syncthreads();
            fft8_pass4_lags(z, tid, twr, mp);          // only the lags the difference function reads
            __syncthreads();
        }
        FRM_TICK(5)
        // FFT(conj Q) = N * conj(acf0 + i acf1); difference function with librosa's clamps (pitch.py::_cumulative_mean_normalized_difference)
        // With the CMND formed in this kernel's epilogue, the FIRST frame's row stays in LDS: the pair's two energy rows are
        // dead once read, and together they are exactly one float64 row (8 (max_period + 1) <= 8 en_stride bytes).  Only the
        // second frame's row makes the trip through global memory and back.
        const bool keep_first = p.cmnd_in_frame != 0;
        if (keep_first) {
            float es[2][3];                  // en[0] + en[tau] of both frames (float32, librosa's clamps), read before the rows are overwritten
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *row = en + (pr + h) * en_stride;
                float en0 = row[0];
                if (fabsf(en0) < 1e-6f) en0 = 0.0f;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    float e = row[min(tid + 256 * u, mp)];
                    if (fabsf(e) < 1e-6f) e = 0.0f;
                    es[h][u] = en0 + e;
                }
            }
            __syncthreads();                 // every thread has read its energies: the two rows become the first frame's d row
            double *__restrict__ d0 = reinterpret_cast<double *>(en + pr * en_stride);
            double *__restrict__ d1 = p.dfn + fidx[1] * (int64_t)p.lag_stride;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int tau = tid + 256 * u;
                if (tau <= mp) {
                    const double2 zz = z[zsw(1024 + tau)];
                    double a0 = zz.x * (1.0 / 8192.0), a1 = -zz.y * (1.0 / 8192.0);      // 1/2048 of the transform, 1/4 of P (see A2, B2)
                    if (fabs(a0) < 1e-6) a0 = 0.0;
                    if (fabs(a1) < 1e-6) a1 = 0.0;
                    if (flive[0]) d0[tau] = (double)es[0][u] - 2.0 * a0;
                    if (flive[1]) d1[tau] = (double)es[1][u] - 2.0 * a1;
                }
            }
        } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!flive[h]) continue;
            const float *row = en + (pr + h) * en_stride;
            float en0 = row[0];
            if (fabsf(en0) < 1e-6f) en0 = 0.0f;
            double *__restrict__ drow = p.dfn + fidx[h] * (int64_t)p.lag_stride;
            for (int tau = tid; tau <= mp; tau += 256) {
                const double2 zz = z[zsw(1024 + tau)];
                double a = (h == 0 ? zz.x : -zz.y) * (1.0 / 8192.0);
                if (fabs(a) < 1e-6) a = 0.0;
                float e = row[tau];
                if (fabsf(e) < 1e-6f) e = 0.0f;
                const float esum = en0 + e;
                drow[tau] = (double)esum - 2.0 * a;
            }
        }
        }
        // the next pair's first write into z (pass 1) comes after that frame's first barrier
        FRM_TICK(6)
    }
    // ---- epilogue: cumulative-mean-normalised difference of ALL the workgroup's frames -------------------------------
    // yin[tau] = d[tau] / (cumsum(d[1:])[tau] / tau + tiny), pitch.py::_cumulative_mean_normalized_difference.  np.cumsum is
    // strictly sequential in float64: one lane per frame walks it, all frames of the workgroup in one wave's lanes at once
    // (pyin_obs_kernel spent a third of its time issuing whole-wave instructions for one lane's walk, frame after frame).
    // The rows of the even frames never left LDS (the pair loop wrote them over the pair's energy rows); those of the odd
    // frames come back from L2 / the fabric (this workgroup wrote them a moment ago) into the FFT buffer, which is dead by
    // now.  Every thread keeps its share of d in registers, the walk turns the LDS copy into the cumsum in place, the
    // quotients run on all threads and go out over d[min_period..]: pyin_obs_kernel reads the CMND directly.
    if (want_pyin && p.cmnd_in_frame) {
        const int RS = frame_cmnd_stride(mp), minp = p.min_period;
        // rows of the even frames: where the pair loop left them (over the pair's energy rows); rows of the odd frames: back
        // from global memory into the FFT buffer / frame / power area
        double *zrows = reinterpret_cast<double *>(fsm);
        auto row_of = [&](int i) { return (i & 1) ? zrows + (i >> 1) * RS : reinterpret_cast<double *>(en + i * en_stride); };
        __syncthreads();                         // the d rows are written (workgroup-scope release), the other LDS buffers are free
        double dv[kFramesPerWg][3];
        int64_t fr[kFramesPerWg];
#pragma unroll
        for (int i = 0; i < kFramesPerWg; ++i) fr[i] = i < nfr ? frow[i] : -1;
#pragma unroll
        for (int i = 0; i < kFramesPerWg; ++i) {
            const double *__restrict__ src = (i & 1) ? p.dfn + (fr[i] < 0 ? 0 : fr[i]) * (int64_t)p.lag_stride : row_of(i);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int tau = tid + 256 * u;
                dv[i][u] = (fr[i] >= 0 && tau <= mp) ? src[tau] : 0.0;
            }
        }
#pragma unroll
        for (int i = 1; i < kFramesPerWg; i += 2) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int tau = tid + 256 * u;
                if (i < nfr && tau <= mp) zrows[(i >> 1) * RS + tau] = dv[i][u];
            }
        }
        __syncthreads();
        FRM_TICK(7)
        if (wid == 0) __builtin_amdgcn_s_setprio(3);
        if (wid == 0 && lane < nfr) cmnd_walk(row_of(lane), mp);
        if (wid == 0) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        FRM_TICK(8)
        const bool troughs = p.troughs != 0;
#pragma unroll
        for (int i = 0; i < kFramesPerWg; ++i) {
            double *__restrict__ drow = p.dfn + (fr[i] < 0 ? 0 : fr[i]) * (int64_t)p.lag_stride;
            double *__restrict__ cs = row_of(i);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int tau = tid + 256 * u;
                if (fr[i] >= 0 && tau >= minp && tau <= mp) {
                    const double yv = dv[i][u] / (cs[tau] / (double)tau + DBL_MIN);
                    if (troughs) cs[tau] = yv;          // the CMND stays in LDS, over the cumsum entry this thread has just read
                    else drow[tau] = yv;
                }
            }
        }
        FRM_TICK(9)
        // ---- troughs of the CMND, compacted (round 4).  pyin_obs_kernel used to load the 495-lag CMND row of every frame
        // from memory and spend a sixth of its time finding the local minima; here the row is in LDS anyway, so every wave
        // takes the frames wid, wid + 4, ... of the workgroup, finds the troughs exactly as pyin_obs_kernel does (util.localmin
        // plus the special first element) and writes, in ascending lag
        // order, each trough's CMND value, its parabolic shift (pitch.py::_parabolic_interpolation: a function of the three
        // CMND values around it) and its lag index: a few hundred bytes per frame instead of the 4 KB row, and the
        // observation kernel starts at the threshold prior.
        if (troughs) {
            __syncthreads();
            const int nl = p.n_lags, KM = trough_km(nl), NR = (nl + 63) >> 6;
            const unsigned long long below = (1ull << lane) - 1ull;
            for (int i = wid; i < nfr; i += 4) {
                if (fr[i] < 0) continue;                                  // (uniform)
                const double *__restrict__ y = row_of(i) + minp;
                double *__restrict__ trow = p.dfn + fr[i] * (int64_t)p.lag_stride;
                int16_t *__restrict__ tiv = reinterpret_cast<int16_t *>(trow + 8 + 2 * KM);
                // lag k = 64 r + lane: consecutive lanes read consecutive doubles (a contiguous chunk per lane, as
                // pyin_obs_kernel walks the row, puts 32 lanes on one pair of banks); ascending lag order = round after
                // round, lane after lane, so a round's ballot and a population count place its troughs
                int K = 0;
                for (int r = 0; r < NR; ++r) {
                    const int k = 64 * r + lane;
                    bool tr = false;
                    double ym = 0.0, y0 = 0.0, yp = 0.0;
                    if (k < nl) {
                        y0 = y[k];
                        if (k > 0) ym = y[k - 1];
                        if (k < nl - 1) yp = y[k + 1];
                        if (k == 0) tr = y0 < yp;
                        else if (k == nl - 1) tr = y0 < ym;
                        else tr = (y0 < ym) && (y0 <= yp);
                    }
                    const unsigned long long m = __ballot(tr);
                    if (tr) {
                        const int pos = K + __popcll(m & below);
                        double shift = 0.0;
                        if (k > 0 && k < nl - 1) {
                            const double a = yp + ym - 2.0 * y0;
                            const double b = (yp - ym) / 2.0;
                            if (fabs(b) < fabs(a)) shift = -b / a;
                        }
                        trow[8 + pos] = y0;
                        trow[8 + KM + pos] = shift;
                        tiv[pos] = (int16_t)k;
                    }
                    K += __popcll(m);
                }
                if (lane == 0) trow[0] = __longlong_as_double((long long)K);
            }
        }
    }
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 128)
    if (blockIdx.x == 1000 && (tid == 0 || tid == 64)) { for (int k = 0; k < 10; ++k) g_frm_dbg[(tid ? 12 : 0) + k] = facc[k]; }
#endif
}
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 128)
hipError_t frame_debug_fetch(long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_frm_dbg), sizeof(long long) * 24); }
#else
hipError_t frame_debug_fetch(long long *dst) { for (int i = 0; i < 24; ++i) dst[i] = 0; return hipSuccess; }
#endif

// ------------------------------------------------------------------------------------------
// Kernel 3: one frame per wave.  Troughs of the CMND, the Beta/Boltzmann threshold prior,
// parabolic refinement, pitch-bin observation row in the log domain.
// ------------------------------------------------------------------------------------------
constexpr int kKMax = 512;    // troughs per frame (n_lags <= 1023)
constexpr int kMaxRounds = 8; // kKMax / 64

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imin(int v) {
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_min_i32(int v) {   // uniform result
    v = dpp_imin<0x111, 0xf>(v);   // row_shr:1
    v = dpp_imin<0x112, 0xf>(v);   // row_shr:2
    v = dpp_imin<0x114, 0xf>(v);   // row_shr:4
    v = dpp_imin<0x118, 0xf>(v);   // row_shr:8
    v = dpp_imin<0x142, 0xa>(v);   // row_bcast:15
    v = dpp_imin<0x143, 0xc>(v);   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// One frame per wave; a workgroup is up to 8 waves that share ONE copy of the scipy tables in LDS (16 waves per CU instead
// of the 10 a private copy allowed) and walk `frames_per_wave` consecutive frames each.  The waves never exchange data: after
// the table load the only synchronisation is inside a wave, where LDS operations complete in order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 256)
__device__ long long g_obs_dbg[16];
#define OBS_TICK(k) { const long long now__ = clock64(); oacc[k] += now__ - olast; olast = now__; }
#else
#define OBS_TICK(k)
#endif
__global__ __launch_bounds__(512, 4) void pyin_obs_kernel(PassParams p, DevTables tb, int frames_per_wave) {
    // dynamic LDS: bfact[KM+1] | bexp[KM+1] | bcum[101] (shared), then per wave y[YN] (CMND, reused as the output row) | U,
    // where U holds first the difference function dd[DN] and later, once the CMND is formed, the trough arrays th[KM],
    // tp[KM], ti[KM], tbin[KM]
    extern __shared__ __align__(16) unsigned char osm[];
    const int nl = p.n_lags, B = p.n_bins;
    const int KM = nl / 2 + 2;
    const int YN = (max(nl, B) + 1) & ~1;
    const int DN = (p.max_period + 1 + 32 + 1) & ~1;      // + look-ahead of the cumsum walk
    const int UN = (max(DN, 2 * KM + (4 * KM + 7) / 8) + 1) & ~1;    // doubles, even: every wave's arrays start on 16 bytes
    const int TN = (2 * (KM + 1) + 101 + 1) & ~1;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int)(blockDim.x >> 6);
    double *bfact = reinterpret_cast<double *>(osm);      // scipy.stats.boltzmann pieces and the Beta mass prefix sums: every lookup
    double *bexp = bfact + (KM + 1);                      // below is data dependent, so they sit in LDS instead of behind a global
    double *bcum = bexp + (KM + 1);                       // load each
    double *y = bfact + TN + (size_t)wid * (YN + UN);
    double *row = y;                       // written only after the last read of y
    double *dd = y + YN;
    double *th = dd;
    double *tp = th + KM;
    int16_t *ti = reinterpret_cast<int16_t *>(tp + KM);
    int16_t *tbin = ti + KM;
    __shared__ double beta_s[104];

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 256)
    long long oacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, olast = clock64();
#endif
    for (int i = threadIdx.x; i < 100; i += blockDim.x) beta_s[i] = tb.beta_probs[i];
    for (int i = threadIdx.x; i <= KM; i += blockDim.x) { bfact[i] = tb.boltz_fact[i]; bexp[i] = tb.boltz_exp[i]; }
    for (int i = threadIdx.x; i <= 100; i += blockDim.x) bcum[i] = tb.beta_cumsum[i];
    __syncthreads();

    int64_t f = 0, fo = 0, sel_end = 0;                   // the wave's frames are normally consecutive frames of one clip:
    for (int it = 0; it < frames_per_wave; ++it) {        // the binary search of map_frame runs once, not per frame
    const int64_t fsel = ((int64_t)blockIdx.x * nwaves + wid) * frames_per_wave + it;
    if (fsel >= geo_n_sel(p)) break;                      // wave-uniform
    if (it > 0 && fsel < sel_end) { ++f; ++fo; }
    else {
        int c;
        int64_t t;
        map_frame(p, fsel, c, t, f);
        fo = out_index(p, c, t);
        sel_end = p.sel_off[c + 1];
    }
    // Cumulative-mean-normalised difference (pitch.py::_cumulative_mean_normalized_difference) from the difference
    // function the frame stage left in HBM: yin[tau] = d[tau] / (cumsum(d[1:])[tau] / tau + tiny).  np.cumsum is strictly
    // sequential in float64, so ONE lane walks it (the other waves of the CU cover its latency); the quotients run on
    // all lanes.  The CMND itself never leaves the CU.
    const int mp = p.max_period, minp = p.min_period;
    const double *__restrict__ dr = p.dfn + f * (int64_t)p.lag_stride;
    wave_sync();                            // the previous frame's output row has been read out of this wave's buffers
    int K = 0;
    double *tsh = y;                        // trough shifts (PassParams::troughs) sit where the CMND row would: read before `row` is written
    if (p.troughs) {
        // the frame kernel's epilogue has found the troughs already (frame_yin_kernel): count, values, parabolic shifts, lags
        const int KMt = trough_km(nl);
        K = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(dr[0]));
        const int16_t *__restrict__ tiv = reinterpret_cast<const int16_t *>(dr + 8 + 2 * KMt);
        for (int k = lane; k < K; k += 64) {
            const double h = dr[8 + k], sh = dr[8 + KMt + k];
            const int16_t ix = tiv[k];
            th[k] = h; tsh[k] = sh; ti[k] = ix;
        }
        wave_sync();
        OBS_TICK(0)
    } else {
    if (p.cmnd_in_frame) {
        // the frame kernel's epilogue has formed the CMND already (dfn[min_period..max_period] holds it): load and go on
        for (int base = lane; base < nl; base += 320) {
            double t5[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) t5[u] = dr[minp + min(base + 64 * u, nl - 1)];
#pragma unroll
            for (int u = 0; u < 5; ++u) if (base + 64 * u < nl) y[base + 64 * u] = t5[u];
        }
        wave_sync();
        OBS_TICK(0)
    } else {
    // five requests per lane in flight, then their five LDS stores (one request per iteration waited a memory latency
    // nine times: 13 k of a frame's 88 k cycles; wider batches cost registers this kernel does not have)
    for (int base = lane; base <= mp; base += 320) {
        double t5[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) t5[u] = dr[min(base + 64 * u, mp)];
#pragma unroll
        for (int u = 0; u < 5; ++u) if (base + 64 * u <= mp) dd[base + 64 * u] = t5[u];
    }
    wave_sync();
    OBS_TICK(0)
    if (lane == 0) {
        // Straight-line groups of 8 lags over two register sets: a group's values are requested (16-byte reads) while
        // the group before it is chained, so the walk costs its dependent adds and little else -- every instruction
        // here takes a whole issue slot for one lane's work.  dd is padded: the look-ahead stays inside the array.
        double cs = dd[1];
        if (minp <= 1) y[1 - minp] = cs;
        int tau = 2;
        constexpr int GL = 8;                 // lags per group (two sets of GL doubles stay in registers)
        double2 ra[GL / 2], rb[GL / 2];
        auto request = [&](double2 (&r)[GL / 2], int t0) {
#pragma unroll
            for (int i = 0; i < GL / 2; ++i) r[i] = *reinterpret_cast<const double2 *>(dd + t0 + 2 * i);
        };
        auto group = [&](auto all_tag, double2 (&cur)[GL / 2], double2 (&nxt)[GL / 2], int t0) {
            constexpr bool ALL = decltype(all_tag)::value;       // every lag of the group is >= min_period
            request(nxt, t0 + GL);
            double *yo = y + (t0 - minp);
#pragma unroll
            for (int i = 0; i < GL / 2; ++i) {
                cs = cs + cur[i].x; if (ALL || t0 + 2 * i >= minp) yo[2 * i] = cs;
                cs = cs + cur[i].y; if (ALL || t0 + 2 * i + 1 >= minp) yo[2 * i + 1] = cs;
            }
        };
        request(ra, tau);
        bool in_a = true;
        for (; tau < minp && tau + GL - 1 <= mp; tau += GL, in_a = !in_a) {   // groups that start below min_period (few)
            if (in_a) group(std::false_type{}, ra, rb, tau); else group(std::false_type{}, rb, ra, tau);
        }
        if (!in_a && tau + GL - 1 <= mp) { group(std::true_type{}, rb, ra, tau); tau += GL; }
        for (; tau + 2 * GL - 1 <= mp; tau += 2 * GL) { group(std::true_type{}, ra, rb, tau); group(std::true_type{}, rb, ra, tau + GL); }
        if (tau + GL - 1 <= mp) { group(std::true_type{}, ra, rb, tau); tau += GL; }
        for (; tau <= mp; ++tau) { cs = cs + dd[tau]; if (tau >= minp) y[tau - minp] = cs; }
    }
    wave_sync();
    OBS_TICK(1)
    for (int i = lane; i < nl; i += 64) {
        const int tau = i + minp;
        y[i] = dd[tau] / (y[i] / (double)tau + DBL_MIN);
    }
    if (p.yin != nullptr) {                 // stage-level parity tests only (AEGIS_DEBUG_STAGES=1)
        wave_sync();
        double *__restrict__ yo = p.yin + f * (int64_t)p.yin_stride;
        for (int i = lane; i < nl; i += 64) yo[i] = y[i];
    }
    wave_sync();
    }   // CMND formed here

    OBS_TICK(2)
    // troughs: util.localmin plus the special first element; contiguous lag chunk per lane
    const int CH = (nl + 63) >> 6;
    unsigned mask = 0;
    int cnt = 0;
    for (int r = 0; r < CH; ++r) {
        const int i = lane * CH + r;
        if (i < nl) {
            const double yi = y[i];
            bool tr;
            if (i == 0) tr = yi < y[1];
            else if (i == nl - 1) tr = yi < y[i - 1];
            else tr = (yi < y[i - 1]) && (yi <= y[i + 1]);
            if (tr) { mask |= 1u << r; ++cnt; }
        }
    }
    const int incl = wave_incl_scan(cnt, lane);
    K = __shfl(incl, 63);
    {
        int k = incl - cnt;
        for (int r = 0; r < CH; ++r)
            if (mask & (1u << r)) { const int i = lane * CH + r; th[k] = y[i]; ti[k] = (int16_t)i; ++k; }
    }
    wave_sync();
    }   // troughs found here

    OBS_TICK(3)
    double vp = 0.0;
    unsigned segm = 0;               // 64-bin segments of the observation row with an observed bin (wave-uniform)
    const int rounds = (K + 63) >> 6;
    // Everything below is unrolled over the rounds of 64 troughs a frame may need (up to 8); nearly every frame has at
    // most 128 troughs, so the body exists twice: the two-round instance is what runs (a quarter of the code: the three
    // frame-stage kernels and the Viterbi share instruction caches), the eight-round one covers the rest.
    auto tail = [&](auto maxr_tag) {
        constexpr int MAXR = decltype(maxr_tag)::value;
        if (K > 0) {
            int jk[MAXR];
            double acc[MAXR];
            int jmin = 101;
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                jk[q] = 101; acc[q] = 0.0;
                if (q < rounds) {
                    const int k = q * 64 + lane;
                    if (k < K) {
                        // first threshold index j with h < thresholds[j+1]; 100 when none
                        const double h = th[k];
                        int g;
                        if (!(h < 1.0)) g = 100;          // thresholds[100] == 1.0 (also NaN)
                        else if (h <= 0.0) g = 0;
                        else g = (int)(h * 100.0);
                        // thresholds = np.linspace(0, 1, 101): i * 0.01 (one rounding), the last one forced to 1.0
                        auto thr = [](int i) { return i >= 100 ? 1.0 : (double)i * 0.01; };
                        while (g < 100 && !(h < thr(g + 1))) ++g;
                        while (g > 0 && h < thr(g)) --g;
                        jk[q] = g;
                        jmin = min(jmin, g);
                    }
                }
            }
            jmin = wave_min_i32(jmin);
            OBS_TICK(4)

            // probs[k] = sum_j [h_k < thr_{j+1}] * boltzmann.pmf(pos_k(j); 2, n_j) * beta_probs[j],
            // products added in ascending j (the order the oracle fixes for librosa's BLAS dot).
            // The set of troughs below threshold j only changes where j passes some trough's first
            // threshold, so the prior is rebuilt at those change points only; inside a stretch every
            // j still contributes its own rounded product, which keeps the sum bit-identical.
            const unsigned long long lt_mask = (1ull << lane) - 1ull;
            int j = jmin;
            while (j < 100) {
                unsigned long long M[MAXR];
                int nj = 0, nxt = 100;
    #pragma unroll
                for (int q = 0; q < MAXR; ++q) {
                    M[q] = 0ull;
                    if (q < rounds) {
                        M[q] = __ballot(jk[q] <= j);
                        nj += __popcll(M[q]);
                        if (jk[q] > j) nxt = min(nxt, jk[q]);
                    }
                }
                nxt = wave_min_i32(nxt);
                const double fact = bfact[nj];
                double prior[MAXR];
                int before = 0;
    #pragma unroll
                for (int q = 0; q < MAXR; ++q) {
                    prior[q] = 0.0;
                    if (q < rounds) {
                        if (jk[q] <= j) prior[q] = fact * bexp[before + __popcll(M[q] & lt_mask)];
                        before += __popcll(M[q]);
                    }
                }
                for (int jj = j; jj < nxt; ++jj) {
                    const double bj = beta_s[jj];
    #pragma unroll
                    for (int q = 0; q < MAXR; ++q)
                        if (q < rounds) acc[q] = acc[q] + prior[q] * bj;
                }
                j = nxt;
            }

            OBS_TICK(5)
            // global minimum trough (first index on ties) gets the no-trough mass
            double bh = INFINITY;
            int bk = kKMax;
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                const int k = q * 64 + lane;
                if (q < rounds && k < K) { const double h = th[k]; if (h < bh) { bh = h; bk = k; } }
            }
    #pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double oh = __shfl_xor(bh, o);
                const int ok = __shfl_xor(bk, o);
                if (oh < bh || (oh == bh && ok < bk)) { bh = oh; bk = ok; }
            }
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                if (q < rounds && q * 64 + lane == bk) {
                    const int nb = jk[q] > 100 ? 100 : jk[q];
                    acc[q] = acc[q] + 0.01 * bcum[nb];
                }
            }

            // parabolic refinement and pitch bin for every trough that carries probability
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                const int k = q * 64 + lane;
                if (q < rounds && k < K) {
                    const double pr = acc[q];
                    int bin = -1;
                    if (pr != 0.0) {
                        const int i = ti[k];
                        double shift = 0.0;
                        if (p.troughs) shift = tsh[k];
                        else if (i > 0 && i < nl - 1) {
                            const double ym = y[i - 1], y0 = y[i], yp = y[i + 1];
                            const double a = yp + ym - 2.0 * y0;
                            const double b = (yp - ym) / 2.0;
                            if (fabs(b) < fabs(a)) shift = -b / a;
                        }
                        const double period = (double)(p.min_period + i) + shift;
                        const double f0c = (double)p.sr / period;
                        double r = rint(120.0 * log2(f0c / p.fmin));
                        r = !(r >= 0.0) ? 0.0 : (r > (double)B ? (double)B : r);      // NaN-safe (np.clip would keep NaN; no finite input gets here with one)
                        bin = (int)r;
                    }
                    tp[k] = pr;
                    tbin[k] = (int16_t)bin;
                }
            }
        }
        OBS_TICK(6)
        wave_sync();                 // last read of y is behind us: the buffer becomes the output row
        for (int b = lane; b < B; b += 64) row[b] = p.log_tiny;
        wave_sync();
        if (K > 0) {
            // observation_probs[bin, t] = probs: on duplicate bins the largest lag wins; bins are
            // non-increasing in lag, so a trough loses exactly when the next trough with
            // probability has the same bin.  Bin == B falls in the unvoiced half and is dropped.
            bool winq[MAXR];
            double prq[MAXR];
            unsigned lseg = 0;           // segments of the row this lane's winners fall in
            // the next trough that carries probability, for every trough at once: one ballot per round of 64 troughs and a
            // per-lane shift instead of a walk over tbin (a data-dependent loop of dependent LDS reads per lane)
            int binq[MAXR];
            unsigned long long hasm[MAXR];
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                const int k = q * 64 + lane;
                binq[q] = (q < rounds && k < K) ? (int)tbin[k] : -1;
                hasm[q] = __ballot(binq[q] >= 0);
            }
    #pragma unroll
            for (int q = 0; q < MAXR; ++q) {
                winq[q] = false; prq[q] = 0.0;
                const int bin = binq[q];
                if (bin >= 0 && bin < B) {
                    const unsigned long long above = lane < 63 ? hasm[q] >> (lane + 1) : 0ull;
                    int k2 = -1;
                    if (above) k2 = q * 64 + lane + (int)__ffsll((long long)above);
                    else {
    #pragma unroll
                        for (int q2 = MAXR - 1; q2 > q; --q2)
                            if (hasm[q2]) k2 = q2 * 64 + (int)__ffsll((long long)hasm[q2]) - 1;
                    }
                    const bool win = (k2 < 0) || ((int)tbin[k2 < 0 ? 0 : k2] != bin);
                    if (win) { prq[q] = tp[q * 64 + lane]; row[bin] = log(prq[q] + DBL_MIN); lseg |= 1u << (bin >> 6); }
                    winq[q] = win;
                }
            }
            // voiced_prob = sum over bins in increasing bin order = decreasing lag order: the winners' probabilities are
            // pulled out of the lanes' registers from the highest trough down (same float64 adds in the same order as a
            // serial walk, without an LDS round trip per trough)
            double s = 0.0;
    #pragma unroll
            for (int q = MAXR - 1; q >= 0; --q) {
                if (q < rounds) {
                    unsigned long long m = __ballot(winq[q]);
                    while (m) {
                        const int l = 63 - __clzll((long long)m);
                        const double v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(prq[q]), l),
                                                          __builtin_amdgcn_readlane(__double2loint(prq[q]), l));
                        s = s + v;
                        m &= ~(1ull << l);
                    }
                }
            }
            vp = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
            for (int sg = 0; sg * 64 < B; ++sg)
                if (__ballot((lseg >> sg) & 1u)) segm |= 1u << sg;
        }
    };
    if (rounds <= 2) tail(std::integral_constant<int, 2>{});
    else tail(std::integral_constant<int, kMaxRounds>{});
    wave_sync();
    OBS_TICK(7)
    // Only the 64-bin segments that hold an observed bin are stored (the Viterbi kernel skips a voiced wave whose segment
    // is all log(tiny) at an easy frame and never loads it); at a hard frame -- unvoiced observation log(tiny) -- every
    // value matters and the whole row goes out.
    const double unv = (1.0 - vp) / (double)B;
    if (unv == 0.0) segm = 0x40000000u | ((1u << ((B + 63) >> 6)) - 1u);
    double *__restrict__ orow = p.logobs + f * (int64_t)p.obs_stride;
    for (int b = lane, sg = 0; b < B; b += 64, ++sg)
        if ((segm >> sg) & 1u) orow[b] = row[b];
    if (lane == 0) {
        p.obs_seg[f] = (int32_t)segm;
        p.logunv[f] = log(unv + DBL_MIN);
        if (p.out_vprob != nullptr) p.out_vprob[fo] = vp;
    }
    OBS_TICK(8)
    }   // frames of this wave
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 256)
    if (blockIdx.x == 100 && threadIdx.x == 0) { for (int k = 0; k < 9; ++k) g_obs_dbg[k] = oacc[k]; g_obs_dbg[9] = frames_per_wave; }
#endif
}
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 256)
hipError_t obs_debug_fetch(long long *dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_obs_dbg), sizeof(long long) * 16); }
#else
hipError_t obs_debug_fetch(long long *dst) { for (int i = 0; i < 16; ++i) dst[i] = 0; return hipSuccess; }
#endif

// ------------------------------------------------------------------------------------------
// Kernel 5a: f0 / voiced decode (pitch.py: f0 = freqs[state % B], voiced = state < B).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(PassParams p, DevTables tb) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= p.n_frames) return;
    const int s = p.states[f];
    const bool voiced = (unsigned)s < (unsigned)p.n_bins;      // anything else decodes as unvoiced
    const int c = find_clip(p.frame_off, p.n_clips, f);
    const int64_t fo = out_index(p, c, f - p.frame_off[c]);
    if (p.out_voiced != nullptr) p.out_voiced[fo] = voiced ? 1 : 0;
    if (p.out_f0 != nullptr) p.out_f0[fo] = voiced ? tb.freqs[s] : p.f0_unvoiced;
    if (p.out_bin != nullptr) p.out_bin[fo] = voiced ? (int16_t)s : (int16_t)-1;
}

// ------------------------------------------------------------------------------------------
// Kernel 5b: power_to_db(ref=np.max, top_db=80) per clip, the per-column broadband test of
// vision.py:11-21, and the [F][n_mels] -> [n_mels][F] transpose of the dB image through LDS.
// 64 frames per workgroup; wave w owns rows w*16 .. w*16+15.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void db_rake_kernel(PassParams p) {
    __shared__ float tile[64][129];
    __shared__ int rclip[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nm = p.n_mels;
    const int64_t fb = (int64_t)blockIdx.x * 64;
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wid * 16 + rr;
        const int64_t f = fb + r;
        if (f >= p.n_frames) { if (lane == 0) rclip[r] = -1; continue; }
        const int c = find_clip(p.frame_off, p.n_clips, f);
        if (lane == 0) rclip[r] = c;
        const float ref = fmaxf(1e-10f, __uint_as_float(p.clipmax[c]));
        const float refdb = 10.0f * (float)log10((double)ref);
        float cmax = -INFINITY;
        for (int m = lane; m < nm; m += 64) {
            const float s = fmaxf(1e-10f, p.melpow[f * nm + m]);
            float v = 10.0f * (float)log10((double)s);
            v = v - refdb;
            v = fmaxf(v, 0.0f - 80.0f);
            tile[r][m] = v;
            cmax = fmaxf(cmax, v);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o));
        int active = 0;
        const float thr = cmax - 20.0f;
        for (int m0 = 0; m0 < nm; m0 += 64) {
            const int m = m0 + lane;
            const bool on = (m < nm) && (tile[r][m] > thr);
            active += __popcll(__ballot(on));
        }
        if (lane == 0) {
            bool cand = false;
            if (!(cmax < -60.0f)) cand = ((double)active / (double)nm) > p.rake_ratio;
            p.rake_raw[f] = cand ? 1 : 0;
        }
    }
    __syncthreads();
    if (p.out_colmean != nullptr && tid < 64 && rclip[tid] >= 0) {
        // np.mean(S_dB, axis=0) and the two half-image means of guitar_specific.py:60-141: float32 sums row after row
        // (NumPy reduces a C-ordered [n_mels, F] array over axis 0 one row at a time), divided by the row count
        const int c = rclip[tid], mid = nm / 2;
        const float *row = tile[tid];
        float a = row[0];
        for (int m = 1; m < mid; ++m) a = a + row[m];
        const float lo_sum = a;
        float hs = row[mid];
        for (int m = mid + 1; m < nm; ++m) hs = hs + row[m];
        for (int m = mid; m < nm; ++m) a = a + row[m];
        const int64_t fo = out_index(p, c, fb + tid - p.frame_off[c]);
        p.out_colmean[fo] = a / (float)nm;
        p.out_colmean[p.out_total + fo] = lo_sum / (float)mid;
        p.out_colmean[2 * p.out_total + fo] = hs / (float)(nm - mid);
    }
    if (p.out_sdb != nullptr) {
        for (int idx = tid; idx < 64 * nm; idx += 256) {
            const int m = idx >> 6, r = idx & 63;
            const int c = rclip[r];
            if (c < 0) continue;
            const int64_t fo = p.frame_off[c];
            const int64_t Fc = p.frame_off[c + 1] - fo;
            const int64_t tl = fb + r - fo;
            p.out_sdb[(int64_t)nm * p.out_off[c] + (int64_t)m * Fc + tl] = tile[r][m];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Kernel 5c: run-length filter of vision.py:27-36.  A candidate frame survives when its run
// is closed before the end of the clip and min_frames <= length <= max_frames.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rake_runs_kernel(PassParams p) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= p.n_frames || p.out_rake == nullptr) return;
    uint8_t keep = 0;
    if (p.rake_raw[f]) {
        const int c = find_clip(p.frame_off, p.n_clips, f);
        const int64_t lo = p.frame_off[c], hi = p.frame_off[c + 1];
        const int lim = p.rake_max_frames + 1;
        int64_t s = f, e = f + 1;
        int steps = 0;
        while (s > lo && p.rake_raw[s - 1] && steps <= lim) { --s; ++steps; }
        while (e < hi && p.rake_raw[e] && steps <= lim) { ++e; ++steps; }
        const int64_t len = e - s;
        const bool closed = e < hi;   // a run still open at the end of the clip is dropped
        if (steps <= lim && closed && len >= p.rake_min_frames && len <= p.rake_max_frames) keep = 1;
    }
    {
        const int c = find_clip(p.frame_off, p.n_clips, f);
        p.out_rake[out_index(p, c, f - p.frame_off[c])] = keep;
    }
}

// ------------------------------------------------------------------------------------------
// Stand-alone rake detection on a caller-supplied dB image [n_mels][F] (the reference's
// AegisEngine.detect_rake_patterns(S_dB), aegis_engine.py:38-39): column test + run filter.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rake_cols_kernel(const float *__restrict__ sdb, int n_mels, int64_t F,
                                                        double ratio, uint8_t *__restrict__ raw) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= F) return;
    float cmax = -INFINITY;
    for (int m = 0; m < n_mels; ++m) cmax = fmaxf(cmax, sdb[(int64_t)m * F + t]);
    bool cand = false;
    if (!(cmax < -60.0f)) {
        const float thr = cmax - 20.0f;
        int active = 0;
        for (int m = 0; m < n_mels; ++m) active += sdb[(int64_t)m * F + t] > thr ? 1 : 0;
        cand = ((double)active / (double)n_mels) > ratio;
    }
    raw[t] = cand ? 1 : 0;
}

__global__ __launch_bounds__(256) void rake_runs_simple_kernel(const uint8_t *__restrict__ raw, int64_t F,
                                                               int min_frames, int max_frames,
                                                               uint8_t *__restrict__ out) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    uint8_t keep = 0;
    if (raw[f]) {
        const int lim = max_frames + 1;
        int64_t s = f, e = f + 1;
        int steps = 0;
        while (s > 0 && raw[s - 1] && steps <= lim) { --s; ++steps; }
        while (e < F && raw[e] && steps <= lim) { ++e; ++steps; }
        const int64_t len = e - s;
        if (steps <= lim && e < F && len >= min_frames && len <= max_frames) keep = 1;
    }
    out[f] = keep;
}

void launch_rake_from_db(const float *sdb, int n_mels, int64_t F, double ratio, int min_frames, int max_frames,
                         uint8_t *raw, uint8_t *out, hipStream_t s) {
    if (F == 0) return;
    const unsigned g = (unsigned)((F + 255) / 256);
    hipLaunchKernelGGL(rake_cols_kernel, dim3(g), dim3(256), 0, s, sdb, n_mels, F, ratio, raw);
    hipLaunchKernelGGL(rake_runs_simple_kernel, dim3(g), dim3(256), 0, s, raw, F, min_frames, max_frames, out);
}

// ------------------------------------------------------------------------------------------
// Streaming graph helpers.  stream_advance_kernel appends the pushed samples (already on the device in
// a fixed staging buffer) to the clip and derives the push's geometry exactly as the host path does:
// a frame is ready when its centred 2048-sample window is complete.  stream_gather_kernel packs the
// push's outputs into a fixed result block: [0] n_frames, then rms[8] f32, vprob[8] f64, live[8] i32.
// ------------------------------------------------------------------------------------------
__global__ void stream_advance_kernel(StreamCtl *ctl, const float *__restrict__ staging, int n_push,
                                      float *__restrict__ pcm, int hop) {
    const int64_t n0 = ctl->n_samples;
    for (int i = threadIdx.x; i < n_push; i += blockDim.x) pcm[n0 + i] = staging[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t n = n0 + n_push;
        const int64_t ready = n >= 1024 ? (n - 1024) / hop + 1 : 0;
        const int64_t lo = ctl->frames_done, hi = ready > lo ? ready : lo;
        ctl->n_samples = n;
        ctl->t_begin = lo; ctl->n_sel = hi - lo;
        ctl->vt_begin = lo; ctl->vt_end = hi;
        ctl->frames_done = hi;
        ctl->meta[1] = n;            // sample_off[1]
        ctl->meta[5] = hi - lo;      // sel_off[1]
    }
}

__global__ void stream_gather_kernel(const StreamCtl *ctl, const float *__restrict__ rms, const double *__restrict__ vprob,
                                     const int32_t *__restrict__ live, unsigned char *__restrict__ result) {
    const int i = threadIdx.x;
    int64_t *cnt = reinterpret_cast<int64_t *>(result);
    float *r = reinterpret_cast<float *>(result + 8);
    double *v = reinterpret_cast<double *>(result + 8 + 32);
    int32_t *l = reinterpret_cast<int32_t *>(result + 8 + 32 + 64);
    const int64_t n = ctl->n_sel, t0 = ctl->t_begin;
    if (i == 0) *cnt = n;
    if (i < 8 && i < n) { r[i] = rms[t0 + i]; v[i] = vprob[t0 + i]; l[i] = live[t0 + i]; }
}

void launch_stream_advance(StreamCtl *ctl, const float *staging, int n_push, float *pcm, int hop, hipStream_t s) {
    hipLaunchKernelGGL(stream_advance_kernel, dim3(1), dim3(256), 0, s, ctl, staging, n_push, pcm, hop);
}
void launch_stream_gather(const StreamCtl *ctl, const float *rms, const double *vprob, const int32_t *live, void *result,
                          hipStream_t s) {
    hipLaunchKernelGGL(stream_gather_kernel, dim3(1), dim3(64), 0, s, ctl, rms, vprob, live, static_cast<unsigned char *>(result));
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
hipError_t viterbi_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(frame_yin_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(pyin_obs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);       // + 832 B static
    if (e != hipSuccess) return e;
    return viterbi_set_lds_limits();
}

// frames per workgroup of a batch launch: as many as keep two workgroups on a CU (16 at the reference's rates)
static int frame_batch_fpw(int max_period) {
    const int stride = frame_en_stride(max_period);
    int fpw = kFramesPerWg;
    while (fpw > 2 && kFrameLdsFixed + (size_t)fpw * stride * 4 > 80 * 1024) fpw -= 2;
    return fpw;
}
int trough_row_doubles_host(int n_lags) { return trough_row_doubles(n_lags); }
bool frame_cmnd_supported(int max_period) {
    // the epilogue keeps three lags per thread and frame in registers and one row per frame in the LDS the loop has freed
    if (max_period + 1 > 3 * 256) return false;
    const int stride = frame_en_stride(max_period), rs = frame_cmnd_stride(max_period);
    if (stride < max_period + 1) return false;                   // a pair's two energy rows hold one float64 row
    for (int fpw : {2, frame_batch_fpw(max_period)})               // the odd frames' rows share the fixed buffers
        if ((size_t)(fpw / 2) * rs * 8 + 64 > kFrameLdsFixed) return false;
    return true;
}
void launch_frame(const PassParams &p, const DevTables &t, hipStream_t s) {
    if (p.n_sel == 0 || !(p.stages & 0xFu)) return;
    const int stride = frame_en_stride(p.max_period);
    // the running-energy prologue is one serial walk per workgroup, so small launches (streaming pushes) take a pair each
    const int fpw = p.n_sel >= 4096 ? frame_batch_fpw(p.max_period) : 2;
    size_t lds = kFrameLdsFixed + (size_t)fpw * stride * 4 + (size_t)fpw * 8;     // + the frames' workspace rows (epilogue)
    // AEGIS_FRAME_LDS_MIN=<bytes> (experiment knob): ask for at least that much LDS, e.g. 100000 keeps one workgroup per CU
    static const size_t lds_min = [] { const char *e = std::getenv("AEGIS_FRAME_LDS_MIN"); return e ? (size_t)std::atol(e) : (size_t)0; }();
    lds = std::max(lds, std::min<size_t>(lds_min, 160 * 1024));
    hipLaunchKernelGGL(frame_yin_kernel, dim3((unsigned)((p.n_sel + fpw - 1) / fpw)), dim3(256), lds, s, p, t, fpw, stride);
}
void launch_pyin_obs(const PassParams &p, const DevTables &t, hipStream_t s) {
    if (p.n_sel == 0) return;
    const int KM = p.n_lags / 2 + 2, YN = (std::max(p.n_lags, p.n_bins) + 1) & ~1, DN = (p.max_period + 1 + 32 + 1) & ~1;
    const int UN = (std::max(DN, 2 * KM + (4 * KM + 7) / 8) + 1) & ~1, TN = (2 * (KM + 1) + 101 + 1) & ~1;
    // eight waves share one copy of the tables (two such workgroups per CU); small launches (streaming pushes) get a wave
    // per frame and one frame per wave
    // (a dense pass -- PassParams::dense -- takes four-wave workgroups: one wave per SIMD fits beside a Viterbi workgroup)
    int waves = (int)std::min<int64_t>(p.dense ? 4 : 8, p.n_sel);
    while (waves > 1 && (size_t)(TN + waves * (YN + UN)) * 8 + 1024 > 80 * 1024) --waves;
    const int fpw = p.n_sel >= 4096 ? (p.dense ? 8 : 4) : 1;     // (dense: half the waves per workgroup, twice the frames per wave)
    const size_t lds = (size_t)(TN + waves * (YN + UN)) * 8;
    const int64_t per_wg = (int64_t)waves * fpw;
    hipLaunchKernelGGL(pyin_obs_kernel, dim3((unsigned)((p.n_sel + per_wg - 1) / per_wg)), dim3(64 * waves), lds, s, p, t, fpw);
}
// librosa.util.valid_audio: every sample finite.  16 bytes per thread and iteration, one atomic per wave that finds one.
__global__ __launch_bounds__(256) void finite_check_kernel(const float *__restrict__ pcm, int64_t n, unsigned long long *first_bad) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    unsigned long long bad = ~0ull;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n && (__float_as_uint(pcm[i + k]) & 0x7f800000u) == 0x7f800000u) bad = min(bad, (unsigned long long)(i + k));
    }
    if (bad != ~0ull) atomicMin(first_bad, bad);
}
void launch_finite_check(const float *pcm, int64_t n, unsigned long long *first_bad, hipStream_t s) {
    if (n <= 0) return;
    const unsigned g = (unsigned)std::min<int64_t>(4096, (n + 1023) / 1024);
    hipLaunchKernelGGL(finite_check_kernel, dim3(g), dim3(256), 0, s, pcm, n, first_bad);
}
void launch_decode(const PassParams &p, const DevTables &t, hipStream_t s) {
    if (p.n_frames == 0 || !(p.stages & 0x4u)) return;
    const unsigned g256 = (unsigned)((p.n_frames + 255) / 256);
    hipLaunchKernelGGL(decode_kernel, dim3(g256), dim3(256), 0, s, p, t);
}
void launch_finalize_mel(const PassParams &p, const DevTables &, hipStream_t s) {
    if (p.n_frames == 0) return;
    const unsigned g256 = (unsigned)((p.n_frames + 255) / 256);
    if (p.stages & 0x3u) {
        hipLaunchKernelGGL(db_rake_kernel, dim3((unsigned)((p.n_frames + 63) / 64)), dim3(256), 0, s, p);
        if (p.stages & 0x2u) hipLaunchKernelGGL(rake_runs_kernel, dim3(g256), dim3(256), 0, s, p);
    }
}

}  // namespace aegis
