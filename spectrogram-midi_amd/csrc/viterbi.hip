// The pYIN Viterbi decode of the analyze path on gfx950 (MI355X / CDNA4, wave64): librosa.sequence.viterbi over the
// 2 x n_pitch_bins states of librosa.pyin (/root/reference/aegis_engine.py:63 calls pyin).
//
// A translation unit of its own because it is built with two code-generation switches the other kernels do not want
// (Makefile, VITFLAGS): the DS load merging of the back end and of the IR load/store vectoriser are off.  The
// candidate chains below read consecutive float64 values from LDS; merged into ds_read2_b64 each pair runs at half the
// LDS rate of two ds_read_b64 on gfx950 (128 vs 256 B/clk/CU), and the step is bound by LDS reads and vector issue
// together (64 clips x 180 s: 69.2 -> 61.4 ms with the merging off).
//
// Built with -ffp-contract=off: every add below rounds exactly where NumPy rounds.
#include <functional>
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace aegis {

// step range of this launch: by-value fields, or the device control block of a graph replay
__device__ __forceinline__ int64_t geo_vt_begin(const PassParams &p) { return p.ctl ? p.ctl->vt_begin : p.vt_begin; }
__device__ __forceinline__ int64_t geo_vt_end(const PassParams &p) { return p.ctl ? p.ctl->vt_end : p.vt_end; }

// ------------------------------------------------------------------------------------------
// Kernel 4: log-domain Viterbi, one workgroup per clip, one thread per HMM state.
//
// librosa's transition matrix is kron(loop(2, .99), local(B, width)) + tiny, dense.  A target
// state (v', b') therefore sees 2*width in-band predecessors with distinct log-probabilities
// and every other state at log(tiny).  Among those out-of-band predecessors only the global
// arg-max of the previous column can win (any in-band candidate built on that arg-max beats
// log(tiny)), so each step evaluates the band exactly and one extra candidate.  arg-max ties
// resolve to the lowest state index, as np.argmax does.
//
// Back-pointers go to HBM; every kViterbiChunk steps the chunk's pointer maps are composed in
// LDS into one map per chunk, so the final back-trace is a short serial walk over chunk maps
// followed by a parallel walk inside the chunks.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void viterbi_kernel(PassParams p, DevTables tb, int lt_in_lds) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int B = p.n_bins, S = 2 * B, H = p.half_width, W = p.width, NC = p.n_cls;
    const int SP = (S + 63) & ~63;
    constexpr int C = kViterbiChunk;
    double *val = reinterpret_cast<double *>(smem_raw);   // [2][SP]
    double *rv = val + 2 * SP;                            // [2][16]
    int *ri = reinterpret_cast<int *>(rv + 32);           // [2][16]
    uint16_t *ring = reinterpret_cast<uint16_t *>(ri + 32);   // [C][S]
    double *ltl = reinterpret_cast<double *>(smem_raw + ((2 * SP + 32) * 8 + 32 * 4 + C * S * 2 + 15) / 16 * 16);

    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    if (lt_in_lds)
        for (int i = tid; i < 4 * NC * W; i += nthr) ltl[i] = tb.lt_band[i];
    const double *lt = lt_in_lds ? ltl : tb.lt_band;

    const int c = p.order[blockIdx.x];
    const int64_t f0 = p.frame_off[c];
    const int T = (int)(p.frame_off[c + 1] - f0);
    const int os = p.obs_stride;
    const double *__restrict__ lobs = p.logobs + f0 * os;
    const double *__restrict__ lunv = p.logunv + f0;
    const int32_t *__restrict__ oseg = p.obs_seg + f0;
    // voiced observation of bin b at frame t: only the row segments obs_seg names are stored, the rest are log(tiny)
    auto voiced_obs = [&](int t, int b) {
        const int sg = oseg[t];
        return (sg & (0x40000000 | (1 << (b >> 6)))) ? lobs[(int64_t)t * os + b] : p.log_tiny;
    };
    uint16_t *__restrict__ ptr = p.ptr + f0 * S;
    const int64_t ch0 = p.chunk_off[c];
    uint16_t *__restrict__ cmap = p.cmap + ch0 * S;
    int32_t *__restrict__ bnd = p.bnd + ch0;
    int32_t *__restrict__ states = p.states + f0;
    const int nch = (T - 1 + C - 1) / C;

    const int j = tid;
    const bool act = j < S;
    const int v2 = (j >= B) ? 1 : 0;
    const int b2 = j - v2 * B;
    const int dlo = max(0, H - b2);
    const int dhi = min(W - 1, B - 1 - b2 + H);

    const int64_t vt_begin = p.clip_t0 ? p.clip_t0[c] : geo_vt_begin(p), vt_end = p.clip_t0 ? p.clip_t1[c] : geo_vt_end(p);
    if (p.ctl && p.ctl->n_sel == 0) return;                     // graph replay of a push that completed no frame
    const int t_lo = (int)(vt_begin > 1 ? vt_begin : 1);
    const int t_hi = (int)(vt_end < T ? vt_end : T);
    if (vt_begin >= T && vt_begin != 0) return;                 // clip finished in an earlier launch
    double *__restrict__ vst = p.vstate + (int64_t)c * S;
    double myv = -INFINITY;
    if (act) {
        if (vt_begin == 0) {
            const double lp = v2 ? lunv[0] : voiced_obs(0, b2);
            myv = lp + (v2 ? p.log_pinit_u : p.log_pinit_v);
        } else {
            myv = vst[j];
        }
        val[j] = myv;
    }
    int par = 0;
    double G;
    int kg;
    auto block_argmax = [&](double v, int ix) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ov = __shfl_down(v, o);
            const int oi = __shfl_down(ix, o);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if (lane == 0) { rv[par * 16 + wid] = v; ri[par * 16 + wid] = ix; }
        __syncthreads();
        G = rv[par * 16]; kg = ri[par * 16];
        for (int w = 1; w < nw; ++w) {
            const double ov = rv[par * 16 + w];
            const int oi = ri[par * 16 + w];
            if (ov > G || (ov == G && oi < kg)) { G = ov; kg = oi; }
        }
        par ^= 1;
    };
    block_argmax(myv, act ? j : 0x7fffffff);
    if (p.live_states != nullptr && tid == 0 && vt_begin == 0) p.live_states[f0] = kg;

    double *cur = val, *nxt = val + SP;
    for (int t = t_lo; t < t_hi; ++t) {
        double lp = 0.0;
        if (act) lp = v2 ? lunv[t] : voiced_obs(t, b2);
        double best = -INFINITY;
        int bi = 0;
        if (act) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const double *cv = cur + v * B;
                const double *ltv = lt + (size_t)(v * 2 + v2) * NC * W;
                for (int d = dlo; d <= dhi; ++d) {
                    const int b = b2 + d - H;
                    const int cl = b < H ? b : (b > B - 1 - H ? b - (B - 1 - 2 * H) : H);
                    const double cand = cv[b] + ltv[cl * W + (W - 1 - d)];
                    if (cand > best) { best = cand; bi = v * B + b; }
                }
            }
            const int bg = kg >= B ? kg - B : kg;
            const int dist = bg > b2 ? bg - b2 : b2 - bg;
            if (dist > H) {
                const double cand = G + p.log_tiny;
                if (cand > best || (cand == best && kg < bi)) { best = cand; bi = kg; }
            }
            myv = lp + best;
            nxt[j] = myv;
            ring[((t - 1) % C) * S + j] = (uint16_t)bi;
            ptr[(int64_t)t * S + j] = (uint16_t)bi;
        }
        block_argmax(myv, act ? j : 0x7fffffff);
        if (p.live_states != nullptr && tid == 0) p.live_states[f0 + t] = kg;
        double *tmp = cur; cur = nxt; nxt = tmp;
        if (t % C == 0 || t == T - 1) {
            const int cc = (t - 1) / C;
            if (act) {
                int s = j;
                for (int tt = t; tt > cc * C; --tt)      // steps of an earlier launch (streaming): pointers from HBM
                    s = tt >= t_lo ? ring[((tt - 1) % C) * S + s] : ptr[(int64_t)tt * S + s];
                cmap[(int64_t)cc * S + j] = (uint16_t)s;
            }
            __syncthreads();
        }
    }

    if (t_hi < T) {                       // more launches follow: hand the column over
        if (act) vst[j] = myv;
        return;
    }
    // back-trace: serial over chunk maps, then parallel inside the chunks
    if (tid == 0) {
        int s = kg;
        for (int cc = nch - 1; cc >= 0; --cc) { bnd[cc] = s; s = cmap[(int64_t)cc * S + s]; }
        states[0] = s;
    }
    __threadfence();
    __syncthreads();
    for (int cc = tid; cc < nch; cc += nthr) {
        const int te = min((cc + 1) * C, T - 1);
        int s = bnd[cc];
        states[te] = s;
        for (int tt = te; tt > cc * C + 1; --tt) { s = ptr[(int64_t)tt * S + s]; states[tt - 1] = s; }
    }
}

// ------------------------------------------------------------------------------------------
// Kernel 4 (band-specialised): the same recurrence with the half width H known at compile
// time.  Thread layout: two voicing halves of BP = roundup(B, 64) threads, so the target's
// voicing v' is wave-uniform.  Sources are split by the class of their transition row:
//   * interior rows (H <= b <= B-1-H) all share one 4 x (2H+1) table: the values live in a
//     -inf padded LDS array, the 2(2H+1) candidates are a fully unrolled loop of
//     ds_read_b64 (immediate offset) + v_add_f64 with a scalar table operand + compare/select;
//   * edge rows (b < H or b > B-1-H) are row-normalised differently: only waves within reach
//     of an edge walk them, one source at a time (source uniform, table lookup per lane).
// Candidate order does not matter for the result: ties are resolved explicitly to the lowest
// state index wherever the evaluation order is not the index order.
// ------------------------------------------------------------------------------------------
// wave64 max of a double via DPP row shifts / row broadcasts (result valid in lane 63)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fmax(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(hi2, lo2));
}
__device__ __forceinline__ double row16_prefix_max(double v) {   // lane 15 of each row = row max
    v = dpp_fmax<0x111, 0xf>(v);   // row_shr:1
    v = dpp_fmax<0x112, 0xf>(v);   // row_shr:2
    v = dpp_fmax<0x114, 0xf>(v);   // row_shr:4
    v = dpp_fmax<0x118, 0xf>(v);   // row_shr:8
    return v;
}
// Wave reductions of float64 values that are all NEGATIVE or -inf (every Viterbi value is: log(p_init + tiny) < 0, every
// log-transition < 0, observations <= log(1 + tiny)): for such values the larger double has the smaller bit pattern, so
// the maximum is an unsigned minimum of the high words followed by one of the low words among the lanes that hold
// that high word.  A 32-bit minimum takes its DPP operand directly (one v_min_u32_dpp per level: identity as the "old"
// value of lanes without a source); the float64 form needs two DPP moves, two copies and a canonicalising max per level.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umin(unsigned v) {
    return min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
    return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ unsigned row16_umin(unsigned v) {      // lane 15 of each row = row minimum
    v = dpp_umin<0x111, 0xf>(v);
    v = dpp_umin<0x112, 0xf>(v);
    v = dpp_umin<0x114, 0xf>(v);
    v = dpp_umin<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_umin(unsigned v) {       // uniform result
    v = row16_umin(v);
    v = dpp_umin<0x142, 0xa>(v);   // row_bcast:15
    v = dpp_umin<0x143, 0xc>(v);   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = dpp_umax<0x111, 0xf>(v);
    v = dpp_umax<0x112, 0xf>(v);
    v = dpp_umax<0x114, 0xf>(v);
    v = dpp_umax<0x118, 0xf>(v);
    v = dpp_umax<0x142, 0xa>(v);
    v = dpp_umax<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// largest value of the wave and the lanes that hold it
__device__ __forceinline__ double wave_max_neg(double v, unsigned long long &at) {
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const unsigned mh = wave_umin(hi);
    const bool top = hi == mh;
    const unsigned long long tb = __ballot(top);
    unsigned ml;
    if ((tb & (tb - 1)) == 0) {            // one lane holds that high word (the usual case): no second reduction
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, (int)__ffsll((long long)tb) - 1);
        at = tb;
    } else {
        ml = wave_umin(top ? lo : 0xffffffffu);
        at = __ballot(top && lo == ml);
    }
    return __hiloint2double((int)mh, (int)ml);
}
// a lower bound, within 2^-20 relative, of the smallest value among the lanes with take set (finite negative values; at
// least one lane takes part): the largest high word with all low bits set.  The list prune only needs a bound.
__device__ __forceinline__ double wave_min_bound_neg(double v, bool take) {
    const unsigned hi = take ? (unsigned)__double2hiint(v) : 0u;
    return __hiloint2double((int)wave_umax(hi), -1);
}
// v_min_f64 as one instruction: fmin() first canonicalises an operand that comes straight from memory (a second
// instruction); the operands here are never NaN.
__device__ __forceinline__ double min_f64_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double read_lane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Interior-row log-transition table, passed BY VALUE: kernel arguments live in the kernarg
// segment, which the compiler reads with scalar loads (s_load_dwordx16) -- the 2(2H+1) table
// operands of a step then sit in SGPRs and cost no vector memory traffic or VGPRs.
// Packed band table kept in LDS: the four (v,v') blocks of log(kron(loop, local) + tiny) are only two distinct ones
// (loop is symmetric: "stay" = v == v', "switch"), and an edge row only reaches the targets that exist, so a block
// is [sentinel][low-edge rows e = 0..H-1: dd = H-e..2H][interior row: dd = 0..2H][high-edge rows e = 0..H-1:
// dd = 0..2H-1-e] = 3H^2 + 3H + 2 entries (15.6 KB at H = 25, 61 KB at H = 50 -- the full [4][2H+1][2H+1] table of
// the 22.05 kHz band would be 326 KB).
template <int H> __host__ __device__ constexpr int pk_lo_start(int e) { return 1 + e * (H + 1) + e * (e - 1) / 2; }
template <int H> __host__ __device__ constexpr int pk_int_start() { return 1 + H * (H + 1) + H * (H - 1) / 2; }
template <int H> __host__ __device__ constexpr int pk_hi_start(int e) { return pk_int_start<H>() + (2 * H + 1) + 2 * H * e - e * (e - 1) / 2; }
template <int H> __host__ __device__ constexpr int pk_size() { return 3 * H * H + 3 * H + 2; }
// The packed layout costs a few scalar multiplies per list entry but 50 KB of LDS instead of 102 at H = 25, which lets a
// frame-stage workgroup share the CU when the batch fills the chip (256 clips: 240 -> 226 ms); measured neutral at 64 clips.
__host__ __device__ constexpr bool band_table_packed(int) { return true; }

template <int H>
struct BandLT {
    double v[4][2 * H + 1];   // [v*2+v'][dd]
    double lmax[4];           // [v*2+v'] largest log-transition of that block over ALL row classes
    double lmax_all;          // largest log-transition of the whole matrix
};

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 64)
__device__ long long g_vit_dbg[16 * 8];
#define VIT_TICK(k) { const long long now__ = clock64(); tacc[k] += now__ - tlast; tlast = now__; }
#else
#define VIT_TICK(k)
#endif

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 512)
// Span statistics of the unvoiced-source arg-max (tools/viterbi_spans.py): [0..127] histogram of r(last lane) - r(first
// lane) per wave-step, [128..191] of the same over aligned groups of 8 lanes, [192] pairs of neighbouring lanes whose
// arg-max source DEcreases with the target bin (none, if the concavity argument holds), [193] wave-steps counted,
// [194] 8-lane groups counted, [200..263] histogram of the arg-max offset |r(b') - b'|.
__device__ unsigned long long g_vit_span[272];
#endif
#ifndef AEGIS_VIT_GATE
#define AEGIS_VIT_GATE 1       // edge rows: reach gates + immediate table offsets (0: compare / select per candidate)
#endif
#ifndef AEGIS_VIT_GROUP50
#define AEGIS_VIT_GROUP50 0    // the same for the 22.05 kHz band (H = 50, 101 candidates): measured slower there in every
                               // group size (10: 61.0, 13: 60.7, 17: 61.5 ms against 58.4 for the index-tracking chain; the kernel sits
                               // at the 128-register limit of 14 waves per CU)
#endif
#ifndef AEGIS_VIT_GROUP
#define AEGIS_VIT_GROUP 7      // candidates per group of the unvoiced-source arg-max (0: index-tracking chain)
#endif
// Two builds of the same body.  viterbi_band_kernel: what a latency-bound pass runs (up to 128 registers per lane).
// viterbi_band_dense_kernel: at most 96 registers per lane (a few values of the step loop live in scratch: the step is 2 %
// slower), which leaves a 128-register wave slot free on every SIMD beside a Viterbi workgroup's 3.5 waves -- in a pass
// with a workgroup on every CU (>= 256 clips) the observation kernel's four-wave workgroups then run ON the Viterbi's CUs,
// in the issue slots its dependent chains leave empty, instead of waiting for a CU of their own (512-clip folder: 337 ->
// 328 ms; 256 x 180 s: 160.5 -> 157.3).
#define VIT_SPLIT 0
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) void viterbi_band_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
#include "viterbi_band.inc"
}
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5))) void viterbi_band_dense_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
#include "viterbi_band.inc"
}
#undef VIT_SPLIT
// Third build of the body: one workgroup per SEGMENT of a clip (time-split passes, see the section below).
#define VIT_SPLIT 1
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) void viterbi_band_split_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
#include "viterbi_band.inc"
}
#undef VIT_SPLIT

// ------------------------------------------------------------------------------------------
// Time-split Viterbi (round 4).  The recurrence is sequential in time, one workgroup per clip: a pass with fewer clips
// than compute units is bound by its longest clip (28 424 steps x 3.1 us = 88 ms for a 330 s clip, whatever else the
// chip has to do).  A clip is therefore cut into segments that run CONCURRENTLY:
//   1. speculative runs: segment k >= 1 starts `warm-up` frames before its boundary from a guessed column (the first-
//      frame formula).  A max-plus recurrence forgets its start as soon as every state's best path runs through one
//      state -- after any voiced note, within a few dozen frames -- and from there on its columns equal the true ones up
//      to ONE additive constant, so its decisions are the true ones;
//   2. lock-on runs: segment k starts again at its boundary, now from the END column of segment k - 1, and runs until its
//      column differs from the stored speculative column by a constant (spread of the differences <= sigma): usually at
//      the first check, 16 steps in.  Its pointers replace the speculative ones up to there;
//   2b. rounds of second speculation: a lock-on run that reaches the end of its segment without meeting the speculative run
//      IS the sequential run up to there.  It happens where the recurrence has nothing to forget with: in a stretch without
//      a voiced note the unvoiced states of the two edge bins (highest stay probability: the transition window is truncated
//      there) run as two rails that never exchange paths, and the offset the last note left between them stays for good -- a
//      run started from the first-frame formula inside the stretch cannot know it (or, simply, the segment was shorter than
//      the convergence took).  The run's last column is the best guess there is for what follows (in a stretch without
//      information the column's shape is stationary): the next segment runs from it directly, the later segments of the
//      clip speculate again from it and lock on again (viterbi_band.inc phases 3 / 4, kSplitRounds rounds; one more segment
//      time per round instead of the rest of the clip in sequence);
//   3. stitch: per-segment pointer maps (state at the segment's end -> state at its boundary), composed per clip from
//      the last segment's arg-max down, then the usual back-trace inside every segment, all in parallel;
//   4. verification and exact resolution: float64 sums are not translation invariant, so a decision of the hybrid run can
//      differ from the sequential run's where two candidates are closer than the accumulated rounding bound (and they are,
//      exactly or nearly: two steps of an unvoiced walk commute, (X + k[a]) + k[c] against (X + k[c]) + k[a]).  Every state
//      a path within that bound of the optimum can occupy is enumerated backwards from each such decision on the decoded
//      path (the "tube") until the tube collapses onto the path again, and recorded; tubes that do not collapse (the two
//      rails of an unvoiced stretch, tied for its whole length) are scanned in parallel and recorded as one entry.  One
//      wave per clip then walks the path forwards carrying the sequential run's EXACT value of the path's state -- two
//      float64 additions per frame outside the tubes, the sequential recurrence over the tube's states with the sequential
//      kernel's tie rule inside them (the previous column's arg-max at log tiny, the one out-of-band source the kernels
//      consider, included where it is within the bound) -- and re-traces the path through the exact pointers.  A clip that
//      cannot be resolved (a tube wider or deeper than a record, a column arg-max in doubt where the out-of-band candidate
//      is in play) is flagged and redone by the sequential kernel.
// Outputs are therefore those of the sequential kernel by construction, not by luck.
// ------------------------------------------------------------------------------------------
// clip_sel (PassParams): the stitch .. exact-walk kernels of a pass run once for the clips whose lock-on runs all met
// (beside the rounds of second speculation of the others) and once for the others, behind their rounds
__device__ __forceinline__ bool clip_selected(const PassParams &p, int c) {
    return p.clip_sel == 0 || ((p.clip_dirty[c] != 0) == (p.clip_sel == 2));
}
__global__ __launch_bounds__(1024) void viterbi_segmap_kernel(PassParams p) {
    constexpr int C = kViterbiChunk;
    const int sg = blockIdx.x, S = 2 * p.n_bins, j = threadIdx.x;
    if (j >= S || !clip_selected(p, p.seg_clip[sg])) return;
    const int T = p.seg_T[sg], st = p.seg_store[sg];
    const uint16_t *__restrict__ cmap = p.cmap + p.seg_ch0[sg] * S;
    int s = j;
    if (T - 1 > st)
        for (int cc = (T - 2) / C; cc >= st / C; --cc) s = cmap[(int64_t)cc * S + s];
    p.seg_map[(int64_t)sg * S + j] = (uint16_t)s;
}
// Between the rounds of a time-split pass: the first segment of every clip whose lock-on run never met its speculative run
// (-1) becomes the round's starting point and is marked -2; tube_count[1] keeps the number of rounds that had work.
__global__ __launch_bounds__(64) void viterbi_round_kernel(PassParams p, int round) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= p.n_clips) return;
    int k = -1;
    for (int sg = p.clip_seg0[c] + 1; sg < p.clip_seg0[c + 1]; ++sg)
        if (p.seg_lock[sg] == -1) { k = sg; break; }
    p.clip_first[c] = k;
    if (round == 0) p.clip_dirty[c] = k >= 0 ? 1 : 0;
    if (k >= 0) { p.seg_lock[k] = -2; atomicMax(p.tube_count + 1, (uint32_t)(round + 1)); }
}
// Hybrid split pass: the sequential kernel ran steps 1 .. hybrid_step of every clip under the frame stage and left its column
// in vstate.  For a clip that goes on behind that step, the column becomes what a speculative first segment would have
// left: the segment's end column (lock-on run of the second segment, exact walk), the stored column of the boundary frame
// with its maximum and arg-max (lowest state on ties, as end_of_step finds it) for the verification kernel.
__global__ __launch_bounds__(1024) void viterbi_seg0_fill_kernel(PassParams p) {
    __shared__ double wmax[16];
    __shared__ int wkg[16];
    const int c = blockIdx.x, S = 2 * p.n_bins, j = threadIdx.x, lane = j & 63, w = j >> 6;
    const int a = p.clip_seg0[c], b = p.clip_seg0[c + 1];
    if (b - a < 2) return;                       // the sequential kernel finished this clip, back-trace included
    const int64_t f = p.frame_off[c] + p.hybrid_step;
    const double v = j < S ? p.vstate[(int64_t)c * S + j] : -INFINITY;
    if (j < S) { p.seg_col[(int64_t)a * S + j] = v; p.colhist[f * (int64_t)S + j] = v; }
    double m = v;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    const unsigned long long at = __ballot(j < S && v == m);
    if (lane == 0) { wmax[w] = m; wkg[w] = at ? w * 64 + (int)__ffsll((long long)at) - 1 : 0x7fffffff; }
    __syncthreads();
    if (j == 0) {
        double G = -INFINITY;
        int kg = 0x7fffffff;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k)
            if (wmax[k] > G || (wmax[k] == G && wkg[k] < kg)) { G = wmax[k]; kg = wkg[k]; }
        p.colG[f] = G; p.colkg[f] = kg; p.seg_kg[a] = kg;
    }
}
__global__ __launch_bounds__(64) void viterbi_stitch_kernel(PassParams p) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= p.n_clips) return;
    const int S = 2 * p.n_bins, a = p.clip_seg0[c], b = p.clip_seg0[c + 1];
    if (p.split_hybrid && b - a < 2) return;     // (hybrid pass: decoded by the sequential kernel already)
    if (!clip_selected(p, c)) return;
    bool bad = false;
    for (int k = a + 1; k < b; ++k) bad |= p.seg_lock[k] == -1;      // (-2: a later round started over from its end column)
    int e = p.seg_kg[b - 1];
    for (int k = b - 1; k >= a; --k) { p.seg_end[k] = e; e = p.seg_map[(int64_t)k * S + e]; }
    p.states[p.frame_off[c]] = e;
    if (bad) atomicOr(&p.clip_flag[c], 1u);
}
__global__ __launch_bounds__(256) void viterbi_segtrace_kernel(PassParams p) {
    constexpr int C = kViterbiChunk;
    const int sg = blockIdx.x, S = 2 * p.n_bins, tid = threadIdx.x;
    const int T = p.seg_T[sg], st = p.seg_store[sg];
    if (T - 1 <= st || !clip_selected(p, p.seg_clip[sg])) return;
    const int64_t f0 = p.seg_f0[sg], ch0 = p.seg_ch0[sg];
    const uint16_t *__restrict__ cmap = p.cmap + ch0 * S;
    const uint16_t *__restrict__ ptr = p.ptr + f0 * S;
    int32_t *__restrict__ bnd = p.bnd + ch0;
    int32_t *__restrict__ states = p.states + f0;
    const int cfirst = st / C, clast = (T - 2) / C;
    if (tid == 0) {
        int s = p.seg_end[sg];
        for (int cc = clast; cc >= cfirst; --cc) { bnd[cc] = s; s = cmap[(int64_t)cc * S + s]; }
    }
    __threadfence();
    __syncthreads();
    for (int cc = cfirst + tid; cc <= clast; cc += blockDim.x) {
        const int te = min((cc + 1) * C, T - 1);
        int s = bnd[cc];
        states[te] = s;
        for (int tt = te; tt > cc * C + 1; --tt) { s = ptr[(int64_t)tt * S + s]; states[tt - 1] = s; }
    }
}

// Verification and exact resolution (step 4 above).
//
// viterbi_verify_kernel: one wave per frame of a split clip behind its first boundary.  A decision of the decoded path is
// AMBIGUOUS when more than one predecessor lies within the rounding bound of the best one; going backwards from the top of
// a stretch of ambiguous decisions the wave enumerates every state a near-optimal path can occupy (the tube) until the
// tube has collapsed onto the decoded path again, and RECORDS it: per frame the states of the tube.
// viterbi_exact_kernel: one wave per clip walks the decoded path forwards from the first boundary -- where the column is
// the sequential run's own, bit for bit -- and carries the sequential run's EXACT value of the path's state: outside the
// tubes a chain of two float64 additions per frame (the value of a state is the rounded sum along its best path, and the
// best path into a state of the decoded path is the decoded path wherever the decision is not ambiguous); inside a tube
// the same recurrence over the tube's states only, with the sequential kernel's tie rule (lowest state index), which
// yields the sequential run's own pointers there, and the path is re-traced through them.  What comes out is the
// sequential run's path, unvoiced bins included -- or the clip is flagged (a tube deeper or wider than the record holds,
// a tube that reaches the first boundary, an out-of-band candidate inside the bound, a segment that never locked on)
// and the sequential kernel decodes it again.
// g_verify_dbg: [0] -, [1] tubes opened, [2] tubes recorded, then the reasons a clip was flagged: [3] tube wider
// than kTubeCap, [5] tube closed on another state than the decoded one, [6] tube open at the exact run, [7] deeper than
// kTubeDepth, [8] largest depth recorded, [9] out-of-band candidate within the bound while the column's arg-max is in doubt, [10] last column's maximum not unique
// within the bound, [11] record buffer full, [12] tubes resolved by the exact walk, [13] of which changed the path,
// [14] tubes with a rail, [15] frames those rails span
//
// RAILS.  A tube does not always collapse: two unvoiced states whose best predecessor is the state itself, frame after frame,
// and whose values tie run side by side for as long as the stretch is unvoiced -- the two EDGE bins of a clip (or tail) that is
// unvoiced throughout are the standing case (the truncated transition window gives both the same, highest, stay probability),
// and a recording that ends in silence is one.  Walking such a tube level by level is a sequential chain as long as the
// Viterbi's own.  So when every member of a level is unvoiced and its only near-best predecessor is itself, the wave checks
// the next frames in PARALLEL (one lane per frame: is `self` alone within the bound for every member?), and the record holds
// the stretch as one entry: the level it hangs under and its length.  The exact walk runs the members' chains side by side
// through it (one lane each: the same two additions per frame as the decoded path's own chain).
constexpr int kTubeCap = 16, kTubeDepth = 48;
// ints of a record: clip, top frame, levels, slot, then per level n + states, then the rail: the level it hangs under (-1: none) and its length
constexpr int kTubeRec = 4 + (kTubeDepth + 1) * (1 + kTubeCap) + 2;
constexpr int kRailAt = kTubeRec - 2, kRailLen = kTubeRec - 1;
__device__ unsigned long long g_verify_dbg[16];

// twice the distance the hybrid run's values can be from the sequential run's at workspace frame fr of a clip that starts
// at frame fc and has nsp segments: one sigma per lock-on splice + two roundings per step and run
__device__ __forceinline__ double split_bound(const PassParams &p, int64_t fr, int64_t fc, int nsp) {
    const double g = fabs(p.colG[fr - 1]) + 1500.0;
    return 2.0 * (nsp * (1e-7 + 1e-13 * g) + 4.5e-16 * (double)(fr - fc) * g);
}
// log-transition of (state si -> state sj) inside the band (|bin difference| <= H)
__device__ __forceinline__ double split_lt(const PassParams &p, const DevTables &tb, int si, int sj) {
    const int B = p.n_bins, H = p.half_width, W = p.width;
    const int v = si >= B ? 1 : 0, v2 = sj >= B ? 1 : 0, bs = si - v * B, b2 = sj - v2 * B;
    const int cl = bs < H ? bs : (bs > B - 1 - H ? bs - (B - 1 - 2 * H) : H);
    return tb.lt_band[((size_t)(v * 2 + v2) * p.n_cls + cl) * W + (b2 - bs + H)];
}
// observation of state j at workspace frame fr, as the Viterbi kernels read it
__device__ __forceinline__ double split_obs(const PassParams &p, int j, int64_t fr) {
    const int B = p.n_bins;
    if (j >= B) return p.logunv[fr];
    return (p.obs_seg[fr] & (0x40000000 | (1 << (j >> 6)))) ? p.logobs[fr * (int64_t)p.obs_stride + j] : p.log_tiny;
}

__global__ __launch_bounds__(256) void viterbi_verify_kernel(PassParams p, DevTables tb) {
    constexpr int CAP = kTubeCap;
    __shared__ int sets[4][2][CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int B = p.n_bins, S = 2 * B, H = p.half_width, W = p.width, NC = p.n_cls;
    // the wave's frame: the idx-th of the frames behind the clips' first boundaries (vf_off: a hybrid pass has none to verify in
    // the clips and the steps its sequential kernel decoded -- half the pass -- and a wave per frame that returns at once is not free)
    const int64_t idx = (int64_t)blockIdx.x * 4 + w;
    if (idx >= p.vf_total) return;
    int lo = 0, hi = p.n_clips;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.vf_off[mid] <= idx) lo = mid; else hi = mid; }
    const int c = lo;
    const int a = p.clip_seg0[c], b = p.clip_seg0[c + 1];
    if (b - a < 2 || !clip_selected(p, c)) return;                // an unsplit clip is the sequential run itself
    const int64_t fc = p.frame_off[c];
    const int64_t fx = p.seg_f0[a + 1] + p.seg_store[a + 1];      // the first boundary: everything up to it is the exact run
    const int64_t f = fx + 1 + (idx - p.vf_off[c]);
    if (f >= p.frame_off[c + 1]) return;
    const int Tc = (int)(p.frame_off[c + 1] - fc), nsp = b - a;
    int why = 0;
    // predecessors of target j at frame fr within thr of the best one, appended to dst (deduplicated)
    auto near_preds = [&](int j, int64_t fr, double thr, int *dst, int &n) {
        const int v2 = j >= B ? 1 : 0, b2 = j - v2 * B;
        const double *__restrict__ col = p.colhist + (fr - 1) * (int64_t)S;
        const double G = p.colG[fr - 1];
        const int kg = p.colkg[fr - 1];
        double cand[4];                       // (voicing, round): up to 2 x 2 candidates per lane (W <= 128)
        double best = -INFINITY;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int d = lane + 64 * r, bs = b2 + d - H;
                double cv = -INFINITY;
                if (d < W && bs >= 0 && bs < B) {
                    const int cl = bs < H ? bs : (bs > B - 1 - H ? bs - (B - 1 - 2 * H) : H);
                    cv = col[v * B + bs] + tb.lt_band[((size_t)(v * 2 + v2) * NC + cl) * W + (W - 1 - d)];
                }
                cand[v * 2 + r] = cv;
                best = fmax(best, cv);
            }
        const int bg = kg >= B ? kg - B : kg;
        const bool oob = (bg > b2 ? bg - b2 : b2 - bg) > H;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) best = fmax(best, __shfl_xor(best, o));
        // The one out-of-band candidate (the previous column's arg-max at log tiny: viterbi_band.inc) joins the set when it is
        // within the bound of the best -- a hard frame whose observed bins lie far from the path: jump at log tiny, or stay and
        // take a log tiny observation.  Its source is the column's arg-max, which has to be beyond doubt: a second state within
        // the bound of the column maximum flags the clip.
        if (oob && G + p.log_tiny >= fmax(best, G + p.log_tiny) - thr) {
            best = fmax(best, G + p.log_tiny);
            int cnt = 0;
            for (int j0 = 0; j0 < S; j0 += 64) cnt += __popcll(__ballot(j0 + lane < S && col[j0 + lane] >= G - thr));
            if (cnt > 1) why = why ? why : 9;
            else {
                bool have = false;
                for (int q = 0; q < n; ++q) have |= dst[q] == kg;
                if (!have) { if (n < CAP) dst[n++] = kg; else why = why ? why : 3; }
            }
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned long long m = __ballot(cand[v * 2 + r] >= best - thr);
                while (m) {
                    const int l = (int)__ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int sidx = v * B + b2 + l + 64 * r - H;
                    bool have = false;
                    for (int q = 0; q < n; ++q) have |= dst[q] == sidx;
                    if (!have) { if (n < CAP) dst[n++] = sidx; else why = why ? why : 3; }
                }
            }
    };
    int *cur = sets[w][0], *nxt = sets[w][1];
    cur[0] = p.states[f];
    int n = 1;
    if (f == fc + Tc - 1) {                   // the end of the path: every state of the last column within the bound of its maximum
        const double *__restrict__ col = p.colhist + f * (int64_t)S;
        const double thr = split_bound(p, f + 1, fc, nsp), G = p.colG[f];
        n = 0;
        for (int j0 = 0; j0 < S; j0 += 64) {
            unsigned long long m = __ballot(j0 + lane < S && col[j0 + lane] >= G - thr);
            while (m) { const int l = (int)__ffsll((long long)m) - 1; m &= m - 1; if (n < CAP) cur[n++] = j0 + l; else why = 3; }
        }
    } else {
        // A tube opened by the decision above this one (frame f + 1) runs through this frame and below: only the TOP of a
        // stretch of ambiguous decisions opens one
        int m1 = 0, w0 = why;
        near_preds(p.states[f + 1], f + 1, split_bound(p, f + 1, fc, nsp), nxt, m1);
        if (why == w0 && m1 > 1) return;
        why = w0;                             // (a failure up there is that wave's to report)
        if (f + 1 == fc + Tc - 1) {           // ... and so does a last column whose maximum is not alone within the bound
            const double *__restrict__ col = p.colhist + (f + 1) * (int64_t)S;
            const double thr = split_bound(p, f + 2, fc, nsp), G = p.colG[f + 1];
            int cnt = 0;
            for (int j0 = 0; j0 < S; j0 += 64) cnt += __popcll(__ballot(j0 + lane < S && col[j0 + lane] >= G - thr));
            if (cnt > 1) return;
        }
    }
    // decisions into frames ftop, ftop - 1, ... (down to the first boundary) at which `self` is the only predecessor within the
    // bound for every one of the n unvoiced states mem[]: how many in a row.  One lane per frame.
    auto rail_scan = [&](const int *mem, int nm, int64_t ftop) -> int {
        int L = 0;
        const int64_t maxL = ftop - fx;
        while (L < maxL) {
            const int64_t fk = ftop - L - lane;
            bool ok = fk > fx;
            if (ok) {
                const double thr = split_bound(p, fk, fc, nsp);
                const double *__restrict__ col = p.colhist + (fk - 1) * (int64_t)S;
                const double G = p.colG[fk - 1];
                const int kg = p.colkg[fk - 1], bg = kg >= B ? kg - B : kg;
                for (int q = 0; q < nm && ok; ++q) {
                    const int b2 = mem[q] - B;
                    // (a lane reads two runs of <= 2H + 1 consecutive doubles of its frame's column; no branch in the loop, so
                    // the loads of several candidates are in flight together)
                    const int blo = b2 - H > 0 ? b2 - H : 0, bhi = b2 + H < B - 1 ? b2 + H : B - 1;
                    const int clj = b2 < H ? b2 : (b2 > B - 1 - H ? b2 - (B - 1 - 2 * H) : H);
                    const double self = col[B + b2] + tb.lt_band[((size_t)3 * NC + clj) * W + H];
                    double other = -INFINITY;
                    for (int v = 0; v < 2; ++v) {
                        const double *__restrict__ cv = col + v * B;
                        const double *__restrict__ ltv = tb.lt_band + (size_t)(v * 2 + 1) * NC * W;
#pragma unroll 8
                        for (int bs = blo; bs <= bhi; ++bs) {
                            const int cl = bs < H ? bs : (bs > B - 1 - H ? bs - (B - 1 - 2 * H) : H);
                            const double x = cv[bs] + ltv[cl * W + (H + b2 - bs)];
                            other = fmax(other, (v == 1 && bs == b2) ? -INFINITY : x);
                        }
                    }
                    ok = other < self - thr && !((bg > b2 ? bg - b2 : b2 - bg) > H && G + p.log_tiny >= self - thr);
                }
            }
            const unsigned long long bad = __ballot(!ok);
            if (bad) { L += (int)__ffsll((long long)bad) - 1; break; }
            L += 64;
        }
        return (int)(L < maxL ? L : maxL);
    };
    int depth = 0, rail_at = -1, rail_len = 0;
    int *rec = nullptr;
    int64_t fr = f;
    while (!why) {
        int m = 0;
        const double thr = split_bound(p, fr, fc, nsp);
        bool rail = n > 1 && rail_at < 0;       // every member unvoiced, and its only near-best predecessor is itself
        for (int q = 0; q < n && !why; ++q) {
            const int m0 = m;
            near_preds(cur[q], fr, thr, nxt, m);
            rail = rail && cur[q] >= B && m == m0 + 1 && nxt[m0] == cur[q];
        }
        if (why) break;
        if (depth == 0) {
            if (m == 1 && n == 1) { if (nxt[0] != p.states[fr - 1]) why = 5; break; }        // an unambiguous decision: the common case
            // a tube opens: take a record
            unsigned slot = 0;
            if (lane == 0) slot = atomicAdd(p.tube_count, 1u);
            slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
            if (slot >= (unsigned)p.tube_cap) { why = 11; break; }
            rec = p.tube_buf + (size_t)slot * kTubeRec;
            if (lane == 0) { rec[0] = c; rec[1] = (int)(f - fc); rec[3] = (int)slot; rec[4] = n; for (int q = 0; q < n; ++q) rec[5 + q] = cur[q]; rec[kRailAt] = -1; rec[kRailLen] = 0; }
        }
        ++depth; --fr;
        if (depth > kTubeDepth) { why = 7; break; }
        if (lane == 0) { int *r = rec + 4 + depth * (1 + CAP); r[0] = m; for (int q = 0; q < m; ++q) r[1 + q] = nxt[q]; }
        int *t2 = cur; cur = nxt; nxt = t2;
        n = m;
        if (n == 1) {                         // collapsed onto the decoded path again
            if (cur[0] != p.states[fr]) why = 5;
            break;
        }
        if (fr <= fx) break;                  // reached the first boundary with the tube still open: the exact column there holds every state's value
        if (rail) {                           // the level just recorded repeats itself: how far?  (cur == the same states, in the same order)
            const int L = rail_scan(cur, n, fr);
            if (L > 0) {
                rail_at = depth; rail_len = L; fr -= L;
                if (lane == 0) { rec[kRailAt] = rail_at; rec[kRailLen] = rail_len; }
                if (fr <= fx) break;
            }
        }
    }
    if (lane == 0 && (depth > 0 || why)) {   // (no per-frame counter: one address for a million waves is ~10 ns each)
        if (depth > 0) atomicAdd(&g_verify_dbg[1], 1ull);
        if (depth > 0 && !why) {
            rec[2] = depth;
            __threadfence();
            // the tube's bottom frame: where the exact walk meets it.  Tubes nest (a decision inside an open tube may be ambiguous
            // itself; its own tube is a subset of the outer one and ends at the same frame or above): the deepest one stays
            const int span = depth + rail_len;
            atomicMax(&p.tube_at[fr], ((span < 127 ? span : 127) << 24) | (rec[3] + 1));
            atomicAdd(&g_verify_dbg[2], 1ull);
            atomicMax(&g_verify_dbg[8], (unsigned long long)depth);
            if (rail_len > 0) { atomicAdd(&g_verify_dbg[14], 1ull); atomicAdd(&g_verify_dbg[15], (unsigned long long)rail_len); }
        }
        if (why) { atomicAdd(&g_verify_dbg[why], 1ull); atomicOr(&p.clip_flag[c], 2u); }
    }
}

__global__ __launch_bounds__(64) void viterbi_exact_kernel(PassParams p, DevTables tb) {
    constexpr int CAP = kTubeCap;
    __shared__ double xv[2][CAP];
    __shared__ int ptrx[kTubeDepth + 1][CAP];
    __shared__ int rec_lds[kTubeRec];
    const int c = blockIdx.x, lane = threadIdx.x;
    const int B = p.n_bins, S = 2 * B, H = p.half_width;
    const int a = p.clip_seg0[c], b = p.clip_seg0[c + 1];
    if (b - a < 2 || !clip_selected(p, c)) return;
    if (__hip_atomic_load(&p.clip_flag[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;     // the sequential kernel redoes it anyway
    const int64_t fc = p.frame_off[c], fx = p.seg_f0[a + 1] + p.seg_store[a + 1];
    const int64_t fend = fc + (p.frame_off[c + 1] - fc) - 1;
    int32_t *__restrict__ states = p.states;
    double X = p.seg_col[(int64_t)a * S + states[fx]];         // the sequential run's own value of the path's state at the first boundary
    int64_t t = fx;
    int n_res = 0, n_chg = 0;
    auto in_band = [&](int si, int sj) {
        const int bi = si >= B ? si - B : si, bj = sj >= B ? sj - B : sj;
        return (bi > bj ? bi - bj : bj - bi) <= H;
    };
    while (t < fend) {
        // up to 64 frames ahead: the chain's operands in parallel, the tube markers of the frames the chain stands on
        const int nblk = (int)min((int64_t)64, fend - t);
        double al = 0.0, ol = 0.0;
        int mark = 0;
        if (lane < nblk) {
            const int64_t fr = t + 1 + lane;
            const int si = states[fr - 1], sj = states[fr];
            al = in_band(si, sj) ? split_lt(p, tb, si, sj) : p.log_tiny;          // (out of band: the column arg-max's candidate)
            ol = split_obs(p, sj, fr);
            mark = p.tube_at[t + lane];
        }
        const unsigned long long mm = __ballot(mark != 0);
        const int k0 = mm ? (int)__ffsll((long long)mm) - 1 : nblk;
        for (int k = 0; k < k0; ++k) {
            const double ak = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(al), k), __builtin_amdgcn_readlane(__double2loint(al), k));
            const double ok = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ol), k), __builtin_amdgcn_readlane(__double2loint(ol), k));
            X = ok + (X + ak);
        }
        t += k0;
        if (!mm) continue;
        // ---- a tube whose bottom is frame t: the recurrence over its states, bottom to top ----------------------------
        const int slot = (__builtin_amdgcn_readlane(mark, k0) & 0xffffff) - 1;
        // the record goes to LDS in one batch of loads: walking its levels in global memory cost three dependent loads per level
        // (a clip with a few hundred tubes: 8 ms of the pass)
        {
            const int *__restrict__ grec = p.tube_buf + (size_t)slot * kTubeRec;
            __syncthreads();
            for (int i = lane; i < kTubeRec; i += 64) rec_lds[i] = grec[i];
            __syncthreads();
        }
        const int *rec = rec_lds;
        const int D = rec[2], RA = rec[kRailAt], RL = RA >= 0 ? rec[kRailLen] : 0;
        const int64_t ttop = fc + rec[1];
        // The common tube -- no rail, at most 15 levels of at most 4 states -- in registers: lane 4 d + q holds state q of level d.
        // Every lane fetches its state's observation, its transition entries from the (<= 4) states of the level below and the
        // path's old state in ONE batch of loads; the recurrence then runs level by level on cross-lane reads, and the new path
        // goes back in one batch of stores.  (Level by level through global memory it was three dependent loads per level:
        // ~20 us per tube, ~400 tubes in a polyphonic clip, one clip after the other.)
        bool fast = RA < 0 && D <= 15;
        {
            const int nl = (lane <= D && lane < 16) ? rec[4 + lane * (1 + CAP)] : 0;
            fast = fast && __ballot(nl > 4) == 0ull;
        }
        if (fast) {
            const int d = lane >> 2, q = lane & 3;
            const int *cs = rec + 4 + (d <= D ? d : D) * (1 + CAP);
            const bool have = d <= D && q < cs[0];
            const int j = have ? cs[1 + q] : 0;
            const int *ps = rec + 4 + (d < D ? d + 1 : D) * (1 + CAP);
            const int npd = (have && d < D) ? ps[0] : 0;
            int si[4];
            double lt[4];
            bool inb[4];
            const int kgp = (have && d < D) ? p.colkg[ttop - d - 1] : -1;     // the one source allowed out of band: the column arg-max below
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                si[i] = i < npd ? ps[1 + i] : 0;
                const bool band = i < npd && in_band(si[i], j);
                inb[i] = band || (i < npd && si[i] == kgp);
                lt[i] = band ? split_lt(p, tb, si[i], j) : p.log_tiny;
            }
            const double ob = (have && d < D) ? split_obs(p, j, ttop - d) : 0.0;
            const int old = lane <= D ? states[ttop - lane] : 0;          // (here the lane is the level)
            double x = 0.0;
            if (have && d == D) x = cs[0] == 1 ? X : p.seg_col[(int64_t)a * S + j];
            int bidx = 0;
            for (int dd = D - 1; dd >= 0; --dd) {
                double xi[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xi[i] = __shfl(x, 4 * (dd + 1) + i);
                if (have && d == dd) {
                    double best = -INFINITY;
                    int bs = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!inb[i]) continue;
                        const double cand = xi[i] + lt[i];
                        if (cand > best || (cand == best && si[i] < bs)) { best = cand; bs = si[i]; bidx = i; }
                    }
                    x = ob + best;
                }
            }
            // the top: one state, or (last frame of the clip) the exact arg-max of the candidates, lowest state first
            const int *ts = rec + 4;
            int top = 0;
            double xt = __shfl(x, 0);
            for (int k = 1; k < ts[0]; ++k) {
                const double xk = __shfl(x, k);
                if (xk > xt || (xk == xt && ts[1 + k] < ts[1 + top])) { top = k; xt = xk; }
            }
            X = xt;
            // the path through the tube by the exact pointers (uniform scalars), every level's new state to the lane of that level
            int idx = top, mynew = old;
            for (int dl = 0; dl <= D; ++dl) {
                const int st = rec[4 + dl * (1 + CAP) + 1 + idx];
                if (lane == dl) mynew = st;
                if (dl < D) idx = __shfl(bidx, 4 * dl + idx);
            }
            const bool diff = lane <= D && mynew != old;
            const unsigned long long dm = __ballot(diff);
            if (diff && lane != D) states[ttop - lane] = mynew;
            if ((dm >> D) & 1ull) {
                // (a change at the bottom can only happen at the first boundary, where the bottom set holds several states: the path
                // enters the exact run in another state -- follow the exact run's own pointers down until the old path is met)
                if (lane == D) {
                    int sx = mynew;
                    int64_t tt = ttop - D;
                    while (states[tt] != sx) { states[tt] = sx; if (tt == fc) break; sx = p.ptr[tt * (int64_t)S + sx]; --tt; }
                }
            }
            if (lane == 0) { n_res += 1; n_chg += dm ? 1 : 0; }
            __syncthreads();
            t = ttop;
            continue;
        }
        {
            const int *bs = rec + 4 + D * (1 + CAP);                   // the bottom set: one state, or several at the first boundary
            if (lane < bs[0]) xv[0][lane] = bs[0] == 1 ? X : p.seg_col[(int64_t)a * S + bs[1 + lane]];
        }
        __syncthreads();
        int pb = 0;
        // the rail under level RA: its states (all unvoiced) stay put for RL frames above frame f_lo, every lane its own chain
        auto rail_steps = [&](int64_t f_lo) {
            const int *rs = rec + 4 + RA * (1 + CAP);
            const bool mine = lane < rs[0];
            double x = mine ? xv[pb][lane] : 0.0;
            const double ls = mine ? split_lt(p, tb, rs[1 + lane], rs[1 + lane]) : 0.0;
            const int64_t f_hi = f_lo + RL;
            double nxt_ou = f_lo + 1 + lane <= f_hi ? p.logunv[f_lo + 1 + lane] : 0.0;
            for (int64_t f1 = f_lo + 1; f1 <= f_hi; f1 += 64) {
                const int nb = (int)min((int64_t)64, f_hi - f1 + 1);
                const double ou = nxt_ou;
                nxt_ou = f1 + 64 + lane <= f_hi ? p.logunv[f1 + 64 + lane] : 0.0;          // the next block's, under this block's chain
                if (nb == 64) {
#pragma unroll
                    for (int k = 0; k < 64; ++k) {
                        const double ok = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ou), k), __builtin_amdgcn_readlane(__double2loint(ou), k));
                        x = ok + (x + ls);
                    }
                } else {
                    for (int k = 0; k < nb; ++k) {
                        const double ok = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ou), k), __builtin_amdgcn_readlane(__double2loint(ou), k));
                        x = ok + (x + ls);
                    }
                }
            }
            __syncthreads();
            if (mine) xv[pb][lane] = x;
            __syncthreads();
        };
        if (RA == D) rail_steps(ttop - D - RL);
        for (int d = D - 1; d >= 0; --d) {
            const int64_t fr = ttop - d - (RA >= 0 && d >= RA ? RL : 0);      // (level RA: the lowest frame of its rail)
            const int *cs = rec + 4 + d * (1 + CAP), *ps = rec + 4 + (d + 1) * (1 + CAP);
            const int nc = cs[0], np = ps[0];
            if (lane < nc) {
                const int j = cs[1 + lane];
                double best = -INFINITY;
                int bs = 0x7fffffff, bidx = 0;
                const int kgp = p.colkg[fr - 1];                        // the one source allowed out of band: the column arg-max below
                for (int i = 0; i < np; ++i) {
                    const int si = ps[1 + i];
                    const bool band = in_band(si, j);
                    if (!band && si != kgp) continue;
                    const double cand = xv[pb][i] + (band ? split_lt(p, tb, si, j) : p.log_tiny);
                    if (cand > best || (cand == best && si < bs)) { best = cand; bs = si; bidx = i; }
                }
                xv[pb ^ 1][lane] = split_obs(p, j, fr) + best;
                ptrx[d][lane] = bidx;
            }
            __syncthreads();
            pb ^= 1;
            if (d == RA) rail_steps(fr);
        }
        // the top: one state, or (last frame of the clip) the exact arg-max of the candidates, lowest state first
        int top = 0;
        {
            const int *ts = rec + 4;
            for (int q = 1; q < ts[0]; ++q)
                if (xv[pb][q] > xv[pb][top] || (xv[pb][q] == xv[pb][top] && ts[1 + q] < ts[1 + top])) top = q;
        }
        X = xv[pb][top];
        // the path through the tube by the exact pointers
        if (lane == 0) {
            bool changed = states[ttop] != rec[5 + top];
            states[ttop] = rec[5 + top];
            int idx = D > 0 ? ptrx[0][top] : top;
            for (int d = 1; d <= D; ++d) {
                const int st = rec[4 + d * (1 + CAP) + 1 + idx];
                const int64_t fhi = ttop - d - (RA >= 0 && d > RA ? RL : 0);    // the level's frame (level RA: the top of its rail)
                changed |= states[fhi] != st;
                if (d == RA && states[fhi] != st)                               // the rail's frames under it (its lowest one is the next level's business at the bottom of the tube)
                    for (int64_t tt = fhi - 1; tt >= fhi - RL + (d == D ? 1 : 0); --tt) states[tt] = st;
                if (d == D) {
                    // (a change here can only happen at the first boundary, where the bottom set holds several states: the path
                    // enters the exact run in another state -- follow the exact run's own pointers down until the old path is met)
                    int sx = st;
                    int64_t tt = ttop - D - RL;
                    if (d == RA) states[fhi] = st;
                    while (states[tt] != sx) { states[tt] = sx; if (tt == fc) break; sx = p.ptr[tt * (int64_t)S + sx]; --tt; }
                } else states[fhi] = st;
                if (d < D) idx = ptrx[d][idx];
            }
            n_res += 1; n_chg += changed ? 1 : 0;
        }
        __syncthreads();
        t = ttop;
    }
    if (lane == 0 && n_res) { atomicAdd(&g_verify_dbg[12], (unsigned long long)n_res); atomicAdd(&g_verify_dbg[13], (unsigned long long)n_chg); }
}

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 512)
hipError_t viterbi_span_fetch(long long *dst) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vit_span), sizeof(long long) * 272);
    if (e == hipSuccess) { static long long z[272]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_vit_span), z, sizeof(z)); }
    return e;
}
#else
hipError_t viterbi_span_fetch(long long *dst) { for (int i = 0; i < 272; ++i) dst[i] = 0; return hipSuccess; }
#endif

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 64)
hipError_t viterbi_debug_fetch(long long *dst, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vit_dbg), sizeof(long long) * 128);
    if (e == hipSuccess && reset) { static long long z[128]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_vit_dbg), z, sizeof(z)); }
    return e;
}
#else
hipError_t viterbi_debug_fetch(long long *dst, bool) { for (int i = 0; i < 128; ++i) dst[i] = 0; return hipSuccess; }
#endif

template <int H>
static size_t viterbi_band_lds(const PassParams &p, bool lt_lds) {
    const int B = p.n_bins, S = 2 * B;
    const int PADB = (B + 2 * H + 64 + 7) & ~7;
    size_t b = (((size_t)(4 * PADB + 8 * H + 96) * 8 + 32 * 4 + (size_t)2 * S * 2 + 15) / 16) * 16;
    if (lt_lds) b += band_table_packed(H) ? (size_t)2 * pk_size<H>() * 8 : (size_t)4 * p.n_cls * (2 * H + 1) * 8;
    if (lt_lds) b += (size_t)(3 * H + 64) * 8;      // reach gates
    return b;
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static size_t viterbi_lds_bytes(const PassParams &p, bool with_lt) {
    const int S = 2 * p.n_bins, SP = (S + 63) & ~63;
    size_t b = ((size_t)(2 * SP + 32) * 8 + 32 * 4 + (size_t)kViterbiChunk * S * 2 + 15) / 16 * 16;
    if (with_lt) b += (size_t)4 * p.n_cls * p.width * 8;
    return b;
}

hipError_t viterbi_set_lds_limits() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_band_kernel<25, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_band_dense_kernel<25, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_band_split_kernel<25, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_band_split_kernel<50, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_band_kernel<50, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// AEGIS_VITERBI_EXCLUSIVE=<clips> (experiment knob, default off): launches of up to that many clips ask for the whole 160 KB
// of LDS, so that no frame-stage workgroup can be placed on the same CU.  Measured: no gain (64 clips: 83.6 vs 78.9 ms) --
// what slows the Viterbi is frame-stage code on the NEIGHBOURING CU (shared instruction cache), which the CU-partitioned
// streams of aegis_api.hip::split_streams avoid.
static size_t viterbi_launch_lds(size_t need, int n_clips) {
    static const int limit = [] { const char *e = std::getenv("AEGIS_VITERBI_EXCLUSIVE"); return e ? std::atoi(e) : 0; }();
    return n_clips <= limit ? std::max<size_t>(need, 160 * 1024) : need;
}

__global__ void chunk_signal_kernel(uint32_t *flag, uint32_t gen) {
    __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_chunk_signal(uint32_t *flag, uint32_t gen, hipStream_t s) {
    hipLaunchKernelGGL(chunk_signal_kernel, dim3(1), dim3(1), 0, s, flag, gen);
}

constexpr int kSplitRounds = 2;
static bool band_geometry(const PassParams &p);
bool viterbi_split_applies(const PassParams &p, const DevTables &t) {
    return viterbi_band_applies(p, t) && p.width <= 128;
}
template <int H>
static hipError_t launch_split_kernels(const PassParams &p, const DevTables &t, const double *host_lt_band, const int32_t *seg_order, int n_spec,
                                       const int32_t *lock_order, int n_lock, hipStream_t s, const std::function<void()> &fork = nullptr) {
    const int BP = (p.n_bins + 63) & ~63;
    BandLT<H> blt;
    for (int q = 0; q < 4; ++q) {
        std::memcpy(blt.v[q], host_lt_band + ((size_t)q * p.n_cls + H) * p.width, sizeof(blt.v[q]));
        blt.lmax[q] = *std::max_element(host_lt_band + (size_t)q * p.n_cls * p.width, host_lt_band + (size_t)(q + 1) * p.n_cls * p.width);
    }
    blt.lmax_all = *std::max_element(blt.lmax, blt.lmax + 4);
    const size_t lds = viterbi_band_lds<H>(p, true) + 64 * 8;     // + the lock-on comparison's per-wave extremes
    PassParams q = p;
    q.split_phase = 1; q.order = seg_order;
    if (n_spec > 0) hipLaunchKernelGGL((viterbi_band_split_kernel<H, true>), dim3((unsigned)n_spec), dim3(2 * BP), lds, s, q, t, blt);
    if (n_lock < 0) return hipGetLastError();       // (speculative runs only: launch_viterbi_split_spec)
    if (n_lock > 0) {
        q.split_phase = 2; q.order = lock_order;
        hipLaunchKernelGGL((viterbi_band_split_kernel<H, true>), dim3((unsigned)n_lock), dim3(2 * BP), lds, s, q, t, blt);
        // rounds of second speculation for the clips with a lock-on run that never met (viterbi_band.inc): workgroups of
        // clips without one return at once.  What is still unmet after the last round flags its clip (stitch kernel).
        for (int round = 0; round < kSplitRounds; ++round) {
            hipLaunchKernelGGL(viterbi_round_kernel, dim3((unsigned)((p.n_clips + 63) / 64)), dim3(64), 0, s, q, round);
            if (round == 0 && fork) fork();       // (clip_dirty is known: the clean clips' stitch .. exact walk start beside the rounds)
            q.split_phase = 3;
            hipLaunchKernelGGL((viterbi_band_split_kernel<H, true>), dim3((unsigned)n_lock), dim3(2 * BP), lds, s, q, t, blt);
            q.split_phase = 4;
            hipLaunchKernelGGL((viterbi_band_split_kernel<H, true>), dim3((unsigned)n_lock), dim3(2 * BP), lds, s, q, t, blt);
        }
    }
    return hipGetLastError();
}
int viterbi_tube_record_ints() { return kTubeRec; }
hipError_t viterbi_verify_fetch(long long *dst, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_verify_dbg), sizeof(long long) * 16);
    if (e == hipSuccess && reset) { static long long z[16]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_verify_dbg), z, sizeof(z)); }
    return e;
}
hipError_t launch_viterbi_split_spec(const PassParams &p, const DevTables &t, const double *host_lt_band, const int32_t *seg_order, int n_spec, hipStream_t s) {
    if (n_spec <= 0) return hipSuccess;
    return p.half_width == 25 ? launch_split_kernels<25>(p, t, host_lt_band, seg_order, n_spec, nullptr, -1, s)
                              : launch_split_kernels<50>(p, t, host_lt_band, seg_order, n_spec, nullptr, -1, s);
}
// stitch, back-trace, verification, exact walk of the clips `sel` selects (PassParams::clip_sel)
static void launch_split_finish(const PassParams &p, const DevTables &t, int sel, hipStream_t s) {
    PassParams q = p;
    q.clip_sel = sel;
    hipLaunchKernelGGL(viterbi_segmap_kernel, dim3((unsigned)p.n_seg), dim3(1024), 0, s, q);
    hipLaunchKernelGGL(viterbi_stitch_kernel, dim3((unsigned)((p.n_clips + 63) / 64)), dim3(64), 0, s, q);
    hipLaunchKernelGGL(viterbi_segtrace_kernel, dim3((unsigned)p.n_seg), dim3(256), 0, s, q);
    if (p.vf_total > 0) hipLaunchKernelGGL(viterbi_verify_kernel, dim3((unsigned)((p.vf_total + 3) / 4)), dim3(256), 0, s, q, t);
    hipLaunchKernelGGL(viterbi_exact_kernel, dim3((unsigned)p.n_clips), dim3(64), 0, s, q, t);
}
hipError_t launch_viterbi_split(const PassParams &p, const DevTables &t, const double *host_lt_band, const int32_t *seg_order, int n_spec,
                                const int32_t *lock_order, int n_lock, hipStream_t s, hipStream_t aux, hipEvent_t *ev) {
    if (p.n_seg == 0) return hipSuccess;
    if (p.split_hybrid) hipLaunchKernelGGL(viterbi_seg0_fill_kernel, dim3((unsigned)p.n_clips), dim3(1024), 0, s, p);
    // The rounds of second speculation keep ONE workgroup per affected clip busy for a segment's time each while the chip
    // waits; everything behind the lock-on runs is per clip, so the clips none of whose lock-on runs failed to meet (all
    // but a few) are stitched, verified and walked on `aux` meanwhile, the others behind their rounds.
    // (a handful of clips: the second set of launches costs more than a round hides -- one clip 6.6 -> 6.7 ms with it)
    const bool two = aux != nullptr && ev != nullptr && n_lock > 0 && aux != s && p.n_clips >= 8;
    bool forked = false;
    hipError_t fe = hipSuccess;
    std::function<void()> fork;
    if (two) fork = [&]() {
        fe = hipEventRecord(ev[0], s);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(aux, ev[0], 0);
        if (fe != hipSuccess) return;
        launch_split_finish(p, t, 1, aux);
        fe = hipEventRecord(ev[1], aux);
        forked = fe == hipSuccess;
    };
    hipError_t e = p.half_width == 25 ? launch_split_kernels<25>(p, t, host_lt_band, seg_order, n_spec, lock_order, n_lock, s, fork)
                                      : launch_split_kernels<50>(p, t, host_lt_band, seg_order, n_spec, lock_order, n_lock, s, fork);
    if (e != hipSuccess) return e;
    if (fe != hipSuccess) return fe;
    launch_split_finish(p, t, forked ? 2 : 0, s);
    if (forked) { fe = hipStreamWaitEvent(s, ev[1], 0); if (fe != hipSuccess) return fe; }
    return hipGetLastError();
}

static bool band_geometry(const PassParams &p) {
    const int BP = (p.n_bins + 63) & ~63;
    return p.n_cls == p.width && 2 * BP <= 1024 && p.n_bins >= 4 * p.half_width + 128;
}
bool viterbi_band_applies(const PassParams &p, const DevTables &t) {      // the two branches of launch_viterbi below
    if (!band_geometry(p)) return false;
    if (p.half_width == 25) return viterbi_band_lds<25>(p, true) <= 160 * 1024;
    if (p.half_width == 50) return t.lt_pack != nullptr && viterbi_band_lds<50>(p, true) <= 160 * 1024;
    return false;
}

hipError_t launch_viterbi(const PassParams &p, const DevTables &t, const double *host_lt_band, hipStream_t s) {
    if (p.n_clips == 0) return hipSuccess;
    const int S = 2 * p.n_bins;
    const int BP = (p.n_bins + 63) & ~63;
    if (band_geometry(p)) {
        // band-specialised kernels for the two hop/sr ratios the reference uses (44.1k and 22.05k at hop 512)
        if (p.half_width == 25 && viterbi_band_lds<25>(p, true) <= 160 * 1024) {
            BandLT<25> blt;
            for (int q = 0; q < 4; ++q) {
                std::memcpy(blt.v[q], host_lt_band + ((size_t)q * p.n_cls + 25) * p.width, sizeof(blt.v[q]));
                blt.lmax[q] = *std::max_element(host_lt_band + (size_t)q * p.n_cls * p.width,
                                                host_lt_band + (size_t)(q + 1) * p.n_cls * p.width);
            }
            blt.lmax_all = *std::max_element(blt.lmax, blt.lmax + 4);
            if (p.dense)
                hipLaunchKernelGGL((viterbi_band_dense_kernel<25, true>), dim3((unsigned)p.n_clips), dim3(2 * BP),
                                   viterbi_launch_lds(viterbi_band_lds<25>(p, true), p.n_clips), s, p, t, blt);
            else
                hipLaunchKernelGGL((viterbi_band_kernel<25, true>), dim3((unsigned)p.n_clips), dim3(2 * BP),
                                   viterbi_launch_lds(viterbi_band_lds<25>(p, true), p.n_clips), s, p, t, blt);
            return hipGetLastError();
        }
        if (p.half_width == 50 && t.lt_pack != nullptr && viterbi_band_lds<50>(p, true) <= 160 * 1024) {
            BandLT<50> blt;
            for (int q = 0; q < 4; ++q) {
                std::memcpy(blt.v[q], host_lt_band + ((size_t)q * p.n_cls + 50) * p.width, sizeof(blt.v[q]));
                blt.lmax[q] = *std::max_element(host_lt_band + (size_t)q * p.n_cls * p.width,
                                                host_lt_band + (size_t)(q + 1) * p.n_cls * p.width);
            }
            blt.lmax_all = *std::max_element(blt.lmax, blt.lmax + 4);
            hipLaunchKernelGGL((viterbi_band_kernel<50, true>), dim3((unsigned)p.n_clips), dim3(2 * BP),
                               viterbi_launch_lds(viterbi_band_lds<50>(p, true), p.n_clips), s, p, t, blt);
            return hipGetLastError();
        }
    }
    const int nthr = (S + 63) & ~63;
    const bool with_lt = viterbi_lds_bytes(p, true) <= 160 * 1024;
    const size_t lds = viterbi_lds_bytes(p, with_lt);
    if (nthr > 1024 || lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(viterbi_kernel, dim3((unsigned)p.n_clips), dim3(nthr), lds, s, p, t, with_lt ? 1 : 0);
    return hipGetLastError();
}
}  // namespace aegis
