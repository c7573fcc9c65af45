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
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) void viterbi_band_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
    constexpr int W = 2 * H + 1;
    constexpr int C = kViterbiChunk;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int B = p.n_bins, S = 2 * B, NC = p.n_cls;
    const int BP = (B + 63) & ~63;
    const int PADB = (B + 2 * H + 64 + 7) & ~7;            // slack: inactive lanes read past B+2H
    // value columns, one block per step parity: [2 v][PADB] interior-indexed (index b + H, -inf padded) then [2 v][2H]
    // for the edge bins (b < H, b > B-1-H), which the interior chains must not see at their band positions
    const int PB = 2 * PADB + 4 * H;
    double *val = reinterpret_cast<double *>(smem_raw);    // [2 parity][PB]
    double *rv = val + 2 * PB;                             // [2][16]  wave maxima (+ 32 doubles spare)
    unsigned long long *omask = reinterpret_cast<unsigned long long *>(rv + 64);   // [2][16] observed-state ballots of the voiced waves
    int *ri = reinterpret_cast<int *>(omask + 32);         // [2][16]
    uint16_t *ring = reinterpret_cast<uint16_t *>(ri + 32);   // [2][S] chunk-origin maps
    double *ltl = reinterpret_cast<double *>(
        smem_raw + (((size_t)(4 * PADB + 8 * H + 96) * 8 + 32 * 4 + (size_t)2 * S * 2 + 15) / 16) * 16);   // [2][NP]
    constexpr int NP = pk_size<H>();
    // Reach gates of the edge rows: gate[i] = -inf below KG, +inf from KG on.  An edge source e reaches only some of a
    // wave's targets; lane and source index the array so that min(table entry, gate) leaves an entry in reach as it is
    // and turns one out of reach into -inf -- one v_min_f64 instead of a compare and selects on the table offset.
    constexpr int KG = H + 63, NGATE = 3 * H + 64;
    double *gate = ltl + (band_table_packed(H) ? 2 * NP : 4 * NC * W);
    // The packed layout costs a few scalar multiplies per list entry (row starts are quadratic in the class); the
    // 44.1 kHz band (H = 25) keeps the full table, whose 83 KB fit; the 22.05 kHz band (H = 50) needs the packing.
    constexpr bool PK = LT_LDS && band_table_packed(H);

    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    const int vp = __builtin_amdgcn_readfirstlane(tid >= BP ? 1 : 0);
    const int b2 = tid - vp * BP;
    const bool act = b2 < B;
    const int b2c = act ? b2 : 0;
    const int j = vp * B + b2c;
    const int wlo = __builtin_amdgcn_readfirstlane(b2 - lane), whi = wlo + 63;
    const bool wave_low = wlo < 2 * H;            // some target of this wave sees low-edge sources
    const bool wave_high = whi >= B - 2 * H;      // ... high-edge sources (never both: B >= 4H+128)
    const bool is_low = b2c < H, is_high = b2c > B - 1 - H;
    const int eidx = is_low ? b2c : b2c - B + 2 * H;

    // Issue priority by expected work: the step ends when the slowest wave reaches the barrier, and the SIMD arbiter
    // otherwise serves the oldest wave first.  Edge waves (25 or 50 extra candidates per source voicing) first, then
    // the low-bin waves (most observed sources: sub-harmonic troughs crowd the low bins), then the rest
    // (a separate, lower level for the voiced-target interior waves starved them: +2 %); measured 76.3 -> 72.5 ms
    // when introduced.  Priorities by list length per step cost more than they gain.
    if (wave_low || wave_high) __builtin_amdgcn_s_setprio(3);
    else if (wlo < 128) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
    for (int i = tid; i < 2 * PB; i += nthr) val[i] = -INFINITY;
    // LDS copy of the band table (edge rows + list lookups).  Slot [class 0][dd = 0] of every (v,v')
    // block is never a real transition (it would be a target bin of -H): it holds the -inf sentinel
    // that out-of-reach (lane, source) pairs are redirected to.
    // Slot 0 of each packed block holds the -inf sentinel that out-of-reach (lane, source) pairs are redirected to.
    const double *lt_e0, *lt_e1;   // blocks (v = 0 -> v' = vp) and (v = 1 -> v' = vp)
    if (PK) {
        for (int i = tid; i < 2 * NP; i += nthr) ltl[i] = (i % NP == 0) ? -INFINITY : tb.lt_pack[i];
        lt_e0 = ltl + (vp ? NP : 0);            // "stay" block first, "switch" block second
        lt_e1 = ltl + (vp ? 0 : NP);
    } else if (LT_LDS) {
        // full table; slot [class 0][dd = 0] of every block is never a real transition (target bin -H): sentinel
        for (int i = tid; i < 4 * NC * W; i += nthr) ltl[i] = (i % (NC * W) == 0) ? -INFINITY : tb.lt_band[i];
        lt_e0 = ltl + (size_t)vp * NC * W;
        lt_e1 = lt_e0 + (size_t)2 * NC * W;
    } else {
        lt_e0 = tb.lt_band + (size_t)vp * NC * W;
        lt_e1 = lt_e0 + (size_t)2 * NC * W;
    }
    if (AEGIS_VIT_GATE && LT_LDS && H == 25) for (int i = tid; i < NGATE; i += nthr) gate[i] = i < KG ? -INFINITY : INFINITY;
    const double *lti0 = blt.v[0 * 2 + vp];   // interior row, source v = 0, target v' = vp
    const double *lti1 = blt.v[1 * 2 + vp];   // source v = 1
    // The interior row of the unvoiced-source block, RESIDENT in scalar registers for the whole kernel.  The row is
    // symmetric bit for bit (lt[d] == lt[2H - d]: the same triangle entry over the same row sum; launch_viterbi checks),
    // so H + 1 values = 2H + 2 SGPRs hold it.  Read from the kernarg segment inside the step loop, as the compiler does
    // for the full 4 x (2H + 1) table, every few candidates wait on an s_load -- and s_waitcnt lgkmcnt(0) is the only
    // way to wait for a scalar load, so each of those waits also drains the LDS reads in flight: the chain ran at the
    // scalar cache's latency (~25 waits of ~100 cycles per step), not at the vector issue rate.  The values are taken from
    // the LDS copy of the table through v_readfirstlane (below, once the copy is complete): nothing the compiler could
    // re-materialise from the kernarg segment inside the loop.
    constexpr int kSentinel = 0;

    const int c = p.order[blockIdx.x];
    const int64_t f0 = p.frame_off[c];
    const int T = (int)(p.frame_off[c + 1] - f0);
    const int os = p.obs_stride;
    const double *__restrict__ lobs = p.logobs + f0 * os;
    const double *__restrict__ lunv = p.logunv + f0;
    const int32_t *__restrict__ oseg = p.obs_seg + f0;
    uint16_t *__restrict__ ptr = p.ptr + f0 * S;
    const int64_t ch0 = p.chunk_off[c];
    uint16_t *__restrict__ cmap = p.cmap + ch0 * S;
    int32_t *__restrict__ bnd = p.bnd + ch0;
    int32_t *__restrict__ states = p.states + f0;
    const int nch = (T - 1 + C - 1) / C;
    __syncthreads();
    double ku[H + 1];         // see above: ku[i] = lt(1 -> vp)[i] = lt(1 -> vp)[2H - i], in SGPRs
    {
        const double *row = lt_e1 + (PK ? pk_int_start<H>() : H * W);
#pragma unroll
        for (int i = 0; i <= H; ++i) {
            const double t = LT_LDS ? row[i] : lti1[i];
            ku[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)), __builtin_amdgcn_readfirstlane(__double2loint(t)));
        }
    }
    auto ku_at = [&](int d) { return ku[d <= H ? d : W - 1 - d]; };     // lt of offset d (compile-time index)

    // Persistent launch: the steps are run chunk by chunk, each run after the chunk's observations are in memory.  Lane 0
    // of the workgroup polls the chunk's flag (agent-scope acquire: the producer ran on other CUs, possibly behind another
    // XCD's L2), sleeping between polls, and publishes the run's last step through LDS; the workgroup barrier behind it
    // orders every wave's loads after the acquire.  The poll is bounded (by the pass's size, see aegis_api.hip): a wait that long means the frame stage is not
    // running beside this kernel, and the kernel must end rather than hold its CU.  Everything the wait needs is parked
    // in LDS (the spare doubles behind the wave maxima), so the step loop carries one flag for it, no pointers.
    const bool chunked = __builtin_amdgcn_readfirstlane(p.chunk_flag != nullptr ? 1 : 0) != 0;
    volatile int *wslot = reinterpret_cast<volatile int *>(rv + 32);                     // [0] end of run (-1: give up), [1] chunk
    volatile unsigned long long *wpar = reinterpret_cast<volatile unsigned long long *>(rv + 34);
    if (chunked && tid == 0) {
        wpar[0] = reinterpret_cast<unsigned long long>(p.chunk_flag);
        wpar[1] = reinterpret_cast<unsigned long long>(p.chunk_lo);
        wpar[2] = reinterpret_cast<unsigned long long>(p.abort_flag);
        wpar[3] = ((unsigned long long)(unsigned)p.n_chunks << 32) | p.chunk_gen;
        wpar[4] = p.wait_ticks ? p.wait_ticks : 150000000ull;
        wslot[1] = 0;
    }
    // returns the step the run that starts at step t0 ends before (at most t_stop), or -1
    auto next_run = [&](int t0, int t_stop) {
        if (tid == 0) {
            const uint32_t *flag = reinterpret_cast<const uint32_t *>(wpar[0]);
            const int64_t *lo = reinterpret_cast<const int64_t *>(wpar[1]);
            const int n = (int)(wpar[3] >> 32);
            const uint32_t gen = (uint32_t)wpar[3];
            int k = wslot[1];
            while (k + 1 < n && lo[k + 1] <= t0) ++k;                   // the chunk of step t0
            int end = t_stop;
            if (k + 1 < n && lo[k + 1] < (int64_t)end) end = (int)lo[k + 1];
            unsigned spins = 0;
            const unsigned long long w0 = wall_clock64();                // 100 MHz
#pragma nounroll
            while (__hip_atomic_load(flag + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen) {
                __builtin_amdgcn_s_sleep(32);
                if ((++spins & 255u) == 0 && wall_clock64() - w0 > wpar[4]) {            // the launch's bound (aegis_api.hip: 0.1 .. 1.5 s)
                    __hip_atomic_store(reinterpret_cast<uint32_t *>(wpar[2]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    end = -1;
                    break;
                }
            }
            wslot[1] = k;
            wslot[0] = end;
        }
        __syncthreads();
        const int end = wslot[0];
        __syncthreads();                  // the slot may be written again only after every wave has read it
        return end;
    };
    // (the launch itself is ordered behind chunk 0's observations: the first column reads frame 0)

    // this lane's slot inside a parity block
    const int sidx = (is_low || is_high) ? 2 * PADB + vp * 2 * H + eidx : vp * PADB + b2c + H;
    auto store_value = [&](int buf, double v) { val[buf * PB + sidx] = v; };

    const int64_t vt_begin = p.clip_t0 ? p.clip_t0[c] : geo_vt_begin(p), vt_end = p.clip_t0 ? p.clip_t1[c] : geo_vt_end(p);
    if (p.ctl && p.ctl->n_sel == 0) return;                     // graph replay of a push that completed no frame
    const int t_lo = (int)(vt_begin > 1 ? vt_begin : 1);
    const int t_hi = (int)(vt_end < T ? vt_end : T);
    if (vt_begin >= T && vt_begin != 0) return;                 // clip finished in an earlier launch
    double *__restrict__ vst = p.vstate + (int64_t)c * S;
    double myv = -INFINITY;
    bool observed = false;        // voiced state whose observation at the column's frame is not log(tiny)
    if (act) {
        const int tprev = vt_begin == 0 ? 0 : t_lo - 1;
        const double lp = vp ? lunv[tprev]
                             : ((oseg[tprev] & (0x40000000 | (1 << (wlo >> 6)))) ? lobs[(int64_t)tprev * os + b2c] : p.log_tiny);
        myv = vt_begin == 0 ? lp + (vp ? p.log_pinit_u : p.log_pinit_v) : vst[j];
        observed = !vp && lp != p.log_tiny;
        store_value(0, myv);
    }
    // Back-pointer chunk maps are composed on the fly: org[s] = state at the start of the current 16-step chunk
    // of the best path into s (one dependent LDS gather per step, double buffered), stored as the chunk map when
    // the chunk closes.  A launch that starts inside a chunk (streaming) rebuilds org from the HBM pointers.
    uint16_t *org = ring;      // [2][S]
    if (act) {
        const int tp = t_lo - 1, c0 = (tp / C) * C;
        int s0 = j;
        for (int tt = tp; tt > c0; --tt) s0 = ptr[(int64_t)tt * S + s0];
        org[j] = (uint16_t)s0;
    }
    double G = INFINITY;      // column max (all states); +inf until the first column is reduced, so that the first step of a
                              // launch sees Gp = +inf (no bound on the dead voiced sources: full chain)
    double Gp = INFINITY;     // the column max one step earlier (unknown at the first step of a launch)
    int kg;
    // End-of-step bookkeeping, one barrier: block arg-max (lowest index on ties; wave max -> first lane holding it ->
    // one LDS slot per wave -> every wave reduces the <= 16 slots) and one ballot mask per voiced wave marking the
    // observed voiced states.  wp = parity of the slots written (the step that follows reads them).
    // The column maximum itself is an LDS atomic: every wave's lane 0 takes the unsigned minimum of its wave maximum's bit
    // pattern (negative values: smaller pattern = larger value) into one of three rotating slots; behind the barrier a wave
    // reads the slot and finds the first wave whose maximum has that pattern -- no second reduction.  Slot (k + 1) mod 3
    // is reset during step k: its last readers passed the previous barrier.
    unsigned long long *gkey = reinterpret_cast<unsigned long long *>(rv + 40);      // [3], in the spare doubles
    if (tid < 3) gkey[tid] = ~0ull;
    __syncthreads();
    int ks = 0;
    auto end_of_step = [&](double v, bool obs, int wp) {
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 2)
        __syncthreads(); G = v; kg = 0; return;
#endif
        unsigned long long eq;
        const double wm = wave_max_neg(v, eq);
        const int kn = ks == 2 ? 0 : ks + 1;
        if (!vp) {                     // wave-uniform
            const unsigned long long om = __ballot(obs);
            if (lane == 0) omask[wp * 16 + wid] = om;
        }
        if (lane == 0) {
            rv[wp * 16 + wid] = wm;
            ri[wp * 16 + wid] = vp * B + wlo + (int)__ffsll((long long)eq) - 1;
            atomicMin(gkey + ks, (unsigned long long)__double_as_longlong(wm));
            if (wid == 0) gkey[kn] = ~0ull;
        }
        __syncthreads();
        const unsigned long long gk = gkey[ks];
        unsigned long long ak = ~0ull;
        int ai = 0x7fffffff;
        if (lane < nw) {
            ak = (unsigned long long)__double_as_longlong(rv[wp * 16 + lane]);
            ai = ri[wp * 16 + lane];
        }
        const unsigned long long eq2 = __ballot(ak == gk) & 0xffffull;           // waves are in state order
        Gp = G;
        G = __longlong_as_double((long long)gk);
        kg = __builtin_amdgcn_readlane(ai, (int)__ffsll((long long)eq2) - 1);
        ks = kn;
    };
    end_of_step(myv, observed, 0);
    if (p.live_states != nullptr && tid == 0 && vt_begin == 0) p.live_states[f0] = kg;

#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 64)
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
    constexpr bool GATED = AEGIS_VIT_GATE && LT_LDS && H == 25;   // at H = 50 the extra loads in flight spill registers
    // Per-lane bases of the edge-row walks (loop invariant): table entry of (source e, this lane) = base + a constant
    // of e, which the DS instruction carries as its immediate offset; lanes without a target act as lane bin 0 (low
    // edge) or B - 1 (high edge), whose reads stay inside the arrays.
    const int ble = act ? b2c : (wave_high ? B - 1 : 0);
    const int elo_t = PK ? ble : ble + H;                         // + pk_lo_start(e)      | + e (W - 1)
    const int ehi_t = ble - B + 2 * H + (PK ? 0 : (H + 1) * W);   // + pk_hi_start(e) - e  | + e (W - 1)
    const double *glo = gate + KG - min(ble - H, H + 63);         // + e:         in reach <=> e >= b' - H
    const double *ghi = gate + KG + max(ble - (B - 2 * H), -63) - (H - 1);   // + H - 1 - e: in reach <=> e <= b' - (B - 2H)
    const int lrlo = max(wlo - H, 0), lrhi = min(whi + H, B - 1);
    const int lw0 = lrlo >> 6, lw1 = lrhi >> 6;
    int lidx[3];              // slot (inside a parity block) of voiced bin 64 (lw0 + u) + lane
    unsigned long long wmask[3];   // bits of word lw0 + u inside [lrlo, lrhi]
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int bw = min(((lw0 + u) << 6) + lane, B - 1);
        lidx[u] = bw < H ? 2 * PADB + bw : (bw > B - 1 - H ? 2 * PADB + bw - B + 2 * H : bw + H);
        unsigned long long m = lw0 + u <= lw1 ? ~0ull : 0ull;
        if (lw0 + u == lw0) m &= ~0ull << (lrlo & 63);
        if (lw0 + u == lw1) m &= ~0ull >> (63 - (lrhi & 63));
        wmask[u] = m;
    }
    // this lane's observation of step t: one pointer and one stride for both voicings (no branch in the step)
    const double *__restrict__ lp_base = vp ? lunv : lobs + b2c;
    const int lp_stride = vp ? 1 : os;
    int ph = 0;               // step parity: columns, origin maps and list slots all alternate with it
    int n_list = 0;           // wave-steps that took the observed-sources-only path (wave-uniform)
    int n_skip = 0;           // voiced wave-steps skipped: every target dead at an easy frame
    // The steps of this launch, cut at the time-chunk boundaries when the launch spans several chunks (persistent
    // launch): the wait sits between two runs of the step loop, not inside it.
    int t = t_lo;
    while (t < t_hi) {
    int t_end = t_hi;
    if (chunked) {
        t_end = next_run(t, t_hi);
        if (t_end < 0) {                  // gave up: leave a decodable path (all unvoiced) behind, the call reports the error
            for (int i = tid; i < T; i += nthr) states[i] = B;
            return;
        }
    }
    // Observations are requested ONE STEP AHEAD (never past the run: the next time chunk's rows may not be written yet):
    // a step starts by deciding, from its own observations, whether this wave has anything to compute.
    // A voiced wave loads its 64-bin segment of the row only where obs_seg says it was stored (its bit, or the hard-frame
    // bit); otherwise the load is pointed at the frame's unvoiced observation -- a line the unvoiced waves fetch anyway --
    // and the step is skipped below, so the value is never looked at.  No branch: one select on the address.
    const int seg_bits = vp ? 0 : (0x40000000 | (1 << (wlo >> 6)));           // unvoiced waves: need = vp = 1 at every step
    int need_n = (__builtin_amdgcn_readfirstlane(oseg[t]) & seg_bits) | vp;
    double lp_n = (need_n ? lp_base + (int64_t)t * lp_stride : lunv + t)[0];
    int sg_n = oseg[min(t + 1, t_end - 1)];       // the segment word runs two steps ahead: it addresses the next step's load
    for (; t < t_end; ++t) {
        VIT_TICK(5)
        // every lane loads (lanes without a state read bin 0): the sum below then needs no wait at a control-flow join
        const double lp = lp_n;
        const int need = need_n;
        {
            const int tn = min(t + 1, t_end - 1);
            need_n = (__builtin_amdgcn_readfirstlane(sg_n) & seg_bits) | vp;
            lp_n = (need_n ? lp_base + (int64_t)tn * lp_stride : lunv + tn)[0];
            sg_n = oseg[min(t + 2, t_end - 1)];
        }
        const int cur = __builtin_amdgcn_readfirstlane(ph);
        // Dead voiced targets.  Call a voiced state dead at frame t when its observation is log(tiny), and frame t easy
        // when its unvoiced observation is not (voiced_prob < 1: the unvoiced observation is then > -43).  A dead state
        // (0, b) of an easy frame collects the same sources as its unvoiced twin (1, b) through rows that differ only by
        // the voicing factor (|log .99 - log .01| = 4.6), so value(1, b) >= value(0, b) + (logunv - log tiny) - 4.6
        // > value(0, b) + 660; as a source for the next column the twin loses at most another 4.6.  Such a state
        // therefore never wins or ties a maximisation, is never the column maximum and never lies on the decoded path:
        // its value may be replaced by -inf and its back-pointer left unwritten without changing any result.  A voiced
        // wave ALL of whose targets are dead at an easy frame (59 % of the voiced wave-steps on the bench clips) does
        // exactly that and goes straight to the end-of-step barrier.
        // (an easy frame's unvoiced observation is log((1 - voiced_prob) / B + tiny) >= log(2^-53 / B) = -42.8, a hard
        // frame's is log(tiny) = -708.4.)  The observation kernel has made the decision already: obs_seg carries one bit
        // per 64-bin segment with an observed bin and the hard-frame bit, and a segment without either was not even stored.
        const bool skip = !need;
        const double *colr = val + cur * PB;        // the column being read
        // observed bins within reach of this wave's targets, [wlo - H, whi + H], span <= 3 mask words: the masks and
        // this lane's share of those words' values (bin 64 w + lane, voiced) are fetched here, far ahead of the list
        // section that tests them
        double xw[3];
        unsigned long long mk[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {         // unconditional: a word past the reach repeats the last one and is masked out below
            mk[u] = omask[cur * 16 + min(lw0 + u, lw1)];
            xw[u] = colr[lidx[u]];
        }

        const double *vi0 = colr + b2c;
        const double *vi1 = colr + PADB + b2c;
        const double *ve0 = colr + 2 * PADB;
        const double *ve1 = ve0 + 2 * H;
        // edge sources: table offset of (source e, this lane) = lane base + e*(W-1); pairs out of
        // reach are redirected to the -inf sentinel (LDS copy) or predicated (global table).
        // The opaque copy of b' keeps these cheap per-step integer ops from being hoisted out of
        // the time loop into ~50 live registers.
        int bl = b2c;
        asm volatile("" : "+v"(bl));
        // full table: [class][dd], class e resp. H+1+e, dd = b' - b + H; packed: row start + (dd - first dd of the row)
        auto eoff_lo = [&](int e) { return PK ? pk_lo_start<H>(e) + bl : (bl + H) + e * (W - 1); };
        auto eoff_hi = [&](int e) { return PK ? pk_hi_start<H>(e) + (bl - B + 2 * H - e) : (bl - B + 2 * H) + (H + 1) * W + e * (W - 1); };
        const int reach_lo = act ? bl - H : 0x7fffffff;           // low source e in reach  <=> e >= reach_lo
        const int reach_hi = act ? bl - (B - 2 * H) : -1;         // high source e in reach <=> e <= reach_hi
        // candidate of edge source e (values ve, table block lt) for this lane's target, -inf when out of reach
        auto edge_lo = [&](const double *ve, const double *lt, int e) {
            if constexpr (GATED) {
                return ve[e] + min_f64_raw((lt + elo_t)[PK ? pk_lo_start<H>(e) : e * (W - 1)], glo[e]);
            } else {
                const bool ok = e >= reach_lo;
                const int off = (LT_LDS && !ok) ? kSentinel : eoff_lo(e);
                return (LT_LDS || ok) ? ve[e] + lt[off] : -INFINITY;
            }
        };
        auto edge_hi = [&](const double *ve, const double *lt, int e) {
            if constexpr (GATED) {
                return ve[H + e] + min_f64_raw((lt + ehi_t)[PK ? pk_hi_start<H>(e) - e : e * (W - 1)], ghi[H - 1 - e]);
            } else {
                const bool ok = e <= reach_hi;
                const int off = (LT_LDS && !ok) ? kSentinel : eoff_hi(e);
                return (LT_LDS || ok) ? ve[H + e] + lt[off] : -INFINITY;
            }
        };

        double best = -INFINITY;
        int bi = 0;
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 32)
        if (vp) {   // timing experiment: voiced waves skip all candidate work (values are wrong)
#else
        if (!skip) {
#endif
        // ---- unvoiced sources (v = 1): always the full band ---------------------------------------------------
        constexpr int HALF = (W + 1) / 2;
        double best1;
        int code1;
        constexpr int GSEL = H > 25 ? AEGIS_VIT_GROUP50 : AEGIS_VIT_GROUP;
        if constexpr (GSEL != 0) {
        // Arg-max in three phases.  Tracking the index beside the maximum costs four vector instructions per candidate
        // (add, compare, select, max) and the step is bound by vector issue.  Phase 1 takes only the maxima of NG
        // groups of GS consecutive candidates (add, max); phase 2 finds the first group holding the overall maximum;
        // phase 3 re-evaluates that one group -- the same additions on the same operands, so the same bits -- and keeps
        // the lowest index whose candidate equals the maximum: the result of the index-tracking chain, ties included.
        // The last group is anchored at W - GS and overlaps its predecessor instead of running past the band: a maximum
        // inside the overlap is found in the earlier group, which phase 2 prefers.
        {
            constexpr int GS = GSEL, NG = (W + GS - 1) / GS;    // NG maxima stay in registers
            double gm[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int d0 = g * GS < W - GS ? g * GS : W - GS;
                double m = vi1[d0] + ku_at(W - 1 - d0);
#pragma unroll
                for (int k = 1; k < GS; ++k) m = fmax(m, vi1[d0 + k] + ku_at(W - 1 - d0 - k));
                gm[g] = m;
            }
            best1 = gm[0];
            int grp = 0;
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                if (gm[g] > best1) grp = g;
                best1 = fmax(best1, gm[g]);
            }
            const int d0 = min(grp * GS, W - GS);
            const double *vg = vi1 + d0;
            // the interior row of the (1 -> vp) block in the LDS copy of the table, read downwards from dd = W - 1 - d0
            const double *tg = lt_e1 + (PK ? pk_int_start<H>() : H * W) + (W - 1) - d0;
            code1 = 0;
#pragma unroll
            for (int k = GS - 1; k >= 0; --k) {
                const double cand = vg[k] + tg[-k];
                if (cand == best1) code1 = d0 + k;
            }
        }
        } else {
            double best1b = -INFINITY;
            int code1b = 0;
            best1 = -INFINITY;
            code1 = 0;
#pragma unroll
            for (int d = 0; d < HALF; ++d) {
                const double cand = vi1[d] + lti1[W - 1 - d];
                if (cand > best1) code1 = d;
                best1 = fmax(best1, cand);
                if (HALF + d < W) {
                    const double candb = vi1[HALF + d] + lti1[W - 1 - HALF - d];
                    if (candb > best1b) code1b = HALF + d;
                    best1b = fmax(best1b, candb);
                }
            }
            if (best1b > best1) { best1 = best1b; code1 = code1b; }
        }
        int src1 = b2c + code1 - H;
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 512)
        if (blockIdx.x < 8) {      // interior-row sources only, as the chain above saw them
            const int r = src1;
            const int rn = __shfl_down(r, 1);
            const bool an = __shfl_down(act ? 1 : 0, 1) != 0;
            if (act && an && lane < 63 && rn < r) atomicAdd(&g_vit_span[192], 1ull);
            const int r0 = __shfl(r, lane & ~7), r7 = __shfl(r, lane | 7);
            const bool a7 = __shfl(act ? 1 : 0, lane | 7) != 0;
            if (a7 && (lane & 7) == 0) { atomicAdd(&g_vit_span[128 + min(63, max(0, r7 - r0))], 1ull); atomicAdd(&g_vit_span[194], 1ull); }
            const int nact = __popcll(__ballot(act));
            const int rl = __shfl(r, nact - 1), rf = __shfl(r, 0);
            if (lane == 0 && nact > 0) { atomicAdd(&g_vit_span[min(127, max(0, rl - rf))], 1ull); atomicAdd(&g_vit_span[193], 1ull); }
            if (act) atomicAdd(&g_vit_span[200 + min(63, abs(code1 - H))], 1ull);
        }
#endif
#if !(defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 1))
        if (wave_low) {     // low-edge sources precede the interior ones in state order: they win ties
            double eb1 = -INFINITY;
            int ec1 = 0;
#pragma unroll
            for (int e = 0; e < H; ++e) {
                const double cand1 = edge_lo(ve1, lt_e1, e);
                if (cand1 > eb1) ec1 = e;
                eb1 = fmax(eb1, cand1);
            }
            if (eb1 >= best1) { best1 = eb1; src1 = ec1; }
        }
        if (wave_high) {    // high-edge sources follow the interior ones: they lose ties
            double eb1 = -INFINITY;
            int ec1 = 0;
#pragma unroll
            for (int e = 0; e < H; ++e) {
                const double cand1 = edge_hi(ve1, lt_e1, e);
                if (cand1 > eb1) ec1 = e;
                eb1 = fmax(eb1, cand1);
            }
            if (eb1 > best1) { best1 = eb1; src1 = B - H + ec1; }
        }
#endif

        VIT_TICK(0)
        // ---- voiced sources (v = 0) ------------------------------------------------------------------
        // Exact pruning.  Voiced states whose previous-frame observation was log(tiny) carry that -708
        // in their value: value = log(tiny) + (best candidate out of the column before), and no candidate out of
        // that column exceeds Gp + lmax_all (Gp = its maximum; rounding is monotone), so MUb = log(tiny) +
        // (Gp + lmax_all) bounds every such state without a reduction over them.  No candidate built on one of
        // them can then exceed MUb + lmax (lmax = largest log-transition of the block).  If that bound is
        // strictly below the unvoiced chain's result in every lane of the wave, such sources can neither
        // win nor tie anywhere in the wave, and only the observed voiced states -- a handful per frame,
        // listed in ascending bin order by the previous step -- remain to be examined.
        int src = 0;
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 8)
        const bool list_only = true;
#elif defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 16)
        const bool list_only = false;
#else
        const double MUb = p.log_tiny + (Gp + blt.lmax_all);
        const bool list_only = LT_LDS && __all(!act || (MUb + blt.lmax[0 * 2 + vp] < best1));
#endif
        n_list += list_only ? 1 : 0;
        if (list_only) {
            // Second exact prune, per source: an observed voiced source of value vo offers no lane more than
            // vo + lmax; when that is strictly below the smallest unvoiced-chain result of the wave it can neither win
            // nor tie (a voiced candidate only beats best1 by being >= it), so the per-lane work of the entry -- table
            // lookup, add, compare, select -- is skipped on a scalar test.  94 % of the entries on the bench clips: most
            // observed bins are sub-harmonic troughs with tiny probabilities.  The test runs on all 64 bins of a mask word
            // at once (each lane holds one bin's value), so only the surviving entries are walked.
            const double wmin1 = wave_min_bound_neg(best1, act);
            const double lmax0 = blt.lmax[0 * 2 + vp];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int w = lw0 + u;
                unsigned long long m = mk[u] & wmask[u];
                m = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)m);
                if (m == 0) continue;
                m &= __ballot(!(xw[u] + lmax0 < wmin1));     // the per-source prune, all 64 bins of the word at once
                while (m) {                                  // ascending bins: strict '>' keeps the lowest index
                    const int bit = (int)__ffsll((long long)m) - 1;
                    const int bo = (w << 6) + bit;
                    m &= m - 1;
                    const bool lo_e = bo < H, hi_e = bo > B - 1 - H;
                    const double vo = read_lane_f64(xw[u], bit);
                    const int rowbase = !PK ? (lo_e ? bo : (hi_e ? bo - (B - 1 - 2 * H) : H)) * W
                                            : lo_e ? pk_lo_start<H>(bo) - (H - bo)
                                                   : (hi_e ? pk_hi_start<H>(bo - (B - H)) : pk_int_start<H>());   // scalar
                    const int dd = b2c - bo + H;
                    const int off = (act && (unsigned)dd < (unsigned)W) ? rowbase + dd : kSentinel;
                    const double cand = vo + lt_e0[off];
                    if (cand > best) src = bo;
                    best = fmax(best, cand);
                }
            }
        } else {
            double besta = -INFINITY, bestb = -INFINITY;
            int codea = 0, codeb = 0;
#pragma unroll
            for (int d = 0; d < HALF; ++d) {
                const double cand = vi0[d] + lti0[W - 1 - d];
                if (cand > besta) codea = d;
                besta = fmax(besta, cand);
                if (HALF + d < W) {
                    const double candb = vi0[HALF + d] + lti0[W - 1 - HALF - d];
                    if (candb > bestb) codeb = HALF + d;
                    bestb = fmax(bestb, candb);
                }
            }
            if (bestb > besta) { besta = bestb; codea = codeb; }
            best = besta;
            src = b2c + codea - H;
#if !(defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 1))
            if (wave_low) {
                double eb = -INFINITY;
                int ec = 0;
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    const double cand = edge_lo(ve0, lt_e0, e);
                    if (cand > eb) ec = e;
                    eb = fmax(eb, cand);
                }
                if (eb >= best) { best = eb; src = ec; }
            }
            if (wave_high) {
                double eb = -INFINITY;
                int ec = 0;
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    const double cand = edge_hi(ve0, lt_e0, e);
                    if (cand > eb) ec = e;
                    eb = fmax(eb, cand);
                }
                if (eb > best) { best = eb; src = B - H + ec; }
            }
#endif
        }
        VIT_TICK(1)
        bi = src;
        if (best1 > best) { best = best1; bi = B + src1; }
        // the one out-of-band candidate that can win: the previous column's arg-max
        {
            const int bg = kg >= B ? kg - B : kg;                       // scalar
            if ((unsigned)(b2c + (H - bg)) > (unsigned)(2 * H)) {       // |b' - bg| > H in one add and one compare
                const double cand = G + p.log_tiny;
                if (cand > best || (cand == best && kg < bi)) { best = cand; bi = kg; }
            }
        }
        }
        const double sum = lp + best;
        myv = (act && !skip) ? sum : -INFINITY;
        observed = act && !vp && !skip && lp != p.log_tiny;       // a skipped step did not load the row at all
        n_skip += skip ? 1 : 0;
        if (skip) {
            if (act) store_value(cur ^ 1, -INFINITY);
        } else if (act) {
            store_value(cur ^ 1, myv);
            ptr[(int64_t)t * S + j] = (uint16_t)bi;
            const uint16_t o = ((t - 1) % C == 0) ? (uint16_t)bi : org[cur * S + bi];
            org[(cur ^ 1) * S + j] = o;
            if (t % C == 0 || t == T - 1) cmap[(int64_t)((t - 1) / C) * S + j] = o;
        }
        VIT_TICK(2)
        end_of_step(myv, observed, cur ^ 1);
        VIT_TICK(3)
        if (p.live_states != nullptr && tid == 0) p.live_states[f0 + t] = kg;
        ph ^= 1;
        VIT_TICK(4)
    }
    }   // runs between chunk boundaries
#if defined(AEGIS_ABLATE) && (AEGIS_ABLATE & 64)
    if (blockIdx.x == 0 && lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd((unsigned long long *)&g_vit_dbg[wid * 8 + k], (unsigned long long)tacc[k]);
        atomicAdd((unsigned long long *)&g_vit_dbg[wid * 8 + 7], (unsigned long long)(t_hi - t_lo));
    }
#endif

    if (p.vstats != nullptr && lane == 0 && t_hi > t_lo) {
        atomicAdd(&p.vstats[0], (unsigned long long)(t_hi - t_lo));
        atomicAdd(&p.vstats[1], (unsigned long long)n_list);
        atomicAdd(&p.vstats[2], (unsigned long long)n_skip);
    }
    if (t_hi < T) {                       // more launches follow: hand the column over
        if (act) vst[j] = myv;
        return;
    }
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
