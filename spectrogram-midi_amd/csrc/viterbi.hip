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
// Two builds of the same body.  viterbi_band_kernel: what a latency-bound pass runs (up to 128 registers per lane).
// viterbi_band_dense_kernel: at most 96 registers per lane (a few values of the step loop live in scratch: the step is 2 %
// slower), which leaves a 128-register wave slot free on every SIMD beside a Viterbi workgroup's 3.5 waves -- in a pass
// with a workgroup on every CU (>= 256 clips) the observation kernel's four-wave workgroups then run ON the Viterbi's CUs,
// in the issue slots its dependent chains leave empty, instead of waiting for a CU of their own (512-clip folder: 337 ->
// 328 ms; 256 x 180 s: 160.5 -> 157.3).
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) void viterbi_band_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
#include "viterbi_band.inc"
}
template <int H, bool LT_LDS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5))) void viterbi_band_dense_kernel(PassParams p, DevTables tb, BandLT<H> blt) {
#include "viterbi_band.inc"
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
