// Device-side parameter blocks and launch wrappers for the analyze kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace aegis {

// Constant tables resident in HBM (built by tables.cpp, uploaded once per handle).
struct DevTables {
    const double *hann;        // [2048]
    const int32_t *mel_start;  // [n_mels]
    const int32_t *mel_len;    // [n_mels]
    const int32_t *mel_off;    // [n_mels]
    const float *mel_w;        // packed triangle weights
    const int32_t *mel_chunk_bin;   // [n_chunks] first bin of each <= 16-bin chunk of a triangle
    const float *mel_chunk_w;       // [n_chunks][16]
    const int32_t *mel_band_chunk;  // [n_mels + 1]
    int32_t mel_chunks;
    const double *thresholds;  // [101]
    const double *beta_probs;  // [100]
    const double *beta_cumsum; // [101]
    const double *beta_suffix; // [101]
    const double *boltz_fact;  // [n]
    const double *boltz_exp;   // [n]
    const double *lt_band;     // [4][n_cls][width]
    const double *lt_pack;     // [2 (stay, switch)][3H^2+3H+2] packed band table (viterbi.hip pk_*), or nullptr
    const double *freqs;       // [n_bins]
    const double2 *twiddle;    // [2048]
};

// Device-resident geometry of a streaming push (captured hipGraph): when PassParams::ctl is set the
// kernels take the selection / step range from here instead of the by-value fields, so the graph's
// kernel nodes never need new parameters.  Advanced by stream_advance_kernel at the head of the graph.
struct StreamCtl {
    int64_t n_samples;        // samples received
    int64_t frames_done;      // frames analysed
    int64_t t_begin, n_sel;   // frames of the current push
    int64_t vt_begin, vt_end; // Viterbi steps of the current push
    int64_t meta[9];          // sample_off[2] | frame_off[2] | sel_off[2] | chunk_off[2] | order
};

// Geometry shared by all kernels of one pass.
struct PassParams {
    // configuration
    int32_t sr, hop, n_mels;
    int32_t min_period, max_period, n_lags;
    int32_t n_bins, half_width, width, n_cls;
    double fmin;
    double log_tiny;
    double log_pinit_v, log_pinit_u;   // log(p_init + tiny) of a voiced / an unvoiced state
    uint32_t stages;
    // batch geometry (device arrays are per pass)
    const float *pcm;            // all clips of the batch
    const int64_t *sample_off;   // [n_clips] first sample of each of this pass's clips in pcm
    const int64_t *sample_len;   // [n_clips] samples per clip
    const int64_t *frame_off;    // [n_clips+1] frame offsets relative to the pass (workspace rows)
    const int64_t *out_off;      // [n_clips] first frame of each clip in the OUTPUT arrays (clips keep the caller's order
                                 //           there, whatever order the passes take them in)
    const int32_t *order;        // [n_clips] clip indices, longest first
    int32_t n_clips;
    int64_t n_frames;            // frames in this pass
    // frame selection of this launch (time-chunked pipeline): clip c contributes its frames
    // t_begin .. t_begin + (sel_off[c+1]-sel_off[c]) - 1; the whole pass is sel_off == frame_off, t_begin == 0
    const int64_t *sel_off;      // [n_clips+1]
    int64_t t_begin;
    // Ragged passes cut every clip into the SAME number of time chunks, each proportional to the clip's length (see
    // aegis_api.hip): clip c then contributes its frames clip_t0[c] .. clip_t0[c] + (sel_off[c+1]-sel_off[c]) - 1 and its
    // Viterbi steps [max(1, clip_t0[c]), min(T, clip_t1[c])).  NULL: t_begin / vt_begin / vt_end for every clip.
    const int64_t *clip_t0, *clip_t1;
    int32_t dense;               // throughput pass (a Viterbi workgroup on every CU): the 96-register Viterbi build and four-wave
                                 // observation workgroups, which share its CUs (viterbi.hip, launch_pyin_obs)
    int64_t n_sel;
    // Viterbi step range of this launch: t in [max(1, vt_begin), min(T, vt_end)); state carried in vstate
    int64_t vt_begin, vt_end;
    double *vstate;              // [n_clips][2*n_bins] column of values at the end of the previous launch
    int32_t *live_states;        // optional [F]: arg-max state of each column as it is produced (streaming preview)
    const StreamCtl *ctl;        // optional: device-resident t_begin / n_sel / vt_begin / vt_end (graph replay)
    unsigned long long *vstats;  // optional [2]: band Viterbi wave-steps, and how many took the observed-sources-only path
    // One Viterbi launch per pass beside the time-chunked frame stage (band kernels only): before the first step of time
    // chunk k the kernel waits until chunk_flag[k] == chunk_gen, which launch_chunk_signal() stores behind the chunk's
    // observation kernel.  A wait that exceeds its bound sets *abort_flag and ends the kernel.
    const uint32_t *chunk_flag;  // optional [n_chunks]
    const int64_t *chunk_lo;     // [n_chunks] first frame of each time chunk (chunk 0 starts at frame 0)
    int32_t n_chunks;
    uint32_t chunk_gen;
    uint32_t *abort_flag;
    uint64_t wait_ticks;         // bound of one chunk wait in 100 MHz wall-clock ticks (0: 1.5 s)
    // Time-split Viterbi (viterbi_band_split_kernel and the stitch / verification kernels, viterbi.hip): a clip is cut into
    // segments that run concurrently, one workgroup each.  Segment s covers the pass's workspace frames seg_f0[s] ..
    // seg_f0[s] + seg_T[s] - 1 (local steps 1 .. seg_T - 1); the first seg_store[s] steps are a warm-up from a guessed
    // column whose results are not kept; local frame seg_store[s] is the boundary it shares with segment seg_prev[s].
    const int64_t *seg_f0;       // [n_seg]
    const int32_t *seg_T;        // [n_seg]
    const int64_t *seg_ch0;      // [n_seg] back-pointer chunk (of the pass) that holds local steps 1..16
    const int32_t *seg_store;    // [n_seg]
    const int32_t *seg_prev;     // [n_seg] the segment before it in its clip, -1 for a clip's first
    const int32_t *seg_clip;     // [n_seg] its clip (index into the pass)
    const int32_t *clip_seg0;    // [n_clips + 1] first segment of each clip
    double *seg_col;             // [n_seg][2 n_bins] end column of the speculative run
    double *seg_col2;            // [n_seg][2 n_bins] end column of a lock-on run that never met the speculative one (the next round speculates from it)
    int32_t *clip_first;         // [n_clips] the clip's first segment whose lock-on run never met (this round; -1: none)
    int32_t *clip_dirty;         // [n_clips] != 0: a lock-on run of the clip never met (set by the first round): its stitch / verification / exact walk wait for the rounds
    const int64_t *vf_off;       // [n_clips + 1] prefix of the frames the verification kernel has to look at (a split clip's frames behind its first boundary)
    int64_t vf_total;            //               = vf_off[n_clips]: its grid covers these, not every frame of the pass
    int32_t clip_sel;            // stitch .. exact walk kernels: 0 every clip, 1 only the clips with clip_dirty == 0, 2 only the others
    int32_t *seg_kg;             // [n_seg] its arg-max
    int32_t *seg_lock;           // [n_seg] local step at which the lock-on run met the speculative run, -1: it did not, -2: it did not and a later round started over from its end column
    int32_t *seg_end;            // [n_seg] decoded state at the segment's last frame (stitch)
    uint16_t *seg_map;           // [n_seg][2 n_bins] state at the segment's last frame -> state at its boundary frame
    double *colhist;             // [F][2 n_bins] the column of every stored frame (lock-on comparison, verification)
    double *colG; int32_t *colkg;   // [F] its maximum and arg-max
    int32_t *tube_buf; int32_t tube_cap; uint32_t *tube_count;   // records of the verification kernel's tubes (viterbi.hip kTubeRec ints each)
    int32_t *tube_at;            // [F] (depth << 24) | (record + 1) of the deepest tube whose bottom (collapse) frame this is, 0: none
    uint32_t *clip_flag;         // [n_clips] != 0: the clip's decode could not be certified, the sequential kernel must redo it
    int32_t split_phase;         // 1: speculative runs, 2: lock-on runs
    int32_t n_seg;
    // Hybrid split pass (aegis_api.hip): the sequential kernel has run every clip's steps 1 .. hybrid_step under the frame
    // stage (a clip's first segment is that run; vstate holds its last column), only the segments behind it are speculative
    int32_t split_hybrid;
    int32_t hybrid_step;
    // workspace (strides in elements)
    double *dfn;   int32_t lag_stride;   // [F][lag_stride]   pyin's difference function d[tau], lags 0..max_period; with
                                         //                   cmnd_in_frame the entries tau >= min_period hold the CMND instead
    int32_t cmnd_in_frame;               // the frame kernel's epilogue forms the CMND (cumsum walk of all its frames at once);
                                         // 0: pyin_obs_kernel walks it frame by frame (stage tests, lag ranges the epilogue cannot hold)
    int32_t troughs;                     // with cmnd_in_frame: the frame kernel also finds the CMND's troughs and leaves each frame's
                                         // trough list (count, values, parabolic shifts, lags: kernels.hip trough_row_doubles) in its dfn row
                                         // instead of the CMND; pyin_obs_kernel starts at the threshold prior
    double *yin;   int32_t yin_stride;   // optional [F][yin_stride]: CMND for lags min..max, written only for the stage tests
    double *logobs; int32_t obs_stride;  // [F][obs_stride]   log(obs+tiny), voiced bins
    double *logunv;                      // [F]               log(unvoiced obs+tiny)
    int32_t *obs_seg;                    // [F]  which 64-bin segments of the logobs row are written: bit s = some bin of [64 s, 64 s + 64)
                                         //      is observed; bit 30 = hard frame (unvoiced observation log(tiny)), the whole row is written.
                                         //      Any other segment is all log(tiny), is never stored and must not be read.
    uint16_t *ptr;                       // [F][2*n_bins]     Viterbi back-pointers
    uint16_t *cmap;                      // [chunks][2*n_bins] composed chunk maps
    int64_t *chunk_off;                  // [n_clips+1]       first chunk of each clip
    int32_t *bnd;                        // [chunks]          state at each chunk end
    int32_t *states;                     // [F]
    float *melpow;                       // [F][n_mels]
    uint32_t *clipmax;                   // [n_clips]  max mel power (float bits)
    uint8_t *rake_raw;                   // [F]
    // outputs of the whole call, indexed through out_off (may be null)
    double *out_f0; uint8_t *out_voiced; double *out_vprob; float *out_rms;
    uint8_t *out_rake; float *out_sdb;   // clip c's dB image starts at n_mels * out_off[c]
    int16_t *out_bin;                    // decoded pitch bin, -1 unvoiced
    float *out_colmean; int64_t out_total;   // [3][out_total] column means of the dB image (all / low half / high half of the mel rows)
    double rake_ratio;
    int32_t rake_min_frames, rake_max_frames;
    double f0_unvoiced;                  // what an unvoiced frame's f0 reads: NaN (librosa.pyin fill_na) or 0.0 (np.nan_to_num)
};

constexpr int kViterbiChunk = 16;   // steps per composed back-pointer map

void launch_frame(const PassParams &p, const DevTables &t, hipStream_t s);
bool frame_cmnd_supported(int max_period);
int trough_row_doubles_host(int n_lags);    // doubles of a dfn row that holds a frame's trough list (PassParams::troughs)   // the frame kernel's LDS holds the CMND rows of a workgroup's frames (PassParams::cmnd_in_frame)
void launch_pyin_obs(const PassParams &p, const DevTables &t, hipStream_t s);
hipError_t launch_viterbi(const PassParams &p, const DevTables &t, const double *host_lt_band, hipStream_t s);
// time-split pass: speculative runs (grid = segments), lock-on runs (grid = segments that have a predecessor, listed in
// lock_order), stitch + back-trace, verification; seg_order (device) lists 0..n_seg-1
// (n_spec: entries of seg_order = speculative runs to launch; a hybrid pass lists the segments behind every clip's first only)
// (aux + ev[2]: when given, the clips without a never-met lock-on run are stitched, verified and walked on `aux` while the
// rounds of second speculation of the others run on `s`; `s` has joined `aux` when the call returns)
hipError_t launch_viterbi_split(const PassParams &p, const DevTables &t, const double *host_lt_band, const int32_t *seg_order, int n_spec,
                                const int32_t *lock_order, int n_lock, hipStream_t s, hipStream_t aux = nullptr, hipEvent_t *ev = nullptr);
// the speculative runs alone (a hybrid pass launches them behind its frame stage, beside the sequential kernel's last chunks,
// and the rest -- launch_viterbi_split with n_spec = 0 -- behind both)
hipError_t launch_viterbi_split_spec(const PassParams &p, const DevTables &t, const double *host_lt_band, const int32_t *seg_order, int n_spec, hipStream_t s);
bool viterbi_split_applies(const PassParams &p, const DevTables &t);
hipError_t viterbi_verify_fetch(long long *dst, bool reset);
int viterbi_tube_record_ints();   // counters of the time-split verification kernel (viterbi.hip g_verify_dbg)
bool viterbi_band_applies(const PassParams &p, const DevTables &t);   // the band-specialised kernels (the ones that can wait for chunk flags) take this geometry
void launch_chunk_signal(uint32_t *flag, uint32_t gen, hipStream_t s);
void launch_decode(const PassParams &p, const DevTables &t, hipStream_t s);
// first_bad (device, preset to ~0): smallest index of a sample of pcm[0..n) that is NaN or infinite
void launch_finite_check(const float *pcm, int64_t n, unsigned long long *first_bad, hipStream_t s);
void launch_finalize_mel(const PassParams &p, const DevTables &t, hipStream_t s);
void launch_rake_from_db(const float *sdb, int n_mels, int64_t F, double ratio, int min_frames, int max_frames,
                         uint8_t *raw, uint8_t *out, hipStream_t s);
void launch_stream_advance(StreamCtl *ctl, const float *staging, int n_push, float *pcm, int hop, hipStream_t s);
void launch_stream_gather(const StreamCtl *ctl, const float *rms, const double *vprob, const int32_t *live, void *result,
                          hipStream_t s);
hipError_t obs_debug_fetch(long long *dst);                  // pyin_obs_kernel section cycles (AEGIS_ABLATE&256)
hipError_t frame_debug_fetch(long long *dst);                // frame_yin_kernel section cycles (AEGIS_ABLATE&128)
hipError_t viterbi_debug_fetch(long long *dst, bool reset);   // per-wave section cycles (zeros unless AEGIS_ABLATE&64)
hipError_t viterbi_span_fetch(long long *dst);               // arg-max span statistics (zeros unless AEGIS_ABLATE&512); reading resets
hipError_t viterbi_configure();   // raises the dynamic-LDS limits once (all kernels)
hipError_t viterbi_set_lds_limits();   // viterbi.hip's share of it

}  // namespace aegis
