/* aegis_hip.h -- C ABI of libaegis_hip.so, the MI355X (gfx950) implementation of
 * the Aegis Engine analyze hot path.
 *
 * The reference has no FFI of its own: its seam is the Python class
 * `AegisEngine` (/root/reference/aegis_engine.py:16-216), which hands the
 * per-frame arithmetic to librosa.  Each entry point below names the reference
 * call(s) it replaces; INTEGRATION.md shows the ctypes stub a maintainer would
 * add to aegis_engine.py.  Plain pointers and sizes only; nothing is thrown
 * across the boundary; every function returns 0 or a negative errno-style code
 * and leaves a message retrievable with aegis_last_error().
 *
 * Frame convention (librosa center=True, pad_mode="constant"):
 *   n_frames(clip) = 1 + n_samples / hop_length           -> aegis_frames_for()
 * Batch outputs are concatenated clip after clip in that frame order.
 */
#ifndef AEGIS_HIP_H
#define AEGIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEGIS_ABI_VERSION 2

/* return codes */
#define AEGIS_OK 0
#define AEGIS_ERR_INVALID (-22)     /* bad argument / unsupported configuration */
#define AEGIS_ERR_NOMEM (-12)       /* host or device allocation failed */
#define AEGIS_ERR_DEVICE (-5)       /* HIP runtime error (no device, launch failure) */
#define AEGIS_ERR_UNSUPPORTED (-95) /* valid request this build cannot serve */

/* stage selection bits for aegis_analyze_*(): which outputs are produced */
#define AEGIS_STAGE_MEL 0x1   /* mel power -> dB image      aegis_engine.py:25-26 */
#define AEGIS_STAGE_RAKE 0x2  /* rake mask (implies MEL)    aegis_engine.py:54, vision.py:3-38 */
#define AEGIS_STAGE_PYIN 0x4  /* f0 / voiced / voiced_prob  aegis_engine.py:63,67, worker.py:9-15 */
#define AEGIS_STAGE_RMS 0x8   /* frame RMS                  aegis_engine.py:70 */
#define AEGIS_STAGE_ALL 0xF
/* options carried in the same bitmask */
#define AEGIS_OPT_CHECK_FINITE 0x10 /* librosa.util.valid_audio (inside librosa.load / pyin): a NaN or infinite sample makes
                                       the call fail with AEGIS_ERR_INVALID, "Audio buffer is not finite everywhere (clip N)".
                                       Checked on the device, where every sample is read anyway; blocking calls only
                                       (aegis_analyze_batch, aegis_analyze_batch_device with sync != 0) */
#define AEGIS_OPT_F0_ZERO 0x20      /* unvoiced f0 = 0.0 instead of NaN: np.nan_to_num(f0), aegis_engine.py:69 */

typedef struct aegis_handle aegis_handle;

#define AEGIS_PYIN_INIT_UNVOICED 0
#define AEGIS_PYIN_INIT_UNIFORM 1

/* Replaces AegisEngine.__init__ (aegis_engine.py:17-20) plus the librosa
 * defaults the reference relies on.  Zero / NaN fields take the default. */
typedef struct aegis_config {
    int32_t sample_rate;        /* 44100 (v2 engine: 22050, aegis_engine_financial.py:36) */
    int32_t hop_length;         /* 512 */
    int32_t n_fft;              /* 2048 (mel STFT; pYIN/RMS frame_length is librosa's fixed 2048) */
    int32_t n_mels;             /* 128 */
    double fmin;                /* pYIN fmin; 0 -> note_to_hz('E2') = 82.4068892282175 */
    double fmax;                /* pYIN fmax; 0 -> note_to_hz('C6') = 1046.5022612023945 */
    int32_t device;             /* HIP device ordinal; -1 = host tables only (no GPU touched:
                                   aegis_get_table/param work, analyze calls fail with AEGIS_ERR_DEVICE) */
    int32_t pyin_init;          /* initial distribution of the pYIN HMM (librosa core/pitch.py::pyin builds p_init, then
                                   sequence.viterbi(observation_probs, transition, p_init=p_init)):
                                   AEGIS_PYIN_INIT_UNVOICED (0, default) = librosa's published code: p_init = zeros(2B),
                                   p_init[B:] = 1/B -- the chain starts unvoiced, voiced states start at log(0 + tiny);
                                   AEGIS_PYIN_INIT_UNIFORM (1) = 1/(2B) on every state (SURVEY.md P11's reading, the
                                   behaviour of ABI version 1).  Every Turbo-Mode chunk (aegis_engine.py:197-210) and
                                   every clip starts its own chain, so the choice shows in their first frames. */
    int64_t max_frames_per_pass; /* workspace bound; 0 -> as many frames as a third of the free device memory holds
                                    (~10.3 KB each), between 2^21 and 2^24 */
} aegis_config;

/* Per-batch outputs, concatenated over clips.  Any pointer may be NULL to skip
 * that output.  Used with host pointers by aegis_analyze_batch() and with device
 * pointers by aegis_analyze_batch_device().  dtypes follow the reference's
 * raw_data dict (aegis_engine.py:72-75). */
typedef struct aegis_outputs {
    double *f0;           /* [F_total]  Hz, NaN where unvoiced (librosa.pyin fill_na) */
    uint8_t *voiced_flag; /* [F_total]  0/1 */
    double *voiced_prob;  /* [F_total] */
    float *rms;           /* [F_total] */
    uint8_t *rake_mask;   /* [F_total]  0/1 */
    float *S_dB;          /* per clip [n_mels, F_clip] C-order, clip after clip (n_mels*F_total) */
    int16_t *pitch_bin;   /* [F_total]  the decoded pitch bin (f0 == freqs[bin], aegis_get_table "freqs"), -1 where unvoiced:
                             lets the event logic take hz_to_midi(f0) from a table of n_pitch_bins entries */
    float *sdb_col_means; /* [3][F_total]  per frame np.mean(S_dB, axis=0), np.mean(S_dB[:n_mels/2], axis=0) and
                             np.mean(S_dB[n_mels/2:], axis=0) in NumPy's order (float32, row after row): all the v2 guitar
                             filters read of the dB image (guitar_specific.py:60-141), without moving the image */
} aegis_outputs;

int aegis_abi_version(void);

/* aegis_engine.py:17-20.  Builds the device tables (Hann window, Slaney mel
 * filterbank, pYIN priors, HMM log-transitions) and the workspace. */
int aegis_create(const aegis_config *cfg, aegis_handle **out);
/* Threads.  Every entry that takes a handle holds the handle's mutex for the whole call: a handle may be SHARED by several
 * threads (the reference's servers share one engine across requests, server.py:51) and their calls are serialised; use one
 * handle per thread for concurrency.  What is NOT supported is driving one call from two threads -- a helper thread issuing
 * HIP work into a handle's streams while another thread is inside a call on it (round 3's feeder-thread experiment hung
 * inside the runtime) -- and aegis_destroy racing with a call in flight.  One handle sizes its workspace from the device
 * memory that was free when it was created (max_frames_per_pass = 0): several handles on one device should be given an
 * explicit max_frames_per_pass.
 * aegis_destroy waits for the handle's own streams (at most ten seconds; work still running then is leaked with a message
 * on stderr rather than waited for), synchronises the device and frees everything; with aegis_stream objects still open it
 * only marks the handle and the last aegis_stream_free() tears it down. */
void aegis_destroy(aegis_handle *h);
const char *aegis_last_error(const aegis_handle *h); /* h may be NULL: last create error */

/* F = 1 + n_samples / hop_length  (librosa framing used by every stage). */
int64_t aegis_frames_for(const aegis_handle *h, int64_t n_samples);

/* aegis_engine.py:50-75 for a batch of decoded clips (float32 mono PCM in host
 * memory): mel -> dB -> rake mask, pYIN, RMS.  Blocking; does H2D, kernels, D2H.
 * `stages` is a bitmask of AEGIS_STAGE_*.  Clips of length 0 produce 1 frame of
 * silence (the Python layer returns None before calling, aegis_engine.py:51). */
int aegis_analyze_batch(aegis_handle *h, const float *const *pcm, const int64_t *n_samples,
                        int32_t n_clips, double rake_sensitivity, uint32_t stages,
                        aegis_outputs *host_out);

/* Same computation with the PCM already resident in device memory:
 * clip c occupies d_pcm[sample_offsets[c] .. sample_offsets[c+1]); `sample_offsets`
 * is a host array of n_clips+1 entries.  Outputs are device pointers.  Work is
 * enqueued on `stream` (a hipStream_t, NULL = the handle's own stream) and the
 * call returns without synchronising unless `sync` is non-zero.  Calls on one handle are
 * serialised by an internal mutex (the reference's servers share one engine across requests);
 * use one handle per thread for concurrency. */
int aegis_analyze_batch_device(aegis_handle *h, const float *d_pcm, const int64_t *sample_offsets,
                               int32_t n_clips, double rake_sensitivity, uint32_t stages,
                               aegis_outputs *device_out, void *stream, int32_t sync);

/* AegisEngine.detect_rake_patterns(S_dB) (aegis_engine.py:38-39 -> vision.py:3-38) on a
 * caller-supplied dB image in host memory, [n_mels, n_frames] C-order; mask_out is
 * uint8[n_frames] in host memory.  Blocking. */
int aegis_rake_patterns(aegis_handle *h, const float *S_dB, int32_t n_mels, int64_t n_frames,
                        double broadband_threshold_ratio, uint8_t *mask_out);

/* --- constant-Q magnitudes (BASELINE.json configs[2]) --------------------------------
 * The reference's only CQT is librosa.feature.chroma_cqt inside the auto-matcher score
 * (aegis_engine_core/auto_matcher.py:68-69).  This computes the DIRECT transform librosa.cqt(hop_length=hop,
 * fmin, n_bins, bins_per_octave, filter_scale, norm=1, window='hann', scale=True, pad_mode='constant')
 * approximates octave by octave (wavelet atoms of librosa 0.10 filters.wavelet), as a block-sparse float32
 * GEMM on the MFMA units.  Host PCM in, |C| out: per clip [n_bins, F_clip] C-order, clip after clip.
 * Zero arguments take the defaults n_bins=84, bins_per_octave=12, fmin=C1, filter_scale=1.  n_bins <= 256 (chroma_cqt's
 * 7 x 36 = 252 bins run as three launches of 11 row tiles); the longest atom may span up to 131 072 taps.  Blocking. */
int aegis_cqt(aegis_handle *h, const float *const *pcm, const int64_t *n_samples, int32_t n_clips,
              int32_t n_bins, int32_t bins_per_octave, double fmin, double filter_scale, float *mag_out);
/* The same with the PCM and the result resident in device memory (clip c = d_pcm[sample_offsets[c] .. sample_offsets[c+1]),
 * `sample_offsets` a host array of n_clips + 1 entries; d_mag_out as mag_out).  Enqueued on `stream` (NULL = the handle's);
 * returns without waiting for the kernel unless sync != 0. */
int aegis_cqt_device(aegis_handle *h, const float *d_pcm, const int64_t *sample_offsets, int32_t n_clips,
                     int32_t n_bins, int32_t bins_per_octave, double fmin, double filter_scale, float *d_mag_out,
                     void *stream, int32_t sync);

/* librosa.feature.chroma_cqt(y, sr) as the auto-matcher calls it (aegis_engine_core/auto_matcher.py:68-69) behind the
 * magnitudes of aegis_cqt: the folding matrix filters.cq_to_chroma -- every CQT bin feeds exactly one chroma class,
 * bin_class[n_bins] with entries in [0, n_chroma) -- and util.normalize(norm=inf) per frame, both on the device, so only
 * n_chroma x F floats per clip come back instead of n_bins x F.  Host PCM in; chroma_out: per clip [n_chroma, F_clip]
 * C-order, clip after clip.  `fmin` is the bank's (tuning-shifted) lowest frequency.  n_chroma <= 24.  Blocking. */
int aegis_chroma_cqt(aegis_handle *h, const float *const *pcm, const int64_t *n_samples, int32_t n_clips,
                     int32_t n_bins, int32_t bins_per_octave, double fmin, double filter_scale, int32_t n_chroma,
                     const int32_t *bin_class, float *chroma_out);

/* --- incremental analysis of one clip (BASELINE.json configs[4]; the reference has no streaming path:
 * financial_app_realtime.py analyses whole files).  Samples are pushed in any chunk sizes; every frame whose
 * centred 2048-sample window is complete is analysed at once (mel, YIN, observation) and the Viterbi advances
 * over it, its column carried exactly between pushes.  aegis_stream_push returns, for the frames it produced,
 * the final rms and voiced_prob plus a zero-lag decode (arg-max state of the current column: `live_state`,
 * < n_pitch_bins = voiced bin, otherwise unvoiced).  aegis_stream_close zero-pads the tail exactly as the
 * batch path does, back-traces, and returns arrays IDENTICAL to aegis_analyze_batch on the whole signal. */
typedef struct aegis_stream aegis_stream;
typedef struct aegis_stream_frames {
    float *rms;           /* [>= frames produced by the push] host arrays, any may be NULL */
    double *voiced_prob;
    int32_t *live_state;
} aegis_stream_frames;
int aegis_stream_open(aegis_handle *h, int64_t max_samples, aegis_stream **out);
int aegis_stream_push(aegis_stream *st, const float *samples, int64_t n, aegis_stream_frames *out, int64_t *n_frames);
int aegis_stream_close(aegis_stream *st, double rake_sensitivity, aegis_outputs *host_out, int64_t *n_frames);
void aegis_stream_free(aegis_stream *st);

/* --- v2 "financial" trend filters on pitch tracks -------------------
 * One op-coded entry over a ragged batch of float64 series in host memory: series i is
 * x[offsets[i] .. offsets[i+1]) (NaN = unvoiced); outputs are host arrays of offsets[n_series]
 * elements (double unless noted).  Blocking.  Each op replaces the reference method named:
 *
 *  op                        params                         outs
 *  AEGIS_TREND_SMA           window                         out          FinancialPitchAnalyzer.simple_moving_average (financial_analysis.py:45-69)
 *  AEGIS_TREND_EMA           span                           out          .exponential_moving_average (:71-107)
 *  AEGIS_TREND_BOLLINGER     window, num_std                ma,upper,lower   .bollinger_bands (:113-146)
 *  AEGIS_TREND_ARTICULATION  window, sensitivity            int8 codes   .detect_articulation_bollinger (:148-197): 0 None 1 normal 2 bend 3 vibrato 4 noise
 *  AEGIS_TREND_MACD          fast, slow, signal             macd,signal,hist .macd (:203-226)
 *  AEGIS_TREND_SLIDES        threshold                      int8 codes   .detect_slides_macd (:228-268): 0 None 1 normal 2 slide_up 3 slide_down
 *  AEGIS_TREND_RSI           period [, averages]            out          .rsi (:274-320); averages != 0: the two Wilder averages
 *                                                          (avg_gain, avg_loss; NaN where the RSI is the constant 50) instead, for callers
 *                                                          that need the RSI at a few positions only (filter_ghost_notes_rsi :322-362)
 *  AEGIS_TREND_SAVGOL        window, symmetric, coef[window] out         FinancialNoiseFilters.savitzky_golay (financial_filters.py:25-59); coef = reversed scipy savgol_coeffs
 *  AEGIS_TREND_KALMAN        process_var, measurement_var   out          .kalman_filter (:62-99)
 *  AEGIS_TREND_HOLT          alpha, beta                    out          .holt_winters (:102-141)
 *  AEGIS_TREND_CONSENSUS     k  (x = k stacked rows, n_series = 1)  median, confidence   multi_filter_consensus (:256-298)
 *  AEGIS_TREND_PITCH_ANALYSIS  sg_window, sg_symmetric, sg_coef[sg_window], kalman q, r, holt alpha, beta, band window, band num_std,
 *                            slide threshold       trend, int8 articulation codes, int8 slide codes, confidence
 *                            FinancialPitchAnalyzer.analyze_pitch_financial (financial_analysis.py:368-423) as ONE call: the
 *                            Savitzky-Golay / Kalman / Holt consensus, the Bollinger articulation and MACD slide state machines and
 *                            the band-width confidence -- the ops above, with the four independent sequential walks (Kalman, Holt,
 *                            NaN compaction, MACD) on four streams at once, one upload and one synchronisation
 *
 * A series shorter than the window is AEGIS_ERR_INVALID for SMA/Bollinger (the reference raises IndexError). */
#define AEGIS_TREND_SMA 1
#define AEGIS_TREND_EMA 2
#define AEGIS_TREND_BOLLINGER 3
#define AEGIS_TREND_ARTICULATION 4
#define AEGIS_TREND_MACD 5
#define AEGIS_TREND_SLIDES 6
#define AEGIS_TREND_RSI 7
#define AEGIS_TREND_SAVGOL 8
#define AEGIS_TREND_KALMAN 9
#define AEGIS_TREND_HOLT 10
#define AEGIS_TREND_CONSENSUS 11
#define AEGIS_TREND_PITCH_ANALYSIS 12
int aegis_trend(aegis_handle *h, int32_t op, const double *x, const int64_t *offsets, int32_t n_series,
                const double *params, int32_t n_params, void *const *outs, int32_t n_outs);

/* The ghost-note filter's RSI lookup for a batch of clips in one call (FinancialPitchAnalyzer.filter_ghost_notes_rsi,
 * /root/reference/aegis_engine_core_v2/financial_analysis.py:322-362).  The reference adds 1 over [int(start*10), int(end*10))
 * per note to a density track of int(max_end*10) elements, takes the RSI of the track (period 14) and reads it at
 * int(start*10) of every note.  Here the tracks are built on the device from the notes' intervals and only the two Wilder
 * averages at each note's own position come back (avg_gain, avg_loss: the caller forms 100 - 100 / (1 + gain / loss), the
 * reference's operations; NaN where the RSI is the constant 50 of the first `period` positions or of a track shorter than
 * period + 1, and where the position lies outside the track).  The same values as AEGIS_TREND_RSI (averages = 1) on the
 * density tracks, without materialising 77 k elements per three-minute clip on the host.
 *   ev_a, ev_b      [event_off[n_series]]  int(start*10), int(end*10) of every note, clip after clip
 *   event_off       [n_series + 1]         first note of each clip
 *   track_len       [n_series]             int(max_end * 10) of each clip (0: the clip is skipped)
 *   avg_gain, avg_loss  [event_off[n_series]]  host, written */
int aegis_ghost_rsi(aegis_handle *h, const int64_t *ev_a, const int64_t *ev_b, const int64_t *event_off, int32_t n_series,
                    const int64_t *track_len, int32_t period, double *avg_gain, double *avg_loss);

/* --- note events and Standard MIDI Files for a batch of clips: host code, no GPU, no handle ----------------
 * SURVEY.md 8(f) rank 1.  aegis_extract_events replaces get_midi_events + detect_articulations
 * (aegis_engine_core/midi_logic.py:32-148, 6-30) from the point where the frame arrays are gated: the caller passes,
 * concatenated clip after clip (clip c = frames frame_off[c] .. frame_off[c+1]),
 *   sounding   u8   voiced_flag & ~(rms_db < noise_gate_db) & (f0 > 0) & ~rake_mask          (midi_logic.py:62-68)
 *   semitones  f64  librosa.hz_to_midi(f0) on the sounding frames (anything elsewhere)      (midi_logic.py:69)
 *              -- or NULL, with pitch_bin i16 (aegis_outputs.pitch_bin) and bin_semitones f64[n_pitch_bins] =
 *              hz_to_midi(freqs): the same values without a logarithm per frame
 *   rms_db     f32  librosa.amplitude_to_db(rms, ref=np.max) of the clip                    (midi_logic.py:51)
 *   probs      f64  voiced_probs
 * (the two logarithms stay with NumPy, whose float32 log10 / float64 log2 kernels are not libm's: the Python binding
 * spectrogram-midi_amd/events_native.py prepares them for a whole batch in a handful of array calls).  Events come back
 * clip after clip; clip_event_off[n_clips + 1] delimits them.  A note whose articulation decision lies within 1e-9 of one
 * of the reference's thresholds (pitch tracks sit on a 0.1-semitone grid: exact ties occur) is not decided here: such
 * runs are listed in risky_runs (clip, start, end), their clips produce no events, and the caller calls again with the
 * reference's own verdicts for them in batch->fits (detect_articulations through np.polyfit, midi_logic.py:6-30).
 * Returns the number of events (which may exceed cap: nothing past cap is written) or a negative code;
 * aegis_events_last_error() has the message (thread-local).
 * aegis_render_smf replaces the SMF block of AegisEngine.extract_events (aegis_engine.py:98-179, mido's writer: type 1,
 * 480 ticks per beat, two tracks main / safe, running status, end_of_track): one file per clip, concatenated in `out`,
 * clip_byte_off[n_clips + 1] delimits them; returns the total size (which may exceed cap: then nothing is complete). */
typedef struct aegis_event {
    int32_t clip, note, start, end, velocity; /* start / end: inclusive frame indices (midi_logic.py:74-79) */
    uint8_t track;                            /* 1 'main', 0 'safe' */
    uint8_t technique;                        /* 0 None 1 vibrato 2 bend 3 slide 4 hammer_on 5 pull_off */
    uint8_t reserved0, reserved1;
    float rms_energy;                         /* dB, the note's first frame */
    int32_t reserved2;
    double confidence, slope;
} aegis_event;
typedef struct aegis_event_params {
    int32_t sample_rate, hop_length;
    double confidence_threshold;              /* 0.70  aegis_engine.py:85 */
    double sustain_ms, min_note_duration_ms;  /* 50, 50  midi_logic.py:37-38 (the noise gate is applied by the caller) */
} aegis_event_params;
typedef struct aegis_run_fit {                /* one note-run and, in aegis_event_batch.fits, its articulation verdict */
    int32_t clip, start, end;                 /* frames start..end inclusive */
    int32_t technique;                        /* 0 None 1 vibrato 2 bend 3 slide */
    double slope;
} aegis_run_fit;
typedef struct aegis_event_batch {
    int32_t n_clips;
    int32_t reserved;
    const int64_t *frame_off;                 /* [n_clips + 1] */
    const uint8_t *sounding;
    const double *semitones;                  /* or NULL with the next two */
    const int16_t *pitch_bin;
    const double *bin_semitones;
    const float *rms_db;
    const double *probs;
    const aegis_run_fit *fits;                /* verdicts for runs an earlier call listed as risky, sorted by (clip, start); may be NULL */
    int64_t n_fits;
} aegis_event_batch;
int64_t aegis_extract_events(const aegis_event_params *params, const aegis_event_batch *batch, aegis_event *events, int64_t cap,
                             int64_t *clip_event_off, aegis_run_fit *risky_runs, int64_t risky_cap, int64_t *n_risky);
int64_t aegis_render_smf(int32_t sample_rate, int32_t hop_length, int32_t midi_program, double vibrato_rate,
                         double vibrato_depth, int32_t n_clips, const aegis_event *events, const int64_t *clip_event_off,
                         uint8_t *out, int64_t cap, int64_t *clip_byte_off);
const char *aegis_events_last_error(void);

/* --- introspection used by the tests (no reference counterpart) ------------- */

/* Host-side copies of the tables the kernels use.  `name` is one of
 * "hann" f64[n_fft], "mel_dense" f32[n_mels*(1+n_fft/2)], "thresholds" f64[101],
 * "beta_probs" f64[100], "beta_cumsum" f64[101], "boltz_fact" f64[n], "boltz_exp" f64[n],
 * "log_trans_band" f64[4*n_cls*width], "log_trans_pack" f64[2*(3H^2+3H+2)] (H = (width-1)/2: the band table
 * without its duplicate (v,v') blocks and unreachable edge-row entries, as the Viterbi kernel keeps it in LDS),
 * "freqs" f64[n_pitch_bins], "twiddle" f64[2*n_fft].
 * Returns the element count (or a negative code); copies min(count, cap) elements. */
int64_t aegis_get_table(const aegis_handle *h, const char *name, void *dst, int64_t cap);

/* Replaces one of the float64 prior tables with caller-supplied values (same length as the
 * built-in one): "beta_probs" [100] (beta_cumsum / beta_suffix are re-derived), "boltz_fact",
 * "boltz_exp", "freqs".  librosa builds these with scipy.stats / numpy at every call
 * (core/pitch.py::pyin); the Python binding passes the same arrays so that the observation
 * probabilities are bit-identical to the reference on that host.  C callers may keep the
 * built-in closed forms (within 1e-14 relative). */
int aegis_set_table(aegis_handle *h, const char *name, const double *data, int64_t count);

/* Scalar parameters derived at create time.  name in {"min_period","max_period",
 * "n_lags","n_pitch_bins","transition_width","n_trans_classes","max_frames_per_pass",
 * "lag_stride","yin_stride","obs_stride","last_frames","pyin_init"}; of the last call (its last pass): "last_passes",
 * "last_chunks", "last_dense", "last_proportional", "last_balanced", "last_persistent", "last_split_segments"; since create:
 * "split_passes", "split_segments", "split_flagged_clips" (clips the sequential kernel decoded again), "split_unlocked_clips";
 * of the last split pass: "split_rounds" (rounds of second speculation that had work), "split_viterbi_us" (measured time of its Viterbi
 * kernels, automatic passes only), "split_cooldown" (calls left that plan sequentially after split passes that did not pay),
 * "last_hybrid_step" (the step up to which the sequential kernel ran every clip under the frame stage before the rest was
 * cut into segments; 0: not a hybrid split pass) -- the time-split Viterbi, csrc/viterbi.hip. */
int64_t aegis_get_param(const aegis_handle *h, const char *name);

/* Copies an intermediate of the most recent pass (device -> host), for stage-level
 * parity tests.  name in {"dfn" f64[F*lag_stride] (pyin's difference function d[tau], the one pYIN intermediate the
 * frame stage leaves in HBM), "yin" f64[F*yin_stride] (the CMND rows: only on handles created under AEGIS_DEBUG_STAGES=1),
 * "logobs" f64[F*obs_stride], "logunv" f64[F], "states" i32[F], "melpow" f32[F*n_mels]}.
 * Returns the element count available; copies min(count, cap).
 * "viterbi_stats" i64[3] (reading resets; "viterbi_stats_peek" does not): wave-steps of the band Viterbi since the last
 * reset, how many of them took the exact observed-sources-only path, and how many were voiced waves that skipped the step
 * because all their targets were dead at an easy frame (bench.py reports the ratios).
 * "persistent_fallbacks" i64[1]: calls this handle repeated with one Viterbi launch per time chunk after its single
 * launch per pass gave up waiting for the frame stage (kernels serialised by a counter-collecting profiler, for one).
 * "fail_allocs": test hook, makes the next `cap` workspace growths fail as hipMalloc would (the out-of-memory retry:
 * an analyze call that cannot allocate halves max_frames_per_pass, down to 2^21 frames, and plans its passes again).
 * "throw_bad_alloc" / "throw_length_error" / "throw_runtime_error" / "throw_int": test hooks of the exception barrier
 * (the body throws; the call returns AEGIS_ERR_NOMEM / AEGIS_ERR_DEVICE like any other failure).
 * Profiling builds only (csrc/Makefile EXTRA=-DAEGIS_ABLATE=64|128, -DCQT_ABLATE=8; zeros otherwise):
 * "viterbi_cycles" i64[16 waves][8], "cqt_cycles" i64[16] -- in-kernel s_memtime
 * section counters read by tools/viterbi_cycles.py, cqt_cycles.py (reading resets them). */
int64_t aegis_debug_fetch(aegis_handle *h, const char *name, void *dst, int64_t cap);

/* Kernel timing of the most recent aegis_analyze_batch_device() with sync != 0,
 * measured with hipEvents on the stream the kernels ran on.  name in
 * {"frame","pyin_obs","viterbi","finalize","total"}; milliseconds,
 * negative when unavailable.  aegis_set_profiling(h, 1) enables the events. */
int aegis_set_profiling(aegis_handle *h, int32_t on);
double aegis_last_kernel_ms(const aegis_handle *h, const char *name);
int aegis_last_kernel_launches(const aegis_handle *h, const char *name); /* launches summed into the figure above */

#ifdef __cplusplus
}
#endif
#endif /* AEGIS_HIP_H */
