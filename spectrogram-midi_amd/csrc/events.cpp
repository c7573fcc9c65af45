// Host-side consumers of the analyze kernels, batched over clips (no GPU): note events from frame arrays and the
// two-track Standard MIDI File, i.e. what the reference runs per clip in Python after every analysis and on every
// slider move (/root/reference/aegis_engine_core/midi_logic.py:6-30, 32-148; /root/reference/aegis_engine.py:98-179).
// once the analysis is ~10^4 x real time these loops are what a folder waits for.
//
// Exactness.  Everything here is IEEE arithmetic in the reference's order, with two inputs prepared by the Python
// layer because they go through NumPy's own float32 log10 / float64 log2 kernels (SIMD variants that are not libm's):
// `rms_db` = amplitude_to_db(rms, ref=np.max) and `semitones` = hz_to_midi(f0) on the sounding frames.  The
// articulation fit is the closed-form least-squares line (np.polyfit solves the same problem through an SVD; the
// slopes agree to ~1e-13); a run whose decision lies within 1e-9 of a threshold is not decided here: it is listed as
// risky, the caller asks the reference's own arithmetic (np.polyfit) for those runs only and calls again with the
// verdicts (pitch tracks sit on a 0.1-semitone grid: exact ties occur, several per clip on real material).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aegis_hip.h"

namespace {

thread_local std::string g_events_error;

enum { TECH_NONE = 0, TECH_VIBRATO = 1, TECH_BEND = 2, TECH_SLIDE = 3, TECH_HAMMER = 4, TECH_PULL = 5 };

struct Fit { int code; double slope; bool risky; };

// detect_articulations (midi_logic.py:6-30) for frames start..end inclusive, every frame sounding; semi(i) = the
// semitone value of frame i (an array element, or the decoded bin's table entry: the same double either way)
template <class Semi>
Fit fit_run(const Semi &semi, int64_t start, int64_t end) {
    const int64_t n = end - start + 1;
    if (n < 3) return {TECH_NONE, 0.0, false};
    double sy = 0.0, sxy = 0.0;
    for (int64_t i = 0; i < n; ++i) { const double y = semi(start + i); sy += y; sxy += (double)i * y; }
    const double nn = (double)n;
    const double sx = nn * (nn - 1) / 2;
    const double sxx = (nn - 1) * nn * (2 * nn - 1) / 6;
    const double slope = (nn * sxy - sx * sy) / (nn * sxx - sx * sx);
    const double icpt = (sy - slope * sx) / nn;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        const double w = semi(start + i) - (slope * (double)i + icpt);
        lo = std::min(lo, w); hi = std::max(hi, w);
    }
    const double spread = hi - lo;
    const double eps = 1e-9;
    const bool risky = std::fabs(spread - 0.3) < eps || std::fabs(slope - 0.05) < eps || std::fabs(std::fabs(slope) - 0.02) < eps;
    int code = TECH_NONE;
    if (spread > 0.3) code = TECH_VIBRATO;
    else if (slope > 0.05) code = TECH_BEND;
    else if (std::fabs(slope) > 0.02) code = TECH_SLIDE;
    return {code, code ? slope : 0.0, risky};
}

// get_midi_events (midi_logic.py:32-148) for one clip; returns false when a decision is too close to call
// semi(i): semitone value of frame i; pitch(i): np.rint of it (half to even) as an integer.
template <class Semi, class Pitch>
bool clip_events_t(const aegis_event_params &P, int64_t F, const uint8_t *sounding, const Semi &semi, const Pitch &pitch_of,
                   const float *rms_db, const double *probs, int32_t clip, const aegis_run_fit *fits,
                   int64_t n_fits, std::vector<aegis_event> &out, std::vector<aegis_run_fit> &risky_out) {
    const int64_t min_frames = (int64_t)((P.min_note_duration_ms / 1000.0) * P.sample_rate / P.hop_length);
    const int64_t sustain_frames = (int64_t)((P.sustain_ms / 1000.0) * P.sample_rate / P.hop_length);
    std::vector<aegis_event> ev;
    bool ok = true;
    int64_t i = 0;
    while (i < F) {
        if (!sounding[i]) { ++i; continue; }
        const int64_t pitch = pitch_of(i);
        int64_t j = i + 1;
        while (j < F && sounding[j] && pitch_of(j) == pitch) ++j;
        const int64_t start = i, end = j - 1;
        i = j;
        if (end - start < min_frames) continue;                         // midi_logic.py:109 (end is inclusive)
        Fit f = fit_run(semi, start, end);
        if (f.risky) {                                                   // the caller's verdict for this run, if it gave one
            const aegis_run_fit key{clip, (int32_t)start, 0, 0, 0.0};
            const aegis_run_fit *hit = std::lower_bound(fits, fits + n_fits, key, [](const aegis_run_fit &a, const aegis_run_fit &b) {
                return a.clip != b.clip ? a.clip < b.clip : a.start < b.start; });
            if (hit != fits + n_fits && hit->clip == clip && hit->start == (int32_t)start && hit->end == (int32_t)end) {
                f = {hit->technique, hit->slope, false};
            } else {
                risky_out.push_back({clip, (int32_t)start, (int32_t)end, 0, 0.0});
                ok = false;
            }
        }
        aegis_event e{};
        e.clip = clip; e.note = (int32_t)pitch; e.start = (int32_t)start; e.end = (int32_t)end;
        e.rms_energy = rms_db[start];
        e.confidence = probs[start];
        float v = (e.rms_energy + 80.0f) * 1.5f;                        // float32 array arithmetic, then np.clip, then astype(int)
        v = v < 0.0f ? 0.0f : (v > 127.0f ? 127.0f : v);
        e.velocity = (int32_t)v;
        e.track = e.confidence >= P.confidence_threshold ? 1 : 0;
        e.technique = (uint8_t)f.code;
        e.slope = f.slope;
        ev.push_back(e);
    }
    if (!ok) return false;
    // join same-pitch neighbours across short gaps while the head carries no technique (midi_logic.py:112-124)
    std::vector<aegis_event> merged;
    for (const aegis_event &e : ev) {
        if (!merged.empty()) {
            aegis_event &head = merged.back();
            if (e.note == head.note && (int64_t)e.start - head.end <= sustain_frames && head.technique == TECH_NONE) {
                head.end = e.end;
                continue;
            }
        }
        merged.push_back(e);
    }
    // hammer-on / pull-off tagging of consecutive notes (midi_logic.py:127-146; the level ratio divides two NEGATIVE
    // float32 dB values, as written there)
    const double frame_ms = ((double)P.hop_length / (double)P.sample_rate) * 1000;
    for (size_t k = 1; k < merged.size(); ++k) {
        const aegis_event &prev = merged[k - 1];
        aegis_event &cur = merged[k];
        if ((double)(cur.start - prev.end) * frame_ms < 30) {
            const int interval = cur.note - prev.note;
            bool softer = (double)cur.velocity / (double)std::max(prev.velocity, 1) < 0.7;
            if (!softer) {
                const float q = cur.rms_energy / prev.rms_energy;       // max(prev, -80) == prev: the dB track is floored at -80
                softer = (double)q < 0.8;
            }
            if (softer && interval > 0 && interval <= 2) { cur.technique = TECH_HAMMER; cur.slope = 0.0; }
            else if (softer && interval >= -2 && interval < 0) { cur.technique = TECH_PULL; cur.slope = 0.0; }
        }
    }
    out.insert(out.end(), merged.begin(), merged.end());
    return true;
}

// get_midi_events for one clip from either form of the pitch track: semitones per frame (semi_in), or the analysis's
// decoded bins with hz_to_midi(freqs) as a table (no per-frame array is built: the table entry IS the frame's value,
// and its rounding is looked up instead of computed per frame)
bool clip_events(const aegis_event_params &P, int64_t F, const uint8_t *sounding, const double *semi_in, const int16_t *bins,
                 const double *bin_semi, int32_t *bin_pitch, const float *rms_db, const double *probs, int32_t clip,
                 const aegis_run_fit *fits, int64_t n_fits, std::vector<aegis_event> &out, std::vector<aegis_run_fit> &risky_out) {
    if (semi_in) {
        auto semi = [semi_in](int64_t i) { return semi_in[i]; };
        auto pitch = [semi_in](int64_t i) { return (int64_t)std::nearbyint(semi_in[i]); };
        return clip_events_t(P, F, sounding, semi, pitch, rms_db, probs, clip, fits, n_fits, out, risky_out);
    }
    auto semi = [bins, bin_semi](int64_t i) { return bins[i] >= 0 ? bin_semi[bins[i]] : 0.0; };
    auto pitch = [bins, bin_semi, bin_pitch](int64_t i) -> int64_t {   // bin_pitch: the worker's lazily filled rint table
        const int b = bins[i];
        if (b < 0) return 0;
        if (bin_pitch[b] == INT32_MIN) bin_pitch[b] = (int32_t)std::nearbyint(bin_semi[b]);
        return bin_pitch[b];
    };
    return clip_events_t(P, F, sounding, semi, pitch, rms_db, probs, clip, fits, n_fits, out, risky_out);
}

// ---- Standard MIDI File (aegis_engine.py:98-179 through mido: type 1, 480 ticks per beat, running status, end_of_track) ----
struct Msg { int64_t tick; uint8_t track, kind; int32_t a, b; };     // kind 0 on, 1 off, 2 pitchwheel

struct TrackBuf {
    std::vector<uint8_t> data;
    int status = -1;
    int64_t clock = 0;
    void varlen(int64_t v) {
        uint8_t g[10];
        int n = 0;
        g[n++] = (uint8_t)(v & 0x7F);
        v >>= 7;
        while (v) { g[n++] = (uint8_t)(0x80 | (v & 0x7F)); v >>= 7; }
        while (n) data.push_back(g[--n]);
    }
    void emit(int64_t tick, int st, int d0, int d1, int nd) {
        varlen(tick - clock);
        clock = tick;
        if (st != status) { data.push_back((uint8_t)st); status = st; }
        data.push_back((uint8_t)(d0 & 0x7F));
        if (nd > 1) data.push_back((uint8_t)(d1 & 0x7F));
    }
};

int64_t render_clip(int sr, int hop, int program, double vib_rate, double vib_depth, const aegis_event *ev, int64_t n,
                    std::vector<uint8_t> &out) {
    const double frame_ticks = (double)hop / (double)sr;
    const double TPS = 960.0;                                           // second2tick(1.0, 480, 500000)
    std::vector<Msg> rows;
    rows.reserve((size_t)n * 4);
    for (int64_t k = 0; k < n; ++k) {
        const aegis_event &e = ev[k];
        const int64_t t_on = (int64_t)((double)e.start * frame_ticks * TPS);
        const int64_t t_off = (int64_t)((double)e.end * frame_ticks * TPS);
        int vel = e.velocity;
        if (e.technique == TECH_HAMMER) vel = (int)((double)vel * 0.6);
        else if (e.technique == TECH_PULL) vel = (int)((double)vel * 0.5);
        const uint8_t trk = e.track ? 1 : 0;
        rows.push_back({t_on, trk, 0, e.note, vel});
        rows.push_back({t_off, trk, 1, e.note, 0});
        const int64_t length = t_off - t_on;
        if (e.technique == TECH_BEND) {
            const double semis = std::min(2.0, std::fabs(e.slope) * 10);
            const int top = (int)((double)(e.slope > 0 ? 1 : -1) * (semis / 2.0) * 8191);
            for (int i = 0; i < 15; ++i) {
                const double u = (double)i / 15;
                rows.push_back({t_on + (int64_t)(u * (double)length), trk, 2, (int)((double)top * (1 - std::pow(1 - u, 2.0))), 0});
            }
            rows.push_back({t_off, trk, 2, 0, 0});
        } else if (e.technique == TECH_VIBRATO) {
            const double secs = (double)length / TPS;
            const int count = std::max(10, std::min(20, (int)(secs * vib_rate * 4)));
            for (int i = 0; i < count; ++i) {
                const double frac = (double)i / (double)count;
                const double angle = frac * secs * vib_rate * 2 * M_PI;
                rows.push_back({t_on + (int64_t)(frac * (double)length), trk, 2, (int)(std::sin(angle) * 8191 * vib_depth), 0});
            }
            rows.push_back({t_off, trk, 2, 0, 0});
        }
    }
    std::stable_sort(rows.begin(), rows.end(), [](const Msg &a, const Msg &b) { return a.tick < b.tick; });
    TrackBuf tr[2];                                                     // [0] safe, [1] main
    for (auto &t : tr) t.emit(0, 0xC0, program, 0, 1);
    for (const Msg &m : rows) {
        TrackBuf &t = tr[m.track];
        if (m.kind == 2) {
            if (m.a < -8192 || m.a > 8191) return -1;
            const int v = m.a + 8192;
            t.emit(m.tick, 0xE0, v & 0x7F, v >> 7, 2);
        } else if (m.kind == 0) t.emit(m.tick, 0x90, m.a, m.b, 2);
        else t.emit(m.tick, 0x80, m.a, m.b, 2);
    }
    auto be32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); };
    auto be16 = [&](uint16_t v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    const size_t before = out.size();
    const char hd[4] = {'M', 'T', 'h', 'd'};
    out.insert(out.end(), hd, hd + 4);
    be32(6); be16(1); be16(2); be16(480);
    for (int q : {1, 0}) {                                              // main first, then safe
        const char tk[4] = {'M', 'T', 'r', 'k'};
        out.insert(out.end(), tk, tk + 4);
        be32((uint32_t)tr[q].data.size() + 4);
        out.insert(out.end(), tr[q].data.begin(), tr[q].data.end());
        const uint8_t eot[4] = {0x00, 0xFF, 0x2F, 0x00};
        out.insert(out.end(), eot, eot + 4);
    }
    return (int64_t)(out.size() - before);
}

int worker_count(int64_t n_items) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    return (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw, (int64_t)16, n_items / 8 + 1}));
}

}  // namespace

extern "C" {

const char *aegis_events_last_error(void) { return g_events_error.c_str(); }

int64_t aegis_extract_events(const aegis_event_params *P, const aegis_event_batch *B, aegis_event *events, int64_t cap,
                             int64_t *clip_event_off, aegis_run_fit *risky_runs, int64_t risky_cap, int64_t *n_risky) {
    try {
    if (!P || !B || B->n_clips < 0 || (B->n_clips > 0 && (!B->frame_off || !clip_event_off)) || B->n_fits < 0 || (B->n_fits > 0 && !B->fits)) {
        g_events_error = "null argument"; return AEGIS_ERR_INVALID;
    }
    const int n_clips = B->n_clips;
    const int64_t *frame_off = B->frame_off;
    if (P->sample_rate <= 0 || P->hop_length <= 0) { g_events_error = "bad sample_rate / hop_length"; return AEGIS_ERR_INVALID; }
    for (int c = 0; c < n_clips; ++c)
        if (frame_off[c + 1] < frame_off[c]) { g_events_error = "frame_off must be non-decreasing"; return AEGIS_ERR_INVALID; }
    if (n_clips > 0 && frame_off[n_clips] > frame_off[0] &&
        (!B->sounding || !B->rms_db || !B->probs || (!B->semitones && (!B->pitch_bin || !B->bin_semitones)))) {
        g_events_error = "null array"; return AEGIS_ERR_INVALID;
    }
    std::vector<std::vector<aegis_event>> per((size_t)n_clips);
    std::vector<std::vector<aegis_run_fit>> risky((size_t)n_clips);
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<int32_t> bin_pitch;                                  // rint(bin_semitones[b]) for the bins met (i16 index)
        if (!B->semitones) bin_pitch.assign(32768, INT32_MIN);
        for (;;) {
            const int c = next.fetch_add(1);
            if (c >= n_clips) break;
            const int64_t a = frame_off[c], F = frame_off[c + 1] - a;
            if (!clip_events(*P, F, B->sounding + a, B->semitones ? B->semitones + a : nullptr, B->pitch_bin ? B->pitch_bin + a : nullptr,
                             B->bin_semitones, bin_pitch.data(), B->rms_db + a, B->probs + a, c, B->fits, B->n_fits, per[c], risky[c]))
                per[c].clear();
        }
    };
    const int nt = worker_count(n_clips);
    if (nt <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    int64_t total = 0, nr = 0;
    if (n_clips > 0) clip_event_off[0] = 0;
    for (int c = 0; c < n_clips; ++c) {
        for (const aegis_event &e : per[c]) { if (events && total < cap) events[total] = e; ++total; }
        clip_event_off[c + 1] = total;
        for (const aegis_run_fit &r : risky[c]) { if (risky_runs && nr < risky_cap) risky_runs[nr] = r; ++nr; }
    }
    if (n_risky) *n_risky = nr;
    return total;
    } catch (const std::bad_alloc &) { g_events_error = "out of host memory"; return AEGIS_ERR_NOMEM; }
    catch (...) { g_events_error = "unexpected C++ exception"; return AEGIS_ERR_DEVICE; }
}

int64_t aegis_render_smf(int32_t sample_rate, int32_t hop_length, int32_t midi_program, double vibrato_rate, double vibrato_depth,
                         int32_t n_clips, const aegis_event *events, const int64_t *clip_event_off, uint8_t *out, int64_t cap,
                         int64_t *clip_byte_off) {
    try {
    if (n_clips < 0 || (n_clips > 0 && (!clip_event_off || !clip_byte_off)) || sample_rate <= 0 || hop_length <= 0) { g_events_error = "bad argument"; return AEGIS_ERR_INVALID; }
    std::vector<std::vector<uint8_t>> blobs((size_t)n_clips);
    std::atomic<int> next{0};
    std::atomic<int> bad{-1};
    auto work = [&]() {
        for (;;) {
            const int c = next.fetch_add(1);
            if (c >= n_clips) break;
            const int64_t a = clip_event_off[c], n = clip_event_off[c + 1] - a;
            if (render_clip(sample_rate, hop_length, midi_program, vibrato_rate, vibrato_depth, events ? events + a : nullptr, n, blobs[c]) < 0) bad = c;
        }
    };
    const int nt = worker_count(n_clips);
    if (nt <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    if (bad >= 0) { g_events_error = "pitchwheel out of range in clip " + std::to_string(bad.load()); return AEGIS_ERR_INVALID; }
    int64_t total = 0;
    if (n_clips > 0) clip_byte_off[0] = 0;
    for (int c = 0; c < n_clips; ++c) {
        const int64_t n = (int64_t)blobs[c].size();
        if (out && total + n <= cap) std::memcpy(out + total, blobs[c].data(), (size_t)n);
        total += n;
        clip_byte_off[c + 1] = total;
    }
    return total;
    } catch (const std::bad_alloc &) { g_events_error = "out of host memory"; return AEGIS_ERR_NOMEM; }
    catch (...) { g_events_error = "unexpected C++ exception"; return AEGIS_ERR_DEVICE; }
}

}  // extern "C"
