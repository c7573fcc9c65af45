/* Plain-C caller of libaegis_hip.so: no Python, no torch.  Synthesises a 2 s, 220 Hz tone (A3), analyses it through
 * the C ABI of include/aegis_hip.h and prints the pitch track's statistics.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/aegis_demo.c -o /tmp/aegis_demo -Lspectrogram-midi_amd -l:libaegis_hip.so \
 *       -Wl,-rpath,$PWD/spectrogram-midi_amd -lm
 *   /tmp/aegis_demo            (needs an MI355X; exits 0 when the tone is found on the pitch grid)
 *   -> abi 1: 173 frames, 172 voiced, 171 on the 220.00 Hz grid point, rms[F/2] = 0.354646 (expect 0.353553)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aegis_hip.h"

int main(void) {
    const int sr = 44100;
    const int64_t n = 2 * sr;
    float *pcm = (float *)malloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i) pcm[i] = 0.5f * (float)sin(2.0 * 3.14159265358979323846 * 220.0 * (double)i / sr);

    aegis_config cfg;
    memset(&cfg, 0, sizeof cfg);                 /* zero fields take the reference's defaults */
    cfg.sample_rate = sr; cfg.hop_length = 512; cfg.n_fft = 2048; cfg.n_mels = 128; cfg.device = 0;
    aegis_handle *h = NULL;
    int rc = aegis_create(&cfg, &h);
    if (rc != AEGIS_OK) { fprintf(stderr, "aegis_create: %d %s\n", rc, aegis_last_error(NULL)); return 2; }

    const int64_t F = aegis_frames_for(h, n);
    double *f0 = (double *)malloc((size_t)F * sizeof(double)), *vp = (double *)malloc((size_t)F * sizeof(double));
    uint8_t *voiced = (uint8_t *)malloc((size_t)F), *rake = (uint8_t *)malloc((size_t)F);
    float *rms = (float *)malloc((size_t)F * sizeof(float));
    aegis_outputs out;
    memset(&out, 0, sizeof out);
    out.f0 = f0; out.voiced_flag = voiced; out.voiced_prob = vp; out.rms = rms; out.rake_mask = rake;   /* S_dB skipped */

    const float *clips[1] = {pcm};
    const int64_t lens[1] = {n};
    rc = aegis_analyze_batch(h, clips, lens, 1, 0.6, AEGIS_STAGE_ALL, &out);
    if (rc != AEGIS_OK) { fprintf(stderr, "aegis_analyze_batch: %d %s\n", rc, aegis_last_error(h)); return 3; }

    int64_t nv = 0, on_grid = 0;
    const double want = 82.4068892282175 * pow(2.0, 170.0 / 120.0);     /* A3 = E2 + 17 semitones = pitch bin 170 */
    for (int64_t t = 0; t < F; ++t)
        if (voiced[t]) { ++nv; if (fabs(f0[t] - want) < 1e-9 * want) ++on_grid; }
    printf("abi %d: %lld frames, %lld voiced, %lld on the 220.00 Hz grid point, rms[F/2] = %.6f (expect %.6f)\n",
           aegis_abi_version(), (long long)F, (long long)nv, (long long)on_grid, rms[F / 2], 0.5 / sqrt(2.0));
    aegis_destroy(h);
    free(pcm); free(f0); free(vp); free(voiced); free(rake); free(rms);
    /* the frames at the clip edges see half a window of padding; a 2048-sample frame holds a non-integer number of periods */
    return (nv > F / 2 && on_grid >= nv - 2 && fabs(rms[F / 2] - 0.5 / sqrt(2.0)) < 2e-3) ? 0 : 1;
}
