/* Oracle (TEST INFRASTRUCTURE): dense log-domain Viterbi, a C restatement of
 * librosa 0.10 `sequence.py::_viterbi` (the numba kernel behind librosa.pyin,
 * which the reference calls at /root/reference/aegis_engine.py:63,67).
 * value[t][j] = log_prob[t][j] + max_k(value[t-1][k] + log_trans[k][j]),
 * argmax = first maximum, backtrack from argmax(value[T-1]).
 * Same float64 operations in the same order as the NumPy fallback in
 * oracle/pyin.py::viterbi_states, so both produce identical states. */
#include <stdint.h>
#include <stdlib.h>

void oracle_viterbi_dense(const double *log_prob, const double *log_trans,
                          const double *log_p_init, int64_t T, int64_t S,
                          int32_t *state)
{
    double *value = (double *)malloc(sizeof(double) * S * 2);
    uint16_t *ptr = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)T * S);
    double *ltT = (double *)malloc(sizeof(double) * S * S);
    double *cur = value, *nxt = value + S;
    for (int64_t k = 0; k < S; ++k)
        for (int64_t j = 0; j < S; ++j)
            ltT[j * S + k] = log_trans[k * S + j];
    for (int64_t j = 0; j < S; ++j)
        cur[j] = log_prob[j] + log_p_init[j];
    for (int64_t t = 1; t < T; ++t) {
        const double *lp = log_prob + t * S;
        uint16_t *pt = ptr + t * S;
        for (int64_t j = 0; j < S; ++j) {
            const double *row = ltT + j * S;
            double best = cur[0] + row[0];
            int64_t bi = 0;
            for (int64_t k = 1; k < S; ++k) {
                double v = cur[k] + row[k];
                if (v > best) { best = v; bi = k; }
            }
            pt[j] = (uint16_t)bi;
            nxt[j] = lp[j] + best;
        }
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    int64_t s = 0;
    for (int64_t j = 1; j < S; ++j)
        if (cur[j] > cur[s]) s = j;
    state[T - 1] = (int32_t)s;
    for (int64_t t = T - 2; t >= 0; --t) {
        s = ptr[(t + 1) * S + s];
        state[t] = (int32_t)s;
    }
    free(value); free(ptr); free(ltT);
}
