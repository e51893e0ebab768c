/*
 * ldpc_oracle.c -- CPU restatement of the reference's decode loops.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (the HIP engine behind include/ldpc_hip.h) never calls into it.
 *
 * What it restates (all paths relative to /root/reference):
 *   ldpc_decoder.py:63-153     BasicMinSumDecoder.decode          (fp64, np.sum order)
 *   neural_2d_decoder.py:133-225  Neural2DMinSumDecoder.forward   (fp32, torch.sum order)
 *   neural_2d_decoder.py:338-434  Neural2DOffsetMinSumDecoder.forward (fp32, OMS form)
 *   rcq_decoder.py:59-121      NonUniformQuantizer.quantize / dequantize
 *   rcq_decoder.py:190-279     RCQMinSumDecoder._decode_flooding
 *   rcq_decoder.py:495-597     WeightedRCQDecoder.forward
 * The loops are written the way the reference writes them (per check, per edge,
 * explicit product of the other signs with sign(0) = 0, explicit leave-one-out
 * sums) and NOT the way the HIP kernels compute them, so that a parity test of
 * kernel vs oracle also tests the kernels' algebraic shortcuts.
 *
 * Third-party arithmetic restated here: torch.sum (fp32, torch 2.10.0 CPU) and
 * np.sum (fp64, numpy 2.2.6) association orders -- see ldpc_oracle_impl.h.
 *
 * Parity pinning: oracle/make_golden.py imports the real reference in the build
 * container, runs it on seeded inputs, checks this library against it and writes
 * tests/golden/\*.npz; tests/test_oracle.py re-checks the library against those
 * fixtures everywhere.  Status: PINNED (see DESIGN.md "Oracle").
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_C2V_NMS 0   /* (beta * min) * prod(signs)                      */
#define ORACLE_C2V_RCQ 1   /* deq(quant((beta * prod(signs)) * min))          */
#define ORACLE_C2V_OMS 2   /* prod(signs) * (relu(min - beta) - alpha_c)      */
#define ORACLE_SUM_TORCH 0
#define ORACLE_SUM_NUMPY 1

typedef struct {
    int32_t n, m, E;
    const int32_t *check_ptr;   /* [m+1]                                   */
    const int32_t *var_idx;     /* [E]  CSR order: variable of each edge   */
    const int32_t *var_ptr;     /* [n+1]                                   */
    const int32_t *csc_edge;    /* [E]  CSR edge id of k-th edge of var j  */
} oracle_graph;

typedef struct {
    int32_t iters;              /* T                                        */
    int32_t early_stop;         /* 1 = reference behaviour                  */
    int32_t c2v_form;           /* ORACLE_C2V_*                             */
    int32_t sum_order;          /* ORACLE_SUM_*                             */
    int32_t n_beta_slots;       /* beta table is [T][n_beta_slots]          */
    int32_t n_alpha_slots;      /* alpha table is [T][n_alpha_slots]        */
    const int32_t *beta_slot;   /* [E] slot of each CSR edge                */
    const int32_t *alpha_slot;  /* [n] slot of each variable                */
    int32_t n_levels;           /* 2^(bc-1) thresholds per quantiser        */
    int32_t n_quantizers;
    const float *thresholds;    /* [Q][n_levels] float32(tau)               */
    const int32_t *q_of_iter;   /* [T] quantiser index per iteration        */
    int32_t n_oms_alpha_slots;  /* OMS only: check-side alpha [T][slots]    */
    const int32_t *oms_alpha_slot; /* [E]                                   */
    const void *oms_alpha;      /* REAL[T][n_oms_alpha_slots] or NULL       */
} oracle_params;

#define REAL float
#define SFX(x) x##_f32
#include "ldpc_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX(x) x##_f64
#include "ldpc_oracle_impl.h"
#undef REAL
#undef SFX

/* NonUniformQuantizer.quantize (rcq_decoder.py:59-91) on a float32 vector;
 * thresholds are float32(tau_j): torch compares an fp32 tensor with a Python
 * scalar in fp32. */
void oracle_quantize_f32(const float *x, int N, const float *thr, int n_levels, int64_t *codes)
{
    for (int i = 0; i < N; ++i) {
        float mag = fabsf(x[i]);
        int lvl = 0;
        for (int q = 0; q < n_levels; ++q) if (mag >= thr[q]) lvl = q;
        int sign_bit = x[i] < 0.0f;
        codes[i] = (int64_t)sign_bit * n_levels + lvl;
    }
}

/* NonUniformQuantizer.dequantize (rcq_decoder.py:93-121) */
void oracle_dequantize_f32(const int64_t *codes, int N, const float *thr, int n_levels, float *out)
{
    for (int i = 0; i < N; ++i) {
        int sb = codes[i] >= n_levels;
        int64_t idx = codes[i] % n_levels;      /* Python % on non-negative codes */
        float mag = (idx >= 0 && idx < n_levels) ? thr[idx] : 0.0f;
        out[i] = (1.0f - 2.0f * (float)sb) * mag;
    }
}

/* RCQMinSumDecoder._decode_layered (rcq_decoder.py:281-350), restated literally -- including the
 * reference's quirk: `c2v_messages` is re-created (all zero) for every check (:323), so the "subtract
 * previous C2V messages" step (:300-302) only ever finds the row of the check processed immediately
 * before; for any code with more than one check that row is a different check and the subtraction is
 * of zeros: posteriors simply accumulate every check's quantised message, check after check.
 * One codeword: llr[n] -> bits[n], *iters, *success. */
static void layered_one(const oracle_graph *g, int T, const float *thresholds, int n_levels, const int32_t *q_of_iter,
                        const float *llr, int32_t *bits, float *post, int32_t *iters, uint8_t *success,
                        float *sg, float *mg, float *last_vals)
{
    const int n = g->n, m = g->m;
    for (int j = 0; j < n; ++j) post[j] = llr[j];
    int last_i = -1, last_dc = 0;
    for (int it = 0; it < T; ++it) {
        const float *thr = thresholds + (size_t)q_of_iter[it] * n_levels;
        for (int i = 0; i < m; ++i) {
            const int e0 = g->check_ptr[i], dc = g->check_ptr[i + 1] - e0;
            if (dc == 0) continue;
            if (i == last_i)            /* the only row c2v_messages still holds */
                for (int t = 0; t < dc && t < last_dc; ++t) post[g->var_idx[e0 + t]] -= last_vals[t];
            for (int t = 0; t < dc; ++t) {
                float in = post[g->var_idx[e0 + t]];
                sg[t] = in > 0 ? 1.0f : (in < 0 ? -1.0f : 0.0f);
                mg[t] = fabsf(in);
            }
            int k = 0;
            for (int t = 1; t < dc; ++t) if (mg[t] < mg[k]) k = t;
            float m1 = mg[k], m2 = m1;
            if (dc > 1) {
                m2 = INFINITY;
                for (int t = 0; t < dc; ++t) if (t != k && mg[t] < m2) m2 = mg[t];
            }
            for (int t = 0; t < dc; ++t) {
                float prod = 1.0f;
                for (int u = 0; u < dc; ++u) if (u != t) prod = prod * sg[u];
                float w = prod * ((t == k) ? m2 : m1);                 /* :331 */
                float mag = fabsf(w);
                int lvl = 0;
                for (int q = 0; q < n_levels; ++q) if (mag >= thr[q]) lvl = q;
                int sb = w < 0.0f;
                last_vals[t] = (1.0f - 2.0f * (float)sb) * thr[lvl];   /* dequantize(quantize(w)) */
            }
            last_i = i; last_dc = dc;
            for (int t = 0; t < dc; ++t) post[g->var_idx[e0 + t]] += last_vals[t];   /* :338-339 */
        }
        int unsat = 0;
        for (int j = 0; j < n; ++j) bits[j] = post[j] < 0 ? 1 : 0;
        for (int i = 0; i < m; ++i) {
            int par = 0;
            for (int e = g->check_ptr[i]; e < g->check_ptr[i + 1]; ++e) par ^= bits[g->var_idx[e]];
            unsat += par;
        }
        if (unsat == 0) { *iters = it + 1; *success = 1; return; }
    }
    for (int j = 0; j < n; ++j) bits[j] = post[j] < 0 ? 1 : 0;
    *iters = T; *success = 0;
}

int oracle_decode_layered_f32(const oracle_graph *g, int T, const float *thresholds, int n_levels,
                              const int32_t *q_of_iter, const float *llr, int B,
                              int32_t *bits, float *post, int32_t *iters, uint8_t *success)
{
    int max_dc = 1;
    for (int i = 0; i < g->m; ++i) { int d = g->check_ptr[i + 1] - g->check_ptr[i]; if (d > max_dc) max_dc = d; }
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)max_dc * 3);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < B; ++b)
            layered_one(g, T, thresholds, n_levels, q_of_iter, llr + (size_t)b * g->n, bits + (size_t)b * g->n,
                        post + (size_t)b * g->n, iters + b, success + b, tmp, tmp + max_dc, tmp + 2 * max_dc);
        free(tmp);
    }
    return 0;
}

/* The layered schedule RCQMinSumDecoder._decode_layered sets out to implement (rcq_decoder.py:281-350 with ONE change: the
 * message matrix c2v_messages is created once, before the loops, instead of inside the check loop at :323), i.e. the layered
 * RCQ decoder of the paper: per check, subtract its previous (dequantised) messages from the posteriors, run the check
 * update on them, store and add the new messages.  PARITY UNPINNED: nothing in the reference executes this. */
static void layered_paper_one(const oracle_graph *g, int T, const float *thresholds, int n_levels, const int32_t *q_of_iter,
                              const float *llr, int32_t *bits, float *post, int32_t *iters, uint8_t *success,
                              float *sg, float *mg, float *c2v)
{
    const int n = g->n, m = g->m;
    for (int j = 0; j < n; ++j) post[j] = llr[j];
    for (int e = 0; e < g->check_ptr[m]; ++e) c2v[e] = 0.0f;
    for (int it = 0; it < T; ++it) {
        const float *thr = thresholds + (size_t)q_of_iter[it] * n_levels;
        for (int i = 0; i < m; ++i) {
            const int e0 = g->check_ptr[i], dc = g->check_ptr[i + 1] - e0;
            if (dc == 0) continue;
            for (int t = 0; t < dc; ++t) post[g->var_idx[e0 + t]] -= c2v[e0 + t];          /* :300-302 */
            for (int t = 0; t < dc; ++t) {
                float in = post[g->var_idx[e0 + t]];
                sg[t] = in > 0 ? 1.0f : (in < 0 ? -1.0f : 0.0f);
                mg[t] = fabsf(in);
            }
            int k = 0;
            for (int t = 1; t < dc; ++t) if (mg[t] < mg[k]) k = t;
            float m1 = mg[k], m2 = m1;
            if (dc > 1) {
                m2 = INFINITY;
                for (int t = 0; t < dc; ++t) if (t != k && mg[t] < m2) m2 = mg[t];
            }
            for (int t = 0; t < dc; ++t) {
                float prod = 1.0f;
                for (int u = 0; u < dc; ++u) if (u != t) prod = prod * sg[u];
                float w = prod * ((t == k) ? m2 : m1);
                float mag = fabsf(w);
                int lvl = 0;
                for (int q = 0; q < n_levels; ++q) if (mag >= thr[q]) lvl = q;
                int sb = w < 0.0f;
                c2v[e0 + t] = (1.0f - 2.0f * (float)sb) * thr[lvl];
            }
            for (int t = 0; t < dc; ++t) post[g->var_idx[e0 + t]] += c2v[e0 + t];          /* :338-339 */
        }
        int unsat = 0;
        for (int j = 0; j < n; ++j) bits[j] = post[j] < 0 ? 1 : 0;
        for (int i = 0; i < m; ++i) {
            int par = 0;
            for (int e = g->check_ptr[i]; e < g->check_ptr[i + 1]; ++e) par ^= bits[g->var_idx[e]];
            unsat += par;
        }
        if (unsat == 0) { *iters = it + 1; *success = 1; return; }
    }
    for (int j = 0; j < n; ++j) bits[j] = post[j] < 0 ? 1 : 0;
    *iters = T; *success = 0;
}

int oracle_decode_layered_paper_f32(const oracle_graph *g, int T, const float *thresholds, int n_levels,
                                    const int32_t *q_of_iter, const float *llr, int B,
                                    int32_t *bits, float *post, int32_t *iters, uint8_t *success)
{
    int max_dc = 1;
    const int E = g->check_ptr[g->m];
    for (int i = 0; i < g->m; ++i) { int d = g->check_ptr[i + 1] - g->check_ptr[i]; if (d > max_dc) max_dc = d; }
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float *tmp = (float *)malloc(sizeof(float) * ((size_t)max_dc * 2 + (size_t)(E > 0 ? E : 1)));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < B; ++b)
            layered_paper_one(g, T, thresholds, n_levels, q_of_iter, llr + (size_t)b * g->n, bits + (size_t)b * g->n,
                              post + (size_t)b * g->n, iters + b, success + b, tmp, tmp + max_dc, tmp + 2 * max_dc);
        free(tmp);
    }
    return 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
