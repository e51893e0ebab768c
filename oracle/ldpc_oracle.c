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
