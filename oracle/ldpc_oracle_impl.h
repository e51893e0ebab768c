/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See ldpc_oracle.c for the header.
 *
 * Type-generic body of the CPU restatement.  Included twice by ldpc_oracle.c:
 *   REAL=float,  SFX(x)=x##_f32   (torch decoders: neural_2d_decoder.py, rcq_decoder.py)
 *   REAL=double, SFX(x)=x##_f64   (numpy decoder:  ldpc_decoder.py BasicMinSumDecoder)
 */

/* ---- torch.sum of a contiguous 1-D fp32 tensor (torch 2.10 CPU, ATen SumKernel.cpp:
 * row_sum / vectorized_inner_sum, 8-lane vectors, ILP 4).  This is what
 * `torch.sum(c2v_messages[other_neighbors, j])` (neural_2d_decoder.py:203,209;
 * rcq_decoder.py:257,263,575,581) evaluates.  Matched bit-for-bit against torch for
 * N = 0..575 (oracle/make_golden.py, tests/test_oracle.py); the cascade level that
 * starts at N >= 576 is not restated, callers reject such degrees. */
static REAL SFX(sum_torch)(const REAL *x, int N)
{
    if (N < 8) {
        REAL p[4] = {0, 0, 0, 0};
        int g = N / 4;
        for (int r = 0; r < g; ++r)
            for (int k = 0; k < 4; ++k) p[k] = p[k] + x[4 * r + k];
        for (int i = 4 * g; i < N; ++i) p[0] = p[0] + x[i];
        for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
        return p[0];
    }
    int V = N / 8, g = V / 4;
    REAL fin = 0;
    for (int k = 8 * V; k < N; ++k) fin = fin + x[k];
    REAL q[8];
    for (int l = 0; l < 8; ++l) {
        REAL p[4] = {0, 0, 0, 0};
        for (int r = 0; r < g; ++r)
            for (int k = 0; k < 4; ++k) p[k] = p[k] + x[(4 * r + k) * 8 + l];
        for (int v = 4 * g; v < V; ++v) p[0] = p[0] + x[v * 8 + l];
        for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
        q[l] = p[0];
    }
    for (int l = 0; l < 8; ++l) fin = fin + q[l];
    return fin;
}

/* ---- np.sum of a contiguous 1-D array (numpy 2.2 pairwise_sum, loops_utils.h.src):
 * what `np.sum(c2v_messages[other_neighbors, j])` (ldpc_decoder.py:131,137,150)
 * evaluates.  n<8: running sum from -0.0; n<=128: 8 accumulators; else split. */
static REAL SFX(pairwise_np)(const REAL *a, int n)
{
    if (n < 8) {
        REAL res = (REAL)-0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    } else if (n <= 128) {
        REAL r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        REAL res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return SFX(pairwise_np)(a, n2) + SFX(pairwise_np)(a + n2, n - n2);
    }
}
static REAL SFX(sum_numpy)(const REAL *a, int n)
{
    if (n == 0) return 0;           /* np.sum(empty) == +0.0 */
    return SFX(pairwise_np)(a, n);
}

static inline REAL SFX(sum_by_order)(int order, const REAL *x, int N)
{
    return order == ORACLE_SUM_NUMPY ? SFX(sum_numpy)(x, N) : SFX(sum_torch)(x, N);
}

static inline REAL SFX(sgn)(REAL x) { return x > 0 ? (REAL)1 : (x < 0 ? (REAL)-1 : (REAL)0); }

/* One codeword, flooding schedule.  Line references: the shared skeleton of
 * ldpc_decoder.py:75-153, neural_2d_decoder.py:145-225, rcq_decoder.py:190-279, 507-597. */
static void SFX(decode_one)(const oracle_graph *g, const oracle_params *p,
                            const REAL *beta, const REAL *alpha,
                            const REAL *llr, int32_t *bits, REAL *post,
                            int32_t *iters, uint8_t *success, uint8_t *code_trace,
                            REAL *v2c, REAL *c2v, REAL *tmp /* >= max(max_dc,max_dv)*3 */)
{
    const int n = g->n, m = g->m, E = g->E;
    const int T = p->iters;
    int max_deg = 1;
    for (int i = 0; i < m; ++i) { int d = g->check_ptr[i + 1] - g->check_ptr[i]; if (d > max_deg) max_deg = d; }
    for (int j = 0; j < n; ++j) { int d = g->var_ptr[j + 1] - g->var_ptr[j]; if (d > max_deg) max_deg = d; }
    REAL *sg = tmp, *mg = tmp + max_deg, *oth = tmp + 2 * max_deg;

    /* "Initialize with channel LLRs": v2c[j,i] = llr[j] on every edge; c2v = 0 */
    for (int e = 0; e < E; ++e) { v2c[e] = llr[g->var_idx[e]]; c2v[e] = 0; }

    int done_at = 0;
    for (int it = 0; it < T; ++it) {
        const REAL *bt = beta + (size_t)it * p->n_beta_slots;
        const REAL *at = alpha + (size_t)it * p->n_alpha_slots;
        const float *thr = p->c2v_form == ORACLE_C2V_RCQ ? p->thresholds + (size_t)p->q_of_iter[it] * p->n_levels : NULL;

        /* ---- check node update */
        for (int i = 0; i < m; ++i) {
            const int e0 = g->check_ptr[i], dc = g->check_ptr[i + 1] - e0;
            if (dc == 0) continue;
            for (int t = 0; t < dc; ++t) {
                REAL in = v2c[e0 + t];
                sg[t] = SFX(sgn)(in);
                mg[t] = in < 0 ? -in : in;       /* abs; -0 -> 0 either way compares equal */
                if (in == 0) mg[t] = 0;
            }
            int k = 0;                           /* argmin: first minimum */
            for (int t = 1; t < dc; ++t) if (mg[t] < mg[k]) k = t;
            REAL m1 = mg[k], m2 = m1;
            if (dc > 1) {
                m2 = (REAL)INFINITY;
                for (int t = 0; t < dc; ++t) if (t != k && mg[t] < m2) m2 = mg[t];
            }
            for (int t = 0; t < dc; ++t) {
                REAL prod = 1;
                for (int u = 0; u < dc; ++u) if (u != t) prod = prod * sg[u];
                REAL raw = (t == k) ? m2 : m1;
                REAL b = bt[p->beta_slot[e0 + t]];
                if (p->c2v_form == ORACLE_C2V_NMS) {
                    /* beta * min * prod(signs)   (ldpc_decoder.py:118-120, neural_2d_decoder.py:189-191) */
                    c2v[e0 + t] = (b * raw) * prod;
                } else if (p->c2v_form == ORACLE_C2V_OMS) {
                    /* prod(signs) * (relu(min - beta) - alpha_c)  (neural_2d_decoder.py:400-401) */
                    REAL a = p->oms_alpha ? ((const REAL *)p->oms_alpha)[(size_t)it * p->n_oms_alpha_slots + p->oms_alpha_slot[e0 + t]] : (REAL)0;
                    REAL d = raw - b;
                    REAL r = d > 0 ? d : (REAL)0;   /* F.relu */
                    c2v[e0 + t] = prod * (r - a);
                } else {
                    /* weighted = beta * prod * raw; quantize; dequantize
                     * (rcq_decoder.py:242-246 with beta == 1, 559-563) */
                    float w = (float)((b * prod) * raw);
                    float mag = fabsf(w);
                    int lvl = 0;
                    for (int q = 0; q < p->n_levels; ++q) if (mag >= thr[q]) lvl = q;   /* rcq_decoder.py:79-85 */
                    int sign_bit = (w < 0.0f) ? 1 : 0;                                  /* sign(x) < 0, :88 */
                    int code = sign_bit * p->n_levels + lvl;                            /* :89 */
                    if (code_trace) code_trace[(size_t)it * E + e0 + t] = (uint8_t)code;
                    /* dequantize :107-119 */
                    int sb = code >= p->n_levels;
                    float rec = thr[code % p->n_levels];
                    float s = 1.0f - 2.0f * (float)sb;
                    c2v[e0 + t] = (REAL)(s * rec);
                }
            }
        }

        /* ---- variable node update: v2c = llr + alpha * sum(c2v[others]) */
        for (int j = 0; j < n; ++j) {
            const int s0 = g->var_ptr[j], dv = g->var_ptr[j + 1] - s0;
            if (dv == 0) continue;
            REAL a = at[p->alpha_slot[j]];
            for (int kk = 0; kk < dv; ++kk) {
                int cnt = 0;
                for (int u = 0; u < dv; ++u) if (u != kk) oth[cnt++] = c2v[g->csc_edge[s0 + u]];
                REAL s = SFX(sum_by_order)(p->sum_order, oth, cnt);
                v2c[g->csc_edge[s0 + kk]] = llr[j] + a * s;
            }
        }

        /* ---- posterior (no alpha), hard decision, syndrome */
        int unsat = 0;
        for (int j = 0; j < n; ++j) {
            const int s0 = g->var_ptr[j], dv = g->var_ptr[j + 1] - s0;
            for (int u = 0; u < dv; ++u) oth[u] = c2v[g->csc_edge[s0 + u]];
            post[j] = llr[j] + SFX(sum_by_order)(p->sum_order, oth, dv);
            bits[j] = post[j] < 0 ? 1 : 0;
        }
        for (int i = 0; i < m; ++i) {
            int par = 0;
            for (int e = g->check_ptr[i]; e < g->check_ptr[i + 1]; ++e) par ^= bits[g->var_idx[e]];
            unsat += par;
        }
        if (unsat == 0 && !done_at) {
            done_at = it + 1;
            if (p->early_stop) { *iters = it + 1; *success = 1; return; }
        }
    }

    /* "Return final decision" (recomputed from the last c2v; equals the loop's value) */
    for (int j = 0; j < n; ++j) {
        const int s0 = g->var_ptr[j], dv = g->var_ptr[j + 1] - s0;
        for (int u = 0; u < dv; ++u) oth[u] = c2v[g->csc_edge[s0 + u]];
        post[j] = llr[j] + SFX(sum_by_order)(p->sum_order, oth, dv);
        bits[j] = post[j] < 0 ? 1 : 0;
    }
    *iters = T;
    if (p->early_stop) {
        *success = 0;
    } else {
        /* fixed-iteration extension: success = final syndrome is zero */
        int unsat = 0;
        for (int i = 0; i < m; ++i) {
            int par = 0;
            for (int e = g->check_ptr[i]; e < g->check_ptr[i + 1]; ++e) par ^= bits[g->var_idx[e]];
            unsat += par;
        }
        *success = unsat == 0;
    }
}

int SFX(oracle_decode)(const oracle_graph *g, const oracle_params *p,
                       const REAL *beta, const REAL *alpha,
                       const REAL *llr, int B,
                       int32_t *bits, REAL *post, int32_t *iters, uint8_t *success,
                       uint8_t *code_trace)
{
    if (p->c2v_form == ORACLE_C2V_RCQ && sizeof(REAL) != sizeof(float)) return -2;
    int max_deg = 1;
    for (int i = 0; i < g->m; ++i) { int d = g->check_ptr[i + 1] - g->check_ptr[i]; if (d > max_deg) max_deg = d; }
    for (int j = 0; j < g->n; ++j) {
        int d = g->var_ptr[j + 1] - g->var_ptr[j];
        if (d > max_deg) max_deg = d;
        if (p->sum_order == ORACLE_SUM_TORCH && d > 575) return -3;   /* cascade level not restated */
    }
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        REAL *v2c = (REAL *)malloc(sizeof(REAL) * (size_t)(g->E + 1));
        REAL *c2v = (REAL *)malloc(sizeof(REAL) * (size_t)(g->E + 1));
        REAL *tmp = (REAL *)malloc(sizeof(REAL) * (size_t)max_deg * 3);
        if (!v2c || !c2v || !tmp) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int b = 0; b < B; ++b) {
                SFX(decode_one)(g, p, beta, alpha, llr + (size_t)b * g->n, bits + (size_t)b * g->n,
                                post + (size_t)b * g->n, iters + b, success + b,
                                code_trace ? code_trace + (size_t)b * p->iters * g->E : NULL,
                                v2c, c2v, tmp);
            }
        }
        free(v2c); free(c2v); free(tmp);
    }
    return fail ? -1 : 0;
}

/* exported for the summation-order known-answer tests */
REAL SFX(oracle_sum)(int order, const REAL *x, int N) { return SFX(sum_by_order)(order, x, N); }
