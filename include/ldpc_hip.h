/*
 * ldpc_hip.h -- C ABI of the MI355X (gfx950) batched LDPC decode engine.
 *
 * The reference (Lalwaniamisha789/Implementation-of-Neural-LDPC-Decoders-...)
 * has no FFI or operator registry: its boundary is the Python class surface
 *     BasicMinSumDecoder.decode            ldpc_decoder.py:63-153
 *     Neural2DMinSumDecoder.forward        neural_2d_decoder.py:133-225
 *     Neural2DOffsetMinSumDecoder.forward  neural_2d_decoder.py:338-434
 *     RCQMinSumDecoder.decode              rcq_decoder.py:169-279
 *     WeightedRCQDecoder.forward           rcq_decoder.py:495-597
 * Every one of those is the same flooding loop with a different C2V rule and
 * weight lookup, so the native boundary is ONE decode entry point driven by a
 * descriptor; the Python classes of the same names (package directory) are thin
 * hosts over it.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C types only; "device" pointers are HIP device pointers on the
 *     device that was current when the graph was created;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it, nothing synchronises, nothing allocates: the
 *     caller provides the workspace (ldpc_decoder_workspace_bytes);
 *   - handles are immutable after creation and may be shared by threads; one
 *     workspace per concurrent ldpc_decode call (two calls that may overlap on the
 *     device -- different streams, different threads -- need two workspaces);
 *   - the library reads no environment variable; measurement and test hooks live in
 *     the separate ldpc_hip_debug.h and are never needed for decoding;
 *   - every function returns LDPC_OK (0) or a negative LDPC_ERR_*;
 *     ldpc_last_error() gives a thread-local message.
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_HIP_ABI_VERSION 1

enum {
    LDPC_OK = 0,
    LDPC_ERR_ARG = -1,          /* bad argument / inconsistent descriptor        */
    LDPC_ERR_HIP = -2,          /* a HIP runtime call failed                     */
    LDPC_ERR_UNSUPPORTED = -3,  /* valid request the engine does not implement   */
    LDPC_ERR_WORKSPACE = -4     /* workspace too small                           */
};

/* arithmetic type of messages, weights, LLRs and posteriors */
enum { LDPC_F32 = 0, LDPC_F64 = 1 };

/* check-to-variable rule; `min` is min1 (min2 on the arg-min edge), `s` the
 * product of the other edges' signs with sign(0) = 0 */
enum {
    LDPC_C2V_NMS = 0,  /* c2v = (beta * min) * s             ldpc_decoder.py:118-120, neural_2d_decoder.py:189-191 */
    LDPC_C2V_RCQ = 1,  /* c2v = deq(quant((beta * s) * min)) rcq_decoder.py:242-246, 559-563 (1-byte codes in HBM) */
    LDPC_C2V_OMS = 2   /* c2v = s * (relu(min - beta) - a_c) neural_2d_decoder.py:400-401                          */
};

/* message schedule.  LAYERED_REF is RCQMinSumDecoder(layered=True) exactly as the reference runs it
 * (rcq_decoder.py:281-350): checks processed in order on running posteriors whose "previous
 * message" is never subtracted (the reference re-creates its message matrix per check); RCQ fp32 only.
 * Two kernels with identical results: LDS-resident (posteriors of a few codewords per one-wave workgroup in LDS, the lanes
 * on the edges of the current check; LDPC_MODE_AUTO / RESIDENT when a posterior vector fits LDS and no check has more than
 * 64 edges) and streaming (posteriors in HBM, a lane per codeword; LDPC_MODE_STREAM and every other case).
 * LAYERED is the schedule that code sets out to implement (and the RCQ paper defines): the check's previous
 * message IS subtracted before its update and the new one added -- an extension with no reference execution
 * to compare against (parity unpinned; checked against an independent CPU restatement only). */
enum { LDPC_SCHED_FLOODING = 0, LDPC_SCHED_LAYERED_REF = 1, LDPC_SCHED_LAYERED = 2 };

typedef struct ldpc_graph ldpc_graph;      /* Tanner graph, CSR + CSC, device resident */
typedef struct ldpc_decoder ldpc_decoder;  /* graph + weight tables + quantiser LUTs    */

/* Replaces the reference's per-node dense scans `np.where(H[i,:]==1)` /
 * `np.where(H[:,j]==1)` (ldpc_decoder.py:92,124).  Input is the CSR edge list
 * (host memory): check i owns edges check_ptr[i]..check_ptr[i+1]-1, var_idx[e]
 * ascending inside a check.  The CSC permutation is derived inside. */
int ldpc_graph_create(ldpc_graph **out, int32_t n, int32_t m, int32_t n_edges,
                      const int32_t *check_ptr, const int32_t *var_idx);
void ldpc_graph_destroy(ldpc_graph *g);
/* n, m, E, max check degree, max variable degree */
int ldpc_graph_info(const ldpc_graph *g, int32_t out5[5]);

/* All pointers are HOST pointers, copied at creation.  Weight tables are
 * dtype-typed ([iters][slots], row t = iteration t); slot arrays say which
 * table column an edge / a variable uses -- the flattened form of
 * _get_beta_weight/_get_alpha_weight (neural_2d_decoder.py:84-131). */
typedef struct {
    int32_t dtype;              /* LDPC_F32 | LDPC_F64                                     */
    int32_t c2v_form;           /* LDPC_C2V_*                                              */
    int32_t iters;              /* T = max_iterations                                      */
    int32_t n_beta_slots;
    const void *beta;           /* [T][n_beta_slots]                                       */
    const int32_t *beta_slot;   /* [E] per CSR edge                                        */
    int32_t n_alpha_slots;
    const void *alpha;          /* [T][n_alpha_slots]  V2C weight: llr + alpha * sum       */
    const int32_t *alpha_slot;  /* [n] per variable                                        */
    /* RCQ only: NonUniformQuantizer tables (rcq_decoder.py:48-57) as float32(tau)  */
    int32_t n_levels;           /* 2^(bc-1), <= 128                                        */
    int32_t n_quantizers;
    const float *thresholds;    /* [n_quantizers][n_levels]                                */
    const int32_t *q_of_iter;   /* [T] quantiser used in iteration t (rcq_decoder.py:156-167) */
    /* OMS only: check-side offset alpha (neural_2d_decoder.py:392,400)              */
    int32_t n_oms_alpha_slots;
    const void *oms_alpha;      /* [T][n_oms_alpha_slots] or NULL (= 0)                    */
    const int32_t *oms_alpha_slot; /* [E]                                                  */
    int32_t schedule;           /* LDPC_SCHED_FLOODING (0) | LDPC_SCHED_LAYERED_REF | LDPC_SCHED_LAYERED */
} ldpc_decoder_desc;

int ldpc_decoder_create(ldpc_decoder **out, const ldpc_graph *g, const ldpc_decoder_desc *desc);

/* Two engines implement the same arithmetic (bit-identical results):
 *   STREAM   : messages in HBM ([tile][edge][W]); any code, fp32/fp64.  One kernel per sweep (check sweep,
 *              variable sweep).  fp32 flooding RCQ decoders have two cheaper forms, STREAM takes the first that applies:
 *                PAIR   -- one beta per check, sorted thresholds, <= 62 levels: both message directions are 1-byte
 *                          codes (the variable sweep quantises with the next iteration's beta and thresholds, the
 *                          check sweep is integer-only): 4E + 4n bytes per codeword and iteration;
 *                GATHER -- variable degree <= 8: ONE fused kernel per iteration that recomputes the variable->check
 *                          messages from the 1-byte check->variable codes and the LLRs (no V2C array);
 *              SWEEPS forces the plain two-sweep form (fp32 V2C rows); GATHER / PAIR force that form (error when the
 *              decoder does not qualify).
 *              On fp32 256-codeword tiles the last variable pass writes the caller's posterior / decision rows itself, and the
 *              PAIR form's entrance pass also codes the LLRs for iteration 0 (no separate layout passes at either end).
 *   RESIDENT : one fused kernel, messages in LDS for all T iterations; fp32 codes with
 *              dv <= 8 whose state fits 160 KiB of LDS (e.g. the (1998,1512) code); for the layered schedule: the
 *              LDS-resident layered kernel (see LDPC_SCHED_LAYERED_REF)
 * AUTO (default) takes RESIDENT when the code qualifies, else STREAM. */
enum { LDPC_MODE_AUTO = 0, LDPC_MODE_STREAM = 1, LDPC_MODE_RESIDENT = 2, LDPC_MODE_SWEEPS = 3, LDPC_MODE_GATHER = 4,
       LDPC_MODE_PAIR = 5 };
int ldpc_decoder_set_mode(ldpc_decoder *d, int32_t mode);
/* out4 = { engine and form a decode would use now (LDPC_MODE_RESIDENT, or the streaming form LDPC_MODE_PAIR /
 * LDPC_MODE_GATHER / LDPC_MODE_SWEEPS), codewords per workgroup, threads per
 * workgroup, LDS bytes per workgroup } -- the last three 0 when the code does not qualify */
int ldpc_decoder_info(const ldpc_decoder *d, int32_t out4[4]);
/* re-upload beta/alpha(/oms_alpha) tables of an existing decoder (same shapes);
 * enqueued on `stream`, host arrays must stay valid until it has run. */
int ldpc_decoder_set_weights(ldpc_decoder *d, const void *beta, const void *alpha,
                             const void *oms_alpha, void *stream);
void ldpc_decoder_destroy(ldpc_decoder *d);

/* bytes of device scratch ldpc_decode needs for a batch of `batch` codewords */
size_t ldpc_decoder_workspace_bytes(const ldpc_decoder *d, int64_t batch);

/* Decode llr[batch][n] (device, row-major, dtype of the decoder).
 *   early_stop != 0 : reference semantics per codeword -- outputs are those of the
 *                     first iteration whose syndrome is zero (iterations 1-based,
 *                     success 1), else of iteration T (success 0);
 *   early_stop == 0 : exactly T iterations; success = final syndrome is zero.
 * Outputs (device, any may be NULL):
 *   bits[batch][n] int32 (posterior < 0), posterior[batch][n] dtype,
 *   iterations[batch] int32, success[batch] uint8,
 *   packed_bits[batch][ceil(n/8)] uint8, bit j of a codeword at byte j/8, bit j%8
 *   (wire format of the multi-GPU all-gather). */
int ldpc_decode(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop,
                int32_t *bits, void *posterior, int32_t *iterations, uint8_t *success,
                uint8_t *packed_bits, void *workspace, size_t workspace_bytes, void *stream);

/* ldpc_decode with at most `max_iterations` (>= 1) of the decoder's iterations: iteration t still uses the decoder's own
 * tables of iteration t (weights, quantiser schedule), a codeword open at the cap reports iterations = cap, success = 0.
 * No reference counterpart: it lets a batched caller with early stop (the Monte-Carlo driver, simulation_framework.py:85-139
 * of the reference run block-wise) decode a block up to the iteration by which most codewords have stopped and finish the few
 * stragglers as a small second batch -- the streaming engine freezes whole 256-codeword tiles only. */
int ldpc_decode_capped(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop,
                       int32_t max_iterations, int32_t *bits, void *posterior, int32_t *iterations,
                       uint8_t *success, uint8_t *packed_bits, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---- gradient (training) path -------------------------------------------------------------
 * Replaces torch autograd through Neural2DMinSumDecoder.forward / NeuralMinSumDecoder.forward
 * (neural_2d_decoder.py:133-225 under loss.backward(), training_framework.py:127-134): the
 * derivative of any loss of the returned posterior with respect to the beta / alpha tables.
 * fp32 LDPC_C2V_NMS and LDPC_C2V_OMS flooding decoders (LDPC_ERR_UNSUPPORTED otherwise; the reference's
 * RCQ quantiser passes no gradient).  Always runs the streaming engine.
 *
 * ldpc_decode_saving : ldpc_decode (same outputs, same arithmetic) that also keeps every
 *   iteration's message rows in `saved` (ldpc_train_saved_bytes: (2T-1) * E * 4 bytes per
 *   codeword, tile-padded).
 * ldpc_backward      : grad_posterior[batch][n] fp32 (d loss / d posterior) and the
 *   iterations[batch] that decode returned -> grad_beta[T][n_beta_slots], grad_alpha[T][n_alpha_slots]
 *   fp32 and, for LDPC_C2V_OMS decoders created with oms_alpha, grad_oms_alpha[T][n_oms_alpha_slots]
 *   and grad_llr[batch][n] fp32 (d loss / d llr, for callers that train what produces the LLRs)
 *   (device, overwritten; any may be NULL).  The decoder's tables must be the ones the
 *   forward call used.  Slots autograd would leave without a gradient come back as 0.
 * Both need ldpc_train_workspace_bytes of 256-byte aligned scratch; `saved` is 256-byte aligned. */
size_t ldpc_train_saved_bytes(const ldpc_decoder *d, int64_t batch);
size_t ldpc_train_workspace_bytes(const ldpc_decoder *d, int64_t batch);
int ldpc_decode_saving(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop,
                       int32_t *bits, void *posterior, int32_t *iterations, uint8_t *success,
                       void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                       void *stream);
int ldpc_backward(const ldpc_decoder *d, const void *saved, size_t saved_bytes, const void *llr,
                  int64_t batch, const int32_t *iterations, const void *grad_posterior,
                  void *grad_beta, void *grad_alpha, void *grad_oms_alpha, void *grad_llr,
                  void *workspace, size_t workspace_bytes, void *stream);

const char *ldpc_last_error(void);
int ldpc_abi_version(void);
/* sha256 (hex) over the sources and the compile recipe this library was built from, embedded at build time
 * (-DLDPC_SRC_HASH=...; "unknown" for a hand build).  The Python loader compares it with the sources on disk and
 * refuses a library that does not match -- file times say nothing after a copy to another machine. */
const char *ldpc_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
