/*
 * ldpc_hip_debug.h -- measurement and test hooks of libldpc_hip.so.
 *
 * NOT part of the product ABI (include/ldpc_hip.h): nothing here is needed to decode, and no
 * reference interface corresponds to it.  bench.py uses ldpc_debug_sweep to time one sweep kernel
 * with HIP events; the parity tests use the two state dumps to compare per-edge check-to-variable
 * messages (for RCQ: the 3-bit quantiser codes the reference emits, rcq_decoder.py:244-246) on BOTH
 * engines.
 */
#ifndef LDPC_HIP_DEBUG_H
#define LDPC_HIP_DEBUG_H

#include "ldpc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Run ONE sweep of iteration `iter` on the state left in `workspace` by a previous
 * ldpc_decode (streaming engine) of the same batch: which = 0 check-node (CN->VN) sweep,
 * 1 variable-node sweep. */
int ldpc_debug_sweep(const ldpc_decoder *d, int64_t batch, int32_t which, int32_t iter,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Byte offsets of the streaming engine's state arrays inside a workspace for `batch`
 * codewords: out8 = { VEC, tiles, llrT, v2c, c2v, postT, bitsT, done }.  Messages are laid out
 * [tile][edge][W] with W = 64*VEC codewords innermost. */
int ldpc_debug_workspace_layout(const ldpc_decoder *d, int64_t batch, int64_t out8[8]);

/* LDS-resident engine: decode llr[batch][n] (posterior[batch][n] and iterations[batch] out, as
 * ldpc_decode) and ALSO copy every codeword's check-to-variable messages of its last executed
 * iteration out of LDS into c2v_out[batch][E] (decoder dtype, CSR edge order; RCQ decoders hold the
 * reconstructed values (1 - 2*sign) * tau[level], from which the test recovers the codes). */
int ldpc_debug_resident_c2v(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop,
                            void *posterior, int32_t *iterations, void *c2v_out, void *stream);

/* The variable sweep of the RCQ code-pair form turns every outgoing value v into the key
 * [m > 0] + [m >= t1] + [m >= t2] + [m >= t3] of m = |beta * v| (thresholds4[0] is not used; device pointers, thresholds
 * within [2^-50, 2^50]).  Runs BOTH device forms of that key on `count` arbitrary values: the float form the 4-level
 * kernels use (clamped differences, csrc/ldpc_kernels.hip key_pair4) and the integer compare chain it replaced. */
int ldpc_debug_key4(const float *values, int64_t count, float beta, const float thresholds4[4], uint8_t *keys_float,
                    uint8_t *keys_compare, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_DEBUG_H */
