// ldpc_resident.hip -- LDS-resident fused decoder (gfx950): all T iterations of G codewords
// run inside ONE workgroup with every message in LDS; HBM sees only the LLRs in and the
// decisions out (~16 KB per codeword instead of T*(16E+4n) ~ 1.15 MB for the (1998,1512) code).
//
// Mapping (the transpose of the streaming kernels): lanes run over the NODES of the graph,
// the G codewords of the workgroup ride along as a G-wide vector per lane (ds_read/write_b64
// for G = 2, b128 for G = 4).
//   msg  [S][G]  fp32   one slot per edge, "ELL-transposed" check-major: slot(p,t) = t*m + p
//                       (p = position of the check after sorting by degree) -> consecutive
//                       lanes hit consecutive LDS words in the check phase (conflict-free)
//   llr_s[n][G]  fp32   channel LLRs in degree-sorted variable order
//   bits_s[n]    u8     hard decisions of the G codewords (bit g), for the syndrome
// One array holds both message directions in place: the check phase turns v2c into c2v slot by
// slot, the variable phase gathers its dv slots (index list vslot), forms the leave-one-out sums
// in the reference's association order and scatters v2c back.  Nodes are sorted by degree so a
// wave runs one compile-time body.  Arithmetic is the streaming kernels' (same helpers), so the
// two engines are bit-identical (the GPU parity tests run every case on both).
//
// Early stop (reference semantics): every iteration a posterior pass + syndrome pass finds the
// codewords that just converged; their posterior is recomputed into their (now dead) LLR slots
// and written out at once; the workgroup leaves when all G are done.
#pragma once

#include "ldpc_kernels.hip"

namespace ldpc {

struct ResidentPlan {
    int n, m, S, max_dc, max_dv;
    const uint8_t *dc_s;          // [m]          degree of the check at sorted position p
    const uint16_t *cvar;         // [max_dc*m]   sorted position of the variable of edge (p,t)
    const uint16_t *bslot;        // [max_dc*m]   beta table column of edge (p,t)
    const uint16_t *bslot_c;      // [m] or null: column shared by all edges of check p (Basic, RCQ,
                                  //              sharing types 2-4) -> one table read per check
    const uint16_t *oaslot;       // [max_dc*m]   OMS alpha column (or null)
    const uint8_t *dv_s;          // [n]          degree of the variable at sorted position q
    const uint16_t *vslot;        // [max_dv*n]   message slot of the k-th (ascending check) edge of q
    const uint16_t *aslot;        // [n]          alpha table column of q
    const uint16_t *inv_perm_v;   // [n]          sorted position of original variable j
};

struct ResidentArgs {
    const float *llr;             // [batch][n]
    long long batch;
    int T, early_stop;
    const float *beta; int n_beta;
    const float *alpha; int n_alpha;
    const float *oms_alpha; int n_oms_alpha;
    const float *thr; int n_levels; const int *q_of_iter;
    int *bits; float *posterior; int *iterations; uint8_t *success; uint8_t *packed;
};

// ---- check phase: v2c -> c2v in place, lane = check -----------------------------------------
template <int G, int FORM>
__device__ __forceinline__ void res_check_phase(const ResidentPlan &pl, float *__restrict__ msg,
                                                const float *__restrict__ beta_row,
                                                const float *__restrict__ oa_row,
                                                const float *__restrict__ thr, int n_levels,
                                                int tid, int nt)
{
    using P = Pack<float, G>;
    P *M = reinterpret_cast<P *>(msg);
    float th[8];
    if (FORM == FORM_RCQ) {
#pragma unroll
        for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
    }
    for (int p = tid; p < pl.m; p += nt) {
        const int dc = pl.dc_s[p];
        const float b_check = pl.bslot_c ? beta_row[pl.bslot_c[p]] : 0.0f;
        float m1[G], m2[G];
        int idx[G];
        uint32_t sm[G], zm[G];
        unsigned nz[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            m1[g] = inf_of<float>(); m2[g] = inf_of<float>(); idx[g] = 0; sm[g] = 0; zm[g] = 0; nz[g] = 0;
        }
        for (int t = 0; t < dc; ++t) {
            const P v = M[t * pl.m + p];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float a = __builtin_fabsf(v.x[g]);
                sm[g] |= signbit_of<float>(v.x[g]) << t;
                if (FORM == FORM_OMS) {
                    const unsigned z = (a == 0.0f) ? 1u : 0u;
                    nz[g] += z;
                    zm[g] |= z << t;
                }
                if (a < m1[g]) { m2[g] = m1[g]; m1[g] = a; idx[g] = t; }
                else if (a < m2[g]) { m2[g] = a; }
            }
        }
        unsigned par[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            par[g] = __popc(sm[g]) & 1u;
            if (dc == 1) m2[g] = m1[g];
        }
        for (int t = 0; t < dc; ++t) {
            const int slot = t * pl.m + p;
            const float b = pl.bslot_c ? b_check : beta_row[pl.bslot[slot]];
            float oa = 0.0f;
            if (FORM == FORM_OMS && oa_row) oa = oa_row[pl.oaslot[slot]];
            P o;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float raw = (t == idx[g]) ? m2[g] : m1[g];
                const unsigned neg = par[g] ^ ((sm[g] >> t) & 1u);
                if (FORM == FORM_NMS) {
                    o.x[g] = flip_sign<float>(b * raw, neg);
                } else if (FORM == FORM_OMS) {
                    const unsigned ownz = (zm[g] >> t) & 1u;
                    const bool nonzero = (nz[g] - ownz) == 0;
                    const float d = raw - b;
                    const float r = d > 0.0f ? d : 0.0f;
                    const float val = r - oa;
                    o.x[g] = nonzero ? flip_sign<float>(val, neg) : 0.0f;
                } else {
                    // quantise + reconstruct in one go: value = (1 - 2*sign_bit) * tau[level]
                    const float w = flip_sign<float>(b * raw, neg);
                    const float mag = __builtin_fabsf(w);
                    float rec = (n_levels <= 8) ? th[0] : thr[0];          // level 0 when nothing matches
                    if (n_levels <= 8) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) rec = (mag >= th[q]) ? th[q] : rec;
                    } else {
                        for (int q = 0; q < n_levels; ++q) rec = (mag >= thr[q]) ? thr[q] : rec;
                    }
                    o.x[g] = flip_sign<float>(rec, (w < 0.0f) ? 1u : 0u);
                }
            }
            M[slot] = o;
        }
    }
}

// ---- variable phase, lane = variable ----------------------------------------------------------
// MODE 0: v2c = llr + alpha * sum(others), scattered back in place
// MODE 1: posterior -> hard-decision byte bits_s[q]; components in `emask` also overwrite their
//         (dead) LLR slot with the posterior so the output pass can read it in original order
template <int G, int DV, int MODE>
__device__ __forceinline__ void res_var_body(const ResidentPlan &pl, float *__restrict__ msg,
                                             float *__restrict__ llr_s, uint8_t *__restrict__ bits_s,
                                             int q, float a, unsigned emask)
{
    using P = Pack<float, G>;
    P *M = reinterpret_cast<P *>(msg);
    P *L = reinterpret_cast<P *>(llr_s);
    int slot[DV > 0 ? DV : 1];
    P x[DV > 0 ? DV : 1];
#pragma unroll
    for (int k = 0; k < DV; ++k) slot[k] = pl.vslot[k * pl.n + q];
#pragma unroll
    for (int k = 0; k < DV; ++k) x[k] = M[slot[k]];
    P l = L[q];
    if constexpr (MODE == 0) {
        P out[DV > 0 ? DV : 1];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float xs[DV > 0 ? DV : 1];
#pragma unroll
            for (int k = 0; k < DV; ++k) xs[k] = x[k].x[g];
            if constexpr (DV >= 1) out[0].x[g] = l.x[g] + a * sum_ct<DV - 1, 0, 0, float>(xs);
            if constexpr (DV >= 2) out[1].x[g] = l.x[g] + a * sum_ct<DV - 1, 1, 0, float>(xs);
            if constexpr (DV >= 3) out[2].x[g] = l.x[g] + a * sum_ct<DV - 1, 2, 0, float>(xs);
            if constexpr (DV >= 4) out[3].x[g] = l.x[g] + a * sum_ct<DV - 1, 3, 0, float>(xs);
            if constexpr (DV >= 5) out[4].x[g] = l.x[g] + a * sum_ct<DV - 1, 4, 0, float>(xs);
            if constexpr (DV >= 6) out[5].x[g] = l.x[g] + a * sum_ct<DV - 1, 5, 0, float>(xs);
            if constexpr (DV >= 7) out[6].x[g] = l.x[g] + a * sum_ct<DV - 1, 6, 0, float>(xs);
            if constexpr (DV >= 8) out[7].x[g] = l.x[g] + a * sum_ct<DV - 1, 7, 0, float>(xs);
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) M[slot[k]] = out[k];
    } else {
        unsigned byte = 0;
        bool store = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float xs[DV > 0 ? DV : 1];
#pragma unroll
            for (int k = 0; k < DV; ++k) xs[k] = x[k].x[g];
            const float post = l.x[g] + sum_ct<DV, -1, 0, float>(xs);
            byte |= (post < 0.0f ? 1u : 0u) << g;
            if ((emask >> g) & 1u) { l.x[g] = post; store = true; }
        }
        bits_s[q] = (uint8_t)byte;
        if (store) L[q] = l;
    }
}

template <int G, int MODE>
__device__ __forceinline__ void res_var_phase(const ResidentPlan &pl, float *__restrict__ msg,
                                              float *__restrict__ llr_s, uint8_t *__restrict__ bits_s,
                                              const float *__restrict__ alpha_row, unsigned emask,
                                              int tid, int nt)
{
    for (int q = tid; q < pl.n; q += nt) {
        const int dv = pl.dv_s[q];
        const float a = (MODE == 0) ? alpha_row[pl.aslot[q]] : 0.0f;
#define LDPC_RV(D) case D: res_var_body<G, D, MODE>(pl, msg, llr_s, bits_s, q, a, emask); break;
        switch (dv) {
            LDPC_RV(0) LDPC_RV(1) LDPC_RV(2) LDPC_RV(3) LDPC_RV(4) LDPC_RV(5) LDPC_RV(6) LDPC_RV(7) LDPC_RV(8)
        default: break;   // host admits only max_dv <= 8 to this engine
        }
#undef LDPC_RV
    }
}

// H @ bits mod 2 per check from the hard-decision bytes; OR of all checks' parities -> *sh_unsat
template <int G>
__device__ __forceinline__ void res_syndrome_phase(const ResidentPlan &pl, const uint8_t *__restrict__ bits_s,
                                                   unsigned *sh_unsat, int tid, int nt)
{
    unsigned acc = 0;
    for (int p = tid; p < pl.m; p += nt) {
        const int dc = pl.dc_s[p];
        unsigned x = 0;
        for (int t = 0; t < dc; ++t) x ^= bits_s[pl.cvar[t * pl.m + p]];
        acc |= x;
    }
    if (acc) atomicOr(sh_unsat, acc);
}

// write codeword g's outputs; its posterior sits in llr_s[.][g] (sorted order)
template <int G>
__device__ __forceinline__ void res_emit(const ResidentPlan &pl, const ResidentArgs &a, const float *__restrict__ llr_s,
                                         long long b, int g, int iters, int success, int tid, int nt)
{
    const int n = pl.n;
    if (a.posterior || a.bits) {
        for (int j = tid; j < n; j += nt) {
            const float p = llr_s[(int)pl.inv_perm_v[j] * G + g];
            if (a.posterior) a.posterior[(size_t)b * n + j] = p;
            if (a.bits) a.bits[(size_t)b * n + j] = p < 0.0f ? 1 : 0;
        }
    }
    if (a.packed) {
        const int nbytes = (n + 7) / 8;
        for (int k = tid; k < nbytes; k += nt) {
            unsigned v = 0;
#pragma unroll
            for (int qb = 0; qb < 8; ++qb) {
                const int j = k * 8 + qb;
                if (j < n && llr_s[(int)pl.inv_perm_v[j] * G + g] < 0.0f) v |= 1u << qb;
            }
            a.packed[(size_t)b * nbytes + k] = (uint8_t)v;
        }
    }
    if (tid == 0) {
        if (a.iterations) a.iterations[b] = iters;
        if (a.success) a.success[b] = (uint8_t)success;
    }
}

template <int G, int FORM>
__global__ __launch_bounds__(1024) void resident_decode(ResidentPlan pl, ResidentArgs a)
{
    extern __shared__ __align__(16) unsigned char res_smem[];
    float *msg = reinterpret_cast<float *>(res_smem);
    float *llr_s = msg + (size_t)pl.S * G;
    uint8_t *bits_s = reinterpret_cast<uint8_t *>(llr_s + (size_t)pl.n * G);
    __shared__ unsigned sh_unsat;
    using P = Pack<float, G>;
    const int tid = threadIdx.x, nt = blockDim.x, n = pl.n;
    const long long b0 = (long long)blockIdx.x * G;
    constexpr unsigned kAll = (1u << G) - 1u;

    // LLRs: coalesced rows from HBM, scattered into degree-sorted order; padding codewords get +1
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool live = b0 + g < a.batch;
        const float *row = a.llr + (size_t)(b0 + g) * n;
        for (int j = tid; j < n; j += nt) llr_s[(int)pl.inv_perm_v[j] * G + g] = live ? row[j] : 1.0f;
    }
    if (tid == 0) sh_unsat = 0;
    __syncthreads();
    // "initialise v2c with the channel LLRs" (T == 0: c2v = 0, the loop never runs)
    {
        P *M = reinterpret_cast<P *>(msg);
        const P *L = reinterpret_cast<const P *>(llr_s);
        for (int q = tid; q < n; q += nt) {
            const int dv = pl.dv_s[q];
            P l = L[q];
            if (a.T == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) l.x[g] = 0.0f;
            }
            for (int k = 0; k < dv; ++k) M[pl.vslot[k * n + q]] = l;
        }
    }
    __syncthreads();

    unsigned done = 0;                                   // block-uniform
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (b0 + g >= a.batch) done |= 1u << g;

    for (int it = 0; it < a.T; ++it) {
        const float *beta_row = a.beta + (size_t)it * a.n_beta;
        const float *oa_row = a.oms_alpha ? a.oms_alpha + (size_t)it * a.n_oms_alpha : nullptr;
        const float *thr = FORM == FORM_RCQ ? a.thr + (size_t)a.q_of_iter[it] * a.n_levels : nullptr;
        res_check_phase<G, FORM>(pl, msg, beta_row, oa_row, thr, a.n_levels, tid, nt);
        __syncthreads();
        if (a.early_stop) {
            res_var_phase<G, 1>(pl, msg, llr_s, bits_s, nullptr, 0u, tid, nt);
            __syncthreads();
            res_syndrome_phase<G>(pl, bits_s, &sh_unsat, tid, nt);
            __syncthreads();
            const unsigned unsat = sh_unsat;
            __syncthreads();
            if (tid == 0) sh_unsat = 0;
            const unsigned newly = ~unsat & ~done & kAll;
            if (newly) {                                 // block-uniform
                res_var_phase<G, 1>(pl, msg, llr_s, bits_s, nullptr, newly, tid, nt);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if ((newly >> g) & 1u) res_emit<G>(pl, a, llr_s, b0 + g, g, it + 1, 1, tid, nt);
                done |= newly;
                if (done == kAll) return;                // every codeword of the block has its outputs
                __syncthreads();
            }
        }
        if (it != a.T - 1) {
            res_var_phase<G, 0>(pl, msg, llr_s, bits_s, a.alpha + (size_t)it * a.n_alpha, 0u, tid, nt);
            __syncthreads();
        }
    }

    // codewords still open after T iterations: outputs of the last iteration
    const unsigned open = ~done & kAll;
    if (!open) return;
    res_var_phase<G, 1>(pl, msg, llr_s, bits_s, nullptr, open, tid, nt);
    __syncthreads();
    unsigned unsat = kAll;
    if (!a.early_stop) {                                 // fixed-T mode: success = final syndrome is zero
        res_syndrome_phase<G>(pl, bits_s, &sh_unsat, tid, nt);
        __syncthreads();
        unsat = sh_unsat;
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
        if ((open >> g) & 1u) res_emit<G>(pl, a, llr_s, b0 + g, g, a.T, ((unsat >> g) & 1u) ? 0 : 1, tid, nt);
}

}  // namespace ldpc
