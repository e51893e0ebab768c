// ldpc_resident.hip -- LDS-resident fused decoder (gfx950): all T iterations of G codewords
// run inside ONE workgroup with every message in LDS; HBM sees only the LLRs in and the
// decisions out (~16 KB per codeword instead of T*(16E+4n) ~ 1.15 MB for the (1998,1512) code).
//
// Mapping (the transpose of the streaming kernels): lanes run over the NODES of the graph,
// the G codewords of the workgroup ride along as a G-wide vector per lane (ds_read/write_b64
// for G = 2, b128 for G = 4).
//   msg  [S][G]  fp32   at LDS offset 0; one slot per edge, "ELL-transposed" check-major:
//                       slot(p,t) = t*m + p  (p = position of the check after sorting by degree)
//                       -> consecutive lanes hit consecutive LDS words in the check phase
//   llr_s[n][G]  fp32   channel LLRs in degree-sorted variable order
//   bits_s[n]    u8     hard decisions of the G codewords (bit g), for the syndrome
//   alpha_s      fp32   the [T][n_alpha] V2C weight table (when it is small)
// One array holds both message directions in place: the check phase turns v2c into c2v slot by
// slot, the variable phase gathers its dv slots (byte offsets, 8 x u16 in one 16-byte load), forms
// the leave-one-out sums in the reference's association order and scatters v2c back.
//
// The engine is bound by instruction issue, not by LDS or HBM, so the phases are written for
// instruction count: nodes are sorted by degree and every wave whose lanes share one degree (all
// but the class-boundary waves) runs wave-uniform control flow -- scalar loop counters, one scalar
// branch into the compile-time body of its degree; min1/min2 by v_med3; the sign product as an XOR
// of raw bit patterns; index lists fetched one variable ahead.
// Arithmetic is the streaming kernels' (same helpers), so the two engines give identical results
// (the GPU parity tests run every case on both).
//
// Early stop (reference semantics): every iteration a posterior pass + syndrome pass finds the
// codewords that just converged; their posterior is recomputed into their (now dead) LLR slots
// and written out at once; the workgroup leaves when all G are done.
#pragma once

#include "ldpc_kernels.hip"

namespace ldpc {

struct ResidentPlan {
    int n, m, S, max_dc, max_dv;
    int mstride;                  // slots between consecutive edges of a check: slot(p,t) = t*mstride + p
                                  // (512 when m <= 512 so that LDS instructions can carry t*stride as an
                                  //  immediate offset; else m)
    const uint8_t *dc_s;          // [m]          degree of the (sub-)check at sorted position p
    const uint8_t *gsz;           // [m] or null  lane-group size of position p: a check wider than the slot layout admits is
                                  //              split over 2^k ADJACENT lanes (sub-checks of contiguous edges); 1 = whole check
    int any_split;                // some check is split (gsz != null): the host launches the SPLIT instantiation of the kernel
    int par_words;                // m when the row stride is a power of two: parity words of the early-stop syndrome (ParScatter)
    int par_shift;                // log2 of the bytes per slot
    const uint16_t *cvar;         // [max_dc*m]   sorted position of the variable of edge (p,t)
    const uint16_t *bslot;        // [max_dc*m]   beta table column of edge (p,t)
    const uint16_t *bslot_c;      // [m] or null: column shared by all edges of check p (Basic, RCQ,
                                  //              sharing types 2-4) -> one table read per check
    const uint16_t *oaslot;       // [max_dc*m]   OMS alpha column (or null)
    const uint32_t *vmeta;        // [n]          degree | alpha column << 8 of sorted variable q
    const uint2 *vslot_lo;        // [n]          LDS byte offsets (= slot * G * 4, all <= 65535) of edges k = 0..3 of q, four
                                  //              16-bit values in 8 bytes
    const uint2 *vslot_hi;        // [n_hi]       ... of edges k = 4..7, for the n_hi leading (highest-degree) variables only
    int n_hi;                     //              variables of degree > 4 (they come first in the degree-sorted order)
                                  // 8 + 8 + 4 bytes per variable: the whole plan of the (1998,1512) code is 26 KB and stays in
                                  // the CU's 32 KiB L1 across iterations (with 32-bit offsets and the hi half read for every
                                  // variable it was 72 KB per workgroup and iteration from L2)
    const uint16_t *inv_perm_v;   // [n]          sorted position of original variable j
    int E;                        // edges of the graph
    const uint32_t *edge_of_slot; // [S]          CSR edge id held by a slot, 0xffffffff for padding slots (test hook
                                  //              ldpc_debug_resident_c2v: dumps the C2V state in CSR order)
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
    int alpha_in_lds;             // T * n_alpha floats staged in LDS
    int unit_alpha;               // every alpha == 1.0f (Basic, RCQ, sharing types 1/3): the multiply is skipped
    int rcq_zero0;                // every quantiser has tau_0 == 0 (true for gamma > 0): per-check quantisation
    int debug_skip;               // phase-timing probes, compiled in only with -DLDPC_RESIDENT_PROBES (tools/resident_probe*.sh)
    void *dbg_c2v;                // [batch][E] or null: C2V values of every codeword's last executed iteration, CSR edge
                                  // order (include/ldpc_hip_debug.h; lets the tests compare per-edge RCQ codes on this engine)
};

constexpr int kResAlphaMax = 1024;   // floats of alpha table kept in LDS
constexpr int kResHeld = 16;         // check degrees up to this keep their values in registers between the two passes
#ifndef LDPC_RES_CHECK_MODE
#define LDPC_RES_CHECK_MODE 0        // 0 per-lane form only | 1 scalar form for single-degree waves | 2 one scalar pass per degree
                                     // (measured on one box, (1998,1512) Basic / RCQ: 0: 3.10 / 3.30 ms, 1: 3.13 / 3.33, 2: 3.20 / 3.48;
                                     //  LDPC_RES_VAR_MODE 1 costs another 0.15-0.25 ms: DESIGN.md 5)
#endif
#ifndef LDPC_RES_FINAL_SCATTER
#define LDPC_RES_FINAL_SCATTER 1      // fixed T: final syndrome by parity scatter (0: decisions through the dead message slots)
#endif
#ifndef LDPC_RES_NO_PLAN_PREFETCH
#define LDPC_RES_NO_PLAN_PREFETCH 0
#endif
#ifndef LDPC_RES_SELECT4
#define LDPC_RES_SELECT4 1           // pass 2 of the check phase in hand-scheduled groups of four values (0: the compiler's form)
#endif
#ifndef LDPC_RES_PLAN_U32
#define LDPC_RES_PLAN_U32 1          // plan prefetch addressed with 32-bit offsets against a scalar base (-0.3 % fp32, measured)
#endif
#ifndef LDPC_RES_F64_MINMAX
#define LDPC_RES_F64_MINMAX 1        // float64 check phase: branch-free min1/min2, products hoisted, pass 2 in groups of four edges
#endif
#ifndef LDPC_RES_VAR_MODE
#define LDPC_RES_VAR_MODE 0          // 0 per-lane dispatch | 1 one scalar pass per distinct degree
#endif

#ifdef LDPC_RESIDENT_PROBES
#define LDPC_PROBE(a, bit) ((a).debug_skip & (bit))
#else
#define LDPC_PROBE(a, bit) 0
#endif

// LDS access by byte offset.  The dynamic LDS block is this kernel's only LDS object (no static
// __shared__), so it starts at LDS address 0 and a message slot's byte offset IS its LDS address:
// addressing through an address-space-3 pointer built from the offset avoids the "+ base" VALU add
// that the generic-pointer form leaves on every access.  resident_decode traps if a static LDS
// allocation ever appears in front of the dynamic block.
template <typename X>
__device__ __forceinline__ X lds_load(unsigned byte_off)
{
    constexpr int N = sizeof(X) / sizeof(float);
    typedef float VT __attribute__((ext_vector_type(N)));
    using LP = __attribute__((address_space(3))) const VT *;
    union { VT v; X k; } u;
    u.v = *(LP)(size_t)byte_off;
    return u.k;
}
template <typename X>
__device__ __forceinline__ void lds_store(unsigned byte_off, const X &v)
{
    constexpr int N = sizeof(X) / sizeof(float);
    typedef float VT __attribute__((ext_vector_type(N)));
    using LP = __attribute__((address_space(3))) VT *;
    union { VT v; X k; } u;
    u.k = v;
    *(LP)(size_t)byte_off = u.v;
}

// LLRs in and decisions out are touched once per codeword: non-temporal, so that they do not evict the plan (the index data
// every iteration re-reads) from the CU's L1
#ifndef LDPC_RES_NT_IO
#define LDPC_RES_NT_IO 1
#endif
template <typename X>
__device__ __forceinline__ X res_stream_load(const X *p)
{
#if LDPC_RES_NT_IO
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename X>
__device__ __forceinline__ void res_stream_store(X *p, X v)
{
#if LDPC_RES_NT_IO
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ void lds_atomic_xor(unsigned byte_off, unsigned v)
{
    using LP = __attribute__((address_space(3))) unsigned *;
    __hip_atomic_fetch_xor((LP)(size_t)byte_off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Early-stop syndrome without a second gather: the variable lane that has just formed a hard decision XORs it into the
// parity word of each of its checks (one LDS atomic per edge).  The check position is recovered from the edge's slot
// offset -- slot(p,t) = t*stride + p with a power-of-two stride: p = (offset >> shift) & mask -- so no index data is
// loaded; `par_off` = 0 switches the scatter off (then the checks gather the decisions, res_syndrome_phase).
struct ParScatter {
    unsigned par_off = 0, shift = 0, mask = 0;
};

__device__ __forceinline__ bool wave_uniform(int v, int &vw)
{
    vw = __builtin_amdgcn_readfirstlane(v);
    return __ballot(v != vw) == 0ull;
}

// ---- check phase: v2c -> c2v in place, lane = check -----------------------------------------
//   pass 1  per value: min1' = med3(a, min1, -inf) = min, min2' = med3(a, min1, min2), signs ^= bits(x)
//   pass 2  re-reads the slot (LDS reads are cheap here): the arg-min edge is recognised by
//           |x| == min1 -- on a tie min2 == min1, so which tied edge "is" the arg-min is
//           value-irrelevant, exactly as with the reference's first-index argmin -- and the sign of
//           the product of the OTHER signs is bit 31 of (signs ^ x).
// quantise-and-reconstruct magnitude: tau[last q with mag >= tau_q], default tau_0; the q = 0
// comparison can never change the outcome and is left out (any threshold order, NaN included)
template <int NL>
__device__ __forceinline__ float res_quant_rec(float mag, const float (&th)[8], const float *__restrict__ thr, int n_levels)
{
    float rec = (n_levels <= 8) ? th[0] : thr[0];
    if constexpr (NL > 0) {                                   // compile-time level count (bc = 3: 4)
#pragma unroll
        for (int q = 1; q < NL; ++q) rec = (mag >= th[q]) ? th[q] : rec;
    } else if (n_levels <= 8) {
#pragma unroll
        for (int q = 1; q < 8; ++q) rec = (mag >= th[q]) ? th[q] : rec;   // NaN padding never matches
    } else {
        for (int q = 1; q < n_levels; ++q) rec = (mag >= thr[q]) ? thr[q] : rec;
    }
    return rec;
}

// ---- wide checks: one check split over a group of 2^k adjacent lanes ---------------------------------------------
// Each lane reduces its own edges (min1 / min2 / sign bits / zero count); the partials are combined across the group by
// an XOR butterfly of wavefront exchanges -- ds_swizzle (bit-mask mode, no LDS memory touched) inside 32 lanes, a
// cross-half shuffle for the last step -- after which every lane of the group holds the whole check's values and emits
// its own edges.  Steps beyond a lane's own group size are exchanged too (the instruction is wave-wide) but not combined.
template <int O>
__device__ __forceinline__ unsigned lane_xchg(unsigned v)
{
    if constexpr (O < 32) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (O << 10) | 0x1f);
    else return (unsigned)__shfl_xor((int)v, 32, 64);
}
template <int O> __device__ __forceinline__ float lane_xchg(float v) { return __uint_as_float(lane_xchg<O>(__float_as_uint(v))); }
template <int O>
__device__ __forceinline__ double lane_xchg(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = lane_xchg<O>((unsigned)u), hi = lane_xchg<O>((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int O, int G, typename T>
__device__ __forceinline__ void group_step(int gs, T (&m1)[G], T (&m2)[G], unsigned (&sg)[G], unsigned (&nz)[G])
{
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const T a1 = lane_xchg<O>(m1[g]), a2 = lane_xchg<O>(m2[g]);
        const unsigned as = lane_xchg<O>(sg[g]), an = lane_xchg<O>(nz[g]);
        if (O < gs) {
            // second smallest of {m1, m2, a1, a2} = min(max(m1, a1), min(m2, a2)); ties keep min2 == min1
            const T hi = m1[g] < a1 ? a1 : m1[g];
            const T lo2 = m2[g] < a2 ? m2[g] : a2;
            m2[g] = hi < lo2 ? hi : lo2;
            m1[g] = m1[g] < a1 ? m1[g] : a1;
            sg[g] ^= as;
            nz[g] += an;
        }
    }
}
template <int G, typename T>
__device__ __forceinline__ void group_combine(int gs, T (&m1)[G], T (&m2)[G], unsigned (&sg)[G], unsigned (&nz)[G])
{
    group_step<1, G, T>(gs, m1, m2, sg, nz);
    group_step<2, G, T>(gs, m1, m2, sg, nz);
    group_step<4, G, T>(gs, m1, m2, sg, nz);
    group_step<8, G, T>(gs, m1, m2, sg, nz);
    group_step<16, G, T>(gs, m1, m2, sg, nz);
    group_step<32, G, T>(gs, m1, m2, sg, nz);
}
// parity words of the syndrome phases
__device__ __forceinline__ unsigned group_xor(int gs, unsigned x)
{
    unsigned y;
    y = lane_xchg<1>(x);  if (1 < gs) x ^= y;
    y = lane_xchg<2>(x);  if (2 < gs) x ^= y;
    y = lane_xchg<4>(x);  if (4 < gs) x ^= y;
    y = lane_xchg<8>(x);  if (8 < gs) x ^= y;
    y = lane_xchg<16>(x); if (16 < gs) x ^= y;
    y = lane_xchg<32>(x); if (32 < gs) x ^= y;
    return x;
}

// Pass 2 of the check phase for TWO edges of a codeword pair (four values), hand-scheduled: out = (|x| == min1 ? o2 : o1) ^
// sign(x).  Written in C the compiler emits compare -> s_nop -> v_cndmask per value (a lane mask needs two instructions
// between its compare and its reader) and keeps the sign-bit constant in an SGPR -- a v_bitop3_b32 with an SGPR operand
// issues in ~4.2 cycles instead of ~2.3 (tools/probes/valu_issue_probe.hip).  Four compares into four lane masks, four
// selects, four bitop3 with the constant in a VGPR: no padding, 10.7 instead of ~15.5 issue cycles per value.
__device__ __forceinline__ void res_select4(float &x0, float &x1, float &x2, float &x3, float m1a, float m1b,
                                            uint32_t o1a, uint32_t o2a, uint32_t o1b, uint32_t o2b, uint32_t sign_v)
{
    unsigned long long c0, c1, c2;
    uint32_t t0, t1, t2, t3;
    asm("v_cmp_eq_f32_e64 %8, |%0|, %11\n\t"
        "v_cmp_eq_f32_e64 %9, |%1|, %12\n\t"
        "v_cmp_eq_f32_e64 %10, |%2|, %11\n\t"
        "v_cmp_eq_f32_e64 vcc, |%3|, %12\n\t"
        "v_cndmask_b32_e64 %4, %13, %14, %8\n\t"
        "v_cndmask_b32_e64 %5, %15, %16, %9\n\t"
        "v_cndmask_b32_e64 %6, %13, %14, %10\n\t"
        "v_cndmask_b32_e32 %7, %15, %16, vcc\n\t"
        "v_bitop3_b32 %0, %4, %0, %17 bitop3:0x78\n\t"
        "v_bitop3_b32 %1, %5, %1, %17 bitop3:0x78\n\t"
        "v_bitop3_b32 %2, %6, %2, %17 bitop3:0x78\n\t"
        "v_bitop3_b32 %3, %7, %3, %17 bitop3:0x78"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(m1a), "v"(m1b), "v"(o1a), "v"(o2a), "v"(o1b), "v"(o2b), "v"(sign_v)
        : "vcc");
}

// the same for FOUR edges of one float64 codeword: compares on the 64-bit values, selects and sign on the 32-bit halves
__device__ __forceinline__ void res_select4_f64(double (&x)[4], double m1, double o1, double o2, uint32_t sign_v)
{
    const unsigned long long b1 = (unsigned long long)__double_as_longlong(o1), b2 = (unsigned long long)__double_as_longlong(o2);
    const uint32_t o1l = (uint32_t)b1, o1h = (uint32_t)(b1 >> 32), o2l = (uint32_t)b2, o2h = (uint32_t)(b2 >> 32);
    uint32_t xh[4], rl[4], rh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xh[i] = (uint32_t)((unsigned long long)__double_as_longlong(x[i]) >> 32);
    unsigned long long c0, c1, c2;
    asm("v_cmp_eq_f64_e64 %[c0], |%[x0]|, %[m1]\n\t"
        "v_cmp_eq_f64_e64 %[c1], |%[x1]|, %[m1]\n\t"
        "v_cmp_eq_f64_e64 %[c2], |%[x2]|, %[m1]\n\t"
        "v_cmp_eq_f64_e64 vcc, |%[x3]|, %[m1]\n\t"
        "v_cndmask_b32_e64 %[l0], %[o1l], %[o2l], %[c0]\n\t"
        "v_cndmask_b32_e64 %[h0], %[o1h], %[o2h], %[c0]\n\t"
        "v_cndmask_b32_e64 %[l1], %[o1l], %[o2l], %[c1]\n\t"
        "v_cndmask_b32_e64 %[h1], %[o1h], %[o2h], %[c1]\n\t"
        "v_cndmask_b32_e64 %[l2], %[o1l], %[o2l], %[c2]\n\t"
        "v_cndmask_b32_e64 %[h2], %[o1h], %[o2h], %[c2]\n\t"
        "v_cndmask_b32_e32 %[l3], %[o1l], %[o2l], vcc\n\t"
        "v_cndmask_b32_e32 %[h3], %[o1h], %[o2h], vcc\n\t"
        "v_bitop3_b32 %[h0], %[h0], %[xh0], %[k] bitop3:0x78\n\t"
        "v_bitop3_b32 %[h1], %[h1], %[xh1], %[k] bitop3:0x78\n\t"
        "v_bitop3_b32 %[h2], %[h2], %[xh2], %[k] bitop3:0x78\n\t"
        "v_bitop3_b32 %[h3], %[h3], %[xh3], %[k] bitop3:0x78"
        : [l0] "=&v"(rl[0]), [h0] "=&v"(rh[0]), [l1] "=&v"(rl[1]), [h1] "=&v"(rh[1]), [l2] "=&v"(rl[2]), [h2] "=&v"(rh[2]),
          [l3] "=&v"(rl[3]), [h3] "=&v"(rh[3]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [m1] "v"(m1), [o1l] "v"(o1l), [o1h] "v"(o1h),
          [o2l] "v"(o2l), [o2h] "v"(o2h), [xh0] "v"(xh[0]), [xh1] "v"(xh[1]), [xh2] "v"(xh[2]), [xh3] "v"(xh[3]), [k] "v"(sign_v)
        : "vcc");
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __longlong_as_double((long long)(((unsigned long long)rh[i] << 32) | rl[i]));
}

// Check update of ONE degree, fully unrolled (fp32, one beta per check): the DC values are read once, stay in registers
// for both passes and are written back in place -- one LDS read and one LDS write per edge instead of two reads and a
// write, no load-to-use wait in the second pass, no loop control.  Entered through a scalar switch on the wave's degree.
template <int G, int FORM, int NL, int DC>
__device__ __forceinline__ void res_check_held(unsigned base, unsigned stride, float b_check, const float (&th)[8],
                                               const float *__restrict__ thr, int n_levels)
{
    using P = Pack<float, G>;
    P v[DC];
#pragma unroll
    for (int t = 0; t < DC; ++t) v[t] = lds_load<P>(base + t * stride);
    float m1[G], m2[G];
    uint32_t sacc[G];
    float ninf = -inf_of<float>();
    asm volatile("" : "+v"(ninf));                    // opaque to constant folding: min as ONE v_med3
#pragma unroll
    for (int g = 0; g < G; ++g) { m1[g] = inf_of<float>(); m2[g] = inf_of<float>(); sacc[g] = 0; }
#pragma unroll
    for (int t = 0; t < DC; ++t) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            sacc[g] ^= __float_as_uint(v[t].x[g]);
            m2[g] = __builtin_amdgcn_fmed3f(__builtin_fabsf(v[t].x[g]), m1[g], m2[g]);
            m1[g] = __builtin_amdgcn_fmed3f(__builtin_fabsf(v[t].x[g]), m1[g], ninf);
        }
    }
    if (DC == 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) m2[g] = m1[g];     // "min2_val = min_val" for a degree-1 check
    }
    uint32_t o1[G], o2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float w1 = b_check * m1[g], w2 = b_check * m2[g];
        if (FORM == FORM_RCQ) {
            // value = (1 - 2*(w < 0)) * tau[level(|w|)]; with tau_0 == 0 a zero magnitude reconstructs to +-0, so
            // applying the edge sign afterwards is value-identical to the reference's order
            const float r1 = res_quant_rec<NL>(__builtin_fabsf(w1), th, thr, n_levels);
            const float r2 = res_quant_rec<NL>(__builtin_fabsf(w2), th, thr, n_levels);
            w1 = flip_sign<float>(r1, (w1 < 0.0f) ? 1u : 0u);
            w2 = flip_sign<float>(r2, (w2 < 0.0f) ? 1u : 0u);
        }
        const uint32_t par = sacc[g] & 0x80000000u;
        o1[g] = __float_as_uint(w1) ^ par;
        o2[g] = __float_as_uint(w2) ^ par;
        asm volatile("" : "+v"(o1[g]), "+v"(o2[g]));  // keep the two per-check values materialised (see the generic form)
    }
    int t0 = 0;
    if constexpr (G == 2 && LDPC_RES_SELECT4 != 0) {
        uint32_t sign_v = 0x80000000u;
        asm volatile("" : "+v"(sign_v));
#pragma unroll
        for (int t = 0; t + 1 < DC; t += 2) {
            res_select4(v[t].x[0], v[t].x[1], v[t + 1].x[0], v[t + 1].x[1], m1[0], m1[1], o1[0], o2[0], o1[1], o2[1], sign_v);
            lds_store<P>(base + t * stride, v[t]);
            lds_store<P>(base + (t + 1) * stride, v[t + 1]);
        }
        t0 = DC & ~1;
    }
#pragma unroll
    for (int t = t0; t < DC; ++t) {
        P o;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t sel = (__builtin_fabsf(v[t].x[g]) == m1[g]) ? o2[g] : o1[g];
            o.x[g] = __uint_as_float(__builtin_amdgcn_bitop3_b32(sel, __float_as_uint(v[t].x[g]), 0x80000000u, 0x78));
        }
        lds_store<P>(base + t * stride, o);
    }
}

template <int G, int FORM, bool BPC, bool UNI, int NL, int MS, typename T, bool SPLIT = false>
__device__ __forceinline__ void res_check_body(const ResidentPlan &pl, unsigned char *smem, int p, int dc, int dcw,
                                               T b_check, const T *__restrict__ beta_row,
                                               const T *__restrict__ oa_row, const float (&th)[8],
                                               const float *__restrict__ thr, int n_levels, bool rcq_zero0, int gs = 1)
{
    // UNI: every lane of the wave has the degree dcw (scalar loops).  !UNI: per-lane degrees, and the lanes of a group
    // (gs > 1) hold the pieces of ONE wide check: their partials are combined between the two passes.
    constexpr int kEl = G * (int)sizeof(T);
    const int trip = UNI ? dcw : dc;
    const unsigned stride = (MS > 0 ? (unsigned)MS : (unsigned)pl.mstride) * kEl;   // compile-time when MS > 0
    const unsigned base = (unsigned)p * kEl;
    if constexpr (!std::is_same<T, float>::value) {
        // fp64 (BasicMinSumDecoder with the reference's own float64 LLRs): normalised min-sum with one factor per
        // check, the same two passes in plain compare/select form (v_med3 / v_bitop3 are 32-bit instructions)
        static_assert(FORM == FORM_NMS && BPC, "the fp64 resident engine covers the per-check normalised form");
        using PD = Pack<T, G>;
        T m1[G], m2[G];
        unsigned par[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { m1[g] = inf_of<T>(); m2[g] = inf_of<T>(); par[g] = 0; }
        auto absorb = [&](const PD &v) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                par[g] ^= signbit_of<T>(v.x[g]);
#if LDPC_RES_F64_MINMAX
                // "if a < min1: (min2, min1) = (min1, a) elif a < min2: min2 = a" (ldpc_decoder.py:96-101) without branches:
                // min2' = max(min(a, min2), min1), min1' = min(a, min1) -- the same values for every ordered input, and a NaN
                // is skipped by both forms (v_min/v_max_f64 return the other operand for a quiet NaN; the initial pass
                // quiets the caller's LLRs, arithmetic never produces a signalling one).  Three instructions instead of two
                // compares, two exec-mask branches and five 64-bit moves.
                T lo, m1n;
                asm("v_min_f64 %0, |%1|, %2" : "=v"(lo) : "v"(v.x[g]), "v"(m2[g]));
                asm("v_min_f64 %0, |%1|, %2" : "=v"(m1n) : "v"(v.x[g]), "v"(m1[g]));
                asm("v_max_f64 %0, %1, %2" : "=v"(m2[g]) : "v"(lo), "v"(m1[g]));
                m1[g] = m1n;
#else
                const T a = abs_of<T>(v.x[g]);
                if (a < m1[g]) { m2[g] = m1[g]; m1[g] = a; }
                else if (a < m2[g]) { m2[g] = a; }
#endif
            }
        };
        {
            int t = 0;
            for (; t + 3 < trip; t += 4) {
                const unsigned addr = base + t * stride;
                const PD va = lds_load<PD>(addr), vb = lds_load<PD>(addr + stride);
                const PD vc = lds_load<PD>(addr + 2 * stride), vd = lds_load<PD>(addr + 3 * stride);
                absorb(va); absorb(vb); absorb(vc); absorb(vd);
            }
            for (; t < trip; ++t) absorb(lds_load<PD>(base + t * stride));
        }
        if constexpr (!UNI && SPLIT) {
            unsigned nzd[G];
#pragma unroll
            for (int g = 0; g < G; ++g) nzd[g] = 0;
            group_combine<G, T>(gs, m1, m2, par, nzd);
        }
        T o1[G], o2[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (trip == 1 && gs == 1) m2[g] = m1[g];        // "min2_val = min_val" for a degree-1 check
            o1[g] = flip_sign<T>(b_check * m1[g], par[g]);
            o2[g] = flip_sign<T>(b_check * m2[g], par[g]);
#if LDPC_RES_F64_MINMAX
            asm volatile("" : "+v"(o1[g]), "+v"(o2[g]));   // two products per check, not one v_mul_f64 per edge (see the fp32 form)
#endif
        }
        int t = 0;
#if LDPC_RES_F64_MINMAX
        if constexpr (G == 1) {
            uint32_t sign_v = 0x80000000u;
            asm volatile("" : "+v"(sign_v));
            for (; t + 3 < trip; t += 4) {
                const unsigned addr = base + t * stride;
                T x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = lds_load<PD>(addr + i * stride).x[0];
                res_select4_f64(x, m1[0], o1[0], o2[0], sign_v);
#pragma unroll
                for (int i = 0; i < 4; ++i) { PD o; o.x[0] = x[i]; lds_store<PD>(addr + i * stride, o); }
            }
        }
#endif
#pragma unroll 4
        for (; t < trip; ++t) {
            const unsigned addr = base + t * stride;
            const PD v = lds_load<PD>(addr);
            PD o;
#pragma unroll
            for (int g = 0; g < G; ++g)
                o.x[g] = flip_sign<T>((abs_of<T>(v.x[g]) == m1[g]) ? o2[g] : o1[g], signbit_of<T>(v.x[g]));
            lds_store<PD>(addr, o);
        }
        return;
    } else {
    using P = Pack<float, G>;
    float m1[G], m2[G];
    uint32_t sacc[G];
    unsigned nz[G];
    float ninf = -inf_of<float>();
    asm volatile("" : "+v"(ninf));                    // opaque to constant folding (see below)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m1[g] = inf_of<float>(); m2[g] = inf_of<float>(); sacc[g] = 0; nz[g] = 0;
    }
#if LDPC_RES_CHECK_MODE != 0
    constexpr bool kPerCheck = BPC && (FORM == FORM_NMS || FORM == FORM_RCQ);
    if (kPerCheck && UNI && trip <= kResHeld && (FORM == FORM_NMS || rcq_zero0)) {
        // The common case -- one beta per check and a wave-uniform degree of at most kResHeld edges: one scalar jump
        // into straight-line code for exactly that degree, the check's values held in registers between the passes.
#define LDPC_RH(D) case D: res_check_held<G, FORM, NL, D>(base, stride, b_check, th, thr, n_levels); return;
        switch (trip) {
            LDPC_RH(1) LDPC_RH(2) LDPC_RH(3) LDPC_RH(4) LDPC_RH(5) LDPC_RH(6) LDPC_RH(7) LDPC_RH(8)
            LDPC_RH(9) LDPC_RH(10) LDPC_RH(11) LDPC_RH(12) LDPC_RH(13) LDPC_RH(14) LDPC_RH(15) LDPC_RH(16)
        default: return;                              // trip == 0: nothing to do
        }
#undef LDPC_RH
    }
#endif
    // generic form (any degree, per-lane trip counts, per-edge beta, OMS): pass 1 streams the slots, pass 2 re-reads them
#pragma unroll 4
    for (int t = 0; t < trip; ++t) {
        const P v = lds_load<P>(base + t * stride);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float a = __builtin_fabsf(v.x[g]);
            sacc[g] ^= __float_as_uint(v.x[g]);
            if (FORM == FORM_OMS) nz[g] += (a == 0.0f) ? 1u : 0u;
            m2[g] = __builtin_amdgcn_fmed3f(a, m1[g], m2[g]);
            m1[g] = __builtin_amdgcn_fmed3f(a, m1[g], ninf);               // = min(a, min1) as ONE v_med3 (a
                                                                           // literal -inf is folded into
                                                                           // canonicalise + v_min: 3 ops)
        }
    }
    if constexpr (!UNI && SPLIT) group_combine<G, float>(gs, m1, m2, sacc, nz);
    if (trip == 1 && gs == 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) m2[g] = m1[g];     // "min2_val = min_val" for a degree-1 check
    }

    if (BPC && (FORM == FORM_NMS || (FORM == FORM_RCQ && rcq_zero0))) {
        // One beta per check: only two outgoing magnitudes exist per codeword, so scaling (and for RCQ the
        // whole quantise-reconstruct) is done twice per check instead of once per edge.  Per edge remains:
        // pick by |x| == min1, then xor in the edge's own sign bit (the parity of all signs is pre-folded).
        uint32_t o1[G], o2[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float w1 = b_check * m1[g], w2 = b_check * m2[g];
            if (FORM == FORM_RCQ) {
                // value = (1 - 2*(w < 0)) * tau[level(|w|)]; with tau_0 == 0 a zero magnitude reconstructs to
                // +-0, so applying the edge sign afterwards is value-identical to the reference's order
                const float r1 = res_quant_rec<NL>(__builtin_fabsf(w1), th, thr, n_levels);
                const float r2 = res_quant_rec<NL>(__builtin_fabsf(w2), th, thr, n_levels);
                w1 = flip_sign<float>(r1, (w1 < 0.0f) ? 1u : 0u);
                w2 = flip_sign<float>(r2, (w2 < 0.0f) ? 1u : 0u);
            }
            const uint32_t par = sacc[g] & 0x80000000u;
            o1[g] = __float_as_uint(w1) ^ par;
            o2[g] = __float_as_uint(w2) ^ par;
            // keep the two per-check values materialised: without the barrier the optimiser sinks the
            // multiply (and the quantiser) back behind the per-edge select
            asm volatile("" : "+v"(o1[g]), "+v"(o2[g]));
        }
        int t = 0;
        if constexpr (G == 2 && LDPC_RES_SELECT4 != 0) {
            uint32_t sign_v = 0x80000000u;
            asm volatile("" : "+v"(sign_v));              // the constant in a VGPR (see res_select4)
            for (; t + 3 < trip; t += 4) {
                const unsigned addr = base + t * stride;
                P va = lds_load<P>(addr), vb = lds_load<P>(addr + stride);
                P vc = lds_load<P>(addr + 2 * stride), vd = lds_load<P>(addr + 3 * stride);
                res_select4(va.x[0], va.x[1], vb.x[0], vb.x[1], m1[0], m1[1], o1[0], o2[0], o1[1], o2[1], sign_v);
                lds_store<P>(addr, va);
                lds_store<P>(addr + stride, vb);
                res_select4(vc.x[0], vc.x[1], vd.x[0], vd.x[1], m1[0], m1[1], o1[0], o2[0], o1[1], o2[1], sign_v);
                lds_store<P>(addr + 2 * stride, vc);
                lds_store<P>(addr + 3 * stride, vd);
            }
            if (t + 1 < trip) {
                const unsigned addr = base + t * stride;
                P va = lds_load<P>(addr), vb = lds_load<P>(addr + stride);
                res_select4(va.x[0], va.x[1], vb.x[0], vb.x[1], m1[0], m1[1], o1[0], o2[0], o1[1], o2[1], sign_v);
                lds_store<P>(addr, va);
                lds_store<P>(addr + stride, vb);
                t += 2;
            }
        }
#pragma unroll 1
        for (; t < trip; ++t) {
            const unsigned addr = base + t * stride;
            const P v = lds_load<P>(addr);
            P o;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t sel = (__builtin_fabsf(v.x[g]) == m1[g]) ? o2[g] : o1[g];
                // sel ^ (x & sign bit) as ONE v_bitop3_b32 (truth table a ^ (b & c) = 0x78); written with & and ^ the
                // compiler emits v_and + v_xor here
                o.x[g] = __uint_as_float(__builtin_amdgcn_bitop3_b32(sel, __float_as_uint(v.x[g]), 0x80000000u, 0x78));
            }
            lds_store<P>(addr, o);
        }
        return;
    }

    int slot = p;
    const int sstride = MS > 0 ? MS : pl.mstride;
#pragma unroll 2
    for (int t = 0; t < trip; ++t) {
        const unsigned addr = base + t * stride;
        const float b = BPC ? b_check : beta_row[pl.bslot[slot]];
        float oa = 0.0f;
        if (FORM == FORM_OMS && oa_row) oa = oa_row[pl.oaslot[slot]];
        const P v = lds_load<P>(addr);
        P o;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float a = __builtin_fabsf(v.x[g]);
            const float raw = (a == m1[g]) ? m2[g] : m1[g];
            const uint32_t sflip = (__float_as_uint(v.x[g]) ^ sacc[g]) & 0x80000000u;
            if (FORM == FORM_NMS) {
                o.x[g] = __uint_as_float(__float_as_uint(b * raw) ^ sflip);
            } else if (FORM == FORM_OMS) {
                const unsigned ownz = (a == 0.0f) ? 1u : 0u;
                const bool nonzero = (nz[g] - ownz) == 0;
                const float d = raw - b;
                const float r = d > 0.0f ? d : 0.0f;
                const float val = r - oa;
                o.x[g] = nonzero ? __uint_as_float(__float_as_uint(val) ^ sflip) : 0.0f;
            } else {
                // quantise + reconstruct in one go: value = (1 - 2*sign_bit) * tau[level]
                const float w = __uint_as_float(__float_as_uint(b * raw) ^ sflip);
                const float rec = res_quant_rec<NL>(__builtin_fabsf(w), th, thr, n_levels);
                o.x[g] = flip_sign<float>(rec, (w < 0.0f) ? 1u : 0u);
            }
        }
        lds_store<P>(addr, o);
        slot += sstride;
    }
    }   // fp32
}

// `dc_pre` / `b_pre` are the first round's degree and per-check beta, fetched by the caller ahead of
// the barrier so their global-memory latency is off the critical path.
template <int G, int FORM, bool BPC, int NL, int MS, typename T, bool SPLIT>
__device__ __forceinline__ void res_check_phase(const ResidentPlan &pl, unsigned char *smem,
                                                const T *__restrict__ beta_row,
                                                const T *__restrict__ oa_row,
                                                const float *__restrict__ thr, int n_levels, bool rcq_zero0,
                                                int dc_pre, T b_pre, int tid, int nt)
{
    float th[8];
    if (FORM == FORM_RCQ) {
#pragma unroll
        for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
    }
    int dc = dc_pre;
    T b_check = b_pre;
    for (int p = tid; p < pl.m; p += nt) {
        if (p != tid) {
            dc = pl.dc_s[p];
            b_check = BPC ? beta_row[pl.bslot_c[p]] : (T)0;
        }
        if constexpr (SPLIT) {
            // a wave holding lane groups runs the per-lane form for all its lanes at once: the group exchange between the
            // passes needs every lane of a group at the same point of the program
            const int gs = pl.gsz[p];
            if (__ballot(gs > 1) != 0ull) {
                res_check_body<G, FORM, BPC, false, NL, MS, T, true>(pl, smem, p, dc, dc, b_check, beta_row, oa_row, th, thr, n_levels, rcq_zero0, gs);
                continue;
            }
        }
#if LDPC_RES_CHECK_MODE == 2
        // One pass per DISTINCT degree in the wave (nodes are sorted by degree: all but the class-boundary waves make one
        // pass), each with a scalar trip count.  The loop itself is uniform -- it runs on the scalar mask of lanes not yet
        // served, every lane stays in it -- and the degree is handed to the body as an opaque scalar copy made BEFORE the
        // comparison: inside `if (dc == dcw)` the optimiser would otherwise substitute the per-lane dc for the scalar and
        // turn the edge loops back into exec-masked vector loops.  (A `for (pending) { rfl; if (==) {...} }` waterfall is not
        // safe here: with the readfirstlane hoisted, lanes of another degree would spin forever, which the optimiser is
        // entitled to assume never happens -- it then drops the comparison and runs every lane with the first lane's degree.)
        for (unsigned long long todo = __ballot(true); todo;) {
            const int dcw = __builtin_amdgcn_readlane(dc, __ffsll((long long)todo) - 1);
            int trip = dcw;
            asm volatile("" : "+s"(trip));
            const bool mine = dc == dcw;
            if (mine)
                res_check_body<G, FORM, BPC, true, NL, MS, T>(pl, smem, p, dc, trip, b_check, beta_row, oa_row, th, thr, n_levels, rcq_zero0);
            todo &= ~__ballot(mine);
        }
#elif LDPC_RES_CHECK_MODE == 1
        // A wave whose lanes all have one degree (nodes are sorted by degree: all but the class-boundary waves) runs the
        // scalar form -- trip count in an SGPR, for the common forms straight-line code of exactly that degree with the
        // values held in registers; a class-boundary wave runs the per-lane form once for all its lanes (exec-masked
        // loops over per-lane trip counts).  The scalar is an opaque copy: were it derived from `dc` visibly, the optimiser
        // would put the per-lane value back (it knows dc == dcw in that branch).
        {
            const int dcw = __builtin_amdgcn_readfirstlane(dc);
            int trip = dcw;
            asm volatile("" : "+s"(trip));
            if (__ballot(dc != dcw) == 0ull)
                res_check_body<G, FORM, BPC, true, NL, MS, T>(pl, smem, p, dc, trip, b_check, beta_row, oa_row, th, thr, n_levels, rcq_zero0);
            else
                res_check_body<G, FORM, BPC, false, NL, MS, T>(pl, smem, p, dc, dc, b_check, beta_row, oa_row, th, thr, n_levels, rcq_zero0);
        }
#else
        // every lane runs the edge loops with ITS degree as trip count (exec-masked vector loops): one pass per wave
        // whatever the mix of degrees.  Measured fastest (see LDPC_RES_CHECK_MODE above): the phase is bound by VALU issue
        // -- six instructions per edge and codeword, which the scalar forms do not reduce -- not by loop control.
        res_check_body<G, FORM, BPC, false, NL, MS, T>(pl, smem, p, dc, dc, b_check, beta_row, oa_row, th, thr, n_levels, rcq_zero0);
#endif
    }
}

// ---- variable phase, lane = variable ----------------------------------------------------------
// MODE 0: v2c = llr + alpha * sum(others), scattered back in place
// MODE 2: the same with alpha == 1 everywhere (1.0f * x == x exactly, the multiply is skipped)
// MODE 1: posterior -> hard-decision byte bits_s[q]; components in `emask` also overwrite their
//         (dead) LLR slot with the posterior so the output pass can read it in original order
// MODE 4 / 6: MODE 0 / 2 plus the hard-decision byte from the same gathered values -- the early-stop
//         iteration of callers that do not ask for the posterior (one gather pass instead of two)
template <int G, int DV, int MODE, typename T>
__device__ __forceinline__ void res_var_body(unsigned char *smem, T *__restrict__ llr_s,
                                             uint8_t *__restrict__ bits_s, int q, const uint4 &slo, const uint4 &shi,
                                             T a, unsigned emask, const ParScatter &ps)
{
    using P = Pack<T, G>;
    constexpr int ORD = std::is_same<T, float>::value ? 0 : 1;       // torch.sum fp32 order / np.sum fp64 order
    P *L = reinterpret_cast<P *>(llr_s);
    const unsigned off[8] = {slo.x, slo.y, slo.z, slo.w, shi.x, shi.y, shi.z, shi.w};
    P x[DV > 0 ? DV : 1];
#pragma unroll
    for (int k = 0; k < DV; ++k) x[k] = lds_load<P>(off[k]);
    P l = L[q];
    if constexpr (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6) {
        P out[DV > 0 ? DV : 1];
        auto v2c = [&](T llr, T sum) { return (MODE & 2) ? llr + sum : llr + a * sum; };
        unsigned byte = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            T xs[DV > 0 ? DV : 1];
#pragma unroll
            for (int k = 0; k < DV; ++k) xs[k] = x[k].x[g];
            if constexpr ((MODE & 4) != 0) byte |= ((l.x[g] + sum_ct<DV, -1, ORD, T>(xs)) < (T)0 ? 1u : 0u) << g;
            if constexpr (DV >= 1) out[0].x[g] = v2c(l.x[g], sum_ct<DV - 1, 0, ORD, T>(xs));
            if constexpr (DV >= 2) out[1].x[g] = v2c(l.x[g], sum_ct<DV - 1, 1, ORD, T>(xs));
            if constexpr (DV >= 3) out[2].x[g] = v2c(l.x[g], sum_ct<DV - 1, 2, ORD, T>(xs));
            if constexpr (DV >= 4) out[3].x[g] = v2c(l.x[g], sum_ct<DV - 1, 3, ORD, T>(xs));
            if constexpr (DV >= 5) out[4].x[g] = v2c(l.x[g], sum_ct<DV - 1, 4, ORD, T>(xs));
            if constexpr (DV >= 6) out[5].x[g] = v2c(l.x[g], sum_ct<DV - 1, 5, ORD, T>(xs));
            if constexpr (DV >= 7) out[6].x[g] = v2c(l.x[g], sum_ct<DV - 1, 6, ORD, T>(xs));
            if constexpr (DV >= 8) out[7].x[g] = v2c(l.x[g], sum_ct<DV - 1, 7, ORD, T>(xs));
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) lds_store<P>(off[k], out[k]);
        if constexpr ((MODE & 4) != 0) {
            bits_s[q] = (uint8_t)byte;
            if (ps.par_off) {
#pragma unroll
                for (int k = 0; k < DV; ++k) lds_atomic_xor(ps.par_off + (((off[k] >> ps.shift) & ps.mask) << 2), byte);
            }
        }
    } else {
        unsigned byte = 0;
        bool store = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            T xs[DV > 0 ? DV : 1];
#pragma unroll
            for (int k = 0; k < DV; ++k) xs[k] = x[k].x[g];
            const T post = l.x[g] + sum_ct<DV, -1, ORD, T>(xs);
            byte |= (post < (T)0 ? 1u : 0u) << g;
            if ((emask >> g) & 1u) { l.x[g] = post; store = true; }
        }
        if constexpr (MODE == 8) {
            // last pass of a fixed-T decode: the c2v values in this variable's slots are dead once read above, so the
            // hard decisions go there (bit g = codeword g) and the final syndrome reads them back with consecutive
            // addresses per check (res_syndrome_slots) instead of gathering bytes through global index loads
#pragma unroll
            for (int k = 0; k < DV; ++k) lds_store<unsigned>(off[k], byte);
        } else {
            bits_s[q] = (uint8_t)byte;
            if (ps.par_off) {
#pragma unroll
                for (int k = 0; k < DV; ++k) lds_atomic_xor(ps.par_off + (((off[k] >> ps.shift) & ps.mask) << 2), byte);
            }
        }
        if (store) L[q] = l;
    }
}

__device__ __forceinline__ uint4 plan_unpack(const uint2 &p)
{
    return make_uint4(p.x & 0xffffu, p.x >> 16, p.y & 0xffffu, p.y >> 16);
}

template <int G, int MODE, typename T>
__device__ __forceinline__ void res_var_dispatch(unsigned char *smem, T *__restrict__ llr_s,
                                                 uint8_t *__restrict__ bits_s, int q, int dv, const uint4 &slo,
                                                 const uint4 &shi, T a, unsigned emask, const ParScatter &ps)
{
#define LDPC_RV(D) case D: res_var_body<G, D, MODE, T>(smem, llr_s, bits_s, q, slo, shi, a, emask, ps); break;
    switch (dv) {
        LDPC_RV(0) LDPC_RV(1) LDPC_RV(2) LDPC_RV(3) LDPC_RV(4) LDPC_RV(5) LDPC_RV(6) LDPC_RV(7) LDPC_RV(8)
    default: break;   // host admits only max_dv <= 8 to this engine
    }
#undef LDPC_RV
}

// Index data (degree, alpha column, the slot offsets) of the NEXT variable of a lane is fetched from
// global memory (L1/L2 resident, shared by every workgroup) while the current one is processed.
// (A two-register-set ping-pong that avoids the hand-over copies doubled the code and measured no faster.)
template <int G, int MODE, typename T>
__device__ __forceinline__ void res_var_phase(const ResidentPlan &pl, unsigned char *smem,
                                              T *__restrict__ llr_s, uint8_t *__restrict__ bits_s,
                                              const T *__restrict__ alpha_lds,
                                              const T *__restrict__ alpha_glb, unsigned emask,
                                              int tid, int nt, const ParScatter ps = ParScatter{})
{
    const int n = pl.n, n_hi = pl.n_hi;
    // Plan loads are UNCONDITIONAL with clamped indices (a lane past the end re-reads the last entry, a variable of degree <= 4
    // reads the last `hi` entry: same cache lines, values unused): with a fixed number of loads per round the compiler waits for
    // the CURRENT variable's entries with vmcnt(3) and leaves the next one's in flight -- behind `if (q < n_hi)` it could not
    // count them and drained the queue (vmcnt(0)) right after issuing the prefetch, every round.
    const int hi_last = (n_hi > 0 ? n_hi : 1) - 1;
    int q = tid;
    unsigned meta = 0;
    uint2 plo = make_uint2(0, 0), phi = make_uint2(0, 0);           // packed offsets, unpacked at the point of use
    if (q < n) {
        meta = pl.vmeta[q];
        plo = pl.vslot_lo[q];
        phi = pl.vslot_hi[min(q, hi_last)];
    }
    while (q < n) {
        const int qn = q + nt;
        unsigned metan;
        uint2 plon, phin;
#if !LDPC_RES_NO_PLAN_PREFETCH                        // tuning builds with more waves per SIMD trade the prefetch for registers
        {
#if LDPC_RES_PLAN_U32
            // 32-bit byte offsets against the scalar base (global_load ... v_off, s[base]): one shift per load instead of a
            // sign extension and a 64-bit add each
            const unsigned qc = (unsigned)min(qn, n - 1), qh = min(qc, (unsigned)hi_last);
            metan = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pl.vmeta) + (size_t)(qc << 2));
            plon = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(pl.vslot_lo) + (size_t)(qc << 3));
            phin = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(pl.vslot_hi) + (size_t)(qh << 3));
#else
            const int qc = min(qn, n - 1);
            metan = pl.vmeta[qc];
            plon = pl.vslot_lo[qc];
            phin = pl.vslot_hi[min(qc, hi_last)];
#endif
        }
#endif
        const int dv = (int)(meta & 0xffu);
        const uint4 slo = plan_unpack(plo), shi = plan_unpack(phi);
        T a = (T)0;                                                      // LDS copy of the table when small
        if (MODE == 0 || MODE == 4) a = alpha_lds ? alpha_lds[meta >> 8] : alpha_glb[meta >> 8];
        // one scalar branch into the body of the wave's degree; a class-boundary wave (two or three degrees)
        // goes round once per distinct degree with the other lanes masked off
#if LDPC_RES_VAR_MODE == 1
        for (unsigned long long todo = __ballot(true); todo;) {      // uniform loop, scalar degree (see res_check_phase)
            const int dvw = __builtin_amdgcn_readlane(dv, __ffsll((long long)todo) - 1);
            int dsel = dvw;
            asm volatile("" : "+s"(dsel));
            const bool mine = dv == dvw;
            if (mine) res_var_dispatch<G, MODE, T>(smem, llr_s, bits_s, q, dsel, slo, shi, a, emask, ps);
            todo &= ~__ballot(mine);
        }
#else
        // every lane jumps to the compile-time body of ITS degree (exec-masked dispatch): one pass per wave whatever the mix
        res_var_dispatch<G, MODE, T>(smem, llr_s, bits_s, q, dv, slo, shi, a, emask, ps);
#endif
#if LDPC_RES_NO_PLAN_PREFETCH
        {
            const int qc = min(qn, n - 1);
            metan = pl.vmeta[qc];
            plon = pl.vslot_lo[qc];
            phin = pl.vslot_hi[min(qc, hi_last)];
        }
#endif
        q = qn; meta = metan; plo = plon; phi = phin;
    }
}

// H @ bits mod 2 per check from the hard-decision bytes; OR of all checks' parities -> *sh_unsat
template <int G, bool SPLIT>
__device__ __forceinline__ void res_syndrome_phase(const ResidentPlan &pl, const uint8_t *__restrict__ bits_s,
                                                   unsigned *sh_unsat, int tid, int nt)
{
    unsigned acc = 0;
    for (int p = tid; p < pl.m; p += nt) {
        const int dc = pl.dc_s[p];
        int dcw;
        unsigned x = 0;
        if (wave_uniform(dc, dcw)) {
#pragma unroll 8
            for (int t = 0; t < dcw; ++t) x ^= bits_s[pl.cvar[t * pl.mstride + p]];
        } else {
            for (int t = 0; t < dc; ++t) x ^= bits_s[pl.cvar[t * pl.mstride + p]];
        }
        if constexpr (SPLIT) x = group_xor(pl.gsz[p], x);   // a split check's parity is the XOR over its lane group
        acc |= x;
    }
    if (acc) atomicOr(sh_unsat, acc);
}

// syndrome from the parity words the variable lanes scattered into (ParScatter): OR of all checks' parities -> *sh_unsat,
// and the words are cleared for the next iteration
template <int G, bool SPLIT>
__device__ __forceinline__ void res_parity_reduce(const ResidentPlan &pl, unsigned par_off, unsigned *sh_unsat, int tid, int nt)
{
    unsigned acc = 0;
    for (int p = tid; p < pl.m; p += nt) {
        unsigned x = lds_load<unsigned>(par_off + 4u * (unsigned)p);
        lds_store<unsigned>(par_off + 4u * (unsigned)p, 0u);
        if constexpr (SPLIT) x = group_xor(pl.gsz[p], x);
        acc |= x;
    }
    if (acc) atomicOr(sh_unsat, acc);
}

// final syndrome of a fixed-T decode from the hard decisions res_var_body<MODE 8> left in the message slots
template <int G, typename T, bool SPLIT>
__device__ __forceinline__ void res_syndrome_slots(const ResidentPlan &pl, unsigned *sh_unsat, int tid, int nt)
{
    constexpr unsigned kSlot = sizeof(Pack<T, G>);
    unsigned acc = 0;
    for (int p = tid; p < pl.m; p += nt) {
        const int dc = pl.dc_s[p];
        int dcw;
        unsigned x = 0;
        if (wave_uniform(dc, dcw)) {
#pragma unroll 8
            for (int t = 0; t < dcw; ++t) x ^= lds_load<unsigned>((unsigned)(t * pl.mstride + p) * kSlot);
        } else {
            for (int t = 0; t < dc; ++t) x ^= lds_load<unsigned>((unsigned)(t * pl.mstride + p) * kSlot);
        }
        if constexpr (SPLIT) x = group_xor(pl.gsz[p], x);
        acc |= x;
    }
    if (acc) atomicOr(sh_unsat, acc);
}

// ---- outputs ---------------------------------------------------------------------------------------------
// Positions are walked in ORIGINAL variable order j = tid + k*nt (coalesced row stores); the inverse-permutation
// entries of a batch of kEmit positions are loaded before the first LDS read (one round trip per batch), and the
// bit-packed row (the multi-GPU wire format) falls out of a wave ballot over 64 consecutive positions: lanes 0..7
// store the 8 bytes.  `mask`: codewords of the workgroup to write.
constexpr int kEmit = 4;

template <int G>
__device__ __forceinline__ void res_store_packed(const ResidentArgs &a, long long b, int j, int n, bool neg)
{
    const unsigned long long m = __ballot(neg);
    const int lane = threadIdx.x & 63;
    const int base = j - lane;                       // first position of this wave's 64
    if (lane < 8 && base + 8 * lane < n) a.packed[(size_t)b * ((n + 7) / 8) + (base >> 3) + lane] = (uint8_t)(m >> (8 * lane));
}

// posterior of the masked codewords sits in llr_s[.][g] (sorted order)
template <int G, typename T>
__device__ __forceinline__ void res_emit(const ResidentPlan &pl, const ResidentArgs &a, const T *__restrict__ llr_s,
                                         long long b0, unsigned mask, int iters, unsigned unsat, int tid, int nt)
{
    using P = Pack<T, G>;
    T *posterior = reinterpret_cast<T *>(a.posterior);
    const int n = pl.n;
    if (a.posterior || a.bits || a.packed) {
        for (int j0 = tid; j0 < n; j0 += kEmit * nt) {
            unsigned ip[kEmit];
#pragma unroll
            for (int k = 0; k < kEmit; ++k) ip[k] = (j0 + k * nt < n) ? pl.inv_perm_v[j0 + k * nt] : 0u;
#pragma unroll
            for (int k = 0; k < kEmit; ++k) {
                const int j = j0 + k * nt;
                if (j - (tid & 63) >= n) break;                       // the whole wave is past the row (wave-uniform)
                const bool in = j < n;
                const P p = reinterpret_cast<const P *>(llr_s)[ip[k]];  // ip = 0 for lanes past the row: harmless read
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!((mask >> g) & 1u)) continue;
                    if (in && a.posterior) res_stream_store(&posterior[(size_t)(b0 + g) * n + j], p.x[g]);
                    if (in && a.bits) res_stream_store(&a.bits[(size_t)(b0 + g) * n + j], p.x[g] < (T)0 ? 1 : 0);
                    if (a.packed) res_store_packed<G>(a, b0 + g, j, n, in && p.x[g] < (T)0);
                }
            }
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!((mask >> g) & 1u)) continue;
            if (a.iterations) a.iterations[b0 + g] = iters;
            if (a.success) a.success[b0 + g] = (uint8_t)(((unsat >> g) & 1u) ? 0 : 1);
        }
    }
}

// outputs when no posterior was asked for: hard decisions straight from bits_s (sorted order, bit g = codeword g)
template <int G>
__device__ __forceinline__ void res_emit_bits(const ResidentPlan &pl, const ResidentArgs &a, const uint8_t *__restrict__ bits_s,
                                              long long b0, unsigned mask, int iters, unsigned unsat, int tid, int nt)
{
    const int n = pl.n;
    if (a.bits || a.packed) {
        for (int j0 = tid; j0 < n; j0 += kEmit * nt) {
            unsigned ip[kEmit];
#pragma unroll
            for (int k = 0; k < kEmit; ++k) ip[k] = (j0 + k * nt < n) ? pl.inv_perm_v[j0 + k * nt] : 0u;
#pragma unroll
            for (int k = 0; k < kEmit; ++k) {
                const int j = j0 + k * nt;
                if (j - (tid & 63) >= n) break;
                const bool in = j < n;
                const unsigned bb = bits_s[ip[k]];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!((mask >> g) & 1u)) continue;
                    const unsigned bit = (bb >> g) & 1u;
                    if (in && a.bits) res_stream_store(&a.bits[(size_t)(b0 + g) * n + j], (int)bit);
                    if (a.packed) res_store_packed<G>(a, b0 + g, j, n, in && bit != 0);
                }
            }
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!((mask >> g) & 1u)) continue;
            if (a.iterations) a.iterations[b0 + g] = iters;
            if (a.success) a.success[b0 + g] = (uint8_t)(((unsat >> g) & 1u) ? 0 : 1);
        }
    }
}

// test hook: C2V values sitting in the message slots -> dbg_c2v[b][e] for the codewords in `mask`
template <int G, typename T>
__device__ __forceinline__ void res_dump_c2v(const ResidentPlan &pl, const ResidentArgs &a, long long b0, unsigned mask,
                                             int tid, int nt)
{
    using P = Pack<T, G>;
    T *out = reinterpret_cast<T *>(a.dbg_c2v);
    for (int s = tid; s < pl.S; s += nt) {
        const unsigned e = pl.edge_of_slot[s];
        if (e == 0xffffffffu) continue;
        const P v = lds_load<P>((unsigned)s * (unsigned)sizeof(P));
#pragma unroll
        for (int g = 0; g < G; ++g)
            if ((mask >> g) & 1u) out[(size_t)(b0 + g) * pl.E + e] = v.x[g];
    }
}

// LDS carve (bytes): msg at 0, then llr_s, alpha_s, bits_s, the syndrome word
__host__ __device__ inline size_t res_off_llr(int S, int G) { return (size_t)S * G * 4; }
__host__ __device__ inline size_t res_off_alpha(int S, int n, int G) { return res_off_llr(S, G) + (size_t)n * G * 4; }
__host__ __device__ inline size_t res_off_bits(int S, int n, int G, int n_alpha_lds) { return res_off_alpha(S, n, G) + (size_t)n_alpha_lds * 4; }
__host__ __device__ inline size_t res_off_flag(int S, int n, int G, int n_alpha_lds) { return (res_off_bits(S, n, G, n_alpha_lds) + n + 3) / 4 * 4; }
__host__ __device__ inline size_t res_off_par(int S, int n, int G, int n_alpha_lds) { return res_off_flag(S, n, G, n_alpha_lds) + 16; }
// `m_par` parity words follow (early-stop syndrome by scatter); 0 when the stride is not a power of two
__host__ __device__ inline size_t res_lds_total(int S, int n, int G, int n_alpha_lds, int m_par) { return res_off_par(S, n, G, n_alpha_lds) + 4 * (size_t)m_par; }

// ES: 0 = fixed-iteration kernel, 1 = early-stop kernel (kept apart so that the fixed-T kernel does not carry
// the posterior/syndrome/emit code of the stop rule: the extra code cost the hot loop ~4 % when merged)
#ifndef LDPC_RES_MAX_THREADS
#define LDPC_RES_MAX_THREADS 1024      // launch bounds of resident_decode: threads per workgroup, waves per SIMD the
#define LDPC_RES_MIN_WAVES 4           // register allocation must leave room for (tuning builds trade registers for waves)
#endif
template <int G, int FORM, bool BPC, int NL, int MS, int ES, typename T = float, bool SPLIT = false>
__global__ __launch_bounds__(LDPC_RES_MAX_THREADS, LDPC_RES_MIN_WAVES) void resident_decode(ResidentPlan pl, ResidentArgs a)
{
    extern __shared__ __align__(16) unsigned char res_smem[];     // the only LDS object: msg starts at offset 0
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();  // lds_load/lds_store rely on that (folds away)
    // T = float: G codewords per workgroup; T = double (fp64 Basic): the same 8-byte slots hold ONE codeword, so the
    // LDS carve is that of GE = G * sizeof(T) / 4 float codewords.  Table / LLR / posterior pointers of ResidentArgs are
    // typed float for the common case and re-read as T here.
    constexpr int GE = G * (int)(sizeof(T) / 4);
    const int n_alpha_lds = a.alpha_in_lds ? a.T * a.n_alpha * (int)(sizeof(T) / 4) : 0;
    T *llr_s = reinterpret_cast<T *>(res_smem + res_off_llr(pl.S, GE));
    T *alpha_s = reinterpret_cast<T *>(res_smem + res_off_alpha(pl.S, pl.n, GE));
    uint8_t *bits_s = res_smem + res_off_bits(pl.S, pl.n, GE, n_alpha_lds);
    unsigned *sh_unsat = reinterpret_cast<unsigned *>(res_smem + res_off_flag(pl.S, pl.n, GE, n_alpha_lds));
    ParScatter ps, psf;                                  // per-iteration syndrome (early stop) / final syndrome (fixed T)
    if (pl.par_words) {
        psf.par_off = (unsigned)res_off_par(pl.S, pl.n, GE, n_alpha_lds);
        psf.shift = (unsigned)pl.par_shift;
        psf.mask = (unsigned)pl.mstride - 1u;
        if (ES) ps = psf;
    }
    const T *g_llr = reinterpret_cast<const T *>(a.llr);
    const T *g_beta = reinterpret_cast<const T *>(a.beta);
    const T *g_alpha = reinterpret_cast<const T *>(a.alpha);
    const T *g_oms_alpha = reinterpret_cast<const T *>(a.oms_alpha);
    using P = Pack<T, G>;
    const int tid = threadIdx.x, nt = blockDim.x, n = pl.n;
    const long long b0 = (long long)blockIdx.x * G;
    constexpr unsigned kAll = (1u << G) - 1u;
    if (LDPC_PROBE(a, 64)) return;                       // launch-overhead probe

    // LLRs: coalesced rows from HBM, scattered into degree-sorted order; padding codewords get +1.
    // All loads of a batch of kPro positions are issued before the first LDS store: one HBM round trip per
    // batch instead of one per element (the plain loop waits for every load before it issues the next).
    constexpr int kPro = 4;
    if (!LDPC_PROBE(a, 16)) {
        for (int j0 = tid; j0 < n; j0 += kPro * nt) {
            unsigned ip[kPro];
            P v[kPro];
#pragma unroll
            for (int k = 0; k < kPro; ++k) {
                const int j = j0 + k * nt;
                ip[k] = 0;
                if (j < n) {
                    ip[k] = pl.inv_perm_v[j];
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        v[k].x[g] = (b0 + g < a.batch) ? res_stream_load(&g_llr[(size_t)(b0 + g) * n + j]) : (T)1;
                }
            }
#pragma unroll
            for (int k = 0; k < kPro; ++k)
                if (j0 + k * nt < n) reinterpret_cast<P *>(llr_s)[ip[k]] = v[k];
        }
    }
    for (int k = tid; k < (a.alpha_in_lds ? a.T * a.n_alpha : 0); k += nt) alpha_s[k] = g_alpha[k];
    if (tid == 0) { sh_unsat[0] = 0; sh_unsat[1] = 0; }
    for (int k = tid; k < pl.par_words; k += nt) lds_store<unsigned>(psf.par_off + 4u * (unsigned)k, 0u);
    __syncthreads();
    // "initialise v2c with the channel LLRs" (T == 0: c2v = 0, the loop never runs)
    {
        const P *L = reinterpret_cast<const P *>(llr_s);
        for (int q0 = tid; q0 < n && !LDPC_PROBE(a, 32); q0 += kPro * nt) {
            int dvk[kPro];
            uint2 plo[kPro], phi[kPro];
#pragma unroll
            for (int k = 0; k < kPro; ++k) {                       // plan loads of the whole batch first
                const int q = q0 + k * nt;
                dvk[k] = 0;
                plo[k] = make_uint2(0, 0);
                phi[k] = make_uint2(0, 0);
                if (q < n) {
                    dvk[k] = (int)(pl.vmeta[q] & 0xffu);
                    plo[k] = pl.vslot_lo[q];
                    if (q < pl.n_hi) phi[k] = pl.vslot_hi[q];
                }
            }
#pragma unroll
            for (int k = 0; k < kPro; ++k) {
                const int q = q0 + k * nt;
                if (q < n) {
                    const uint4 slo = plan_unpack(plo[k]), shi = plan_unpack(phi[k]);
                    const unsigned off[8] = {slo.x, slo.y, slo.z, slo.w, shi.x, shi.y, shi.z, shi.w};
                    P l = L[q];
                    if constexpr (!std::is_same<T, float>::value) {
#pragma unroll
                        for (int g = 0; g < G; ++g) l.x[g] = __builtin_canonicalize(l.x[g]);   // quiet NaNs (see the fp64 check phase)
                    }
                    if (a.T == 0) {
#pragma unroll
                        for (int g = 0; g < G; ++g) l.x[g] = (T)0;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e < dvk[k]) lds_store<P>(off[e], l);
                }
            }
        }
    }
    // first-round check degree (iteration-invariant) and per-check beta of iteration 0, in registers
    const int dc_pre = tid < pl.m ? pl.dc_s[tid] : 0;
    T b_pre = (BPC && tid < pl.m && a.T > 0) ? g_beta[pl.bslot_c[tid]] : (T)0;
    __syncthreads();

    unsigned done = 0;                                   // block-uniform
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (b0 + g >= a.batch) done |= 1u << g;

    for (int it = 0; it < a.T; ++it) {
        const T *beta_row = g_beta + (size_t)it * a.n_beta;
        const T *oa_row = a.oms_alpha ? g_oms_alpha + (size_t)it * a.n_oms_alpha : nullptr;
        const float *thr = FORM == FORM_RCQ ? a.thr + (size_t)a.q_of_iter[it] * a.n_levels : nullptr;
        const T *alpha_lds = a.alpha_in_lds ? alpha_s + it * a.n_alpha : nullptr;
        const T *alpha_glb = g_alpha + (size_t)it * a.n_alpha;
        if (!LDPC_PROBE(a, 1))
            res_check_phase<G, FORM, BPC, NL, MS, T, SPLIT>(pl, res_smem, beta_row, oa_row, thr, a.n_levels, a.rcq_zero0 != 0,
                                                  dc_pre, b_pre, tid, nt);
        if (BPC && tid < pl.m && it + 1 < a.T)           // next iteration's beta: in flight across the phases below
            b_pre = g_beta[(size_t)(it + 1) * a.n_beta + pl.bslot_c[tid]];
        if (!LDPC_PROBE(a, 128)) __syncthreads();   // probe 128: timing without the two barriers of an iteration (results are then wrong)
        if (ES && !a.posterior) {
            // reference stop rule without a second gather pass: the variable phase also yields this iteration's
            // hard decisions (the posterior shares the gathered C2V values); outputs are bits only
            const bool last = it == a.T - 1;
            if (last) res_var_phase<G, 1, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, 0u, tid, nt, ps);
            else if (a.unit_alpha) res_var_phase<G, 6, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, 0u, tid, nt, ps);
            else res_var_phase<G, 4, T>(pl, res_smem, llr_s, bits_s, alpha_lds, alpha_glb, 0u, tid, nt, ps);
            __syncthreads();
            // the flag word alternates between iterations: this one's is read after ONE barrier while thread 0 already clears
            // the other one for the next iteration (no third barrier)
            unsigned *su = sh_unsat + (it & 1);
            if (ps.par_off) res_parity_reduce<G, SPLIT>(pl, ps.par_off, su, tid, nt);    // the variable lanes scattered the parities
            else res_syndrome_phase<G, SPLIT>(pl, bits_s, su, tid, nt);
            __syncthreads();
            const unsigned unsat = *su;
            if (tid == 0) sh_unsat[(it + 1) & 1] = 0;
            const unsigned newly = ~unsat & ~done & kAll;
            if (newly) {                                 // block-uniform
                res_emit_bits<G>(pl, a, bits_s, b0, newly, it + 1, 0u, tid, nt);
                done |= newly;
                if (done == kAll) return;
                __syncthreads();                         // bits_s is rewritten by the next iteration
            }
            continue;
        }
        if (ES) {
            res_var_phase<G, 1, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, 0u, tid, nt, ps);
            __syncthreads();
            unsigned *su = sh_unsat + (it & 1);
            if (ps.par_off) res_parity_reduce<G, SPLIT>(pl, ps.par_off, su, tid, nt);
            else res_syndrome_phase<G, SPLIT>(pl, bits_s, su, tid, nt);
            __syncthreads();
            const unsigned unsat = *su;
            if (tid == 0) sh_unsat[(it + 1) & 1] = 0;
            const unsigned newly = ~unsat & ~done & kAll;
            if (newly) {                                 // block-uniform
                if (a.dbg_c2v) res_dump_c2v<G, T>(pl, a, b0, newly, tid, nt);      // slots still hold this iteration's C2V
                res_var_phase<G, 1, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, newly, tid, nt);
                __syncthreads();
                res_emit<G, T>(pl, a, llr_s, b0, newly, it + 1, 0u, tid, nt);
                done |= newly;
                if (done == kAll) return;                // every codeword of the block has its outputs
                __syncthreads();
            }
        }
        if (it != a.T - 1 && !LDPC_PROBE(a, 2)) {
            if (a.unit_alpha) res_var_phase<G, 2, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, 0u, tid, nt);
            else res_var_phase<G, 0, T>(pl, res_smem, llr_s, bits_s, alpha_lds, alpha_glb, 0u, tid, nt);
            if (!LDPC_PROBE(a, 128)) __syncthreads();   // probe 128: timing without the two barriers of an iteration (results are then wrong)
        }
    }

    // codewords still open after T iterations: outputs of the last iteration
    const unsigned open = ~done & kAll;
    if (!open) return;
    if (a.dbg_c2v) {                                     // slots hold the C2V of iteration T-1 (zeros when T == 0)
        unsigned real = 0;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (b0 + g < a.batch) real |= 1u << g;
        res_dump_c2v<G, T>(pl, a, b0, open & real, tid, nt);
        __syncthreads();                                 // the final posterior pass overwrites the slots
    }
    if (ES && !a.posterior && a.T > 0) {                 // bits_s hold iteration T's decisions already
        res_emit_bits<G>(pl, a, bits_s, b0, open, a.T, kAll, tid, nt);
        return;
    }
    // fixed-T mode: success = final syndrome is zero.  With parity words (power-of-two stride) the posterior pass scatters the
    // decisions into them (one LDS atomic per edge, then m words are read); otherwise the decisions go into the dead message
    // slots and every check reads its row of them back (MODE 8 / res_syndrome_slots)
    const bool scatter = LDPC_RES_FINAL_SCATTER && !ES && psf.par_off != 0;
    if (!LDPC_PROBE(a, 8)) {
        if (ES) res_var_phase<G, 1, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, open, tid, nt);
        else if (scatter) res_var_phase<G, 1, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, open, tid, nt, psf);
        else res_var_phase<G, 8, T>(pl, res_smem, llr_s, bits_s, (const T *)nullptr, (const T *)nullptr, open, tid, nt);
    }
    __syncthreads();
    unsigned unsat = kAll;
    if (!ES && !LDPC_PROBE(a, 8)) {
        if (scatter) res_parity_reduce<G, SPLIT>(pl, psf.par_off, sh_unsat, tid, nt);
        else res_syndrome_slots<G, T, SPLIT>(pl, sh_unsat, tid, nt);
        __syncthreads();
        unsat = *sh_unsat;
    }
    if (LDPC_PROBE(a, 4)) return;
    res_emit<G, T>(pl, a, llr_s, b0, open, a.T, unsat, tid, nt);
}

}  // namespace ldpc
