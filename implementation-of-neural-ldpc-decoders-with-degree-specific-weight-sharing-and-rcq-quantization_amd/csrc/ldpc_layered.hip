// ldpc_layered.hip -- LDS-resident layered RCQ decode (gfx950): lanes run over the EDGES of one check.
//
// Schedule: RCQMinSumDecoder._decode_layered as the reference executes it (rcq_decoder.py:281-350; its per-check
// message matrix is re-created for every check, so nothing is ever subtracted: posteriors accumulate every check's
// quantised message, check after check, inside and across iterations -- LDPC_SCHED_LAYERED_REF).  The walk over the
// checks is ONE dependent chain per codeword (check i+1 reads what check i wrote), m*T steps long; what can run in
// parallel is the batch and the edges of a check.  The streaming kernel (layered_rcq, ldpc_kernels.hip) keeps the
// posteriors in HBM and gives a lane a codeword: at 65536 codewords that is one wave per SIMD with a memory round trip
// per step.  Here a workgroup is ONE wave holding CW codewords' posterior vectors in LDS (4n bytes each):
//
//   lane = (row, t):  row = lane / LW  -> codeword of the wave,   t = lane % LW -> edge of the current check
//   LW = 2^k >= max check degree (1..64),  CW = 64 / LW codewords per wave (fewer when 4n*CW would exceed LDS)
//
// Per check: one ds_read per lane (posterior of the edge's variable), min1 / min2 / sign parity over the LW lanes by an
// XOR butterfly (DPP quad_perm / row_half_mirror / row_mirror inside 16 lanes, ds_swizzle / shuffle beyond), the
// quantise-reconstruct of sign * min-of-the-others per lane, one ds_write.  LDS instructions of a wave execute in order,
// so the chain needs no barrier; the plan entry of a lane (LDS byte offset of its variable, prefetched kPf checks ahead)
// is the only global-memory access of the loop.  LLRs come straight from the caller's rows and the outputs go straight
// back (no tile transposes, no workspace).  Several single-wave workgroups share a CU (five on the (1998,1512) code).
//
// What a step costs: ~36 instructions, half of them with a DPP operand (the butterfly), issued by ONE wave per SIMD -- a wave
// issues a dependent VALU instruction every ~9 cycles, a DPP one every ~16 with its hazard slot
// (tools/probes/valu_latency_probe.hip) -- so the quantise/select tail is scheduled by hand (ten instructions, no hazard
// padding) and the step is bounded by VALU time, not by the LDS round trip.
//
// A/B form (-DLDPC_LAY_EARLY=1, measured slower at full occupancy, see the macro below): the LDS round trip taken OFF the
// chain where the graph allows it -- the posteriors of check i+1 are read one step
// EARLY (before check i has written), which is exact for every lane whose variable check i does not touch.  The host knows
// which lanes those are; per plan row it records how the row depends on the previous one:
//   NONE  no common variable                         -> the early values are the values
//   FWD   exactly one common variable, sitting in the LAST lane of the previous row and the last-but-one of this row (edges
//         are laid out right-aligned in ascending variable order, so this is the dual-diagonal parity chain of IRA /
//         DVB-S2-like codes: check i = {..., p_i-1, p_i})  -> that lane takes the value the previous step just wrote from
//         its neighbour lane (one DPP move), all others keep the early value
//   LATE  anything else                              -> the row is read again after the previous write (the plain in-order form;
//         decided per GROUP of four rows so that the fast groups carry no test and no branch per step)
// (the plan always carries these flags; the default in-order form ignores them).
//
// Arithmetic is that of layered_rcq (same helpers): results are identical to the streaming kernel's.
#pragma once

#include "ldpc_kernels.hip"

namespace ldpc {

struct LayeredPlan {
    int n, m, lw, cw;              // lanes per check, codewords per wave
    int m_pad;                     // plan rows walked: m rounded up to a multiple of kLayPf with no-op rows (every lane at +inf)
    int has_deg1;                  // some check has exactly one edge (its entries carry bit 31)
    int zero0;                     // every quantiser reconstructs magnitude 0 as 0 (tau_0 == 0, the other thresholds > 0)
    int sorted;                    // tau_1 <= tau_2 <= ... under every quantiser
    int row_shift;                 // > 0: a codeword's LDS region is 2^row_shift bytes (its address is then offset | row bits: one
                                   // v_and_or_b32); 0: (n + 1) * 4 bytes, packed
    const uint32_t *off;           // [m_pad + 2 * kLayPf][lw]  bits 0..28: LDS byte offset (4 * variable) of the edge in lane t of check
                                   //          i (edges right-aligned, ascending variable order); lanes without an
                                   //          edge point at word n of the codeword's vector, which holds +inf for ever (it
                                   //          is neutral for min and parity, and inf + message = inf is written back);
                                   //          bit 31 (kLayFwdBit): THIS lane takes its value from its upper neighbour lane of the
                                   //          previous step (FWD rows: the one receiving lane); bit 30 (kLayLateBit), on every lane
                                   //          of the FIRST row of a group of kLayPf rows: some row of the group must be read after
                                   //          the previous row's write (the whole group then runs in order; group 0 always);
                                   //          bit 29 (kLayDeg1Bit) on the entries of a degree-1 check ("min2 = min", :312-313);
                                   //          no-op rows up to m_pad (inf in, inf out), then 2 * kLayPf more that only the
                                   //          prefetch past the last check reads
};

constexpr int kLayPf = 4;          // plan entries in flight ahead of the check being processed (= the unroll of the walk)
#ifndef LDPC_LAY_EARLY
#define LDPC_LAY_EARLY 0           // 1: early reads + DPP forwarding of the parity chain (header); 0: every row is read after the previous
                                   // write.  Same-box A/B on the (1998,1512) code, 65536 codewords, T = 10
                                   // (profiles/r03_layered_variants.txt): in order 8.87-8.90 ms, early form 9.23-9.25 ms (both 0.66-0.68 ms
                                   // at 4096 codewords, where a wave has its SIMD to itself): a step is bounded by the VALU time of its
                                   // ~19 DPP operations, not by the LDS round trip, and the in-order form's LDS waits are what the
                                   // fifth wave of a CU fills.
#endif
constexpr uint32_t kLayOffMask = 0x1fffffffu;
constexpr uint32_t kLayFwdBit = 0x80000000u, kLayLateBit = 0x40000000u, kLayDeg1Bit = 0x20000000u;

// LDS bytes of one codeword: n posteriors + the +inf word
__host__ __device__ inline size_t lay_row_bytes(int n) { return ((size_t)n + 1) * 4; }

template <int O>
__device__ __forceinline__ unsigned lay_xchg(unsigned v)
{
    // partner lane ^ O for a butterfly whose earlier steps ran in order (1, 2, 4, ...): after steps 1 and 2 the four lanes of
    // a quad agree, so "the other quad of my half row" may be ANY lane of it (row_half_mirror), likewise row_mirror for 8.
    // bound_ctrl: no `old` operand to materialise (every lane is active and every source lane exists)
    if constexpr (O == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (O == 4) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    else if constexpr (O == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);  // row_mirror
    else if constexpr (O == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1f);
    else return (unsigned)__shfl_xor((int)v, 32, 64);
}

// One butterfly step on (min1, min2, XOR of the raw bit patterns).  Magnitudes are non-negative floats, so their BIT
// PATTERNS order like unsigned integers (+inf above every finite value): v_min_u32 / v_max_u32 take the exchanged operand
// through DPP directly and need no canonicalisation.  (A NaN input orders above +inf here; the float compares of the
// streaming kernel skip it -- results agree on every non-NaN input.)
template <int O, int LW>
__device__ __forceinline__ void lay_step(unsigned &m1, unsigned &m2, unsigned &par)
{
    if constexpr (O == 1 && O < LW) {                            // first step: min2 is still +inf on every lane
        const unsigned a = m1;
        m2 = max(a, lay_xchg<1>(a));
        m1 = min(a, lay_xchg<1>(a));
        par ^= lay_xchg<1>(par);
    } else if constexpr (O < LW) {
        const unsigned hi = max(m1, lay_xchg<O>(m1));            // each exchanged value has ONE use: folds into the DPP operand
        const unsigned lo2 = min(m2, lay_xchg<O>(m2));
        par ^= lay_xchg<O>(par);
        m1 = min(m1, lay_xchg<O>(m1));
        m2 = min(hi, lo2);                                       // second smallest of {m1, m2, o1, o2}; ties keep min2 == min1
    }
}
template <int O, int LW>
__device__ __forceinline__ void lay_step_xor(unsigned &x)
{
    if constexpr (O < LW) x ^= lay_xchg<O>(x);
}

__device__ __forceinline__ float lay_lds_ld(unsigned byte_off)
{
    return *(__attribute__((address_space(3))) const float *)(size_t)byte_off;
}
__device__ __forceinline__ void lay_lds_st(unsigned byte_off, float v)
{
    *(__attribute__((address_space(3))) float *)(size_t)byte_off = v;
}

// NL: compile-time level count (4 = bc 3), 0 = run-time (up to 8 in registers, more from global memory)
// ES: early stop (per-iteration syndrome, frozen codewords keep their posteriors); D1: the code has degree-1 checks;
// Z0: magnitude 0 reconstructs to 0 under every quantiser -- the message sign is then the parity of the other signs without
//     the reference's "w < 0" test (it differs only in the sign of an exact zero message, which no later operation observes
//     as a value: x + (+-0) == x, |.|, the compares; same argument as ldpc_resident.hip's per-check quantisation)
// SORTED (NL = 4): tau_1 <= tau_2 <= tau_3 -- the level is then found by a two-deep select tree, and BOTH candidate outputs
//     (reconstruction of min1 and of min2) are formed side by side right after the butterfly; the lane only picks.  On one
//     wave a dependent VALU instruction issues every ~9 cycles and a compare -> select pair costs ~21
//     (tools/probes/valu_latency_probe.hip), so the depth of this tail, not its instruction count, is what a step costs.
template <int LW, int NL, bool ES, bool D1, bool Z0, bool SORTED = false, bool P2 = false>
__global__ __launch_bounds__(kWave) void layered_lds(LayeredPlan pl, const float *__restrict__ llr, long long batch,
                                                     const float *__restrict__ thresholds, int n_levels,
                                                     const int *__restrict__ q_of_iter, int T,
                                                     int *__restrict__ bits, float *__restrict__ posterior,
                                                     int *__restrict__ iterations, uint8_t *__restrict__ success,
                                                     uint8_t *__restrict__ packed)
{
    extern __shared__ __align__(16) unsigned char lay_smem[];       // the only LDS object: posteriors start at offset 0
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();  // lay_lds_ld / lay_lds_st rely on that (folds away)
    const int lane = threadIdx.x;
    const int n = pl.n, m = pl.m_pad, cw = pl.cw;                  // m: plan rows incl. the no-op padding
    const int row = lane / LW, t = lane % LW;
    const long long b0 = (long long)blockIdx.x * cw;
    // lane groups beyond the wave's codewords (cw < 64 / LW: large n) SHADOW the last one: same reads, same state, the same
    // values written to the same addresses -- they must take every decision (frozen, latch) exactly as the group they shadow
    const int row_eff = min(row, cw - 1);
    const bool row_live = b0 + row_eff < batch;                     // this lane's codeword exists (padding rows of the last wave do not)
    const unsigned row_words = P2 ? (1u << pl.row_shift) / 4u : (unsigned)n + 1u;
    const unsigned row_base = (unsigned)row_eff * row_words * 4u;
    // LDS address of a plan entry: P2 -> the row bits are disjoint from the offset bits, mask and combine are one instruction
    auto lds_addr = [&](uint32_t o) { return P2 ? ((o & kLayOffMask) | row_base) : (row_base + (o & kLayOffMask)); };

    // LLRs: the caller's rows, coalesced (all 64 lanes over one row at a time); "posteriors = llr.clone()" (:288)
    for (int r = 0; r < cw; ++r) {
        const bool have = b0 + r < batch;
        const float *src = llr + (size_t)(b0 + r) * n;
        for (int j = lane; j < n; j += kWave)
            lay_lds_st(((unsigned)r * row_words + (unsigned)j) * 4u, have ? __builtin_nontemporal_load(src + j) : 1.0f);
        if (lane == 0) lay_lds_st(((unsigned)r * row_words + (unsigned)n) * 4u, inf_of<float>());
    }
    asm volatile("" ::: "memory");

    unsigned frozen = row_live ? 0u : 1u;                           // row-uniform: this lane's codeword has stopped (or is padding)
    int my_iters = T;
    const uint32_t *plan = pl.off + t;

    // syndrome of the current posteriors: 1 when some check of this lane's codeword is unsatisfied (row-uniform);
    // nothing depends on the previous check here, so the loop pipelines (hard decision = "posterior < 0", :341;
    // the +inf word of a lane without an edge contributes 0)
    auto syndrome = [&]() {
        unsigned unsat = 0;
#pragma unroll 4
        for (int i = 0; i < m; ++i) {
            const uint32_t o = plan[(size_t)i * LW];
            unsigned s = lay_lds_ld(lds_addr(o)) < 0.0f ? 1u : 0u;
            lay_step_xor<1, LW>(s); lay_step_xor<2, LW>(s); lay_step_xor<4, LW>(s);
            lay_step_xor<8, LW>(s); lay_step_xor<16, LW>(s); lay_step_xor<32, LW>(s);
            unsat |= s;
        }
        return unsat & 1u;
    };

    for (int it = 0; it < T; ++it) {
        if (ES && __ballot(frozen == 0u) == 0ull) break;
        const float *thr = thresholds + (size_t)q_of_iter[it] * n_levels;
        float th[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (NL > 0) th[q] = q < NL ? thr[q] : 0.0f;
            else th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
            asm volatile("" : "+v"(th[q]));                         // in VGPRs for the whole walk (a select takes one scalar operand: its mask)
        }
        // one check: o = this lane's plan entry, on = its entry of the NEXT row (whose posteriors are requested here, one step
        // early); xe = this row's early value, upd_prev = what this lane wrote in the previous step
        float xe = 0.0f, upd_prev = 0.0f;
        auto step = [&](uint32_t o, uint32_t on, auto late_tag) {
            constexpr bool kLate = decltype(late_tag)::value;
            const unsigned addr = lds_addr(o);
            // the NEXT row's posteriors first -- before this row's write, a whole step before they are needed (the scheduling
            // barrier keeps the compiler from sinking the request towards its use)
#if LDPC_LAY_EARLY
            const float x_early = xe;
            xe = lay_lds_ld(lds_addr(on));
            __builtin_amdgcn_sched_barrier(0);
            float x = x_early;
#else
            (void)on;
            float x = lay_lds_ld(addr);                             // in order: after the previous row's write
#endif
            if constexpr (!LDPC_LAY_EARLY) {
            } else if constexpr (kLate) {
                // a group with a LATE row (or the first group of an iteration): every row of it is read after the previous
                // write.  The wait sits INSIDE the asm, so the compiler's own counters -- and with them the fast groups --
                // never drain the LDS queue.
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr) : "memory");
            } else if (LW >= 2) {
                // FWD: the flagged lane <- its upper neighbour's value of the previous step (row_shl:1)
                const float f = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(upd_prev), 0x101, 0xf, 0xf, true));
                x = ((int)o < 0) ? f : x;
            }
            const unsigned xb = __float_as_uint(x), a = xb & 0x7fffffffu;
            unsigned m1 = a, m2 = 0x7f800000u, par = xb;
            lay_step<1, LW>(m1, m2, par); lay_step<2, LW>(m1, m2, par); lay_step<4, LW>(m1, m2, par);
            lay_step<8, LW>(m1, m2, par); lay_step<16, LW>(m1, m2, par); lay_step<32, LW>(m1, m2, par);
            if (D1 && (o & kLayDeg1Bit)) m2 = m1;                   // degree-1 check: "min2_val = min_val" (:312-313)
            float msg;
            if constexpr (NL == 4 && SORTED && Z0) {
                // raw = min over the OTHER edges (arg-min edge takes min2; ties make min2 == min1); rec = tau[last q with raw >=
                // tau_q]; sorted thresholds: c3 => c2 => c1, so rec = c2 ? (c3 ? tau3 : tau2) : (c1 ? tau1 : tau0).  Scheduled by
                // hand -- ten instructions, every compare two or more instructions ahead of the select that reads its mask; the
                // compiler routes every compare/select pair through vcc with an s_nop each.  (One wave issues an instruction
                // every 6-9 cycles whatever it is -- tools/probes/valu_latency_probe.hip -- so a step costs its instruction
                // COUNT: forming both candidates side by side was measured slower than this.)
                float rec, raw, lo, hi;
                unsigned long long ca, cb;
                unsigned sgn;                                        // par ^ x: formed inside the block, in a hazard slot
                asm("v_cmp_eq_u32_e32 vcc, %[a], %[m1]\n\t"
                    "v_xor_b32_e32 %[sgn], %[par], %[xb]\n\t"
                    "s_nop 0\n\t"
                    "v_cndmask_b32_e32 %[raw], %[m1], %[m2], vcc\n\t"
                    "v_cmp_le_f32_e64 %[ca], %[t1], %[raw]\n\t"
                    "v_cmp_le_f32_e64 %[cb], %[t3], %[raw]\n\t"
                    "v_cmp_le_f32_e32 vcc, %[t2], %[raw]\n\t"
                    "v_cndmask_b32_e64 %[lo], %[t0], %[t1], %[ca]\n\t"
                    "v_cndmask_b32_e64 %[hi], %[t2], %[t3], %[cb]\n\t"
                    "v_cndmask_b32_e32 %[rec], %[lo], %[hi], vcc"
                    : [rec] "=v"(rec), [raw] "=&v"(raw), [lo] "=&v"(lo), [hi] "=&v"(hi), [sgn] "=&v"(sgn), [ca] "=&s"(ca), [cb] "=&s"(cb)
                    : [t0] "v"(th[0]), [t1] "v"(th[1]), [t2] "v"(th[2]), [t3] "v"(th[3]), [m1] "v"(m1), [m2] "v"(m2), [a] "v"(a),
                      [par] "v"(par), [xb] "v"(xb)
                    : "vcc");
                msg = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(rec), sgn, 0x80000000u, 0x78));   // a ^ (b & c)
            } else {
            const float raw = __uint_as_float((a == m1) ? m2 : m1);  // arg-min edge; ties make min2 == min1
            float rec;
            if constexpr (NL > 0) {
                rec = th[0];
#pragma unroll
                for (int q = 1; q < NL; ++q) rec = (raw >= th[q]) ? th[q] : rec;
            } else if (n_levels <= 8) {
                rec = th[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) rec = (raw >= th[q]) ? th[q] : rec;      // NaN padding never matches
            } else {
                rec = thr[0];
                for (int q = 1; q < n_levels; ++q) rec = (raw >= thr[q]) ? thr[q] : rec;
            }
            // message = (1 - 2*sign_bit) * tau[level] (:107-119), sign_bit = (sign * raw < 0): the parity of the OTHER edges'
            // sign bits (bit 31 of par ^ x), counted only for a non-zero magnitude
            if constexpr (Z0) {
                msg = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(rec), par ^ xb, 0x80000000u, 0x78));   // a ^ (b & c)
            } else {
                const unsigned neg = (raw > 0.0f) ? ((par ^ xb) & 0x80000000u) : 0u;
                msg = __uint_as_float(__float_as_uint(rec) ^ neg);
            }
            }
            float upd = x + msg;                                    // "posteriors[j] += c2v_messages[i, j]" (:337-338)
            if (ES) upd = frozen ? x : upd;                         // a stopped codeword keeps its posteriors
            lay_lds_st(addr, upd);
            upd_prev = upd;
        };
        // Plan entries are requested a group ahead: rows 1.. of the NEXT group, and row 0 of the group after it (row 0 of the
        // next group is needed already by the LAST step of this one, for its early read -- it was requested a group ago).
        // m is a multiple of kLayPf and 2 * kLayPf more rows follow: no bounds tests.
        uint32_t cur[kLayPf], nxt[kLayPf], nn0;
#pragma unroll
        for (int k = 0; k < kLayPf; ++k) cur[k] = plan[(size_t)k * LW];
        nxt[0] = plan[(size_t)kLayPf * LW];
        for (int i0 = 0; i0 < m; i0 += kLayPf) {
            const uint32_t *nx = plan + (size_t)(i0 + kLayPf) * LW;
#pragma unroll
            for (int k = 1; k < kLayPf; ++k) nxt[k] = nx[(size_t)k * LW];
            nn0 = nx[(size_t)kLayPf * LW];
            // one wave-uniform test per GROUP: the host flags the first row of a group that holds a LATE row (and group 0:
            // its first row follows the previous iteration's last check); such a group runs the in-order form
            if (LDPC_LAY_EARLY && (__builtin_amdgcn_readfirstlane(cur[0]) & kLayLateBit)) {
#pragma unroll
                for (int k = 0; k < kLayPf; ++k) step(cur[k], k + 1 < kLayPf ? cur[k + 1] : nxt[0], std::true_type{});
            } else {
#pragma unroll
                for (int k = 0; k < kLayPf; ++k) step(cur[k], k + 1 < kLayPf ? cur[k + 1] : nxt[0], std::false_type{});
            }
#pragma unroll
            for (int k = 0; k < kLayPf; ++k) cur[k] = nxt[k];
            nxt[0] = nn0;
        }
        if (ES) {
            const unsigned unsat = syndrome();
            if (frozen == 0u && unsat == 0u) { frozen = 1u; my_iters = it + 1; }      // first zero-syndrome iteration latches (:344-345)
        }
    }
    asm volatile("" ::: "memory");

    // early stop: success = latched at a zero syndrome, iterations = that iteration (else T, :348-349);
    // fixed T: success = the final syndrome is zero, iterations = T
    unsigned ok;
    if (ES) ok = (row_live && frozen != 0u) ? 1u : 0u;
    else ok = syndrome() == 0u ? 1u : 0u;
    if (row_live && row < cw && t == 0) {
        if (iterations) iterations[b0 + row] = (ES && ok) ? my_iters : T;
        if (success) success[b0 + row] = (uint8_t)ok;
    }
    // outputs straight into the caller's rows, coalesced
    const int nbytes = (n + 7) / 8;
    for (int r = 0; r < cw; ++r) {
        if (b0 + r >= batch) break;
        const size_t ob = (size_t)(b0 + r) * n;
        for (int j0 = 0; j0 < n; j0 += kWave) {
            const int j = j0 + lane;
            const bool in = j < n;
            const float v = in ? lay_lds_ld(((unsigned)r * row_words + (unsigned)j) * 4u) : 0.0f;
            const bool neg = in && v < 0.0f;
            if (in && posterior) __builtin_nontemporal_store(v, posterior + ob + j);
            if (in && bits) __builtin_nontemporal_store(neg ? 1 : 0, bits + ob + j);
            if (packed) {
                const unsigned long long mk = __ballot(neg);
                if (lane < 8 && j0 + 8 * lane < n) packed[(size_t)(b0 + r) * nbytes + (j0 >> 3) + lane] = (uint8_t)(mk >> (8 * lane));
            }
        }
    }
}

}  // namespace ldpc
