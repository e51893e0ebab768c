// ldpc_kernels.hip -- gfx950 (MI355X) kernels of the batched flooding decoder.
//
// Mapping (DESIGN.md "Data layout"): a wavefront is 64 lanes x VEC consecutive
// codewords sitting on ONE node of the Tanner graph.  Messages live in HBM as
// [tile][edge][W] with W = 64*VEC codewords innermost, so every message access of a
// wave is one contiguous W*sizeof(T) row (1 KiB for fp32/VEC=4) and every graph
// index, weight and threshold is wave-uniform (scalar loads, SGPRs).  Both sweeps
// are therefore coalesced; the CSC permutation of the variable sweep only selects
// WHICH row.  No MFMA: there is no contraction here, the kernels are HBM streams.
//
// Arithmetic follows the reference exactly: unfused multiply/add
// (-ffp-contract=off), torch.sum's fp32 association order / np.sum's fp64 order for
// the variable sums, first-minimum arg-min, float32 thresholds.  The product of the
// other edges' signs is kept as sign-bit parity (value-equivalent to the reference's
// sign(0)=0 product for the NMS/RCQ rules; the OMS rule tracks zeros explicitly).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ldpc {

constexpr int kWave = 64;
constexpr int kBlock = 256;                 // 4 waves: 4 consecutive nodes of one tile
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kWideCheck = 32;              // checks of higher degree go to cn_sweep_wide (one check per block)

enum { FORM_NMS = 0, FORM_RCQ = 1, FORM_OMS = 2 };

struct GraphDev {
    int n, m, E;
    const int *check_ptr;   // [m+1]
    const int *var_idx;     // [E]  variable of CSR edge
    const int *var_ptr;     // [n+1]
    const int *csc_edge;    // [E]  CSR edge id of k-th edge of a variable
};

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T x[V];
};

// Message rows are streamed exactly once per sweep (a sweep moves GBs, far beyond the 256 MiB
// Infinity Cache), so loads/stores may carry the non-temporal hint; knobs for A/B timing.
#ifndef LDPC_NT_LOAD
#define LDPC_NT_LOAD 1
#endif
#ifndef LDPC_NT_STORE
#define LDPC_NT_STORE 1
#endif
#ifndef LDPC_CN_UNROLL
#define LDPC_CN_UNROLL 4
#endif
#ifndef LDPC_GATHER_GRP
#define LDPC_GATHER_GRP 2      // edges per load group of the fused RCQ iteration kernel (cn_gather); measured on the
                               // (16200,7200) code: 2 edges / 5 waves 2.22 ms per launch, 3/4: 2.40, 4/4: 2.43, 4/3: 2.87,
                               // 6/2: 3.85, 8/2: 3.94 (occupancy beats deeper groups: the kernel is issue- and latency-bound)
#endif
#ifndef LDPC_GATHER_WAVES
#define LDPC_GATHER_WAVES 5    // waves per SIMD the register allocation of cn_gather must leave room for
#endif

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld(const T *p)
{
    typedef T VT __attribute__((ext_vector_type(V)));
    union { VT v; Pack<T, V> k; } u;
#if LDPC_NT_LOAD
    u.v = __builtin_nontemporal_load(reinterpret_cast<const VT *>(p));
#else
    u.v = *reinterpret_cast<const VT *>(p);
#endif
    return u.k;
}
template <typename T, int V>
__device__ __forceinline__ void st(T *p, const Pack<T, V> &v)
{
    typedef T VT __attribute__((ext_vector_type(V)));
    union { VT v; Pack<T, V> k; } u;
    u.k = v;
#if LDPC_NT_STORE
    __builtin_nontemporal_store(u.v, reinterpret_cast<VT *>(p));
#else
    *reinterpret_cast<VT *>(p) = u.v;
#endif
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename T> struct Bits;
template <> struct Bits<float> {
    using U = uint32_t;
    static __device__ __forceinline__ U get(float f) { return __float_as_uint(f); }
    static __device__ __forceinline__ float put(U u) { return __uint_as_float(u); }
    static constexpr int kSignShift = 31;
};
template <> struct Bits<double> {
    using U = uint64_t;
    static __device__ __forceinline__ U get(double f) { return (U)__double_as_longlong(f); }
    static __device__ __forceinline__ double put(U u) { return __longlong_as_double((long long)u); }
    static constexpr int kSignShift = 63;
};

template <typename T>
__device__ __forceinline__ unsigned signbit_of(T v)
{
    return (unsigned)(Bits<T>::get(v) >> Bits<T>::kSignShift);
}
template <typename T>
__device__ __forceinline__ T flip_sign(T v, unsigned neg)
{
    using U = typename Bits<T>::U;
    return Bits<T>::put(Bits<T>::get(v) ^ ((U)neg << Bits<T>::kSignShift));
}
template <typename T> __device__ __forceinline__ T abs_of(T v);
template <> __device__ __forceinline__ float abs_of<float>(float v) { return __builtin_fabsf(v); }
template <> __device__ __forceinline__ double abs_of<double>(double v) { return __builtin_fabs(v); }
template <typename T> __device__ __forceinline__ T inf_of();
template <> __device__ __forceinline__ float inf_of<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_of<double>() { return __builtin_huge_val(); }

// per-lane "frozen" flags of the VEC codewords of this lane (early-stop latch)
template <int VEC>
struct Frozen {
    unsigned bits;   // bit c = codeword c of this lane is done
    __device__ __forceinline__ bool all() const { return bits == ((1u << VEC) - 1u); }
    __device__ __forceinline__ bool none() const { return bits == 0; }
    __device__ __forceinline__ bool one(int c) const { return (bits >> c) & 1u; }
};

// Returns true when every codeword of the wave is frozen (wave-uniform).
template <int VEC>
__device__ __forceinline__ bool load_frozen(const uint64_t *done, int tile, int lane, Frozen<VEC> &f)
{
    f.bits = 0;
    if (!done) return false;
    bool all = true;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        uint64_t w = done[(size_t)tile * VEC + c];
        all = all && (w == ~0ull);
        f.bits |= (unsigned)((w >> lane) & 1ull) << c;
    }
    return all;
}

template <typename T, int VEC>
__device__ __forceinline__ void store_masked(T *p, const Pack<T, VEC> &v, const Frozen<VEC> &f)
{
    if (f.none()) {
        st<T, VEC>(p, v);
    } else if (!f.all()) {
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            if (!f.one(c)) p[c] = v.x[c];
    }
}

// Training forward (ldpc_decode_saving): every iteration writes its own slice, so a stopped codeword's latched
// values are carried forward from the previous iteration's slice instead of being left in place.
template <typename T, int VEC>
__device__ __forceinline__ void store_latched(T *p, const T *prev, Pack<T, VEC> v, const Frozen<VEC> &f)
{
    if (!prev) {
        store_masked<T, VEC>(p, v, f);
        return;
    }
    if (!f.none()) {
        const Pack<T, VEC> old = ld<T, VEC>(prev);
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            if (f.one(c)) v.x[c] = old.x[c];
    }
    st<T, VEC>(p, v);
}

// ------------------------------------------------------------------------------------------
// Check-node (CN -> VN) sweep.  One wave = one check x W codewords.
//   pass 1: stream the dc incoming rows, keep min1/min2/arg-min and the sign bits
//   pass 2: emit dc outgoing rows  (fp rows, or 1-byte RCQ codes)
// FIRST: iteration 0 reads llr[var] instead of v2c (the reference's "initialize with
// channel LLRs", neural_2d_decoder.py:153-157, folded into the first sweep).
// ------------------------------------------------------------------------------------------
// BPC (RCQ only): one beta per check -- only two outgoing (magnitude, sign) pairs exist per codeword, so the
// multiply and the quantiser run twice per check and every edge just picks one of four precomputed codes.
// CPW: consecutive checks one wave walks.  2 amortises the wave prologue (latch word, thresholds, row bases) on
// graphs of small check degree (+3 % on the dc 4-7 code); on the dc ~ 13 code it costs 2-6 %, so the host picks.
template <typename T, int VEC, int FORM, bool FIRST, int NL = 0, bool BPC = false, int CPW = 1>
__global__ __launch_bounds__(kBlock) void cn_sweep(GraphDev g, const T *__restrict__ src,
                                                   void *__restrict__ c2v_out,
                                                   const T *__restrict__ beta_row,
                                                   const int *__restrict__ beta_slot,
                                                   const float *__restrict__ thr, int n_levels,
                                                   const T *__restrict__ oms_alpha_row,
                                                   const int *__restrict__ oms_alpha_slot,
                                                   const uint64_t *__restrict__ done, int check_blocks,
                                                   const void *__restrict__ prev_out = nullptr, int skip_wide = 0)
{
    constexpr int W = kWave * VEC;
    using OutT = typename std::conditional<FORM == FORM_RCQ, uint8_t, T>::type;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / check_blocks);
    const int ibase = uni(((blockIdx.x % check_blocks) * kWavesPerBlock + (threadIdx.x >> 6)) * CPW);
    if (ibase >= g.m) return;

    Frozen<VEC> fz;
    const bool all_frozen = load_frozen<VEC>(done, tile, lane, fz);
    if (all_frozen && !prev_out) return;
    float th[8];
    if (FORM == FORM_RCQ) {
#pragma unroll
        for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
    }
    const size_t lane_off = (size_t)lane * VEC;

#pragma unroll
    for (int cc_ = 0; cc_ < CPW; ++cc_) {
    const int i = ibase + cc_;
    if (i >= g.m) break;
    const int e0 = uni(g.check_ptr[i]);
    const int dc = uni(g.check_ptr[i + 1]) - e0;
    if (dc == 0) continue;
    if (skip_wide && dc > kWideCheck) continue;    // wide checks are split over the waves of a block by cn_sweep_wide
    if (all_frozen) {                              // saving mode: carry the tile's latched rows into this slice
        const size_t off = ((size_t)tile * g.E + e0) * W + (size_t)lane * VEC;
        for (int t = 0; t < dc; ++t)
            st<OutT, VEC>(reinterpret_cast<OutT *>(c2v_out) + off + (size_t)t * W,
                          ld<OutT, VEC>(reinterpret_cast<const OutT *>(prev_out) + off + (size_t)t * W));
        continue;
    }
    const T *in_base = FIRST ? src + (size_t)tile * g.n * W + lane_off
                             : src + ((size_t)tile * g.E + e0) * W + lane_off;

    T m1[VEC], m2[VEC];
    int idx[VEC];
    uint32_t sm[VEC], zm[VEC];
    unsigned par[VEC], nz[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        m1[c] = inf_of<T>(); m2[c] = inf_of<T>(); idx[c] = 0; sm[c] = 0; zm[c] = 0; par[c] = 0; nz[c] = 0;
    }

#pragma unroll LDPC_CN_UNROLL
    for (int t = 0; t < dc; ++t) {
        const T *row = FIRST ? in_base + (size_t)g.var_idx[e0 + t] * W : in_base + (size_t)t * W;
        Pack<T, VEC> v = ld<T, VEC>(row);
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            T a = abs_of<T>(v.x[c]);
            unsigned sb = signbit_of<T>(v.x[c]);
            par[c] ^= sb;
            sm[c] |= sb << (t & 31);
            if (FORM == FORM_OMS) {
                unsigned z = (a == (T)0) ? 1u : 0u;
                nz[c] += z;
                zm[c] |= z << (t & 31);
            }
            if (a < m1[c]) { m2[c] = m1[c]; m1[c] = a; idx[c] = t; }
            else if (a < m2[c]) { m2[c] = a; }
        }
    }
    if (dc == 1) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) m2[c] = m1[c];   // "min2_val = min_val" (neural_2d_decoder.py:181-182)
    }

    const bool wide = dc > 32;   // sign masks hold 32 edges; wider checks re-read their inputs
    OutT *out_base = reinterpret_cast<OutT *>(c2v_out) + ((size_t)tile * g.E + e0) * W + lane_off;
    const OutT *prev_base = prev_out ? reinterpret_cast<const OutT *>(prev_out) + ((size_t)tile * g.E + e0) * W + lane_off : nullptr;

    if constexpr (FORM == FORM_RCQ && BPC) {
        const float b = (float)beta_row[beta_slot[e0]];
        auto level = [&](float mag) {
            int lvl = 0;
            if constexpr (NL > 0) {
#pragma unroll
                for (int q = 1; q < NL; ++q) lvl = (mag >= th[q]) ? q : lvl;
            } else if (n_levels <= 8) {
#pragma unroll
                for (int q = 1; q < 8; ++q) lvl = (mag >= th[q]) ? q : lvl;
            } else {
                for (int q = 1; q < n_levels; ++q) lvl = (mag >= thr[q]) ? q : lvl;
            }
            return lvl;
        };
        // cc[k][c]: byte 0 = code of candidate k when the other signs multiply to +, byte 1 when to -
        // (code = (w < 0) * L + level(|w|), w = +-(b * min): the sign bit counts only for a non-zero magnitude)
        unsigned cc1[VEC], cc2[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const float p1 = b * (float)m1[c], p2 = b * (float)m2[c];
            const float a1 = __builtin_fabsf(p1), a2 = __builtin_fabsf(p2);
            const unsigned l1 = (unsigned)level(a1), l2 = (unsigned)level(a2);
            const unsigned s1 = signbit_of<float>(p1), s2 = signbit_of<float>(p2);
            const unsigned z1 = a1 > 0.0f ? (unsigned)n_levels : 0u, z2 = a2 > 0.0f ? (unsigned)n_levels : 0u;
            cc1[c] = (l1 + (s1 ? z1 : 0u)) | ((l1 + (s1 ? 0u : z1)) << 8);
            cc2[c] = (l2 + (s2 ? z2 : 0u)) | ((l2 + (s2 ? 0u : z2)) << 8);
        }
#pragma unroll LDPC_CN_UNROLL
        for (int t = 0; t < dc; ++t) {
            Pack<T, VEC> re;
            if (wide) {
                const T *row = FIRST ? in_base + (size_t)g.var_idx[e0 + t] * W : in_base + (size_t)t * W;
                re = ld<T, VEC>(row);
            }
            Pack<OutT, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const unsigned own = wide ? signbit_of<T>(re.x[c]) : ((sm[c] >> (t & 31)) & 1u);
                const unsigned neg = par[c] ^ own;
                const unsigned cc = (t == idx[c]) ? cc2[c] : cc1[c];
                o.x[c] = (OutT)((cc >> (neg * 8u)) & 0xffu);
            }
            store_latched<OutT, VEC>(out_base + (size_t)t * W, prev_base ? prev_base + (size_t)t * W : nullptr, o, fz);
        }
        continue;
    }
#pragma unroll LDPC_CN_UNROLL
    for (int t = 0; t < dc; ++t) {
        const T b = beta_row[beta_slot[e0 + t]];
        T oa = (T)0;
        if (FORM == FORM_OMS && oms_alpha_row) oa = oms_alpha_row[oms_alpha_slot[e0 + t]];
        Pack<T, VEC> re;
        if (wide) {
            const T *row = FIRST ? in_base + (size_t)g.var_idx[e0 + t] * W : in_base + (size_t)t * W;
            re = ld<T, VEC>(row);
        }
        Pack<OutT, VEC> o;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const T raw = (t == idx[c]) ? m2[c] : m1[c];
            unsigned own = wide ? signbit_of<T>(re.x[c]) : ((sm[c] >> (t & 31)) & 1u);
            unsigned neg = par[c] ^ own;
            if (FORM == FORM_NMS) {
                o.x[c] = (OutT)flip_sign<T>(b * raw, neg);
            } else if (FORM == FORM_OMS) {
                unsigned ownz = wide ? ((re.x[c] == (T)0) ? 1u : 0u) : ((zm[c] >> (t & 31)) & 1u);
                bool nonzero = (nz[c] - ownz) == 0;         // no OTHER edge carries a zero
                T d = raw - b;
                T r = d > (T)0 ? d : (T)0;
                T val = r - oa;
                o.x[c] = (OutT)(nonzero ? flip_sign<T>(val, neg) : (T)0);
            } else {
                float w = flip_sign<float>((float)(b * raw), neg);
                float mag = __builtin_fabsf(w);
                // rcq_decoder.py:79-85: level = last q with mag >= tau_q, default 0 -- the q = 0
                // comparison cannot change the outcome and is left out
                int lvl = 0;
                if constexpr (NL > 0) {                                   // compile-time level count (bc = 3: 4)
#pragma unroll
                    for (int q = 1; q < NL; ++q) lvl = (mag >= th[q]) ? q : lvl;
                } else if (n_levels <= 8) {
#pragma unroll
                    for (int q = 1; q < 8; ++q) lvl = (mag >= th[q]) ? q : lvl;       // NaN padding never matches
                } else {
                    for (int q = 1; q < n_levels; ++q) lvl = (mag >= thr[q]) ? q : lvl;
                }
                int code = ((w < 0.0f) ? n_levels : 0) + lvl;                         // :88-89
                o.x[c] = (OutT)code;
            }
        }
        store_latched<OutT, VEC>(out_base + (size_t)t * W, prev_base ? prev_base + (size_t)t * W : nullptr, o, fz);
    }
    }   // checks of this wave
}

// ------------------------------------------------------------------------------------------
// Check-node sweep, fp32 normalised min-sum, 256-codeword tiles, check degrees <= 16 -- the form the benchmark codes run
// (BASELINE config 2 on the streaming engine: the north_star's CN->VN sweep).  Same wave mapping and the same values as
// cn_sweep, written the way cn_sweep_q4 is: one straight-line body per degree, ALL rows of the check requested before the
// first use and held in registers for both passes (13.5 KB in flight per wave instead of one 1 KB row at a time, no row is
// read twice, no sign / arg-min masks), and branch-free arithmetic: min1 / min2 by v_med3 (ties make min2 == min1, so
// "|x| == min1" picks the arg-min edge's output exactly as the first-index arg-min does), the sign product as an XOR of the
// raw bit patterns, the edge's own sign folded in with one v_bitop3.  BPC (one beta per check): the two scaled magnitudes
// are formed once per check.  The generic cn_sweep keeps every other case (fp64, RCQ, OMS, wider checks, the saving forward).
// ------------------------------------------------------------------------------------------
template <int DC, bool FIRST, bool BPC, bool ES>
__device__ __forceinline__ void cn_f4_check(const GraphDev &g, int e0, const float *__restrict__ in_base,
                                            float *__restrict__ out_base, const float *__restrict__ beta_row,
                                            const int *__restrict__ beta_slot, const Frozen<4> &fz)
{
    constexpr int VEC = 4, W = kWave * VEC;
    Pack<float, VEC> v[DC];
#pragma unroll
    for (int t = 0; t < DC; ++t)
        v[t] = ld<float, VEC>(FIRST ? in_base + (size_t)g.var_idx[e0 + t] * W : in_base + (size_t)t * W);
    float m1[VEC], m2[VEC];
    uint32_t sacc[VEC];
    float ninf = -inf_of<float>();
    asm volatile("" : "+v"(ninf));                    // opaque to constant folding: min as ONE v_med3
#pragma unroll
    for (int c = 0; c < VEC; ++c) { m1[c] = inf_of<float>(); m2[c] = inf_of<float>(); sacc[c] = 0; }
#pragma unroll
    for (int t = 0; t < DC; ++t) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const float a = __builtin_fabsf(v[t].x[c]);
            sacc[c] ^= __float_as_uint(v[t].x[c]);
            m2[c] = __builtin_amdgcn_fmed3f(a, m1[c], m2[c]);
            m1[c] = __builtin_amdgcn_fmed3f(a, m1[c], ninf);
        }
    }
    if (DC == 1) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) m2[c] = m1[c];   // "min2_val = min_val" (neural_2d_decoder.py:181-182)
    }
    if constexpr (BPC) {
        const float b = beta_row[beta_slot[e0]];
        uint32_t o1[VEC], o2[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const uint32_t par = sacc[c] & 0x80000000u;
            o1[c] = __float_as_uint(b * m1[c]) ^ par;
            o2[c] = __float_as_uint(b * m2[c]) ^ par;
        }
#pragma unroll
        for (int t = 0; t < DC; ++t) {
            Pack<float, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const uint32_t sel = (__builtin_fabsf(v[t].x[c]) == m1[c]) ? o2[c] : o1[c];
                o.x[c] = __uint_as_float(__builtin_amdgcn_bitop3_b32(sel, __float_as_uint(v[t].x[c]), 0x80000000u, 0x78));   // sel ^ (x & sign)
            }
            if constexpr (ES) store_masked<float, VEC>(out_base + (size_t)t * W, o, fz);
            else st<float, VEC>(out_base + (size_t)t * W, o);
        }
    } else {
#pragma unroll
        for (int t = 0; t < DC; ++t) {
            const float b = beta_row[beta_slot[e0 + t]];
            Pack<float, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const float raw = (__builtin_fabsf(v[t].x[c]) == m1[c]) ? m2[c] : m1[c];
                const uint32_t flip = (sacc[c] ^ __float_as_uint(v[t].x[c])) & 0x80000000u;       // parity of the OTHER signs
                o.x[c] = __uint_as_float(__float_as_uint(b * raw) ^ flip);
            }
            if constexpr (ES) store_masked<float, VEC>(out_base + (size_t)t * W, o, fz);
            else st<float, VEC>(out_base + (size_t)t * W, o);
        }
    }
}

#ifndef LDPC_CNF4_WAVES
#define LDPC_CNF4_WAVES 5          // waves per SIMD the register allocation leaves room for (16 rows x 4 floats are held per lane)
#endif
template <bool FIRST, bool BPC, bool ES>
__global__ __launch_bounds__(kBlock, LDPC_CNF4_WAVES) void cn_sweep_f4(GraphDev g, const float *__restrict__ src, float *__restrict__ c2v_out,
                                                      const float *__restrict__ beta_row, const int *__restrict__ beta_slot,
                                                      const uint64_t *__restrict__ done, int check_blocks)
{
    constexpr int VEC = 4, W = kWave * VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / check_blocks);
    const int i = uni((blockIdx.x % check_blocks) * kWavesPerBlock + (threadIdx.x >> 6));
    if (i >= g.m) return;
    Frozen<VEC> fz;
    fz.bits = 0;
    if constexpr (ES) { if (load_frozen<VEC>(done, tile, lane, fz)) return; }
    const int e0 = uni(g.check_ptr[i]);
    const int dc = uni(g.check_ptr[i + 1]) - e0;
    const size_t lane_off = (size_t)lane * VEC;
    const float *in_base = FIRST ? src + (size_t)tile * g.n * W + lane_off : src + ((size_t)tile * g.E + e0) * W + lane_off;
    float *out_base = c2v_out + ((size_t)tile * g.E + e0) * W + lane_off;
#define LDPC_CF_CASE(D) case D: cn_f4_check<D, FIRST, BPC, ES>(g, e0, in_base, out_base, beta_row, beta_slot, fz); break;
    switch (dc) {
        LDPC_CF_CASE(1) LDPC_CF_CASE(2) LDPC_CF_CASE(3) LDPC_CF_CASE(4) LDPC_CF_CASE(5) LDPC_CF_CASE(6) LDPC_CF_CASE(7) LDPC_CF_CASE(8)
        LDPC_CF_CASE(9) LDPC_CF_CASE(10) LDPC_CF_CASE(11) LDPC_CF_CASE(12) LDPC_CF_CASE(13) LDPC_CF_CASE(14) LDPC_CF_CASE(15)
        LDPC_CF_CASE(16)
    default: break;                                   // dc == 0: nothing to do; dc > 16: the host does not launch this kernel
    }
#undef LDPC_CF_CASE
}

// ------------------------------------------------------------------------------------------
// Check-node sweep for WIDE checks (degree > kWideCheck): lanes still run over codewords, but the check's edges are
// split over the four waves of the block -- wave w streams the rows of its quarter, keeping min1 / min2 / first
// arg-min / sign parity / zero count of that quarter; the partials meet in LDS, every wave combines the four in
// edge order (so the first-minimum rule and the tie behaviour are those of one sequential scan) and emits its own
// quarter.  A quarter of at most 32 edges keeps its sign / zero masks in registers: checks up to degree 128 are
// read exactly once (the one-wave kernel re-reads every input row of a check wider than 32); beyond that a quarter
// re-reads its rows in pass 2.  One block = one (tile, wide check).  Any C2V rule, same arithmetic as cn_sweep.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int FORM, bool FIRST, int NL = 0>
__global__ __launch_bounds__(kBlock) void cn_sweep_wide(GraphDev g, const int *__restrict__ wide_checks, int n_wide,
                                                        const T *__restrict__ src, void *__restrict__ c2v_out,
                                                        const T *__restrict__ beta_row,
                                                        const int *__restrict__ beta_slot,
                                                        const float *__restrict__ thr, int n_levels,
                                                        const T *__restrict__ oms_alpha_row,
                                                        const int *__restrict__ oms_alpha_slot,
                                                        const uint64_t *__restrict__ done,
                                                        const void *__restrict__ prev_out)
{
    constexpr int W = kWave * VEC;
    using OutT = typename std::conditional<FORM == FORM_RCQ, uint8_t, T>::type;
    __shared__ T s_m1[kWavesPerBlock][W], s_m2[kWavesPerBlock][W];
    __shared__ int s_idx[kWavesPerBlock][W];
    __shared__ unsigned s_par[kWavesPerBlock][W], s_nz[kWavesPerBlock][W];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = uni(threadIdx.x >> 6);
    const int tile = uni(blockIdx.x / n_wide);
    const int i = uni(wide_checks[blockIdx.x % n_wide]);
    const int e0 = uni(g.check_ptr[i]);
    const int dc = uni(g.check_ptr[i + 1]) - e0;
    const int quarter = (dc + kWavesPerBlock - 1) / kWavesPerBlock;
    const int t_beg = min(dc, wave * quarter), t_end = min(dc, t_beg + quarter);

    Frozen<VEC> fz;
    const bool all_frozen = load_frozen<VEC>(done, tile, lane, fz);      // tile-uniform: the whole block agrees
    if (all_frozen && !prev_out) return;
    const size_t lane_off = (size_t)lane * VEC;
    OutT *out_base = reinterpret_cast<OutT *>(c2v_out) + ((size_t)tile * g.E + e0) * W + lane_off;
    const OutT *prev_base = prev_out ? reinterpret_cast<const OutT *>(prev_out) + ((size_t)tile * g.E + e0) * W + lane_off : nullptr;
    if (all_frozen) {                              // saving mode: carry the tile's latched rows into this slice
        for (int t = t_beg; t < t_end; ++t)
            st<OutT, VEC>(out_base + (size_t)t * W, ld<OutT, VEC>(prev_base + (size_t)t * W));
        return;
    }
    float th[8];
    if (FORM == FORM_RCQ) {
#pragma unroll
        for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
    }
    const T *in_base = FIRST ? src + (size_t)tile * g.n * W + lane_off : src + ((size_t)tile * g.E + e0) * W + lane_off;
    auto row_of = [&](int t) { return FIRST ? in_base + (size_t)g.var_idx[e0 + t] * W : in_base + (size_t)t * W; };

    T m1[VEC], m2[VEC];
    int idx[VEC];
    uint32_t sm[VEC], zm[VEC];
    unsigned par[VEC], nz[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { m1[c] = inf_of<T>(); m2[c] = inf_of<T>(); idx[c] = 0; sm[c] = 0; zm[c] = 0; par[c] = 0; nz[c] = 0; }
#pragma unroll 4
    for (int t = t_beg; t < t_end; ++t) {
        const Pack<T, VEC> v = ld<T, VEC>(row_of(t));
        const int k = (t - t_beg) & 31;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const T a = abs_of<T>(v.x[c]);
            const unsigned sb = signbit_of<T>(v.x[c]);
            par[c] ^= sb;
            sm[c] |= sb << k;
            if (FORM == FORM_OMS) {
                const unsigned z = (a == (T)0) ? 1u : 0u;
                nz[c] += z;
                zm[c] |= z << k;
            }
            if (a < m1[c]) { m2[c] = m1[c]; m1[c] = a; idx[c] = t; }
            else if (a < m2[c]) { m2[c] = a; }
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const int w = lane * VEC + c;
        s_m1[wave][w] = m1[c]; s_m2[wave][w] = m2[c]; s_idx[wave][w] = idx[c]; s_par[wave][w] = par[c]; s_nz[wave][w] = nz[c];
    }
    __syncthreads();
    // the four quarters in edge order: a strict "<" keeps the FIRST minimum, and a partial whose minimum is not
    // smaller can only lower min2
    T M1[VEC], M2[VEC];
    int IDX[VEC];
    unsigned PAR[VEC], NZ[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const int w = lane * VEC + c;
        M1[c] = inf_of<T>(); M2[c] = inf_of<T>(); IDX[c] = 0; PAR[c] = 0; NZ[c] = 0;
#pragma unroll
        for (int q = 0; q < kWavesPerBlock; ++q) {
            const T p1 = s_m1[q][w], p2 = s_m2[q][w];
            if (p1 < M1[c]) {
                M2[c] = M1[c] < p2 ? M1[c] : p2;
                M1[c] = p1;
                IDX[c] = s_idx[q][w];
            } else {
                M2[c] = p1 < M2[c] ? p1 : M2[c];
            }
            PAR[c] ^= s_par[q][w];
            NZ[c] += s_nz[q][w];
        }
    }
    const bool reread = (t_end - t_beg) > 32;      // this quarter's masks hold 32 edges
#pragma unroll 2
    for (int t = t_beg; t < t_end; ++t) {
        const T b = beta_row[beta_slot[e0 + t]];
        T oa = (T)0;
        if (FORM == FORM_OMS && oms_alpha_row) oa = oms_alpha_row[oms_alpha_slot[e0 + t]];
        Pack<T, VEC> re;
        if (reread) re = ld<T, VEC>(row_of(t));
        const int k = (t - t_beg) & 31;
        Pack<OutT, VEC> o;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const T raw = (t == IDX[c]) ? M2[c] : M1[c];
            const unsigned own = reread ? signbit_of<T>(re.x[c]) : ((sm[c] >> k) & 1u);
            const unsigned neg = PAR[c] ^ own;
            if (FORM == FORM_NMS) {
                o.x[c] = (OutT)flip_sign<T>(b * raw, neg);
            } else if (FORM == FORM_OMS) {
                const unsigned ownz = reread ? ((re.x[c] == (T)0) ? 1u : 0u) : ((zm[c] >> k) & 1u);
                const bool nonzero = (NZ[c] - ownz) == 0;         // no OTHER edge carries a zero
                const T d = raw - b;
                const T r = d > (T)0 ? d : (T)0;
                const T val = r - oa;
                o.x[c] = (OutT)(nonzero ? flip_sign<T>(val, neg) : (T)0);
            } else {
                const float w = flip_sign<float>((float)(b * raw), neg);
                const float mag = __builtin_fabsf(w);
                int lvl = 0;
                if constexpr (NL > 0) {
#pragma unroll
                    for (int q = 1; q < NL; ++q) lvl = (mag >= th[q]) ? q : lvl;
                } else if (n_levels <= 8) {
#pragma unroll
                    for (int q = 1; q < 8; ++q) lvl = (mag >= th[q]) ? q : lvl;
                } else {
                    for (int q = 1; q < n_levels; ++q) lvl = (mag >= thr[q]) ? q : lvl;
                }
                o.x[c] = (OutT)(((w < 0.0f) ? n_levels : 0) + lvl);
            }
        }
        store_latched<OutT, VEC>(out_base + (size_t)t * W, prev_base ? prev_base + (size_t)t * W : nullptr, o, fz);
    }
}

// ------------------------------------------------------------------------------------------
// Compile-time sums in the reference's association order.
//   ORDER 0: torch.sum fp32 (ATen row_sum, ILP 4; N == 8 takes the 8-lane vector path)
//   ORDER 1: np.sum fp64 pairwise (sequential below 8, 8 accumulators at 8)
// x has DV elements; element u of the summed list is x[u] with index SKIP removed.
// ------------------------------------------------------------------------------------------
template <int N, int SKIP, int ORDER, typename T, int DV>
__device__ __forceinline__ T sum_ct(const T (&x)[DV])
{
    // The additions of the reference's zero-initialised accumulators ("0 + x", "+ 0") are left out:
    // they change nothing but the sign of an exact zero, which no later operation can observe as a
    // value (|.|, comparisons, the sign-parity of a zero-magnitude message, quantize(+-0) = 0).
#define LDPC_AT(u) x[((SKIP) >= 0 && (u) >= (SKIP)) ? (u) + 1 : (u)]
    if constexpr (N == 0) {
        return (T)0;
    } else if constexpr (ORDER == 0) {
        if constexpr (N < 4) {
            T p0 = LDPC_AT(0);
#pragma unroll
            for (int u = 1; u < N; ++u) p0 = p0 + LDPC_AT(u);
            return p0;
        } else if constexpr (N < 8) {
            T p0 = LDPC_AT(0);
#pragma unroll
            for (int u = 4; u < N; ++u) p0 = p0 + LDPC_AT(u);      // remainder joins partial 0 first
            p0 = p0 + LDPC_AT(1);
            p0 = p0 + LDPC_AT(2);
            p0 = p0 + LDPC_AT(3);
            return p0;
        } else {
            static_assert(N == 8, "compile-time torch order only up to 8 operands");
            T fin = LDPC_AT(0);                                    // one 8-lane vector: lanes added in order
#pragma unroll
            for (int l = 1; l < 8; ++l) fin = fin + LDPC_AT(l);
            return fin;
        }
    } else {
        if constexpr (N < 8) {
            T res = LDPC_AT(0);
#pragma unroll
            for (int u = 1; u < N; ++u) res = res + LDPC_AT(u);
            return res;
        } else {
            static_assert(N == 8, "compile-time numpy order only up to 8 operands");
            return ((LDPC_AT(0) + LDPC_AT(1)) + (LDPC_AT(2) + LDPC_AT(3))) +
                   ((LDPC_AT(4) + LDPC_AT(5)) + (LDPC_AT(6) + LDPC_AT(7)));
        }
    }
#undef LDPC_AT
}

// Run-time sums for variables of degree > 8: operands are re-fetched through `get`
// (L1/L2 hits), VEC codewords at a time.  Same association orders, any N the host
// admits (torch order N <= 575, numpy order N <= 128).
template <int ORDER, typename T, int VEC, typename Get>
__device__ __forceinline__ Pack<T, VEC> sum_rt(int N, Get get)
{
    Pack<T, VEC> fin;
    auto add = [](Pack<T, VEC> &a, const Pack<T, VEC> &b) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) a.x[c] = a.x[c] + b.x[c];
    };
    auto zero = [](Pack<T, VEC> &a, T v) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) a.x[c] = v;
    };
    if (ORDER == 0) {
        if (N < 8) {
            Pack<T, VEC> p[4];
            for (int k = 0; k < 4; ++k) zero(p[k], (T)0);
            const int G = N / 4;
            for (int r = 0; r < G; ++r)
                for (int k = 0; k < 4; ++k) add(p[k], get(4 * r + k));
            for (int u = 4 * G; u < N; ++u) add(p[0], get(u));
            add(p[0], p[1]); add(p[0], p[2]); add(p[0], p[3]);
            return p[0];
        }
        const int V = N / 8, G = V / 4;
        zero(fin, (T)0);
        for (int u = 8 * V; u < N; ++u) add(fin, get(u));
        for (int l = 0; l < 8; ++l) {
            Pack<T, VEC> p0, p1, p2, p3;
            zero(p0, (T)0); zero(p1, (T)0); zero(p2, (T)0); zero(p3, (T)0);
            for (int r = 0; r < G; ++r) {
                add(p0, get((4 * r + 0) * 8 + l));
                add(p1, get((4 * r + 1) * 8 + l));
                add(p2, get((4 * r + 2) * 8 + l));
                add(p3, get((4 * r + 3) * 8 + l));
            }
            for (int v = 4 * G; v < V; ++v) add(p0, get(v * 8 + l));
            add(p0, p1); add(p0, p2); add(p0, p3);
            add(fin, p0);
        }
        return fin;
    } else {
        if (N == 0) { zero(fin, (T)0); return fin; }
        if (N < 8) {
            zero(fin, (T)-0.0);
            for (int u = 0; u < N; ++u) add(fin, get(u));
            return fin;
        }
        // 8 <= N <= 128 (host rejects larger degrees for this order)
        Pack<T, VEC> r0 = get(0), r1 = get(1), r2 = get(2), r3 = get(3), r4 = get(4), r5 = get(5), r6 = get(6), r7 = get(7);
        int u = 8;
        for (; u < N - (N % 8); u += 8) {
            add(r0, get(u)); add(r1, get(u + 1)); add(r2, get(u + 2)); add(r3, get(u + 3));
            add(r4, get(u + 4)); add(r5, get(u + 5)); add(r6, get(u + 6)); add(r7, get(u + 7));
        }
        add(r0, r1); add(r2, r3); add(r4, r5); add(r6, r7);
        add(r0, r2); add(r4, r6);
        add(r0, r4);
        for (; u < N; ++u) add(r0, get(u));
        return r0;
    }
}

// ------------------------------------------------------------------------------------------
// Variable-node sweep.  One wave = one variable x W codewords: gathers the dv C2V rows
// (CSC order), forms the dv leave-one-out sums and the posterior in the reference's
// association order, writes dv V2C rows, the hard-decision ballots and (LAST) the posterior.
// ------------------------------------------------------------------------------------------
// Reconstruction LUTs of ALL quantisers sit in LDS ([Q][2L] signed values); `off[c]` selects
// the table a codeword uses.  It is the current iteration's quantiser, except in the LAST
// sweep for codewords the early-stop latch froze earlier: their codes were produced by the
// quantiser of the iteration they stopped in and must be reconstructed with that one.
template <int VEC>
struct Lut {
    const float *base;
    int off[VEC];
};

template <typename T, int VEC, bool CODES>
__device__ __forceinline__ Pack<T, VEC> load_c2v(const void *c2v, size_t elem_off, const Lut<VEC> &lut)
{
    if constexpr (CODES) {
        Pack<uint8_t, VEC> q = ld<uint8_t, VEC>(reinterpret_cast<const uint8_t *>(c2v) + elem_off);
        Pack<T, VEC> r;
#pragma unroll
        for (int c = 0; c < VEC; ++c) r.x[c] = (T)lut.base[lut.off[c] + q.x[c]];
        return r;
    } else {
        return ld<T, VEC>(reinterpret_cast<const T *>(c2v) + elem_off);
    }
}

template <typename T, int VEC, bool CODES, int ORDER, bool LAST, int DV>
__device__ __forceinline__ void vn_body(const GraphDev &g, int tile, int j, int s0, int lane,
                                        const void *__restrict__ c2v, const T *__restrict__ llrT,
                                        T *__restrict__ v2c, T a, const Lut<VEC> &lut,
                                        uint64_t *__restrict__ bitsT, T *__restrict__ postT,
                                        const Frozen<VEC> &fz)
{
    constexpr int W = kWave * VEC;
    const size_t lane_off = (size_t)lane * VEC;
    const size_t tileE = (size_t)tile * g.E;
    int e[DV > 0 ? DV : 1];
    Pack<T, VEC> x[DV > 0 ? DV : 1];
#pragma unroll
    for (int k = 0; k < DV; ++k) e[k] = g.csc_edge[s0 + k];
#pragma unroll
    for (int k = 0; k < DV; ++k) x[k] = load_c2v<T, VEC, CODES>(c2v, (tileE + e[k]) * W + lane_off, lut);
    const Pack<T, VEC> l = ld<T, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);

    Pack<T, VEC> post;
    Pack<T, VEC> out[DV > 0 ? DV : 1];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        T xs[DV > 0 ? DV : 1];
#pragma unroll
        for (int k = 0; k < DV; ++k) xs[k] = x[k].x[c];
        post.x[c] = l.x[c] + sum_ct<DV, -1, ORDER, T>(xs);           // posterior: no alpha (:206-209)
        if constexpr (!LAST) {
            // v2c = llr + alpha * sum(others)  (neural_2d_decoder.py:200-203)
            if constexpr (DV >= 1) out[0].x[c] = l.x[c] + a * sum_ct<DV - 1, 0, ORDER, T>(xs);
            if constexpr (DV >= 2) out[1].x[c] = l.x[c] + a * sum_ct<DV - 1, 1, ORDER, T>(xs);
            if constexpr (DV >= 3) out[2].x[c] = l.x[c] + a * sum_ct<DV - 1, 2, ORDER, T>(xs);
            if constexpr (DV >= 4) out[3].x[c] = l.x[c] + a * sum_ct<DV - 1, 3, ORDER, T>(xs);
            if constexpr (DV >= 5) out[4].x[c] = l.x[c] + a * sum_ct<DV - 1, 4, ORDER, T>(xs);
            if constexpr (DV >= 6) out[5].x[c] = l.x[c] + a * sum_ct<DV - 1, 5, ORDER, T>(xs);
            if constexpr (DV >= 7) out[6].x[c] = l.x[c] + a * sum_ct<DV - 1, 6, ORDER, T>(xs);
            if constexpr (DV >= 8) out[7].x[c] = l.x[c] + a * sum_ct<DV - 1, 7, ORDER, T>(xs);
        }
    }
    if constexpr (!LAST) {
#pragma unroll
        for (int k = 0; k < DV; ++k) store_masked<T, VEC>(v2c + (tileE + e[k]) * W + lane_off, out[k], fz);
    }
    // hard decision: one 64-bit ballot per codeword position c  (posterior < 0)
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        uint64_t mask = __ballot(post.x[c] < (T)0);
        if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = mask;
    }
    if constexpr (LAST) {
        if (postT) st<T, VEC>(postT + ((size_t)tile * g.n + j) * W + lane_off, post);   // null: hard decisions only
    }
}

template <typename T, int VEC, bool CODES, int ORDER, bool LAST>
__device__ __forceinline__ void vn_generic(const GraphDev &g, int tile, int j, int s0, int dv, int lane,
                                        const void *__restrict__ c2v, const T *__restrict__ llrT,
                                        T *__restrict__ v2c, T a, const Lut<VEC> &lut,
                                        uint64_t *__restrict__ bitsT, T *__restrict__ postT,
                                        const Frozen<VEC> &fz)
{
    constexpr int W = kWave * VEC;
    const size_t lane_off = (size_t)lane * VEC;
    const size_t tileE = (size_t)tile * g.E;
    const Pack<T, VEC> l = ld<T, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);
    auto fetch = [&](int k) {
        return load_c2v<T, VEC, CODES>(c2v, (tileE + g.csc_edge[s0 + k]) * W + lane_off, lut);
    };
    Pack<T, VEC> post = sum_rt<ORDER, T, VEC>(dv, fetch);
#pragma unroll
    for (int c = 0; c < VEC; ++c) post.x[c] = l.x[c] + post.x[c];
    if (!LAST) {
        for (int k = 0; k < dv; ++k) {
            auto others = [&](int u) { return fetch(u < k ? u : u + 1); };
            Pack<T, VEC> s = sum_rt<ORDER, T, VEC>(dv - 1, others);
            Pack<T, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) o.x[c] = l.x[c] + a * s.x[c];
            store_masked<T, VEC>(v2c + (tileE + g.csc_edge[s0 + k]) * W + lane_off, o, fz);
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        uint64_t mask = __ballot(post.x[c] < (T)0);
        if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = mask;
    }
    if (LAST && postT) st<T, VEC>(postT + ((size_t)tile * g.n + j) * W + lane_off, post);
}

template <typename T, int VEC, bool CODES, bool LAST>
__global__ __launch_bounds__(kBlock) void vn_sweep(GraphDev g, const void *__restrict__ c2v,
                                                   const T *__restrict__ llrT, T *__restrict__ v2c,
                                                   const T *__restrict__ alpha_row,
                                                   const int *__restrict__ alpha_slot,
                                                   const float *__restrict__ lut_global, int lut_total,
                                                   int lut_cur_off, int lut_stride,
                                                   const int *__restrict__ q_of_iter,
                                                   const int *__restrict__ iters_ws,
                                                   uint64_t *__restrict__ bitsT, T *__restrict__ postT,
                                                   const uint64_t *__restrict__ done, int var_blocks)
{
    constexpr int ORDER = sizeof(T) == 8 ? 1 : 0;   // fp64 = numpy decoder, fp32 = torch decoders
    constexpr int W = kWave * VEC;
    extern __shared__ float lut_s[];
    if (CODES) {
        for (int k = threadIdx.x; k < lut_total; k += kBlock) lut_s[k] = lut_global[k];
        __syncthreads();
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / var_blocks);
    const int j = uni((blockIdx.x % var_blocks) * kWavesPerBlock + (threadIdx.x >> 6));
    if (j >= g.n) return;
    const int s0 = uni(g.var_ptr[j]);
    const int dv = uni(g.var_ptr[j + 1]) - s0;

    Frozen<VEC> fz;
    const bool all_frozen = load_frozen<VEC>(done, tile, lane, fz);
    if (!LAST && all_frozen) return;    // LAST still has to publish the latched posterior
    const T a = alpha_row[alpha_slot[j]];
    Lut<VEC> lut;
    lut.base = lut_s;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        lut.off[c] = lut_cur_off;
        if (CODES && LAST && done && fz.one(c)) {
            const int it = iters_ws[(size_t)tile * W + lane * VEC + c];      // 1-based stop iteration
            lut.off[c] = q_of_iter[it > 0 ? it - 1 : 0] * lut_stride;
        }
    }

#define LDPC_VN_CASE(D) \
    case D: vn_body<T, VEC, CODES, ORDER, LAST, D>(g, tile, j, s0, lane, c2v, llrT, v2c, a, lut, bitsT, postT, fz); break;
    switch (dv) {
        LDPC_VN_CASE(0) LDPC_VN_CASE(1) LDPC_VN_CASE(2) LDPC_VN_CASE(3) LDPC_VN_CASE(4)
        LDPC_VN_CASE(5) LDPC_VN_CASE(6) LDPC_VN_CASE(7) LDPC_VN_CASE(8)
    default:
        vn_generic<T, VEC, CODES, ORDER, LAST>(g, tile, j, s0, dv, lane, c2v, llrT, v2c, a, lut, bitsT, postT, fz);
    }
#undef LDPC_VN_CASE
}

// ------------------------------------------------------------------------------------------
// LAST variable pass that writes the CALLER's rows (fp32, 256-codeword tiles): posterior [B][n] and int32 decisions
// [B][n] leave this kernel directly -- no tile-major posterior array, no transpose_out pass (on the (16200,7200) code
// that pair moved 2.1 GB out and 2.1 GB back in and then wrote 4.2 GB at 3.2 TB/s).  One block = one tile x kRowsVars
// consecutive variables: each wave forms the posteriors of its 16 variables exactly as vn_sweep<LAST> does (same
// association order, same per-codeword LUT choice for latched codewords, same ballots), stages them in LDS as
// s[row][variable] with row = c * 64 + lane (codeword 4 * lane + c of the tile) and a row stride of kRowsVars + 1 floats
// -- conflict-free for the lanes' writes and for the 16 threads x 4 rows of a wave reading 16 bytes each -- and the
// block then writes 256-byte runs of the callers' rows.
// ------------------------------------------------------------------------------------------
template <typename T, int BYTES> struct VecB { typedef T type __attribute__((ext_vector_type(BYTES / sizeof(T)))); };
template <typename T> struct VecB<T, sizeof(T)> { typedef T type; };
template <typename V, typename T> __device__ __forceinline__ T vec_get(const V &v, int q)
{
    if constexpr (sizeof(V) == sizeof(T)) return v; else return v[q];
}
template <typename V, typename T> __device__ __forceinline__ void vec_set(V &v, int q, T x)
{
    if constexpr (sizeof(V) == sizeof(T)) v = x; else v[q] = x;
}

constexpr int kRowsVars = 64;
constexpr int kRowsStride = kRowsVars + 1;
#ifndef LDPC_ROWS_XCD
#define LDPC_ROWS_XCD 1              // vn_last_rows: the variable chunks of a tile are dealt to the XCDs in CONTIGUOUS ranges (workgroups go
                                     // round-robin over the 8 XCDs, each with its own L2): a caller row is 4n bytes, not a multiple of the
                                     // 128-byte line, so every chunk boundary splits a line between two workgroups -- on the same XCD the
                                     // two halves meet in one L2, written with plain (temporal) stores.  Config 5, per launch
                                     // (tools/experiments/rows_xcd_round.sh and the r03 profiles): plain order + non-temporal 2.20 ms,
                                     // XCD-contiguous + non-temporal 2.16, XCD-contiguous + temporal 1.96, plain order + temporal 2.40.
                                     // transpose_in_q4 keeps the plain order (1.16 ms; XCD-contiguous 1.31: its tile-row writes lose
                                     // their order).
#endif
#ifndef LDPC_ROWS_NT_STORE
#define LDPC_ROWS_NT_STORE 0         // caller rows of vn_last_rows with non-temporal stores (A/B knob; see above)
#endif
// blockIdx -> (tile, variable chunk) of the two boundary kernels; the grid is tiles x rows_grid_chunks(var_blocks, xcd)
__host__ __device__ inline int rows_grid_chunks(int var_blocks, bool xcd) { return xcd ? (var_blocks + 7) / 8 * 8 : var_blocks; }
template <bool XCD>
__device__ __forceinline__ bool rows_block(int var_blocks, int &tile, int &chunk)
{
    const int gc = rows_grid_chunks(var_blocks, XCD);
    tile = uni((int)(blockIdx.x / gc));
    const int k = uni((int)(blockIdx.x % gc));
    chunk = XCD ? (k % 8) * (gc / 8) + k / 8 : k;                   // XCD x (= k % 8) owns chunks [x * gc/8, (x + 1) * gc/8)
    return chunk < var_blocks;
}
template <typename V>
__device__ __forceinline__ void rows_store(V v, V *p)
{
#if LDPC_ROWS_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
constexpr int kRowsThreads = 1024;   // 16 waves x 4 variables: the staging tile (66.5 KB) admits two blocks per CU, and the first
                                     // phase needs every wave slot of the CU to keep enough row loads in flight (with 256-thread
                                     // blocks -- 8 waves per CU -- the kernel ran at 3.7 TB/s)
__host__ __device__ inline size_t vn_rows_stage_bytes() { return (size_t)256 * kRowsStride * sizeof(float); }

// PRE: the edge ids of the wave's variables were fetched once with the lanes in parallel (`ev`, lane = position in the wave's
// CSC range); this variable's are lanes off .. off + DV - 1 -- no dependent scalar load in front of the row loads
template <typename T, int VEC, bool CODES, int ORDER, int DV, bool PRE = false>
__device__ __forceinline__ Pack<T, VEC> vn_post_ct(const GraphDev &g, int tile, int j, int s0, int lane,
                                                   const void *__restrict__ c2v, const T *__restrict__ llrT,
                                                   const Lut<VEC> &lut, int ev = 0, int off = 0)
{
    constexpr int W = kWave * VEC;
    const size_t lane_off = (size_t)lane * VEC;
    const size_t tileE = (size_t)tile * g.E;
    int e[DV > 0 ? DV : 1];
    Pack<T, VEC> x[DV > 0 ? DV : 1];
#pragma unroll
    for (int k = 0; k < DV; ++k) e[k] = PRE ? __builtin_amdgcn_readlane(ev, off + k) : g.csc_edge[s0 + k];
#pragma unroll
    for (int k = 0; k < DV; ++k) x[k] = load_c2v<T, VEC, CODES>(c2v, (tileE + e[k]) * W + lane_off, lut);
    const Pack<T, VEC> l = ld<T, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);
    Pack<T, VEC> post;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        T xs[DV > 0 ? DV : 1];
#pragma unroll
        for (int k = 0; k < DV; ++k) xs[k] = x[k].x[c];
        post.x[c] = l.x[c] + sum_ct<DV, -1, ORDER, T>(xs);           // posterior: no alpha (:206-209)
    }
    return post;
}

// QV: floats per store on the caller side (4 / 2 / 1: what the row length and the buffers' alignment admit)
template <bool CODES, int QV>
__global__ __launch_bounds__(kRowsThreads, 8) void vn_last_rows(GraphDev g, const void *__restrict__ c2v,
                                                       const float *__restrict__ llrT,
                                                       const float *__restrict__ lut_global, int lut_total,
                                                       int lut_cur_off, int lut_stride,
                                                       const int *__restrict__ q_of_iter,
                                                       const int *__restrict__ iters_ws,
                                                       uint64_t *__restrict__ bitsT, const uint64_t *__restrict__ done,
                                                       float *__restrict__ posterior, int *__restrict__ bits,
                                                       long long batch, int var_blocks)
{
    constexpr int VEC = 4, W = kWave * VEC;
    constexpr int kPerWave = kRowsVars / (kRowsThreads / kWave);
    extern __shared__ float rows_smem[];
    float *stage = rows_smem;
    float *lut_s = rows_smem + (size_t)W * kRowsStride;
    if (CODES) {
        for (int k = threadIdx.x; k < lut_total; k += kRowsThreads) lut_s[k] = lut_global[k];
        __syncthreads();
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int tile, chunk;
    if (!rows_block<LDPC_ROWS_XCD != 0>(var_blocks, tile, chunk)) return;   // block-uniform (padding blocks of the XCD-contiguous grid)
    const int j0 = chunk * kRowsVars;

    Frozen<VEC> fz;
    load_frozen<VEC>(done, tile, lane, fz);
    Lut<VEC> lut;
    lut.base = lut_s;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        lut.off[c] = lut_cur_off;
        if (CODES && done && fz.one(c)) {
            const int it = iters_ws[(size_t)tile * W + lane * VEC + c];      // 1-based stop iteration
            lut.off[c] = q_of_iter[it > 0 ? it - 1 : 0] * lut_stride;
        }
    }
    // index data of the wave's kPerWave variables, fetched ONCE with the lanes in parallel (as vn_sweep_q4): var_ptr[jbase ..
    // jbase + kPerWave] -> CSC edge ids; per variable these were three dependent round trips in front of the row loads
    const int jbase = j0 + wave * kPerWave;
    const int nv = max(0, min(kPerWave, g.n - jbase));
    const int vp = g.var_ptr[min(jbase + min(lane, kPerWave), g.n)];
    const int s_base = __builtin_amdgcn_readfirstlane(vp);
    const int n_edges = __builtin_amdgcn_readlane(vp, nv) - s_base;
    const bool pre = n_edges <= kWave;                             // degrees <= 8 always qualify (4 x 8 edges); else per-variable loads
    int ev = 0;
    if (pre && lane < n_edges) ev = g.csc_edge[s_base + lane];
    for (int u = 0; u < nv; ++u) {
        const int jj = wave * kPerWave + u, j = j0 + jj;
        const int s0 = __builtin_amdgcn_readlane(vp, u);
        const int dv = __builtin_amdgcn_readlane(vp, u + 1) - s0;
        Pack<float, VEC> post;
        auto generic_post = [&]() {                                 // degree > 8: run-time sums, operands re-fetched (L1 / L2 hits)
            const size_t lane_off = (size_t)lane * VEC, tileE = (size_t)tile * g.E;
            auto fetch = [&](int k) {
                return load_c2v<float, VEC, CODES>(c2v, (tileE + g.csc_edge[s0 + k]) * W + lane_off, lut);
            };
            Pack<float, VEC> r = sum_rt<0, float, VEC>(dv, fetch);
            const Pack<float, VEC> l = ld<float, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);
#pragma unroll
            for (int c = 0; c < VEC; ++c) r.x[c] = l.x[c] + r.x[c];
            return r;
        };
        if (pre) {
#define LDPC_VPP_CASE(D) case D: post = vn_post_ct<float, VEC, CODES, 0, D, true>(g, tile, j, s0, lane, c2v, llrT, lut, ev, s0 - s_base); break;
            switch (dv) {
                LDPC_VPP_CASE(0) LDPC_VPP_CASE(1) LDPC_VPP_CASE(2) LDPC_VPP_CASE(3) LDPC_VPP_CASE(4)
                LDPC_VPP_CASE(5) LDPC_VPP_CASE(6) LDPC_VPP_CASE(7) LDPC_VPP_CASE(8)
            default: post = generic_post();
            }
#undef LDPC_VPP_CASE
        } else {
#define LDPC_VP_CASE(D) case D: post = vn_post_ct<float, VEC, CODES, 0, D>(g, tile, j, s0, lane, c2v, llrT, lut); break;
            switch (dv) {
                LDPC_VP_CASE(0) LDPC_VP_CASE(1) LDPC_VP_CASE(2) LDPC_VP_CASE(3) LDPC_VP_CASE(4)
                LDPC_VP_CASE(5) LDPC_VP_CASE(6) LDPC_VP_CASE(7) LDPC_VP_CASE(8)
            default: post = generic_post();
            }
#undef LDPC_VP_CASE
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const uint64_t mask = __ballot(post.x[c] < 0.0f);        // the syndrome pass reads the ballot words
            if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = mask;
            stage[(size_t)(c * kWave + lane) * kRowsStride + jj] = post.x[c];
        }
    }
    __syncthreads();
    using FV = typename VecB<float, QV * 4>::type;
    using IV = typename VecB<int, QV * 4>::type;
    constexpr int kTpr = kRowsVars / QV;                            // threads per staged row
    constexpr int kRpp = kRowsThreads / kTpr;                       // rows per pass (a wave: 64 / kTpr rows x kTpr threads,
                                                                    //   banks (row + first float) mod 64 all distinct)
    const int jq = (threadIdx.x % kTpr) * QV;
    if (j0 + jq >= g.n) return;                                     // n is a multiple of QV (host check): whole vectors
#pragma unroll 4
    for (int p = 0; p < W / kRpp; ++p) {
        const int r = p * kRpp + threadIdx.x / kTpr;                // staged row: codeword 4 * (r % 64) + r / 64 of the tile
        const long long b = (long long)tile * W + (r & 63) * VEC + (r >> 6);
        if (b >= batch) continue;
        const float *src = stage + (size_t)r * kRowsStride + jq;
        FV v;
        IV d;
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            vec_set<FV, float>(v, q, src[q]);
            vec_set<IV, int>(d, q, src[q] < 0.0f ? 1 : 0);
        }
        const size_t o = (size_t)b * g.n + j0 + jq;
        if (posterior) rows_store(v, reinterpret_cast<FV *>(posterior + o));
        if (bits) rows_store(d, reinterpret_cast<IV *>(bits + o));
    }
}

// ------------------------------------------------------------------------------------------
// RCQ with ONE beta per check, "code pair" form: BOTH message directions travel as one byte per edge.
//
// The check update of iteration t+1 (rcq_decoder.py:242-246 / :559-563) sees a variable->check message v only through
//   sign(v)   and   level_{t+1}(| beta * min_{others} |v| |),
// and with one beta per check and non-decreasing thresholds  x -> level(|beta * x|)  is non-decreasing in x, so
//   level(|beta * min_u |v_u||) = min_u level(|beta * |v_u||)        (exactly: the same float product is rounded).
// The variable sweep therefore applies the NEXT iteration's beta and quantiser to each outgoing value itself and stores
//   byte = key | sign << 6,   key = 0 when |beta * |v|| is not > 0 (zero / underflow: the check output is code
//                             level(0), its sign bit cleared, rcq_decoder.py:88),  else 1 + level(|beta * |v||)
// and the check sweep is integer-only: min1 / min2 of the keys, XOR of the sign bits, one table-free code per edge.
// Per codeword and iteration: 4E + 4n bytes (E = both code arrays read and written once, 4n the LLR rows) against
// 10E + 4n for the fp32 V2C array and 5E + sum dv(dv-1) issued by the gather form.
// Iteration 0 stays cn_sweep<FIRST> on the LLR rows, the last variable pass stays vn_sweep<CODES, LAST>.
// ------------------------------------------------------------------------------------------
constexpr unsigned kKeyMask = 63u;          // n_levels <= 62 (host check)

template <int NL>
__device__ __forceinline__ unsigned v2c_code(float val, float b, const float (&th)[8], const float *thr, int n_levels)
{
    const float mag = __builtin_fabsf(b * __builtin_fabsf(val));
    unsigned lvl = 0;
    if constexpr (NL > 0) {
#pragma unroll
        for (int q = 1; q < NL; ++q) lvl = (mag >= th[q]) ? (unsigned)q : lvl;
    } else if (n_levels <= 8) {
#pragma unroll
        for (int q = 1; q < 8; ++q) lvl = (mag >= th[q]) ? (unsigned)q : lvl;       // NaN padding never matches
    } else {
        for (int q = 1; q < n_levels; ++q) lvl = (mag >= thr[q]) ? (unsigned)q : lvl;
    }
    const unsigned key = (mag > 0.0f) ? lvl + 1u : 0u;
    return key | (signbit_of<float>(val) << 6);
}

template <int VEC, int NL, int DV>
__device__ __forceinline__ void vn_q_body(const GraphDev &g, int tile, int j, int s0, int lane,
                                          const uint8_t *__restrict__ c2v, const float *__restrict__ llrT,
                                          uint8_t *__restrict__ v2c, float a, const Lut<VEC> &lut,
                                          const float *__restrict__ beta_next, const int *__restrict__ beta_slot,
                                          const float (&th)[8], const float *__restrict__ thr, int n_levels,
                                          uint64_t *__restrict__ bitsT, const Frozen<VEC> &fz)
{
    constexpr int W = kWave * VEC;
    constexpr int D = DV > 0 ? DV : 1;
    const size_t lane_off = (size_t)lane * VEC;
    const size_t tileE = (size_t)tile * g.E;
    int e[D];
    float bb[D];
    Pack<float, VEC> x[D];
#pragma unroll
    for (int k = 0; k < DV; ++k) e[k] = g.csc_edge[s0 + k];
#pragma unroll
    for (int k = 0; k < DV; ++k) x[k] = load_c2v<float, VEC, true>(c2v, (tileE + e[k]) * W + lane_off, lut);
    const Pack<float, VEC> l = ld<float, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);
#pragma unroll
    for (int k = 0; k < DV; ++k) bb[k] = beta_next[beta_slot[e[k]]];

    Pack<float, VEC> post;
    Pack<uint8_t, VEC> out[D];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        float xs[D];
#pragma unroll
        for (int k = 0; k < DV; ++k) xs[k] = x[k].x[c];
        post.x[c] = l.x[c] + sum_ct<DV, -1, 0, float>(xs);
        float v[D];
        if constexpr (DV >= 1) v[0] = l.x[c] + a * sum_ct<DV - 1, 0, 0, float>(xs);
        if constexpr (DV >= 2) v[1] = l.x[c] + a * sum_ct<DV - 1, 1, 0, float>(xs);
        if constexpr (DV >= 3) v[2] = l.x[c] + a * sum_ct<DV - 1, 2, 0, float>(xs);
        if constexpr (DV >= 4) v[3] = l.x[c] + a * sum_ct<DV - 1, 3, 0, float>(xs);
        if constexpr (DV >= 5) v[4] = l.x[c] + a * sum_ct<DV - 1, 4, 0, float>(xs);
        if constexpr (DV >= 6) v[5] = l.x[c] + a * sum_ct<DV - 1, 5, 0, float>(xs);
        if constexpr (DV >= 7) v[6] = l.x[c] + a * sum_ct<DV - 1, 6, 0, float>(xs);
        if constexpr (DV >= 8) v[7] = l.x[c] + a * sum_ct<DV - 1, 7, 0, float>(xs);
#pragma unroll
        for (int k = 0; k < DV; ++k) out[k].x[c] = (uint8_t)v2c_code<NL>(v[k], bb[k], th, thr, n_levels);
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) store_masked<uint8_t, VEC>(v2c + (tileE + e[k]) * W + lane_off, out[k], fz);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        uint64_t mask = __ballot(post.x[c] < 0.0f);
        if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = mask;
    }
}

// Variable sweep of the code-pair form (not the last one): C2V codes of iteration `it` in, V2C codes for iteration
// it + 1 out.  `lut_cur` = the 2L signed reconstruction values of iteration it's quantiser; `beta_next` / `thr_next` = row
// and thresholds of iteration it + 1.
template <int VEC, int NL>
__global__ __launch_bounds__(kBlock) void vn_sweep_q(GraphDev g, const uint8_t *__restrict__ c2v,
                                                     const float *__restrict__ llrT, uint8_t *__restrict__ v2c,
                                                     const float *__restrict__ alpha_row, const int *__restrict__ alpha_slot,
                                                     const float *__restrict__ lut_cur, int lut_entries,
                                                     const float *__restrict__ beta_next, const int *__restrict__ beta_slot,
                                                     const float *__restrict__ thr_next, int n_levels,
                                                     uint64_t *__restrict__ bitsT, const uint64_t *__restrict__ done,
                                                     int var_blocks)
{
    constexpr int W = kWave * VEC;
    extern __shared__ float lut_s[];
    for (int k = threadIdx.x; k < lut_entries; k += kBlock) lut_s[k] = lut_cur[k];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / var_blocks);
    const int j = uni((blockIdx.x % var_blocks) * kWavesPerBlock + (threadIdx.x >> 6));
    if (j >= g.n) return;
    const int s0 = uni(g.var_ptr[j]);
    const int dv = uni(g.var_ptr[j + 1]) - s0;

    Frozen<VEC> fz;
    if (load_frozen<VEC>(done, tile, lane, fz)) return;
    const float a = alpha_row[alpha_slot[j]];
    float th[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr_next[q] : __builtin_nanf("");
    Lut<VEC> lut;
    lut.base = lut_s;
#pragma unroll
    for (int c = 0; c < VEC; ++c) lut.off[c] = 0;

#define LDPC_VQ_CASE(D) \
    case D: vn_q_body<VEC, NL, D>(g, tile, j, s0, lane, c2v, llrT, v2c, a, lut, beta_next, beta_slot, th, thr_next, n_levels, bitsT, fz); break;
    switch (dv) {
        LDPC_VQ_CASE(0) LDPC_VQ_CASE(1) LDPC_VQ_CASE(2) LDPC_VQ_CASE(3) LDPC_VQ_CASE(4)
        LDPC_VQ_CASE(5) LDPC_VQ_CASE(6) LDPC_VQ_CASE(7) LDPC_VQ_CASE(8)
    default: {
        const size_t lane_off = (size_t)lane * VEC;
        const size_t tileE = (size_t)tile * g.E;
        const Pack<float, VEC> l = ld<float, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);
        auto fetch = [&](int k) {
            return load_c2v<float, VEC, true>(c2v, (tileE + g.csc_edge[s0 + k]) * W + lane_off, lut);
        };
        Pack<float, VEC> post = sum_rt<0, float, VEC>(dv, fetch);
        for (int k = 0; k < dv; ++k) {
            auto others = [&](int u) { return fetch(u < k ? u : u + 1); };
            const Pack<float, VEC> sm = sum_rt<0, float, VEC>(dv - 1, others);
            const int e = g.csc_edge[s0 + k];
            const float b = beta_next[beta_slot[e]];
            Pack<uint8_t, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) o.x[c] = (uint8_t)v2c_code<NL>(l.x[c] + a * sm.x[c], b, th, thr_next, n_levels);
            store_masked<uint8_t, VEC>(v2c + (tileE + e) * W + lane_off, o, fz);
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            uint64_t mask = __ballot(l.x[c] + post.x[c] < 0.0f);
            if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = mask;
        }
    }
    }
#undef LDPC_VQ_CASE
}

// Check sweep of the code-pair form (iterations >= 1): V2C codes in, C2V codes out, integer arithmetic only.
// DCMAX > 0: the check's rows stay in registers between the two passes (graphs with max check degree <= DCMAX);
// DCMAX == 0: any degree, pass 2 re-reads the row it is about to answer.
template <int VEC, int CPW, int DCMAX>
__global__ __launch_bounds__(kBlock) void cn_sweep_q(GraphDev g, const uint8_t *__restrict__ v2c,
                                                     uint8_t *__restrict__ c2v_out, const float *__restrict__ beta_row,
                                                     const int *__restrict__ beta_slot, const float *__restrict__ thr,
                                                     int n_levels, const uint64_t *__restrict__ done, int check_blocks)
{
    constexpr int W = kWave * VEC;
    constexpr int R = DCMAX > 0 ? DCMAX : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / check_blocks);
    const int ibase = uni(((blockIdx.x % check_blocks) * kWavesPerBlock + (threadIdx.x >> 6)) * CPW);
    if (ibase >= g.m) return;
    Frozen<VEC> fz;
    if (load_frozen<VEC>(done, tile, lane, fz)) return;
    constexpr unsigned lvl0 = 0;                       // level(0): thresholds 1.. are > 0 (host check)
    const size_t lane_off = (size_t)lane * VEC;

#pragma unroll
    for (int cc_ = 0; cc_ < CPW; ++cc_) {
        const int i = ibase + cc_;
        if (i >= g.m) break;
        const int e0 = uni(g.check_ptr[i]);
        const int dc = uni(g.check_ptr[i + 1]) - e0;
        if (dc == 0) continue;
        const uint8_t *in_base = v2c + ((size_t)tile * g.E + e0) * W + lane_off;
        uint8_t *out_base = c2v_out + ((size_t)tile * g.E + e0) * W + lane_off;
        const unsigned sgn_b = signbit_of<float>(beta_row[beta_slot[e0]]) ? (unsigned)n_levels : 0u;   // sign of beta * min

        Pack<uint8_t, VEC> in[R];
        unsigned m1[VEC], m2[VEC], par[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) { m1[c] = 255u; m2[c] = 255u; par[c] = 0; }
        if constexpr (DCMAX > 0) {
#pragma unroll
            for (int t = 0; t < DCMAX; ++t)
                if (t < dc) in[t] = ld<uint8_t, VEC>(in_base + (size_t)t * W);
#pragma unroll
            for (int t = 0; t < DCMAX; ++t) {
                if (t < dc) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {
                        const unsigned q = in[t].x[c], key = q & kKeyMask;
                        par[c] ^= q;
                        m2[c] = min(m2[c], max(m1[c], key));
                        m1[c] = min(m1[c], key);
                    }
                }
            }
        } else {
#pragma unroll LDPC_CN_UNROLL
            for (int t = 0; t < dc; ++t) {
                const Pack<uint8_t, VEC> v = ld<uint8_t, VEC>(in_base + (size_t)t * W);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const unsigned q = v.x[c], key = q & kKeyMask;
                    par[c] ^= q;
                    m2[c] = min(m2[c], max(m1[c], key));
                    m1[c] = min(m1[c], key);
                }
            }
        }
        if (dc == 1) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) m2[c] = m1[c];       // "min2_val = min_val" (rcq_decoder.py:233-234)
        }
        auto emit = [&](int t, const Pack<uint8_t, VEC> &own) {
            Pack<uint8_t, VEC> o;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const unsigned q = own.x[c], key = q & kKeyMask;
                const unsigned k = (key == m1[c]) ? m2[c] : m1[c];            // min over the OTHER edges (ties: m2 == m1)
                const unsigned neg = ((par[c] ^ q) >> 6) & 1u;                 // sign parity of the others
                const unsigned sgn = neg ? ((unsigned)n_levels - sgn_b) : sgn_b;   // (w < 0) * L
                o.x[c] = (uint8_t)(k == 0u ? lvl0 : k - 1u + sgn);
            }
            store_masked<uint8_t, VEC>(out_base + (size_t)t * W, o, fz);
        };
        if constexpr (DCMAX > 0) {
#pragma unroll
            for (int t = 0; t < DCMAX; ++t)
                if (t < dc) emit(t, in[t]);
        } else {
#pragma unroll LDPC_CN_UNROLL
            for (int t = 0; t < dc; ++t) emit(t, ld<uint8_t, VEC>(in_base + (size_t)t * W));
        }
    }
}

// ---- 256-codeword tiles (VEC = 4): the four codewords of a lane are the four bytes of one dword, and both sweeps
// work on the dword where they can (the generic kernels above spend 13 / 25 VALU instructions per codeword and edge,
// which is more than the byte traffic leaves room for; these take ~6 / ~15).
// ES = false (fixed iteration count): no latch words, plain stores, no per-iteration hard decisions.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

template <bool ES>
__device__ __forceinline__ void store_codes4(uint8_t *p, unsigned o, const Frozen<4> &fz)
{
    if constexpr (ES) store_masked<uint8_t, 4>(p, __builtin_bit_cast(Pack<uint8_t, 4>, o), fz);
    else st<uint8_t, 4>(p, __builtin_bit_cast(Pack<uint8_t, 4>, o));
}

// Check sweep, VEC = 4.  Keys of codewords 0/2 and 1/3 sit in the 16-bit halves of two registers (v_pk_min/max_u16);
// the four possible answers of a check -- (min over others = m1 | m2) x (sign parity of the others 0 | 1) -- are built once
// per check as byte vectors, and every edge picks per byte with three v_perm_b32 (parity of the others, own key == m1).
// Thresholds 1.. are > 0 (host check), so a zero-magnitude output is code 0.
// DC > 0: exactly DC edges, straight-line code (DCMAX <= 8 graphs: one body per degree); DC == 0: up to DCMAX edges under
// `t < dc` predicates (DCMAX 16 / 32), or any degree with the rows re-read in pass 2 (DCMAX == 0).
template <int DC, int DCMAX, bool ES>
__device__ __forceinline__ void cn_q4_check(int dc, const uint8_t *__restrict__ in_base, uint8_t *__restrict__ out_base,
                                            unsigned s_pos, unsigned s_neg, const Frozen<4> &fz)
{
    constexpr int VEC = 4, W = kWave * VEC;
    constexpr int R = DC > 0 ? DC : (DCMAX > 0 ? DCMAX : 1);
    constexpr bool HELD = DC > 0 || DCMAX > 0;
    auto row = [](const uint8_t *p) { return __builtin_bit_cast(unsigned, ld<uint8_t, VEC>(p)); };
    auto live = [&](int t) { return DC > 0 ? true : t < dc; };
    unsigned in[R];
    unsigned par = 0;
    u16x2 m1l = as_u16x2(0x00ff00ffu), m2l = m1l, m1h = m1l, m2h = m1l;
    auto absorb = [&](unsigned q) {
        const u16x2 kl = as_u16x2(q & 0x003f003fu), kh = as_u16x2((q >> 8) & 0x003f003fu);
        par ^= q;
        m2l = __builtin_elementwise_min(m2l, __builtin_elementwise_max(m1l, kl));
        m1l = __builtin_elementwise_min(m1l, kl);
        m2h = __builtin_elementwise_min(m2h, __builtin_elementwise_max(m1h, kh));
        m1h = __builtin_elementwise_min(m1h, kh);
    };
    if constexpr (HELD) {
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (live(t)) in[t] = row(in_base + (size_t)t * W);
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (live(t)) absorb(in[t]);
    } else {
#pragma unroll LDPC_CN_UNROLL
        for (int t = 0; t < dc; ++t) absorb(row(in_base + (size_t)t * W));
    }
    if (DC == 1 || (DC == 0 && dc == 1)) { m2l = m1l; m2h = m1h; }       // "min2_val = min_val" (rcq_decoder.py:233-234)
    // answers: code = 0 for key 0, else key - 1 + (w < 0) * L
    auto answer = [&](u16x2 kl, u16x2 kh, unsigned sgn) {
        const u16x2 one = as_u16x2(0x00010001u), s2 = as_u16x2(sgn);
        const u16x2 al = __builtin_elementwise_sub_sat(kl, one) + __builtin_elementwise_min(kl, one) * s2;
        const u16x2 ah = __builtin_elementwise_sub_sat(kh, one) + __builtin_elementwise_min(kh, one) * s2;
        return as_u32(al) | (as_u32(ah) << 8);
    };
    const unsigned a1p = answer(m1l, m1h, s_pos), a1n = answer(m1l, m1h, s_neg);
    const unsigned a2p = answer(m2l, m2h, s_pos), a2n = answer(m2l, m2h, s_neg);
    const unsigned m1b = as_u32(m1l) | (as_u32(m1h) << 8);
    auto emit = [&](int t, unsigned q) {
        const unsigned sel = (((par ^ q) >> 4) & 0x04040404u) | 0x03020100u;            // byte c: c + 4 * parity of the others
        const unsigned p1 = __builtin_amdgcn_perm(a1n, a1p, sel), p2 = __builtin_amdgcn_perm(a2n, a2p, sel);
        const unsigned x = ((q & 0x3f3f3f3fu) ^ m1b) + 0x7f7f7f7fu;                   // bit 7 of a byte: own key != m1
        const unsigned sel2 = ((x >> 5) & 0x04040404u) | 0x03020100u;                   // byte c: c + 4 * (own key != m1)
        store_codes4<ES>(out_base + (size_t)t * W, __builtin_amdgcn_perm(p1, p2, sel2), fz);
    };
    if constexpr (HELD) {
#pragma unroll
        for (int t = 0; t < R; ++t)
            if (live(t)) emit(t, in[t]);
    } else {
#pragma unroll LDPC_CN_UNROLL
        for (int t = 0; t < dc; ++t) emit(t, row(in_base + (size_t)t * W));
    }
}

template <int CPW, int DCMAX, bool ES>
__global__ __launch_bounds__(kBlock) void cn_sweep_q4(GraphDev g, const uint8_t *__restrict__ v2c,
                                                      uint8_t *__restrict__ c2v_out, const float *__restrict__ beta_row,
                                                      const int *__restrict__ beta_slot, int n_levels,
                                                      const uint64_t *__restrict__ done, int check_blocks)
{
    constexpr int VEC = 4, W = kWave * VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / check_blocks);
    const int ibase = uni(((blockIdx.x % check_blocks) * kWavesPerBlock + (threadIdx.x >> 6)) * CPW);
    if (ibase >= g.m) return;
    Frozen<VEC> fz;
    fz.bits = 0;
    if constexpr (ES) { if (load_frozen<VEC>(done, tile, lane, fz)) return; }
    const size_t lane_off = (size_t)lane * VEC;

#pragma unroll
    for (int cc_ = 0; cc_ < CPW; ++cc_) {
        const int i = ibase + cc_;
        if (i >= g.m) break;
        const int e0 = uni(g.check_ptr[i]);
        const int dc = uni(g.check_ptr[i + 1]) - e0;
        if (dc == 0) continue;
        const uint8_t *in_base = v2c + ((size_t)tile * g.E + e0) * W + lane_off;
        uint8_t *out_base = c2v_out + ((size_t)tile * g.E + e0) * W + lane_off;
        const unsigned sb = signbit_of<float>(beta_row[beta_slot[e0]]) ? (unsigned)n_levels : 0u;   // sign of beta * min
        const unsigned s_pos = sb * 0x00010001u, s_neg = ((unsigned)n_levels - sb) * 0x00010001u;   // (w < 0) * L, both halves
        if constexpr (DCMAX == 8) {
#define LDPC_CQ_CASE(D) case D: cn_q4_check<D, 8, ES>(dc, in_base, out_base, s_pos, s_neg, fz); break;
            switch (dc) {
                LDPC_CQ_CASE(1) LDPC_CQ_CASE(2) LDPC_CQ_CASE(3) LDPC_CQ_CASE(4)
                LDPC_CQ_CASE(5) LDPC_CQ_CASE(6) LDPC_CQ_CASE(7) LDPC_CQ_CASE(8)
            default: break;
            }
#undef LDPC_CQ_CASE
        } else {
            cn_q4_check<0, DCMAX, ES>(dc, in_base, out_base, s_pos, s_neg, fz);
        }
    }
}

// key of one outgoing value, integer form: the magnitude's bit pattern against the thresholds' (all non-negative, so the
// unsigned order is the float order; NaN is squashed to 0 first, as `NaN >= tau` and `NaN > 0` are false).
// NL = 4: three compares into separate lane masks, then three add-with-carry -- hand-scheduled, because a VALU read of a
// lane mask needs two instructions after the compare that wrote it and the compiler pads every pair with s_nop.
template <int NL>
__device__ __forceinline__ unsigned key_of(float val, float b, const unsigned (&tb)[8])
{
    const float mag = __builtin_fmaxf(__builtin_fabsf(b * __builtin_fabsf(val)), 0.0f);
    const unsigned m = __float_as_uint(mag);
    unsigned key;
    if constexpr (NL == 4) {
        unsigned long long c1, c2;
        asm("v_cmp_le_u32_e64 %1, %4, %3\n\t"
            "v_cmp_le_u32_e64 %2, %5, %3\n\t"
            "v_cmp_le_u32_e32 vcc, %6, %3\n\t"
            "v_min_u32_e32 %0, 1, %3\n\t"
            "v_addc_co_u32_e64 %0, %1, 0, %0, %1\n\t"
            "v_addc_co_u32_e64 %0, %2, 0, %0, %2\n\t"
            "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
            : "=&v"(key), "=&s"(c1), "=&s"(c2)
            : "v"(m), "s"(tb[1]), "s"(tb[2]), "s"(tb[3])
            : "vcc");
    } else {
        key = min(m, 1u);
#pragma unroll
        for (int q = 1; q < 8; ++q) key += (m >= tb[q]) ? 1u : 0u;                        // NaN padding: above every magnitude
    }
    return key;
}

// Float form of the same key for 4 levels (NL == kKeyFloat4): key = [m > 0] + [m >= t1] + [m >= t2] + [m >= t3] with every
// bracket an exact 0.0f / 1.0f from ONE fast instruction.  On gfx950 v_add/v_mul/v_fma_f32 with VGPR operands (abs / neg /
// clamp modifiers included) issue in ~2.2 cycles when two or more waves share the SIMD, compares, add-with-carry, min/max,
// shifts and everything with an SGPR operand in ~4.2 (tools/probes/valu_issue_probe.hip, profiles/r03_valu_issue_probe.txt):
// the compare chain above costs 9 slow instructions per value, this form 2 packed + 4 fast + 3 adds.
//   G = (b * v) * K, K = 2^75 (power of two: exact, also for a subnormal product; a product >= 2^53 becomes inf, which passes
//   every bracket as it must);  [m >= t] = clamp(|G| - prev(t) * K): m >= t  <=>  m > prev(t), and then the difference is at
//   least ulp(prev(t)) * K >= 1 (the host admits thresholds in [2^-50, 2^50] to this form), else it is <= 0;
//   [m > 0] = clamp(|G| * K) (m >= 2^-149 -> >= 2);  NaN -> every bracket 0 (DX10 clamp), as the compare form's squash.
constexpr int kKeyFloat4 = 104;
#ifndef LDPC_KEY_FLOAT
#define LDPC_KEY_FLOAT 1          // 0: 4-level decoders keep the compare chain (A/B builds)
#endif
struct KeyTab {
    unsigned tb[8];               // compare forms: threshold bit patterns (NaN padding)
    float K, c1, c2, c3;          // float form: 2^75 and prev(t_q) * 2^75
};
template <int NL>
__device__ __forceinline__ void key_tab_init(KeyTab &kt, const float *__restrict__ thr, int n_levels)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) kt.tb[q] = (q < n_levels) ? __float_as_uint(thr[q]) : 0x7fc00000u;
    kt.K = 0x1p75f; kt.c1 = kt.c2 = kt.c3 = 0.0f;
    if constexpr (NL == kKeyFloat4) {
        kt.c1 = __uint_as_float(kt.tb[1] - 1u) * 0x1p75f;
        kt.c2 = __uint_as_float(kt.tb[2] - 1u) * 0x1p75f;
        kt.c3 = __uint_as_float(kt.tb[3] - 1u) * 0x1p75f;
    }
}
// keys of a codeword pair's scaled values G into bytes B0 / B0 + 1 of `keys`: the eight brackets, their sums (exact small
// integers) and the two float -> byte insertions as ONE block, so that the scheduler cannot spread the fourteen temporaries of
// a degree-8 variable's sixteen values over the whole body (it did: 170 VGPRs)
template <int B0>
__device__ __forceinline__ unsigned key_pair4(f32x2 G, const KeyTab &kt, unsigned keys)
{
    float a0, a1, a2, a3, b0, b1, b2, b3;
    asm("v_mul_f32_e64 %1, |%9|, %11 clamp\n\t"
        "v_add_f32_e64 %2, |%9|, -%12 clamp\n\t"
        "v_add_f32_e64 %3, |%9|, -%13 clamp\n\t"
        "v_add_f32_e64 %4, |%9|, -%14 clamp\n\t"
        "v_mul_f32_e64 %5, |%10|, %11 clamp\n\t"
        "v_add_f32_e64 %6, |%10|, -%12 clamp\n\t"
        "v_add_f32_e64 %7, |%10|, -%13 clamp\n\t"
        "v_add_f32_e64 %8, |%10|, -%14 clamp\n\t"
        "v_add_f32_e32 %1, %1, %2\n\t"
        "v_add_f32_e32 %3, %3, %4\n\t"
        "v_add_f32_e32 %5, %5, %6\n\t"
        "v_add_f32_e32 %7, %7, %8\n\t"
        "v_add_f32_e32 %1, %1, %3\n\t"
        "v_add_f32_e32 %5, %5, %7\n\t"
        "v_cvt_pk_u8_f32 %0, %1, %15, %0\n\t"
        "v_cvt_pk_u8_f32 %0, %5, %16, %0"
        : "+v"(keys), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(G.x), "v"(G.y), "v"(kt.K), "v"(kt.c1), "v"(kt.c2), "v"(kt.c3), "n"(B0), "n"(B0 + 1));
    return keys;
}

// test hook (ldpc_debug_key4): both key forms on arbitrary values, one value pair per thread
__global__ void debug_key4(const float *__restrict__ vals, long long n, float b, const float *__restrict__ thr,
                           uint8_t *__restrict__ out_float, uint8_t *__restrict__ out_compare)
{
    KeyTab kt;
    key_tab_init<kKeyFloat4>(kt, thr, 4);
    const long long i = 2 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const f32x2 v = {vals[i], i + 1 < n ? vals[i + 1] : 0.0f};
    const f32x2 G = (v * b) * kt.K;
    const unsigned kf = key_pair4<0>(G, kt, 0u);
    out_float[i] = (uint8_t)(kf & 0xffu);
    out_compare[i] = (uint8_t)key_of<4>(v.x, b, kt.tb);
    if (i + 1 < n) {
        out_float[i + 1] = (uint8_t)((kf >> 8) & 0xffu);
        out_compare[i + 1] = (uint8_t)key_of<4>(v.y, b, kt.tb);
    }
}

// the LUT is the only LDS object of vn_sweep_q4 (it starts at LDS address 0): read it by absolute byte offset
__device__ __forceinline__ float lut_at(unsigned byte_off)
{
    return *(__attribute__((address_space(3))) const float *)(size_t)byte_off;
}

// INIT: the pass before iteration 0 -- every outgoing message is the LLR itself ("initialize with channel LLRs",
// rcq_decoder.py:514-518), coded with iteration 0's beta and thresholds; no codes are read.
template <int NL, bool ES, bool INIT, int DV>
__device__ __forceinline__ void vn_q4_body(const GraphDev &g, int tile, int j, int off, int lane,
                                           const uint8_t *__restrict__ c2v, const float *__restrict__ llrT,
                                           uint8_t *__restrict__ v2c, float a, int ev, float bv,
                                           const KeyTab &kt, uint64_t *__restrict__ bitsT, const Frozen<4> &fz,
                                           const Pack<float, 4> *l_ext = nullptr)
{
    constexpr int VEC = 4, W = kWave * VEC;
    constexpr int D = DV > 0 ? DV : 1;
    const size_t lane_off = (size_t)lane * VEC;
    const size_t tileE = (size_t)tile * g.E;
    int e[D];
    float bb[D];
    unsigned q[D];
    // edge ids and betas of this variable: lanes off .. off + DV - 1 of the wave's prefetched index registers
#pragma unroll
    for (int k = 0; k < DV; ++k) {
        e[k] = __builtin_amdgcn_readlane(ev, off + k);
        bb[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), off + k));
    }
#pragma unroll
    for (int k = 0; k < DV; ++k)
        q[k] = INIT ? 0u : __builtin_bit_cast(unsigned, ld<uint8_t, VEC>(c2v + (tileE + e[k]) * W + lane_off)) << 2;
    // l_ext: the LLRs come from the caller (transpose_in_q4 holds them in LDS), else from the tile-major array
    const Pack<float, VEC> l = l_ext ? *l_ext : ld<float, VEC>(llrT + ((size_t)tile * g.n + j) * W + lane_off);

    unsigned keys[D], sgns[D];
#pragma unroll
    for (int k = 0; k < DV; ++k) { keys[k] = 0; sgns[k] = 0; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                 // codewords 2h, 2h + 1 as one float pair (v_pk_add/mul_f32)
        f32x2 xs[D];
        if constexpr (!INIT) {
#pragma unroll
            for (int k = 0; k < DV; ++k) {        // codes < 16 here: (code << 2) is the byte offset into the LUT
                xs[k].x = lut_at((q[k] >> (16 * h)) & 0xffu);
                xs[k].y = lut_at((q[k] >> (16 * h + 8)) & 0xffu);
            }
        }
        const f32x2 lh = {l.x[2 * h], l.x[2 * h + 1]};
        if constexpr (ES && !INIT) {
            const f32x2 post = lh + sum_ct<DV, -1, 0, f32x2>(xs);
            const uint64_t b0 = __ballot(post.x < 0.0f), b1 = __ballot(post.y < 0.0f);
            if (lane == 0) {
                bitsT[((size_t)tile * g.n + j) * VEC + 2 * h] = b0;
                bitsT[((size_t)tile * g.n + j) * VEC + 2 * h + 1] = b1;
            }
        }
        f32x2 v[D];
        if constexpr (INIT) {
#pragma unroll
            for (int k = 0; k < DV; ++k) v[k] = lh;
        } else {
        if constexpr (DV >= 1) v[0] = lh + a * sum_ct<DV - 1, 0, 0, f32x2>(xs);
        if constexpr (DV >= 2) v[1] = lh + a * sum_ct<DV - 1, 1, 0, f32x2>(xs);
        if constexpr (DV >= 3) v[2] = lh + a * sum_ct<DV - 1, 2, 0, f32x2>(xs);
        if constexpr (DV >= 4) v[3] = lh + a * sum_ct<DV - 1, 3, 0, f32x2>(xs);
        if constexpr (DV >= 5) v[4] = lh + a * sum_ct<DV - 1, 4, 0, f32x2>(xs);
        if constexpr (DV >= 6) v[5] = lh + a * sum_ct<DV - 1, 5, 0, f32x2>(xs);
        if constexpr (DV >= 7) v[6] = lh + a * sum_ct<DV - 1, 6, 0, f32x2>(xs);
        if constexpr (DV >= 8) v[7] = lh + a * sum_ct<DV - 1, 7, 0, f32x2>(xs);
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            const unsigned u0 = __float_as_uint(v[k].x), u1 = __float_as_uint(v[k].y);
            if constexpr (NL == kKeyFloat4) {
                const f32x2 G = (v[k] * bb[k]) * kt.K;                                      // two packed multiplies per codeword pair
                keys[k] = h == 0 ? key_pair4<0>(G, kt, keys[k]) : key_pair4<2>(G, kt, keys[k]);     // bytes 2h, 2h + 1
            } else {
                const unsigned k0 = key_of<NL>(v[k].x, bb[k], kt.tb), k1 = key_of<NL>(v[k].y, bb[k], kt.tb);
                if (h == 0) keys[k] = k0 | (k1 << 8);
                else keys[k] |= (k0 << 16) | (k1 << 24);
            }
            if (h == 0) {
                sgns[k] = __builtin_amdgcn_perm(u1, u0, 0x0c0c0703u);                       // byte 0 = top of u0, byte 1 = top of u1
            } else {
                sgns[k] = __builtin_amdgcn_perm(u0, sgns[k], 0x0c070100u);
                sgns[k] = __builtin_amdgcn_perm(u1, sgns[k], 0x07020100u);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DV; ++k)
        store_codes4<ES>(v2c + (tileE + e[k]) * W + lane_off, keys[k] | ((sgns[k] >> 1) & 0x40404040u), fz);
}

// Variable sweep, VEC = 4, degrees <= 8, at most 8 levels (the host launches vn_sweep_q<4> otherwise).
// VPW consecutive variables per wave: a degree-2 variable is ~300 instructions of work, less than the wave's prologue
// (kernel arguments, LUT staging + barrier, thresholds) -- unlike the fp32 sweeps this kernel is not HBM-bound.
#ifndef LDPC_VNQ_VPW
#define LDPC_VNQ_VPW 8
#endif
#ifndef LDPC_VNQ_WAVES
#define LDPC_VNQ_WAVES 8          // waves per SIMD the register allocation must leave room for
#endif
template <int NL, bool ES, int VPW, bool INIT = false>
__global__ __launch_bounds__(kBlock, LDPC_VNQ_WAVES) void vn_sweep_q4(GraphDev g, const uint8_t *__restrict__ c2v,
                                                      const float *__restrict__ llrT, uint8_t *__restrict__ v2c,
                                                      const float *__restrict__ alpha_row, const int *__restrict__ alpha_slot,
                                                      const float *__restrict__ lut_cur, int lut_entries,
                                                      const float *__restrict__ beta_next, const int *__restrict__ beta_slot,
                                                      const float *__restrict__ thr_next, int n_levels,
                                                      uint64_t *__restrict__ bitsT, const uint64_t *__restrict__ done,
                                                      int var_blocks)
{
    constexpr int VEC = 4;
    extern __shared__ float lut_s[];
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();      // lut_at relies on that (folds away)
    if constexpr (!INIT) {
        for (int k = threadIdx.x; k < lut_entries; k += kBlock) lut_s[k] = lut_cur[k];
        __syncthreads();
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / var_blocks);
    const int jbase = uni(((blockIdx.x % var_blocks) * kWavesPerBlock + (threadIdx.x >> 6)) * VPW);
    if (jbase >= g.n) return;
    Frozen<VEC> fz;
    fz.bits = 0;
    if constexpr (ES) { if (load_frozen<VEC>(done, tile, lane, fz)) return; }
    KeyTab kt;
    key_tab_init<NL>(kt, thr_next, n_levels);
    // Index data of the wave's VPW variables, fetched ONCE with the lanes in parallel (degrees <= 8: at most 64 edges):
    // var_ptr[jbase .. jbase + VPW] -> CSC edge ids -> beta slots -> betas, alpha slots -> alphas.  Per variable these would
    // be three dependent scalar round trips in front of the row loads; here the variables only pick lanes (v_readlane).
    static_assert(VPW <= 8, "edge ids of a wave's variables are held one per lane");
    const int nv = min(VPW, g.n - jbase);
    const int vp = g.var_ptr[min(jbase + min(lane, VPW), g.n)];
    const int s_base = __builtin_amdgcn_readfirstlane(vp);
    const int n_edges = __builtin_amdgcn_readlane(vp, nv) - s_base;
    int ev = 0;
    float bv = 0.0f, av = 0.0f;
    if (lane < n_edges) {
        ev = g.csc_edge[s_base + lane];
        bv = beta_next[beta_slot[ev]];
    }
    if (!INIT && lane < nv) av = alpha_row[alpha_slot[jbase + lane]];
#pragma unroll 1
    for (int u = 0; u < nv; ++u) {
        const int j = jbase + u;
        const int s0 = __builtin_amdgcn_readlane(vp, u);
        const int dv = __builtin_amdgcn_readlane(vp, u + 1) - s0;
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), u));
#define LDPC_VQ_CASE(D) \
    case D: vn_q4_body<NL, ES, INIT, D>(g, tile, j, s0 - s_base, lane, c2v, llrT, v2c, a, ev, bv, kt, bitsT, fz); break;
        switch (dv) {
            LDPC_VQ_CASE(0) LDPC_VQ_CASE(1) LDPC_VQ_CASE(2) LDPC_VQ_CASE(3) LDPC_VQ_CASE(4)
            LDPC_VQ_CASE(5) LDPC_VQ_CASE(6) LDPC_VQ_CASE(7) LDPC_VQ_CASE(8)
        default: break;
        }
#undef LDPC_VQ_CASE
    }
}

// ------------------------------------------------------------------------------------------
// Layout change + the code pair form's initial pass in ONE kernel (fp32, 256-codeword tiles, 16-byte-aligned caller rows):
// caller rows llr[B][n] -> tile-major llrT[tile][n][256] AND the V2C codes of iteration 0 (every outgoing message is the LLR
// itself, rcq_decoder.py:514-518, coded with iteration 0's beta and thresholds) -- the LLRs are in the block's LDS tile
// anyway, so the separate vn_sweep_q4<INIT> pass (another read of the 4n-byte rows) disappears.
// One block = one tile x kRowsVars variables, staged like vn_last_rows: s[row][variable], row = c * 64 + lane for codeword
// 4 * lane + c, row stride kRowsVars + 1 floats.  Load: pass p covers the 64 codewords of one c (16 threads x 16 bytes per
// codeword; a wave's four codewords have consecutive lanes -> 64 distinct banks per stored float).
// ------------------------------------------------------------------------------------------
template <int NL>
__global__ __launch_bounds__(kRowsThreads, 8) void transpose_in_q4(GraphDev g, const float *__restrict__ llr,
                                                                   float *__restrict__ llrT, uint8_t *__restrict__ v2c,
                                                                   const float *__restrict__ beta0,
                                                                   const int *__restrict__ beta_slot,
                                                                   const float *__restrict__ thr0, int n_levels,
                                                                   long long batch, int var_blocks)
{
    constexpr int VEC = 4, W = kWave * VEC;
    constexpr int kWaves = kRowsThreads / kWave;
    constexpr int kPerWave = kRowsVars / kWaves;                    // variables per wave (4)
    constexpr int kTpr = kRowsVars / 4;                             // threads per codeword run (16 x 16 bytes)
    constexpr int kRpp = kRowsThreads / kTpr;                       // codewords per pass (64)
    static_assert(kRpp == kWave && W / kRpp == VEC, "one pass = the 64 codewords of one c");
    extern __shared__ float rows_smem[];
    float *stage = rows_smem;
    typedef float F4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int tile, chunk;
    if (!rows_block<false>(var_blocks, tile, chunk)) return;        // plain chunk order (see LDPC_ROWS_XCD)
    const int j0 = chunk * kRowsVars;
    {
        const int jq = (threadIdx.x % kTpr) * 4, l = threadIdx.x / kTpr;       // codeword 4 * l + c of the tile
        F4 v[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const long long b = (long long)tile * W + l * VEC + c;
            v[c] = (F4){1.0f, 1.0f, 1.0f, 1.0f};                    // padding codewords: benign positive LLR
            if (b < batch && j0 + jq < g.n)
                v[c] = __builtin_nontemporal_load(reinterpret_cast<const F4 *>(llr + (size_t)b * g.n + j0 + jq));
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float *dst = stage + (size_t)(c * kWave + l) * kRowsStride + jq;
            dst[0] = v[c][0]; dst[1] = v[c][1]; dst[2] = v[c][2]; dst[3] = v[c][3];
        }
    }
    __syncthreads();
    const int jbase = j0 + wave * kPerWave;
    if (jbase >= g.n) return;
    KeyTab kt;
    key_tab_init<NL>(kt, thr0, n_levels);
    // index data of the wave's variables, fetched once with the lanes in parallel (as vn_sweep_q4; <= 32 edges here)
    const int nv = min(kPerWave, g.n - jbase);
    const int vp = g.var_ptr[min(jbase + min(lane, kPerWave), g.n)];
    const int s_base = __builtin_amdgcn_readfirstlane(vp);
    const int n_edges = __builtin_amdgcn_readlane(vp, nv) - s_base;
    int ev = 0;
    float bv = 0.0f;
    if (lane < n_edges) {
        ev = g.csc_edge[s_base + lane];
        bv = beta0[beta_slot[ev]];
    }
    Frozen<VEC> fz;
    fz.bits = 0;
#pragma unroll 1
    for (int u = 0; u < nv; ++u) {
        const int j = jbase + u, jj = j - j0;
        const int s0 = __builtin_amdgcn_readlane(vp, u);
        const int dv = __builtin_amdgcn_readlane(vp, u + 1) - s0;
        Pack<float, VEC> l;
#pragma unroll
        for (int c = 0; c < VEC; ++c) l.x[c] = stage[(size_t)(c * kWave + lane) * kRowsStride + jj];
        st<float, VEC>(llrT + ((size_t)tile * g.n + j) * W + (size_t)lane * VEC, l);
#define LDPC_TQ_CASE(D) \
    case D: vn_q4_body<NL, false, true, D>(g, tile, j, s0 - s_base, lane, nullptr, nullptr, v2c, 0.0f, ev, bv, kt, nullptr, fz, &l); break;
        switch (dv) {
            LDPC_TQ_CASE(0) LDPC_TQ_CASE(1) LDPC_TQ_CASE(2) LDPC_TQ_CASE(3) LDPC_TQ_CASE(4)
            LDPC_TQ_CASE(5) LDPC_TQ_CASE(6) LDPC_TQ_CASE(7) LDPC_TQ_CASE(8)
        default: break;
        }
#undef LDPC_TQ_CASE
    }
}

// ------------------------------------------------------------------------------------------
// Fused RCQ iteration ("gather" form): ONE kernel per iteration, no V2C array at all.
//
// With 1-byte C2V codes the variable->check messages are cheaper to RECOMPUTE than to store: the check sweep of
// iteration t needs, for its edge (i, j),   v2c = llr[j] + alpha_{t-1} * sum_{i' != i} deq_{t-1}(code_{t-1}[i', j])
// (rcq_decoder.py:257 / :575), i.e. the LLR row of j and the dv(j)-1 code rows of j's OTHER edges -- 4 + (dv-1)
// bytes per edge and codeword instead of writing and re-reading a 4-byte V2C value, and the variable sweep
// disappears from the iteration.  Per codeword and iteration the kernel issues 4E + sum_j dv(dv-1) bytes of reads
// (345 KB on the (16200,7200) code, most of them cache hits: the distinct bytes are 4n + E) and writes E code
// bytes, against 10E + 4n = 551 KB in HBM for the two-sweep form.  Codes are double-buffered (every check
// reads the previous iteration's codes of its neighbours' other edges).
//
// Same wave mapping as cn_sweep (wave = one check x W codewords, every index wave-uniform); per-edge gather
// metadata is precomputed by the host: meta[e] = {variable, offset into nbr, dv-1, alpha column}, nbr[] = CSR
// edge ids of the variable's other edges in ascending check order (the reference's summation order).
// ------------------------------------------------------------------------------------------
//   gat_meta [E + 1]              (one padding entry: the prefetch of "the edge after the last" stays in bounds)
//   gat_nbr  [sum_j dv(dv-1) + 8] (padded: eight ids are always fetched)

// Row loads of the gather kernel: buffer_load with the tile's base in a wave-uniform 128-bit descriptor, the row's byte
// offset in an SGPR (soffset) and the lane's offset inside the row in one VGPR (voffset) -- no VALU address arithmetic
// and no 64-bit VGPR address pairs.  aux = 2: non-temporal, as the other message streams.
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> buf_ld(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_byte, unsigned row_byte)
{
    constexpr int kBytes = (int)sizeof(T) * V;
    static_assert(kBytes == 1 || kBytes == 4 || kBytes == 16, "row element of 1, 4 or 16 bytes per lane");
    union { Pack<T, V> k; unsigned char b; unsigned w; unsigned q __attribute__((ext_vector_type(4))); } u;
    if constexpr (kBytes == 1) u.b = __builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)lane_byte, (int)row_byte, 2);
    else if constexpr (kBytes == 4) u.w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_byte, (int)row_byte, 2);
    else u.q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_byte, (int)row_byte, 2);
    return u.k;
}

// v2c = llr + alpha * sum(reconstructed codes) for CNT other edges whose code rows are already in registers
template <int VEC, int CNT>
__device__ __forceinline__ Pack<float, VEC> gather_v2c(const Pack<uint8_t, VEC> (&q)[7], const Pack<float, VEC> &l, float a,
                                                       const float *__restrict__ lut)
{
    Pack<float, VEC> out;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        float xs[CNT > 0 ? CNT : 1];
#pragma unroll
        for (int k = 0; k < CNT; ++k) xs[k] = lut[q[k].x[c]];
        out.x[c] = l.x[c] + a * sum_ct<CNT, -1, 0, float>(xs);      // same expression as vn_body (alpha * 0 for dv = 1)
    }
    return out;
}

// GRP: edges whose rows (LLR + code rows of the other edges) are requested together before any of them is
// consumed.  The kernel is bound by memory latency, not bandwidth (its reads are L2/MALL hits of 256 B - 1 KiB rows):
// one exposed round trip per GRP edges instead of one per edge.
template <int VEC, int NL, bool BPC, int CPW, int GRP>
__global__ __launch_bounds__(kBlock, LDPC_GATHER_WAVES) void cn_gather(GraphDev g, const int4 *__restrict__ gat_meta,
                                                    const int *__restrict__ gat_nbr, const float *__restrict__ llrT,
                                                    const uint8_t *__restrict__ codes_in,
                                                    uint8_t *__restrict__ codes_out,
                                                    const float *__restrict__ beta_row,
                                                    const int *__restrict__ beta_slot,
                                                    const float *__restrict__ alpha_prev_row,
                                                    const float *__restrict__ thr, int n_levels,
                                                    const float *__restrict__ lut_prev, int lut_entries,
                                                    const uint64_t *__restrict__ done, int check_blocks, int xcd_tiles)
{
    constexpr int W = kWave * VEC;
    // reconstruction values of the PREVIOUS iteration's quantiser (the one that produced codes_in), at LDS offset 0.
    // Codewords the early-stop latch froze earlier would need their own quantiser's table -- but their outputs are
    // never stored (store_latched keeps their old codes), so any table will do for them.
    extern __shared__ float gather_lut_s[];
    for (int k = threadIdx.x; k < lut_entries; k += kBlock) gather_lut_s[k] = lut_prev[k];
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    // xcd_tiles > 0: XCD-affine mapping -- workgroups are dealt round-robin to the 8 XCDs, so workgroup b works on tile
    // 8*(b/8/check_blocks) + b%8: every tile is processed by ONE XCD and the re-reads of its rows (each LLR row dv times, each
    // code row dv-1 times) meet in that XCD's L2 instead of in eight
    int tile_, cblk_;
    if (xcd_tiles > 0) {
        const int k = blockIdx.x >> 3;
        tile_ = (k / check_blocks) * 8 + (blockIdx.x & 7);
        cblk_ = k % check_blocks;
        if (tile_ >= xcd_tiles) return;
    } else {
        tile_ = blockIdx.x / check_blocks;
        cblk_ = blockIdx.x % check_blocks;
    }
    const int tile = uni(tile_);
    const int ibase = uni((cblk_ * kWavesPerBlock + (threadIdx.x >> 6)) * CPW);
    if (ibase >= g.m) return;

    Frozen<VEC> fz;
    const bool all_frozen = load_frozen<VEC>(done, tile, lane, fz);
    float th[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) th[q] = (q < n_levels) ? thr[q] : __builtin_nanf("");
    const unsigned lane_f = (unsigned)lane * VEC * (unsigned)sizeof(float);     // byte offset inside an LLR row
    const unsigned lane_b = (unsigned)lane * VEC;                               // byte offset inside a code row
    const uint8_t *cin_tile = codes_in + (size_t)tile * g.E * W;
    uint8_t *cout_tile = codes_out + (size_t)tile * g.E * W;
    // descriptors of this tile's LLR rows and code rows (the host admits only graphs whose tile fits 31 bits of bytes)
    const __amdgpu_buffer_rsrc_t llr_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(llrT + (size_t)tile * g.n * W), 0, g.n * W * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t cin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(cin_tile), 0, g.E * W, 0x00020000);

    for (int cc_ = 0; cc_ < CPW; ++cc_) {
    const int i = ibase + cc_;
    if (i >= g.m) break;
    const int e0 = uni(g.check_ptr[i]);
    const int dc = uni(g.check_ptr[i + 1]) - e0;
    if (dc == 0) continue;
    if (all_frozen) {                              // the codes are double-buffered: carry the tile's latched rows over
        for (int t = 0; t < dc; ++t)
            st<uint8_t, VEC>(cout_tile + (size_t)(e0 + t) * W + lane_b, ld<uint8_t, VEC>(cin_tile + (size_t)(e0 + t) * W + lane_b));
        continue;
    }

    // variable->check messages of the edges t0 .. t0+GRP-1 of this check, recomputed from the previous codes
    auto group = [&](int t0, auto &&consume) {
        int4 md[GRP];
        int nb[GRP][8];
        Pack<float, VEC> l[GRP];
        Pack<uint8_t, VEC> q[GRP][7];
        // scalar phase: the group's metadata, then its neighbour lists (eight ids each, the lists are padded), each
        // batch issued back to back -- two scalar round trips per group
#pragma unroll
        for (int u = 0; u < GRP; ++u) md[u] = gat_meta[min(e0 + t0 + u, g.E)];          // entry E is padding (no other edges)
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int *np = gat_nbr + md[u].y;
#pragma unroll
            for (int k = 0; k < 8; ++k) nb[u][k] = np[k];
        }
        // vector phase: every row of the group requested before the first is consumed
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            if (t0 + u < dc) {
                l[u] = buf_ld<float, VEC>(llr_rs, lane_f, (unsigned)md[u].x * (unsigned)(W * sizeof(float)));
                const int cnt = md[u].z;
#pragma unroll
                for (int k = 0; k < 7; ++k)
                    if (k < cnt) q[u][k] = buf_ld<uint8_t, VEC>(cin_rs, lane_b, (unsigned)nb[u][k] * (unsigned)W);
            }
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            if (t0 + u < dc) {
                const float a = alpha_prev_row[md[u].w];
                Pack<float, VEC> v;
                switch (md[u].z) {
                case 0: v = gather_v2c<VEC, 0>(q[u], l[u], a, gather_lut_s); break;
                case 1: v = gather_v2c<VEC, 1>(q[u], l[u], a, gather_lut_s); break;
                case 2: v = gather_v2c<VEC, 2>(q[u], l[u], a, gather_lut_s); break;
                case 3: v = gather_v2c<VEC, 3>(q[u], l[u], a, gather_lut_s); break;
                case 4: v = gather_v2c<VEC, 4>(q[u], l[u], a, gather_lut_s); break;
                case 5: v = gather_v2c<VEC, 5>(q[u], l[u], a, gather_lut_s); break;
                case 6: v = gather_v2c<VEC, 6>(q[u], l[u], a, gather_lut_s); break;
                default: v = gather_v2c<VEC, 7>(q[u], l[u], a, gather_lut_s); break;   // host admits dv <= 8 here
                }
                consume(u, v);
            }
        }
    };

    float m1[VEC], m2[VEC];
    int idx[VEC];
    uint32_t sm[VEC];
    unsigned par[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { m1[c] = inf_of<float>(); m2[c] = inf_of<float>(); idx[c] = 0; sm[c] = 0; par[c] = 0; }

    for (int t0 = 0; t0 < dc; t0 += GRP) {
        group(t0, [&](int u, const Pack<float, VEC> &v) {
            const int t = t0 + u;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const float a = __builtin_fabsf(v.x[c]);
                const unsigned sb = signbit_of<float>(v.x[c]);
                par[c] ^= sb;
                sm[c] |= sb << (t & 31);
                if (a < m1[c]) { m2[c] = m1[c]; m1[c] = a; idx[c] = t; }
                else if (a < m2[c]) { m2[c] = a; }
            }
        });
    }
    if (dc == 1) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) m2[c] = m1[c];
    }
    const bool wide = dc > 32;                     // sign masks hold 32 edges; wider checks recompute their inputs

    auto level = [&](float mag) {
        int lvl = 0;
        if constexpr (NL > 0) {
#pragma unroll
            for (int q = 1; q < NL; ++q) lvl = (mag >= th[q]) ? q : lvl;
        } else if (n_levels <= 8) {
#pragma unroll
            for (int q = 1; q < 8; ++q) lvl = (mag >= th[q]) ? q : lvl;
        } else {
            for (int q = 1; q < n_levels; ++q) lvl = (mag >= thr[q]) ? q : lvl;
        }
        return lvl;
    };
    unsigned cc1[VEC], cc2[VEC];
    if constexpr (BPC) {                           // one beta per check: four candidate codes per codeword (see cn_sweep)
        const float b = beta_row[beta_slot[e0]];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const float p1 = b * m1[c], p2 = b * m2[c];
            const float a1 = __builtin_fabsf(p1), a2 = __builtin_fabsf(p2);
            const unsigned l1 = (unsigned)level(a1), l2 = (unsigned)level(a2);
            const unsigned s1 = signbit_of<float>(p1), s2 = signbit_of<float>(p2);
            const unsigned z1 = a1 > 0.0f ? (unsigned)n_levels : 0u, z2 = a2 > 0.0f ? (unsigned)n_levels : 0u;
            cc1[c] = (l1 + (s1 ? z1 : 0u)) | ((l1 + (s1 ? 0u : z1)) << 8);
            cc2[c] = (l2 + (s2 ? z2 : 0u)) | ((l2 + (s2 ? 0u : z2)) << 8);
        }
    }
    for (int t0 = 0; t0 < dc; t0 += GRP) {
        Pack<float, VEC> re[GRP];
        if (wide) group(t0, [&](int u, const Pack<float, VEC> &v) { re[u] = v; });
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int t = t0 + u;
            if (t >= dc) break;
            Pack<uint8_t, VEC> o;
            float b = 0.0f;
            if constexpr (!BPC) b = beta_row[beta_slot[e0 + t]];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const unsigned own = wide ? signbit_of<float>(re[u].x[c]) : ((sm[c] >> (t & 31)) & 1u);
                const unsigned neg = par[c] ^ own;
                if constexpr (BPC) {
                    const unsigned cc = (t == idx[c]) ? cc2[c] : cc1[c];
                    o.x[c] = (uint8_t)((cc >> (neg * 8u)) & 0xffu);
                } else {
                    const float raw = (t == idx[c]) ? m2[c] : m1[c];
                    const float w = flip_sign<float>(b * raw, neg);
                    const int lvl = level(__builtin_fabsf(w));
                    o.x[c] = (uint8_t)(((w < 0.0f) ? n_levels : 0) + lvl);                    // rcq_decoder.py:88-89
                }
            }
            store_latched<uint8_t, VEC>(cout_tile + (size_t)(e0 + t) * W + lane_b, done ? cin_tile + (size_t)(e0 + t) * W + lane_b : nullptr, o, fz);
        }
    }
    }   // checks of this wave
}

// ------------------------------------------------------------------------------------------
// Syndrome + early-stop latch.  One block per tile; H @ bits mod 2 becomes an XOR of
// the ballot words of a check's variables (64 codewords per word).
//   latch = true : done |= (syndrome == 0); iterations = it+1 for newly done codewords
//   latch = false: fixed-T mode, done := (syndrome == 0)   (reported as `success`)
// ------------------------------------------------------------------------------------------
// XOR of the decision words of check i's variables.  Groups of four edges: the four variable indices are requested together, then
// the four word rows -- two load round trips per group instead of eight dependent ones (the kernel is bound by that chain).
template <int VEC>
__device__ __forceinline__ void syndrome_row(const GraphDev &g, const uint64_t *__restrict__ b, int i, uint64_t (&x)[VEC])
{
    const int e1 = g.check_ptr[i + 1];
    for (int e = g.check_ptr[i]; e < e1; e += 4) {
        const int last = e1 - 1, cnt = e1 - e;                       // a short last group re-reads its last edge and masks it out
        const int v0 = g.var_idx[e], v1 = g.var_idx[min(e + 1, last)], v2 = g.var_idx[min(e + 2, last)], v3 = g.var_idx[min(e + 3, last)];
        const uint64_t *w0 = b + (size_t)v0 * VEC, *w1 = b + (size_t)v1 * VEC, *w2 = b + (size_t)v2 * VEC, *w3 = b + (size_t)v3 * VEC;
        uint64_t t0[VEC], t1[VEC], t2[VEC], t3[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) { t0[c] = w0[c]; t1[c] = w1[c]; t2[c] = w2[c]; t3[c] = w3[c]; }
        const uint64_t k1 = cnt > 1 ? ~0ull : 0ull, k2 = cnt > 2 ? ~0ull : 0ull, k3 = cnt > 3 ? ~0ull : 0ull;
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c] ^= (t0[c] ^ (t1[c] & k1)) ^ ((t2[c] & k2) ^ (t3[c] & k3));
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void syndrome_latch(GraphDev g, const uint64_t *__restrict__ bitsT,
                                                         uint64_t *__restrict__ done,
                                                         int *__restrict__ iters, int it_plus_1, int latch)
{
    constexpr int W = kWave * VEC;
    __shared__ unsigned long long unsat[VEC];
    const int tile = blockIdx.x;
    if (threadIdx.x < VEC) unsat[threadIdx.x] = 0ull;
    __syncthreads();
    if (latch) {
        bool all = true;
#pragma unroll
        for (int c = 0; c < VEC; ++c) all = all && (done[(size_t)tile * VEC + c] == ~0ull);
        if (all) return;
    }
    uint64_t acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0;
    const uint64_t *b = bitsT + (size_t)tile * g.n * VEC;
    for (int i = threadIdx.x; i < g.m; i += kBlock) {
        uint64_t x[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c] = 0;
        syndrome_row<VEC>(g, b, i, x);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] |= x[c];
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c)
        if (acc[c]) atomicOr(&unsat[c], (unsigned long long)acc[c]);
    __syncthreads();
    if ((int)threadIdx.x < W) {
        const int w = threadIdx.x, c = w % VEC, l = w / VEC;
        const uint64_t sat = ~(uint64_t)unsat[c];
        if (latch) {
            const uint64_t was = done[(size_t)tile * VEC + c];
            if (((sat & ~was) >> l) & 1ull) iters[(size_t)tile * W + w] = it_plus_1;
        }
    }
    __syncthreads();   // all reads of `done` above precede the update below
    if ((int)threadIdx.x < VEC) {
        const int c = threadIdx.x;
        const uint64_t sat = ~(uint64_t)unsat[c];
        done[(size_t)tile * VEC + c] = latch ? (done[(size_t)tile * VEC + c] | sat) : sat;
    }
}

// The same for FEW tiles (small batches of a code that does not fit LDS: the reference's one-codeword call on the (16200,7200)
// code ran one 256-thread block over all 9000 checks, ~60 us per iteration): `chunks` blocks per tile, each over every
// chunks-th group of 256 checks; partial words are OR-ed into unsat_g[tile][VEC], and the block that draws the last ticket of its
// tile applies the latch and leaves unsat_g / ticket zeroed for the next launch.  No block can read `done` after the latch of
// its own launch: the latch needs every block's ticket, and a tile whose codewords are all done draws none at all.
template <int VEC>
__global__ __launch_bounds__(kBlock) void syndrome_latch_chunks(GraphDev g, const uint64_t *__restrict__ bitsT,
                                                                uint64_t *__restrict__ done, int *__restrict__ iters,
                                                                int it_plus_1, int latch, int chunks,
                                                                unsigned long long *__restrict__ unsat_g, int *__restrict__ ticket)
{
    constexpr int W = kWave * VEC;
    __shared__ unsigned long long unsat[VEC];
    __shared__ int is_last;
    const int tile = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    if (threadIdx.x < VEC) unsat[threadIdx.x] = 0ull;
    __syncthreads();
    if (latch) {
        bool all = true;
#pragma unroll
        for (int c = 0; c < VEC; ++c) all = all && (done[(size_t)tile * VEC + c] == ~0ull);
        if (all) return;
    }
    uint64_t acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0;
    const uint64_t *b = bitsT + (size_t)tile * g.n * VEC;
    for (int i = chunk * kBlock + threadIdx.x; i < g.m; i += chunks * kBlock) {
        uint64_t x[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c] = 0;
        syndrome_row<VEC>(g, b, i, x);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] |= x[c];
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c)
        if (acc[c]) atomicOr(&unsat[c], (unsigned long long)acc[c]);
    __syncthreads();
    if ((int)threadIdx.x < VEC && unsat[threadIdx.x]) atomicOr(&unsat_g[(size_t)tile * VEC + threadIdx.x], unsat[threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&ticket[tile], 1) == chunks - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if ((int)threadIdx.x < VEC) unsat[threadIdx.x] = atomicOr(&unsat_g[(size_t)tile * VEC + threadIdx.x], 0ull);   // the tile's OR, read at L2
    __syncthreads();
    if ((int)threadIdx.x < W) {
        const int w = threadIdx.x, c = w % VEC, l = w / VEC;
        const uint64_t sat = ~(uint64_t)unsat[c];
        if (latch) {
            const uint64_t was = done[(size_t)tile * VEC + c];
            if (((sat & ~was) >> l) & 1ull) iters[(size_t)tile * W + w] = it_plus_1;
        }
    }
    __syncthreads();   // all reads of `done` above precede the update below
    if ((int)threadIdx.x < VEC) {
        const int c = threadIdx.x;
        const uint64_t sat = ~(uint64_t)unsat[c];
        done[(size_t)tile * VEC + c] = latch ? (done[(size_t)tile * VEC + c] | sat) : sat;
        unsat_g[(size_t)tile * VEC + c] = 0ull;
    }
    if (threadIdx.x == 0) ticket[tile] = 0;
}

// ------------------------------------------------------------------------------------------
// Reference-compatible "layered" RCQ schedule (RCQMinSumDecoder._decode_layered,
// rcq_decoder.py:281-350).  As written there the per-check message matrix is re-created for every
// check, so the "subtract the previous message" step never finds anything to subtract on a code
// with more than one check: posteriors accumulate every check's quantised message, check after
// check, inside and across iterations.  That sequential update is what this kernel reproduces.
// One wave owns W codewords (one tile of the posterior array post[tile][n][W], initialised with the
// LLRs) and walks the checks in order, so no inter-wave synchronisation exists at all; the
// parallelism is the batch.  Per check: pass 1 gathers the dc posterior rows (min1/min2/sign
// parity), pass 2 re-reads each row, quantises-reconstructs sign*min and adds it in place.
// ------------------------------------------------------------------------------------------
// PAPER = false: the schedule as the reference EXECUTES it (see above: the previous message is never subtracted).
// PAPER = true : the layered schedule the reference's comments describe and the RCQ paper defines (LDPC_SCHED_LAYERED):
//                the check's previous message (its 1-byte code, kept per edge in `codes`, reconstructed with the
//                quantiser of the iteration that produced it) is subtracted from the posterior before the update,
//                the new one is added and its code stored.  Nothing in the reference executes this: parity unpinned.
template <int VEC, bool PAPER>
__global__ __launch_bounds__(kWave) void layered_rcq(GraphDev g, float *__restrict__ post,
                                                     const float *__restrict__ thresholds, int n_levels,
                                                     const int *__restrict__ q_of_iter, int T, int early_stop,
                                                     uint64_t *__restrict__ bitsT, uint64_t *__restrict__ done,
                                                     int *__restrict__ iters, int max_dc,
                                                     uint8_t *__restrict__ codes)
{
    constexpr int W = kWave * VEC;
    const int lane = threadIdx.x, tile = blockIdx.x;
    float *P = post + (size_t)tile * g.n * W + (size_t)lane * VEC;
    uint8_t *Cd = PAPER ? codes + (size_t)tile * g.E * W + (size_t)lane * VEC : nullptr;
    unsigned frozen = 0;                                    // bit c: codeword c of this lane has stopped
#pragma unroll
    for (int c = 0; c < VEC; ++c) frozen |= (unsigned)((done[(size_t)tile * VEC + c] >> lane) & 1ull) << c;
    const unsigned kAll = (1u << VEC) - 1u;

    auto syndrome = [&]() {                                 // bit c set: some check of codeword c is unsatisfied
        unsigned unsat = 0;
        for (int i = 0; i < g.m; ++i) {
            const int e0 = g.check_ptr[i], e1 = g.check_ptr[i + 1];
            unsigned par = 0;
            for (int e = e0; e < e1; ++e) {
                const Pack<float, VEC> v = ld<float, VEC>(P + (size_t)g.var_idx[e] * W);
#pragma unroll
                for (int c = 0; c < VEC; ++c) par ^= (v.x[c] < 0.0f ? 1u : 0u) << c;
            }
            unsat |= par;
        }
        return unsat;
    };
    // value of a stored code under the quantiser that produced it: (1 - 2*sign) * tau[level]
    auto rec_of = [&](const float *thr_q, unsigned code) {
        const unsigned lvl = code >= (unsigned)n_levels ? code - (unsigned)n_levels : code;
        float mag = thr_q[0];
        for (int q = 1; q < n_levels; ++q) mag = (lvl == (unsigned)q) ? thr_q[q] : mag;
        return flip_sign<float>(mag, code >= (unsigned)n_levels ? 1u : 0u);
    };

    for (int it = 0; it < T; ++it) {
        if (early_stop && __ballot(frozen != kAll) == 0ull) break;
        const float *thr = thresholds + (size_t)q_of_iter[it] * n_levels;
        const float *thr_prev = thresholds + (size_t)q_of_iter[it > 0 ? it - 1 : 0] * n_levels;
        if constexpr (VEC == 1) {
            // The walk is ONE dependent chain per wave (a check reads what the previous one wrote), so what counts is
            // memory round trips per check.  With at most kHeld edges per check: the indices of check i+1 are fetched
            // while check i is processed, all rows of a check are requested at once and stay in registers for the
            // update -- one exposed round trip per check instead of index loads, unroll groups and re-reads in series.
            constexpr int kHeld = 32;
            if (max_dc <= kHeld) {
                int var_n[kHeld];
                int dc_n = 0, e0_n = 0;
                auto fetch_idx = [&](int i) {
                    e0_n = g.check_ptr[i];
                    dc_n = g.check_ptr[i + 1] - e0_n;
#pragma unroll
                    for (int t = 0; t < kHeld; ++t)
                        if (t < max_dc) var_n[t] = g.var_idx[min(e0_n + t, g.E - 1)];      // past the check: unused
                };
                if (g.m > 0) fetch_idx(0);
                for (int i = 0; i < g.m; ++i) {
                    int var[kHeld];
                    const int dc = uni(dc_n), e0 = uni(e0_n);
#pragma unroll
                    for (int t = 0; t < kHeld; ++t) var[t] = var_n[t];
                    if (i + 1 < g.m) fetch_idx(i + 1);
                    float x[kHeld];
                    unsigned old[kHeld];
#pragma unroll
                    for (int t = 0; t < kHeld; ++t)
                        if (t < dc) {
                            x[t] = P[(size_t)var[t] * W];
                            if (PAPER && it > 0) old[t] = Cd[(size_t)(e0 + t) * W];
                        }
                    if (PAPER && it > 0) {
#pragma unroll
                        for (int t = 0; t < kHeld; ++t)
                            if (t < dc) x[t] = x[t] - rec_of(thr_prev, old[t]);          // "subtract previous C2V messages" (:300-302)
                    }
                    float m1 = inf_of<float>(), m2 = inf_of<float>();
                    unsigned par = 0;
#pragma unroll
                    for (int t = 0; t < kHeld; ++t)
                        if (t < dc) {
                            const float a = __builtin_fabsf(x[t]);
                            par ^= signbit_of<float>(x[t]);
                            if (a < m1) { m2 = m1; m1 = a; }
                            else if (a < m2) { m2 = a; }
                        }
                    if (dc == 1) m2 = m1;
                    if (frozen != kAll) {
#pragma unroll
                        for (int t = 0; t < kHeld; ++t)
                            if (t < dc) {
                                const float a = __builtin_fabsf(x[t]);
                                const float raw = (a == m1) ? m2 : m1;     // arg-min edge; ties make min2 == min1
                                const float w = flip_sign<float>(raw, par ^ signbit_of<float>(x[t]));
                                const float mag = __builtin_fabsf(w);
                                float rec = thr[0];
                                unsigned lvl = 0;
                                for (int q = 1; q < n_levels; ++q) { const bool ge = mag >= thr[q]; rec = ge ? thr[q] : rec; lvl = ge ? (unsigned)q : lvl; }
                                P[(size_t)var[t] * W] = x[t] + flip_sign<float>(rec, (w < 0.0f) ? 1u : 0u);
                                if (PAPER) Cd[(size_t)(e0 + t) * W] = (uint8_t)(((w < 0.0f) ? (unsigned)n_levels : 0u) + lvl);
                            }
                    }
                }
                goto checks_done;
            }
        }
        for (int i = 0; i < g.m; ++i) {
            const int e0 = uni(g.check_ptr[i]);
            const int dc = uni(g.check_ptr[i + 1]) - e0;
            if (dc == 0) continue;
            float m1[VEC], m2[VEC];
            unsigned par[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) { m1[c] = inf_of<float>(); m2[c] = inf_of<float>(); par[c] = 0; }
            // PAPER: the posterior minus the check's previous message, recomputed identically in both passes
            auto input = [&](int t) {
                Pack<float, VEC> v = ld<float, VEC>(P + (size_t)g.var_idx[e0 + t] * W);
                if (PAPER && it > 0) {
                    const Pack<uint8_t, VEC> oc = ld<uint8_t, VEC>(Cd + (size_t)(e0 + t) * W);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) v.x[c] = v.x[c] - rec_of(thr_prev, oc.x[c]);
                }
                return v;
            };
#pragma unroll 4
            for (int t = 0; t < dc; ++t) {
                const Pack<float, VEC> v = input(t);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const float a = __builtin_fabsf(v.x[c]);
                    par[c] ^= signbit_of<float>(v.x[c]);
                    if (a < m1[c]) { m2[c] = m1[c]; m1[c] = a; }
                    else if (a < m2[c]) { m2[c] = a; }
                }
            }
            if (dc == 1) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) m2[c] = m1[c];
            }
#pragma unroll 2
            for (int t = 0; t < dc; ++t) {
                float *row = P + (size_t)g.var_idx[e0 + t] * W;
                Pack<float, VEC> v = input(t);
                Pack<uint8_t, VEC> nc;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const float a = __builtin_fabsf(v.x[c]);
                    const float raw = (a == m1[c]) ? m2[c] : m1[c];       // arg-min edge; ties make min2 == min1
                    const float w = flip_sign<float>(raw, par[c] ^ signbit_of<float>(v.x[c]));
                    const float mag = __builtin_fabsf(w);
                    float rec = thr[0];
                    unsigned lvl = 0;
                    for (int q = 1; q < n_levels; ++q) { const bool ge = mag >= thr[q]; rec = ge ? thr[q] : rec; lvl = ge ? (unsigned)q : lvl; }
                    const float msg = flip_sign<float>(rec, (w < 0.0f) ? 1u : 0u);
                    nc.x[c] = (uint8_t)(((w < 0.0f) ? (unsigned)n_levels : 0u) + lvl);
                    if ((frozen >> c) & 1u) {
                        // a stopped codeword keeps posterior AND code: put back what `input` took off
                        if (PAPER && it > 0) v.x[c] = ld<float, VEC>(row).x[c];
                    } else {
                        v.x[c] = v.x[c] + msg;
                    }
                }
                if (frozen != kAll) {
                    st<float, VEC>(row, v);
                    if (PAPER) {
                        if (frozen == 0) {
                            st<uint8_t, VEC>(Cd + (size_t)(e0 + t) * W, nc);
                        } else {
#pragma unroll
                            for (int c = 0; c < VEC; ++c)
                                if (!((frozen >> c) & 1u)) Cd[(size_t)(e0 + t) * W + c] = nc.x[c];
                        }
                    }
                }
            }
        }
    checks_done:
        if (early_stop) {
            const unsigned newly = ~syndrome() & ~frozen & kAll;
#pragma unroll
            for (int c = 0; c < VEC; ++c)
                if ((newly >> c) & 1u) iters[(size_t)tile * W + lane * VEC + c] = it + 1;
            frozen |= newly;
        }
    }
    unsigned ok = frozen;
    if (!early_stop) ok = ~syndrome() & kAll;               // fixed-T mode: success = final syndrome is zero
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const uint64_t m = __ballot((ok >> c) & 1u);
        if (lane == 0) done[(size_t)tile * VEC + c] = m;
    }
    for (int j = 0; j < g.n; ++j) {                         // hard decisions as ballots, like the sweep engine
        const Pack<float, VEC> v = ld<float, VEC>(P + (size_t)j * W);
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const uint64_t m = __ballot(v.x[c] < 0.0f);
            if (lane == 0) bitsT[((size_t)tile * g.n + j) * VEC + c] = m;
        }
    }
}

// done masks: padding codewords (>= batch) start frozen; iterations start at T
template <int VEC>
__global__ void init_state(uint64_t *__restrict__ done, int *__restrict__ iters, long long batch,
                           int tiles, int T, unsigned long long *__restrict__ unsat_g, int *__restrict__ ticket)
{
    constexpr int W = kWave * VEC;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < (long long)tiles * W) iters[gid] = T;
    if (gid < tiles) ticket[gid] = 0;
    if (gid < (long long)tiles * VEC) {
        unsat_g[gid] = 0ull;
        const int tile = (int)(gid / VEC), c = (int)(gid % VEC);
        uint64_t mask = 0;
        for (int l = 0; l < kWave; ++l) {
            long long b = (long long)tile * W + (long long)l * VEC + c;
            if (b >= batch) mask |= 1ull << l;
        }
        done[gid] = mask;
    }
}

// ------------------------------------------------------------------------------------------
// Layout changes at the API edge: llr[B][n] -> llrT[tile][n][W]; posterior / bits back.
// ------------------------------------------------------------------------------------------
// Layout changes at the boundary: caller rows [B][n] <-> workspace tiles [tile][n][W].  One block moves 64 codewords x
// kTransposeBytes / sizeof(T) variables through LDS, so that BOTH sides are touched in runs of at least 256 contiguous
// bytes (a wave reads 256 B of one codeword's row and writes the 64-codeword run of one variable's tile row) -- with
// 32-variable blocks the caller side was touched 128 B at a time and the two kernels ran at 2.5-2.7 TB/s.
constexpr int kTransposeBytes = 512;
template <typename T> constexpr int transpose_vars() { return kTransposeBytes / (int)sizeof(T); }

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void transpose_in(const T *__restrict__ llr, T *__restrict__ llrT,
                                                       long long batch, int n, int var_chunks)
{
    constexpr int W = kWave * VEC;
    constexpr int JT = transpose_vars<T>();
    __shared__ T s[JT][kWave + 1];
    const int chunk = blockIdx.x % var_chunks, sub = (blockIdx.x / var_chunks) % VEC, tile = blockIdx.x / (var_chunks * VEC);
    const int j0 = chunk * JT, w0 = sub * kWave;
    for (int idx = threadIdx.x; idx < kWave * JT; idx += kBlock) {
        const int r = idx / JT, jj = idx % JT;
        const long long b = (long long)tile * W + w0 + r;
        T v = (T)1;                                     // padding codewords: benign positive LLR
        if (b < batch && j0 + jj < n) v = llr[(size_t)b * n + j0 + jj];
        s[jj][r] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < kWave * JT; idx += kBlock) {
        const int jj = idx / kWave, w = idx % kWave;
        if (j0 + jj < n) llrT[((size_t)tile * n + j0 + jj) * W + w0 + w] = s[jj][w];
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void transpose_out(const T *__restrict__ postT,
                                                        const uint64_t *__restrict__ bitsT,
                                                        T *__restrict__ posterior, int *__restrict__ bits,
                                                        long long batch, int n, int var_chunks)
{
    constexpr int W = kWave * VEC;
    constexpr int JT = transpose_vars<T>();
    __shared__ T s[JT][kWave + 1];
    __shared__ uint64_t sb[JT][VEC];
    const int chunk = blockIdx.x % var_chunks, sub = (blockIdx.x / var_chunks) % VEC, tile = blockIdx.x / (var_chunks * VEC);
    const int j0 = chunk * JT, w0 = sub * kWave;
    if (posterior) {
        for (int idx = threadIdx.x; idx < kWave * JT; idx += kBlock) {
            const int jj = idx / kWave, w = idx % kWave;
            if (j0 + jj < n) s[jj][w] = postT[((size_t)tile * n + j0 + jj) * W + w0 + w];
        }
    }
    if (bits) {
        for (int idx = threadIdx.x; idx < JT * VEC; idx += kBlock) {
            const int jj = idx / VEC, c = idx % VEC;
            sb[jj][c] = (j0 + jj < n) ? bitsT[((size_t)tile * n + j0 + jj) * VEC + c] : 0ull;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < kWave * JT; idx += kBlock) {
        const int r = idx / JT, jj = idx % JT;
        const int rw = w0 + r;                          // codeword inside the tile: lane rw / VEC, slot rw % VEC
        const long long b = (long long)tile * W + rw;
        if (b < batch && j0 + jj < n) {
            if (posterior) posterior[(size_t)b * n + j0 + jj] = s[jj][r];
            if (bits) bits[(size_t)b * n + j0 + jj] = (int)((sb[jj][rw % VEC] >> (rw / VEC)) & 1ull);
        }
    }
}

// Vector versions of the two layout changes: every thread issues ALL its loads before the first use, so that a block
// keeps 32 KB in flight -- the scalar kernels above have one or two 256-byte requests in flight per wave and run at half
// the sweep rate.  The workspace side (tile rows) always moves 16 bytes per lane; the CALLER side moves VB bytes per lane,
// the widest of 16 / 8 / 4 that the row length and the buffer alignment admit (n = 1998 floats: rows are 8-byte aligned,
// VB = 8) -- a wave still touches 256 contiguous bytes of a codeword's row whatever VB is.
template <typename T, int VEC, int VB>
__global__ __launch_bounds__(kBlock) void transpose_in_v(const T *__restrict__ llr, T *__restrict__ llrT,
                                                         long long batch, int n, int var_chunks)
{
    constexpr int W = kWave * VEC;
    constexpr int JT = transpose_vars<T>();
    constexpr int VT = VB / (int)sizeof(T);            // elements per caller-side vector
    constexpr int CV = JT / VT;                        // vectors per codeword run
    constexpr int PER = kWave * CV / kBlock;           // caller-side vectors per thread
    constexpr int WT = 16 / (int)sizeof(T);            // elements per workspace-side (16-byte) vector
    constexpr int WV = kWave / WT;                     // vectors per variable run
    using V = typename VecB<T, VB>::type;
    using V16 = typename VecB<T, 16>::type;
    __shared__ T s[JT][kWave + 1];
    const int chunk = blockIdx.x % var_chunks, sub = (blockIdx.x / var_chunks) % VEC, tile = blockIdx.x / (var_chunks * VEC);
    const int j0 = chunk * JT, w0 = sub * kWave;
    V v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = threadIdx.x + k * kBlock, r = idx / CV, c = idx % CV;
        const long long b = (long long)tile * W + w0 + r;
        v[k] = (T)1;                                    // padding codewords: benign positive LLR
        if (b < batch && j0 + c * VT < n) v[k] = __builtin_nontemporal_load(reinterpret_cast<const V *>(llr + (size_t)b * n + j0 + c * VT));
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = threadIdx.x + k * kBlock, r = idx / CV, c = idx % CV;
#pragma unroll
        for (int q = 0; q < VT; ++q) s[c * VT + q][r] = vec_get<V, T>(v[k], q);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < JT * WV / kBlock; ++k) {
        const int idx = threadIdx.x + k * kBlock, jj = idx / WV, wv = idx % WV;
        V16 o;
#pragma unroll
        for (int q = 0; q < WT; ++q) o[q] = s[jj][wv * WT + q];
        if (j0 + jj < n) *reinterpret_cast<V16 *>(llrT + ((size_t)tile * n + j0 + jj) * W + w0 + wv * WT) = o;
    }
}

template <typename T, int VEC, int VB>
__global__ __launch_bounds__(kBlock) void transpose_out_v(const T *__restrict__ postT,
                                                          const uint64_t *__restrict__ bitsT,
                                                          T *__restrict__ posterior, int *__restrict__ bits,
                                                          long long batch, int n, int var_chunks)
{
    constexpr int W = kWave * VEC;
    constexpr int JT = transpose_vars<T>();
    constexpr int VT = VB / (int)sizeof(T);            // caller side
    constexpr int CV = JT / VT;
    constexpr int WT = 16 / (int)sizeof(T);            // workspace side
    constexpr int WV = kWave / WT;
    constexpr int PER = JT * WV / kBlock;
    constexpr int PB = kWave + 4;                      // byte row of the expanded hard decisions (rows stay 4-byte aligned)
    using V = typename VecB<T, VB>::type;
    using V16 = typename VecB<T, 16>::type;
    using IV = typename VecB<int, VT * (int)sizeof(int)>::type;
    typedef uint8_t BV __attribute__((ext_vector_type(WT)));
    __shared__ T s[JT][kWave + 1];
    __shared__ __align__(4) uint8_t pb[JT][PB];        // hard decision of (variable, codeword of this run), one byte each
    const int chunk = blockIdx.x % var_chunks, sub = (blockIdx.x / var_chunks) % VEC, tile = blockIdx.x / (var_chunks * VEC);
    const int j0 = chunk * JT, w0 = sub * kWave;
    if (posterior) {
        V16 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x + k * kBlock, jj = idx / WV, wv = idx % WV;
            v[k] = (T)0;
            if (j0 + jj < n) v[k] = __builtin_nontemporal_load(reinterpret_cast<const V16 *>(postT + ((size_t)tile * n + j0 + jj) * W + w0 + wv * WT));
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x + k * kBlock, jj = idx / WV, wv = idx % WV;
#pragma unroll
            for (int q = 0; q < WT; ++q) s[jj][wv * WT + q] = v[k][q];
        }
    }
    if (bits) {
        // ballot words -> one byte per (variable, codeword): the thread that owns WT consecutive codewords of a variable
        // reads the (at most WT) words holding them -- the 64 / WT threads of a variable read the same words, one request
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int idx = threadIdx.x + k * kBlock, jj = idx / WV, wv = idx % WV;
            BV o;
#pragma unroll
            for (int q = 0; q < WT; ++q) {
                const int rw = w0 + wv * WT + q;        // codeword inside the tile: ballot lane rw / VEC, word rw % VEC
                const uint64_t word = (j0 + jj < n) ? bitsT[((size_t)tile * n + j0 + jj) * VEC + (rw % VEC)] : 0ull;
                const unsigned half = (rw / VEC) < 32 ? (unsigned)word : (unsigned)(word >> 32);
                o[q] = (uint8_t)((half >> ((rw / VEC) & 31)) & 1u);
            }
            *reinterpret_cast<BV *>(&pb[jj][wv * WT]) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWave * CV / kBlock; ++k) {
        const int idx = threadIdx.x + k * kBlock, r = idx / CV, c = idx % CV;
        const long long b = (long long)tile * W + w0 + r;
        if (b < batch && j0 + c * VT < n) {
            if (posterior) {
                V o;
#pragma unroll
                for (int q = 0; q < VT; ++q) vec_set<V, T>(o, q, s[c * VT + q][r]);
                __builtin_nontemporal_store(o, reinterpret_cast<V *>(posterior + (size_t)b * n + j0 + c * VT));
            }
            if (bits) {
                IV o;
#pragma unroll
                for (int q = 0; q < VT; ++q) vec_set<IV, int>(o, q, (int)pb[c * VT + q][r]);
                __builtin_nontemporal_store(o, reinterpret_cast<IV *>(bits + (size_t)b * n + j0 + c * VT));
            }
        }
    }
}

// iterations / success / packed hard decisions, one thread per (codeword, output byte)
template <int VEC>
__global__ void finalize_out(const uint64_t *__restrict__ done, const int *__restrict__ iters_ws,
                             const uint64_t *__restrict__ bitsT, int *__restrict__ iterations,
                             uint8_t *__restrict__ success, uint8_t *__restrict__ packed,
                             long long batch, int n)
{
    constexpr int W = kWave * VEC;
    const int nbytes = (n + 7) / 8;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < batch) {
        const int tile = (int)(gid / W), w = (int)(gid % W);
        if (iterations) iterations[gid] = iters_ws[gid];
        if (success) success[gid] = (uint8_t)((done[(size_t)tile * VEC + (w % VEC)] >> (w / VEC)) & 1ull);
    }
    if (packed) {
        const long long total = batch * nbytes;
        for (long long k = gid; k < total; k += (long long)gridDim.x * blockDim.x) {
            const long long b = k / nbytes;
            const int byte = (int)(k % nbytes);
            const int tile = (int)(b / W), w = (int)(b % W);
            unsigned v = 0;
            for (int q = 0; q < 8; ++q) {
                const int j = byte * 8 + q;
                if (j < n) v |= (unsigned)((bitsT[((size_t)tile * n + j) * VEC + (w % VEC)] >> (w / VEC)) & 1ull) << q;
            }
            packed[k] = (uint8_t)v;
        }
    }
}

}  // namespace ldpc
