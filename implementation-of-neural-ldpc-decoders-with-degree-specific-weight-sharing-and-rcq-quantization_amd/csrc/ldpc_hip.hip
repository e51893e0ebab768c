// ldpc_hip.hip -- C ABI (include/ldpc_hip.h) over the gfx950 kernels in ldpc_kernels.hip.
//
// Host side only: handle lifetime, table upload, workspace carving and the launch
// sequence of one decode.  Nothing here allocates or synchronises inside ldpc_decode
// (graph-capturable); all work goes to the caller's stream.
#include "ldpc_kernels.hip"
#include "ldpc_resident.hip"
#include "ldpc_train.hip"
#include "ldpc_layered.hip"

#include <algorithm>
#include <climits>
#include <mutex>
#include <set>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "../../include/ldpc_hip_debug.h"

using namespace ldpc;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(LDPC_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

template <typename X>
int upload(X **dst, const X *src, size_t count)
{
    *dst = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(X)));
    if (src) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(X), hipMemcpyHostToDevice));
    return LDPC_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct ldpc_graph {
    int device = 0;
    int n = 0, m = 0, E = 0, max_dc = 0, max_dv = 0;
    int *check_ptr = nullptr, *var_idx = nullptr, *var_ptr = nullptr, *csc_edge = nullptr;
    int *wide_checks = nullptr;    // checks of degree > kWideCheck (cn_sweep_wide splits each over a block)
    int n_wide = 0;
    std::vector<int> h_check_ptr, h_var_idx, h_var_ptr, h_csc, h_check_of_edge;   // host copies (resident-plan builder)
    GraphDev dev() const { return GraphDev{n, m, E, check_ptr, var_idx, var_ptr, csc_edge}; }
};

struct ldpc_decoder {
    const ldpc_graph *g = nullptr;
    int dtype = LDPC_F32, form = LDPC_C2V_NMS, T = 0;
    int n_beta = 0, n_alpha = 0, n_levels = 0, n_quant = 0, n_oms_alpha = 0;
    void *beta = nullptr, *alpha = nullptr, *oms_alpha = nullptr;   // device, dtype-typed [T][slots]
    int *beta_slot = nullptr, *alpha_slot = nullptr, *oms_alpha_slot = nullptr;
    float *thresholds = nullptr;   // device [Q][L]
    float *lut = nullptr;          // device [Q][2L] signed reconstruction values
    std::vector<int> q_of_iter;    // host
    int *q_of_iter_dev = nullptr;  // device copy (frozen codewords look their quantiser up)
    size_t elem() const { return dtype == LDPC_F64 ? 8 : 4; }
    // LDS-resident engine (ldpc_resident.hip): built at creation when the code qualifies
    int schedule = 0;              // LDPC_SCHED_*
    int mode = 0;                  // LDPC_MODE_*
    bool res_ok = false;
    int res_G = 0, res_NT = 0;
    bool unit_alpha = false, rcq_zero0 = false;   // table properties the resident kernel exploits
    bool beta_per_check = false;                  // every edge of a check uses the same beta slot
    size_t res_lds = 0;
    ResidentPlan res{};
    std::vector<void *> res_bufs;  // device allocations owned by the plan
    // inverse slot maps of the gradient path (reduce_table_grads): items of slot s = inv_items[inv_ptr[s] .. inv_ptr[s+1])
    int *beta_inv_ptr = nullptr, *beta_inv_items = nullptr;
    int *alpha_inv_ptr = nullptr, *alpha_inv_items = nullptr;
    int *oms_inv_ptr = nullptr, *oms_inv_items = nullptr;
    // fused RCQ iteration of the streaming engine (cn_gather): per-edge gather metadata, built at creation for fp32
    // flooding RCQ decoders on graphs with variable degree <= 8
    bool gat_ok = false;
    bool pair_ok = false;          // RCQ code-pair form (vn_sweep_q / cn_sweep_q): one beta per check, sorted thresholds
    bool key_float4 = false;       // ... with 4 levels and every threshold 1.. in [2^-50, 2^50]: the float form of the key (kKeyFloat4)
    int4 *gat_meta = nullptr;      // [E + 1]
    int *gat_nbr = nullptr;        // [sum dv(dv-1) + 8]
    // LDS-resident layered decode (ldpc_layered.hip): LDPC_SCHED_LAYERED_REF on codes whose posteriors fit LDS
    bool lay_ok = false;
    LayeredPlan lay{};
    uint32_t *lay_off = nullptr;
    size_t lay_lds = 0;
};

namespace {

bool use_resident(const ldpc_decoder *d) { return d->res_ok && (d->mode == LDPC_MODE_AUTO || d->mode == LDPC_MODE_RESIDENT); }
// layered schedule: the LDS-resident kernel unless a streaming mode is forced (then layered_rcq, posteriors in HBM)
bool use_layered_lds(const ldpc_decoder *d) { return d->lay_ok && (d->mode == LDPC_MODE_AUTO || d->mode == LDPC_MODE_RESIDENT); }
// streaming engine, RCQ: one fused kernel per iteration (cn_gather) unless the two-sweep form is forced
// streaming forms of an fp32 RCQ decoder, best first: code pair (4E + 4n bytes per iteration), fused gather, two sweeps
bool use_pair(const ldpc_decoder *d) { return d->pair_ok && d->mode != LDPC_MODE_SWEEPS && d->mode != LDPC_MODE_GATHER; }
bool use_gather(const ldpc_decoder *d) { return d->gat_ok && !use_pair(d) && d->mode != LDPC_MODE_SWEEPS && d->mode != LDPC_MODE_PAIR; }

// tile width: 64 lanes x VEC codewords.  fp32: VEC 4 (16 B per lane) for real batches,
// VEC 1 for latency-mode batches <= 64; fp64: VEC 2 / 1.
int pick_vec(const ldpc_decoder *d, int64_t batch)
{
#ifdef LDPC_RESIDENT_PROBES                          // tuning builds: LDPC_STREAM_VEC=1 forces 64-codeword tiles
    { const char *ev = getenv("LDPC_STREAM_VEC"); if (ev && atoi(ev) == 1) return 1; }
#endif
    if (batch <= 64) return 1;
    if (d->schedule != LDPC_SCHED_FLOODING) return 1;      // layered: one dependent chain per wave, as many waves as possible
    return d->dtype == LDPC_F64 ? 2 : 4;
}

// ldpc_decode_capped: the calling thread's decode runs at most this many iterations (INT_MAX outside such a call).  The decode
// entry points only enqueue work, so the cap lives exactly as long as the call that set it.
thread_local int tl_iter_cap = INT_MAX;
inline int capped_T(const ldpc_decoder *d) { return std::min(d->T, tl_iter_cap); }

struct Workspace {
    int vec = 0, tiles = 0;
    char *llrT = nullptr, *v2c = nullptr, *c2v = nullptr, *postT = nullptr;
    char *c2v_prev = nullptr;        // saving forward only: previous iteration's c2v slice (latched rows are carried over)
    uint64_t *bitsT = nullptr, *done = nullptr;
    int *iters = nullptr;
    unsigned long long *unsat = nullptr;     // [tiles][vec] partial syndrome words and
    int *ticket = nullptr;                   // [tiles] block tickets of syndrome_latch_chunks (both zero between launches)
    size_t total = 0;
};

Workspace carve(const ldpc_decoder *d, int64_t batch, void *base)
{
    Workspace w;
    w.vec = pick_vec(d, batch);
    const int W = 64 * w.vec;
    w.tiles = (int)((batch + W - 1) / W);
    if (w.tiles < 1) w.tiles = 1;
    const size_t es = d->elem();
    const size_t n = d->g->n, E = d->g->E, tw = (size_t)w.tiles * W;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t o_llr = take(tw * n * es);
    const size_t o_v2c = take(tw * std::max<size_t>(E, 1) * ((use_gather(d) || use_pair(d)) ? 1 : es));   // gather / code-pair form: the second code buffer
    const size_t o_c2v = take(tw * std::max<size_t>(E, 1) * (d->form == LDPC_C2V_RCQ ? 1 : es));
    const size_t o_post = take(tw * n * es);
    const size_t o_bits = take((size_t)w.tiles * n * w.vec * sizeof(uint64_t));
    const size_t o_done = take((size_t)w.tiles * w.vec * sizeof(uint64_t));
    const size_t o_it = take(tw * sizeof(int));
    const size_t o_unsat = take((size_t)w.tiles * w.vec * sizeof(uint64_t));
    const size_t o_ticket = take((size_t)w.tiles * sizeof(int));
    w.total = off;
    if (base) {
        char *b = (char *)base;
        w.llrT = b + o_llr; w.v2c = b + o_v2c; w.c2v = b + o_c2v; w.postT = b + o_post;
        w.bitsT = (uint64_t *)(b + o_bits); w.done = (uint64_t *)(b + o_done); w.iters = (int *)(b + o_it);
        w.unsat = (unsigned long long *)(b + o_unsat); w.ticket = (int *)(b + o_ticket);
    }
    return w;
}

// ---- launch helpers, one per kernel family ------------------------------------------------
// syndrome + latch: one block per tile when there are many tiles, else `chunks` blocks per tile (syndrome_latch_chunks)
template <int VEC>
void launch_syndrome(const GraphDev &g, const Workspace &w, int it_plus_1, int latch, hipStream_t s)
{
    const int max_chunks = (g.m + kBlock - 1) / kBlock;
    // 64 tiles or more keep the one-block form (measured equal there: 173 vs 182 us at 128 tiles of the (16200,7200) code)
    const int chunks = w.tiles >= 64 ? 1 : std::min(max_chunks, std::max(1, 256 / std::max(w.tiles, 1)));
    if (chunks <= 1)
        hipLaunchKernelGGL((syndrome_latch<VEC>), dim3(w.tiles), dim3(kBlock), 0, s, g, w.bitsT, w.done, w.iters, it_plus_1, latch);
    else
        hipLaunchKernelGGL((syndrome_latch_chunks<VEC>), dim3((unsigned)w.tiles * chunks), dim3(kBlock), 0, s, g, w.bitsT, w.done,
                           w.iters, it_plus_1, latch, chunks, w.unsat, w.ticket);
}

template <typename T, int VEC>
int launch_cn(const ldpc_decoder *d, const Workspace &w, int it, bool use_done, hipStream_t s)
{
    const GraphDev g = d->g->dev();
    // small check degrees: two consecutive checks per wave (cn_sweep CPW), for the two forms the benchmarks use
    const int cpw = (g.E < 8 * (long long)g.m && (d->form == LDPC_C2V_NMS || (d->form == LDPC_C2V_RCQ && d->n_levels == 4))) ? 2 : 1;
    const int per_block = kWavesPerBlock * cpw;
    const int cb = (g.m + per_block - 1) / per_block;
    if (cb == 0 || g.E == 0) return LDPC_OK;
    const dim3 grid((unsigned)((size_t)w.tiles * cb)), block(kBlock);
    const bool first = it == 0;
    const int n_wide = d->g->n_wide;                 // > 0: the sweep below skips them, cn_sweep_wide follows
    const T *src = first ? (const T *)w.llrT : (const T *)w.v2c;
    const T *beta_row = (const T *)d->beta + (size_t)it * d->n_beta;
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        // fp32 normalised min-sum on check degrees <= 16: the register-held straight-line form (cn_sweep_f4)
        if (d->form == LDPC_C2V_NMS && d->g->max_dc <= 16 && !w.c2v_prev) {
            const int cb4 = (g.m + kWavesPerBlock - 1) / kWavesPerBlock;
            const dim3 grid4((unsigned)((size_t)w.tiles * cb4));
            const uint64_t *done4 = use_done ? w.done : nullptr;
#define LDPC_CF(FIRST_, BPC_, ES_)                                                                                    \
    hipLaunchKernelGGL((cn_sweep_f4<FIRST_, BPC_, ES_>), grid4, block, 0, s, g, (const float *)src, (float *)w.c2v,     \
                       (const float *)beta_row, (const int *)d->beta_slot, done4, cb4)
            switch ((first ? 4 : 0) + (d->beta_per_check ? 2 : 0) + (done4 ? 1 : 0)) {
            case 0: LDPC_CF(false, false, false); break;
            case 1: LDPC_CF(false, false, true); break;
            case 2: LDPC_CF(false, true, false); break;
            case 3: LDPC_CF(false, true, true); break;
            case 4: LDPC_CF(true, false, false); break;
            case 5: LDPC_CF(true, false, true); break;
            case 6: LDPC_CF(true, true, false); break;
            default: LDPC_CF(true, true, true); break;
            }
#undef LDPC_CF
            HIP_TRY(hipGetLastError());
            return LDPC_OK;
        }
    }
    const T *oa_row = d->oms_alpha ? (const T *)d->oms_alpha + (size_t)it * d->n_oms_alpha : nullptr;
    const float *thr = d->form == LDPC_C2V_RCQ ? d->thresholds + (size_t)d->q_of_iter[it] * d->n_levels : nullptr;
    const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_CN_X(FORM, FIRST, NL_, BPC_, CPW_)                                                                       \
    hipLaunchKernelGGL((cn_sweep<T, VEC, FORM, FIRST, NL_, BPC_, CPW_>), grid, block, 0, s, g, src, (void *)w.c2v,    \
                       beta_row, d->beta_slot, thr, d->n_levels, oa_row, d->oms_alpha_slot, done, cb,             \
                       (const void *)w.c2v_prev, n_wide)
#define LDPC_CN(FORM, FIRST) LDPC_CN_X(FORM, FIRST, 0, false, 1)
    if (d->form == LDPC_C2V_NMS) {
        if (cpw == 2) { if (first) LDPC_CN_X(FORM_NMS, true, 0, false, 2); else LDPC_CN_X(FORM_NMS, false, 0, false, 2); }
        else { if (first) LDPC_CN(FORM_NMS, true); else LDPC_CN(FORM_NMS, false); }
    } else if (d->form == LDPC_C2V_OMS) {
        if (first) LDPC_CN(FORM_OMS, true); else LDPC_CN(FORM_OMS, false);
    } else {
        if constexpr (sizeof(T) == 4) {
            if (d->n_levels == 4) {                      // bc = 3: compare chain with a compile-time length
                const int variant = (d->beta_per_check ? 2 : 0) + (cpw == 2 ? 1 : 0);
                switch (variant * 2 + (first ? 1 : 0)) {
                case 0: LDPC_CN_X(FORM_RCQ, false, 4, false, 1); break;
                case 1: LDPC_CN_X(FORM_RCQ, true, 4, false, 1); break;
                case 2: LDPC_CN_X(FORM_RCQ, false, 4, false, 2); break;
                case 3: LDPC_CN_X(FORM_RCQ, true, 4, false, 2); break;
                case 4: LDPC_CN_X(FORM_RCQ, false, 4, true, 1); break;
                case 5: LDPC_CN_X(FORM_RCQ, true, 4, true, 1); break;
                case 6: LDPC_CN_X(FORM_RCQ, false, 4, true, 2); break;
                default: LDPC_CN_X(FORM_RCQ, true, 4, true, 2); break;
                }
            } else {
                if (first) LDPC_CN(FORM_RCQ, true); else LDPC_CN(FORM_RCQ, false);
            }
        } else {
            return fail(LDPC_ERR_UNSUPPORTED, "RCQ messages are fp32 only");
        }
    }
#undef LDPC_CN_X
#undef LDPC_CN
    HIP_TRY(hipGetLastError());
    if (n_wide) {
        const dim3 wgrid((unsigned)((size_t)w.tiles * n_wide));
#define LDPC_CW(FORM, FIRST, NL_)                                                                                     \
    hipLaunchKernelGGL((cn_sweep_wide<T, VEC, FORM, FIRST, NL_>), wgrid, block, 0, s, g, (const int *)d->g->wide_checks, \
                       n_wide, src, (void *)w.c2v, beta_row, d->beta_slot, thr, d->n_levels, oa_row,                  \
                       d->oms_alpha_slot, done, (const void *)w.c2v_prev)
        if (d->form == LDPC_C2V_NMS) { if (first) LDPC_CW(FORM_NMS, true, 0); else LDPC_CW(FORM_NMS, false, 0); }
        else if (d->form == LDPC_C2V_OMS) { if (first) LDPC_CW(FORM_OMS, true, 0); else LDPC_CW(FORM_OMS, false, 0); }
        else {
            if constexpr (sizeof(T) == 4) {
                if (d->n_levels == 4) { if (first) LDPC_CW(FORM_RCQ, true, 4); else LDPC_CW(FORM_RCQ, false, 4); }
                else { if (first) LDPC_CW(FORM_RCQ, true, 0); else LDPC_CW(FORM_RCQ, false, 0); }
            }
        }
#undef LDPC_CW
        HIP_TRY(hipGetLastError());
    }
    return LDPC_OK;
}

// where the last variable pass may put the caller's rows itself (vn_last_rows) instead of a tile-major posterior array
struct RowsOut {
    float *posterior = nullptr;
    int32_t *bits = nullptr;
    long long batch = 0;
    int qv = 4;                    // floats per store: 4 / 2 / 1
};

int allow_full_lds(const void *kfn, int device);

template <typename T, int VEC>
int launch_vn(const ldpc_decoder *d, const Workspace &w, int it, bool last, bool use_done, hipStream_t s,
              bool store_posterior = true, const RowsOut *rows = nullptr)
{
    const GraphDev g = d->g->dev();
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        if (rows && last) {
            const int row = it < d->T ? it : 0;
            const bool codes = d->form == LDPC_C2V_RCQ;
            const int lut_stride = 2 * d->n_levels;
            const int lut_total = codes ? d->n_quant * lut_stride : 0;
            const int lut_cur = codes ? d->q_of_iter[row] * lut_stride : 0;
            const size_t shmem = vn_rows_stage_bytes() + (size_t)lut_total * sizeof(float);
            const int vb = (g.n + kRowsVars - 1) / kRowsVars;
            const dim3 grid((unsigned)((size_t)w.tiles * rows_grid_chunks(vb, LDPC_ROWS_XCD != 0))), block(kRowsThreads);
            const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_VR(CODES)                                                                                          \
    do {                                                                                                        \
        auto kfn = rows->qv == 4 ? vn_last_rows<CODES, 4> : rows->qv == 2 ? vn_last_rows<CODES, 2> : vn_last_rows<CODES, 1>; \
        if (int rc_ = allow_full_lds((const void *)kfn, d->g->device)) return rc_;                              \
        hipLaunchKernelGGL(kfn, grid, block, shmem, s, g, (const void *)w.c2v, (const float *)w.llrT,           \
                           (const float *)d->lut, lut_total, lut_cur, lut_stride, (const int *)d->q_of_iter_dev, \
                           (const int *)w.iters, w.bitsT, done, rows->posterior, rows->bits, rows->batch, vb);  \
    } while (0)
            if (codes) LDPC_VR(true); else LDPC_VR(false);
#undef LDPC_VR
            HIP_TRY(hipGetLastError());
            return LDPC_OK;
        }
    }
    const int vb = (g.n + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid((unsigned)((size_t)w.tiles * vb)), block(kBlock);
    const int row = it < d->T ? it : 0;     // T == 0: posterior-only pass, alpha unused
    const T *alpha_row = (const T *)d->alpha + (size_t)row * d->n_alpha;
    const bool codes = d->form == LDPC_C2V_RCQ;
    const int lut_stride = 2 * d->n_levels;
    const int lut_total = codes ? d->n_quant * lut_stride : 0;
    const int lut_cur = codes ? d->q_of_iter[row] * lut_stride : 0;
    const size_t shmem = (size_t)lut_total * sizeof(float);
    const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_VN(CODES, LAST)                                                                           \
    hipLaunchKernelGGL((vn_sweep<T, VEC, CODES, LAST>), grid, block, shmem, s, g, (const void *)w.c2v,  \
                       (const T *)w.llrT, (T *)w.v2c, alpha_row, d->alpha_slot, (const float *)d->lut,  \
                       lut_total, lut_cur, lut_stride, (const int *)d->q_of_iter_dev,                   \
                       (const int *)w.iters, w.bitsT, store_posterior ? (T *)w.postT : (T *)nullptr, done, vb)
    if (codes) {
        if constexpr (sizeof(T) == 4) {
            if (last) LDPC_VN(true, true); else LDPC_VN(true, false);
        } else {
            return fail(LDPC_ERR_UNSUPPORTED, "RCQ messages are fp32 only");
        }
    } else {
        if (last) LDPC_VN(false, true); else LDPC_VN(false, false);
    }
#undef LDPC_VN
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

// code-pair form, iterations >= 1: V2C codes (w.v2c) -> C2V codes (w.c2v)
template <int VEC>
int launch_cn_q(const ldpc_decoder *d, const Workspace &w, int it, bool use_done, hipStream_t s)
{
    const GraphDev g = d->g->dev();
    const int dmax = d->g->max_dc;
    const int cpw = (dmax <= 8 && g.E < 8 * (long long)g.m) ? 2 : 1;
    const int per_block = kWavesPerBlock * cpw;
    const int cb = (g.m + per_block - 1) / per_block;
    if (cb == 0 || g.E == 0) return LDPC_OK;
    const dim3 grid((unsigned)((size_t)w.tiles * cb)), block(kBlock);
    const float *beta_row = (const float *)d->beta + (size_t)it * d->n_beta;
    const float *thr = d->thresholds + (size_t)d->q_of_iter[it] * d->n_levels;
    const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_CQ(CPW_, DCMAX_)                                                                                        \
    hipLaunchKernelGGL((cn_sweep_q<VEC, CPW_, DCMAX_>), grid, block, 0, s, g, (const uint8_t *)w.v2c, (uint8_t *)w.c2v, \
                       beta_row, (const int *)d->beta_slot, thr, d->n_levels, done, cb)
#define LDPC_CQ4(CPW_, DCMAX_)                                                                                       \
    do {                                                                                                             \
        if (done)                                                                                                    \
            hipLaunchKernelGGL((cn_sweep_q4<CPW_, DCMAX_, true>), grid, block, 0, s, g, (const uint8_t *)w.v2c,       \
                               (uint8_t *)w.c2v, beta_row, (const int *)d->beta_slot, d->n_levels, done, cb);        \
        else                                                                                                         \
            hipLaunchKernelGGL((cn_sweep_q4<CPW_, DCMAX_, false>), grid, block, 0, s, g, (const uint8_t *)w.v2c,      \
                               (uint8_t *)w.c2v, beta_row, (const int *)d->beta_slot, d->n_levels, done, cb);        \
    } while (0)
    if constexpr (VEC == 4) {
        if (dmax <= 8) { if (cpw == 2) LDPC_CQ4(2, 8); else LDPC_CQ4(1, 8); }
        else if (dmax <= 16) LDPC_CQ4(1, 16);
        else if (dmax <= 32) LDPC_CQ4(1, 32);
        else LDPC_CQ4(1, 0);
    } else {
        if (dmax <= 8) { if (cpw == 2) LDPC_CQ(2, 8); else LDPC_CQ(1, 8); }
        else if (dmax <= 16) LDPC_CQ(1, 16);
        else if (dmax <= 32) LDPC_CQ(1, 32);
        else LDPC_CQ(1, 0);
    }
#undef LDPC_CQ4
#undef LDPC_CQ
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

// 256-codeword tiles of a decoder whose variable sweep can run vn_sweep_q4 (degrees <= 8, at most 8 levels)
template <int VEC>
bool pair_q4(const ldpc_decoder *d) { return VEC == 4 && d->g->max_dv <= 8 && d->n_levels <= 8; }

// code-pair form, iterations < T-1: C2V codes of iteration `it` (w.c2v) -> V2C codes for iteration it + 1 (w.v2c).
// it == -1 (pair_q4 decoders only): the pass before iteration 0, LLRs -> V2C codes for iteration 0.
template <int VEC>
int launch_vn_q(const ldpc_decoder *d, const Workspace &w, int it, bool use_done, hipStream_t s)
{
    const bool init = it < 0;
    if (init) it = 0;
    const GraphDev g = d->g->dev();
    const int vb = (g.n + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 grid((unsigned)((size_t)w.tiles * vb)), block(kBlock);
    const float *alpha_row = (const float *)d->alpha + (size_t)it * d->n_alpha;
    const int nxt = init ? 0 : it + 1;
    const float *beta_next = (const float *)d->beta + (size_t)nxt * d->n_beta;
    const float *thr_next = d->thresholds + (size_t)d->q_of_iter[nxt] * d->n_levels;
    const int lut_entries = 2 * d->n_levels;
    const float *lut_cur = d->lut + (size_t)d->q_of_iter[it] * lut_entries;
    const size_t shmem = (size_t)lut_entries * sizeof(float);
    const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_VQ(NL_)                                                                                                  \
    hipLaunchKernelGGL((vn_sweep_q<VEC, NL_>), grid, block, shmem, s, g, (const uint8_t *)w.c2v, (const float *)w.llrT, \
                       (uint8_t *)w.v2c, alpha_row, (const int *)d->alpha_slot, lut_cur, lut_entries, beta_next,        \
                       (const int *)d->beta_slot, thr_next, d->n_levels, w.bitsT, done, vb)
    const int vb4 = (g.n + kWavesPerBlock * LDPC_VNQ_VPW - 1) / (kWavesPerBlock * LDPC_VNQ_VPW);
    const dim3 grid4((unsigned)((size_t)w.tiles * vb4));
#define LDPC_VQ4(NL_, ES_)                                                                                            \
    hipLaunchKernelGGL((vn_sweep_q4<NL_, ES_, LDPC_VNQ_VPW>), grid4, block, shmem, s, g, (const uint8_t *)w.c2v,         \
                       (const float *)w.llrT, (uint8_t *)w.v2c, alpha_row, (const int *)d->alpha_slot, lut_cur,         \
                       lut_entries, beta_next, (const int *)d->beta_slot, thr_next, d->n_levels, w.bitsT, done, vb4)
    if (init) {
        if (!pair_q4<VEC>(d)) return fail(LDPC_ERR_ARG, "internal: code-pair initial pass on a decoder without vn_sweep_q4");
#define LDPC_VQI(NL_)                                                                                                 \
    hipLaunchKernelGGL((vn_sweep_q4<NL_, false, LDPC_VNQ_VPW, true>), grid4, block, 0, s, g, (const uint8_t *)w.c2v,     \
                       (const float *)w.llrT, (uint8_t *)w.v2c, alpha_row, (const int *)d->alpha_slot, lut_cur,         \
                       lut_entries, beta_next, (const int *)d->beta_slot, thr_next, d->n_levels, w.bitsT, done, vb4)
        if (d->key_float4) LDPC_VQI(kKeyFloat4); else if (d->n_levels == 4) LDPC_VQI(4); else LDPC_VQI(0);
#undef LDPC_VQI
    } else if (pair_q4<VEC>(d)) {
        if (d->key_float4) { if (done) LDPC_VQ4(kKeyFloat4, true); else LDPC_VQ4(kKeyFloat4, false); }
        else if (d->n_levels == 4) { if (done) LDPC_VQ4(4, true); else LDPC_VQ4(4, false); }
        else { if (done) LDPC_VQ4(0, true); else LDPC_VQ4(0, false); }
    }
    else if (d->n_levels == 4) LDPC_VQ(4); else LDPC_VQ(0);
#undef LDPC_VQ4
#undef LDPC_VQ
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

// fused RCQ iteration `it` >= 1: codes of iteration it-1 (`cin`) -> codes of iteration it (`cout`)
template <int VEC>
int launch_gather(const ldpc_decoder *d, const Workspace &w, int it, bool use_done, const char *cin, char *cout,
                  hipStream_t s)
{
    const GraphDev g = d->g->dev();
    const int cpw = (g.E < 8 * (long long)g.m) ? 2 : 1;
    const int per_block = kWavesPerBlock * cpw;
    const int cb = (g.m + per_block - 1) / per_block;
    if (cb == 0 || g.E == 0) return LDPC_OK;
    // XCD-affine tile mapping when the rows one tile touches (LLRs + both code buffers) fit an XCD's 4 MiB L2: the re-reads
    // then hit there instead of going through the fabric (measured, (1998,1512) RCQ: 6.28 -> 5.85 ms per decode; on the
    // (16200,7200) code, 29 MB per tile, it costs 3 %: gpurun_out/xcd1)
    constexpr int W = 64 * VEC;
    int xcd_tiles = (w.tiles >= 16 && ((size_t)4 * g.n + 2 * (size_t)g.E) * W <= (4u << 20)) ? w.tiles : 0;
#ifdef LDPC_RESIDENT_PROBES                          // tuning builds: LDPC_GATHER_XCD=0/1 overrides
    { const char *ex = getenv("LDPC_GATHER_XCD"); if (ex) xcd_tiles = atoi(ex) > 0 ? w.tiles : 0; }
#endif
    const size_t tiles_padded = xcd_tiles ? (size_t)((w.tiles + 7) / 8) * 8 : (size_t)w.tiles;
    const dim3 grid((unsigned)(tiles_padded * cb)), block(kBlock);
    const float *beta_row = (const float *)d->beta + (size_t)it * d->n_beta;
    const float *alpha_prev = (const float *)d->alpha + (size_t)(it - 1) * d->n_alpha;
    const float *thr = d->thresholds + (size_t)d->q_of_iter[it] * d->n_levels;
    const int lut_entries = 2 * d->n_levels;
    const float *lut_prev = d->lut + (size_t)d->q_of_iter[it - 1] * lut_entries;     // the quantiser that produced `cin`
    const size_t shmem = (size_t)lut_entries * sizeof(float);
    const uint64_t *done = use_done ? w.done : nullptr;
#define LDPC_GA(NL_, BPC_, CPW_)                                                                                     \
    hipLaunchKernelGGL((cn_gather<VEC, NL_, BPC_, CPW_, LDPC_GATHER_GRP>), grid, block, shmem, s, g,                  \
                       (const int4 *)d->gat_meta, (const int *)d->gat_nbr, (const float *)w.llrT, (const uint8_t *)cin, \
                       (uint8_t *)cout, beta_row, (const int *)d->beta_slot, alpha_prev, thr, d->n_levels, lut_prev,   \
                       lut_entries, done, cb, xcd_tiles)
    const int variant = (d->n_levels == 4 ? 4 : 0) + (d->beta_per_check ? 2 : 0) + (cpw == 2 ? 1 : 0);
    switch (variant) {
    case 0: LDPC_GA(0, false, 1); break;
    case 1: LDPC_GA(0, false, 2); break;
    case 2: LDPC_GA(0, true, 1); break;
    case 3: LDPC_GA(0, true, 2); break;
    case 4: LDPC_GA(4, false, 1); break;
    case 5: LDPC_GA(4, false, 2); break;
    case 6: LDPC_GA(4, true, 1); break;
    default: LDPC_GA(4, true, 2); break;
    }
#undef LDPC_GA
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

// Saved forward state of the training path: c2v rows of iterations 0..T-1, then the v2c input rows of
// iterations 1..T-1 (iteration 0 reads the LLRs), each slice [tile][E][W] fp32.
struct SavedLayout {
    size_t slice = 0;
    int T = 0;
    size_t c2v_off(int it) const { return (size_t)it * slice; }
    size_t v2c_off(int it) const { return ((size_t)T + (size_t)(it - 1)) * slice; }      // it >= 1
    size_t total() const { return T > 0 ? (size_t)(2 * T - 1) * slice : 0; }
};
SavedLayout saved_layout(const ldpc_decoder *d, int tiles, int W)
{
    SavedLayout sl;
    sl.T = d->T;
    sl.slice = align_up((size_t)tiles * W * std::max(d->g->E, 1) * sizeof(float));
    return sl;
}

template <typename T, int VEC>
int decode_impl(const ldpc_decoder *d, const void *llr, int64_t batch, bool early_stop, int32_t *bits,
                void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed,
                const Workspace &w, hipStream_t s, char *saved = nullptr)
{
    constexpr int W = 64 * VEC;
    constexpr int JT = transpose_vars<T>();
    const GraphDev g = d->g->dev();
    const int T_it = capped_T(d);
    const bool need_post = bits || posterior;          // packed decisions / iterations only: the last pass stores no posterior rows
    const int vc = (g.n + JT - 1) / JT;
    const dim3 tgrid((unsigned)((size_t)w.tiles * VEC * vc));       // transposes: one block per (tile, 64-codeword run, chunk)

    // vector forms of the layout changes: the caller side moves the widest of 16 / 8 / 4 bytes per lane that the row length
    // and the buffers' alignment admit (0: not even element-aligned -- cannot happen for tensors, kept as the scalar kernels)
    auto vec_bytes = [&](const void *a, const void *b) {
        for (int vb = 16; vb >= (int)sizeof(T); vb >>= 1)
            if (((size_t)g.n * sizeof(T)) % vb == 0 && ((uintptr_t)a & (vb - 1)) == 0 && ((uintptr_t)b & (vb - 1)) == 0) return vb;
        return 0;
    };
    auto layout_in = [&]() {
        switch (vec_bytes(llr, nullptr)) {
        case 16: hipLaunchKernelGGL((transpose_in_v<T, VEC, 16>), tgrid, dim3(kBlock), 0, s, (const T *)llr, (T *)w.llrT, (long long)batch, g.n, vc); break;
        case 8: hipLaunchKernelGGL((transpose_in_v<T, VEC, 8>), tgrid, dim3(kBlock), 0, s, (const T *)llr, (T *)w.llrT, (long long)batch, g.n, vc); break;
        case 4:
            if constexpr (sizeof(T) == 4) {
                hipLaunchKernelGGL((transpose_in_v<T, VEC, 4>), tgrid, dim3(kBlock), 0, s, (const T *)llr, (T *)w.llrT, (long long)batch, g.n, vc);
                break;
            }
            [[fallthrough]];
        default: hipLaunchKernelGGL((transpose_in<T, VEC>), tgrid, dim3(kBlock), 0, s, (const T *)llr, (T *)w.llrT, (long long)batch, g.n, vc);
        }
    };
    auto layout_out = [&](const T *srcT) {
        switch (vec_bytes(posterior, bits)) {
        case 16: hipLaunchKernelGGL((transpose_out_v<T, VEC, 16>), tgrid, dim3(kBlock), 0, s, srcT, w.bitsT, (T *)posterior, bits, (long long)batch, g.n, vc); break;
        case 8: hipLaunchKernelGGL((transpose_out_v<T, VEC, 8>), tgrid, dim3(kBlock), 0, s, srcT, w.bitsT, (T *)posterior, bits, (long long)batch, g.n, vc); break;
        case 4:
            if constexpr (sizeof(T) == 4) {
                hipLaunchKernelGGL((transpose_out_v<T, VEC, 4>), tgrid, dim3(kBlock), 0, s, srcT, w.bitsT, (T *)posterior, bits, (long long)batch, g.n, vc);
                break;
            }
            [[fallthrough]];
        default: hipLaunchKernelGGL((transpose_out<T, VEC>), tgrid, dim3(kBlock), 0, s, srcT, w.bitsT, (T *)posterior, bits, (long long)batch, g.n, vc);
        }
    };
    // code-pair form on 256-codeword tiles: the layout change also codes the LLRs for iteration 0 (transpose_in_q4)
    bool fused_init = false;
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        if (use_pair(d) && !saved && pair_q4<VEC>(d) && T_it > 0 && d->schedule == LDPC_SCHED_FLOODING && vec_bytes(llr, nullptr) == 16) {
            const int vb = (g.n + kRowsVars - 1) / kRowsVars;
            const float *beta0 = (const float *)d->beta;
            const float *thr0 = d->thresholds + (size_t)d->q_of_iter[0] * d->n_levels;
#define LDPC_TIQ(NL_)                                                                                                \
    do {                                                                                                             \
        auto kfn = transpose_in_q4<NL_>;                                                                             \
        if (int rc_ = allow_full_lds((const void *)kfn, d->g->device)) return rc_;                                   \
        hipLaunchKernelGGL(kfn, dim3((unsigned)((size_t)w.tiles * rows_grid_chunks(vb, false))), dim3(kRowsThreads), vn_rows_stage_bytes(), s, g, \
                           (const float *)llr, (float *)w.llrT, (uint8_t *)w.v2c, beta0, (const int *)d->beta_slot, \
                           thr0, d->n_levels, (long long)batch, vb);                                                \
    } while (0)
            if (d->key_float4) LDPC_TIQ(kKeyFloat4); else if (d->n_levels == 4) LDPC_TIQ(4); else LDPC_TIQ(0);
#undef LDPC_TIQ
            fused_init = true;
        }
    }
    if (!fused_init) layout_in();
    {
        const long long cnt = (long long)w.tiles * W;
        hipLaunchKernelGGL((init_state<VEC>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, w.done,
                           w.iters, (long long)batch, w.tiles, T_it, w.unsat, w.ticket);
    }
    HIP_TRY(hipGetLastError());

    if (d->schedule != LDPC_SCHED_FLOODING) {
        if constexpr (sizeof(T) == 4) {
            // posteriors start as the LLRs and are updated in place, check after check, by one wave per tile;
            // LDPC_SCHED_LAYERED keeps every edge's message code in the c2v buffer (subtracted before the check's update)
            if (d->schedule == LDPC_SCHED_LAYERED)
                hipLaunchKernelGGL((layered_rcq<VEC, true>), dim3(w.tiles), dim3(kWave), 0, s, g, (float *)w.llrT, d->thresholds,
                                   d->n_levels, (const int *)d->q_of_iter_dev, T_it, early_stop ? 1 : 0, w.bitsT, w.done, w.iters,
                                   d->g->max_dc, (uint8_t *)w.c2v);
            else
                hipLaunchKernelGGL((layered_rcq<VEC, false>), dim3(w.tiles), dim3(kWave), 0, s, g, (float *)w.llrT, d->thresholds,
                                   d->n_levels, (const int *)d->q_of_iter_dev, T_it, early_stop ? 1 : 0, w.bitsT, w.done, w.iters,
                                   d->g->max_dc, (uint8_t *)nullptr);
            HIP_TRY(hipGetLastError());
            if (early_stop && T_it == 0) HIP_TRY(hipMemsetAsync(w.done, 0, (size_t)w.tiles * VEC * sizeof(uint64_t), s));
            if (bits || posterior) layout_out((const T *)w.llrT);
            if (iterations || success || packed) {
                long long threads = batch;
                if (packed) threads = std::max<long long>(threads, std::min<long long>(batch * ((g.n + 7) / 8), 1ll << 22));
                hipLaunchKernelGGL((finalize_out<VEC>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w.done,
                                   w.iters, w.bitsT, iterations, success, packed, (long long)batch, g.n);
            }
            HIP_TRY(hipGetLastError());
            return LDPC_OK;
        } else {
            return fail(LDPC_ERR_UNSUPPORTED, "layered schedule is fp32 only");
        }
    }
    // fp32, 256-codeword tiles, 16-byte-aligned caller rows: the last variable pass writes posterior / decisions into the
    // caller's rows itself (vn_last_rows) -- no tile-major posterior array, no transpose_out pass
    RowsOut rows_out;
    const RowsOut *rows = nullptr;
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        if (!saved && T_it > 0 && (bits || posterior) && vec_bytes(posterior, bits) >= 4) {
            rows_out.posterior = (float *)posterior; rows_out.bits = bits; rows_out.batch = (long long)batch;
            rows_out.qv = vec_bytes(posterior, bits) / 4;
            rows = &rows_out;
        }
    }
    if (T_it == 0) {
        // no iteration ran: c2v == 0, posterior = llr + 0  (loop skipped, ldpc_decoder.py:147-153)
        const size_t c2v_bytes = (size_t)w.tiles * W * std::max(g.E, 1) * (d->form == LDPC_C2V_RCQ ? 1 : sizeof(T));
        HIP_TRY(hipMemsetAsync(w.c2v, 0, c2v_bytes, s));
        int rc = launch_vn<T, VEC>(d, w, 0, /*last=*/true, /*use_done=*/false, s);
        if (rc) return rc;
    }
    if (use_pair(d) && !saved) {
        if constexpr (sizeof(T) == 4) {
            // RCQ, code-pair form: both directions are 1-byte codes (iteration 0 of decoders without vn_sweep_q4 is the plain
            // check sweep on the LLRs) (vn_sweep_q quantises with the next iteration's beta and thresholds, cn_sweep_q is integer-only);
            // the last variable pass is the ordinary posterior pass over the C2V codes.
            const bool q4 = pair_q4<VEC>(d);      // LLRs -> V2C codes first, then iteration 0 is a code sweep like the others
            if (q4 && T_it > 0 && !fused_init) {
                int rc = launch_vn_q<VEC>(d, w, -1, false, s);
                if (rc) return rc;
            }
            for (int it = 0; it < T_it; ++it) {
                int rc = (it == 0 && !q4) ? launch_cn<T, VEC>(d, w, 0, early_stop, s) : launch_cn_q<VEC>(d, w, it, early_stop, s);
                if (rc) return rc;
                rc = it == T_it - 1 ? launch_vn<T, VEC>(d, w, it, /*last=*/true, early_stop, s, /*store_posterior=*/need_post, rows)
                                    : launch_vn_q<VEC>(d, w, it, early_stop, s);
                if (rc) return rc;
                if (early_stop)
                    launch_syndrome<VEC>(g, w, it + 1, 1, s);
            }
        }
    } else if (use_gather(d) && !saved) {
        if constexpr (sizeof(T) == 4) {
            // RCQ, fused form: iteration 0 is the plain check sweep on the LLRs, every later iteration ONE cn_gather
            // launch (codes ping-pong between the two code buffers; iteration `it` writes buffer it & 1).  The hard
            // decisions the stop rule needs come from a posterior-only variable pass per iteration (early stop), or
            // once at the end (fixed T).
            char *buf[2] = {w.c2v, w.v2c};
            for (int it = 0; it < T_it; ++it) {
                int rc;
                if (it == 0) {
                    rc = launch_cn<T, VEC>(d, w, 0, early_stop, s);
                } else {
                    rc = launch_gather<VEC>(d, w, it, early_stop, buf[(it - 1) & 1], buf[it & 1], s);
                }
                if (rc) return rc;
                const bool last = it == T_it - 1;
                if (early_stop || last) {
                    Workspace wv = w;
                    wv.c2v = buf[it & 1];
                    rc = launch_vn<T, VEC>(d, wv, it, /*last=*/true, early_stop, s, /*store_posterior=*/last && need_post, last ? rows : nullptr);
                    if (rc) return rc;
                }
                if (early_stop)
                    launch_syndrome<VEC>(g, w, it + 1, 1, s);
            }
        }
    } else
    for (int it = 0; it < T_it; ++it) {
        // training forward: iteration `it` writes its c2v rows, and the v2c rows the next one reads, straight into
        // their slices of `saved` (SavedLayout).  A stopped codeword's rows stay latched because the check sweep
        // carries them over from the previous slice -- the final variable sweep forms its posterior from them.
        Workspace wc = w, wv = w;
        if (saved) {
            const SavedLayout sl = saved_layout(d, w.tiles, W);
            wc.c2v = wv.c2v = saved + sl.c2v_off(it);
            if (it > 0) {
                wc.v2c = saved + sl.v2c_off(it);
                if (early_stop) wc.c2v_prev = saved + sl.c2v_off(it - 1);
            }
            if (it + 1 < T_it) wv.v2c = saved + sl.v2c_off(it + 1);
        }
        int rc = launch_cn<T, VEC>(d, wc, it, early_stop, s);
        if (rc) return rc;
        rc = launch_vn<T, VEC>(d, wv, it, it == T_it - 1, early_stop, s, /*store_posterior=*/it == T_it - 1 ? need_post : true, rows);
        if (rc) return rc;
        if (early_stop) {
            launch_syndrome<VEC>(g, w, it + 1, 1, s);
        }
    }
    if (!early_stop) {
        launch_syndrome<VEC>(g, w, T_it, 0, s);
    } else if (T_it == 0) {
        // reference: the loop never ran, success False, iterations 0 -> clear the (padding) latch bits
        HIP_TRY(hipMemsetAsync(w.done, 0, (size_t)w.tiles * VEC * sizeof(uint64_t), s));
    }
    HIP_TRY(hipGetLastError());

    if ((bits || posterior) && !rows) layout_out((const T *)w.postT);
    if (iterations || success || packed) {
        long long threads = batch;
        if (packed) threads = std::max<long long>(threads, std::min<long long>(batch * ((g.n + 7) / 8), 1ll << 22));
        hipLaunchKernelGGL((finalize_out<VEC>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w.done,
                           w.iters, w.bitsT, iterations, success, packed, (long long)batch, g.n);
    }
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

template <typename T>
int decode_dispatch(const ldpc_decoder *d, const void *llr, int64_t batch, bool early_stop, int32_t *bits,
                    void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed,
                    const Workspace &w, hipStream_t s, char *saved = nullptr)
{
    switch (w.vec) {
    case 1: return decode_impl<T, 1>(d, llr, batch, early_stop, bits, posterior, iterations, success, packed, w, s, saved);
    case 2:
        if constexpr (sizeof(T) == 8)
            return decode_impl<T, 2>(d, llr, batch, early_stop, bits, posterior, iterations, success, packed, w, s, saved);
        break;
    case 4:
        if constexpr (sizeof(T) == 4)
            return decode_impl<T, 4>(d, llr, batch, early_stop, bits, posterior, iterations, success, packed, w, s, saved);
        break;
    }
    return fail(LDPC_ERR_ARG, "internal: bad tile width");
}


// alpha table identically 1.0f / every quantiser's tau_0 == 0: exact shortcuts in the resident kernel
void resident_table_flags(ldpc_decoder *d, const void *alpha_host, const float *thr_host)
{
    if (alpha_host) {
        const size_t cnt = (size_t)std::max(d->T, 0) * d->n_alpha;
        bool unit = true;
        if (d->dtype == LDPC_F32) {
            const float *a = (const float *)alpha_host;
            for (size_t k = 0; k < cnt; ++k) unit = unit && a[k] == 1.0f;
        } else {
            const double *a = (const double *)alpha_host;
            for (size_t k = 0; k < cnt; ++k) unit = unit && a[k] == 1.0;
        }
        d->unit_alpha = unit;
    }
    if (thr_host) {
        bool z = true;
        for (int q = 0; q < d->n_quant; ++q) z = z && thr_host[(size_t)q * d->n_levels] == 0.0f;
        d->rcq_zero0 = z;
    }
}

// ---- LDS-resident engine: plan (host) ---------------------------------------------------------
constexpr size_t kLdsBytes = 160 * 1024;          // LDS per CU; one workgroup may take all of it
constexpr int kResSubDegreeCap = 32;              // checks up to this degree stay whole in the resident engine ...
constexpr int kResSubDegree = 16;                 // ... wider ones are split into lane groups of sub-checks this long

template <typename X>
int plan_upload(ldpc_decoder *d, const X **dst, const std::vector<X> &src)
{
    X *p = nullptr;
    int rc = upload(&p, src.data(), src.size());
    if (rc) return rc;
    d->res_bufs.push_back((void *)p);
    *dst = p;
    return LDPC_OK;
}

int resident_alpha_floats(const ldpc_decoder *d)
{
    if (d->dtype != LDPC_F32) return 0;                 // the fp64 kernel reads alpha from global memory
    const long long cnt = (long long)d->T * d->n_alpha;
    return cnt <= kResAlphaMax ? (int)cnt : 0;
}

// G codewords per workgroup fit when the state is within LDS and slot byte offsets fit 16 bits
bool resident_fits(const ldpc_decoder *d, long long S, int G, int blocks, int m_par)
{
    if (S * G * 4 > 65535) return false;
    return blocks * res_lds_total((int)S, d->g->n, G, resident_alpha_floats(d), m_par) <= kLdsBytes;
}
bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// ---- LDS bank-conflict-aware lane assignment -------------------------------------------------
// The check phase touches consecutive slots (conflict-free by construction); the variable phase
// gathers/scatters the slots of a variable's edges, so which variables share a wave decides how
// many LDS passes those accesses take.  Any order of the variables INSIDE a degree class is valid,
// so a seeded hill climb swaps variables between lane groups whenever that lowers
//     sum over read groups  (32 lanes) of  max multiplicity of (slot mod RM)
//   + sum over write groups (WG lanes) of  max multiplicity of (slot mod WM)
// (RM/WG/WM follow the instruction's banking: ds_read_b64 64 banks, ds_write_b64 32 banks in
// 16-lane groups; MI355X_MICROARCH.md "LDS").  On the (1998,1512) code the gathers go from 3.3 to
// ~2.1 passes per instruction, the scatters from 2.9 to ~2.0.  Purely a performance choice.
struct LaneCost {
    const std::vector<std::vector<int>> &vs;   // slots of each variable, CSC order
    const std::vector<int> &order;
    int rg, rm, wg, wm;
    int group(int first, int count, int mod) const
    {
        const int last = std::min<int>(first + count, (int)order.size());
        int kmax = 0;
        for (int i = first; i < last; ++i) kmax = std::max<int>(kmax, (int)vs[order[i]].size());
        int cost = 0;
        unsigned char cnt[64];
        for (int k = 0; k < kmax; ++k) {
            std::memset(cnt, 0, sizeof(cnt));
            int mx = 0;
            for (int i = first; i < last; ++i) {
                const auto &v = vs[order[i]];
                if ((int)v.size() > k) mx = std::max<int>(mx, ++cnt[v[k] % mod]);
            }
            cost += mx;
        }
        return cost;
    }
    int around(int x, int y) const            // cost of every group containing position x or y
    {
        int c = group(x / rg * rg, rg, rm) + group(x / wg * wg, wg, wm);
        if (y / rg != x / rg) c += group(y / rg * rg, rg, rm);
        if (y / wg != x / wg) c += group(y / wg * wg, wg, wm);
        return c;
    }
};

void optimise_lane_order(std::vector<int> &order, const std::vector<std::vector<int>> &vs, int G)
{
    if (G != 1 && G != 2) return;
#ifdef LDPC_RESIDENT_PROBES                         // tuning builds only (tools/): the product library reads no environment
    const char *off = getenv("LDPC_RESIDENT_NO_LANE_OPT");
    if (off && atoi(off)) return;
#endif
    LaneCost lc{vs, order, 32, 32, G == 2 ? 16 : 32, G == 2 ? 16 : 32};
    const int n = (int)order.size();
    std::vector<std::pair<int, int>> classes;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && vs[order[j]].size() == vs[order[i]].size()) ++j;
        if (j - i >= 2 && !vs[order[i]].empty()) classes.push_back({i, j});
        i = j;
    }
    if (classes.empty()) return;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    const long trials = std::min<long>(400000, 150L * n);
    for (long it = 0; it < trials; ++it) {
        const auto &c = classes[next() % classes.size()];
        const int x = c.first + (int)(next() % (uint64_t)(c.second - c.first));
        const int y = c.first + (int)(next() % (uint64_t)(c.second - c.first));
        if (x / lc.wg == y / lc.wg) continue;
        const int before = lc.around(x, y);
        std::swap(order[x], order[y]);
        if (lc.around(x, y) > before) std::swap(order[x], order[y]);
    }
}

// Sort checks and variables by degree (stable, descending), lay the edges out ELL-transposed.
int build_resident_plan(ldpc_decoder *d, const ldpc_decoder_desc *desc)
{
    const ldpc_graph *g = d->g;
    d->res_ok = false;
    if (g->E == 0 || g->max_dv > 8) return LDPC_OK;
    // fp64 (the reference's BasicMinSumDecoder dtype): the normalised form with one factor per check; a codeword's
    // 8-byte slots take the place of a float codeword PAIR, so the geometry below must come out at G = 2
    const bool f64 = d->dtype == LDPC_F64;
    if (f64 && (d->form != LDPC_C2V_NMS || !d->beta_per_check)) return LDPC_OK;
    const int n = g->n;
    if (n > 65535 || d->n_beta > 65535 || d->n_alpha >= (1 << 24) || d->n_oms_alpha > 65535) return LDPC_OK;

    // Lane positions of the check phase are VIRTUAL checks.  A check of degree <= kResSubDegreeCap is one of them; a wider
    // one is split into 2^k sub-checks of contiguous edges (balanced, at most kResSubDegree each) that sit on ADJACENT lanes
    // and are combined by wavefront exchanges (ldpc_resident.hip: group_combine).  Groups come first, by descending size --
    // every group then starts at a multiple of its size, so it never straddles a wave -- then the whole checks.
    struct VCheck { int check, e0, dc, gs; };
    std::vector<VCheck> vc;
    {
        auto dc_real = [&](int i) { return g->h_check_ptr[i + 1] - g->h_check_ptr[i]; };
        std::vector<int> wide_ids, plain_ids;
        for (int i = 0; i < g->m; ++i) (dc_real(i) > kResSubDegreeCap ? wide_ids : plain_ids).push_back(i);
        auto group_of = [&](int i) { int k = 1; while (k * kResSubDegree < dc_real(i)) k <<= 1; return k; };
        for (int i : wide_ids)
            if (group_of(i) > 64) return LDPC_OK;                 // wider than a wavefront of sub-checks
        std::stable_sort(wide_ids.begin(), wide_ids.end(), [&](int a, int b) { return group_of(a) > group_of(b); });
        for (int i : wide_ids) {
            const int k = group_of(i), dc = dc_real(i), base = dc / k, rem = dc % k;
            int e = g->h_check_ptr[i];
            for (int j = 0; j < k; ++j) {
                const int len = base + (j < rem ? 1 : 0);
                vc.push_back({i, e, len, k});
                e += len;
            }
        }
        std::stable_sort(plain_ids.begin(), plain_ids.end(), [&](int a, int b) { return dc_real(a) > dc_real(b); });
        for (int i : plain_ids) vc.push_back({i, g->h_check_ptr[i], dc_real(i), 1});
    }
    const int m = (int)vc.size();
    const bool any_split = m != g->m;
    int max_sub = 0;
    for (const VCheck &v : vc) max_sub = std::max(max_sub, v.dc);
    if (m > 65535 || max_sub > 255) return LDPC_OK;

    // geometry: G codewords per workgroup, NT threads, and the row stride of the slot layout.
    // Two 512-thread workgroups per CU (G = 2, ds_read/write_b64) let one workgroup's barrier wait overlap
    // the other's phase -- measured best on the (1998,1512) code; larger codes fall back to one workgroup
    // per CU or G = 1.  A row stride of 512 slots (instead of m) lets LDS instructions carry t*stride as an
    // immediate offset; it is taken when it costs neither G nor workgroups per CU.
    // LDPC_RESIDENT_G / _NT override for tuning (read only by -DLDPC_RESIDENT_PROBES builds).
    auto geometry = [&](int stride, int &G_out, int &blocks_out) {
        const long long S_ = (long long)max_sub * stride;
        const int mp = is_pow2(stride) ? m : 0;           // parity words of the early-stop syndrome (power-of-two strides)
        if (S_ > 65535 || !resident_fits(d, S_, 1, 1, mp)) return false;
        int G_ = 0;
#ifdef LDPC_RESIDENT_PROBES
        const char *eg = getenv("LDPC_RESIDENT_G");
        if (eg) G_ = atoi(eg);
#endif
        if (!((G_ == 1 || G_ == 2) && resident_fits(d, S_, G_, 1, mp))) G_ = resident_fits(d, S_, 2, 1, mp) ? 2 : 1;
        int b_ = 1;
        while (b_ < 8 && resident_fits(d, S_, G_, b_ + 1, mp)) ++b_;
        G_out = G_; blocks_out = b_;
        return true;
    };
    int G = 0, blocks = 0, mstride = m;
    if (!geometry(m, G, blocks)) return LDPC_OK;
    if (f64 && G != 2) return LDPC_OK;
    if (m <= 512) {
        int G5 = 0, b5 = 0;
        if (geometry(512, G5, b5) && G5 == G && std::min(b5, 2) == std::min(blocks, 2)) { mstride = 512; blocks = b5; }
    }
    const long long S = (long long)max_sub * mstride;
    int NT = 0;
#ifdef LDPC_RESIDENT_PROBES
    { const char *en = getenv("LDPC_RESIDENT_NT"); if (en) NT = atoi(en); }
#endif
    if (NT < 64 || NT > 1024 || NT % 64) NT = blocks >= 2 ? 512 : 1024;

    std::vector<int> perm_v(n), pos_v(n);
    for (int j = 0; j < n; ++j) perm_v[j] = j;
    auto dv_of = [&](int j) { return g->h_var_ptr[j + 1] - g->h_var_ptr[j]; };
    std::stable_sort(perm_v.begin(), perm_v.end(), [&](int a, int b) { return dv_of(a) > dv_of(b); });
    std::vector<int> slot_of_edge(g->E);
    for (int p = 0; p < m; ++p)
        for (int t = 0; t < vc[p].dc; ++t) slot_of_edge[vc[p].e0 + t] = t * mstride + p;
    {   // slots are fixed by the check order alone; choose the variable order inside each degree class
        std::vector<std::vector<int>> vs(n);
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < dv_of(j); ++k) vs[j].push_back(slot_of_edge[g->h_csc[g->h_var_ptr[j] + k]]);
        optimise_lane_order(perm_v, vs, G);
    }
    for (int q = 0; q < n; ++q) pos_v[perm_v[q]] = q;

    std::vector<uint8_t> dc_s(m), gsz(m);
    std::vector<uint16_t> cvar((size_t)S, 0), bslot((size_t)S, 0), oaslot((size_t)S, 0), bslot_c(m, 0), inv(n);
    std::vector<uint32_t> vmeta(n);
    std::vector<uint2> vslot_lo(n), vslot_hi(std::max(n, 1));
    int n_hi = 0;
    std::vector<uint32_t> edge_of_slot((size_t)S, 0xffffffffu);
    bool per_check = true;
    for (int p = 0; p < m; ++p) {
        const VCheck &v = vc[p];
        const int first = g->h_check_ptr[v.check];                  // the WHOLE check's first edge decides "one beta per check"
        dc_s[p] = (uint8_t)v.dc;
        gsz[p] = (uint8_t)v.gs;
        for (int t = 0; t < v.dc; ++t) {
            const int e = v.e0 + t, slot = t * mstride + p;
            edge_of_slot[slot] = (uint32_t)e;
            cvar[slot] = (uint16_t)pos_v[g->h_var_idx[e]];
            bslot[slot] = (uint16_t)desc->beta_slot[e];
            if (desc->beta_slot[e] != desc->beta_slot[first]) per_check = false;
            if (d->form == LDPC_C2V_OMS && desc->oms_alpha) oaslot[slot] = (uint16_t)desc->oms_alpha_slot[e];
        }
        bslot_c[p] = v.dc ? (uint16_t)desc->beta_slot[first] : 0;
    }
    for (int q = 0; q < n; ++q) {
        const int j = perm_v[q], s0 = g->h_var_ptr[j], dv = dv_of(j);
        vmeta[q] = (uint32_t)dv | ((uint32_t)desc->alpha_slot[j] << 8);
        inv[j] = (uint16_t)q;
        uint32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < dv; ++k) off[k] = (uint32_t)slot_of_edge[g->h_csc[s0 + k]] * G * 4;
        vslot_lo[q] = make_uint2(off[0] | (off[1] << 16), off[2] | (off[3] << 16));      // offsets <= 65535 (resident_fits)
        vslot_hi[q] = make_uint2(off[4] | (off[5] << 16), off[6] | (off[7] << 16));
        if (dv > 4) n_hi = q + 1;                                                     // degree-sorted: they come first
    }
    ResidentPlan &pl = d->res;
    pl = ResidentPlan{};
    pl.n = n; pl.m = m; pl.S = (int)S; pl.max_dc = max_sub; pl.max_dv = g->max_dv; pl.mstride = mstride; pl.E = g->E;
    pl.any_split = any_split ? 1 : 0;
    pl.n_hi = n_hi;
    pl.par_words = is_pow2(mstride) ? m : 0;
    pl.par_shift = G == 2 ? 3 : 2;                     // slot byte offset = slot * G * 4
    int rc = plan_upload(d, &pl.dc_s, dc_s);
    if (!rc && any_split) rc = plan_upload(d, &pl.gsz, gsz);
    if (!rc) rc = plan_upload(d, &pl.cvar, cvar);
    if (!rc) rc = plan_upload(d, &pl.bslot, bslot);
    if (!rc && per_check) rc = plan_upload(d, &pl.bslot_c, bslot_c);
    if (!rc && d->form == LDPC_C2V_OMS && desc->oms_alpha) rc = plan_upload(d, &pl.oaslot, oaslot);
    if (!rc) rc = plan_upload(d, &pl.vmeta, vmeta);
    if (!rc) rc = plan_upload(d, &pl.vslot_lo, vslot_lo);
    vslot_hi.resize((size_t)std::max(n_hi, 1));
    if (!rc) rc = plan_upload(d, &pl.vslot_hi, vslot_hi);
    if (!rc) rc = plan_upload(d, &pl.inv_perm_v, inv);
    if (!rc) rc = plan_upload(d, &pl.edge_of_slot, edge_of_slot);
    if (rc) return rc;
    d->res_G = G; d->res_NT = NT;
    d->res_lds = res_lds_total((int)S, n, G, resident_alpha_floats(d), pl.par_words);
    d->res_ok = true;
    return LDPC_OK;
}

// A kernel's dynamic-LDS ceiling is process-wide state of that kernel on a device: it is raised ONCE per
// instantiation and device to the full 160 KiB (not per launch, and not to one decoder's size -- another decoder
// of a larger code launches the same instantiation).
int allow_full_lds(const void *kfn, int device)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kfn, device})) return LDPC_OK;
    HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    done.insert({kfn, device});
    return LDPC_OK;
}

template <int G>
int launch_resident(const ldpc_decoder *d, const ResidentArgs &a, hipStream_t s)
{
    size_t lds = d->res_lds;
    if (d->dtype == LDPC_F64) {                       // one fp64 codeword per workgroup in the slots of a float pair
        if (G != 2 || !d->res.bslot_c) return fail(LDPC_ERR_ARG, "internal: fp64 resident geometry");
        const unsigned blocks64 = (unsigned)a.batch;
#define LDPC_RES64(MS, SPLIT)                                                                             \
    do {                                                                                                  \
        auto kfn = a.early_stop ? resident_decode<1, FORM_NMS, true, 0, MS, 1, double, SPLIT>             \
                                : resident_decode<1, FORM_NMS, true, 0, MS, 0, double, SPLIT>;            \
        if (int rc_ = allow_full_lds((const void *)kfn, d->g->device)) return rc_;                        \
        hipLaunchKernelGGL(kfn, dim3(blocks64), dim3(d->res_NT), lds, s, d->res, a);                      \
    } while (0)
        // codes with split (wide) checks run the generic-stride instantiation with the lane-group exchange compiled in
        if (d->res.any_split) LDPC_RES64(0, true);
        else if (d->res.mstride == 512) LDPC_RES64(512, false);
        else LDPC_RES64(0, false);
#undef LDPC_RES64
        HIP_TRY(hipGetLastError());
        return LDPC_OK;
    }
    const unsigned blocks = (unsigned)((a.batch + G - 1) / G);
#ifdef LDPC_RESIDENT_PROBES
    { const char *pad = getenv("LDPC_RES_LDS_PAD"); if (pad && atoi(pad) > 0) lds = std::min<size_t>(lds + atoi(pad), 160 * 1024); }  // occupancy experiments
#endif
#define LDPC_RES_MS(FORM, NL, MS, SPLIT)                                                                 \
    do {                                                                                                 \
        auto kfn = d->res.bslot_c ? (a.early_stop ? resident_decode<G, FORM, true, NL, MS, 1, float, SPLIT>     \
                                                  : resident_decode<G, FORM, true, NL, MS, 0, float, SPLIT>)    \
                                  : (a.early_stop ? resident_decode<G, FORM, false, NL, MS, 1, float, SPLIT>    \
                                                  : resident_decode<G, FORM, false, NL, MS, 0, float, SPLIT>);  \
        if (int rc_ = allow_full_lds((const void *)kfn, d->g->device)) return rc_;                       \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(d->res_NT), lds, s, d->res, a);                       \
    } while (0)
    // codes with split (wide) checks: the generic instantiation (run-time stride and level count) with the lane-group
    // exchange compiled in -- the specialised ones stay exactly as lean as without the feature
#define LDPC_RES(FORM, NL)                                                                               \
    do {                                                                                                 \
        if (d->res.any_split) LDPC_RES_MS(FORM, 0, 0, true);                                             \
        else if (d->res.mstride == 512) LDPC_RES_MS(FORM, NL, 512, false);                               \
        else LDPC_RES_MS(FORM, NL, 0, false);                                                            \
    } while (0)
    if (d->form == LDPC_C2V_NMS) LDPC_RES(FORM_NMS, 0);
    else if (d->form == LDPC_C2V_OMS) LDPC_RES(FORM_OMS, 0);
    else if (d->n_levels == 4) LDPC_RES(FORM_RCQ, 4);       // bc = 3, the paper's and the benchmark's quantiser
    else LDPC_RES(FORM_RCQ, 0);
#undef LDPC_RES_MS
#undef LDPC_RES
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int decode_resident(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop, int32_t *bits,
                    void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed_bits, void *dbg_c2v,
                    void *stream)
{
    DeviceGuard guard(d->g->device);
    ResidentArgs a{};
    a.llr = (const float *)llr; a.batch = batch; a.T = capped_T(d); a.early_stop = early_stop != 0;
    a.beta = (const float *)d->beta; a.n_beta = d->n_beta;
    a.alpha = (const float *)d->alpha; a.n_alpha = d->n_alpha;
    a.oms_alpha = (const float *)d->oms_alpha; a.n_oms_alpha = d->n_oms_alpha;
    a.thr = d->thresholds; a.n_levels = d->n_levels; a.q_of_iter = d->q_of_iter_dev;
    a.bits = bits; a.posterior = (float *)posterior; a.iterations = iterations; a.success = success;
    a.packed = packed_bits;
    a.dbg_c2v = dbg_c2v;
#ifdef LDPC_RESIDENT_PROBES                          // phase-timing probes of tools/resident_probe*.sh; never in the product build
    { const char *dbg = getenv("LDPC_RES_DEBUG"); a.debug_skip = dbg ? atoi(dbg) : 0; }
#endif
    a.alpha_in_lds = resident_alpha_floats(d) > 0;
    a.unit_alpha = d->unit_alpha; a.rcq_zero0 = d->rcq_zero0;
    hipStream_t rs = (hipStream_t)stream;
    switch (d->res_G) {
    case 1: return launch_resident<1>(d, a, rs);
    default: return launch_resident<2>(d, a, rs);
    }
}

// ---- LDS-resident layered decode: plan (host) and launch -----------------------------------------------------
int build_layered_plan(ldpc_decoder *d, const ldpc_decoder_desc *desc)
{
    const ldpc_graph *g = d->g;
    d->lay_ok = false;
    if (d->schedule != LDPC_SCHED_LAYERED_REF || d->form != LDPC_C2V_RCQ || d->dtype != LDPC_F32) return LDPC_OK;
    if (g->E == 0 || g->m == 0 || g->max_dc > 64) return LDPC_OK;     // a check wider than a wavefront: streaming kernel
    int lw = 1;
    while (lw < g->max_dc) lw <<= 1;
    int cw = 64 / lw;
    while (cw > 1 && (size_t)cw * lay_row_bytes(g->n) > kLdsBytes) cw >>= 1;
    if ((size_t)cw * lay_row_bytes(g->n) > kLdsBytes) return LDPC_OK;  // one codeword's posteriors exceed LDS
    const int m_pad = (g->m + kLayPf - 1) / kLayPf * kLayPf;
    if ((size_t)g->n * 4u + 4u > kLayOffMask) return LDPC_OK;
    const uint32_t none = (uint32_t)g->n * 4u;                        // the codeword's +inf word
    std::vector<uint32_t> off((size_t)(m_pad + 2 * kLayPf) * lw, none);
    bool deg1 = false;
    // edges right-aligned in ascending variable order: the last lane holds the highest variable of the check
    auto var_at = [&](int row, int t) -> int {                       // variable in lane t of plan row `row`, -1: none
        if (row >= g->m) return -1;
        const int e0 = g->h_check_ptr[row], dc = g->h_check_ptr[row + 1] - e0, k = t - (lw - dc);
        return k >= 0 ? g->h_var_idx[e0 + k] : -1;
    };
    std::vector<char> group_late((size_t)m_pad / kLayPf, 0);
    group_late[0] = 1;                                                // its first row follows the previous iteration's last check
    for (int i = 0; i < m_pad; ++i) {
        const int dc = i < g->m ? g->h_check_ptr[i + 1] - g->h_check_ptr[i] : 0;
        deg1 = deg1 || dc == 1;
        // dependence on the previous plan row
        const int p = (i + m_pad - 1) % m_pad;
        int shared = 0, src_lane = -1, dst_lane = -1;
        for (int t = 0; t < lw; ++t) {
            const int v = var_at(i, t);
            if (v < 0) continue;
            for (int u = 0; u < lw; ++u)
                if (var_at(p, u) == v) { ++shared; src_lane = u; dst_lane = t; }
        }
        // one common variable whose lane in this row is the lower neighbour of its lane in the previous row: forwarded by a
        // DPP row_shl:1 (row = 16 lanes; the pair must not straddle a DPP row).  Right-aligned ascending order makes the
        // parity chain of staircase codes exactly that.  Anything else with a common variable: LATE (its whole group).
        const bool fwd = shared == 1 && src_lane == dst_lane + 1 && (src_lane / 16 == dst_lane / 16);
        if (shared > 0 && !fwd) group_late[(size_t)i / kLayPf] = 1;
        for (int t = 0; t < lw; ++t) {
            const int v = var_at(i, t);
            const uint32_t o = v >= 0 ? (uint32_t)v * 4u : none;
            off[(size_t)i * lw + t] = o | ((fwd && t == dst_lane) ? kLayFwdBit : 0u) | (dc == 1 ? kLayDeg1Bit : 0u);
        }
    }
    for (size_t gi = 0; gi < group_late.size(); ++gi)
        if (group_late[gi])
            for (int t = 0; t < lw; ++t) off[gi * kLayPf * lw + t] |= kLayLateBit;
    // magnitude 0 reconstructs to 0 under every quantiser (rcq_decoder.py:79-85, :107-119): tau_0 == 0 and no later
    // threshold <= 0 -- true for the reference's C * (j / (2^(bc-1) - 1))^gamma with gamma > 0
    bool zero0 = true;
    for (int q = 0; q < d->n_quant; ++q) {
        float rec = desc->thresholds[(size_t)q * d->n_levels];
        for (int k = 1; k < d->n_levels; ++k)
            if (0.0f >= desc->thresholds[(size_t)q * d->n_levels + k]) rec = desc->thresholds[(size_t)q * d->n_levels + k];
        zero0 = zero0 && rec == 0.0f;
    }
    int rc = upload(&d->lay_off, off.data(), off.size());
    if (rc) return rc;
    bool sorted = true;                                               // tau_1 <= tau_2 <= ... under every quantiser
    for (int q = 0; q < d->n_quant; ++q)
        for (int k = 2; k < d->n_levels; ++k)
            sorted = sorted && desc->thresholds[(size_t)q * d->n_levels + k - 1] <= desc->thresholds[(size_t)q * d->n_levels + k];
    // A power-of-two region per codeword (address = offset | row bits, one v_and_or_b32) was measured and dropped: 4 x 8192 bytes
    // leave room for only four workgroups per CU instead of five and put the four rows of a wave on the same LDS banks
    // ((1998,1512): 11.2 vs 9.6 ms, profiles/r03_layered_variants.txt).  The kernel keeps the template flag for A/B builds.
    const int row_shift = 0;
    d->lay = LayeredPlan{g->n, g->m, lw, cw, m_pad, deg1 ? 1 : 0, zero0 ? 1 : 0, sorted ? 1 : 0, row_shift, d->lay_off};
    d->lay_lds = row_shift ? ((size_t)cw << row_shift) : (size_t)cw * lay_row_bytes(g->n);
    d->lay_ok = true;
    return LDPC_OK;
}

int decode_layered_lds(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop, int32_t *bits,
                       void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed_bits, void *stream)
{
    DeviceGuard guard(d->g->device);
    hipStream_t s = (hipStream_t)stream;
    const LayeredPlan &pl = d->lay;
    const unsigned blocks = (unsigned)((batch + pl.cw - 1) / pl.cw);
#define LDPC_LAY_K(LW_, NL_, ES_, D1_, Z0_, SO_)                                                                     \
    do {                                                                                                             \
        auto kfn = pl.row_shift ? layered_lds<LW_, NL_, ES_, D1_, Z0_, SO_, true> : layered_lds<LW_, NL_, ES_, D1_, Z0_, SO_, false>; \
        if (int rc_ = allow_full_lds((const void *)kfn, d->g->device)) return rc_;                                   \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(kWave), d->lay_lds, s, pl, (const float *)llr, (long long)batch, \
                           (const float *)d->thresholds, d->n_levels, (const int *)d->q_of_iter_dev, capped_T(d),   \
                           bits, (float *)posterior, iterations, success, packed_bits);                             \
    } while (0)
    // the common case (bc = 3, no degree-1 check, zero reconstructs to zero) gets the lean instantiation per stop mode;
    // everything else the general one (run-time level count, degree-1 flag, the reference's "w < 0" sign test)
#define LDPC_LAY_NL(LW_)                                                                                             \
    do {                                                                                                             \
        const bool lean = d->n_levels == 4 && !pl.has_deg1 && pl.zero0 && pl.sorted;                                 \
        if (lean) { if (early_stop) LDPC_LAY_K(LW_, 4, true, false, true, true); else LDPC_LAY_K(LW_, 4, false, false, true, true); } \
        else { if (early_stop) LDPC_LAY_K(LW_, 0, true, true, false, false); else LDPC_LAY_K(LW_, 0, false, true, false, false); } \
    } while (0)
    switch (pl.lw) {
    case 1: LDPC_LAY_NL(1); break;
    case 2: LDPC_LAY_NL(2); break;
    case 4: LDPC_LAY_NL(4); break;
    case 8: LDPC_LAY_NL(8); break;
    case 16: LDPC_LAY_NL(16); break;
    case 32: LDPC_LAY_NL(32); break;
    default: LDPC_LAY_NL(64); break;
    }
#undef LDPC_LAY_NL
#undef LDPC_LAY_K
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

}  // namespace

// =========================================================================================== C ABI
extern "C" {

const char *ldpc_last_error(void) { return g_err; }
int ldpc_abi_version(void) { return LDPC_HIP_ABI_VERSION; }
#ifndef LDPC_SRC_HASH
#define LDPC_SRC_HASH unknown
#endif
#define LDPC_STR2(x) #x
#define LDPC_STR(x) LDPC_STR2(x)
const char *ldpc_source_hash(void) { return LDPC_STR(LDPC_SRC_HASH); }

// Host-side containers may throw; no exception crosses the C boundary.
#define LDPC_NOTHROW(call)                                                                     \
    try {                                                                                      \
        return call;                                                                           \
    } catch (const std::bad_alloc &) {                                                         \
        return fail(LDPC_ERR_HIP, "out of host memory");                                       \
    } catch (const std::exception &e) {                                                        \
        return fail(LDPC_ERR_HIP, "internal error: %s", e.what());                             \
    } catch (...) {                                                                            \
        return fail(LDPC_ERR_HIP, "internal error");                                           \
    }

static int graph_create_impl(ldpc_graph **out, int32_t n, int32_t m, int32_t E, const int32_t *check_ptr,
                             const int32_t *var_idx)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n < 0 || m < 0 || E < 0 || !check_ptr || (E > 0 && !var_idx)) return fail(LDPC_ERR_ARG, "bad graph sizes");
    if (check_ptr[0] != 0 || check_ptr[m] != E) return fail(LDPC_ERR_ARG, "check_ptr must span [0, E]");
    std::vector<int> dv(n, 0);
    int max_dc = 0;
    for (int i = 0; i < m; ++i) {
        const int a = check_ptr[i], b = check_ptr[i + 1];
        if (b < a || b > E) return fail(LDPC_ERR_ARG, "check_ptr not monotone at check %d", i);
        max_dc = std::max(max_dc, b - a);
        for (int e = a; e < b; ++e) {
            const int j = var_idx[e];
            if (j < 0 || j >= n) return fail(LDPC_ERR_ARG, "var_idx[%d]=%d out of range", e, j);
            if (e > a && var_idx[e - 1] >= j) return fail(LDPC_ERR_ARG, "edges of check %d not strictly ascending", i);
            dv[j]++;
        }
    }
    std::vector<int> var_ptr(n + 1, 0), csc(std::max(E, 1), 0), fill(n, 0);
    int max_dv = 0;
    for (int j = 0; j < n; ++j) {
        var_ptr[j + 1] = var_ptr[j] + dv[j];
        max_dv = std::max(max_dv, dv[j]);
    }
    // scanning CSR edges in order visits the checks of every variable in ascending order
    for (int e = 0; e < E; ++e) {
        const int j = var_idx[e];
        csc[var_ptr[j] + fill[j]++] = e;
    }
    ldpc_graph *g = new (std::nothrow) ldpc_graph();
    if (!g) return fail(LDPC_ERR_ARG, "out of host memory");
    g->n = n; g->m = m; g->E = E; g->max_dc = max_dc; g->max_dv = max_dv;
    g->h_check_ptr.assign(check_ptr, check_ptr + m + 1);
    g->h_var_idx.assign(var_idx, var_idx + E);
    g->h_var_ptr = var_ptr;
    g->h_csc.assign(csc.begin(), csc.begin() + E);
    g->h_check_of_edge.resize(E);
    for (int i = 0; i < m; ++i)
        for (int e = check_ptr[i]; e < check_ptr[i + 1]; ++e) g->h_check_of_edge[e] = i;
    if (hipGetDevice(&g->device) != hipSuccess) {
        delete g;
        return fail(LDPC_ERR_HIP, "no HIP device available");
    }
    int rc = upload(&g->check_ptr, check_ptr, (size_t)m + 1);
    if (!rc) rc = upload(&g->var_idx, var_idx, (size_t)E);
    if (!rc) rc = upload(&g->var_ptr, var_ptr.data(), (size_t)n + 1);
    if (!rc) rc = upload(&g->csc_edge, csc.data(), (size_t)E);
    std::vector<int> wide;
    for (int i = 0; i < m; ++i)
        if (check_ptr[i + 1] - check_ptr[i] > kWideCheck) wide.push_back(i);
#ifdef LDPC_NO_WIDE_KERNEL                            // A/B timing builds of tools/time_wide.py only
    wide.clear();
#endif
    g->n_wide = (int)wide.size();
    if (!rc && g->n_wide) rc = upload(&g->wide_checks, wide.data(), wide.size());
    if (rc) {
        ldpc_graph_destroy(g);
        return rc;
    }
    *out = g;
    return LDPC_OK;
}

int ldpc_graph_create(ldpc_graph **out, int32_t n, int32_t m, int32_t E, const int32_t *check_ptr,
                      const int32_t *var_idx)
{
    LDPC_NOTHROW(graph_create_impl(out, n, m, E, check_ptr, var_idx))
}

void ldpc_graph_destroy(ldpc_graph *g)
{
    if (!g) return;
    DeviceGuard guard(g->device);
    (void)hipFree(g->check_ptr); (void)hipFree(g->var_idx); (void)hipFree(g->var_ptr); (void)hipFree(g->csc_edge);
    (void)hipFree(g->wide_checks);
    delete g;
}

int ldpc_graph_info(const ldpc_graph *g, int32_t out5[5])
{
    if (!g || !out5) return fail(LDPC_ERR_ARG, "NULL argument");
    out5[0] = g->n; out5[1] = g->m; out5[2] = g->E; out5[3] = g->max_dc; out5[4] = g->max_dv;
    return LDPC_OK;
}

static int decoder_create_impl(ldpc_decoder **out, const ldpc_graph *g, const ldpc_decoder_desc *desc)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!g || !desc) return fail(LDPC_ERR_ARG, "NULL argument");
    if (desc->dtype != LDPC_F32 && desc->dtype != LDPC_F64) return fail(LDPC_ERR_ARG, "bad dtype");
    if (desc->c2v_form < LDPC_C2V_NMS || desc->c2v_form > LDPC_C2V_OMS) return fail(LDPC_ERR_ARG, "bad c2v_form");
    if (desc->iters < 0) return fail(LDPC_ERR_ARG, "iters < 0");
    if (desc->n_beta_slots < 1 || desc->n_alpha_slots < 1 || !desc->beta || !desc->alpha ||
        (g->E > 0 && !desc->beta_slot) || (g->n > 0 && !desc->alpha_slot))
        return fail(LDPC_ERR_ARG, "weight tables missing");
    for (int e = 0; e < g->E; ++e)
        if (desc->beta_slot[e] < 0 || desc->beta_slot[e] >= desc->n_beta_slots) return fail(LDPC_ERR_ARG, "beta_slot[%d] out of range", e);
    for (int j = 0; j < g->n; ++j)
        if (desc->alpha_slot[j] < 0 || desc->alpha_slot[j] >= desc->n_alpha_slots) return fail(LDPC_ERR_ARG, "alpha_slot[%d] out of range", j);
    // association orders restated on the device: torch.sum below its cascade level, np.sum one block
    if (desc->dtype == LDPC_F32 && g->max_dv > 575) return fail(LDPC_ERR_UNSUPPORTED, "variable degree %d > 575 (fp32 sum order)", g->max_dv);
    if (desc->dtype == LDPC_F64 && g->max_dv > 128) return fail(LDPC_ERR_UNSUPPORTED, "variable degree %d > 128 (fp64 sum order)", g->max_dv);
    if (desc->schedule < LDPC_SCHED_FLOODING || desc->schedule > LDPC_SCHED_LAYERED) return fail(LDPC_ERR_ARG, "bad schedule");
    if (desc->schedule != LDPC_SCHED_FLOODING) {
        if (desc->c2v_form != LDPC_C2V_RCQ || desc->dtype != LDPC_F32)
            return fail(LDPC_ERR_UNSUPPORTED, "the layered schedule exists for the fp32 RCQ decoder only (rcq_decoder.py:281-350)");
        if (g->m == 1 && desc->schedule == LDPC_SCHED_LAYERED_REF)
            return fail(LDPC_ERR_UNSUPPORTED, "the reference's layered schedule on a single-check code");
    }
    if (desc->c2v_form == LDPC_C2V_RCQ) {
        if (desc->dtype != LDPC_F32) return fail(LDPC_ERR_UNSUPPORTED, "RCQ messages are fp32 only");
        if (desc->n_levels < 1 || desc->n_levels > 128) return fail(LDPC_ERR_UNSUPPORTED, "n_levels %d outside 1..128 (bc 1..8)", desc->n_levels);
        if (desc->n_quantizers < 1 || !desc->thresholds || (desc->iters > 0 && !desc->q_of_iter)) return fail(LDPC_ERR_ARG, "quantiser tables missing");
        for (int t = 0; t < desc->iters; ++t)
            if (desc->q_of_iter[t] < 0 || desc->q_of_iter[t] >= desc->n_quantizers) return fail(LDPC_ERR_ARG, "q_of_iter[%d] out of range", t);
    }
    if (desc->c2v_form == LDPC_C2V_OMS && desc->oms_alpha) {
        if (desc->n_oms_alpha_slots < 1 || (g->E > 0 && !desc->oms_alpha_slot)) return fail(LDPC_ERR_ARG, "oms_alpha tables missing");
        for (int e = 0; e < g->E; ++e)
            if (desc->oms_alpha_slot[e] < 0 || desc->oms_alpha_slot[e] >= desc->n_oms_alpha_slots) return fail(LDPC_ERR_ARG, "oms_alpha_slot[%d] out of range", e);
    }

    DeviceGuard guard(g->device);
    ldpc_decoder *d = new (std::nothrow) ldpc_decoder();
    if (!d) return fail(LDPC_ERR_ARG, "out of host memory");
    d->g = g; d->dtype = desc->dtype; d->form = desc->c2v_form; d->T = desc->iters; d->schedule = desc->schedule;
    d->n_beta = desc->n_beta_slots; d->n_alpha = desc->n_alpha_slots;
    const size_t es = d->elem();
    const size_t rows = (size_t)std::max(d->T, 1);
    int rc = LDPC_OK;
    auto up_bytes = [&](void **dst, const void *src, size_t bytes) {
        char *p = nullptr;
        int r = upload(&p, (const char *)src, bytes);
        *dst = p;
        return r;
    };
    // tables have max(T,1) rows on the device; with T == 0 row 0 is never read for arithmetic
    if (d->T > 0) {
        rc = up_bytes(&d->beta, desc->beta, rows * d->n_beta * es);
        if (!rc) rc = up_bytes(&d->alpha, desc->alpha, rows * d->n_alpha * es);
    } else {
        rc = up_bytes(&d->beta, nullptr, (size_t)d->n_beta * es);
        if (!rc) rc = up_bytes(&d->alpha, nullptr, (size_t)d->n_alpha * es);
    }
    if (!rc) rc = upload(&d->beta_slot, desc->beta_slot, (size_t)g->E);
    d->beta_per_check = true;
    for (int i = 0; i < g->m && d->beta_per_check; ++i)
        for (int e = g->h_check_ptr[i] + 1; e < g->h_check_ptr[i + 1]; ++e)
            if (desc->beta_slot[e] != desc->beta_slot[g->h_check_ptr[i]]) { d->beta_per_check = false; break; }
    if (!rc) rc = upload(&d->alpha_slot, desc->alpha_slot, (size_t)g->n);
    if (!rc && d->form == LDPC_C2V_RCQ) {
        d->n_levels = desc->n_levels; d->n_quant = desc->n_quantizers;
        d->q_of_iter.assign(desc->q_of_iter, desc->q_of_iter + d->T);
        if (d->q_of_iter.empty()) d->q_of_iter.push_back(0);
        const size_t L = d->n_levels, Q = d->n_quant;
        rc = upload(&d->thresholds, desc->thresholds, Q * L);
        // signed reconstruction LUT: value of code c = (1 - 2*sign_bit) * tau[c mod L]  (rcq_decoder.py:107-119)
        std::vector<float> lut(Q * 2 * L);
        for (size_t q = 0; q < Q; ++q)
            for (size_t c = 0; c < 2 * L; ++c) {
                const float sb = c >= L ? 1.0f : 0.0f;
                lut[q * 2 * L + c] = (1.0f - 2.0f * sb) * desc->thresholds[q * L + (c % L)];
            }
        if (!rc) rc = upload(&d->lut, lut.data(), lut.size());
        if (!rc) rc = upload(&d->q_of_iter_dev, d->q_of_iter.data(), d->q_of_iter.size());
        if (!rc && Q * 2 * L * sizeof(float) > 64 * 1024) rc = fail(LDPC_ERR_UNSUPPORTED, "quantiser LUTs exceed 64 KiB of LDS");
    }
    if (!rc && d->form == LDPC_C2V_OMS && desc->oms_alpha && d->T > 0) {
        d->n_oms_alpha = desc->n_oms_alpha_slots;
        rc = up_bytes(&d->oms_alpha, desc->oms_alpha, rows * d->n_oms_alpha * es);
        if (!rc) rc = upload(&d->oms_alpha_slot, desc->oms_alpha_slot, (size_t)g->E);
    }
    if (!rc && d->dtype == LDPC_F32 && d->form != LDPC_C2V_RCQ && d->schedule == LDPC_SCHED_FLOODING) {
        auto invert = [&](const int32_t *slot, int count, int n_slots, int **ptr_dev, int **items_dev) {
            std::vector<int> ptr((size_t)n_slots + 1, 0), items((size_t)std::max(count, 1), 0);
            for (int x = 0; x < count; ++x) ptr[slot[x] + 1]++;
            for (int k = 0; k < n_slots; ++k) ptr[k + 1] += ptr[k];
            std::vector<int> fill(ptr.begin(), ptr.end() - 1);
            for (int x = 0; x < count; ++x) items[fill[slot[x]]++] = x;         // ascending x inside a slot
            int r = upload(ptr_dev, ptr.data(), ptr.size());
            if (!r) r = upload(items_dev, items.data(), items.size());
            return r;
        };
        rc = invert(desc->beta_slot, g->E, d->n_beta, &d->beta_inv_ptr, &d->beta_inv_items);
        if (!rc) rc = invert(desc->alpha_slot, g->n, d->n_alpha, &d->alpha_inv_ptr, &d->alpha_inv_items);
        if (!rc && d->oms_alpha) rc = invert(desc->oms_alpha_slot, g->E, d->n_oms_alpha, &d->oms_inv_ptr, &d->oms_inv_items);
    }
    if (!rc && d->form == LDPC_C2V_RCQ && d->dtype == LDPC_F32 && d->schedule == LDPC_SCHED_FLOODING &&
        g->E > 0 && g->max_dv <= 8 && (long long)g->E * 256 < (1ll << 31) && (long long)g->n * 1024 < (1ll << 31)) {
        // gather metadata of the fused RCQ iteration (cn_gather): per CSR edge its variable, where the list of
        // the variable's OTHER edges starts (ascending check order = CSC order), how many there are, alpha column
        std::vector<int4> meta((size_t)g->E + 1);
        std::vector<int> nbr;
        nbr.reserve((size_t)g->E * 3 + 8);
        for (int e = 0; e < g->E; ++e) {
            const int j = g->h_var_idx[e], s0 = g->h_var_ptr[j], dv = g->h_var_ptr[j + 1] - s0;
            meta[e] = make_int4(j, (int)nbr.size(), dv - 1, desc->alpha_slot[j]);
            for (int k = 0; k < dv; ++k)
                if (g->h_csc[s0 + k] != e) nbr.push_back(g->h_csc[s0 + k]);
        }
        meta[g->E] = make_int4(0, 0, 0, 0);           // what the prefetch past the last edge of the last check reads
        nbr.resize(nbr.size() + 8, 0);
        rc = upload(&d->gat_meta, meta.data(), meta.size());
        if (!rc) rc = upload(&d->gat_nbr, nbr.data(), nbr.size());
        d->gat_ok = !rc;
    }
    if (!rc && d->form == LDPC_C2V_RCQ && d->dtype == LDPC_F32 && d->schedule == LDPC_SCHED_FLOODING && g->E > 0 &&
        d->beta_per_check && d->n_levels <= 62) {
        // code-pair form: level(|beta * x|) must be non-decreasing in x, i.e. thresholds 1.. of every quantiser sorted, and
        // positive (level(0) = 0; the magnitude compares run on the bit patterns)
        // (threshold 0 never decides a level, rcq_decoder.py:79-85)
        bool sorted = true;
        for (int q = 0; q < d->n_quant && sorted; ++q)
            for (int k = 1; k < d->n_levels; ++k) {
                const float t = desc->thresholds[(size_t)q * d->n_levels + k];
                if (!(t > 0.0f) || (k > 1 && t < desc->thresholds[(size_t)q * d->n_levels + k - 1])) { sorted = false; break; }
            }
        d->pair_ok = sorted;
        bool in_range = sorted && d->n_levels == 4 && LDPC_KEY_FLOAT != 0;
        for (int q = 0; q < d->n_quant && in_range; ++q)
            for (int k = 1; k < d->n_levels; ++k) {
                const float t = desc->thresholds[(size_t)q * d->n_levels + k];
                if (!(t >= 0x1p-50f && t <= 0x1p50f)) in_range = false;
            }
        d->key_float4 = in_range;
    }
    resident_table_flags(d, desc->alpha, d->form == LDPC_C2V_RCQ ? desc->thresholds : nullptr);
    if (!rc && d->schedule == LDPC_SCHED_FLOODING) rc = build_resident_plan(d, desc);
    if (!rc) rc = build_layered_plan(d, desc);
    if (rc) {
        ldpc_decoder_destroy(d);
        return rc;
    }
    *out = d;
    return LDPC_OK;
}

int ldpc_decoder_create(ldpc_decoder **out, const ldpc_graph *g, const ldpc_decoder_desc *desc)
{
    LDPC_NOTHROW(decoder_create_impl(out, g, desc))
}

int ldpc_decoder_set_mode(ldpc_decoder *d, int32_t mode)
{
    if (!d) return fail(LDPC_ERR_ARG, "NULL decoder");
    if (mode < LDPC_MODE_AUTO || mode > LDPC_MODE_PAIR) return fail(LDPC_ERR_ARG, "bad mode");
    if (mode == LDPC_MODE_GATHER && !d->gat_ok)
        return fail(LDPC_ERR_UNSUPPORTED, "the fused RCQ iteration needs an fp32 flooding RCQ decoder with variable degree <= 8");
    if (mode == LDPC_MODE_PAIR && !d->pair_ok)
        return fail(LDPC_ERR_UNSUPPORTED, "the code-pair form needs an fp32 flooding RCQ decoder with one beta per check, "
                                          "sorted thresholds and at most 62 levels");
    if (mode == LDPC_MODE_RESIDENT && !d->res_ok && !d->lay_ok)
        return fail(LDPC_ERR_UNSUPPORTED, "code does not qualify for the LDS-resident engine "
                                          "(dv <= 8, check degree <= 1024, state within 160 KiB of LDS)");
    d->mode = mode;
    return LDPC_OK;
}

int ldpc_decoder_info(const ldpc_decoder *d, int32_t out4[4])
{
    if (!d || !out4) return fail(LDPC_ERR_ARG, "NULL argument");
    out4[0] = (use_resident(d) || use_layered_lds(d)) ? LDPC_MODE_RESIDENT : use_pair(d) ? LDPC_MODE_PAIR : use_gather(d) ? LDPC_MODE_GATHER : LDPC_MODE_SWEEPS;
    out4[1] = d->res_ok ? (d->dtype == LDPC_F64 ? 1 : d->res_G) : 0;      // fp64: one codeword in a float pair's slots
    out4[2] = d->res_ok ? d->res_NT : 0;
    out4[3] = d->res_ok ? (int32_t)d->res_lds : 0;
    if (d->lay_ok) { out4[1] = d->lay.cw; out4[2] = kWave; out4[3] = (int32_t)d->lay_lds; }   // layered: codewords per (one-wave) workgroup
    return LDPC_OK;
}

int ldpc_decoder_set_weights(ldpc_decoder *d, const void *beta, const void *alpha, const void *oms_alpha,
                             void *stream)
{
    if (!d) return fail(LDPC_ERR_ARG, "NULL decoder");
    if (d->T == 0) return LDPC_OK;
    DeviceGuard guard(d->g->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t es = d->elem(), rows = (size_t)d->T;
    if (beta) HIP_TRY(hipMemcpyAsync(d->beta, beta, rows * d->n_beta * es, hipMemcpyHostToDevice, s));
    if (alpha) {
        HIP_TRY(hipMemcpyAsync(d->alpha, alpha, rows * d->n_alpha * es, hipMemcpyHostToDevice, s));
        resident_table_flags(d, alpha, nullptr);
    }
    if (oms_alpha) {
        if (!d->oms_alpha) return fail(LDPC_ERR_ARG, "decoder was created without oms_alpha");
        HIP_TRY(hipMemcpyAsync(d->oms_alpha, oms_alpha, rows * d->n_oms_alpha * es, hipMemcpyHostToDevice, s));
    }
    return LDPC_OK;
}

void ldpc_decoder_destroy(ldpc_decoder *d)
{
    if (!d) return;
    DeviceGuard guard(d->g->device);
    (void)hipFree(d->beta); (void)hipFree(d->alpha); (void)hipFree(d->oms_alpha);
    (void)hipFree(d->beta_slot); (void)hipFree(d->alpha_slot); (void)hipFree(d->oms_alpha_slot);
    (void)hipFree(d->thresholds); (void)hipFree(d->lut); (void)hipFree(d->q_of_iter_dev);
    for (void *p : d->res_bufs) (void)hipFree(p);
    (void)hipFree(d->gat_meta); (void)hipFree(d->gat_nbr); (void)hipFree(d->lay_off);
    (void)hipFree(d->beta_inv_ptr); (void)hipFree(d->beta_inv_items); (void)hipFree(d->alpha_inv_ptr);
    (void)hipFree(d->alpha_inv_items); (void)hipFree(d->oms_inv_ptr); (void)hipFree(d->oms_inv_items);
    delete d;
}

size_t ldpc_decoder_workspace_bytes(const ldpc_decoder *d, int64_t batch)
{
    if (!d || batch < 0) return 0;
    if (use_resident(d) || use_layered_lds(d)) return kAlign;   // the LDS-resident engines keep their state in LDS
    return carve(d, batch, nullptr).total;
}

int ldpc_decode(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop, int32_t *bits,
                void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed_bits,
                void *workspace, size_t workspace_bytes, void *stream)
{
    if (!d) return fail(LDPC_ERR_ARG, "NULL decoder");
    if (batch < 0) return fail(LDPC_ERR_ARG, "batch < 0");
    if (batch == 0) return LDPC_OK;
    if (!llr || !workspace) return fail(LDPC_ERR_ARG, "NULL llr/workspace");
    if (d->g->n == 0) return LDPC_OK;
    if (((uintptr_t)workspace % kAlign) != 0) return fail(LDPC_ERR_ARG, "workspace must be %zu-byte aligned", kAlign);
    if (use_resident(d))
        return decode_resident(d, llr, batch, early_stop, bits, posterior, iterations, success, packed_bits, nullptr, stream);
    if (use_layered_lds(d))
        return decode_layered_lds(d, llr, batch, early_stop, bits, posterior, iterations, success, packed_bits, stream);
    const Workspace w = carve(d, batch, workspace);
    if (w.total > workspace_bytes) return fail(LDPC_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.total);
    if ((size_t)w.tiles * ((d->g->n + 3) / 4) > 0x7fffffffull) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one launch");
    DeviceGuard guard(d->g->device);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == LDPC_F64)
        return decode_dispatch<double>(d, llr, batch, early_stop != 0, bits, posterior, iterations, success, packed_bits, w, s);
    return decode_dispatch<float>(d, llr, batch, early_stop != 0, bits, posterior, iterations, success, packed_bits, w, s);
}

int ldpc_decode_capped(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop, int32_t max_iterations,
                       int32_t *bits, void *posterior, int32_t *iterations, uint8_t *success, uint8_t *packed_bits,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    if (max_iterations < 1) return fail(LDPC_ERR_ARG, "max_iterations must be >= 1");
    struct Cap {
        explicit Cap(int c) { tl_iter_cap = c; }
        ~Cap() { tl_iter_cap = INT_MAX; }
    } cap(max_iterations);
    return ldpc_decode(d, llr, batch, early_stop, bits, posterior, iterations, success, packed_bits, workspace, workspace_bytes,
                       stream);
}

}  // extern "C"

// ---- gradient (training) path: ldpc_train.hip ----------------------------------------------------------------
namespace {
int train_supported(const ldpc_decoder *d)
{
    if (!d) return fail(LDPC_ERR_ARG, "NULL decoder");
    if (d->dtype != LDPC_F32 || d->form == LDPC_C2V_RCQ || d->schedule != LDPC_SCHED_FLOODING)
        return fail(LDPC_ERR_UNSUPPORTED, "gradients exist for the fp32 normalised / offset min-sum flooding decoders "
                                          "(the reference's RCQ quantiser passes no gradient)");
    return LDPC_OK;
}

struct BackwardWs {
    int vec = 0, tiles = 0;
    float *llrT = nullptr, *gpostT = nullptr, *gv2c = nullptr, *gc2v = nullptr, *gbeta = nullptr, *galpha = nullptr;
    float *goa = nullptr;             // offset form: per-edge partials of the check-side alpha
    float *gllrT = nullptr;           // accumulator of d loss/d llr (starts as a copy of gpostT)
    size_t part_bytes = 0, total = 0;
};
BackwardWs carve_backward(const ldpc_decoder *d, int64_t batch, void *base)
{
    BackwardWs w;
    w.vec = pick_vec(d, batch);
    const int W = 64 * w.vec;
    w.tiles = (int)std::max<int64_t>((batch + W - 1) / W, 1);
    const size_t n = d->g->n, E = std::max(d->g->E, 1), tw = (size_t)w.tiles * W, T = std::max(d->T, 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_llr = take(tw * n * 4), o_gp = take(tw * n * 4), o_gv = take(tw * E * 4), o_gc = take(tw * E * 4);
    const size_t o_gb = take(T * w.tiles * E * 4), o_ga = take(T * w.tiles * n * 4);
    const size_t o_goa = take(d->form == LDPC_C2V_OMS ? T * w.tiles * E * 4 : 0);
    w.part_bytes = off - o_gb;
    const size_t o_gl = take(tw * n * 4);
    w.total = off;
    if (base) {
        char *b = (char *)base;
        w.llrT = (float *)(b + o_llr); w.gpostT = (float *)(b + o_gp); w.gv2c = (float *)(b + o_gv);
        w.gc2v = (float *)(b + o_gc); w.gbeta = (float *)(b + o_gb); w.galpha = (float *)(b + o_ga);
        w.goa = (float *)(b + o_goa);
        w.gllrT = (float *)(b + o_gl);
    }
    return w;
}

template <int VEC>
int backward_impl(const ldpc_decoder *d, const char *saved, const float *llr, int64_t batch, const int32_t *iterations,
                  const float *grad_posterior, float *grad_beta, float *grad_alpha, float *grad_oms_alpha,
                  float *grad_llr, const BackwardWs &w, hipStream_t s)
{
    constexpr int W = 64 * VEC;
    constexpr int JT = transpose_vars<float>();
    const GraphDev g = d->g->dev();
    const int T = d->T, vc = (g.n + JT - 1) / JT;
    const SavedLayout sl = saved_layout(d, w.tiles, W);
    const dim3 tgrid((unsigned)((size_t)w.tiles * VEC * vc)), blk(kBlock);
    hipLaunchKernelGGL((transpose_in<float, VEC>), tgrid, blk, 0, s, llr, w.llrT, (long long)batch, g.n, vc);
    hipLaunchKernelGGL((transpose_in<float, VEC>), tgrid, blk, 0, s, grad_posterior, w.gpostT, (long long)batch, g.n, vc);
    HIP_TRY(hipMemsetAsync(w.gbeta, 0, w.part_bytes, s));
    if (grad_llr)
        HIP_TRY(hipMemcpyAsync(w.gllrT, w.gpostT, (size_t)w.tiles * W * g.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    const int cb = (g.m + kWavesPerBlock - 1) / kWavesPerBlock, vb = (g.n + kWavesPerBlock - 1) / kWavesPerBlock;
    const int vbb = (g.n + kWavesPerBlock * kVnbVarsPerWave - 1) / (kWavesPerBlock * kVnbVarsPerWave);   // vn_backward: several variables per wave
    const dim3 cgrid((unsigned)((size_t)w.tiles * cb)), vgrid((unsigned)((size_t)w.tiles * vb)), vbgrid((unsigned)((size_t)w.tiles * vbb));
    const size_t epart = (size_t)w.tiles * g.E, vpart = (size_t)w.tiles * g.n;
    for (int t = T - 1; t >= 0; --t) {
        const float *beta_row = (const float *)d->beta + (size_t)t * d->n_beta;
        // d loss/d c2v_t is in w.gc2v (written by the variable pass of step t+1; unread at t == T-1)
        const bool oms = d->form == LDPC_C2V_OMS;
        float *goa = (oms && d->oms_alpha) ? w.goa + (size_t)t * epart : nullptr;
#define LDPC_CNB(FIRST_, FORM_, SRC_, OUT_)                                                                            \
    hipLaunchKernelGGL((cn_backward<VEC, FIRST_, FORM_>), cgrid, blk, 0, s, g, (const float *)(SRC_),                  \
                       (const float *)w.gc2v, (const float *)w.gpostT, iterations, (long long)batch, t, beta_row,     \
                       (const int *)d->beta_slot, (float *)(OUT_), w.gbeta + (size_t)t * epart, goa, cb)
        float *gv0 = grad_llr ? w.gv2c : nullptr;        // d loss/d v2c_0 is needed only for the input gradient
        if (t == 0) {
            if (oms) LDPC_CNB(true, FORM_OMS, w.llrT, gv0); else LDPC_CNB(true, FORM_NMS, w.llrT, gv0);
        } else {
            if (oms) LDPC_CNB(false, FORM_OMS, saved + sl.v2c_off(t), w.gv2c);
            else LDPC_CNB(false, FORM_NMS, saved + sl.v2c_off(t), w.gv2c);
            const float *alpha_row = (const float *)d->alpha + (size_t)(t - 1) * d->n_alpha;
            hipLaunchKernelGGL((vn_backward<VEC>), vbgrid, blk, 0, s, g, (const float *)(saved + sl.c2v_off(t - 1)),
                               (const float *)w.gv2c, iterations, (long long)batch, t, alpha_row, (const int *)d->alpha_slot,
                               w.gc2v, w.galpha + (size_t)(t - 1) * vpart, vbb);
        }
        if (grad_llr)                                     // g_llr += sum over the edges of every variable of g_v2c_t
            hipLaunchKernelGGL((llr_backward_accumulate<VEC>), vgrid, blk, 0, s, g, (const float *)w.gv2c, w.gllrT, vb);
#undef LDPC_CNB
    }
    HIP_TRY(hipGetLastError());
    if (grad_llr)
        hipLaunchKernelGGL((transpose_out<float, VEC>), tgrid, blk, 0, s, (const float *)w.gllrT, (const uint64_t *)nullptr, grad_llr,
                           (int *)nullptr, (long long)batch, g.n, vc);     // [tile][n][W] -> [batch][n], padding rows dropped
    // fixed-order reductions (one wave per slot and iteration): every table entry is written, no memset, no atomics
    if (grad_beta)
        hipLaunchKernelGGL(reduce_table_grads, dim3((unsigned)d->n_beta, (unsigned)T), dim3(kWave), 0, s,
                           (const float *)w.gbeta, w.tiles, g.E, (const int *)d->beta_inv_ptr,
                           (const int *)d->beta_inv_items, d->n_beta, grad_beta);
    if (grad_oms_alpha && d->form == LDPC_C2V_OMS && d->oms_alpha)
        hipLaunchKernelGGL(reduce_table_grads, dim3((unsigned)d->n_oms_alpha, (unsigned)T), dim3(kWave), 0, s,
                           (const float *)w.goa, w.tiles, g.E, (const int *)d->oms_inv_ptr,
                           (const int *)d->oms_inv_items, d->n_oms_alpha, grad_oms_alpha);
    if (grad_alpha)
        hipLaunchKernelGGL(reduce_table_grads, dim3((unsigned)d->n_alpha, (unsigned)T), dim3(kWave), 0, s,
                           (const float *)w.galpha, w.tiles, g.n, (const int *)d->alpha_inv_ptr,
                           (const int *)d->alpha_inv_items, d->n_alpha, grad_alpha);
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}
}  // namespace

extern "C" {

size_t ldpc_train_saved_bytes(const ldpc_decoder *d, int64_t batch)
{
    if (!d || batch <= 0) return 0;
    const int vec = pick_vec(d, batch), W = 64 * vec;
    const size_t total = saved_layout(d, (int)((batch + W - 1) / W), W).total();
    return total ? total : kAlign;
}

size_t ldpc_train_workspace_bytes(const ldpc_decoder *d, int64_t batch)
{
    if (!d || batch < 0) return 0;
    return std::max(carve(d, batch, nullptr).total, carve_backward(d, batch, nullptr).total);
}

int ldpc_decode_saving(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop, int32_t *bits,
                       void *posterior, int32_t *iterations, uint8_t *success, void *saved, size_t saved_bytes,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = train_supported(d)) return rc;
    if (batch < 0) return fail(LDPC_ERR_ARG, "batch < 0");
    if (batch == 0 || d->g->n == 0) return LDPC_OK;
    if (!llr || !workspace || !saved) return fail(LDPC_ERR_ARG, "NULL llr/workspace/saved");
    if (((uintptr_t)workspace % kAlign) != 0 || ((uintptr_t)saved % kAlign) != 0)
        return fail(LDPC_ERR_ARG, "workspace and saved state must be %zu-byte aligned", kAlign);
    const Workspace w = carve(d, batch, workspace);
    if (w.total > workspace_bytes) return fail(LDPC_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.total);
    if (ldpc_train_saved_bytes(d, batch) > saved_bytes) return fail(LDPC_ERR_WORKSPACE, "saved-state buffer too small");
    if ((size_t)w.tiles * ((d->g->n + 3) / 4) > 0x7fffffffull) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one launch");
    DeviceGuard guard(d->g->device);
    return decode_dispatch<float>(d, llr, batch, early_stop != 0, bits, posterior, iterations, success, nullptr, w,
                                  (hipStream_t)stream, (char *)saved);
}

int ldpc_backward(const ldpc_decoder *d, const void *saved, size_t saved_bytes, const void *llr, int64_t batch,
                  const int32_t *iterations, const void *grad_posterior, void *grad_beta, void *grad_alpha,
                  void *grad_oms_alpha, void *grad_llr, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = train_supported(d)) return rc;
    if (batch < 0) return fail(LDPC_ERR_ARG, "batch < 0");
    if (!grad_beta && !grad_alpha && !grad_oms_alpha && !grad_llr) return LDPC_OK;
    DeviceGuard guard(d->g->device);
    hipStream_t s = (hipStream_t)stream;
    if (batch == 0 || d->g->n == 0 || d->T == 0 || d->g->E == 0) {       // no iteration ran: the posterior is the LLR
        if (grad_beta) HIP_TRY(hipMemsetAsync(grad_beta, 0, (size_t)std::max(d->T, 1) * d->n_beta * 4, s));
        if (grad_alpha) HIP_TRY(hipMemsetAsync(grad_alpha, 0, (size_t)std::max(d->T, 1) * d->n_alpha * 4, s));
        if (grad_oms_alpha && d->n_oms_alpha > 0)
            HIP_TRY(hipMemsetAsync(grad_oms_alpha, 0, (size_t)std::max(d->T, 1) * d->n_oms_alpha * 4, s));
        if (grad_llr && batch > 0 && d->g->n > 0) {              // no iteration ran: posterior == llr
            if (!grad_posterior) return fail(LDPC_ERR_ARG, "NULL grad_posterior");
            HIP_TRY(hipMemcpyAsync(grad_llr, grad_posterior, (size_t)batch * d->g->n * 4, hipMemcpyDeviceToDevice, s));
        }
        return LDPC_OK;
    }
    if (!saved || !llr || !iterations || !grad_posterior || !workspace) return fail(LDPC_ERR_ARG, "NULL argument");
    if (((uintptr_t)workspace % kAlign) != 0 || ((uintptr_t)saved % kAlign) != 0)
        return fail(LDPC_ERR_ARG, "workspace and saved state must be %zu-byte aligned", kAlign);
    if (ldpc_train_saved_bytes(d, batch) > saved_bytes) return fail(LDPC_ERR_WORKSPACE, "saved-state buffer too small");
    const BackwardWs w = carve_backward(d, batch, workspace);
    if (w.total > workspace_bytes) return fail(LDPC_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.total);
    if ((size_t)w.tiles * ((d->g->n + 3) / 4) > 0x7fffffffull) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one launch");
    if (w.vec == 1)
        return backward_impl<1>(d, (const char *)saved, (const float *)llr, batch, iterations, (const float *)grad_posterior,
                                (float *)grad_beta, (float *)grad_alpha, (float *)grad_oms_alpha, (float *)grad_llr, w, s);
    return backward_impl<4>(d, (const char *)saved, (const float *)llr, batch, iterations, (const float *)grad_posterior,
                            (float *)grad_beta, (float *)grad_alpha, (float *)grad_oms_alpha, (float *)grad_llr, w, s);
}

int ldpc_debug_key4(const float *values, int64_t count, float beta, const float thresholds4[4], uint8_t *keys_float,
                    uint8_t *keys_compare, void *stream)
{
    if (!values || !thresholds4 || !keys_float || !keys_compare || count <= 0) return fail(LDPC_ERR_ARG, "bad argument");
    const long long pairs = (count + 1) / 2;
    hipLaunchKernelGGL(debug_key4, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values,
                       (long long)count, beta, thresholds4, keys_float, keys_compare);
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int ldpc_debug_workspace_layout(const ldpc_decoder *d, int64_t batch, int64_t out8[8])
{
    if (!d || !out8 || batch <= 0) return fail(LDPC_ERR_ARG, "bad argument");
    if (use_resident(d) || use_layered_lds(d)) return fail(LDPC_ERR_ARG, "workspace layout exists only in LDPC_MODE_STREAM");
    char *base = reinterpret_cast<char *>(kAlign);   // any non-null base: only differences are used
    const Workspace w = carve(d, batch, base);
    out8[0] = w.vec; out8[1] = w.tiles;
    // fused RCQ form: the codes ping-pong between the two buffers, iteration T-1 leaves them in buffer (T-1) & 1
    const char *c2v_final = (use_gather(d) && d->T > 0 && ((d->T - 1) & 1)) ? w.v2c : w.c2v;
    out8[2] = w.llrT - base; out8[3] = w.v2c - base; out8[4] = c2v_final - base; out8[5] = w.postT - base;
    out8[6] = (char *)w.bitsT - base; out8[7] = (char *)w.done - base;
    return LDPC_OK;
}

int ldpc_debug_resident_c2v(const ldpc_decoder *d, const void *llr, int64_t batch, int32_t early_stop,
                            void *posterior, int32_t *iterations, void *c2v_out, void *stream)
{
    if (!d || !llr || !posterior || !c2v_out || batch <= 0) return fail(LDPC_ERR_ARG, "bad argument");
    if (!use_resident(d)) return fail(LDPC_ERR_ARG, "the decoder is not on the LDS-resident engine");
    if (d->g->n == 0) return LDPC_OK;
    return decode_resident(d, llr, batch, early_stop, nullptr, posterior, iterations, nullptr, nullptr, c2v_out, stream);
}

int ldpc_debug_sweep(const ldpc_decoder *d, int64_t batch, int32_t which, int32_t iter, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    if (!d || !workspace || batch <= 0) return fail(LDPC_ERR_ARG, "bad argument");
    if (iter < 0 || iter >= d->T) return fail(LDPC_ERR_ARG, "iter outside [0, T)");
    if (use_resident(d) || use_layered_lds(d)) return fail(LDPC_ERR_ARG, "debug sweeps need LDPC_MODE_STREAM");
    const Workspace w = carve(d, batch, workspace);
    if (w.total > workspace_bytes) return fail(LDPC_ERR_WORKSPACE, "workspace too small");
    DeviceGuard guard(d->g->device);
    hipStream_t s = (hipStream_t)stream;
    const bool f64 = d->dtype == LDPC_F64;
    if (use_pair(d) && !f64) {
        // code-pair form: which = 0 the check sweep (integer-only from iteration 1), which = 1 the variable sweep
        // (code-producing except in the last iteration)
        if (which == 0 && (iter >= 1 || (w.vec == 4 && pair_q4<4>(d))))
            return w.vec == 1 ? launch_cn_q<1>(d, w, iter, false, s) : launch_cn_q<4>(d, w, iter, false, s);
        if (which != 0 && iter < d->T - 1) return w.vec == 1 ? launch_vn_q<1>(d, w, iter, false, s) : launch_vn_q<4>(d, w, iter, false, s);
    }
    if (use_gather(d) && !f64) {
        // fused RCQ form: which = 0 is the fused iteration kernel (iter >= 1; iteration 0 is the plain check sweep),
        // which = 1 the posterior-only variable pass
        char *buf[2] = {w.c2v, w.v2c};
        if (which == 0 && iter >= 1)
            return w.vec == 1 ? launch_gather<1>(d, w, iter, false, buf[(iter - 1) & 1], buf[iter & 1], s)
                              : launch_gather<4>(d, w, iter, false, buf[(iter - 1) & 1], buf[iter & 1], s);
        if (which != 0) {
            Workspace wv = w;
            wv.c2v = buf[iter & 1];
            return w.vec == 1 ? launch_vn<float, 1>(d, wv, iter, true, false, s) : launch_vn<float, 4>(d, wv, iter, true, false, s);
        }
    }
#define LDPC_DBG(REAL_, V_)                                                                 \
    return which == 0 ? launch_cn<REAL_, V_>(d, w, iter, false, s)                           \
                      : launch_vn<REAL_, V_>(d, w, iter, iter == d->T - 1, false, s)
    if (f64) {
        if (w.vec == 1) { LDPC_DBG(double, 1); } else { LDPC_DBG(double, 2); }
    } else {
        if (w.vec == 1) { LDPC_DBG(float, 1); } else { LDPC_DBG(float, 4); }
    }
#undef LDPC_DBG
}

}  // extern "C"
