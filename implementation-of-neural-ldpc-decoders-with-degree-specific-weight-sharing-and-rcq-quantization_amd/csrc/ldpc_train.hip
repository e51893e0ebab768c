// ldpc_train.hip -- gfx950 kernels of the gradient (training) path: d loss / d beta, d loss / d alpha of the
// normalised min-sum decoders, for a loss that is a function of the posterior the decode returned.
//
// What is differentiated (reference: neural_2d_decoder.py:133-225 run under torch autograd, which is what
// training_framework.py:127-134 does with it):
//   c2v_t[e]   = beta_t[slot(e)] * minval_t(e) * prod_{e' != e} sign(v2c_t[e'])           (:186-191)
//   v2c_t+1[e] = llr[j] + alpha_t[slot(j)] * sum_{e' != e at j} c2v_t[e']                  (:203)
//   posterior  = llr[j] + sum_{e at j} c2v_s[e]     at the iteration s the codeword stopped (:206-216)
// autograd's rules for those operations: d|x| = sign(x) with sign(0) = 0, d sign = 0, the minimum passes its
// gradient to the arg-min element (first index; `magnitudes[min_idx]`), the second minimum to the arg-min of
// the remaining elements, `min2_val = min_val` for a degree-1 check sends both to that one edge.
//
// Same mapping as the forward sweeps (ldpc_kernels.hip): one wave = one node x W codewords, message and
// gradient rows [tile][edge][W] streamed coalesced; the forward pass has saved every iteration's v2c and c2v
// rows (ldpc_decode_saving), nothing is recomputed but the per-check min/min2/sign state.  Table gradients are
// reduced over the W codewords of the wave (DPP/shuffle tree) into per-(tile, edge|variable) partials and
// summed over tiles and slots by reduce_table_grads in a fixed order: no atomics anywhere, the gradients are
// bit-identical from run to run.
//
// Ties follow autograd too: `torch.min(temp_mags)` (neural_2d_decoder.py:179) is a full reduction whose backward splits
// the gradient evenly among ALL elements that hold the minimum -- the second minimum's gradient is divided by the number
// of edges tied for it (the set is tracked as a bit mask during the scan); `magnitudes[min_idx]` sends the minimum's to
// the first arg-min edge.  Pinned by tests/golden/grad_ties.npz (the reference under autograd on half-integer LLRs).
//
// Deviation from torch autograd on the reference's graph (documented, pinned by tests/test_gpu_training.py):
//  * parameters that cannot influence the returned posterior (iterations after every codeword stopped) receive a
//    zero gradient where the reference leaves `.grad` None (an optimizer with momentum keeps moving those).
#pragma once

#include "ldpc_kernels.hip"

namespace ldpc {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// state of codeword (lane, c) at backward step t:  0 = takes no part (stopped before t, or padding),
// 1 = c2v_t feeds v2c_t+1 (gradient arrives from the variable pass), 2 = c2v_t feeds the returned posterior
template <int VEC>
__device__ __forceinline__ void load_states(const int *__restrict__ iterations, long long batch, int tile, int lane,
                                            int t, int (&state)[VEC])
{
    constexpr int W = kWave * VEC;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const long long b = (long long)tile * W + (long long)lane * VEC + c;
        const int stop = (b < batch) ? iterations[b] - 1 : -1;      // index of the iteration whose posterior was returned
        state[c] = t < stop ? 1 : (t == stop ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------
// Check-node backward of iteration t.  in: v2c_t rows (llr rows when FIRST), d loss/d c2v_t.
// out: d loss/d v2c_t rows (not when FIRST: v2c_0 = llr has no parameters upstream),
//      per-edge partial of d loss/d beta_t.
// ------------------------------------------------------------------------------------------
//
// FORM_NMS: c2v = beta * minval * prod(signs).   FORM_OMS: c2v = prod(signs) * (relu(minval - beta) - alpha_c)
// (neural_2d_decoder.py:389-401, neural_minsum_decoder.py:245-253): d/d beta = -g*ps*[minval > beta],
// d/d alpha_c = -g*ps (second per-edge partial, goa_part), d/d minval = g*ps*[minval > beta]; relu'(0) = 0.
template <int VEC, bool FIRST, int FORM>
__global__ __launch_bounds__(kBlock) void cn_backward(GraphDev g, const float *__restrict__ src,
                                                      const float *__restrict__ gc2v,
                                                      const float *__restrict__ gpostT,
                                                      const int *__restrict__ iterations, long long batch, int t,
                                                      const float *__restrict__ beta_row,
                                                      const int *__restrict__ beta_slot,
                                                      float *__restrict__ gv2c_out,
                                                      float *__restrict__ gbeta_part,
                                                      float *__restrict__ goa_part, int check_blocks)
{
    constexpr int W = kWave * VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / check_blocks);
    const int i = uni((blockIdx.x % check_blocks) * kWavesPerBlock + (threadIdx.x >> 6));
    if (i >= g.m) return;
    const int e0 = uni(g.check_ptr[i]);
    const int dc = uni(g.check_ptr[i + 1]) - e0;
    if (dc == 0) return;

    int state[VEC];
    load_states<VEC>(iterations, batch, tile, lane, t, state);
    bool mine = false, from_vn = false, from_post = false;
#pragma unroll
    for (int c = 0; c < VEC; ++c) { mine |= state[c] != 0; from_vn |= state[c] == 1; from_post |= state[c] == 2; }
    const bool any = __ballot(mine) != 0ull;
    const bool any_vn = __ballot(from_vn) != 0ull, any_post = __ballot(from_post) != 0ull;

    const size_t lane_off = (size_t)lane * VEC;
    const size_t erow = ((size_t)tile * g.E + e0) * W + lane_off;
    float *out_base = gv2c_out ? gv2c_out + erow : nullptr;      // FIRST: only when d loss/d llr is wanted
    if (!any) {                                   // nothing of this wave is live: the consumers still read zeros
        Pack<float, VEC> z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) z.x[c] = 0.0f;
        for (int u = 0; u < dc; ++u) {
            if (out_base) st<float, VEC>(out_base + (size_t)u * W, z);
            if (lane == 0) {
                gbeta_part[(size_t)tile * g.E + e0 + u] = 0.0f;
                if (FORM == FORM_OMS && goa_part) goa_part[(size_t)tile * g.E + e0 + u] = 0.0f;
            }
        }
        return;
    }

    const float *in_base = FIRST ? src + (size_t)tile * g.n * W + lane_off : src + erow;
    auto in_row = [&](int u) { return FIRST ? in_base + (size_t)g.var_idx[e0 + u] * W : in_base + (size_t)u * W; };

    float m1[VEC], m2[VEC];
    int idx[VEC], k2[VEC];          // arg-min edge (first index); number of edges tied for the second minimum
    uint32_t sm[VEC], zm[VEC], tm[VEC];   // sign bits, zero flags, and the edges tied for the second minimum (bit u & 31)
    unsigned par[VEC], nz[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        m1[c] = inf_of<float>(); m2[c] = inf_of<float>(); idx[c] = 0; k2[c] = 0; sm[c] = 0; zm[c] = 0; tm[c] = 0; par[c] = 0; nz[c] = 0;
    }
#pragma unroll 4
    for (int u = 0; u < dc; ++u) {
        const Pack<float, VEC> v = ld<float, VEC>(in_row(u));
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const float a = __builtin_fabsf(v.x[c]);
            const unsigned sb = signbit_of<float>(v.x[c]);
            const unsigned z = (a == 0.0f) ? 1u : 0u;
            par[c] ^= sb; nz[c] += z;
            sm[c] |= sb << (u & 31); zm[c] |= z << (u & 31);
            const uint32_t ub = 1u << (u & 31);
            if (u == 0) {
                m1[c] = a;                                                    // idx = 0; no second minimum yet (k2 == 0)
            } else if (a < m1[c]) {
                // the old minimum joins the second-minimum set: alone if it lies below the old second, else as one more tie
                if (m1[c] < m2[c] || k2[c] == 0) { m2[c] = m1[c]; tm[c] = 1u << (idx[c] & 31); k2[c] = 1; }
                else { tm[c] |= 1u << (idx[c] & 31); k2[c] += 1; }
                m1[c] = a; idx[c] = u;
            } else if (a < m2[c] || k2[c] == 0) {
                m2[c] = a; tm[c] = ub; k2[c] = 1;
            } else if (a == m2[c]) {
                tm[c] |= ub; k2[c] += 1;
            }
        }
    }
    if (dc == 1) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) { m2[c] = m1[c]; tm[c] = 1u; k2[c] = 1; }   // min2_val = min_val (:181-182): both to that edge
    }
    const bool wide = dc > 32;                     // the sign / zero masks hold 32 edges; wider checks re-read

    float acc1[VEC], acc2[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { acc1[c] = 0.0f; acc2[c] = 0.0f; }
    for (int u = 0; u < dc; ++u) {
        const float b = beta_row[beta_slot[e0 + u]];
        Pack<float, VEC> gv, gp, re;
#pragma unroll
        for (int c = 0; c < VEC; ++c) { gv.x[c] = 0.0f; gp.x[c] = 0.0f; re.x[c] = 0.0f; }
        if (any_vn) gv = ld<float, VEC>(gc2v + erow + (size_t)u * W);
        if (any_post) gp = ld<float, VEC>(gpostT + ((size_t)tile * g.n + g.var_idx[e0 + u]) * W + lane_off);
        if (wide) re = ld<float, VEC>(in_row(u));
        float gb = 0.0f, goa = 0.0f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const unsigned own = wide ? signbit_of<float>(re.x[c]) : ((sm[c] >> (u & 31)) & 1u);
            const unsigned ownz = wide ? ((re.x[c] == 0.0f) ? 1u : 0u) : ((zm[c] >> (u & 31)) & 1u);
            // product of the OTHER signs, sign(0) = 0
            const float ps = (nz[c] - ownz) != 0 ? 0.0f : ((par[c] ^ own) ? -1.0f : 1.0f);
            const float gin = state[c] == 1 ? gv.x[c] : (state[c] == 2 ? gp.x[c] : 0.0f);
            const float minval = (u == idx[c]) ? m2[c] : m1[c];
            const float gps = gin * ps;
            if (state[c] != 0) {
                float gm;
                if (FORM == FORM_OMS) {
                    const float open = (minval - b) > 0.0f ? gps : 0.0f;      // relu gate
                    gb -= open;
                    goa -= gps;
                    gm = open;
                } else {
                    gb += gps * minval;
                    gm = gps * b;
                }
                if (u == idx[c]) acc2[c] += gm; else acc1[c] += gm;
            }
        }
        gb = wave_sum(gb);
        if (FORM == FORM_OMS && goa_part) goa = wave_sum(goa);
        if (lane == 0) {
            gbeta_part[(size_t)tile * g.E + e0 + u] = gb;
            if (FORM == FORM_OMS && goa_part) goa_part[(size_t)tile * g.E + e0 + u] = goa;
        }
    }
    if (!out_base) return;
    for (int u = 0; u < dc; ++u) {
        Pack<float, VEC> re, o;
#pragma unroll
        for (int c = 0; c < VEC; ++c) re.x[c] = 0.0f;
        if (wide) re = ld<float, VEC>(in_row(u));
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const unsigned own = wide ? signbit_of<float>(re.x[c]) : ((sm[c] >> (u & 31)) & 1u);
            const unsigned ownz = wide ? ((re.x[c] == 0.0f) ? 1u : 0u) : ((zm[c] >> (u & 31)) & 1u);
            const float sg = ownz ? 0.0f : (own ? -1.0f : 1.0f);                           // d|x|/dx
            // member of the second-minimum set: the mask bit (<= 32 edges), else by value (the arg-min edge is not in it)
            const bool tied = wide ? (u != idx[c] && __builtin_fabsf(re.x[c]) == m2[c]) : (((tm[c] >> (u & 31)) & 1u) != 0);
            const float gmag = ((u == idx[c]) ? acc1[c] : 0.0f) + (tied ? acc2[c] / (float)k2[c] : 0.0f);
            o.x[c] = state[c] != 0 ? gmag * sg : 0.0f;
        }
        st<float, VEC>(out_base + (size_t)u * W, o);
    }
}

// ------------------------------------------------------------------------------------------
// Variable-node backward of the update that produced v2c_t (t >= 1) from c2v_t-1 with alpha_t-1.
// in: c2v_t-1 rows, d loss/d v2c_t rows.  out: d loss/d c2v_t-1 rows, per-variable partial of
// d loss/d alpha_t-1.  Leave-one-out sums are formed as (total - own) with the totals in fp64.
// ------------------------------------------------------------------------------------------
constexpr int kVnbVarsPerWave = 4;

template <int VEC>
__global__ __launch_bounds__(kBlock) void vn_backward(GraphDev g, const float *__restrict__ c2v_prev,
                                                      const float *__restrict__ gv2c,
                                                      const int *__restrict__ iterations, long long batch, int t,
                                                      const float *__restrict__ alpha_row,
                                                      const int *__restrict__ alpha_slot,
                                                      float *__restrict__ gc2v_out,
                                                      float *__restrict__ galpha_part, int var_blocks)
{
    constexpr int W = kWave * VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / var_blocks);
    const int jbase = uni(((blockIdx.x % var_blocks) * kWavesPerBlock + (threadIdx.x >> 6)) * kVnbVarsPerWave);
    if (jbase >= g.n) return;
    int state[VEC];
    load_states<VEC>(iterations, batch, tile, lane, t, state);     // v2c_t exists for codewords with state != 0
    bool mine = false;
#pragma unroll
    for (int c = 0; c < VEC; ++c) mine |= state[c] != 0;
    const bool any = __ballot(mine) != 0ull;
    const size_t base = (size_t)tile * g.E * W + (size_t)lane * VEC;
    for (int vv = 0; vv < kVnbVarsPerWave; ++vv) {       // a wave's work per variable is small (degree 2-3): several per wave
    const int j = jbase + vv;
    if (j >= g.n) break;
    const int k0 = uni(g.var_ptr[j]);
    const int dv = uni(g.var_ptr[j + 1]) - k0;
    if (dv == 0) {
        if (lane == 0) galpha_part[(size_t)tile * g.n + j] = 0.0f;
        continue;
    }
    if (!any) {
        Pack<float, VEC> z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) z.x[c] = 0.0f;
        for (int k = 0; k < dv; ++k) st<float, VEC>(gc2v_out + base + (size_t)g.csc_edge[k0 + k] * W, z);
        if (lane == 0) galpha_part[(size_t)tile * g.n + j] = 0.0f;
        continue;
    }
    const float alpha = alpha_row[alpha_slot[j]];
    double totc[VEC], totg[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { totc[c] = 0.0; totg[c] = 0.0; }
    float ga = 0.0f;
    constexpr int kHeld = 8;                       // variables up to this degree keep their rows in registers
    if (dv <= kHeld) {
        Pack<float, VEC> cv[kHeld], gv[kHeld];
        size_t row[kHeld];
#pragma unroll
        for (int k = 0; k < kHeld; ++k) {
            if (k < dv) {
                row[k] = base + (size_t)g.csc_edge[k0 + k] * W;
                cv[k] = ld<float, VEC>(c2v_prev + row[k]);
                gv[k] = ld<float, VEC>(gv2c + row[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < kHeld; ++k) {
            if (k < dv) {
#pragma unroll
                for (int c = 0; c < VEC; ++c)
                    if (state[c] != 0) { totc[c] += (double)cv[k].x[c]; totg[c] += (double)gv[k].x[c]; }
            }
        }
#pragma unroll
        for (int k = 0; k < kHeld; ++k) {
            if (k < dv) {
                Pack<float, VEC> o;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    o.x[c] = 0.0f;
                    if (state[c] != 0) {
                        ga += gv[k].x[c] * (float)(totc[c] - (double)cv[k].x[c]);
                        o.x[c] = alpha * (float)(totg[c] - (double)gv[k].x[c]);
                    }
                }
                st<float, VEC>(gc2v_out + row[k], o);
            }
        }
        ga = wave_sum(ga);
        if (lane == 0) galpha_part[(size_t)tile * g.n + j] = ga;
        continue;
    }
#pragma unroll 4
    for (int k = 0; k < dv; ++k) {
        const size_t row = base + (size_t)g.csc_edge[k0 + k] * W;
        const Pack<float, VEC> cv = ld<float, VEC>(c2v_prev + row);
        const Pack<float, VEC> gv = ld<float, VEC>(gv2c + row);
#pragma unroll
        for (int c = 0; c < VEC; ++c)
            if (state[c] != 0) { totc[c] += (double)cv.x[c]; totg[c] += (double)gv.x[c]; }
    }
#pragma unroll 4
    for (int k = 0; k < dv; ++k) {
        const size_t row = base + (size_t)g.csc_edge[k0 + k] * W;
        const Pack<float, VEC> cv = ld<float, VEC>(c2v_prev + row);
        const Pack<float, VEC> gv = ld<float, VEC>(gv2c + row);
        Pack<float, VEC> o;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            o.x[c] = 0.0f;
            if (state[c] != 0) {
                ga += gv.x[c] * (float)(totc[c] - (double)cv.x[c]);          // d v2c[e] / d alpha = sum of the others
                o.x[c] = alpha * (float)(totg[c] - (double)gv.x[c]);         // c2v[e] feeds every OTHER edge of j
            }
        }
        st<float, VEC>(gc2v_out + row, o);
    }
    ga = wave_sum(ga);
    if (lane == 0) galpha_part[(size_t)tile * g.n + j] = ga;
    }   // variables of this wave
}

// d loss/d llr: the LLR of variable j enters the returned posterior directly and every v2c message of j
// (v2c_0 = llr, v2c_t = llr + alpha * sum(...)), so  g_llr[j] = g_post[j] + sum over iterations and edges of g_v2c.
// One wave = one variable x W codewords; called once per backward step with that step's g_v2c rows (zero for
// codewords that take no part), accumulating in place in gllrT [tile][n][W] (initialised with g_post).
template <int VEC>
__global__ __launch_bounds__(kBlock) void llr_backward_accumulate(GraphDev g, const float *__restrict__ gv2c,
                                                                  float *__restrict__ gllrT, int var_blocks)
{
    constexpr int W = kWave * VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int tile = uni(blockIdx.x / var_blocks);
    const int j = uni((blockIdx.x % var_blocks) * kWavesPerBlock + (threadIdx.x >> 6));
    if (j >= g.n) return;
    const int k0 = uni(g.var_ptr[j]);
    const int dv = uni(g.var_ptr[j + 1]) - k0;
    if (dv == 0) return;
    const size_t base = (size_t)tile * g.E * W + (size_t)lane * VEC;
    float *acc_row = gllrT + ((size_t)tile * g.n + j) * W + (size_t)lane * VEC;
    Pack<float, VEC> acc = ld<float, VEC>(acc_row);
#pragma unroll 4
    for (int k = 0; k < dv; ++k) {
        const Pack<float, VEC> gv = ld<float, VEC>(gv2c + base + (size_t)g.csc_edge[k0 + k] * W);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc.x[c] += gv.x[c];
    }
    st<float, VEC>(acc_row, acc);
}

// grad_table[t][s] = sum over the items x of slot s (x = edge for beta, variable for alpha) and over the tiles of
// part[t][tile][x].  One wave per (slot, iteration), grid = (n_slots, T): lane l takes the items l, l+64, ... of the
// slot's list (built by the host: slot_ptr / slot_items, the inverse of the slot map) in that order, sums in double,
// and the 64 lane sums are combined by a fixed butterfly -- no atomics, so the result is bit-identical from run to
// run.  Slots without items (and slots whose partials are all zero) come out as exactly 0.
__global__ __launch_bounds__(kWave) void reduce_table_grads(const float *__restrict__ part, int tiles, int count,
                                                            const int *__restrict__ slot_ptr,
                                                            const int *__restrict__ slot_items, int n_slots,
                                                            float *__restrict__ grad)
{
    const int s = blockIdx.x, t = blockIdx.y;
    const int i0 = slot_ptr[s], i1 = slot_ptr[s + 1];
    const float *p = part + ((size_t)t * tiles) * count;
    double acc = 0.0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += kWave) {
        const float *q = p + slot_items[i];
        double a = 0.0;
        for (int k = 0; k < tiles; ++k) a += (double)q[(size_t)k * count];
        acc += a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if (threadIdx.x == 0) grad[(size_t)t * n_slots + s] = (float)acc;
}

}  // namespace ldpc
