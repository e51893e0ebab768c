// valu_latency_probe.hip -- cycles per instruction of dependent chains on gfx950, one wave on an idle CU (s_memtime):
// plain VALU, VALU with a DPP operand, v_cmp -> v_cndmask pairs, s_nop, and an LDS read-after-read pointer chase.
// Guides the layered kernel (ldpc_layered.hip), whose per-check step is one dependent chain of ~45 instructions.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_latency_probe tools/probes/valu_latency_probe.hip && tools/probes/valu_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int kIters = 256;

__global__ void probe(unsigned *out, long long *cyc, int variant, const unsigned *chase)
{
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned)((i * 67 + 4 * 33) % 4096) * 4u;   // byte offsets, a permutation-ish chase
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u, y = x ^ 0x5bd1e995u;
    long long t0 = clock64();
    if (variant == 0) {           // dependent plain VALU
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(y));) }
    } else if (variant == 1) {    // dependent VALU with DPP operand (quad_perm), the compiler-visible hazard handled by s_nop 1
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));) }
    } else if (variant == 2) {    // the same with row_mirror
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));) }
    } else if (variant == 3) {    // v_cmp -> s_nop 1 -> v_cndmask
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");) }
    } else if (variant == 4) {    // s_nop 0 alone
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_nop 0");) }
    } else if (variant == 5) {    // independent VALU (two chains interleaved)
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("v_min_u32 %0, %0, %2\n\tv_max_u32 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(threadIdx.x));) }
    } else if (variant == 6) {    // LDS pointer chase: ds_read -> wait -> ds_read
        unsigned a = (threadIdx.x * 4u) & 16383u;
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a));) }
        x = a;
    } else if (variant == 7) {    // DPP mov then plain op (mov_dpp + min), as the compiler emits for a two-use exchange
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_min_u32 %0, %0, %1" : "+v"(x), "+v"(y));) }
    } else if (variant == 8) {    // ds_write then dependent ds_read of the same address (in-order LDS): write->read round trip
        unsigned a = (threadIdx.x * 4u) & 16383u;
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(a));) }
    }
    else if (variant == 9) {      // a TAKEN scalar branch per unit (skipping two instructions), as the per-step "not LATE" jump was
        for (int i = 0; i < kIters; ++i) { REP16(asm volatile("s_cmp_eq_u32 %1, %1\n\ts_cbranch_scc1 1f\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n1:" : "+v"(x) : "s"(variant));) }
    } else if (variant == 10) {   // dependent global loads, 64 KB footprint (L2 resident, beyond the 32 KB L1)
        unsigned a = threadIdx.x & 15u;
        for (int i = 0; i < kIters; ++i) { REP16(a = chase[a];) }
        x = a;
    } else if (variant == 11) {   // dependent global loads, 4 KB footprint (L1 resident)
        unsigned a = threadIdx.x & 15u;
        for (int i = 0; i < kIters; ++i) { REP16(a = chase[a] & 1023u;) }
        x = a;
    }
    else if (variant == 12) {     // THROUGHPUT of independent plain VALU: 16 different destinations, one source
        unsigned r0, r1, r2, r3;
        for (int i = 0; i < kIters; ++i) {
            asm volatile("v_min_u32 %0, %4, %5\n\tv_max_u32 %1, %4, %5\n\tv_min_u32 %2, %4, %5\n\tv_max_u32 %3, %4, %5\n\t"
                         "v_min_u32 %0, %4, %5\n\tv_max_u32 %1, %4, %5\n\tv_min_u32 %2, %4, %5\n\tv_max_u32 %3, %4, %5\n\t"
                         "v_min_u32 %0, %4, %5\n\tv_max_u32 %1, %4, %5\n\tv_min_u32 %2, %4, %5\n\tv_max_u32 %3, %4, %5\n\t"
                         "v_min_u32 %0, %4, %5\n\tv_max_u32 %1, %4, %5\n\tv_min_u32 %2, %4, %5\n\tv_max_u32 %3, %4, %5"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x), "v"(y));
        }
        x = r0 + r1 + r2 + r3;
    } else if (variant == 13) {   // THROUGHPUT of independent VALU with a DPP operand: the same, src0 through quad_perm
        unsigned r0, r1, r2, r3;
        for (int i = 0; i < kIters; ++i) {
#define D4 "v_min_u32_dpp %0, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
           "v_max_u32_dpp %1, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
           "v_min_u32_dpp %2, %4, %5 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
           "v_max_u32_dpp %3, %4, %5 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            asm volatile(D4 D4 D4 D4 "s_nop 0" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x), "v"(y));
#undef D4
        }
        x = r0 + r1 + r2 + r3;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) cyc[variant] = t1 - t0;
}

int main()
{
    unsigned *out; long long *cyc; unsigned *chase;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16 * 8); hipMalloc(&chase, 16384 * 4);
    {
        unsigned h[16384];
        for (int i = 0; i < 16384; ++i) h[i] = (unsigned)((i * 4099 + 64) % 16384);   // stride walk through 64 KB
        hipMemcpy(chase, h, sizeof(h), hipMemcpyHostToDevice);
    }
    const char *names[] = {"dependent v_min_u32", "dependent s_nop1 + v_min_u32_dpp quad_perm", "dependent s_nop1 + v_min_u32_dpp row_mirror",
                           "v_cmp + s_nop1 + v_cndmask", "s_nop 0", "two independent VALU chains (per pair)", "LDS pointer chase (ds_read + wait)",
                           "s_nop1 + v_mov_dpp + v_min (per triple)", "ds_write + ds_read same address + wait",
                           "taken s_cbranch (cmp + branch over 2 instr)", "global load chase, 64 KB (L2)", "global load chase, 4 KB (L1)",
                           "independent plain VALU (per instruction)", "independent VALU with DPP operand (per instruction)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 14; ++v) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, v, chase);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, cyc + v, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-52s %8.2f clock64 ticks per unit\n", names[v], (double)c / (kIters * 16));
        }
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    int wkhz = 0; hipDeviceGetAttribute(&wkhz, hipDeviceAttributeWallClockRate, 0);
    printf("device clock %d kHz, wall clock rate %d kHz (clock64 = s_memtime)\n", khz, wkhz);
    return 0;
}
