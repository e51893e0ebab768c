// valu_issue_probe.hip -- ISSUE cost of independent vector instructions on gfx950 when 1, 2, 4 or 8 waves share a SIMD
// (one workgroup of 4 * w waves on one CU, every wave runs the same stream of independent instructions; s_memtime).
// Answers what "VALU issue bound" means for the kernels here: cycles a SIMD needs per wave-instruction of each kind,
// at the occupancies the kernels run at (the single-wave figures are in valu_latency_probe.hip).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_issue_probe tools/probes/valu_issue_probe.hip && tools/probes/valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

constexpr int kIters = 512;
// sixteen independent instructions: four destinations in rotation, sources never written inside the block
#define B4(I) I("%0") I("%1") I("%2") I("%3")
#define B16(I) B4(I) B4(I) B4(I) B4(I)
#define RUN(I)                                                                                         \
    for (int i = 0; i < kIters; ++i)                                                                   \
        asm volatile(B16(I) "" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x), "v"(y), "s"(sx) : "vcc", "s20", "s21", "s22");
#define RUN2(I)                                                                                        \
    for (int i = 0; i < kIters; ++i)                                                                   \
        asm volatile(B16(I) "" : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(px), "v"(py) : "vcc");

#define I_FMA(d) "v_fma_f32 " d ", %4, %5, %5\n\t"
#define I_ADD(d) "v_add_f32 " d ", %4, %5\n\t"
#define I_MULABS(d) "v_mul_f32_e64 " d ", %6, |%5|\n\t"
#define I_MINU(d) "v_min_u32 " d ", %4, %5\n\t"
#define I_MED3(d) "v_med3_f32 " d ", |%4|, %5, %5\n\t"
#define I_PERM(d) "v_perm_b32 " d ", %4, %5, %5\n\t"
#define I_BITOP(d) "v_bitop3_b32 " d ", %4, %5, %5 bitop3:0x78\n\t"
#define I_CMPADDC(d) "v_cmp_le_u32 vcc, %4, %5\n\tv_addc_co_u32 " d ", vcc, 0, %5, vcc\n\t"
#define I_CMP(d) "v_cmp_le_u32_e64 s[20:21], %4, %5\n\t"
#define I_CNDMASK(d) "v_cndmask_b32 " d ", %4, %5, vcc\n\t"
#define I_CVTU8(d) "v_cvt_pk_u8_f32 " d ", %4, 1, %5\n\t"
#define I_DPP(d) "v_min_u32_dpp " d ", %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_BFE(d) "v_bfe_u32 " d ", %4, 6, 8\n\t"
#define I_OR3(d) "v_or3_b32 " d ", %4, %5, %5\n\t"
#define I_LSHLOR(d) "v_lshl_or_b32 " d ", %4, 8, %5\n\t"
#define I_PKFMA(d) "v_pk_fma_f32 " d ", %4, %5, %5\n\t"
#define I_PKADD(d) "v_pk_add_f32 " d ", %4, %5\n\t"
#define I_PKMUL(d) "v_pk_mul_f32 " d ", %4, %5\n\t"
#define I_PKFMAC(d) "v_pk_fma_f32 " d ", %4, %5, %5 clamp\n\t"
#define I_ADD64(d) "v_lshl_add_u64 " d ", %4, 0, %5\n\t"
#define I_PKMINU16(d) "v_pk_min_u16 " d ", %4, %5\n\t"
#define I_CND64(d) "v_cndmask_b32_e64 " d ", %4, %5, s[20:21]\n\t"
#define I_CNDK(d) "v_cndmask_b32_e64 " d ", 0, 1, vcc\n\t"
#define I_CMP3CND(d) "v_cmp_lt_u32 vcc, %4, %5\n\tv_mov_b32 " d ", %4\n\tv_mov_b32 " d ", %5\n\tv_cndmask_b32 " d ", %4, %5, vcc\n\t"
#define I_ADDC(d) "v_addc_co_u32 " d ", vcc, 0, %5, vcc\n\t"
#define I_MAXF(d) "v_max_f32 " d ", %4, %5\n\t"
#define I_MULF(d) "v_mul_f32 " d ", %4, %5\n\t"
#define I_XOR(d) "v_xor_b32 " d ", %4, %5\n\t"
#define I_AND(d) "v_and_b32 " d ", %4, %5\n\t"
#define I_LSHL(d) "v_lshlrev_b32 " d ", 3, %5\n\t"
#define I_ADDU(d) "v_add_u32 " d ", %4, %5\n\t"
#define I_MOV(d) "v_mov_b32 " d ", %5\n\t"
#define I_CVT(d) "v_cvt_u32_f32 " d ", %5\n\t"
#define I_RDLANE(d) "v_readlane_b32 s22, %5, 3\n\t"
#define I_MINF(d) "v_min_f32 " d ", %4, %5\n\t"
#define I_MAX3(d) "v_max3_f32 " d ", %4, %5, %5\n\t"
#define I_MEDU(d) "v_med3_u32 " d ", %4, %5, %5\n\t"
#define I_MINUE64(d) "v_min_u32_e64 " d ", %4, %6\n\t"
#define I_FMAAC(d) "v_fma_f32 " d ", |%4|, %6, -%5 clamp\n\t"
#define I_FMAA(d) "v_fma_f32 " d ", |%4|, %5, %5\n\t"
#define I_MULC(d) "v_mul_f32_e64 " d ", %4, %5 clamp\n\t"
#define I_MULS(d) "v_mul_f32_e64 " d ", %6, %5\n\t"
#define I_ADDA(d) "v_add_f32_e64 " d ", |%4|, %5\n\t"
#define I_OR(d) "v_or_b32 " d ", %4, %5\n\t"
#define I_SUBU(d) "v_sub_u32 " d ", %4, %5\n\t"
#define I_SUBF(d) "v_sub_f32 " d ", %4, %5\n\t"
#define I_FMAK(d) "v_fmac_f32 " d ", %4, %5\n\t"
#define I_ADDS(d) "v_add_f32 " d ", %6, %5\n\t"
#define I_ANDL(d) "v_and_b32 " d ", 0x40404040, %5\n\t"
#define I_ANDI(d) "v_and_b32 " d ", 15, %5\n\t"
#define I_XORS(d) "v_xor_b32 " d ", %6, %5\n\t"
#define I_FMAI(d) "v_fma_f32 " d ", |%4|, 2.0, -%5 clamp\n\t"
#define I_BITOPS(d) "v_bitop3_b32 " d ", %4, %5, %6 bitop3:0x78\n\t"
#define I_MULI(d) "v_mul_f32 " d ", 2.0, %5\n\t"
#define I_ADD3(d) "v_add3_u32 " d ", %4, %5, %5\n\t"

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned *out, long long *cyc, int variant)
{
    unsigned x = threadIdx.x * 2654435761u | 1u, y = x ^ 0x5bd1e995u, sx = 0x3f000000u;
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    f2 px = {1.0f + threadIdx.x, 2.0f}, py = {0.5f, 0.25f}, p0 = px, p1 = px, p2 = px, p3 = px;
    __syncthreads();
    long long t0 = clock64();
    switch (variant) {
    case 0: RUN(I_FMA) break;
    case 1: RUN(I_ADD) break;
    case 2: RUN(I_MULABS) break;
    case 3: RUN(I_MINU) break;
    case 4: RUN(I_MED3) break;
    case 5: RUN(I_PERM) break;
    case 6: RUN(I_BITOP) break;
    case 7: RUN(I_CMPADDC) break;
    case 8: RUN(I_CMP) break;
    case 9: RUN(I_CNDMASK) break;
    case 10: RUN(I_CVTU8) break;
    case 11: RUN(I_DPP) break;
    case 12: RUN(I_BFE) break;
    case 13: RUN(I_OR3) break;
    case 14: RUN(I_LSHLOR) break;
    case 15: RUN2(I_PKFMA) break;
    case 16: RUN2(I_PKADD) break;
    case 17: RUN2(I_PKMUL) break;
    case 18: RUN2(I_PKFMAC) break;
    case 19: RUN2(I_ADD64) break;
    case 20: RUN(I_PKMINU16) break;
    case 21: asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); RUN(I_CND64) break;
    case 22: RUN(I_CNDK) break;
    case 23: RUN(I_CMP3CND) break;
    case 24: RUN(I_ADDC) break;
    case 25: RUN(I_MAXF) break;
    case 26: RUN(I_MULF) break;
    case 27: RUN(I_XOR) break;
    case 28: RUN(I_AND) break;
    case 29: RUN(I_LSHL) break;
    case 30: RUN(I_ADDU) break;
    case 31: RUN(I_MOV) break;
    case 32: RUN(I_CVT) break;
    case 33: RUN(I_RDLANE) break;
    case 34: RUN(I_MINF) break;
    case 35: RUN(I_MAX3) break;
    case 36: RUN(I_MEDU) break;
    case 37: RUN(I_MINUE64) break;
    case 38: RUN(I_ADD3) break;
    case 39: RUN(I_FMAAC) break;
    case 40: RUN(I_FMAA) break;
    case 41: RUN(I_MULC) break;
    case 42: RUN(I_MULS) break;
    case 43: RUN(I_ADDA) break;
    case 44: RUN(I_OR) break;
    case 45: RUN(I_SUBU) break;
    case 46: RUN(I_SUBF) break;
    case 47: RUN(I_ADDS) break;
    case 48: RUN(I_ANDL) break;
    case 49: RUN(I_ANDI) break;
    case 50: RUN(I_XORS) break;
    case 51: RUN(I_FMAI) break;
    case 52: RUN(I_BITOPS) break;
    case 53: RUN(I_MULI) break;
    }
    long long t1 = clock64();
    __syncthreads();
    out[threadIdx.x] = r0 + r1 + r2 + r3 + (unsigned)(p0.x + p1.y + p2.x + p3.y);
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main()
{
    static const char *names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32 e64 |abs|", "v_min_u32", "v_med3_f32 |abs|", "v_perm_b32",
                                  "v_bitop3_b32", "v_cmp_le_u32 + v_addc_co_u32 (pair)", "v_cmp_le_u32_e64 -> sgpr", "v_cndmask_b32",
                                  "v_cvt_pk_u8_f32", "v_min_u32 dpp quad_perm", "v_bfe_u32", "v_or3_b32", "v_lshl_or_b32",
                                  "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32 clamp", "v_lshl_add_u64", "v_pk_min_u16",
                                  "v_cndmask_b32_e64 sgpr pair", "v_cndmask_b32 0,1,vcc", "v_cmp + 2 v_mov + v_cndmask (four)", "v_addc_co_u32 alone",
                                  "v_max_f32", "v_mul_f32 e32", "v_xor_b32", "v_and_b32", "v_lshlrev_b32", "v_add_u32", "v_mov_b32",
                                  "v_cvt_u32_f32", "v_readlane_b32", "v_min_f32", "v_max3_f32", "v_med3_u32", "v_min_u32_e64 sgpr", "v_add3_u32",
                                  "v_fma_f32 |a|, s, -c clamp", "v_fma_f32 |a|, b, c", "v_mul_f32_e64 clamp", "v_mul_f32_e64 sgpr, v", "v_add_f32_e64 |a|, b",
                                  "v_or_b32", "v_sub_u32", "v_sub_f32",
                                  "v_add_f32 sgpr, v", "v_and_b32 literal, v", "v_and_b32 inline, v", "v_xor_b32 sgpr, v", "v_fma_f32 |a|, 2.0, -c clamp",
                                  "v_bitop3_b32 v, v, sgpr", "v_mul_f32 2.0, v"};
    unsigned *out; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    int dev_khz = 0, wall_khz = 0;
    hipDeviceGetAttribute(&dev_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    const double tick = 1.0;      // clock64() advances once per shader clock on this device (a dependent v_min_u32 chain reads 9 per step)
    printf("device clock %d kHz, s_memtime %d kHz; SIMD cycles per wave-instruction (per pair where noted), w waves per SIMD\n", dev_khz, wall_khz);
    printf("%-40s %8s %8s %8s %8s\n", "instruction", "w=1", "w=2", "w=4", "w=8");
    for (int v = 0; v <= 53; ++v) {
        printf("%-40s", names[v]);
        for (int w : {1, 2, 4, 8}) {
            const int waves = 4 * w;
            if (waves * 64 > 1024) {     // 8 waves per SIMD = two 1024-thread workgroups on the CU: launch two blocks, hope they share a CU
                printf(" %8s", "-");
                continue;
            }
            long long h[64];
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(waves * 64), 0, 0, out, cyc, v);
                hipDeviceSynchronize();
                hipMemcpy(h, cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
                long long mx = 0;
                for (int i = 0; i < waves; ++i) mx = std::max(mx, h[i]);
                best = std::min(best, (double)mx);
            }
            printf(" %8.2f", best * tick / ((double)kIters * 16.0 * w));
        }
        printf("\n");
    }
    return 0;
}
