// Micro-probe: LDS throughput of the operations an atomics-based check-node accumulation would use
// (random addresses over a 9000-word table, 1024-thread workgroup, one workgroup per CU).
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int M = 9000;          // words in the table (checks of the (16200,7200) code)
constexpr int REPS = 2000;

template <int OP>
__global__ __launch_bounds__(1024) void probe(const unsigned* __restrict__ idx, unsigned* out, int per_thread)
{
    extern __shared__ unsigned tab[];
    for (int i = threadIdx.x; i < 3 * M + 8192; i += 1024) tab[i] = 0x7f800000u;
    __syncthreads();
    unsigned acc = 0;
    unsigned char* bytes = reinterpret_cast<unsigned char*>(tab + 3 * M);
    constexpr int PT = 48;
    unsigned ii[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) ii[k] = idx[(k * 1024 + threadIdx.x)];
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const unsigned i = ii[k];
            const unsigned v = (i * 2654435761u + r) & 0x7fffffffu;
            if (OP == 0) tab[i] = v;                                   // plain store
            if (OP == 1) acc += tab[i];                                // plain load
            if (OP == 2) atomicMin(&tab[i], v);                        // ds_min_u32 (no return)
            if (OP == 3) atomicXor(&tab[M + i], v & 0x80000000u);      // ds_xor_b32
            if (OP == 4) { atomicMin(&tab[i], v); atomicXor(&tab[M + i], v & 0x80000000u); }
            if (OP == 5) acc += atomicMin(&tab[i], v);                 // returning atomic
            if (OP == 6) bytes[(i * 5 + k) & 0x7fff] = (unsigned char)v;   // byte store
            if (OP == 7) acc += bytes[(i * 5 + k) & 0x7fff];               // byte load
            if (OP == 8) { const unsigned m1 = tab[i]; if (v > m1) atomicMin(&tab[2 * M + i], v); else atomicAdd(&tab[M + i], 1u); }
        }
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x + 1] = tab[(blockIdx.x * 7) % M];
}

template <int OP>
void run(const char* name, const unsigned* d_idx, unsigned* d_out, int per_thread, int blocks)
{
    const size_t lds = (3 * M + 8192) * sizeof(unsigned);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<OP><<<blocks, 1024, lds>>>(d_idx, d_out, per_thread);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<OP><<<blocks, 1024, lds>>>(d_idx, d_out, per_thread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double waveops = double(REPS) * per_thread * 16;            // per workgroup (= per CU)
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-34s %8.3f ms  %6.2f clk per wave-instruction per CU (%s)\n", name, ms, clk / waveops, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int per_thread = 48;                                        // ~48599 edges / 1024 lanes
    std::vector<unsigned> idx(per_thread * 1024);
    unsigned s = 12345;
    for (auto& x : idx) { s = s * 1664525u + 1013904223u; x = (s >> 8) % M; }
    unsigned *d_idx, *d_out;
    hipMalloc(&d_idx, idx.size() * 4); hipMalloc(&d_out, 4096 * 4);
    hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256;
    run<0>("ds_write_b32 random", d_idx, d_out, per_thread, blocks);
    run<1>("ds_read_b32 random", d_idx, d_out, per_thread, blocks);
    run<2>("ds_min_u32 (no return) random", d_idx, d_out, per_thread, blocks);
    run<3>("ds_xor_b32 (no return) random", d_idx, d_out, per_thread, blocks);
    run<4>("ds_min + ds_xor", d_idx, d_out, per_thread, blocks);
    run<5>("ds_min_rtn_u32 random", d_idx, d_out, per_thread, blocks);
    run<6>("ds_write_b8 random", d_idx, d_out, per_thread, blocks);
    run<7>("ds_read_u8 random", d_idx, d_out, per_thread, blocks);
    run<8>("read + (min | add) pass-B shape", d_idx, d_out, per_thread, blocks);
    return 0;
}
