// tools/bench_gather_policy.hip — rate of random 8-byte gathers from an L2-resident table (2 MiB: one fp16 level of the hash grid) under
// the cache-policy bits of global_load (sc0 / sc1 / nt): does any of them lift the ~0.44 lane-loads per clock and CU the encoder's
// fine levels run at?  hipcc --offload-arch=gfx950 -O3 -o tools/bench_gather_policy tools/bench_gather_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int POLICY>
__device__ __forceinline__ uint2 ld8(const void *p) {
    uint2 v;
    if constexpr (POLICY == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 6) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 7) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POLICY, int BYTES>
__global__ void __launch_bounds__(256) k_gather(const char *__restrict__ table, uint32_t mask, uint32_t rounds, uint32_t *__restrict__ out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t row = (x >> 8) & mask;
            v[j] = ld8<POLICY>(table + (uint64_t)row * BYTES);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int POLICY>
static void run(const char *name, const char *table, uint32_t rows, uint32_t *out) {
    const uint32_t blocks = 256 * 16, rounds = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<POLICY, 8>), dim3(blocks), dim3(256), 0, 0, table, rows - 1, rounds, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_gather<POLICY, 8>), dim3(blocks), dim3(256), 0, 0, table, rows - 1, rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double loads = 5.0 * blocks * 256.0 * rounds * 8;
    printf("%-14s rows %8u (%6.1f MiB): %7.1f G lane-loads/s  = %.3f per clock and CU (2.4 GHz, 256 CUs)\n", name, rows, rows * 8.0 / 1048576, loads / ms * 1e-6,
           loads / (ms * 1e-3) / (2.4e9 * 256));
}

int main() {
    for (uint32_t rows : {1u << 18, 1u << 21, 1u << 24}) {       // 2 MiB (one level, L2-resident), 16 MiB (all levels), 128 MiB (Infinity Cache)
        char *table; uint32_t *out;
        hipMalloc(&table, (size_t)rows * 8); hipMalloc(&out, 64);
        hipMemset(table, 1, (size_t)rows * 8);
        run<0>("plain", table, rows, out);
        run<1>("nt", table, rows, out);
        run<2>("sc0", table, rows, out);
        run<3>("sc1", table, rows, out);
        run<4>("sc0 sc1", table, rows, out);
        run<5>("sc0 sc1 nt", table, rows, out);
        run<6>("sc0 nt", table, rows, out);
        run<7>("sc1 nt", table, rows, out);
        hipFree(table); hipFree(out);
    }
    return 0;
}
