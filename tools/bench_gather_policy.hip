// Does a cache-policy modifier change the divergent-gather rate on gfx950?  (companion of bench_gather.hip)
// Every lane of a wave reads a different 128-byte line of an L2-resident table (2 MiB) or of one that only fits the Infinity Cache (48 MiB);
// the load carries no modifier, sc0, sc1, nt, or combinations. Build: hipcc -O3 --offload-arch=gfx950 tools/bench_gather_policy.hip -o tools/bench_gather_policy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int POL> __device__ __forceinline__ uint32_t ld(const char *p) {
    uint32_t v;
    if (POL == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 1) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 3) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 5) asm volatile("global_load_dword %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 6) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, uint32_t mask16, uint32_t iters, uint32_t *__restrict__ out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    const uint32_t key = tid * 0x9e3779b9u;
    for (uint32_t it = 0; it < iters; it += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ld<POL>(reinterpret_cast<const char *>(table + (mix(key + (it + u) * 0x85ebca6bu) & mask16)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int POL> static void run(const uint4 *table, uint32_t mask16, uint32_t *out, const char *what, const char *pol) {
    const uint32_t blocks = 256 * 32, iters = 256;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<POL>), dim3(blocks), dim3(256), 0, 0, table, mask16, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_gather<POL>), dim3(blocks), dim3(256), 0, 0, table, mask16, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double loads = 5.0 * blocks * 256.0 * iters;
    printf("%-14s %-12s : %8.1f G lane-loads/s\n", what, pol, loads / ms / 1e6);
}
int main() {
    for (int mb : {2, 48}) {
        const size_t bytes = (size_t)mb << 20;
        uint4 *table; uint32_t *out;
        CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 4)); CHECK(hipMemset(table, 1, bytes));
        const uint32_t mask16 = (uint32_t)(bytes / 16 - 1);
        char what[64]; snprintf(what, sizeof(what), "table %d MiB", mb);
        run<0>(table, mask16, out, what, "(none)"); run<1>(table, mask16, out, what, "sc0"); run<2>(table, mask16, out, what, "sc1"); run<3>(table, mask16, out, what, "nt");
        run<4>(table, mask16, out, what, "sc0 sc1"); run<5>(table, mask16, out, what, "sc0 nt"); run<6>(table, mask16, out, what, "sc1 nt"); run<7>(table, mask16, out, what, "sc0 sc1 nt");
        CHECK(hipFree(table)); CHECK(hipFree(out));
    }
    return 0;
}
