// Gather-rate microbenchmark for gfx950: how many divergent loads per second can the chip retire when every lane of a wave
// reads a different cache line of an L2-resident table, as a function of the load width (4 / 8 / 16 B), and how much it helps
// when groups of lanes share lines. Build: hipcc -O3 --offload-arch=gfx950 tools/bench_gather.hip -o tools/bench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// W = bytes per load (4, 8, 16); SHARE = lanes per distinct address group (1 = all different); table_mask in units of 16 B
template <int W, int SHARE>
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, uint32_t mask16, uint32_t iters, uint32_t *__restrict__ out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    uint32_t key = (tid / SHARE) * 0x9e3779b9u;
    for (uint32_t it = 0; it < iters; it += 8) {
        uint32_t idx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) idx[u] = mix(key + (it + u) * 0x85ebca6bu) & mask16;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const char *p = reinterpret_cast<const char *>(table + idx[u]);
            if (W == 4) acc += *reinterpret_cast<const uint32_t *>(p);
            else if (W == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x ^ v.y; }
            else { const uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W, int SHARE>
static void run(const uint4 *table, uint32_t mask16, uint32_t *out, const char *what) {
    const uint32_t blocks = 256 * 32, iters = 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<W, SHARE>), dim3(blocks), dim3(256), 0, 0, table, mask16, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_gather<W, SHARE>), dim3(blocks), dim3(256), 0, 0, table, mask16, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double loads = 5.0 * blocks * 256.0 * iters;
    printf("%-28s width %2d B  share %2d : %8.1f G lane-loads/s  (%.3f ms per 268 M loads)\n", what, W, SHARE, loads / ms / 1e6, 268.4e6 / (loads / ms));
}

int main() {
    for (int mb : {2, 48}) {
        const size_t bytes = (size_t)mb << 20;
        uint4 *table; uint32_t *out;
        CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 4));
        CHECK(hipMemset(table, 1, bytes));
        const uint32_t mask16 = (uint32_t)(bytes / 16 - 1);
        char what[64]; snprintf(what, sizeof(what), "table %d MiB", mb);
        run<4, 1>(table, mask16, out, what); run<8, 1>(table, mask16, out, what); run<16, 1>(table, mask16, out, what);
        run<4, 2>(table, mask16, out, what); run<4, 4>(table, mask16, out, what); run<4, 16>(table, mask16, out, what); run<4, 64>(table, mask16, out, what);
        CHECK(hipFree(table)); CHECK(hipFree(out));
    }
    return 0;
}
