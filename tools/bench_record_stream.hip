// bench_record_stream.hip — how fast can 0.7 GB of records that the previous kernel has just written be read back?
// The binned grid backward's reduce (csrc/gridencoder.hip, k_gbin_reduce) streams its records at ~3.1 TB/s; the microarchitecture guide measures
// 6.0-6.3 TB/s for an in-order sweep of HBM. Variants of the READ side (the write side is always the same plain fill):
//   A  reduce-like: one 1024-thread workgroup per 32768-record chunk of 8 B (chunk = contiguous 256 KiB), UNR 8-byte loads in flight per lane
//   B  the same with 16-byte loads
//   C  256-thread workgroups, 4 per chunk
//   D  grid-stride sweep of the whole buffer (consecutive workgroups read consecutive 16 KiB), 16-byte loads
// build: hipcc -O3 --offload-arch=gfx950 tools/bench_record_stream.hip -o tools/bench_record_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fill(uint2 *p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = make_uint2((uint32_t)i, (uint32_t)(i >> 7));
}

template <int THREADS, int UNR>
__global__ void __launch_bounds__(THREADS) k_read8_chunks(const uint2 *__restrict__ p, uint64_t n, uint32_t chunk, uint32_t *out, uint32_t lds_pad) {
    extern __shared__ uint32_t pad[];
    const uint32_t per = chunk / (1024 / THREADS);                        // records per workgroup
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t acc = 0;
    for (uint64_t base = lo; base < hi; base += (uint64_t)THREADS * UNR) {
        uint2 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { const uint64_t i = base + threadIdx.x + (uint64_t)u * THREADS; v[u] = i < hi ? p[i] : make_uint2(0u, 0u); }
#pragma unroll
        for (int u = 0; u < UNR; u++) acc += v[u].x ^ v[u].y;
    }
    if (lds_pad && threadIdx.x == 0) pad[0] = acc;
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UNR>
__global__ void __launch_bounds__(1024) k_read16_chunks(const uint4 *__restrict__ p, uint64_t n16, uint32_t chunk16, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint64_t lo = (uint64_t)blockIdx.x * chunk16, hi = lo + chunk16 < n16 ? lo + chunk16 : n16;
    uint32_t acc = 0;
    for (uint64_t base = lo; base < hi; base += 1024ull * UNR) {
        uint4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { const uint64_t i = base + threadIdx.x + (uint64_t)u * 1024; v[u] = i < hi ? p[i] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < UNR; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (threadIdx.x == 0) pad[0] = acc;
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_read16_sweep(const uint4 *__restrict__ p, uint64_t n16, uint32_t *out) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t base = (uint64_t)blockIdx.x * 1024; base < n16; base += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint64_t i = base + threadIdx.x + (uint64_t)u * 256; v[u] = i < n16 ? p[i] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t bytes = 720ull << 20, n = bytes / 8, n16 = bytes / 16;
    uint2 *buf; uint32_t *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t chunk = 32768, n_chunks = (uint32_t)((n + chunk - 1) / chunk);
    auto timed = [&](const char *name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 6; rep++) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, n);      // "the scatter": the buffer is freshly written every time
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-58s %7.3f ms (best %7.3f)  %6.2f TB/s\n", name, sum / 5, best, bytes / (sum / 5 * 1e-3) / 1e12);
    };
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_read8_chunks<1024, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_read8_chunks<1024, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_read16_chunks<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
    timed("A  1024-thread WG per 32768-record chunk, 8 B x 4, 1 WG/CU", [&] { hipLaunchKernelGGL((k_read8_chunks<1024, 4>), dim3(n_chunks), dim3(1024), 129 * 1024, 0, buf, n, chunk, out, 1u); });
    timed("A' the same, 8 B x 8 in flight, 1 WG/CU", [&] { hipLaunchKernelGGL((k_read8_chunks<1024, 8>), dim3(n_chunks), dim3(1024), 129 * 1024, 0, buf, n, chunk, out, 1u); });
    timed("A2 the same kernel without the LDS (2 WG/CU)", [&] { hipLaunchKernelGGL((k_read8_chunks<1024, 4>), dim3(n_chunks), dim3(1024), 64, 0, buf, n, chunk, out, 1u); });
    timed("B  1024-thread WG per chunk, 16 B x 4, 1 WG/CU", [&] { hipLaunchKernelGGL((k_read16_chunks<4>), dim3(n_chunks), dim3(1024), 129 * 1024, 0, (const uint4 *)buf, n16, chunk / 2, out); });
    timed("C  256-thread WGs, 4 per chunk, 8 B x 4", [&] { hipLaunchKernelGGL((k_read8_chunks<256, 4>), dim3(n_chunks * 4), dim3(256), 64, 0, buf, n, chunk, out, 0u); });
    timed("D  grid-stride sweep, 16 B x 4, 2048 WGs", [&] { hipLaunchKernelGGL(k_read16_sweep, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, n16, out); });
    timed("D' grid-stride sweep, 16 B x 4, 8192 WGs", [&] { hipLaunchKernelGGL(k_read16_sweep, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, n16, out); });
    return 0;
}
