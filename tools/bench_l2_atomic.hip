// Microbenchmark: rate of scattered global atomics (no return) as a function of the footprint and of who shares it.
//   mode 0: every lane adds to a random row of ONE table shared by the whole chip (the reference's gradient scatter)
//   mode 1: every XCD adds to its OWN copy of the table (XCC_ID) — would such atomics stay in the XCD's L2?
// ops: pk_add_f16 (one row of a C = 2 fp16 table), add_f32, add_u32. rows: table size (4-byte rows).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
template <int OP, int PRIVATE>
__global__ void __launch_bounds__(256) k(uint32_t *table, uint32_t rows_mask, uint32_t rows_per_copy, int iters) {
    uint32_t xcc = 0;
    if (PRIVATE) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
    uint32_t *t = table + (size_t)xcc * rows_per_copy;
    uint32_t r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; it++) {
        r = r * 1664525u + 1013904223u;
        const uint32_t row = (r >> 8) & rows_mask;
        if (OP == 0) { v2h v = {(_Float16)1.0f, (_Float16)0.5f}; (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) v2h *)(t + row), v); }
        else if (OP == 1) (void)__hip_atomic_fetch_add(reinterpret_cast<float *>(t + row), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 2) (void)__hip_atomic_fetch_add(t + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 3) (void)__hip_atomic_fetch_add(t + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 4) t[row] = r;                                   // plain scattered 4-byte stores, for scale
    }
}
template <int OP, int PRIVATE> void run(const char *name, uint32_t *d, uint32_t rows) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 64, blocks = 256 * 16;
    hipLaunchKernelGGL((k<OP, PRIVATE>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, rows, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PRIVATE>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, rows, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-22s %-8s rows %8u (%6.1f MB%s)  %8.3f ms  %8.1f G atomics/s\n", name, PRIVATE ? "per-XCD" : "shared", rows, rows * 4.0 / 1e6, PRIVATE ? " x8" : "", ms,
           (double)blocks * 256 * iters / ms / 1e6);
}
int main() {
    uint32_t *d; const size_t bytes = (size_t)8 * (1u << 23) * 4;
    hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    for (uint32_t rows : {1u << 14, 1u << 19, 1u << 23}) {
        run<0, 0>("pk_add_f16 agent", d, rows); run<0, 1>("pk_add_f16 agent", d, rows);
        run<1, 0>("add_f32 agent", d, rows);    run<1, 1>("add_f32 agent", d, rows);
        run<2, 0>("add_u32 agent", d, rows);    run<2, 1>("add_u32 agent", d, rows);
        run<3, 0>("add_u32 workgroup", d, rows); run<3, 1>("add_u32 workgroup", d, rows);
        run<4, 0>("store b32", d, rows);        run<4, 1>("store b32", d, rows);
    }
    return 0;
}
