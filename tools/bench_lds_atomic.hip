// Microbenchmark: LDS atomic add throughput (diagnostic for the binned grid backward).
// rows: number of distinct rows a wave's 64 lanes can hit per instruction (64 lanes / rows = same-address multiplicity).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(512) k(const uint32_t* idx, float* out, int iters, uint32_t spread_mask) {
  __shared__ double accd[16384];
  float* acc = (float*)accd;
  for (int i = threadIdx.x; i < 32768; i += 512) acc[i] = 0;
  __syncthreads();
  uint32_t r = idx[blockIdx.x * 512 + threadIdx.x];
  for (int it = 0; it < iters; it++) {
    r = r * 1664525u + 1013904223u;
    // lanes of a wave share the high bits (one random base per wave-iteration), spread_mask picks how many rows they differ in
    const uint32_t base = __builtin_amdgcn_readfirstlane(r >> 8) & 8191u;
    const uint32_t a = (base + ((r >> 12) & spread_mask)) & 8191u;
    if (MODE == 0) { atomicAdd(&acc[2 * a], 1.0f); atomicAdd(&acc[2 * a + 1], 2.0f); }
    else if (MODE == 1) { atomicAdd((int*)&acc[2 * a], 1); atomicAdd((int*)&acc[2 * a + 1], 2); }
    else if (MODE == 3) { atomicAdd((unsigned long long*)&accd[2 * a], 3ull); atomicAdd((unsigned long long*)&accd[2 * a + 1], 5ull); }
    else if (MODE == 4) { atomicAdd(&accd[2 * a], 1.0); atomicAdd(&accd[2 * a + 1], 2.0); }
    else if (MODE == 5) {        // one packed half2 add per record (both channels): ds_pk_add_f16
      typedef _Float16 v2h __attribute__((ext_vector_type(2)));
      v2h v = {(_Float16)1.0f, (_Float16)2.0f};
      (void)__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) v2h *)&acc[a], v);
    }
    else if (MODE == 6) { atomicAdd((unsigned long long*)&accd[a], 3ull); atomicAdd((unsigned long long*)&accd[8192 + a], 5ull); }   // planar u64
    else if (MODE == 7) { atomicAdd((int*)&acc[a], 1); }                                                                                  // one u32 add per record
  }
  __syncthreads();
  float s = 0; for (int i = threadIdx.x; i < 32768; i += 512) s += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, uint32_t* d, float* o, uint32_t mask) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 256, blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, o, iters, mask);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, o, iters, mask); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double recs = (double)blocks * 512 * iters;
  printf("%-18s rows/wave<=%-5u %8.3f ms  %9.2f G records/s\n", name, mask + 1, ms, recs / ms / 1e6);
}
int main() {
  uint32_t* d; float* o; hipMalloc(&d, 512 * 2048 * 4); hipMalloc(&o, 512 * 2048 * 4);
  uint32_t* h = (uint32_t*)malloc(512 * 2048 * 4); for (int i = 0; i < 512 * 2048; i++) h[i] = i * 2654435761u + 12345u;
  hipMemcpy(d, h, 512 * 2048 * 4, hipMemcpyHostToDevice);
  const uint32_t masks[] = {8191u, 63u, 7u, 0u};
  for (uint32_t m : masks) {
    run<0>("2 x ds_add_f32", d, o, m); run<1>("2 x ds_add_u32", d, o, m); run<3>("2 x ds_add_u64", d, o, m); run<4>("2 x ds_add_f64", d, o, m);
    run<5>("1 x ds_pk_add_f16", d, o, m); run<6>("2 x ds_add_u64 planar", d, o, m); run<7>("1 x ds_add_u32", d, o, m);
  }
  return 0;
}
