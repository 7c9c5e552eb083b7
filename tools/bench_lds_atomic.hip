// Microbenchmark: LDS atomic add throughput on random addresses (diagnostic for the binned grid backward).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(512) k(const uint32_t* idx, float* out, int iters) {
  __shared__ float acc[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) acc[i] = 0;
  __syncthreads();
  uint32_t r = idx[blockIdx.x * 512 + threadIdx.x];
  for (int it = 0; it < iters; it++) {
    r = r * 1664525u + 1013904223u;
    const uint32_t a = (r >> 8) & 8191u;
    if (MODE == 0) { atomicAdd(&acc[2 * a], 1.0f); atomicAdd(&acc[2 * a + 1], 2.0f); }
    else if (MODE == 1) { atomicAdd((int*)&acc[2 * a], 1); atomicAdd((int*)&acc[2 * a + 1], 2); }
    else if (MODE == 2) { acc[2 * a] = 1.0f; acc[2 * a + 1] = 2.0f; }
    else if (MODE == 3) { atomicAdd((unsigned long long*)&acc[2 * a], 0x100000001ull); }
    else if (MODE == 4) { atomicAdd((double*)&acc[2 * a], 1.0); }
  }
  __syncthreads();
  float s = 0; for (int i = threadIdx.x; i < 16384; i += 512) s += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, uint32_t* d, float* o) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 512, blocks = 512 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, o, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double recs = (double)blocks * 512 * iters;
  printf("%-28s %8.3f ms  %8.2f G records/s\n", name, ms, recs / ms / 1e6);
}
int main() {
  uint32_t* d; float* o; hipMalloc(&d, 512 * 2048 * 4); hipMalloc(&o, 512 * 2048 * 4);
  uint32_t* h = (uint32_t*)malloc(512 * 2048 * 4); for (int i = 0; i < 512 * 2048; i++) h[i] = i * 2654435761u + 12345u;
  hipMemcpy(d, h, 512 * 2048 * 4, hipMemcpyHostToDevice);
  run<0>("2 x ds_add_f32", d, o); run<1>("2 x ds_add_u32", d, o); run<2>("2 x ds_write_b32", d, o); run<3>("1 x ds_add_u64", d, o); run<4>("1 x ds_add_f64", d, o);
  return 0;
}
