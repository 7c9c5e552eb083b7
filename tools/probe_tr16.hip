// Probe of ds_read_b64_tr_b16 lane/element mapping (diagnostic; not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + threadIdx.x * 4));
  for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) { printf("lane %2d:", l); for (int e = 0; e < 4; e++) printf(" (src lane %2d, elem %d)", h[l*4+e] >> 2, h[l*4+e] & 3); printf("\n"); }
  return 0;
}
