// Diagnostic: the dW kernel's transposed fragment reads vs plain 2-byte reads of the same LDS tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4 lds_s4;
#define W 72
__global__ void k(int* bad, short* dump) {
  __shared__ __attribute__((aligned(16))) short sD[64][W];
  for (int i = threadIdx.x; i < 64 * W; i += 256) sD[i / W][i % W] = (short)((i / W) * 100 + (i % W));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int mt = wave & 1;
  const int q = (lane & 15) >> 2, p = lane & 3, cg = 16 * ((lane >> 4) & 1);
  int nb = 0;
  for (int ks = 0; ks < 4; ks++) {
    const int k0 = 16 * ks + 8 * h + q;
    const s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sD[k0][32 * mt + cg + 4 * p]);
    const s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4 *)&sD[k0 + 4][32 * mt + cg + 4 * p]);
    for (int e = 0; e < 8; e++) {
      const short got = e < 4 ? a0[e] : a1[e - 4];
      const short want = sD[16 * ks + 8 * h + e][32 * mt + r];
      if (got != want) nb++;
      if (wave == 0 && ks == 0) dump[lane * 8 + e] = got;
    }
  }
  atomicAdd(bad, nb);
}
int main() {
  int* d; short* dd; (void)hipMalloc(&d, 4); (void)hipMalloc(&dd, 1024); (void)hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dd);
  int h; short hd[512]; (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  printf("mismatches: %d\n", h);
  for (int l = 0; l < 64; l += 9) { printf("lane %2d:", l); for (int e = 0; e < 8; e++) printf(" %5d", hd[l*8+e]); printf("\n"); }
  return 0;
}
