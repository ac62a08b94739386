#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define RS 96
__global__ void k(short* out) {
  __shared__ short lds[16 * RS];
  for (int i = threadIdx.x; i < 16 * RS; i += 64) lds[i] = (short)((i / RS) * 100 + (i % RS));
  __syncthreads();
  const int lane = threadIdx.x, lh = lane >> 5;
  const int g = lane >> 4, gi = lane & 15, tq = gi >> 2, tp = gi & 3;
  const int tr_ch = 16 * (g & 1) + 4 * tp;
  const int tr_px = 8 * lh + tq;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const short* p0 = lds + tr_px * RS + tr_ch;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * RS));
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = v[j]; out[lane * 8 + 4 + j] = w[j]; }
}
int main() {
  short* d; (void)hipMalloc(&d, 1024); short h[512];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    for (int j = 0; j < 8; ++j) {
      int px = h[l*8+j] / 100, ch = h[l*8+j] % 100;
      int epx = 8 * (l >> 5) + j, ech = l & 31;
      if (px != epx || ch != ech) { if (bad < 12) printf("lane %d j %d: got (px %d, ch %d) expected (px %d, ch %d)\n", l, j, px, ch, epx, ech); ++bad; }
    }
  }
  printf("bad = %d\n", bad);
  return 0;
}
