#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RS 96
__device__ __forceinline__ bf16x8 tr_frag(const short* p0, const short* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
  const bf16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p1);
  return __builtin_shufflevector(v, w, 0, 1, 2, 3, 4, 5, 6, 7);
}
// A = tile[pixel][ch] transposed-read -> A[ch][pixel]; B = identity-ish: D = A * B with B[k][col] = (k == col) so D[ch][col] = A[ch][k=col] for col<16
__global__ void k(float* out) {
  __shared__ short lds[16 * RS];
  __shared__ short eye[16 * RS];
  for (int i = threadIdx.x; i < 16 * RS; i += 64) {
    int px = i / RS, ch = i % RS;
    float v = (float)(px * 32 + ch);   // exactly representable in bf16? up to 16*32+95 < 1024: needs 10 bits -> not exact; use small values
    v = (float)((px * 7 + ch * 3) % 61);
    __bf16 h = (__bf16)v; lds[i] = __builtin_bit_cast(short, h);
    __bf16 e = (__bf16)((px == ch) ? 1.0f : 0.0f); eye[i] = __builtin_bit_cast(short, e);
  }
  __syncthreads();
  const int lane = threadIdx.x, lh = lane >> 5;
  const int g = lane >> 4, gi = lane & 15, tq = gi >> 2, tp = gi & 3;
  const int tr_ch = 16 * (g & 1) + 4 * tp;
  const int tr_px = 8 * lh + tq;
  const short* p0 = lds + tr_px * RS + tr_ch;
  bf16x8 a = tr_frag(p0, p0 + 4 * RS);
  const short* q0 = eye + tr_px * RS + tr_ch;
  bf16x8 b = tr_frag(q0, q0 + 4 * RS);     // B[k=pixel][col=ch] = eye
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 64 * 16 * 4); float h[1024];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);   // D[row = ch][col]
    float exp = col < 16 ? (float)((col * 7 + row * 3) % 61) : 0.f;  // A[ch=row][k=col]
    if (h[l * 16 + r] != exp) { if (bad < 8) printf("D[%d][%d] = %g expected %g\n", row, col, h[l*16+r], exp); ++bad; }
  }
  printf("bad = %d\n", bad);
  return 0;
}
