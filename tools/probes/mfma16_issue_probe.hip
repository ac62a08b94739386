// Issue probe for v_mfma_f32_16x16x32_bf16 (round 4): cycles per MFMA of a wave that issues, behind every MFMA, k plain VALU
// instructions (k = 0..4), or one ds_read_b128 every third MFMA, alone on its SIMD or beside a partner wave of the same
// workgroup that runs a VALU-only / MFMA-only / idle loop.  s_memtime around 256 (MFMA + fillers) groups, 8 independent
// accumulators in rotation.   hipcc --offload-arch=gfx950 -O2 mfma16_issue_probe.hip -o mfma16_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define M(i) "v_mfma_f32_16x16x32_bf16 %" #i ", %8, %9, %" #i "\n"
#define V1 "v_fma_f32 %10, %11, %11, %10\n"
#define V2 V1 "v_fma_f32 %12, %11, %11, %12\n"
#define V3 V2 "v_fma_f32 %13, %11, %11, %13\n"
#define V4 V3 "v_fma_f32 %14, %11, %11, %14\n"
#define GROUP8(F) M(0) F M(1) F M(2) F M(3) F M(4) F M(5) F M(6) F M(7) F

// MODE of the partner half (waves 4-7 when 512 threads): 0 idle (exits), 1 VALU loop, 2 MFMA loop
#define KERNEL(NAME, F)                                                                                        \
  __global__ __launch_bounds__(512) void NAME(unsigned long long* out, float* sink, int dummy, int partner) { \
    __shared__ float lds[4096];                                                                                \
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;                                            \
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;                         \
    bf8 a, b;                                                                                                  \
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i + dummy); b[i] = (__bf16)(float)(i * 3 + dummy); } \
    float x = threadIdx.x + dummy, y = x + 1.f, z = y + 1.f, w = z + 1.f, s = 1.0001f;                            \
    __syncthreads();                                                                                           \
    const int wave = threadIdx.x >> 6;                                                                         \
    unsigned long long t0 = 0, t1 = 0;                                                                         \
    if (wave < 4) {                                                                                            \
      t0 = __builtin_amdgcn_s_memtime();                                                                       \
      asm volatile(REP32(GROUP8(F)) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
                   : "v"(a), "v"(b), "v"(x), "v"(s), "v"(y), "v"(z), "v"(w));                                   \
      asm volatile("s_nop 0" ::: "memory");                                                                    \
      t1 = __builtin_amdgcn_s_memtime();                                                                       \
    } else if (partner == 1) {                                                                                 \
      for (int it = 0; it < 6; ++it)                                                                            \
        asm volatile(REP32(REP8(V4)) : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b), "v"(x), "v"(s), "v"(y), "v"(z), "v"(w)); \
    } else if (partner == 2) {                                                                                 \
      for (int it = 0; it < 2; ++it)                                                                            \
        asm volatile(REP32(GROUP8("")) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) \
                     : "v"(a), "v"(b), "v"(x), "v"(s), "v"(y), "v"(z), "v"(w));                                 \
    }                                                                                                          \
    if (threadIdx.x % 64 == 0 && wave < 4) out[blockIdx.x * 4 + wave] = t1 - t0;                                 \
    f4 r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;                                                               \
    if (r[0] + r[1] + x + y + z + w == 12345.678f) sink[0] = r[2] + lds[threadIdx.x];                           \
  }

KERNEL(k_v0, "")
KERNEL(k_v1, V1)
KERNEL(k_v2, V2)
KERNEL(k_v3, V3)
KERNEL(k_v4, V4)
KERNEL(k_nop1, "s_nop 0\n")
KERNEL(k_v2nop, V2 "s_nop 0\n")
KERNEL(k_salu1, "s_add_u32 s20, s20, 1\n")

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 64);
  struct { const char* n; void (*k)(unsigned long long*, float*, int, int); } ks[] = {
      {"MFMA only", k_v0}, {"MFMA + 1 VALU", k_v1}, {"MFMA + 2 VALU", k_v2}, {"MFMA + 3 VALU", k_v3}, {"MFMA + 4 VALU", k_v4},
      {"MFMA + s_nop", k_nop1}, {"MFMA + 2 VALU + s_nop", k_v2nop}, {"MFMA + 1 SALU", k_salu1}};
  const char* pn[] = {"alone (256 threads)", "partner idle", "partner VALU loop", "partner MFMA loop"};
  for (auto& e : ks)
    for (int mode = 0; mode < 4; ++mode) {
      const int threads = mode == 0 ? 256 : 512, partner = mode == 0 ? 0 : mode - 1;
      std::vector<unsigned long long> h(256 * 4);
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, sink, 0, partner);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> v(h.begin(), h.end());
      std::sort(v.begin(), v.end());
      printf("%-24s %-22s median %6.2f cycles per MFMA group (p10 %.2f, p90 %.2f)\n", e.n, pn[mode], v[v.size() / 2] / 256.0,
             v[v.size() / 10] / 256.0, v[v.size() * 9 / 10] / 256.0);
    }
  return 0;
}
