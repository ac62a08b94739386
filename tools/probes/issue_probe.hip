// Issue-cost probe: cycles per wave64 instruction for the instruction kinds the conv epilogue/staging code uses,
// measured with s_memtime around 256 back-to-back copies, at 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 issue_probe.hip -o issue_probe && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

#define PROBE(NAME, ASM)                                                                 \
  __global__ void NAME(unsigned long long* out, float* sink, int dummy) {                \
    float a = threadIdx.x * 1.5f + dummy, b = a + 1.f, c = b + 2.f, d = c + 3.f;          \
    float e = d * 2.f, f = e + 1.f, g = f + 1.f, h = g + 1.f;                             \
    __syncthreads();                                                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    asm volatile(REP256(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "s20", "s21", "vcc"); \
    asm volatile("s_nop 0" ::: "memory");                                                 \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    if (a + b + c + d + e + f + g + h == 12345.678f) sink[0] = a;                         \
  }

// independent pairs (alternating destination registers) and dependent chains
PROBE(k_fma_ind, "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3\n")
PROBE(k_fma_dep, "v_fma_f32 %0, %0, %2, %0\n v_fma_f32 %0, %0, %2, %0\n")
PROBE(k_cvt_ind, "v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n")
PROBE(k_cvt_dep, "v_cvt_pk_bf16_f32 %0, %0, %2\n v_cvt_pk_bf16_f32 %0, %0, %2\n")
PROBE(k_dpp_ind, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
PROBE(k_dpp_dep, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n")
PROBE(k_perm_ind, "v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %4, %1, %2, %3\n")
PROBE(k_cnd_ind, "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %1, %2, vcc\n")
PROBE(k_pkmax_ind, "v_pk_max_i16 %0, %1, %2\n v_pk_max_i16 %3, %1, %2\n")
PROBE(k_and_ind, "v_and_b32 %0, %1, %2\n v_and_b32 %3, %1, %2\n")
PROBE(k_mullo_ind, "v_mul_lo_u32 %0, %1, %2\n v_mul_lo_u32 %3, %1, %2\n")

PROBE(k_cnd64_ind, "v_cndmask_b32_e64 %0, %1, %2, s[20:21]\n v_cndmask_b32_e64 %3, %1, %2, s[20:21]\n")
PROBE(k_cnd_same, "v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n")
PROBE(k_add_sgpr, "v_add_f32 %0, s20, %1\n v_add_f32 %3, s20, %2\n")
PROBE(k_and_sgpr, "v_and_b32 %0, s20, %1\n v_and_b32 %3, s21, %2\n")
PROBE(k_cmp, "v_cmp_gt_i32 vcc, %0, %1\n v_cmp_gt_i32 vcc, %2, %3\n")
PROBE(k_cmp_cnd, "v_cmp_gt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc\n")
PROBE(k_bfe, "v_bfe_i32 %0, %1, 3, 1\n v_bfe_i32 %3, %2, 4, 1\n")
PROBE(k_lshl, "v_lshlrev_b32 %0, 16, %1\n v_lshlrev_b32 %3, 16, %2\n")
PROBE(k_andlit, "v_and_b32 %0, 0xffff0000, %1\n v_and_b32 %3, 0xffff0000, %2\n")
PROBE(k_salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
PROBE(k_mix, "v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1\n")

// packed fp32 needs register pairs
#define PROBE2(NAME, ASM)                                                                 \
  __global__ void NAME(unsigned long long* out, float* sink, int dummy) {                \
    typedef float f2 __attribute__((ext_vector_type(2)));                                 \
    f2 a = {threadIdx.x * 1.5f + dummy, 2.f}, b = a + 1.f, c = b + 2.f, d = c + 3.f;      \
    __syncthreads();                                                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    asm volatile(REP256(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));                       \
    asm volatile("s_nop 0" ::: "memory");                                                 \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.678f) sink[0] = a.x;       \
  }
PROBE2(k_pkfma_ind, "v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %3, %1, %2, %3\n")
PROBE2(k_pkfma_dep, "v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %0, %0, %2, %0\n")
PROBE2(k_pkadd_ind, "v_pk_add_f32 %0, %1, %2\n v_pk_add_f32 %3, %1, %2\n")

// AGPR reads
__global__ void k_accread(unsigned long long* out, float* sink, int dummy) {
  float a = threadIdx.x + dummy, b = 0.f;
  asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n s_nop 4\n" ::"v"(a) : "a0", "a1");
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile(REP256("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n") : "+v"(a), "+v"(b));
  asm volatile("s_nop 0" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a + b == 12345.678f) sink[0] = a;
}

typedef void (*kern_t)(unsigned long long*, float*, int);
struct Item { const char* name; kern_t k; };

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 1 << 20); hipMalloc(&sink, 64);
  Item items[] = {{"v_fma_f32 indep", k_fma_ind}, {"v_fma_f32 dep", k_fma_dep}, {"v_cvt_pk_bf16_f32 indep", k_cvt_ind},
                  {"v_cvt_pk_bf16_f32 dep", k_cvt_dep}, {"v_mov_b32_dpp indep", k_dpp_ind},
                  {"v_mov_b32_dpp dep(+s_nop 1)", k_dpp_dep}, {"v_perm_b32 indep", k_perm_ind},
                  {"v_cndmask_b32 indep", k_cnd_ind}, {"v_pk_max_i16 indep", k_pkmax_ind}, {"v_and_b32 indep", k_and_ind},
                  {"v_mul_lo_u32 indep", k_mullo_ind}, {"v_pk_fma_f32 indep", k_pkfma_ind}, {"v_pk_fma_f32 dep", k_pkfma_dep},
                  {"v_pk_add_f32 indep", k_pkadd_ind}, {"v_accvgpr_read_b32", k_accread}, {"v_cndmask_b32_e64 sgpr mask", k_cnd64_ind}, {"v_cndmask_b32 vcc dst=src0", k_cnd_same}, {"v_add_f32 with SGPR src", k_add_sgpr}, {"v_and_b32 with SGPR src", k_and_sgpr}, {"v_cmp_gt_i32 -> vcc", k_cmp}, {"v_cmp + v_cndmask pair (per instr)", k_cmp_cnd}, {"v_bfe_i32", k_bfe}, {"v_lshlrev_b32 imm", k_lshl}, {"v_and_b32 literal", k_andlit}, {"s_add_u32", k_salu}, {"v_fma + s_add pair (per instr)", k_mix}};
  for (int wpw = 1; wpw <= 2; ++wpw) {   // waves per SIMD
    const int threads = 256 * wpw;
    printf("--- %d wave(s) per SIMD (one %d-thread block per CU), cycles per instruction (s_memtime ticks / 512)\n", wpw, threads);
    for (auto& it : items) {
      std::vector<unsigned long long> h(256 * 8);
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(it.k, dim3(256), dim3(threads), 0, 0, out, sink, 0);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), out, 256 * 4 * wpw * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 256 * 4 * wpw; ++i) s += h[i];
      printf("%-32s %.2f\n", it.name, s / (256 * 4 * wpw) / 512.0);
    }
  }
  return 0;
}
