// Private declarations shared by the 16-bit convolution translation units (fu_conv_bf16.hip, fu_conv_bf16_fast.hip).
//
// ONE source, TWO element types.  The Makefile compiles both files twice: as they stand (bf16: v_mfma_f32_32x32x16_bf16) and
// with -DFU_HALF=1 (IEEE fp16: v_mfma_f32_32x32x16_f16, the same MFMA rate).  Tiles, LDS images, schedules, address math
// and the transposing reads are identical -- a 16-bit element is a 16-bit element -- so the only things that change are
// the four conversions (f2e / e2f_lo / e2f_hi / pack_e2), the fragment vector type and the MFMA / ds_read_tr builtins,
// all defined below.  Device buffers stay raw 16-bit storage (`bf16_t` = unsigned short in both builds); the fp16 build's
// entry points and kernels carry `f16` in their names (the #defines at the end of the FU_HALF block), the testing hooks
// and their globals live in the bf16 objects only.
#pragma once
#include "fu_common.h"

#ifndef FU_HALF
#define FU_HALF 0
#endif
#if FU_HALF
// names of the fp16 build (the bf16 build keeps the names as written in the sources)
#define launch_pack_conv3x3_bf16 launch_pack_conv3x3_f16
#define launch_conv3x3_bf16 launch_conv3x3_f16
#define launch_conv3x3_wgrad_bf16 launch_conv3x3_wgrad_f16
#define launch_conv3x3_bf16_fast launch_conv3x3_f16_fast
#define conv3x3_bf16_fast_eligible conv3x3_f16_fast_eligible
#define conv3x3_num_stat_tiles_bf16 conv3x3_num_stat_tiles_f16
#define conv3x3_wgrad_slab_elems_bf16 conv3x3_wgrad_slab_elems_f16
#define k_pack_bf16 k_pack_f16
#define k_conv3x3_bf16 k_conv3x3_f16
#define k_conv3x3_bf16_fast k_conv3x3_f16_fast
#define k_wgrad_bf16 k_wgrad_f16
#define k_wgrad_bf16_pp k_wgrad_f16_pp
#define k_wgrad_bf16_c8 k_wgrad_f16_c8
#define conv3x3_rs_eligible conv3x3_rs_eligible_f16
#define launch_conv3x3_rs launch_conv3x3_rs_f16
#define conv3x3_c8_eligible conv3x3_c8_eligible_f16
#define launch_conv3x3_c8 launch_conv3x3_c8_f16
#define conv3x3_pp_eligible conv3x3_pp_eligible_f16
#define launch_conv3x3_pp launch_conv3x3_pp_f16
#define conv3x3_pp_preferred conv3x3_pp_preferred_f16
#define conv3x3_pp_preferred_bnb conv3x3_pp_preferred_bnb_f16
#endif

#include <type_traits>

namespace fu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#if FU_HALF
typedef _Float16 frag8_t __attribute__((ext_vector_type(8)));     // one MFMA operand fragment: 8 consecutive k per lane
typedef __fp16 tr4_t __attribute__((ext_vector_type(4)));         // (the element type the ds_read_tr16 builtin is declared with)
#define FU_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FU_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4f16(p)
#else
typedef __bf16 frag8_t __attribute__((ext_vector_type(8)));
typedef __bf16 tr4_t __attribute__((ext_vector_type(4)));
#define FU_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define FU_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p)
#endif

// compile-time loop: every array index below is a constant expression, so the staging registers are never
// demoted to scratch (runtime-indexed private arrays are -- cdna guide rule 20)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define FU_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#ifdef __HIPCC__
// ---- the element conversions (the only arithmetic that depends on the 16-bit format) ----
#if FU_HALF
typedef _Float16 e16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2e(float f) { return f2h(f); }                                   // RNE
__device__ __forceinline__ float e2f_lo(unsigned v) { return h2f((unsigned short)(v & 0xffffu)); }  // v_cvt_f32_f16
__device__ __forceinline__ float e2f_hi(unsigned v) { return h2f((unsigned short)(v >> 16)); }      // v_cvt_f32_f16 sdwa WORD_1
__device__ __forceinline__ f32x2 e2f_pair(unsigned v) {
  return __builtin_convertvector(__builtin_bit_cast(e16x2, v), f32x2);
}
#else
typedef __bf16 e16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2e(float f) { return f2bf(f); }
__device__ __forceinline__ float e2f_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float e2f_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ f32x2 e2f_pair(unsigned v) {
  f32x2 x;
  x.x = __uint_as_float(v << 16);
  x.y = __uint_as_float(v & 0xffff0000u);
  return x;
}
#endif
__device__ __forceinline__ unsigned pack_e2(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, e16x2));    // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (RNE)
}

// relu(a * x + b) on the two 16-bit channels of one dword; one rounding to the element type
#ifndef FU_PACKED_BN
#define FU_PACKED_BN 0
#endif
__device__ __forceinline__ unsigned bn_relu_pair(unsigned v, f32x2 a, f32x2 b) {
#if !FU_PACKED_BN
  // plain v_fma_f32 / v_max_f32 per channel.  The packed-f32 form below is three instructions shorter per pair and was
  // round 1's choice, but packed f32 VALU is slow beside a co-resident wave's MFMAs (MI355X_MICROARCH.md, "price of one
  // filler": +22 cycles per v_pk_fma_f32): s_memtime stamps of the row-stationary kernel's staging phase, 10 units per
  // thread and chunk: 6000 cycles packed, 3300 plain (tools/stamp_rs.py)
  const float lo = fmaxf(fmaf(a.x, e2f_lo(v), b.x), 0.f), hi = fmaxf(fmaf(a.y, e2f_hi(v), b.y), 0.f);
  return pack_e2(f32x2{lo, hi});
#endif
  f32x2 x = e2f_pair(v);
  x = __builtin_elementwise_fma(a, x, b);                                    // v_pk_fma_f32 (= bn_act_fused per lane)
  const s16x2 h = __builtin_bit_cast(s16x2, pack_e2(x));
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, z));      // v_pk_max_i16: the sign bit test of bf16 AND of
}                                                                            // fp16 (sign-magnitude, bit 15) = relu

#endif

// forward / dgrad launch parameters (see launch_conv3x3_bf16)
struct BConvP {
  const bf16_t* src0; const bf16_t* src1; const float* a0; const float* b0;
  const bf16_t* wpk;   // [9][N][Cin]
  const float* bias;
  bf16_t* dst0; bf16_t* dst1; float* stats;
  int C0, C1, Cin, N, D0, D1, B, H, W, tilesX, tilesY, nPix, nCo;
  unsigned rcp_nPix, rcp_tilesX, rcp_tilesY;   // fast path: floor(2^32 / d) + 1 (0 for d == 1)
  unsigned rcp_nCo;                            // row-stationary kernel: channel tile fastest in the workgroup order
  unsigned long long* dbg;   // optional s_memtime stamps per workgroup (tools/stamp_test.py; FU_CONV_STAMPS builds)
  int center_only;           // 1: every tap but the centre one of wpk is zero (embedded 1x1): the fast kernel skips them
  // row-stationary kernel, dgrad into a BatchNorm's output gradient (BnbFuse, fu_common.h): the raw conv output y of that
  // BatchNorm, its a / b / mean / invstd, and the per-tile sums [nPix][N][2]; all null otherwise
  const bf16_t* bnb_y; const float* bnb_a; const float* bnb_b; const float* bnb_mean; const float* bnb_invstd; float* bnb_part;
  int nTiles;                // persistent ping-pong kernel (fu_conv_pp.hip): nPix * nCo, walked by gridDim.x workgroups
};

// aligned-shape fast path (fu_conv_bf16_fast.hip)
bool conv3x3_bf16_fast_eligible(const BConvP& P);
int launch_conv3x3_bf16_fast(BConvP& P, const LaunchOpts& o, hipStream_t s);
// 8-input-channel forward kernel (fu_conv_rs.hip): the network's first conv
bool conv3x3_c8_eligible(const BConvP& P);
int launch_conv3x3_c8(BConvP& P, const LaunchOpts& o, hipStream_t s);
// row-stationary 16x16x32 kernel (fu_conv_rs.hip)
bool conv3x3_rs_eligible(const BConvP& P);
int launch_conv3x3_rs(BConvP& P, const LaunchOpts& o, hipStream_t s);
// persistent ping-pong row-stationary kernel (fu_conv_pp.hip): 8 waves, two LDS stages, one workgroup per CU
bool conv3x3_pp_eligible(const BConvP& P);
bool conv3x3_pp_preferred(const BConvP& P);   // ... and worth it: at least one tile per CU
bool conv3x3_pp_preferred_bnb(const BConvP& P);   // ... when the launch is asked for fused BatchNorm-backward sums
int launch_conv3x3_pp(BConvP& P, const LaunchOpts& o, hipStream_t s);

}  // namespace fu

