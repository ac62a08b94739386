// Private declarations shared by the bf16 convolution translation units (fu_conv_bf16.hip, fu_conv_bf16_fast.hip).
#pragma once
#include "fu_common.h"

#include <type_traits>

namespace fu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#ifdef __HIPCC__
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32
}

// relu(a * x + b) on the two bf16 channels of one dword; one rounding to bf16
__device__ __forceinline__ unsigned bn_relu_pair(unsigned v, f32x2 a, f32x2 b) {
  f32x2 x;
  x.x = __uint_as_float(v << 16);
  x.y = __uint_as_float(v & 0xffff0000u);
  x = __builtin_elementwise_fma(a, x, b);                                    // v_pk_fma_f32 (= bn_act_fused per lane)
  const s16x2 h = __builtin_bit_cast(s16x2, pack_bf16x2(x));
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, z));      // v_pk_max_i16: bf16 sign test = relu
}

#endif

// forward / dgrad launch parameters (see launch_conv3x3_bf16)
struct BConvP {
  const bf16_t* src0; const bf16_t* src1; const float* a0; const float* b0;
  const bf16_t* wpk;   // [9][N][Cin]
  const float* bias;
  bf16_t* dst0; bf16_t* dst1; float* stats;
  int C0, C1, Cin, N, D0, D1, B, H, W, tilesX, tilesY, nPix, nCo;
  unsigned rcp_nPix, rcp_tilesX, rcp_tilesY;   // fast path: floor(2^32 / d) + 1 (0 for d == 1)
  unsigned long long* dbg;   // optional s_memtime stamps per workgroup (tools/stamp_test.py; FU_CONV_STAMPS builds)
  int center_only;           // 1: every tap but the centre one of wpk is zero (embedded 1x1): the fast kernel skips them
};

// aligned-shape fast path (fu_conv_bf16_fast.hip)
bool conv3x3_bf16_fast_eligible(const BConvP& P);
int launch_conv3x3_bf16_fast(BConvP& P, hipStream_t s);

}  // namespace fu
