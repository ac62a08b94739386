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

// forward / dgrad launch parameters (see launch_conv3x3_bf16)
struct BConvP {
  const bf16_t* src0; const bf16_t* src1; const float* a0; const float* b0;
  const bf16_t* wpk;   // [9][N][Cin]
  const float* bias;
  bf16_t* dst0; bf16_t* dst1; float* stats;
  int C0, C1, Cin, N, D0, D1, B, H, W, tilesX, tilesY, nPix, nCo;
  unsigned rcp_nPix, rcp_tilesX, rcp_tilesY;   // fast path: floor(2^32 / d) + 1 (0 for d == 1)
  unsigned long long* dbg;   // optional s_memtime stamps per workgroup (tools/stamp_test.py; FU_CONV_STAMPS builds)
};

// aligned-shape fast path (fu_conv_bf16_fast.hip)
bool conv3x3_bf16_fast_eligible(const BConvP& P);
int launch_conv3x3_bf16_fast(BConvP& P, hipStream_t s);

}  // namespace fu
