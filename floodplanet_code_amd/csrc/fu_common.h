// Shared declarations for libfloodunet (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace fu {

typedef unsigned short bf16_t;  // raw bf16 storage
struct f16_t { unsigned short v; };   // raw IEEE fp16 storage (a distinct type, so that the kernels can be instantiated on it)

// PREC_F16: fp16 activations / weights / gradient maps on the matrix cores (v_mfma_f32_32x32x16_f16, the bf16 rate), fp32
// accumulation, statistics, loss and master weights; the gradient maps carry a power-of-two loss scale (see LossScale).
enum Prec { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };

// ---- error plumbing (thread-local message, never exceptions across the ABI) ----------------
void set_error(const char* fmt, ...);
const char* get_error();

#define FU_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      ::fu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                              \
    }                                                                                        \
  } while (0)

#define FU_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      ::fu::set_error(__VA_ARGS__);    \
      return 1;                        \
    }                                  \
  } while (0)

#define FU_TRY(expr)           \
  do {                         \
    int _s = (expr);           \
    if (_s != 0) return _s;    \
  } while (0)

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline unsigned host_rcp(int d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1) << 32) / (unsigned)d + 1); }

// ---- device helpers -------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ float h2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }   // v_cvt_f32_f16
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }  // RNE
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_h2(float a, float b) {     // two floats -> packed fp16 pair (RNE)
  const float2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2_t));
}
__device__ __forceinline__ void unpack_h2(unsigned u, float& a, float& b) {
  const float2_t v = __builtin_convertvector(__builtin_bit_cast(half2_t, u), float2_t);
  a = v.x; b = v.y;
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static constexpr int VEC = 4;  // elements per 16-byte access
  __device__ static __forceinline__ void load4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
  __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    t.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = t;
  }
  __device__ static __forceinline__ float load1(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void store1(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct ElemIO<f16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void load4(const f16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    unpack_h2(t.x, v[0], v[1]); unpack_h2(t.y, v[2], v[3]);
  }
  __device__ static __forceinline__ void store4(f16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = pack_h2(v[0], v[1]); t.y = pack_h2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
  }
  __device__ static __forceinline__ float load1(const f16_t* p) { return h2f(p->v); }
  __device__ static __forceinline__ void store1(f16_t* p, float v) { p->v = f2h(v); }
};

// The activation every consumer re-derives from a stored pre-BN value y: relu(a*y + b), with a = gamma*invstd and
// b = beta - mean*a.  ATen's CPU batch_norm (what the reference runs, and what the golden fixtures were made with)
// forms exactly these two coefficients and evaluates x*alpha + beta as a multiply and a separate add, so the parity
// definition is TWO roundings, never an fma: with it the activations are bit-identical to torch's given identical
// y / mean / invstd, and ReLU / max-pool near-ties fall the same way (measured: with an fma here the worst gradient
// tensor of the 300x300 fixture moved from 0.6 % to 1.2 % off the reference).  The bf16 conv staging rounds the
// result to bf16 anyway and uses the fused form (bn_act_fused).
__device__ __forceinline__ float bn_act_pre(float a, float y, float b) {
#pragma clang fp contract(off)   // (HIP's __fmul_rn / __fadd_rn are plain operators and DO get fused under -ffp-contract=fast)
  const float p = a * y;
  return p + b;
}
__device__ __forceinline__ float bn_act(float a, float y, float b) { return fmaxf(bn_act_pre(a, y, b), 0.f); }
__device__ __forceinline__ float bn_act_fused(float a, float y, float b) { return fmaxf(fmaf(a, y, b), 0.f); }

// 16-byte vector access: V elements of T as floats (V = 4 for fp32, 8 for bf16)
template <typename T> struct VecIO;
template <> struct VecIO<float> {
  static constexpr int V = 4;
  __device__ static __forceinline__ void load(const float* p, float (&v)[4]) { ElemIO<float>::load4(p, v); }
  __device__ static __forceinline__ void store(float* p, const float (&v)[4]) { ElemIO<float>::store4(p, v); }
};
template <> struct VecIO<bf16_t> {
  static constexpr int V = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    t.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    t.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
    t.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

template <> struct VecIO<f16_t> {
  static constexpr int V = 8;
  __device__ static __forceinline__ void load(const f16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    unpack_h2(t.x, v[0], v[1]); unpack_h2(t.y, v[2], v[3]); unpack_h2(t.z, v[4], v[5]); unpack_h2(t.w, v[6], v[7]);
  }
  __device__ static __forceinline__ void store(f16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack_h2(v[0], v[1]); t.y = pack_h2(v[2], v[3]); t.z = pack_h2(v[4], v[5]); t.w = pack_h2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

// n / d for n * d < 2^32 with rcp = floor(2^32 / d) + 1 made on the host (0 encodes d == 1)
__device__ __forceinline__ int fast_div(int n, int d, unsigned rcp) {
  (void)d;
  return rcp ? (int)__umulhi((unsigned)n, rcp) : n;
}

// XCD-aware bijective remap of a linear block id: blocks b and b+8 share an XCD (observed round
// robin), so give each XCD a contiguous chunk of the logical id space (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#endif  // __HIPCC__

// Optional event pair recorded immediately around the main kernel of ONE conv / wgrad launch (LaunchOpts::prof: filled by
// the API layer from the context's profiler when profiling is on; per launch, never process-wide state).
struct ProfSlot { hipEvent_t start = nullptr; hipEvent_t stop = nullptr; };

// Exact data-parallel mode (fu_set_exact_sync): at every statistics reduction the partial sums are summed over the
// ranks before they are finalised.  The API layer points g_sync at the context's descriptor for the duration of a
// call; the three launchers with such a reduction (BN forward statistics, BN backward sums, CE loss sums) check it.
struct SyncDesc {
  int (*hook)(void* user, int64_t n_elems, int is_double) = nullptr;   // sums xbuf[0..n) over the ranks, in place
  void* user = nullptr;
  void* xbuf = nullptr;        // caller-owned device exchange buffer
  int64_t xbytes = 0;
  int world = 1;
};
extern thread_local const SyncDesc* g_sync;
// payload (device, n elements of float or double) -> xbuf, hook, back; no-op without an active descriptor
int sync_sum_over_ranks(void* payload, int64_t n_elems, bool is_double, hipStream_t s);
static inline int sync_world() { return (g_sync && g_sync->hook) ? g_sync->world : 1; }

// fp16 mode: the gradient maps (fp16) carry a power-of-two loss scale S chosen on the device from max|dL/dlogits| at the
// start of every backward (launch_loss_grad_eff).  scale[0] = S, scale[1] = 1/S.  The three places that write PARAMETER
// gradients (weight-gradient transpose + bias tail, BN backward finalize, head backward finalize) multiply by 1/S, so the
// flat gradient buffer always holds true gradients.  The API layer points g_grad_unscale at scale + 1 for the duration of
// a backward call in fp16 mode; nullptr (fp32 / bf16) means 1.
extern thread_local const float* g_grad_unscale;

// BatchNorm-backward sums from the PRODUCER of dL/dz (round 2): a dgrad launch whose single destination g = dL/d relu(bn(y))
// feeds a BatchNorm backward can emit that backward's two sums (sum g*m, sum g*m*xhat, m = [a*y + b > 0]) per workgroup
// tile from its epilogue, where g is still in registers -- k_bn_bwd_reduce (one read of g and one of y, 14 launches per
// step) then disappears for that layer.  The API layer passes this request with the launch (ConvIn::opt.bnb);
// a kernel that honours it (the row-stationary 16-bit kernel, single destination, <= max_tiles workgroup tiles) writes
// part[tile][channel][2] and *tiles_out = number of tiles; everything else leaves *tiles_out untouched (0 = not fused).
struct BnbFuse {
  const void* y = nullptr;                 // raw conv output of the BatchNorm, NHWC, the destination's shape
  const float *a = nullptr, *b = nullptr, *mean = nullptr, *invstd = nullptr;
  float* part = nullptr;
  int64_t max_elems = 0;                   // capacity of part (floats)
  int* tiles_out = nullptr;
  bool skip_g = false;                     // head backward only: g itself is not stored (HeadGrad below recomputes it)
};
// The head's data gradient g[p][c] = sum_k dl[p][k] w[k][c] is two or three FMAs per element: when the head backward has
// left the BatchNorm-backward sums of g behind (BnbFuse), the apply pass of that BatchNorm recomputes g from dl and w
// instead of reading a stored copy -- one 134 MB write and one 134 MB read less per step at the bench shape.
struct HeadGrad {
  const float* dl = nullptr;               // dL/dlogits, NHWC fp32 [npix][ncls]
  const float* w = nullptr;                // head weight [ncls][C]
  int ncls = 0;
};
// per-launch options of the conv / wgrad launchers (host side only; travels inside ConvIn)
struct LaunchOpts {
  ProfSlot prof;                  // events around the main kernel
  const BnbFuse* bnb = nullptr;   // dgrad launches: request for the destination BatchNorm's backward sums
};
// out = dlogits * (*up_scale_dev or 1) * (S or 1); scale != null (fp16): S chosen from max|dlogits * up| and written to scale[0..1]
int launch_loss_grad_eff(const float* dlogits, float* out, int64_t n, const float* up_scale_dev,
                         float* partials /* >= 256 floats */, float* scale /* [2] or null */, hipStream_t s,
                         const int* guard = nullptr /* fp16 guard words: [2] = back-off exponent */);
// fp16 guard: guard[0] <- 1 if a gradient is not finite; bookkeeping after the (possibly skipped) Adam launch
int launch_grad_finite_check(const float* g, int64_t n, int* guard, hipStream_t s);
int launch_guard_book(int* guard, hipStream_t s);

#ifdef FU_EXPERIMENTS   // ablation-by-skip in variant builds only (tools/build_variant.sh exp -DFU_EXPERIMENTS): FU_EXP_SKIP bit mask,
// 1 = BN forward finalize, 2 = BN backward finalize, 4 = wgrad slab reduce + transpose, 8 = weight pack, 16 = BN backward apply,
// 32 = bilinear backward, 64 = pool-folded BN backward passes.  The results are WRONG; only the step time means something.
#include <stdlib.h>
static inline int exp_skip() {
  static int m = -1;
  if (m < 0) { const char* e = getenv("FU_EXP_SKIP"); m = e ? atoi(e) : 0; }
  return m;
}
#define FU_EXP_SKIP(bit) (::fu::exp_skip() & (bit))
#else
#define FU_EXP_SKIP(bit) 0
#endif

// ---- kernel launchers (implemented in the .hip files) ----------------------------------------
// All pointers are device pointers; T-typed buffers are `void*` + Prec.

struct ConvIn {              // the (virtual) input of a 3x3 convolution
  const void* src0; int C0;  // first C0 channels (skip / plain input)
  const float* a0;           // optional BN scale for src0: x = relu(a0*src0 + b0)
  const float* b0;
  const void* src1; int C1;  // next C1 channels, taken as they are (may be null/0)
  bool center_only = false;  // the packed weights are a 1x1 conv embedded as the centre tap (late-fusion convs)
  LaunchOpts opt;            // events / fused-sum request of this launch
};

// conv as implicit GEMM: out[p][n] = sum_{tap,c} in[p+tap][c] * wpk[tap][c][n] (+bias[n])
// wpk layout is precision specific (see the pack kernels).  Output channels [0,D0) -> dst0, [D0,D0+D1) -> dst1.
// stats: optional [nStatTiles][N][2] partial (sum, sumsq) of the bias-free result per pixel tile;
// *n_stat_tiles receives the number of tiles written (<= conv3x3_num_stat_tiles()).
int conv3x3_num_stat_tiles(Prec p, int B, int H, int W);
int launch_conv3x3(Prec p, const ConvIn& in, const void* wpk, const float* bias, void* dst0, int D0, void* dst1,
                   int D1, float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s);
// dW partial slabs + fixed-order reduce into fp32 OIHW (+ bias gradient from db_partials [n][Cout]).
int64_t conv3x3_wgrad_slab_elems(Prec p, int Cin, int Cout, int B, int H, int W);
int launch_conv3x3_wgrad(Prec p, const ConvIn& in, const void* dy, int Cout, float* slab, float* dw_oihw,
                         int cin_real, const float* db_partials, int n_db_partials, float* db, int B, int H,
                         int W, hipStream_t s);

// weight packing: fp32 OIHW [Cout][cin_real][3][3] -> fwd pack (GEMM K = cin padded to cin_pad, N = Cout)
// and dgrad pack (taps reversed, K = Cout, N = cin_pad).  Either destination may be null.
int64_t conv3x3_pack_elems(Prec p, int cin_pad, int Cout);
int launch_pack_conv3x3(Prec p, const float* w_oihw, int Cout, int cin_real, int cin_pad, void* wfwd, void* wdgrad,
                        hipStream_t s);

// precision-specific implementations (fu_conv_f32.hip / fu_conv_bf16.hip)
int conv3x3_num_stat_tiles_f32(int B, int H, int W);
int launch_conv3x3_f32(const ConvIn& in, const float* wpk, const float* bias, float* dst0, int D0, float* dst1, int D1,
                       float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s);
int64_t conv3x3_wgrad_slab_elems_f32(int Cin, int Cout, int B, int H, int W);
int launch_conv3x3_wgrad_f32(const ConvIn& in, const float* dy, int Cout, float* slab, float* dw_oihw, int cin_real,
                             const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                             hipStream_t s);
int launch_pack_conv3x3_f32(const float* w_oihw, int Cout, int cin_real, int cin_pad, float* wfwd, float* wdgrad,
                            hipStream_t s);
int conv3x3_num_stat_tiles_bf16(int B, int H, int W);
int64_t conv3x3_wgrad_slab_elems_bf16(int Cin, int Cout, int B, int H, int W);
int launch_conv3x3_bf16(const ConvIn& in, const bf16_t* wpk, const float* bias, bf16_t* dst0, int D0, bf16_t* dst1,
                        int D1, float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s);
int launch_conv3x3_wgrad_bf16(const ConvIn& in, const bf16_t* dy, int Cout, float* slab, float* dw_oihw, int cin_real,
                              const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                              hipStream_t s);
int launch_pack_conv3x3_bf16(const float* w_oihw, int Cout, int cin_real, int cin_pad, bf16_t* wfwd, bf16_t* wdgrad,
                             hipStream_t s);
// the same sources compiled for IEEE fp16 (fu_conv_bf16.h, -DFU_HALF=1); buffers are raw 16-bit storage
int conv3x3_num_stat_tiles_f16(int B, int H, int W);
int64_t conv3x3_wgrad_slab_elems_f16(int Cin, int Cout, int B, int H, int W);
int launch_conv3x3_f16(const ConvIn& in, const bf16_t* wpk, const float* bias, bf16_t* dst0, int D0, bf16_t* dst1,
                       int D1, float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s);
int launch_conv3x3_wgrad_f16(const ConvIn& in, const bf16_t* dy, int Cout, float* slab, float* dw_oihw, int cin_real,
                             const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                             hipStream_t s);
int launch_pack_conv3x3_f16(const float* w_oihw, int Cout, int cin_real, int cin_pad, bf16_t* wfwd, bf16_t* wdgrad,
                            hipStream_t s);
extern int g_bf16_force_cfg;

int launch_nchw_to_nhwc(Prec p, const float* src, void* dst, int B, int C, int H, int W, int c_pad, hipStream_t s,
                        int src_channels = 0, int src_channel_offset = 0);
int launch_nhwc_to_nchw(Prec p, const void* src, float* dst, int B, int C, int H, int W, int c_pad, hipStream_t s);
// up to 8 fp32 NCHW tensors [B, c[k], H, W] taken side by side along the channel axis; coff = prefix sums of c
struct SrcList { const float* p[8]; int c[8]; int coff[9]; int n; };
// dst [B,H,W,c_pad] <- channels [ch_off, ch_off + C) of the virtual concatenation (zero padding above C)
int launch_gather_nchw_to_nhwc(Prec p, const SrcList& S, void* dst, int B, int C, int H, int W, int c_pad, int ch_off,
                               hipStream_t s);

// BN: finalize forward statistics.  partials [nTiles][C][2]; count = B*H*W.
// training: writes mean/invstd/a/b, updates running stats (momentum 0.1, unbiased var) and nbt.
int launch_bn_finalize(const float* partials, int nTiles, int C, int64_t count, const float* conv_bias,
                       const float* gamma, const float* beta, float eps, float momentum, float* mean, float* invstd,
                       float* a, float* b, float* running_mean, float* running_var, int64_t* nbt, double* dscratch,
                       hipStream_t s);
// fp64 scratch needed by the two-level reductions: elements for a layer with C channels
static inline int64_t reduce_scratch_elems(int C) { return (int64_t)32 * C * 2; }
// backward: g (in place -> dy).  Two launches + finalize inside.  partials scratch: >= bn_bwd_partial_elems.
int64_t bn_bwd_partial_elems(int C, int64_t npix);
// g_pool != null: y's 2x2 max-pool ran in forward and g_pool [B, H/2, W/2, C] is dL/d(pooled): the pool's backward is folded
// into the two passes (g is then the skip gradient only)
int launch_bn_bwd(Prec p, void* g, const void* y, int C, int64_t npix, const float* a, const float* b,
                  const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                  float* partials, float* coef, float* db_partials, int* n_db_partials, double* dscratch,
                  hipStream_t s, const void* g_pool = nullptr, int B = 0, int H = 0, int W = 0, int ext_partials = 0,
                  const HeadGrad* head = nullptr);

int launch_maxpool2(Prec p, const void* src, const float* a, const float* b, void* dst, int B, int H, int W, int C,
                    hipStream_t s);
struct UpTables {  // device tables for one bilinear x2 resize (H x W -> 2H x 2W, placed inside outH x outW)
  const int* y_i0; const int* y_i1; const float* y_w1;  // [2H]
  const int* x_i0; const int* x_i1; const float* x_w1;  // [2W]
  // backward gather lists: for input index i up to UP_BWD_MAX (o, w) pairs
  const int* yb_o; const float* yb_w;  // [H][UP_BWD_MAX]
  const int* xb_o; const float* xb_w;  // [W][UP_BWD_MAX]
  float scale_y, scale_x;              // (in - 1) / (out - 1) in float, as ATen: the forward kernel evaluates the
};                                     // index/weight expressions itself (same values as y_*/x_*)
static constexpr int UP_BWD_MAX = 6;
int launch_upsample2(Prec p, const void* src, const float* a, const float* b, void* dst, int B, int H, int W, int C,
                     int outH, int outW, const UpTables& t, hipStream_t s);
int launch_upsample2_bwd(Prec p, const void* g_dst, void* g_src, int B, int H, int W, int C, int outH, int outW,
                         const UpTables& t, hipStream_t s);

// ConvTranspose2d(k2,s2) helpers (bilinear=False variant): the four phase GEMMs run as one 1x1 conv with 4 cout channels at
// the low resolution; depth-to-space (+ F.pad) / space-to-depth (- pad) move between [B,h,w,4C] and [B,2h(+pad),2w(+pad),C];
// pad-region zeroing, bias-gradient partial sums, weight <-> embedded-centre-tap conversions
int launch_depth_to_space(Prec p, const void* y4, void* up, int B, int h, int w, int C, int outH, int outW, hipStream_t s);
int launch_space_to_depth(Prec p, const void* gup, void* g4, int B, int h, int w, int C, int outH, int outW, hipStream_t s);
int launch_colsum_partials(const float* partials, int n, int C, float* out, hipStream_t s);
int launch_channel_partial_sums(Prec p, const void* g, int C, int64_t npix, float* partials, int* n_partials,
                                hipStream_t s);
int launch_convT_to_w3(const float* w, const float* b, int Cin, int Cout, float* w3, float* bias4, hipStream_t s);
int launch_convT_grad_from_w3(const float* dw3, int Cin, int Cout, float* dw, hipStream_t s);
int launch_copy_channels(Prec p, const void* src, int srcC, int src_off, const float* a, const float* b, void* dst,
                         int dstC, int dst_off, int C, int64_t npix, hipStream_t s);
int launch_center_to_w3(const float* w, int64_t n, float* w3, hipStream_t s);
int launch_center_from_w3(const float* dw3, int64_t n, float* dw, hipStream_t s);

// head: 1x1 conv + bias on relu(a*y+b); logits fp32 NHWC [npix][ncls] (+ optional NCHW copy)
int launch_head_fwd(Prec p, const void* y, const float* a, const float* b, const float* w, const float* bias, int C,
                    int ncls, int B, int H, int W, float* logits_nhwc, float* logits_nchw, hipStream_t s);
static constexpr int HEAD_MAX_CLS = 8;
// CE loss over stored logits.  partials: [nblk][2] (loss sum, valid count as float) -> finalize
int launch_ce_loss(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                   float* partials, float* loss_out, int64_t* n_valid_dev, int64_t* confusion_accum,
                   int64_t* n_valid_out, unsigned long long* conf_tmp /* [64], zero-initialised */, hipStream_t s);
// dlogits (NHWC fp32) from CE: (softmax - onehot)/n_valid on valid pixels
int launch_ce_grad(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                   const int64_t* n_valid_dev, float* dlogits_nhwc, hipStream_t s);
// BCE + soft Dice on softmax(z)[1]; partials >= 5*400 floats, coef 4 floats; dlogits may be null (eval)
int launch_bce_dice(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                    float dice_w, float* partials, float* coef, float* loss_out, int64_t* n_valid_dev,
                    float* dlogits_nhwc, hipStream_t s);
int launch_dlogits_from_nchw(const float* dlogits_nchw, float* dlogits_nhwc, int ncls, int B, int H, int W,
                             hipStream_t s);
// head backward: G[y] = dlogits * W ; dW = sum dlogits (x) z ; db = sum dlogits
int64_t head_bwd_partial_elems(int C, int ncls);
int launch_head_bwd(Prec p, const float* dlogits_nhwc, const void* y, const float* a, const float* b, const float* w,
                    int C, int ncls, int64_t npix, void* g, float* partials, float* dw, float* db, hipStream_t s,
                    const BnbFuse* fuse = nullptr);

int launch_stitch_add(const float* logits_nhwc, int ncls, int cropW, float* canvas, float* weight, int canvasW, int h0,
                      int w0, int dh, int dw, hipStream_t s);
int launch_stitch_finalize(float* canvas, const float* weight, int ncls, int64_t npix, int64_t* argmax_out,
                           hipStream_t s);
int launch_assemble_tiles(const float* const* srcs, const int* src_channels, int n_src, int B, int H, int W, const int* vh,
                          const int* vw, int norm_mode, const float* gmean, const float* gstd, float pad_value, float* out,
                          float* mean_out, float* std_out, hipStream_t s);
int launch_resize_lanczos4_tiles(const float* win, int B, int C, int win_h, int win_w, const int* iy, const float* wy,
                                 const int* ix, const float* wx, int TH, int TW, int scale_mode, float* out, hipStream_t s);
int launch_augment(const float* img, const int64_t* tgt, float* img_o, int64_t* tgt_o, const int* flags,
                   const float* angle, int B, int C, int H, int W, int64_t target_fill, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                double eps, int64_t step, double grad_scale, hipStream_t s, const int* skip = nullptr);
void adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale, float out[7]);
int launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* scalars_dev, hipStream_t s,
                    const int* skip = nullptr);

}  // namespace fu
