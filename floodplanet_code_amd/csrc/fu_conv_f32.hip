// fp32 3x3 convolutions for gfx950 as im2col-free implicit GEMMs on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
//   forward / dgrad : out[p][n] = sum_{tap,c} in[p+tap][c] * wpk[tap][c][n]
//                     M = output pixels (8x16 or 4x16 tile per workgroup), N = 64 channels, K = 9*C in chunks
//                     of 16 channels.  The halo tile of the (virtual, two-source) NHWC input is staged through
//                     LDS once per chunk and reused by the 9 taps; the previous layer's BatchNorm+ReLU is applied
//                     while staging (x = relu(a*y+b)), so normalised activations never exist in HBM.  The epilogue
//                     adds the bias, stores NHWC and emits per-tile (sum, sum of squares) partials for this
//                     layer's own batch statistics.  dgrad is the same kernel on tap-reversed, K/N-swapped weights.
//   wgrad           : dW[tap][c][n] = sum_p in[p+tap][c] * dy[p][n]  -- M = 64 in-channels, N = 64 out-channels,
//                     K = pixels (4x16 tiles), 9 accumulator tiles (one per tap) per wave; split-K over pixel
//                     tiles into fp32 slabs that a second kernel sums in a fixed order (deterministic).
#include "fu_common.h"

namespace fu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FU_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

// ------------------------------------------------------------------------------------------------
// weight packing (fp32): OIHW -> wf[tap][ci][co] and wd[8-tap][co][ci]
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_f32(const float* __restrict__ w, int Cout, int cin_real, int cin_pad, float* __restrict__ wf,
                           float* __restrict__ wd, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Cout);
    const int64_t r = idx / Cout;
    const int ci = (int)(r % cin_pad);
    const int tap = (int)(r / cin_pad);
    const float v = ci < cin_real ? w[((int64_t)co * cin_real + ci) * 9 + tap] : 0.f;
    if (wf) wf[idx] = v;
    if (wd) wd[((int64_t)(8 - tap) * Cout + co) * cin_pad + ci] = v;
  }
}

int launch_pack_conv3x3_f32(const float* w_oihw, int Cout, int cin_real, int cin_pad, float* wfwd, float* wdgrad,
                            hipStream_t s) {
  const int64_t total = (int64_t)9 * cin_pad * Cout;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_pack_f32, dim3(g), dim3(256), 0, s, w_oihw, Cout, cin_real, cin_pad, wfwd, wdgrad, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ------------------------------------------------------------------------------------------------
struct ConvP {
  const float* src0; const float* src1; const float* a0; const float* b0;
  const float* wpk; const float* bias;
  float* dst0; float* dst1; float* stats;
  int C0, C1, Cin, N, D0, D1, B, H, W, tilesX, tilesY, nPix, nCo;
};

template <int TH>
struct ConvCfg {
  static constexpr int TW = 16, BN = 64, KC = 16, NT = 256;
  static constexpr int HWd = TW + 2, HHt = TH + 2, NHP = HHt * HWd;
  static constexpr int PS = ((NHP + 7) / 8) * 8 + 2;  // == 2 (mod 8): conflict-free transposing ds_write_b32
  static constexpr int A_UNITS = NHP * (KC / 4);
  static constexpr int A_ITERS = (A_UNITS + NT - 1) / NT;
  static constexpr int MT = TH / 4;  // 32-pixel m-tiles (2 rows x 16) per wave; waves laid out 2 (m) x 2 (n)
  static constexpr int SMEM_FLOATS = KC * PS + 9 * KC * BN;
};

template <int TH>
__global__ __launch_bounds__(256) void k_conv3x3_f32(ConvP P) {
  using Cfg = ConvCfg<TH>;
  constexpr int TW = Cfg::TW, BN = Cfg::BN, KC = Cfg::KC, HWd = Cfg::HWd, PS = Cfg::PS;
  constexpr int A_ITERS = Cfg::A_ITERS, MT = Cfg::MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;             // [KC][PS]   channel-major halo tile
  float* sW = smem + KC * PS;   // [9][KC][BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int coT = logical / P.nPix;
  const int pixT = logical - coT * P.nPix;
  const int tx = pixT % P.tilesX;
  const int t2 = pixT / P.tilesX;
  const int ty = t2 % P.tilesY;
  const int bb = t2 / P.tilesY;
  const int x0 = tx * TW, y0 = ty * TH, n0 = coT * BN;

  // ---- per-thread staging descriptors -------------------------------------------------------
  const int aq = tid & 3;  // channel quad inside the chunk (256 % 4 == 0 -> same for every iteration)
  int a_hp[A_ITERS];
  int64_t a_pix[A_ITERS];
  bool a_ok[A_ITERS];
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int u = tid + it * 256;
    const int hp = u >> 2;
    const int hy = hp / HWd, hx = hp - hy * HWd;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    a_hp[it] = hp;
    a_ok[it] = (u < Cfg::A_UNITS) && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    a_pix[it] = ((int64_t)bb * P.H + iy) * P.W + ix;
  }
  const int w_ci = tid >> 4;
  const int w_n = n0 + 4 * (tid & 15);
  const bool w_nok = w_n < P.N;

  float4 ra[A_ITERS];
  float4 rw[9];
  const bool has_bn = P.a0 != nullptr;

  auto load_chunk = [&](int k0) {
    const int c = k0 + 4 * aq;
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool from0 = c < P.C0;
    if (has_bn && from0) {
      av = *reinterpret_cast<const float4*>(P.a0 + c);
      bv = *reinterpret_cast<const float4*>(P.b0 + c);
    }
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_ok[it] && c < P.Cin) {
        if (from0) {
          v = *reinterpret_cast<const float4*>(P.src0 + a_pix[it] * P.C0 + c);
          if (has_bn) {
            v.x = bn_act(av.x, v.x, bv.x);
            v.y = bn_act(av.y, v.y, bv.y);
            v.z = bn_act(av.z, v.z, bv.z);
            v.w = bn_act(av.w, v.w, bv.w);
          }
        } else {
          v = *reinterpret_cast<const float4*>(P.src1 + a_pix[it] * P.C1 + (c - P.C0));
        }
      }
      ra[it] = v;
    }
    const int ci = k0 + w_ci;
    const bool wok = w_nok && ci < P.Cin;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      rw[tap] = wok ? *reinterpret_cast<const float4*>(P.wpk + ((int64_t)tap * P.Cin + ci) * P.N + w_n)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (tid + it * 256 < Cfg::A_UNITS) {
        float* d = sA + (4 * aq) * PS + a_hp[it];
        d[0] = ra[it].x;
        d[PS] = ra[it].y;
        d[2 * PS] = ra[it].z;
        d[3 * PS] = ra[it].w;
      }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      *reinterpret_cast<float4*>(sW + (tap * KC + w_ci) * BN + 4 * (tid & 15)) = rw[tap];
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = ((wm * MT + mt) * 2 + (l31 >> 4)) * HWd + (l31 & 15) + lh * PS;
  const int boff = lh * BN + wn * 32 + l31;

  const int nChunks = (P.Cin + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < nChunks; ++ch) {
    __syncthreads();  // previous chunk's LDS reads are done
    store_chunk();
    __syncthreads();
    if (ch + 1 < nChunks) load_chunk((ch + 1) * KC);  // in flight under the MFMAs below
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * HWd + (tap % 3);
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const float bfr = sW[(tap * KC + 2 * kk) * BN + boff];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float afr = sA[(2 * kk) * PS + aoff[mt] + toff];
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, bfr, acc[mt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, NHWC store (two destinations), batch-statistics partials ----------------
  const int n = n0 + wn * 32 + l31;
  const bool nok = n < P.N;
  const float bias = (P.bias && nok) ? P.bias[n] : 0.f;
  float* dst;
  int dstride, dn;
  if (n < P.D0) { dst = P.dst0; dstride = P.D0; dn = n; }
  else { dst = P.dst1; dstride = P.D1; dn = n - P.D0; }
  float ssum = 0.f, ssq = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;  // row of the 32x32 tile held by this register
      const int oy = y0 + (wm * MT + mt) * 2 + (p >> 4);
      const int ox = x0 + (p & 15);
      if (nok && oy < P.H && ox < P.W) {
        const float v = acc[mt][r];
        ssum += v;
        ssq += v * v;
        dst[(((int64_t)bb * P.H + oy) * P.W + ox) * dstride + dn] = v + bias;
      }
    }
  }
  if (P.stats) {
    ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    __syncthreads();  // all MFMA-phase LDS reads finished -> reuse sA
    if (lh == 0) {
      sA[(wm * 64 + wn * 32 + l31) * 2 + 0] = ssum;
      sA[(wm * 64 + wn * 32 + l31) * 2 + 1] = ssq;
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < P.N) {
      float* o = P.stats + ((int64_t)pixT * P.N + n0 + tid) * 2;
      o[0] = sA[tid * 2 + 0] + sA[(64 + tid) * 2 + 0];
      o[1] = sA[tid * 2 + 1] + sA[(64 + tid) * 2 + 1];
    }
  }
}

static inline int conv_pick_th(int B, int H, int W, int N) {
  const int nCo = ceil_div(N, 64);
  const int64_t wg8 = (int64_t)B * ceil_div(H, 8) * ceil_div(W, 16) * nCo;
  return wg8 >= 768 ? 8 : 4;
}

int conv3x3_num_stat_tiles_f32(int B, int H, int W) { return B * ceil_div(H, 4) * ceil_div(W, 16); }

int launch_conv3x3_f32(const ConvIn& in, const float* wpk, const float* bias, float* dst0, int D0, float* dst1, int D1,
                       float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s) {
  ConvP P;
  P.src0 = (const float*)in.src0; P.src1 = (const float*)in.src1; P.a0 = in.a0; P.b0 = in.b0;
  P.wpk = wpk; P.bias = bias; P.dst0 = dst0; P.dst1 = dst1; P.stats = stats;
  P.C0 = in.C0; P.C1 = in.src1 ? in.C1 : 0; P.Cin = P.C0 + P.C1; P.N = D0 + D1; P.D0 = D0; P.D1 = D1;
  P.B = B; P.H = H; P.W = W;
  FU_REQUIRE(P.C0 % 4 == 0 && P.C1 % 4 == 0 && P.N % 4 == 0 && D0 % 4 == 0,
             "conv3x3_f32: channel counts must be multiples of 4 (C0=%d C1=%d N=%d)", P.C0, P.C1, P.N);
  const int th = conv_pick_th(B, H, W, P.N);
  P.tilesX = ceil_div(W, 16); P.tilesY = ceil_div(H, th); P.nPix = B * P.tilesX * P.tilesY; P.nCo = ceil_div(P.N, 64);
  const int grid = P.nPix * P.nCo;
  if (n_stat_tiles) *n_stat_tiles = P.nPix;
  const ProfSlot ps = in.opt.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  if (th == 8) {
    const size_t sh = ConvCfg<8>::SMEM_FLOATS * sizeof(float);
    hipLaunchKernelGGL(k_conv3x3_f32<8>, dim3(grid), dim3(256), sh, s, P);
  } else {
    const size_t sh = ConvCfg<4>::SMEM_FLOATS * sizeof(float);
    hipLaunchKernelGGL(k_conv3x3_f32<4>, dim3(grid), dim3(256), sh, s, P);
  }
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
struct WgP {
  const float* src0; const float* src1; const float* a0; const float* b0; const float* dy;
  float* slab;
  int C0, C1, Cin, Cout, B, H, W, tilesX, tilesY, nPix, nCi, nCo, S, perSplit;
};

static constexpr int WG_PTH = 4, WG_PTW = 16, WG_HW = 18, WG_NHP = 6 * 18, WG_CT = 64;

__global__ __launch_bounds__(256) void k_wgrad_f32(WgP P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                     // [108][64]
  float* sD = smem + WG_NHP * WG_CT;    // [64][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int mi = wave >> 1, ni = wave & 1;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nT = P.nCi * P.nCo;
  const int split = logical / nT;
  const int t = logical - split * nT;
  const int ciT = t / P.nCo, coT = t - ciT * P.nCo;
  const int ci0 = ciT * WG_CT, co0 = coT * WG_CT;

  f32x16 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  const bool has_bn = P.a0 != nullptr;
  const int q = tid & 15;
  const int cX = ci0 + 4 * q;   // this thread's input-channel quad (virtual concat index)
  const int cD = co0 + 4 * q;
  float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool from0 = cX < P.C0;
  if (has_bn && from0 && cX < P.Cin) {
    av = *reinterpret_cast<const float4*>(P.a0 + cX);
    bv = *reinterpret_cast<const float4*>(P.b0 + cX);
  }

  const int pt0 = split * P.perSplit;
  const int pt1 = min(P.nPix, pt0 + P.perSplit);
  for (int pt = pt0; pt < pt1; ++pt) {
    const int tx = pt % P.tilesX;
    const int t2 = pt / P.tilesX;
    const int ty = t2 % P.tilesY;
    const int bb = t2 / P.tilesY;
    const int x0 = tx * WG_PTW, y0 = ty * WG_PTH;
    __syncthreads();
    // stage X halo tile: 108 pixels x 16 quads
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int u = tid + it * 256;
      const int hp = u >> 4;
      if (hp < WG_NHP) {
        const int hy = hp / WG_HW, hx = hp - hy * WG_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W && cX < P.Cin) {
          const int64_t pix = ((int64_t)bb * P.H + iy) * P.W + ix;
          if (from0) {
            v = *reinterpret_cast<const float4*>(P.src0 + pix * P.C0 + cX);
            if (has_bn) {
              v.x = bn_act(av.x, v.x, bv.x);
              v.y = bn_act(av.y, v.y, bv.y);
              v.z = bn_act(av.z, v.z, bv.z);
              v.w = bn_act(av.w, v.w, bv.w);
            }
          } else {
            v = *reinterpret_cast<const float4*>(P.src1 + pix * P.C1 + (cX - P.C0));
          }
        }
        *reinterpret_cast<float4*>(sX + hp * WG_CT + 4 * q) = v;
      }
    }
    // stage dy tile: 64 pixels x 16 quads
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = (tid + it * 256) >> 4;
      const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < P.H && ox < P.W && cD < P.Cout)
        v = *reinterpret_cast<const float4*>(P.dy + (((int64_t)bb * P.H + oy) * P.W + ox) * P.Cout + cD);
      *reinterpret_cast<float4*>(sD + p * WG_CT + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < WG_PTH; ++r) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int cx = 2 * s8 + lh;
        const float bfr = sD[(r * 16 + cx) * WG_CT + ni * 32 + l31];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float afr = sX[((r + tap / 3) * WG_HW + cx + tap % 3) * WG_CT + mi * 32 + l31];
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, bfr, acc[tap], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: slab[split][tap][ci][co]
  const int co = co0 + ni * 32 + l31;
  if (co < P.Cout) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < P.Cin) P.slab[(((int64_t)split * 9 + tap) * P.Cin + ci) * P.Cout + co] = acc[tap][r];
      }
    }
  }
}

// dw[co][ci][tap] = sum_s slab[s][tap][ci][co]  (fixed order);  db[co] = sum_i dbp[i][co]
// Bandwidth kernel: 256 threads = SL split lanes x (256/SL) float4 columns; a block owns 4*256/SL consecutive slab
// elements (co fastest -> coalesced rows); every lane keeps 4 independent float4 loads in flight; lanes meet in
// LDS in a fixed order (deterministic), then the block scatters its elements to OIHW.
template <int SL>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int S, int Cin, int Cout,
                                                      int cin_real, float* __restrict__ dw,
                                                      const float* __restrict__ dbp, int ndb, float* __restrict__ db,
                                                      const float* __restrict__ unscale) {
  constexpr int COLS = 256 / SL, ELEMS = 4 * COLS;
  __shared__ float sm[SL][ELEMS];
  const int64_t nSlab = (int64_t)9 * Cin * Cout;          // elements of one slab ([tap][ci][co], Cout % 4 == 0)
  const int64_t nBlocksW = (nSlab + ELEMS - 1) / ELEMS;
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  if ((int64_t)blockIdx.x < nBlocksW) {
    const int64_t e0 = (int64_t)blockIdx.x * ELEMS + col * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (e0 < nSlab) {
      const float* p = slab + e0;
      int s = sl;
      for (; s + 3 * SL < S; s += 4 * SL) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)s * nSlab);
        const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(s + SL) * nSlab);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 2 * SL) * nSlab);
        const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 3 * SL) * nSlab);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; s < S; s += SL) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)s * nSlab);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
    }
    sm[sl][col * 4 + 0] = (a0.x + a1.x) + (a2.x + a3.x);
    sm[sl][col * 4 + 1] = (a0.y + a1.y) + (a2.y + a3.y);
    sm[sl][col * 4 + 2] = (a0.z + a1.z) + (a2.z + a3.z);
    sm[sl][col * 4 + 3] = (a0.w + a1.w) + (a2.w + a3.w);
    __syncthreads();
    // the block's summed elements go back IN PLACE into slab 0 (only this block ever touches them), coalesced;
    // k_wgrad_transpose then turns [tap][ci][co] into OIHW with full-line writes
    if (threadIdx.x < COLS && e0 < nSlab) {
      float4 o;
      float* op = &o.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < SL; ++k) acc += (double)sm[k][col * 4 + j];
        op[j] = (float)acc;
      }
      *reinterpret_cast<float4*>(const_cast<float*>(slab) + e0) = o;
    }
  } else if (db) {
    // bias gradient: 256 threads = 16 channels x 16 partial lanes, 8 independent loads in flight per lane
    // (one thread walking all ~2000 partials is a 100+ us latency chain that would set this kernel's duration)
    __shared__ double smd[16][16];
    const int64_t bblk = (int64_t)blockIdx.x - nBlocksW;
    const int c16 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int co = (int)(bblk * 16 + c16);
    double acc = 0.0;
    if (co < Cout) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int i = pl;
      for (; i + 7 * 16 < ndb; i += 8 * 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += dbp[(int64_t)(i + u * 16) * Cout + co];
      }
      for (; i < ndb; i += 16) a[0] += dbp[(int64_t)i * Cout + co];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)a[u];
    }
    smd[pl][c16] = acc;
    __syncthreads();
    if (pl == 0 && co < Cout) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += smd[k][c16];
      if (unscale) t *= (double)*unscale;       // fp16 mode: dy carried the loss scale
      db[co] = (float)t;
    }
  }
}

// packed [tap][Cin][Cout] (slab 0 after the reduce) -> dw OIHW [Cout][cin_real][9].
// block tile: 32 co x 8 ci x 9 taps through LDS: 128-byte reads along co, 288-byte writes along (ci, tap).
// CI4: the 16-bit weight-gradient kernels write their slabs as [tap][Cin / 4][Cout][4] -- a lane of the 32x32 accumulator
// tile holds four consecutive c_in of one c_out, so each lane stores 16 bytes and a wave 512 contiguous bytes (with
// [tap][Cin][Cout] the epilogue was 144 four-byte stores per thread, issue-bound on the texture path).
template <bool CI4>
__global__ __launch_bounds__(256) void k_wgrad_transpose(const float* __restrict__ packed, int Cin, int Cout,
                                                         int cin_real, float* __restrict__ dw,
                                                         const float* __restrict__ unscale) {
  __shared__ float t[72][33];
  const float us = unscale ? *unscale : 1.f;      // fp16 mode: 1 / loss scale (a power of two: exact)
  const int nCo = (Cout + 31) / 32;
  const int co0 = (blockIdx.x % nCo) * 32, ci0 = (blockIdx.x / nCo) * 8;
  for (int i = threadIdx.x; i < 72 * 32; i += 256) {
    if constexpr (CI4) {
      const int k = i & 3, c = (i >> 2) & 31, q = (i >> 7) & 1, tap = i >> 8;     // 512 contiguous bytes per (tap, quad)
      const int ci = ci0 + 4 * q + k, co = co0 + c;
      t[tap * 8 + 4 * q + k][c] =
          (ci < Cin && co < Cout) ? packed[(((int64_t)tap * (Cin >> 2) + (ci >> 2)) * Cout + co) * 4 + k] : 0.f;
    } else {
      const int row = i >> 5, c = i & 31;          // row = tap*8 + ci_local
      const int tap = row >> 3, ci = ci0 + (row & 7), co = co0 + c;
      t[row][c] = (ci < Cin && co < Cout) ? packed[((int64_t)tap * Cin + ci) * Cout + co] : 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 72; i += 256) {
    const int c = i / 72, j = i - c * 72;        // j = ci_local*9 + tap  (OIHW order inside the tile)
    const int cil = j / 9, tap = j - cil * 9;
    const int ci = ci0 + cil, co = co0 + c;
    if (ci < cin_real && co < Cout) dw[((int64_t)co * cin_real + ci) * 9 + tap] = t[tap * 8 + cil][c] * us;
  }
}

template <int SL>
static int launch_wgrad_reduce_sl(const float* slab, int S, int Cin, int Cout, int cin_real, float* dw,
                                  const float* dbp, int ndb, float* db, hipStream_t s) {
  constexpr int ELEMS = 4 * (256 / SL);
  const int64_t nSlab = (int64_t)9 * Cin * Cout;
  const int64_t blocks = (nSlab + ELEMS - 1) / ELEMS + (db ? (Cout + 15) / 16 : 0);
  hipLaunchKernelGGL(k_wgrad_reduce<SL>, dim3((unsigned)blocks), dim3(256), 0, s, slab, S, Cin, Cout, cin_real, dw,
                     dbp, ndb, db, g_grad_unscale);
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_reduce(const float* slab, int S, int Cin, int Cout, int cin_real, float* dw, const float* dbp,
                        int ndb, float* db, hipStream_t s, bool ci4) {
  FU_REQUIRE(!ci4 || Cin % 4 == 0, "wgrad_reduce: the interleaved slab layout needs c_in %% 4 == 0");
  if (FU_EXP_SKIP(4)) return 0;
  int st;
  if (S >= 64) st = launch_wgrad_reduce_sl<16>(slab, S, Cin, Cout, cin_real, dw, dbp, ndb, db, s);
  else if (S >= 16) st = launch_wgrad_reduce_sl<4>(slab, S, Cin, Cout, cin_real, dw, dbp, ndb, db, s);
  else st = launch_wgrad_reduce_sl<1>(slab, S, Cin, Cout, cin_real, dw, dbp, ndb, db, s);
  if (st) return st;
  const int blocks = ceil_div(Cout, 32) * ceil_div(Cin, 8);
  if (ci4) hipLaunchKernelGGL(k_wgrad_transpose<true>, dim3(blocks), dim3(256), 0, s, slab, Cin, Cout, cin_real, dw, g_grad_unscale);
  else hipLaunchKernelGGL(k_wgrad_transpose<false>, dim3(blocks), dim3(256), 0, s, slab, Cin, Cout, cin_real, dw, g_grad_unscale);
  FU_LAUNCH_CHECK();
  return 0;
}

void wgrad_split_shared(int Cin, int Cout, int B, int H, int W, int* nPix, int* S, int* perSplit);
static inline void wgrad_split(int Cin, int Cout, int B, int H, int W, int* nPix, int* S, int* perSplit) {
  wgrad_split_shared(Cin, Cout, B, H, W, nPix, S, perSplit);
}
void wgrad_split_shared(int Cin, int Cout, int B, int H, int W, int* nPix, int* S, int* perSplit) {
  const int np = B * ceil_div(H, WG_PTH) * ceil_div(W, WG_PTW);
  const int nT = ceil_div(Cin, WG_CT) * ceil_div(Cout, WG_CT);
  int s = ceil_div(512, nT);
  if (s > np) s = np;
  if (s < 1) s = 1;
  const int per = ceil_div(np, s);
  *nPix = np; *perSplit = per; *S = ceil_div(np, per);
}

int64_t conv3x3_wgrad_slab_elems_f32(int Cin, int Cout, int B, int H, int W) {
  int np, S, per;
  wgrad_split(Cin, Cout, B, H, W, &np, &S, &per);
  return (int64_t)S * 9 * Cin * Cout;
}

int launch_conv3x3_wgrad_f32(const ConvIn& in, const float* dy, int Cout, float* slab, float* dw_oihw, int cin_real,
                             const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                             hipStream_t s) {
  WgP P;
  P.src0 = (const float*)in.src0; P.src1 = (const float*)in.src1; P.a0 = in.a0; P.b0 = in.b0; P.dy = dy; P.slab = slab;
  P.C0 = in.C0; P.C1 = in.src1 ? in.C1 : 0; P.Cin = P.C0 + P.C1; P.Cout = Cout; P.B = B; P.H = H; P.W = W;
  FU_REQUIRE(P.C0 % 4 == 0 && P.C1 % 4 == 0 && Cout % 4 == 0, "wgrad_f32: channel counts must be multiples of 4");
  P.tilesX = ceil_div(W, WG_PTW); P.tilesY = ceil_div(H, WG_PTH);
  wgrad_split(P.Cin, Cout, B, H, W, &P.nPix, &P.S, &P.perSplit);
  P.nCi = ceil_div(P.Cin, WG_CT); P.nCo = ceil_div(Cout, WG_CT);
  const int grid = P.nCi * P.nCo * P.S;
  const size_t sh = (size_t)(WG_NHP * WG_CT + 64 * WG_CT) * sizeof(float);
  const ProfSlot ps = in.opt.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL(k_wgrad_f32, dim3(grid), dim3(256), sh, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return launch_wgrad_reduce(slab, P.S, P.Cin, Cout, cin_real, dw_oihw, db_partials, n_db_partials, db, s, false);
}

}  // namespace fu
