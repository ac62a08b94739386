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
            v.x = fmaxf(av.x * v.x + bv.x, 0.f);
            v.y = fmaxf(av.y * v.y + bv.y, 0.f);
            v.z = fmaxf(av.z * v.z + bv.z, 0.f);
            v.w = fmaxf(av.w * v.w + bv.w, 0.f);
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
  const ProfSlot ps = g_prof_slot;
  g_prof_slot = ProfSlot();
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
              v.x = fmaxf(av.x * v.x + bv.x, 0.f);
              v.y = fmaxf(av.y * v.y + bv.y, 0.f);
              v.z = fmaxf(av.z * v.z + bv.z, 0.f);
              v.w = fmaxf(av.w * v.w + bv.w, 0.f);
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
// block = 64 consecutive outputs (co fastest -> coalesced slab reads) x 4 split lanes; each lane sums every 4th
// split with 4 independent fp64 accumulators, lanes meet in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int S, int Cin, int Cout,
                                                      int cin_real, float* __restrict__ dw,
                                                      const float* __restrict__ dbp, int ndb, float* __restrict__ db) {
  __shared__ double sm[4][64];
  const int64_t nW = (int64_t)9 * cin_real * Cout;
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + o;
  double acc = 0.0;
  int64_t out_index = -1;
  float* out_ptr = nullptr;
  if (idx < nW) {
    const int co = (int)(idx % Cout);
    const int64_t r = idx / Cout;
    const int ci = (int)(r % cin_real);
    const int tap = (int)(r / cin_real);
    const int64_t sstride = (int64_t)9 * Cin * Cout;
    const float* p = slab + ((int64_t)tap * Cin + ci) * Cout + co;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int s = sl;
    for (; s + 12 < S; s += 16) {
      a0 += (double)p[(int64_t)s * sstride];
      a1 += (double)p[(int64_t)(s + 4) * sstride];
      a2 += (double)p[(int64_t)(s + 8) * sstride];
      a3 += (double)p[(int64_t)(s + 12) * sstride];
    }
    for (; s < S; s += 4) a0 += (double)p[(int64_t)s * sstride];
    acc = (a0 + a1) + (a2 + a3);
    out_index = ((int64_t)co * cin_real + ci) * 9 + tap;
    out_ptr = dw;
  } else if (db && idx < nW + Cout) {
    const int co = (int)(idx - nW);
    for (int i = sl; i < ndb; i += 4) acc += (double)dbp[(int64_t)i * Cout + co];
    out_index = co;
    out_ptr = db;
  }
  sm[sl][o] = acc;
  __syncthreads();
  if (sl == 0 && out_ptr) out_ptr[out_index] = (float)((sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]));
}

int launch_wgrad_reduce(const float* slab, int S, int Cin, int Cout, int cin_real, float* dw, const float* dbp,
                        int ndb, float* db, hipStream_t s) {
  const int64_t nOut = (int64_t)9 * cin_real * Cout + (db ? Cout : 0);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((nOut + 63) / 64)), dim3(256), 0, s, slab, S, Cin, Cout,
                     cin_real, dw, dbp, ndb, db);
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
  const ProfSlot ps = g_prof_slot;
  g_prof_slot = ProfSlot();
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL(k_wgrad_f32, dim3(grid), dim3(256), sh, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return launch_wgrad_reduce(slab, P.S, P.Cin, Cout, cin_real, dw_oihw, db_partials, n_db_partials, db, s);
}

}  // namespace fu
