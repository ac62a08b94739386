// bf16 3x3 convolutions for gfx950: im2col-free implicit GEMMs on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//
//   forward / dgrad : out[p][n] = sum_{tap,c} in[p+tap][c] * w[tap][n][c]
//       workgroup tile = (WM*64) output pixels (TH x 16) x (WN*64) channels, K chunks of 32 input channels;
//       each wave owns 64 px x 64 ch = 2x2 MFMA tiles (64 accumulator registers).  The halo tile of the virtual
//       two-source NHWC input is staged once per chunk as [pixel][32ch + 8 pad] (80-byte rows: conflict-free
//       ds_read_b128 fragments) with the producer's BatchNorm+ReLU applied on the way (fp32 math, one rounding to
//       bf16); weights as [tap][n][32ch + 8 pad].  Next chunk's global loads are issued before the MFMA block.
//       Epilogue: bias, bf16 NHWC stores (two destinations for dgrad), fp32 (sum, sumsq) partials per tile.
//   wgrad : dW[tap][c][n] = sum_p in[p+tap][c] * dy[p][n]
//       M = 64 c_in, N = 64 c_out, K = pixels.  Both operands need K (= pixels) contiguous per lane while memory is
//       channel-contiguous: the tiles are staged untransposed ([pixel][64ch + 32 pad], 192-byte rows) and the
//       fragments are fetched with the hardware transposing read ds_read_b64_tr_b16.  9 accumulator tiles per
//       wave (one per tap); split-K slabs in fp32, reduced by the shared fixed-order kernel.
#include "fu_common.h"
#include "fu_conv_bf16.h"

#include <stdlib.h>
#include <type_traits>

namespace fu {


// ------------------------------------------------------------------------------------------------
// weight packing (bf16): OIHW fp32 -> wf[tap][co][ci_pad] and wd[8-tap][ci_pad][co]
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_bf16(const float* __restrict__ w, int Cout, int cin_real, int cin_pad,
                            bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % cin_pad);
    const int64_t r = idx / cin_pad;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    const float v = ci < cin_real ? w[((int64_t)co * cin_real + ci) * 9 + tap] : 0.f;
    const bf16_t h = f2e(v);
    if (wf) wf[idx] = h;                                                    // [tap][co][ci]
    if (wd) wd[((int64_t)(8 - tap) * cin_pad + ci) * Cout + co] = h;        // [8-tap][ci][co]
  }
}

int launch_pack_conv3x3_bf16(const float* w_oihw, int Cout, int cin_real, int cin_pad, bf16_t* wfwd, bf16_t* wdgrad,
                             hipStream_t s) {
  const int64_t total = (int64_t)9 * cin_pad * Cout;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_pack_bf16, dim3(g), dim3(256), 0, s, w_oihw, Cout, cin_real, cin_pad, wfwd, wdgrad, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------------------

template <int WM, int WN, int NTW>
struct BCfg {
  // WM x WN waves; each wave owns 64 pixels x (32*NTW) channels
  static constexpr int TW = 16, TH = 4 * WM, BN = 32 * NTW * WN, KC = 32, KCP = 40;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int HWd = TW + 2, HHt = TH + 2, NHP = HHt * HWd;
  static constexpr int A_UNITS = NHP * 4, W_UNITS = 9 * BN * 4;   // 16-byte units (8 channels)
  static constexpr int A_ITERS = (A_UNITS + NT - 1) / NT, W_ITERS = (W_UNITS + NT - 1) / NT;
  static constexpr int AB_FLOATS = 2 * 1024;                      // BN scale/shift of source 0 (C0 <= 1024)
  static constexpr int SMEM_BYTES = (NHP + 9 * BN) * KCP * 2 + AB_FLOATS * 4;
};

__device__ __forceinline__ uint4 bn_relu_pack8(uint4 v, const float4& a0, const float4& a1, const float4& b0,
                                               const float4& b1) {
  float x[8];
  x[0] = e2f_lo(v.x); x[1] = e2f_hi(v.x);
  x[2] = e2f_lo(v.y); x[3] = e2f_hi(v.y);
  x[4] = e2f_lo(v.z); x[5] = e2f_hi(v.z);
  x[6] = e2f_lo(v.w); x[7] = e2f_hi(v.w);
  x[0] = bn_act_fused(a0.x, x[0], b0.x); x[1] = bn_act_fused(a0.y, x[1], b0.y);
  x[2] = bn_act_fused(a0.z, x[2], b0.z); x[3] = bn_act_fused(a0.w, x[3], b0.w);
  x[4] = bn_act_fused(a1.x, x[4], b1.x); x[5] = bn_act_fused(a1.y, x[5], b1.y);
  x[6] = bn_act_fused(a1.z, x[6], b1.z); x[7] = bn_act_fused(a1.w, x[7], b1.w);
  uint4 o;
  o.x = (unsigned)f2e(x[0]) | ((unsigned)f2e(x[1]) << 16);
  o.y = (unsigned)f2e(x[2]) | ((unsigned)f2e(x[3]) << 16);
  o.z = (unsigned)f2e(x[4]) | ((unsigned)f2e(x[5]) << 16);
  o.w = (unsigned)f2e(x[6]) | ((unsigned)f2e(x[7]) << 16);
  return o;
}

template <int WM, int WN, int NTW>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3x3_bf16(BConvP P) {
  using Cfg = BCfg<WM, WN, NTW>;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, KC = Cfg::KC, KCP = Cfg::KCP, NT = Cfg::NT;
  constexpr int HWd = Cfg::HWd, NHP = Cfg::NHP, A_ITERS = Cfg::A_ITERS, W_ITERS = Cfg::W_ITERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);   // [NHP][KCP]
  bf16_t* sW = sA + NHP * KCP;                        // [9][BN][KCP]
  float* sAB = reinterpret_cast<float*>(sW + 9 * BN * KCP);  // [2][1024] BN scale / shift of source 0

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int coT = logical / P.nPix;
  const int pixT = logical - coT * P.nPix;
  const int tx = pixT % P.tilesX;
  const int t2 = pixT / P.tilesX;
  const int ty = t2 % P.tilesY;
  const int bb = t2 / P.tilesY;
  const int x0 = tx * TW, y0 = ty * TH, n0 = coT * BN;

  const bool has_bn = P.a0 != nullptr;
  if (has_bn) {
    for (int c = tid; c < P.C0; c += NT) { sAB[c] = P.a0[c]; sAB[1024 + c] = P.b0[c]; }
  }

  // ---- staging descriptors: loads are UNCONDITIONAL from clamped addresses (no branch, no early wait);
  //      masking to zero happens when the registers are written to LDS -----------------------------------
  const int aq = tid & 3;  // channel octet inside the chunk (NT % 4 == 0)
  int a_pix[A_ITERS];
  unsigned a_okmask = 0;
  static_for<0, A_ITERS>([&](auto I) {
    constexpr int it = decltype(I)::value;
    const int u = tid + it * NT;
    const int hp = u >> 2;
    const int hy = hp / HWd, hx = hp - hy * HWd;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = (u < Cfg::A_UNITS) && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    a_okmask |= ok ? (1u << it) : 0u;
    a_pix[it] = ok ? ((bb * P.H + iy) * P.W + ix) : 0;
  });
  int w_off[W_ITERS];
  unsigned w_okmask = 0;
  static_for<0, W_ITERS>([&](auto I) {
    constexpr int it = decltype(I)::value;
    const int u = tid + it * NT;
    const int q = u & 3;
    const int co = (u >> 2) % BN;
    const int tap = u / (4 * BN);
    const int n = n0 + co;
    const bool ok = (u < Cfg::W_UNITS) && n < P.N;
    w_okmask |= ok ? (1u << it) : 0u;
    w_off[it] = ok ? ((tap * P.N + n) * P.Cin + 8 * q) : 0;
  });
  uint4 ra[A_ITERS];
  uint4 rw[W_ITERS];

  auto load_chunk = [&](int k0) {
    const int c = k0 + 8 * aq;
    const bool cval = c < P.Cin;
    const bool from0 = c < P.C0;
    const bf16_t* base = (from0 || !cval) ? P.src0 : P.src1;
    const int cs = (from0 || !cval) ? P.C0 : P.C1;
    const int cc = !cval ? 0 : (from0 ? c : c - P.C0);
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      ra[it] = *reinterpret_cast<const uint4*>(base + (int64_t)a_pix[it] * cs + cc);
    });
    // weights: ci = k0 + 8q; the last chunk of a ragged Cin is clamped to offset 0 and masked at store time
    static_for<0, W_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int q = (tid + it * NT) & 3;
      const bool wv = k0 + 8 * q < P.Cin;
      rw[it] = *reinterpret_cast<const uint4*>(P.wpk + (wv ? (int64_t)w_off[it] + k0 : 0));
    });
  };

  auto store_chunk = [&](int k0) {
    const int c = k0 + 8 * aq;
    const bool cval = c < P.Cin;
    const bool bn = has_bn && c < P.C0;
    float4 av0, av1, bv0, bv1;
    if (bn) {
      av0 = *reinterpret_cast<const float4*>(sAB + c);
      av1 = *reinterpret_cast<const float4*>(sAB + c + 4);
      bv0 = *reinterpret_cast<const float4*>(sAB + 1024 + c);
      bv1 = *reinterpret_cast<const float4*>(sAB + 1024 + c + 4);
    }
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      if (u < Cfg::A_UNITS) {
        uint4 v = ra[it];
        if (bn) v = bn_relu_pack8(v, av0, av1, bv0, bv1);
        const bool keep = cval && ((a_okmask >> it) & 1u);   // component-wise: a ?: on uint4 lvalues would take
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u;         // addresses and demote the arrays to scratch
        v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        *reinterpret_cast<uint4*>(sA + (u >> 2) * KCP + 8 * aq) = v;
      }
    });
    static_for<0, W_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      if (u < Cfg::W_UNITS) {
        const bool wv = (k0 + 8 * (u & 3) < P.Cin) && ((w_okmask >> it) & 1u);
        uint4 v = rw[it];
        v.x = wv ? v.x : 0u; v.y = wv ? v.y : 0u; v.z = wv ? v.z : 0u; v.w = wv ? v.w : 0u;
        *reinterpret_cast<uint4*>(sW + (u >> 2) * KCP + 8 * (u & 3)) = v;
      }
    });
  };

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment base offsets (bf16 elements)
  int aoff[2], boff[NTW];
  // m-tile = 2 image rows x 16 columns.  Lanes 16..31 (second row) take their columns ROTATED by HWd mod 16:
  // the halo pitch (18 pixels) would otherwise put rows 12..15 of the first image row and 4..11 of the second on the
  // same LDS bank slots inside every 16-lane ds_read_b128 group (2-way conflict on each A read; measured 38 % of the
  // LDS-active cycles).  With the rotation the 16 lanes of a group hit 16 distinct slots.
  const int mrow = l31 >> 4;
  const int mcol = mrow ? ((l31 - 16 - (HWd & 15)) & 15) : l31;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    aoff[mt] = (((wm * 2 + mt) * 2 + mrow) * HWd + mcol) * KCP + 8 * lh;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) boff[nt] = (wn * 32 * NTW + nt * 32 + l31) * KCP + 8 * lh;

  const int nChunks = (P.Cin + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < nChunks; ++ch) {
    __syncthreads();            // previous chunk's fragment reads are done (and sAB is visible on the first pass)
    store_chunk(ch * KC);
    __syncthreads();
    if (ch + 1 < nChunks) load_chunk((ch + 1) * KC);   // raw loads stay in flight under the MFMA block
    // 18 k-steps (9 taps x 2 halves of the 32-channel chunk), software-pipelined by hand: the fragments of step
    // s+1 are requested from LDS before the MFMAs of step s are issued (hipcc otherwise issues each step's
    // ds_reads just in time and exposes one LDS latency per 4 MFMAs).
    frag8_t af[2][2], bfr[2][NTW];
    auto load_frags = [&](auto Sc, auto Bc) {
      constexpr int st = decltype(Sc)::value, buf = decltype(Bc)::value;
      constexpr int tap = st >> 1, ks = st & 1;
      constexpr int toff = ((tap / 3) * HWd + (tap % 3)) * KCP + ks * 16;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *reinterpret_cast<const frag8_t*>(sA + aoff[mt] + toff);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        bfr[buf][nt] = *reinterpret_cast<const frag8_t*>(sW + tap * BN * KCP + boff[nt] + ks * 16);
    };
    load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    static_for<0, 18>([&](auto S) {
      constexpr int st = decltype(S)::value, buf = st & 1;
      if constexpr (st + 1 < 18) {
        load_frags(std::integral_constant<int, st + 1>{}, std::integral_constant<int, buf ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = FU_MFMA32(af[buf][mt], bfr[buf][nt], acc[mt][nt]);
    });
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  // The accumulator holds one channel per lane and 16 pixels in registers; a 2-byte store per value would make the
  // epilogue store-issue bound.  Two DPP exchanges inside each lane quad (xor 1, then xor 2) transpose 4 pixels x
  // 4 channels so that every lane owns 4 consecutive channels of ONE pixel: one 8-byte store per 4 registers,
  // each wave instruction writing 8 pixels x 64 contiguous bytes.
  float ssum[NTW], ssq[NTW];
  const int qj = l31 & 3;
  const bool q_even = !(l31 & 1), q_lo = qj < 2;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    ssum[nt] = 0.f; ssq[nt] = 0.f;
    const int n = n0 + wn * 32 * NTW + nt * 32 + l31;
    const bool nok = n < P.N;
    const float bias = (P.bias && nok) ? P.bias[n] : 0.f;
    const int nq = n & ~3;                       // first channel of this lane quad (N, D0 are multiples of 8)
    bf16_t* dst;
    int dstride, dn;
    if (nq < P.D0) { dst = P.dst0; dstride = P.D0; dn = nq; }
    else { dst = P.dst1; dstride = P.D1; dn = nq - P.D0; }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = acc[mt][nt][4 * g + k];
          const int p = k + 8 * g + 4 * lh;                       // MFMA row -> pixel (second row rotated, see aoff)
          const int oy = y0 + (wm * 2 + mt) * 2 + (p >> 4);
          const int ox = x0 + ((p >> 4) ? ((p - 16 - (HWd & 15)) & 15) : p);
          if (nok && oy < P.H && ox < P.W) { ssum[nt] += a; ssq[nt] += a * a; }
          v[k] = a + bias;
        }
        // level 1: pairs of channels
        const float s01 = q_even ? v[1] : v[0];
        const float s23 = q_even ? v[3] : v[2];
        const float r01 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s01), 0xB1, 0xF, 0xF, true));
        const float r23 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s23), 0xB1, 0xF, 0xF, true));
        const unsigned A = q_even ? ((unsigned)f2e(v[0]) | ((unsigned)f2e(r01) << 16))
                                  : ((unsigned)f2e(r01) | ((unsigned)f2e(v[1]) << 16));
        const unsigned Bq = q_even ? ((unsigned)f2e(v[2]) | ((unsigned)f2e(r23) << 16))
                                   : ((unsigned)f2e(r23) | ((unsigned)f2e(v[3]) << 16));
        // level 2: pairs of channel pairs
        const unsigned send = q_lo ? Bq : A;
        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);
        uint2 o;
        o.x = q_lo ? A : recv;
        o.y = q_lo ? recv : Bq;
        const int p = qj + 8 * g + 4 * lh;       // this lane now owns pixel row qj of the register quad
        const int oy = y0 + (wm * 2 + mt) * 2 + (p >> 4);
        const int ox = x0 + ((p >> 4) ? ((p - 16 - (HWd & 15)) & 15) : p);
        if (nq < P.N && oy < P.H && ox < P.W)
          *reinterpret_cast<uint2*>(dst + (((int64_t)bb * P.H + oy) * P.W + ox) * dstride + dn) = o;
      }
    }
  }
  if (P.stats) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [WM][BN][2]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      ssum[nt] += __shfl_xor(ssum[nt], 32, 64);
      ssq[nt] += __shfl_xor(ssq[nt], 32, 64);
    }
    __syncthreads();
    if (lh == 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int cn = wn * 32 * NTW + nt * 32 + l31;
        red[(wm * BN + cn) * 2 + 0] = ssum[nt];
        red[(wm * BN + cn) * 2 + 1] = ssq[nt];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < P.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int m = 0; m < WM; ++m) { s += red[(m * BN + tid) * 2 + 0]; q += red[(m * BN + tid) * 2 + 1]; }
      float* o = P.stats + ((int64_t)pixT * P.N + n0 + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

int conv3x3_num_stat_tiles_bf16(int B, int H, int W) { return B * ceil_div(H, 16) * ceil_div(W, 16); }

template <int WM, int WN, int NTW>
static int launch_cfg(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  using Cfg = BCfg<WM, WN, NTW>;
  P.tilesX = ceil_div(P.W, Cfg::TW); P.tilesY = ceil_div(P.H, Cfg::TH);
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = ceil_div(P.N, Cfg::BN);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16<WM, WN, NTW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL((k_conv3x3_bf16<WM, WN, NTW>), dim3(P.nPix * P.nCo), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

#if FU_HALF      // the hooks live in the bf16 objects; the fp16 kernels obey the same switches
extern int g_bf16_force_general, g_bf16_force_full_taps, g_wgrad_force_lockstep;
extern unsigned long long* g_conv_dbg;
#else
int g_bf16_force_cfg = -1;  // testing hook: 0 = 256x64 tile, 2 = 256x32 tile
int g_bf16_force_general = 0;   // testing hook (fu_test_force_general_conv): skip the aligned-shape fast kernel
int g_bf16_force_full_taps = 0; // testing hook (fu_test_force_full_taps): embedded 1x1 convs run all nine taps
unsigned long long* g_conv_dbg = nullptr;
#endif

int launch_conv3x3_bf16(const ConvIn& in, const bf16_t* wpk, const float* bias, bf16_t* dst0, int D0, bf16_t* dst1,
                        int D1, float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s) {
  BConvP P;
  P.src0 = (const bf16_t*)in.src0; P.src1 = (const bf16_t*)in.src1; P.a0 = in.a0; P.b0 = in.b0;
  P.wpk = wpk; P.bias = bias; P.dst0 = dst0; P.dst1 = dst1; P.stats = stats;
  P.C0 = in.C0; P.C1 = in.src1 ? in.C1 : 0; P.Cin = P.C0 + P.C1; P.N = D0 + D1; P.D0 = D0; P.D1 = D1;
  P.B = B; P.H = H; P.W = W;
  P.dbg = g_conv_dbg;
  P.center_only = (in.center_only && !g_bf16_force_full_taps) ? 1 : 0;
  P.bnb_y = nullptr; P.bnb_a = P.bnb_b = P.bnb_mean = P.bnb_invstd = nullptr; P.bnb_part = nullptr;
  FU_REQUIRE(P.C0 % 8 == 0 && P.C1 % 8 == 0, "conv3x3_bf16: input channel counts must be multiples of 8 (C0=%d C1=%d)",
             P.C0, P.C1);
  FU_REQUIRE(P.a0 == nullptr || P.C0 <= 1024, "conv3x3_bf16: at most 1024 BN-activated channels in source 0 (got %d)",
             P.C0);   // (the LDS table of BN coefficients; a plain source has no such limit)
  // tile choice (all tiles are 16x16 = 256 output pixels).  Measured on MI355X (profiles/): two 4-wave workgroups
  // per CU (256x64 tile, 80 KB LDS) overlap one group's LDS staging with the other's MFMA block and beat the
  // 8-wave 256x128 tile (higher FLOP/byte but lock-step phases) on every layer that yields >= 512 workgroups;
  // 256x32 keeps the small deep levels at >= 256 workgroups.  (Also tried and measured slower, hence not built: the
  // 8-wave 256x128 tile with single or double-buffered 16-channel LDS stages (810-900 TF where this one reaches 870-1040)
  // and a warp-specialised 4 loader + 4 compute wave version with two LDS stages (790-915 TF).  Removing the per-chunk
  // staging altogether lets the same MFMA loop run at 1200-1430 TF, so staging costs ~30 % on the deep layers.)
  if (!g_bf16_force_general && conv3x3_bf16_fast_eligible(P)) {
    const int st = launch_conv3x3_bf16_fast(P, in.opt, s);
    if (n_stat_tiles) *n_stat_tiles = P.nPix;
    return st;
  }
  const int64_t t256 = (int64_t)B * ceil_div(H, 16) * ceil_div(W, 16);
  int cfg;
  if (P.N >= 64 && t256 * ceil_div(P.N, 64) >= 512) cfg = 0;
  else cfg = 2;
  if (g_bf16_force_cfg == 0 || g_bf16_force_cfg == 2) cfg = g_bf16_force_cfg;
  int st;
  if (cfg == 0) st = launch_cfg<4, 1, 2>(P, in.opt, s);
  else st = launch_cfg<4, 1, 1>(P, in.opt, s);
  if (n_stat_tiles) *n_stat_tiles = P.nPix;
  return st;
}

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
struct BWgP {
  const bf16_t* src0; const bf16_t* src1; const float* a0; const float* b0; const bf16_t* dy;
  float* slab;
  int C0, C1, Cin, Cout, B, H, W, tilesX, tilesY, nPix, nCi, nCo, S, perSplit;
  unsigned rcp_tilesX, rcp_tilesY;   // k_wgrad_bf16_pp<true>: floor(2^32 / d) + 1 (0 for d == 1), as in BConvP
  unsigned long long* dbg;   // FU_CONV_STAMPS builds: per-workgroup phase sums (tools/stamp_wgrad.py)
};

// Two transposing reads -> one MFMA fragment.  NOTE (hipcc / ROCm 7.2): the v4i16 form of the builtin followed by
// per-element bit casts to __bf16 is miscompiled (element 0 is replicated); the v4bf16 form + shufflevector is correct
// (checked on hardware, tools/probes/tr_probe3.hip).
__device__ __forceinline__ frag8_t tr_frag(const bf16_t* p0, const bf16_t* p1) {
  typedef __attribute__((address_space(3))) tr4_t lds_tr4;
  const tr4_t v = FU_TR16((lds_tr4*)p0);
  const tr4_t w = FU_TR16((lds_tr4*)p1);
  return __builtin_bit_cast(frag8_t, __builtin_shufflevector(v, w, 0, 1, 2, 3, 4, 5, 6, 7));
}

// WMI waves along c_in (32 each) x 2 waves along c_out (32 each); pixel stage = PTH x 16 pixels with halo.
template <int WMI, int PTH>
struct WCfg {
  static constexpr int CI_T = 32 * WMI, CO_T = 64, NT = 128 * WMI, PTW = 16;
  static constexpr int HWd = PTW + 2, NHP = (PTH + 2) * HWd, NPX = PTH * PTW;
  static constexpr int RSX = CI_T + 32, RSD = CO_T + 32;    // row strides (elements): 64-byte residue mod 256 B
  static constexpr int XQ = CI_T / 8, DQ = CO_T / 8;        // 16-byte units per pixel row
  static constexpr int X_UNITS = NHP * XQ, D_UNITS = NPX * DQ;
  static constexpr int X_ITERS = (X_UNITS + NT - 1) / NT, D_ITERS = (D_UNITS + NT - 1) / NT;
  static constexpr int SMEM_BYTES = (NHP * RSX + NPX * RSD) * 2 + 2 * CI_T * 4;
};

// TAPS = 9: the 3x3 weight gradient.  TAPS = 1: only its centre tap (the embedded 1x1 fusion convs of the late-fusion
// net): 8 MFMAs per stage instead of 72; the other eight tap slabs are left unwritten and must not be read.
template <int WMI, int PTH, int TAPS = 9>
__global__ __launch_bounds__(128 * WMI) void k_wgrad_bf16(BWgP P) {
  using Cfg = WCfg<WMI, PTH>;
  constexpr int CI_T = Cfg::CI_T, CO_T = Cfg::CO_T, NT = Cfg::NT, HWd = Cfg::HWd, NHP = Cfg::NHP, NPX = Cfg::NPX;
  constexpr int RSX = Cfg::RSX, RSD = Cfg::RSD, XQ = Cfg::XQ, DQ = Cfg::DQ;
  constexpr int X_ITERS = Cfg::X_ITERS, D_ITERS = Cfg::D_ITERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sX = reinterpret_cast<bf16_t*>(smem_raw);   // [NHP][RSX]
  bf16_t* sD = sX + NHP * RSX;                        // [NPX][RSD]
  float* sAB = reinterpret_cast<float*>(sD + NPX * RSD);  // [2][CI_T] BN scale / shift of this c_in tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int mi = wave >> 1, ni = wave & 1;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nT = P.nCi * P.nCo;
  const int split = logical / nT;
  const int t = logical - split * nT;
  const int ciT = t / P.nCo, coT = t - ciT * P.nCo;
  const int ci0 = ciT * CI_T, co0 = coT * CO_T;

  f32x16 acc[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // transposing-read lane roles: group g = lane>>4 -> channel block 16*(g&1), pixel half g>>1 (= lh);
  // within the group lane 4q+p supplies the address of pixel q, channels 4p..4p+3
  const int g = lane >> 4, gi = lane & 15, tq = gi >> 2, tp = gi & 3;
  const int tr_ch = 16 * (g & 1) + 4 * tp;
  const int tr_px = 8 * lh + tq;   // column inside the 16-pixel row (second read: +4)

  const bool has_bn = P.a0 != nullptr;
  if (has_bn) {
    for (int c = tid; c < CI_T; c += NT) {
      const int cc = ci0 + c;
      const bool ok = cc < P.C0;
      sAB[c] = ok ? P.a0[cc] : 1.f;
      sAB[CI_T + c] = ok ? P.b0[cc] : 0.f;
    }
  }
  // this thread's channel octets (NT % XQ == 0 and NT % DQ == 0)
  const int xq = tid % XQ, dq = tid % DQ;
  const int cX = ci0 + 8 * xq;
  const int cD = co0 + 8 * dq;
  const bool xval = cX < P.Cin, dval = cD < P.Cout;
  const bool from0 = cX < P.C0;
  const bool xbn = has_bn && from0 && xval;
  const bf16_t* xbase = (from0 || !xval) ? P.src0 : P.src1;
  const int xcs = (from0 || !xval) ? P.C0 : P.C1;
  const int xcc = !xval ? 0 : (from0 ? cX : cX - P.C0);
  const int dcc = dval ? cD : 0;

  uint4 rx[X_ITERS], rd[D_ITERS];
  unsigned xmask = 0, dmask = 0;

  auto load_tile = [&](int pt) {
    const int tx = pt % P.tilesX;
    const int t2 = pt / P.tilesX;
    const int bb = t2 / P.tilesY;
    const int y0 = (t2 % P.tilesY) * PTH;
    const int x0 = tx * Cfg::PTW;
    xmask = 0; dmask = 0;
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      const int hp = u / XQ;
      const int hy = hp / HWd, hx = hp - hy * HWd;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = (u < Cfg::X_UNITS) && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      xmask |= ok ? (1u << it) : 0u;
      const int pix = ok ? ((bb * P.H + iy) * P.W + ix) : 0;
      rx[it] = *reinterpret_cast<const uint4*>(xbase + (int64_t)pix * xcs + xcc);
    });
    static_for<0, D_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      const int p = u / DQ;
      const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
      const bool ok = (u < Cfg::D_UNITS) && oy < P.H && ox < P.W;
      dmask |= ok ? (1u << it) : 0u;
      const int pix = ok ? ((bb * P.H + oy) * P.W + ox) : 0;
      rd[it] = *reinterpret_cast<const uint4*>(P.dy + (int64_t)pix * P.Cout + dcc);
    });
  };
  auto store_tile = [&]() {
    float4 av0, av1, bv0, bv1;
    if (xbn) {
      av0 = *reinterpret_cast<const float4*>(sAB + 8 * xq);
      av1 = *reinterpret_cast<const float4*>(sAB + 8 * xq + 4);
      bv0 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq);
      bv1 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq + 4);
    }
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      if (u < Cfg::X_UNITS) {
        uint4 v = rx[it];
        if (xbn) v = bn_relu_pack8(v, av0, av1, bv0, bv1);
        const bool keep = xval && ((xmask >> it) & 1u);
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        *reinterpret_cast<uint4*>(sX + (u / XQ) * RSX + 8 * xq) = v;
      }
    });
    static_for<0, D_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int u = tid + it * NT;
      if (u < Cfg::D_UNITS) {
        uint4 v = rd[it];
        const bool keep = dval && ((dmask >> it) & 1u);
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        *reinterpret_cast<uint4*>(sD + (u / DQ) * RSD + 8 * dq) = v;
      }
    });
  };

  const int pt0 = split * P.perSplit;
  const int pt1 = min(P.nPix, pt0 + P.perSplit);
#ifdef FU_CONV_STAMPS
  unsigned long long tStage = 0, tIssue = 0, tMfma = 0, tW = 0, tS = 0, tA = __builtin_amdgcn_s_memtime(), tStart = tA;
#endif
  if (pt0 < pt1) load_tile(pt0);
  for (int pt = pt0; pt < pt1; ++pt) {
#ifdef FU_CONV_STAMPS
    tA = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();            // previous stage's fragment reads are done (sAB visible on the first pass)
#ifdef FU_CONV_STAMPS
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
    store_tile();
#ifdef FU_CONV_STAMPS
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long tB = __builtin_amdgcn_s_memtime();
    if (pt > pt0) { tW += t1 - tA; tS += t3 - t2; }
#endif
    // in flight under the MFMA block.  ALWAYS issued (the last stage re-reads its own tile): under `if (pt + 1 < pt1)`
    // the staging registers are phis of a loaded and a not-loaded path, hipcc copies some of them right behind the
    // loads and waits for them (vmcnt) in front of the MFMA block -- with one workgroup per CU nothing covers that
    load_tile(min(pt + 1, pt1 - 1));
#ifdef FU_CONV_STAMPS
    const unsigned long long tC = __builtin_amdgcn_s_memtime();
#endif
    // Walk the halo rows once: the X fragment of (halo row hr, column shift dx) feeds up to three taps
    // (dy = 0..2 with pixel row r = hr - dy), so every fragment is fetched from LDS once; dy fragments of the last
    // three pixel rows stay in a 4-deep register ring.  Next step's fragment is requested before this step's MFMAs.
    frag8_t Af[2], Bf[4];
    auto loadA = [&](auto Sc) {
      constexpr int st = decltype(Sc)::value, hr = st / 3, dx = st % 3;
      const bf16_t* ad = sX + (hr * HWd + tr_px + dx) * RSX + mi * 32 + tr_ch;
      Af[st & 1] = tr_frag(ad, ad + 4 * RSX);
    };
    auto loadB = [&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      const bf16_t* bd = sD + (r * 16 + tr_px) * RSD + ni * 32 + tr_ch;
      Bf[r & 3] = tr_frag(bd, bd + 4 * RSD);
    };
    if constexpr (TAPS == 9) {
      loadB(std::integral_constant<int, 0>{});
      loadA(std::integral_constant<int, 0>{});
      static_for<0, 3 * (PTH + 2)>([&](auto S) {
        constexpr int st = decltype(S)::value, hr = st / 3, dx = st % 3;
        if constexpr (st + 1 < 3 * (PTH + 2)) loadA(std::integral_constant<int, st + 1>{});
        if constexpr (dx == 0 && hr + 1 < PTH) loadB(std::integral_constant<int, hr + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 3>([&](auto DY) {
          constexpr int dy = decltype(DY)::value, r = hr - dy;
          if constexpr (r >= 0 && r < PTH)
            acc[dy * 3 + dx] = FU_MFMA32(Af[st & 1], Bf[r & 3], acc[dy * 3 + dx]);
        });
      });
    } else {
      // centre tap only: pixel row r pairs halo row r + 1, column shift 1 (stage index 3 (r + 1) + 1 of loadA)
      loadB(std::integral_constant<int, 0>{});
      loadA(std::integral_constant<int, 4>{});
      static_for<0, PTH>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (r + 1 < PTH) {
          loadA(std::integral_constant<int, 3 * (r + 2) + 1>{});
          loadB(std::integral_constant<int, r + 1>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = FU_MFMA32(Af[(3 * (r + 1) + 1) & 1], Bf[r & 3], acc[0]);
      });
    }
#ifdef FU_CONV_STAMPS
    const unsigned long long tD = __builtin_amdgcn_s_memtime();
    if (pt > pt0) { tStage += tB - tA; tIssue += tC - tB; tMfma += tD - tC; }
#endif
  }
#ifdef FU_CONV_STAMPS
  const unsigned long long tE = __builtin_amdgcn_s_memtime();
#endif
  // slab[split][tap][Cin / 4][Cout][4] (see k_wgrad_transpose<true>): accumulator registers 4j .. 4j+3 of a lane are c_in
  // 8j + 4 lh + {0..3} of its c_out -- one 16-byte store, 512 contiguous bytes per 32 lanes
  const int co = co0 + ni * 32 + l31;
  if (co < P.Cout) {
    const int cq = P.Cin >> 2;                       // Cin % 8 == 0 (launch_conv3x3_wgrad_bf16)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int tap = TAPS == 9 ? t : 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + mi * 32 + 8 * j + 4 * lh;
        if (ci < P.Cin)
          *reinterpret_cast<float4*>(P.slab + ((((int64_t)split * 9 + tap) * cq + (ci >> 2)) * P.Cout + co) * 4) =
              make_float4(acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]);
      }
    }
  }
#ifdef FU_CONV_STAMPS
  if (P.dbg && lane == 0 && blockIdx.x < 256) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = tStage; d[1] = tIssue; d[2] = tMfma; d[3] = (unsigned long long)max(pt1 - pt0 - 1, 0);
    d[4] = __builtin_amdgcn_s_memtime() - tE; d[5] = __builtin_amdgcn_s_memtime() - tStart; d[6] = tW; d[7] = tS;
  }
#endif
}


// ------------------------------------------------------------------------------------------------
// wgrad, ping-pong version for c_in tiles of 128 (8 waves, one workgroup per CU)
//
// k_wgrad_bf16<4,8> spends half of every stage with the MFMA pipe idle (tools/stamp_wgrad.py: per stage 4450 cycles of
// MFMA work for the two waves of a SIMD, 2400 BN/ReLU + LDS store, 1900 load issue at the texture unit's 64 B/clk, all
// in lock step because the single LDS stage needs two barriers).  Here the stage is double-buffered (rows unpadded and
// XOR-swizzled by 64-byte chunk instead, 2 x 62.5 KB) and the 8 waves form two groups half a stage apart: while waves
// 0-3 multiply stage n (one wave per SIMD feeds the MFMA pipe alone), waves 4-7 activate and store their half of the
// next stage and issue the loads after that, then the roles swap.  Every wave runs the same instruction stream
// { MFMA(n) ; barrier ; store(n+1+grp), load(n+2+grp) ; barrier }; group 1 is shifted by one pre-loop staging step and
// group 0 pays the matching barrier after the loop.  Accumulators, split-K slabs and the reduce are unchanged, so the
// results are bit-identical to k_wgrad_bf16<4,8>.
// ------------------------------------------------------------------------------------------------
struct WPCfg {
  static constexpr int CI_T = 128, CO_T = 64, NT = 512, PTH = 8, PTW = 16, GT = 256;
  static constexpr int HWd = PTW + 2, NHP = (PTH + 2) * HWd, NPX = PTH * PTW;
  static constexpr int RSX = CI_T, RSD = CO_T;               // unpadded rows (elements)
  static constexpr int XQ = CI_T / 8, DQ = CO_T / 8;
  static constexpr int XH_UNITS = NHP * XQ / 2, DH_UNITS = NPX * DQ / 2;   // per group
  static constexpr int X_ITERS = (XH_UNITS + GT - 1) / GT, D_ITERS = (DH_UNITS + GT - 1) / GT;
  static constexpr int BUF_ELEMS = NHP * RSX + NPX * RSD;
  static constexpr int SMEM_BYTES = 2 * BUF_ELEMS * 2 + 2 * CI_T * 4;
  static_assert((NHP * XQ) % 2 == 0 && XH_UNITS % XQ == 0 && DH_UNITS % DQ == 0, "halves split on row boundaries");
};

//
// FAST (round 2): the staging half of a stage was ~470 instructions per thread against 72 MFMAs of the other group -- the
// stage was issue-bound on tile decode (three integer divisions), per-slot bounds tests, 64-bit address products, per-slot
// select masks and one convert + one permute per CHANNEL.  With whole tiles (H % 8 == 0, W % 16 == 0), whole channel tiles
// (Cin % 128 == 0, Cout % 64 == 0) and < 2^24 pixels everything per slot is a thread constant: the byte offset of the slot
// from the tile's origin pixel (relX / relD), four bit sets naming the slots that fall off the image when the tile touches
// the top / bottom / left / right border, and LDS addresses that differ by immediates.  Per stage the decode is scalar
// (multiply-high by host reciprocals), a load is one add (+ the 64-bit base), interior tiles carry no masks at all, and
// the BatchNorm + ReLU converts channel PAIRS (v_cvt_pk with both operands).  Lanes of an un-normalised second source take
// a = 1, b = 0 and a NaN floor (v_max_f32 returns the other operand), so the stream has no divergent branch.  The
// arithmetic per element is unchanged: results are bit-identical to FAST = false, which keeps every other shape.
template <bool FAST>
__global__ __launch_bounds__(512) void k_wgrad_bf16_pp(BWgP P) {
  using Cfg = WPCfg;
  constexpr int CI_T = Cfg::CI_T, CO_T = Cfg::CO_T, GT = Cfg::GT, HWd = Cfg::HWd, NHP = Cfg::NHP;
  constexpr int PTH = Cfg::PTH, RSX = Cfg::RSX, RSD = Cfg::RSD, XQ = Cfg::XQ, DQ = Cfg::DQ;
  constexpr int X_ITERS = Cfg::X_ITERS, D_ITERS = Cfg::D_ITERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sBuf = reinterpret_cast<bf16_t*>(smem_raw);                       // 2 x { X [NHP][RSX], D [NPX][RSD] }
  float* sAB = reinterpret_cast<float*>(sBuf + 2 * Cfg::BUF_ELEMS);         // [2][CI_T]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & (GT - 1);
  // staging halves: group 1 stages the TOP half of a stage (halo rows 0-4, dy rows 0-3), group 0 the bottom half.  A group
  // stores its half of its next stage right before it multiplies that stage; the other half was stored by the other
  // group one interval earlier, i.e. in front of a barrier this group has already passed.  With the top half coming from
  // the OTHER group, the first fragments of the next MFMA phase (row 0) can be requested before the barrier that ends
  // the staging, and their LDS latency (the phase measured 2650 cycles for 2304 of MFMA work with an idle partner: the
  // pipe waits ~250 cycles for its first fragments) overlaps the barrier wait.
  const int hg = 1 - grp;
  const int mi = wave & 3;                 // c_in block of this wave; both c_out blocks: see ni below
  // waves 0-3 and 4-7 must split the MFMA work so that each group covers all (mi, ni): wave = 4*grp + w, w = 0..3
  // -> group g owns c_out block ni = g, all four c_in blocks
  const int ni = grp;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int nT = P.nCi * P.nCo;
  const int split = logical / nT;
  const int t = logical - split * nT;
  const int ciT = t / P.nCo, coT = t - ciT * P.nCo;
  const int ci0 = ciT * CI_T, co0 = coT * CO_T;

  f32x16 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  const int g = lane >> 4, gi = lane & 15, tq = gi >> 2, tp = gi & 3;
  const int tr_ch = 16 * (g & 1) + 4 * tp;
  const int tr_px = 8 * lh + tq;
  // swizzled fragment offsets (elements).  X: 64-byte chunk mi of row r sits at chunk mi ^ (r & 3); the rows of a
  // fragment are c + tr_px (+4) with c a compile-time constant, so four lane offsets cover c & 3 = 0..3.
  int aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aoff[j] = tr_px * RSX + ((mi ^ ((j + tr_px) & 3)) * 32) + tr_ch;
  // D: 128-byte rows, chunk ni of row r at chunk ni ^ ((r >> 1) & 1); rows are 16 r + tr_px (+4)
  const int boff = tr_px * RSD + ((ni ^ ((tr_px >> 1) & 1)) * 32) + tr_ch;

  const bool has_bn = P.a0 != nullptr;
  if (has_bn) {
    for (int c = tid; c < CI_T; c += Cfg::NT) {
      const int cc = ci0 + c;
      const bool ok = cc < P.C0;
      sAB[c] = ok ? P.a0[cc] : 1.f;
      sAB[CI_T + c] = ok ? P.b0[cc] : 0.f;
    }
  }
  const int xq = tid & (XQ - 1), dq = tid & (DQ - 1);
  const int cX = ci0 + 8 * xq;
  const int cD = co0 + 8 * dq;
  const bool xval = cX < P.Cin, dval = cD < P.Cout;
  const bool from0 = cX < P.C0;
  const bool xbn = has_bn && from0 && xval;
  const bf16_t* xbase = (from0 || !xval) ? P.src0 : P.src1;
  const int xcs = (from0 || !xval) ? P.C0 : P.C1;
  const int xcc = !xval ? 0 : (from0 ? cX : cX - P.C0);
  const int dcc = dval ? cD : 0;

  uint4 rx[X_ITERS], rd[D_ITERS];
  unsigned xmask = 0, dmask = 0;

  auto load_half = [&](auto pt) __attribute__((always_inline)) {      // (generic: only instantiated where used, i.e. for FAST = false)
    const int tx = pt % P.tilesX;
    const int t2 = pt / P.tilesX;
    const int bb = t2 / P.tilesY;
    const int y0 = (t2 % P.tilesY) * PTH;
    const int x0 = tx * Cfg::PTW;
    xmask = 0; dmask = 0;
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int ul = tg + it * GT;
      const int hp = (hg * Cfg::XH_UNITS + ul) / XQ;
      const int hy = hp / HWd, hx = hp - hy * HWd;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const unsigned ok = unsigned(ul < Cfg::XH_UNITS) & unsigned((unsigned)iy < (unsigned)P.H) &
                          unsigned((unsigned)ix < (unsigned)P.W);   // bitwise: no exec-mask branches
      xmask |= ok << it;
      const int pix = ok ? ((bb * P.H + iy) * P.W + ix) : 0;
      rx[it] = *reinterpret_cast<const uint4*>(xbase + (int64_t)pix * xcs + xcc);
    });
    static_for<0, D_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int ul = tg + it * GT;
      const int p = (hg * Cfg::DH_UNITS + ul) / DQ;
      const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
      const unsigned ok = unsigned(ul < Cfg::DH_UNITS) & unsigned(oy < P.H) & unsigned(ox < P.W);
      dmask |= ok << it;
      const int pix = ok ? ((bb * P.H + oy) * P.W + ox) : 0;
      rd[it] = *reinterpret_cast<const uint4*>(P.dy + (int64_t)pix * P.Cout + dcc);
    });
  };
  auto store_half = [&](auto* sX) __attribute__((always_inline)) {
    bf16_t* sD = sX + NHP * RSX;
    float4 av0, av1, bv0, bv1;
    if (xbn) {
      av0 = *reinterpret_cast<const float4*>(sAB + 8 * xq);
      av1 = *reinterpret_cast<const float4*>(sAB + 8 * xq + 4);
      bv0 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq);
      bv1 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq + 4);
    }
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int ul = tg + it * GT;
      if (ul < Cfg::XH_UNITS) {
        uint4 v = rx[it];
        // (the packed form of the fast conv kernel -- v_pk_fma_f32 / v_pk_max_i16 -- is 90 instructions shorter per
        // stage here and was measured 10 % SLOWER, same box: 8.30 -> 9.15 ms per 98 launches)
        if (xbn) v = bn_relu_pack8(v, av0, av1, bv0, bv1);
        const bool keep = xval && ((xmask >> it) & 1u);
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        const int hp = (hg * Cfg::XH_UNITS + ul) / XQ;
        *reinterpret_cast<uint4*>(sX + hp * RSX + 8 * (xq ^ ((hp & 3) << 2))) = v;
      }
    });
    static_for<0, D_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int ul = tg + it * GT;
      if (ul < Cfg::DH_UNITS) {
        uint4 v = rd[it];
        const bool keep = dval && ((dmask >> it) & 1u);
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        const int p = (hg * Cfg::DH_UNITS + ul) / DQ;
        *reinterpret_cast<uint4*>(sD + p * RSD + 8 * (dq ^ (((p >> 1) & 1) << 2))) = v;
      }
    });
  };

  // ---- FAST staging (see the kernel's header comment)
  constexpr int X_FULL = Cfg::XH_UNITS / GT;                 // slots every thread of the group owns (the last one is partial)
  unsigned relX[X_ITERS], relD[D_ITERS];                     // byte offsets from the tile's origin pixel (mod 2^32)
  unsigned mT = 0, mB = 0, mL = 0, mR = 0;                   // bit it: slot it lies in the halo row / column of that side
  unsigned xbad = 0;                                         // slots of the tile in registers that are outside the image
  bool xborder = false;                                      // (uniform) ... and whether there is any
  const unsigned cs2 = (unsigned)xcs * 2u;
  const char* xb = reinterpret_cast<const char*>(xbase + xcc);
  const char* db = reinterpret_cast<const char*>(P.dy + dcc);
  const float relu_floor = (has_bn && !from0) ? __builtin_nanf("") : 0.f;
  unsigned ldsX0 = 0, ldsD0 = 0;                             // LDS byte offsets of slot 0 inside a buffer
  uint4 rdf0, rdf1;                                          // the two dy units in flight (named: as an array they went to scratch)
  if constexpr (FAST) {
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int ul = tg + it * GT;
      const bool own = ul < Cfg::XH_UNITS;
      const int hp = (hg * Cfg::XH_UNITS + ul) / XQ;
      const int hy = hp / HWd, hx = hp - hy * HWd;
      relX[it] = own ? (unsigned)(((hy - 1) * P.W + (hx - 1)) * (int)cs2) : 0u;     // not owned: the origin pixel, never stored
      mT |= (own && hy == 0) ? (1u << it) : 0u;  mB |= (own && hy == PTH + 1) ? (1u << it) : 0u;
      mL |= (own && hx == 0) ? (1u << it) : 0u;  mR |= (own && hx == HWd - 1) ? (1u << it) : 0u;
    });
    static_for<0, D_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int p = (hg * Cfg::DH_UNITS + tg + it * GT) / DQ;
      relD[it] = (unsigned)(((p >> 4) * P.W + (p & 15)) * P.Cout * 2);
    });
    const int hp0 = (hg * Cfg::XH_UNITS + tg) / XQ;         // slot it: halo pixel hp0 + 16 it, same swizzle
    ldsX0 = (unsigned)(hp0 * RSX + 8 * (xq ^ ((hp0 & 3) << 2))) * 2u;
    const int p0 = (hg * Cfg::DH_UNITS + tg) / DQ;          // slot it: pixel p0 + 32 it, same swizzle
    ldsD0 = (unsigned)((NHP * RSX) + p0 * RSD + 8 * (dq ^ (((p0 >> 1) & 1) << 2))) * 2u;
  }
  auto load_half_fast = [&](int pt) __attribute__((always_inline)) {                        // pt uniform
    const int t2 = fast_div(pt, P.tilesX, P.rcp_tilesX);
    const int tx = pt - t2 * P.tilesX;
    const int bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
    const int y0 = (t2 - bb * P.tilesY) * PTH, x0 = tx * Cfg::PTW;
    const unsigned tile_pix = (unsigned)((bb * P.H + y0) * P.W + x0);
    const unsigned ft = y0 == 0 ? ~0u : 0u, fb = y0 + PTH == P.H ? ~0u : 0u;
    const unsigned fl = x0 == 0 ? ~0u : 0u, fr = x0 + Cfg::PTW == P.W ? ~0u : 0u;
    xborder = (ft | fb | fl | fr) != 0u;
    xbad = (mT & ft) | (mB & fb) | (mL & fl) | (mR & fr);
    const unsigned tX = (tile_pix & 0xffffffu) * (cs2 & 0xffffffu);                // v_mul_u32_u24 (eligibility: < 2^24 each)
    const char* dbt = db + (size_t)tile_pix * (size_t)(P.Cout * 2);                // uniform
    if (xborder) {
      const unsigned good = ~xbad;
      static_for<0, X_ITERS>([&](auto I) {
        constexpr int it = decltype(I)::value;
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)good, it, 1);      // bit it -> 0 / 0xffffffff
        rx[it] = *reinterpret_cast<const uint4*>(xb + (tX + (relX[it] & m)));      // off the image: the origin pixel
      });
    } else {
      static_for<0, X_ITERS>([&](auto I) {
        constexpr int it = decltype(I)::value;
        rx[it] = *reinterpret_cast<const uint4*>(xb + (tX + relX[it]));
      });
    }
    static_assert(D_ITERS == 2, "two dy units per thread and stage");
    rdf0 = *reinterpret_cast<const uint4*>(dbt + relD[0]);
    rdf1 = *reinterpret_cast<const uint4*>(dbt + relD[1]);
  };
  auto store_half_fast = [&](bf16_t* sXb, auto Mc, auto Bc) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(Mc)::value, BNR = decltype(Bc)::value;
    unsigned char* lx = reinterpret_cast<unsigned char*>(sXb) + ldsX0;
    unsigned char* ld = reinterpret_cast<unsigned char*>(sXb) + ldsD0;
    float4 av0, av1, bv0, bv1;
    if constexpr (BNR) {
      av0 = *reinterpret_cast<const float4*>(sAB + 8 * xq);
      av1 = *reinterpret_cast<const float4*>(sAB + 8 * xq + 4);
      bv0 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq);
      bv1 = *reinterpret_cast<const float4*>(sAB + CI_T + 8 * xq + 4);
    }
    const unsigned good = ~xbad;
    static_for<0, X_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < X_FULL || tg + it * GT < Cfg::XH_UNITS) {
        uint4 v = rx[it];
        if constexpr (BNR) {
          auto act = [&](unsigned w, float a_lo, float a_hi, float b_lo, float b_hi) {
            const float lo = __builtin_fmaxf(fmaf(a_lo, e2f_lo(w), b_lo), relu_floor);
            const float hi = __builtin_fmaxf(fmaf(a_hi, e2f_hi(w), b_hi), relu_floor);
            return pack_e2(f32x2{lo, hi});
          };
          v.x = act(v.x, av0.x, av0.y, bv0.x, bv0.y); v.y = act(v.y, av0.z, av0.w, bv0.z, bv0.w);
          v.z = act(v.z, av1.x, av1.y, bv1.x, bv1.y); v.w = act(v.w, av1.z, av1.w, bv1.z, bv1.w);
        }
        if constexpr (MASKED) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)good, it, 1);
          v.x &= m; v.y &= m; v.z &= m; v.w &= m;
        }
        *reinterpret_cast<uint4*>(lx + it * (16 * RSX * 2)) = v;
      }
    });
    *reinterpret_cast<uint4*>(ld) = rdf0;
    *reinterpret_cast<uint4*>(ld + 32 * RSD * 2) = rdf1;
  };
  auto load_half_any = [&](int pt) __attribute__((always_inline)) {
    if constexpr (FAST) load_half_fast(pt); else load_half(pt);
  };
  auto store_half_any = [&](bf16_t* sXb) __attribute__((always_inline)) {
    if constexpr (FAST) {
      if (xborder) { if (has_bn) store_half_fast(sXb, std::true_type{}, std::true_type{}); else store_half_fast(sXb, std::true_type{}, std::false_type{}); }
      else { if (has_bn) store_half_fast(sXb, std::false_type{}, std::true_type{}); else store_half_fast(sXb, std::false_type{}, std::false_type{}); }
    } else {
      store_half(sXb);
    }
  };
  // x fragments: a ring APD steps ahead of the MFMAs (depths 1-3 measured the same once the first fragments are in flight
  // before the phase starts)
#ifndef FU_WGRAD_APD
#define FU_WGRAD_APD 2
#endif
  constexpr int APD = FU_WGRAD_APD, NA = APD + 1, NST = 3 * (PTH + 2);
  static_assert(APD <= 3, "the fragments requested ahead of the barrier must lie in halo row 0");
  frag8_t Af[NA], Bf[4];
  // fragment addresses = one lane base per (buffer, row residue) + an immediate (ds offsets reach 64 KB, a buffer is 62.5 KB:
  // the second buffer gets its own bases; they are opaque to the compiler, which otherwise materialises base + constant
  // per fragment as loop invariants and spills them)
  int aoffp[2][4], boffp[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    aoffp[0][j] = aoff[j];
    aoffp[1][j] = aoff[j] + Cfg::BUF_ELEMS;
    asm volatile("" : "+v"(aoffp[0][j]), "+v"(aoffp[1][j]));
  }
  boffp[0] = boff + NHP * RSX;
  boffp[1] = boff + NHP * RSX + Cfg::BUF_ELEMS;
  asm volatile("" : "+v"(boffp[0]), "+v"(boffp[1]));
  auto loadA = [&](auto Par, auto Sc) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value, st = decltype(Sc)::value, hr = st / 3, dx = st % 3, c = hr * HWd + dx;
    const bf16_t* ad = sBuf + aoffp[par][c & 3] + c * RSX;
    Af[st % NA] = tr_frag(ad, ad + 4 * RSX);
  };
  auto loadB = [&](auto Par, auto Rc) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value, r = decltype(Rc)::value;
    const bf16_t* bd = sBuf + boffp[par] + r * 16 * RSD;
    Bf[r & 3] = tr_frag(bd, bd + 4 * RSD);
  };
  auto mfma_prefetch = [&](auto Par) __attribute__((always_inline)) {   // row 0 of x and of dy: staged by the other group (see hg)
    loadB(Par, std::integral_constant<int, 0>{});
    static_for<0, APD>([&](auto Sc) { loadA(Par, Sc); });
  };
  auto mfma_stage = [&](auto Par) __attribute__((always_inline)) {      // after mfma_prefetch(Par)
    static_for<0, NST>([&](auto S) {
      constexpr int st = decltype(S)::value, hr = st / 3, dx = st % 3;
      if constexpr (st + APD < NST) loadA(Par, std::integral_constant<int, st + APD>{});
      if constexpr (dx == 0 && hr + 1 < PTH) loadB(Par, std::integral_constant<int, hr + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 3>([&](auto DY) {
        constexpr int dy = decltype(DY)::value, r = hr - dy;
        if constexpr (r >= 0 && r < PTH)
          acc[dy * 3 + dx] = FU_MFMA32(Af[st % NA], Bf[r & 3], acc[dy * 3 + dx]);
      });
    });
  };

  const int pt0 = split * P.perSplit;
  const int pt1 = min(P.nPix, pt0 + P.perSplit);
  const int T = pt1 - pt0;
  if (T <= 0) return;       // uniform over the workgroup (cannot happen with the launcher's split)
  bf16_t* buf0 = sBuf;
  bf16_t* buf1 = sBuf + Cfg::BUF_ELEMS;
  __syncthreads();                               // sAB
  load_half_any(pt0);
  store_half_any(buf0);
  load_half_any(min(pt0 + 1, pt1 - 1));
  __syncthreads();                               // stage 0 complete
  if (grp) {                                     // group 1 runs half a stage ahead with its staging
    store_half_any(buf1);
    load_half_any(min(pt0 + 2, pt1 - 1));
    __syncthreads();
  }
#ifdef FU_CONV_STAMPS
  unsigned long long tM = 0, tB1 = 0, tSt = 0, tLd = 0, tB2 = 0, tGap = 0, sPrev = 0;
#endif
  // The loop is unrolled by the buffer parity: with `(n & 1) ? buf1 : buf0` every LDS address of the MFMA phase existed in
  // two variants that hipcc kept in spilled SGPRs and selected at the top of each iteration -- 60 scalar instructions between
  // the barrier and the first fragment read.
  bf16_t* const sbuf0 = grp ? buf0 : buf1;       // where this wave stages while it multiplies an even stage: stage n+1+grp
  bf16_t* const sbuf1 = grp ? buf1 : buf0;
  mfma_prefetch(std::integral_constant<int, 0>{});
  // Barrier of the loop: this wave's LDS traffic done (lgkmcnt), then s_barrier.  __syncthreads() also waits with vmcnt(0),
  // i.e. for the loads of stage n+2 that the staging half has just issued -- they are needed one whole interval later (round 4:
  // found with the persistent conv kernel, fu_conv_pp.hip; here the staging interval ended with an HBM latency in it).
#ifdef FU_WGRAD_OLD_BARRIER     // (A/B builds)
  auto wg_barrier = []() __attribute__((always_inline)) { __syncthreads(); };
#else
  auto wg_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif
  auto body = [&](auto Par, int n) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
#ifdef FU_CONV_STAMPS
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
    mfma_stage(Par);
#ifdef FU_CONV_STAMPS
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
    wg_barrier();
#ifdef FU_CONV_STAMPS
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
    // stage n+1+grp (already in registers) -> the other buffer of that stage's parity; past the last stage this stores a
    // copy of the last tile into a buffer nobody reads any more (unconditional on purpose: conditional loads become
    // phis that hipcc waits for in front of the MFMA block)
    // (measured and dropped, twice: storing slot by slot with the next tile's load of the same slot issued in between -- the
    //  texture path takes ~45 cycles per 1 KB wave load, 1250 cycles for the eight of a thread -- and a raised s_setprio for
    //  the staging group.  The staging phase shrinks by a third in the stamps; the kernel gets 2-3 % SLOWER, one register
    //  spill included.)
    store_half_any(par ? sbuf1 : sbuf0);
#ifdef FU_CONV_STAMPS
    const unsigned long long s3 = __builtin_amdgcn_s_memtime();
#endif
    load_half_any(min(pt0 + n + 2 + grp, pt1 - 1));
    mfma_prefetch(std::integral_constant<int, 1 - par>{});   // (past the last stage: reads of a valid buffer, never used)
#ifdef FU_CONV_STAMPS
    const unsigned long long s4 = __builtin_amdgcn_s_memtime();
#endif
    wg_barrier();
#ifdef FU_CONV_STAMPS
    const unsigned long long s5 = __builtin_amdgcn_s_memtime();
    if (n > 0) { tM += s1 - s0; tB1 += s2 - s1; tSt += s3 - s2; tLd += s4 - s3; tB2 += s5 - s4; tGap += s0 - sPrev; }
    sPrev = s5;
#endif
  };
  for (int n = 0; n < T; n += 2) {
    body(std::integral_constant<int, 0>{}, n);
    if (n + 1 >= T) break;
    body(std::integral_constant<int, 1>{}, n + 1);
  }
  if (!grp) wg_barrier();                        // group 1's pre-loop barrier
#ifdef FU_CONV_STAMPS
  if (P.dbg && lane == 0 && blockIdx.x < 256) {
    unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = tM; d[1] = tB1; d[2] = tSt; d[3] = (unsigned long long)max(T - 1, 0); d[4] = tLd; d[5] = tB2; d[6] = tGap; d[7] = 0;
  }
#endif

  const int co = co0 + ni * 32 + l31;             // slab[split][tap][Cin / 4][Cout][4], as in k_wgrad_bf16
  if (co < P.Cout) {
    const int cq = P.Cin >> 2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + mi * 32 + 8 * j + 4 * lh;
        if (ci < P.Cin)
          *reinterpret_cast<float4*>(P.slab + ((((int64_t)split * 9 + tap) * cq + (ci >> 2)) * P.Cout + co) * 4) =
              make_float4(acc[tap][4 * j], acc[tap][4 * j + 1], acc[tap][4 * j + 2], acc[tap][4 * j + 3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad of the network's first conv: 8 input channels (the image bands, no BatchNorm in front), 64 output channels
//
// On the kernels above this layer pads c_in to a 64-row tile: 7 of 8 MFMA rows multiply zeros, 58-60 us against ~30 us
// of traffic (134 MB of dy + 17 MB of x).  With 8 channels a pixel of x is ONE 16-byte vector and the whole 3x3 window
// is M = 72 = 9 taps x 8 channels, so the GEMM is (72 x pixels) . (pixels x 64) and the kernel is a stream over dy:
//   * MFMA 16x16x32, K = one tile row of 32 pixels; M tile j = taps 2j, 2j + 1 (the fifth holds tap 8 twice, its second
//     half is never stored), N tile s = 16 output channels: 5 x 4 accumulator tiles = 80 registers per wave;
//   * the 4 waves split K: wave w multiplies tile rows 2w, 2w + 1 (10 + 8 transposing fragment reads per 20 MFMAs) and
//     the workgroup adds its four partial sums once, at the end, through LDS in a fixed order -- one split-K slab per
//     workgroup, same slab layout as the other kernels (the reduce / transpose passes are shared);
//   * dy goes global -> LDS by LDS-DMA (no staging registers, no ds_write): [pixel][64 channels] rows of 128 bytes, the
//     32-byte segment s of pixel p stored at s ^ (p & 3) ^ ((p >> 3) & 1), which spreads the four (eight) pixels of a
//     ds_read_b64_tr_b16 lane group over all banks; the swizzle is applied on the SOURCE side (a lane of the DMA picks
//     its global address, its LDS slot is fixed);
//   * x ([10][38 px][8 ch]; 38: the one tap pair that straddles two halo rows, (0,2) / (1,0), lands on disjoint banks)
//     passes through two registers per thread, zeroed off the image;
//   * two stages of 38 KB, two workgroups per CU; per tile: wait for the own DMA, store x, ONE barrier, issue the next
//     tile's DMA and loads, multiply.
// Eligible: C_in = 8 from one un-normalised source, C_out = 64, H % 8 = 0, W % 32 = 0 (launch_conv3x3_wgrad_bf16).
// ------------------------------------------------------------------------------------------------
#if FU_HALF
#define FU_MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
#define FU_MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef float f32x4w __attribute__((ext_vector_type(4)));

struct WC8 {
  static constexpr int NT = 256, TH = 8, TW = 32, XRS = 38, XROWS = TH + 2, XCOLS = TW + 2;
  static constexpr int X_UNITS = XROWS * XCOLS;              // 340 halo pixels of 16 bytes
  static constexpr int D_BYTES = TH * TW * 128;              // 32768
  static constexpr int X_BYTES = XROWS * XRS * 16;           // 6080
  static constexpr int STAGE = D_BYTES + X_BYTES;            // 38848
  static constexpr int SMEM_BYTES = 2 * STAGE;               // 77696: two workgroups per CU
  static_assert(3 * 20 * 1024 <= SMEM_BYTES, "the final reduction of three waves' accumulators reuses the stages");
};

__global__ __launch_bounds__(256, 2) void k_wgrad_bf16_c8(BWgP P) {
  using C = WC8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int split = blockIdx.x;
  const int pt0 = split * P.perSplit;
  const int pt1 = min(P.nPix, pt0 + P.perSplit);
  const int T = pt1 - pt0;
  if (T <= 0) return;                                        // (uniform; cannot happen with the launcher's split)

  f32x4w acc[5][4];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[j][s] = f32x4w{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (bytes from the stage base): lane 4q + p of k-group g supplies pixel 8g + q (+ 4 in the second
  // read) and "channels" 4p .. 4p + 3 of the 16 rows / columns of the operand tile
  int abase[2][5], bbase[2][4];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + (p >> 1), 8), ky = tap / 3, kx = tap - 3 * ky;
    abase[0][j] = C::D_BYTES + ((2 * wave + ky) * C::XRS + 8 * g + q + kx) * 16 + 8 * (p & 1);
    abase[1][j] = abase[0][j] + C::STAGE;
    asm volatile("" : "+v"(abase[0][j]), "+v"(abase[1][j]));
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bbase[0][s] = (2 * wave * C::TW + 8 * g + q) * 128 + ((s ^ q ^ (g & 1)) * 32) + 8 * p;
    bbase[1][s] = bbase[0][s] + C::STAGE;
    asm volatile("" : "+v"(bbase[0][s]), "+v"(bbase[1][s]));
  }

  // dy DMA: wave-instruction i = 8 wave + it covers pixels 8i .. 8i + 7 of the tile (row i >> 2, columns 8 (i & 3) ..),
  // lane = (pixel lane >> 3, 16-byte slot lane & 7); the slot holds source unit 2 ((slot >> 1) ^ sw) + (slot & 1),
  // sw = (pixel & 3) ^ (i & 1)
  unsigned dlane[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int lp = lane >> 3, sl = lane & 7, sw = (lp & 3) ^ par;
    dlane[par] = (unsigned)(lp * 128 + (2 * ((sl >> 1) ^ sw) + (sl & 1)) * 16);
  }
  // x: halo pixel u = tid (+ 256): byte offset from the tile's origin pixel, border membership, LDS address
  int xrel[2], xlds[2];
  unsigned xT = 0, xB = 0, xL = 0, xR = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = min(tid + k * C::NT, C::X_UNITS - 1);
    const int hy = u / C::XCOLS, hx = u - hy * C::XCOLS;
    xrel[k] = ((hy - 1) * P.W + (hx - 1)) * 16;
    xlds[k] = C::D_BYTES + (hy * C::XRS + hx) * 16;
    xT |= (hy == 0) ? (1u << k) : 0u;  xB |= (hy == C::XROWS - 1) ? (1u << k) : 0u;
    xL |= (hx == 0) ? (1u << k) : 0u;  xR |= (hx == C::XCOLS - 1) ? (1u << k) : 0u;
  }
  const bool x2 = tid + C::NT < C::X_UNITS;                  // this thread owns a second halo pixel
  const char* xb = reinterpret_cast<const char*>(P.src0);
  const char* db = reinterpret_cast<const char*>(P.dy);
  const unsigned rowB = (unsigned)P.W * 128u;

  uint4 rx0, rx1;
  unsigned xbad = 0;
  auto issue = [&](int pt, int st) __attribute__((always_inline)) {        // pt, st uniform
    const int t2 = fast_div(pt, P.tilesX, P.rcp_tilesX);
    const int tx = pt - t2 * P.tilesX;
    const int bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
    const int y0 = (t2 - bb * P.tilesY) * C::TH, x0 = tx * C::TW;
    const size_t tile_pix = (size_t)((bb * P.H + y0) * P.W + x0);
    const char* dt = db + tile_pix * 128 + (size_t)(2 * wave) * rowB;
    unsigned char* ls = smem_raw + st * C::STAGE + wave * 8192;
    unsigned d0 = dlane[0], d1 = dlane[1];
    asm volatile("" : "+v"(d0), "+v"(d1));                   // (opaque: see fu_conv_rs.hip, hoisted 64-bit lane addresses)
#pragma unroll
    for (int it = 0; it < 8; ++it)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(dt + (size_t)(it >> 2) * rowB + (size_t)((it & 3) * 1024) + (size_t)((it & 1) ? d1 : d0)),
          (__attribute__((address_space(3))) void*)(ls + it * 1024), 16, 0, 0);
    const unsigned ft = y0 == 0 ? ~0u : 0u, fb = y0 + C::TH == P.H ? ~0u : 0u;
    const unsigned fl = x0 == 0 ? ~0u : 0u, fr = x0 + C::TW == P.W ? ~0u : 0u;
    xbad = (xT & ft) | (xB & fb) | (xL & fl) | (xR & fr);
    const char* xt = xb + tile_pix * 16;
    rx0 = *reinterpret_cast<const uint4*>(xt + ((xbad & 1u) ? 0 : xrel[0]));   // off the image: the origin pixel, zeroed below
    rx1 = *reinterpret_cast<const uint4*>(xt + ((xbad & 2u) ? 0 : xrel[1]));
  };
  auto store_x = [&](int st) __attribute__((always_inline)) {
    const unsigned m0 = (xbad & 1u) ? 0u : ~0u, m1 = (xbad & 2u) ? 0u : ~0u;
    uint4 v0 = rx0, v1 = rx1;
    v0.x &= m0; v0.y &= m0; v0.z &= m0; v0.w &= m0;
    v1.x &= m1; v1.y &= m1; v1.z &= m1; v1.w &= m1;
    *reinterpret_cast<uint4*>(smem_raw + st * C::STAGE + xlds[0]) = v0;
    if (x2) *reinterpret_cast<uint4*>(smem_raw + st * C::STAGE + xlds[1]) = v1;
  };
  // the fragments of both tile rows of this wave are requested BEFORE the next tile's DMA is issued and multiplied after it:
  // hipcc orders every LDS read behind an outstanding LDS-DMA (s_waitcnt vmcnt) -- reads that follow the issue in program
  // order would wait for the tile that has just been requested
  frag8_t Bf[2][4], Af[2][5];
  auto read_frags = [&](auto Par) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
    static_for<0, 2>([&](auto RR) {
      constexpr int rr = decltype(RR)::value;
      static_for<0, 4>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const bf16_t* bd = reinterpret_cast<const bf16_t*>(smem_raw + bbase[par][s] + rr * (C::TW * 128));
        Bf[rr][s] = tr_frag(bd, bd + 4 * 64);
      });
      static_for<0, 5>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const bf16_t* ad = reinterpret_cast<const bf16_t*>(smem_raw + abase[par][j] + rr * (C::XRS * 16));
        Af[rr][j] = tr_frag(ad, ad + 4 * 8);
      });
    });
  };
  auto multiply = [&]() __attribute__((always_inline)) {
    static_for<0, 2>([&](auto RR) {
      constexpr int rr = decltype(RR)::value;
      static_for<0, 5>([&](auto J) {
        constexpr int j = decltype(J)::value;
        static_for<0, 4>([&](auto S) {
          constexpr int s = decltype(S)::value;
          acc[j][s] = FU_MFMA16W(Af[rr][j], Bf[rr][s], acc[j][s]);
        });
      });
    });
  };
  auto wg_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto body = [&](auto Par, int n) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));    // vmcnt(0): this lane's share of stage n (the builtin, so that hipcc knows the DMA has landed)
    store_x(par);
    wg_barrier();                                            // stage n complete; everybody is done with stage n - 1
    read_frags(Par);
    __builtin_amdgcn_sched_barrier(0);
    issue(pt0 + min(n + 1, T - 1), 1 - par);                 // (past the last tile: the last tile again, never read)
    __builtin_amdgcn_sched_barrier(0);
    multiply();
  };
  issue(pt0, 0);
  for (int n = 0; n < T; n += 2) {
    body(std::integral_constant<int, 0>{}, n);
    if (n + 1 >= T) break;
    body(std::integral_constant<int, 1>{}, n + 1);
  }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));      // the trailing DMA writes LDS: it must have landed before the reuse
  wg_barrier();

  // waves 1-3 park their partial sums, wave 0 adds them in wave order and writes the workgroup's slab
  float* red = reinterpret_cast<float*>(smem_raw);
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        *reinterpret_cast<f32x4w*>(red + (((wave - 1) * 20 + j * 4 + s) * 64 + lane) * 4) = acc[j][s];
  }
  wg_barrier();
  if (wave == 0) {
    // D tile: lane = column (output channel 16 s + lane % 16), registers = rows 4 g + i = tap 2j + (g >> 1), channels
    // 4 (g & 1) + i  ->  slab[split][tap][c_in / 4][c_out][4]: one 16-byte store per accumulator tile
    const int co = lane & 15;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = 2 * j + (g >> 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f32x4w v = acc[j][s];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4w*>(red + ((w * 20 + j * 4 + s) * 64 + lane) * 4);
        if (tap < 9)
          *reinterpret_cast<f32x4w*>(P.slab + ((((int64_t)split * 9 + tap) * 2 + (g & 1)) * 64 + 16 * s + co) * 4) = v;
      }
    }
  }
}

static bool wgrad_c8_eligible(const BWgP& P) {
  const int64_t npx = (int64_t)P.B * P.H * P.W;
  return P.C0 == 8 && P.C1 == 0 && P.a0 == nullptr && P.Cout == 64 && P.H % WC8::TH == 0 && P.W % WC8::TW == 0 &&
         npx * 128 < (int64_t(1) << 40) && npx < (int64_t(1) << 31);
}

static int launch_wgrad_c8(BWgP& P, int target_wgs, const LaunchOpts& o, hipStream_t s) {
  using C = WC8;
  P.tilesX = P.W / C::TW; P.tilesY = P.H / C::TH;
  P.nPix = P.B * P.tilesX * P.tilesY;
  P.nCi = 1; P.nCo = 1;
  int S = target_wgs < P.nPix ? target_wgs : P.nPix;
  if (S < 1) S = 1;
  P.perSplit = ceil_div(P.nPix, S);
  P.S = ceil_div(P.nPix, P.perSplit);     // every split owns at least one tile
  P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_bf16_c8),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES));
    attr_set = true;
  }
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL(k_wgrad_bf16_c8, dim3(P.S), dim3(C::NT), C::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_wgrad_reduce(const float* slab, int S, int Cin, int Cout, int cin_real, float* dw, const float* dbp,
                        int ndb, float* db, hipStream_t s, bool ci4);

template <int WMI, int PTH, int TAPS = 9>
static int launch_wgrad_cfg(BWgP& P, int target_wgs, const LaunchOpts& o, hipStream_t s) {
  using Cfg = WCfg<WMI, PTH>;
  P.tilesX = ceil_div(P.W, Cfg::PTW); P.tilesY = ceil_div(P.H, PTH);
  P.nPix = P.B * P.tilesX * P.tilesY;
  P.nCi = ceil_div(P.Cin, Cfg::CI_T); P.nCo = ceil_div(P.Cout, Cfg::CO_T);
  const int nT = P.nCi * P.nCo;
  int S = ceil_div(target_wgs, nT);
  if (S > P.nPix) S = P.nPix;
  if (S < 1) S = 1;
  P.perSplit = ceil_div(P.nPix, S);
  P.S = ceil_div(P.nPix, P.perSplit);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_bf16<WMI, PTH, TAPS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL((k_wgrad_bf16<WMI, PTH, TAPS>), dim3(nT * P.S), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

#ifndef FU_WGRAD_LOCKSTEP_DEFAULT
#define FU_WGRAD_LOCKSTEP_DEFAULT 0   // A/B builds: -DFU_WGRAD_LOCKSTEP_DEFAULT=1
#endif
#if !FU_HALF
int g_wgrad_force_lockstep = FU_WGRAD_LOCKSTEP_DEFAULT;   // testing hook (fu_test_force_lockstep_wgrad): 1 = k_wgrad_bf16<4,8> instead of the ping-pong kernel, 2 = the ping-pong kernel with its general staging
#endif

static int launch_wgrad_pp(BWgP& P, int target_wgs, const LaunchOpts& o, hipStream_t s) {
  using Cfg = WPCfg;
  P.tilesX = ceil_div(P.W, Cfg::PTW); P.tilesY = ceil_div(P.H, Cfg::PTH);
  P.nPix = P.B * P.tilesX * P.tilesY;
  P.nCi = ceil_div(P.Cin, Cfg::CI_T); P.nCo = ceil_div(P.Cout, Cfg::CO_T);
  const int nT = P.nCi * P.nCo;
  int S = ceil_div(target_wgs, nT);
  if (S > P.nPix) S = P.nPix;
  if (S < 1) S = 1;
  P.perSplit = ceil_div(P.nPix, S);
  P.S = ceil_div(P.nPix, P.perSplit);     // every split owns at least one stage
  P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_bf16_pp<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_bf16_pp<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  // whole pixel tiles, whole channel tiles, 24-bit pixel indices and pixel strides, 32-bit byte offsets inside a source
  const int64_t npx = (int64_t)P.B * P.H * P.W;
  const int cmax = P.C0 > P.C1 ? P.C0 : P.C1;
  const bool fast = g_wgrad_force_lockstep != 2 && P.H % Cfg::PTH == 0 && P.W % Cfg::PTW == 0 && P.Cin % Cfg::CI_T == 0 &&
                    P.Cout % Cfg::CO_T == 0 && npx < (1 << 24) && npx * cmax * 2 < (int64_t(1) << 32) &&
                    npx * P.Cout * 2 < (int64_t(1) << 32) && (int64_t)P.nPix * nT < (int64_t(1) << 31);
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  if (fast) hipLaunchKernelGGL(k_wgrad_bf16_pp<true>, dim3(nT * P.S), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  else hipLaunchKernelGGL(k_wgrad_bf16_pp<false>, dim3(nT * P.S), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

#ifdef FU_EXPERIMENTS
static int wgrad_mode_env() {
  static int m = -1;
  if (m < 0) { const char* e = getenv("FU_WGRAD_MODE"); m = e ? atoi(e) : 0; }
  return m;
}
#endif

// Workgroups of the ping-pong kernel.  Its 8-wave workgroups (248 registers, 125 KB of LDS) take a CU each and live for the
// whole launch (45-290 us).  At 256 of them -- one per CU, round 2 -- every kernel of the main backward chain that starts while
// a weight-gradient launch is in flight (BatchNorm-backward finalize / apply, bilinear backward, the next dgrad) waits for
// those workgroups to retire before it gets a single CU: in the two-stream trace the 36 finalize launches took 494 us against
// 190 alone, the four bilinear backwards 525 against 105 (profiles/r3_*).  Launching FEWER, longer workgroups leaves CUs to
// the critical chain from the first cycle, and the split-K slabs shrink with the workgroup count (75 -> 52 MB written and
// re-read per layer).  Measured, same box, ms per step | conv class TFLOP/s event-timed in the step: 256: 5.74 / 5.75 | 835 /
// 856, 208: 5.61 / 5.64 | 830 / 827, 192: 5.58 / 5.61 | 790 / 790, 176: 5.59 / 5.61 | 793 / 789, 160: 5.56 / 5.57 | 780 / 790,
// 128: 5.67 (another box, against 5.78 at 256).  Below 208 the step gains another 0.5 % while the dgrad launches -- which then
// share the GPU with the longer-running weight-gradient launch of the layer before -- lose 5 %: 208 takes most of the one
// without the other.  (FU_WGRAD_TARGET overrides it in -DFU_EXPERIMENTS builds.)
// Round 4: 160.  The persistent conv kernel (fu_conv_pp.hip) takes a whole CU per workgroup like this one, so a dgrad launch that
// starts beside a weight-gradient launch runs on the CUs this kernel leaves and the rest of its grid waits; measured again with
// that dispatch, two boxes, ms per step | conv class TFLOP/s event-timed in the step: 256: 5.60 | 913, 208: 5.40-5.47 | 890-922,
// 192: 5.34-5.39 | 853-866, 176: 5.33-5.40 | 848-867, 160: 5.28-5.35 | 834-843, 144: 5.38 | 825, 128: 5.57 | 825 -- the step is the
// product's metric (-1.8 % at 160); the conv launches' event-timed rate falls with it by construction (they share more of their
// own duration), their rate with the GPU to themselves (roofline.achieved_serial) does not change.  Capping the dgrad launches'
// grid to the complement instead (FU_PP_GRID: 128 + 128, 112 + 144, 96 + 160) is no better (5.35-5.40) and slower alone.
static int wgrad_pp_target() {
#ifdef FU_EXPERIMENTS
  static int t = -1;
  if (t < 0) { const char* e = getenv("FU_WGRAD_TARGET"); t = e ? atoi(e) : 160; }
  return t;
#else
  return 160;
#endif
}

static int wgrad_c64_target() {
#ifdef FU_EXPERIMENTS
  static int t = -1;
  if (t < 0) { const char* e = getenv("FU_WGRAD_TARGET64"); t = e ? atoi(e) : 512; }
  return t;
#else
  return 512;
#endif
}

// upper bound of the slab size over the two configurations below
int64_t conv3x3_wgrad_slab_elems_bf16(int Cin, int Cout, int B, int H, int W) {
  const int64_t npix = (int64_t)B * ceil_div(H, 8) * ceil_div(W, 16);
  const int nT128 = ceil_div(Cin, 128) * ceil_div(Cout, 64);
  const int nT64 = ceil_div(Cin, 64) * ceil_div(Cout, 64);
  int64_t s128 = ceil_div(256, nT128), s64 = ceil_div(512, nT64);
  if (s128 > npix) s128 = npix;
  if (s64 > npix) s64 = npix;
  const int64_t smax = s128 > s64 ? s128 : s64;
  return (smax + 1) * 9 * (int64_t)Cin * Cout;
}

int launch_conv3x3_wgrad_bf16(const ConvIn& in, const bf16_t* dy, int Cout, float* slab, float* dw_oihw, int cin_real,
                              const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                              hipStream_t s) {
  BWgP P;
  P.src0 = (const bf16_t*)in.src0; P.src1 = (const bf16_t*)in.src1; P.a0 = in.a0; P.b0 = in.b0; P.dy = dy;
  P.slab = slab;
  P.C0 = in.C0; P.C1 = in.src1 ? in.C1 : 0; P.Cin = P.C0 + P.C1; P.Cout = Cout; P.B = B; P.H = H; P.W = W;
  P.dbg = g_conv_dbg;
  FU_REQUIRE(P.C0 % 8 == 0 && P.C1 % 8 == 0 && Cout % 8 == 0, "wgrad_bf16: channel counts must be multiples of 8");
  int st;
  // embedded 1x1 (late-fusion convs): the stage is all staging, so the lock-step kernel (all 8 waves stage together) wins
  // over the ping-pong one; the eight unwritten tap slabs reach only taps of dw_oihw that the caller never reads
  if (in.center_only && !g_bf16_force_full_taps && P.Cin > 64) st = launch_wgrad_cfg<4, 8, 1>(P, 256, in.opt, s);
  else if (in.center_only && !g_bf16_force_full_taps) st = launch_wgrad_cfg<2, 8, 1>(P, 512, in.opt, s);
#ifdef FU_EXPERIMENTS   // A/B knob (tools/ab_*.sh): FU_WGRAD_MODE=1 the 256-thread 64 x 64 kernel everywhere (512 WGs), 2 = the same at 256 WGs
  else if (wgrad_mode_env() == 1) st = launch_wgrad_cfg<2, 8>(P, 512, in.opt, s);
  else if (wgrad_mode_env() == 2) st = launch_wgrad_cfg<2, 8>(P, 256, in.opt, s);
#endif
  else if (!g_wgrad_force_lockstep && wgrad_c8_eligible(P)) st = launch_wgrad_c8(P, wgrad_c64_target(), in.opt, s);   // the 8-band first conv: K = 72 stream over dy
  else if (P.Cin > 64 && g_wgrad_force_lockstep != 1) st = launch_wgrad_pp(P, wgrad_pp_target(), in.opt, s);   // 512 threads, 128 c_in x 64 c_out, one WG per CU
  else if (P.Cin > 64) st = launch_wgrad_cfg<4, 8>(P, wgrad_pp_target(), in.opt, s);   // (same split as the ping-pong kernel: bit-identical sums)
  else st = launch_wgrad_cfg<2, 8>(P, wgrad_c64_target(), in.opt, s);   // 256 threads, 64 x 64, two WGs per CU
  if (st) return st;
  return launch_wgrad_reduce(slab, P.S, P.Cin, Cout, cin_real, dw_oihw, db_partials, n_db_partials, db, s, true);
}

}  // namespace fu

#if !FU_HALF
extern "C" void fu_test_force_general_conv(int on) { fu::g_bf16_force_general = on; }
extern "C" void fu_test_force_lockstep_wgrad(int on) { fu::g_wgrad_force_lockstep = on; }
extern "C" void fu_test_force_full_taps(int on) { fu::g_bf16_force_full_taps = on; }
extern "C" void fu_debug_set_conv_stamps(void* p) { fu::g_conv_dbg = (unsigned long long*)p; }
#endif
