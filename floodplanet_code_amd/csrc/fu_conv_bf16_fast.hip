// bf16 3x3 convolution, forward / dgrad, aligned-shape fast path for gfx950.
//
// Same implicit GEMM, LDS image and MFMA schedule as k_conv3x3_bf16 (fu_conv_bf16.hip), rewritten around one
// measurement: on the 64..128-channel layers the general kernel issues ~900 VALU/SALU instructions per 32-channel
// chunk and ~1100 in its epilogue against 72 MFMAs per chunk (2300 MFMA cycles), i.e. the waves are VALU-issue bound
// (s_memtime stamps, tools/stamp_test.py: prologue 6100 + epilogue 6200 cycles around a 10200-cycle main loop).  This
// kernel keeps every per-load / per-store quantity in a register that is computed once per tile:
//   * global loads   : uniform (SGPR) chunk base + one 32-bit byte offset per staging slot, no per-chunk address math;
//   * BN+ReLU        : v_pk_fma_f32 on channel pairs, v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the packed pair;
//   * zero padding   : only border tiles / ragged chunks take the masked variant (workgroup-uniform branch);
//   * epilogue       : packed statistics (v_pk_add/fma_f32), convert first and transpose the 4x4 lane quad on PACKED
//                      pairs (2 DPP + 2 v_perm + 1 DPP + 3 selects per 4 registers), stores from a uniform base + a
//                      per-lane byte offset computed once, immediate offsets for the second channel tile.
// Shapes it takes (conv3x3_bf16_fast_eligible): a second source only if C0 % 32 == 0, a second destination only if
// D0 % BN == 0, every tensor < 2 GiB.  Everything else (and only that) runs on the general kernel.
#include "fu_conv_bf16.h"
#include <stdlib.h>

#ifndef FU_FAST_LOADS_PER_STEP
#define FU_FAST_LOADS_PER_STEP 2   // the next chunk's 15 global loads go out 2 per k-step behind that step's MFMAs: as one
#endif                             // burst in front of the block they hold the wave ~1900 cycles before its first MFMA
                                   // (texture path: 60 KB per workgroup at 64 B/clk); measured 787 -> 817 TF (0 = burst).
#ifndef FU_FAST_FRAG_DIST           // Raising the MFMA waves' priority (s_setprio) was measured too: no gain.
#define FU_FAST_FRAG_DIST 1        // k-steps of fragment prefetch (1: two register buffers, 2: three)
#endif
#ifndef FU_FAST_STORE16
#define FU_FAST_STORE16 1   // interior tiles: 16-byte epilogue stores (0: the 8-byte form); measured 825 -> 858 TF
#endif
#ifndef FU_FAST_ROWPERM
#define FU_FAST_ROWPERM 1
#endif
#ifndef FU_FAST_DBG
#define FU_FAST_DBG 0   // diagnostic builds (make EXTRA=-DFU_FAST_DBG=4): 4 = epilogue without its global stores
#endif                  // (64->64 @256^2: 114.8 -> 92.2 us).  -DFU_CONV_STAMPS: s_memtime stamps for
                        // tools/stamp_test.py.  Neither is ever defined in the shipped build.

namespace fu {

// TAPS = 9: the 3x3 convolution.  TAPS = 1: only the centre tap of the same packed [9][N][Cin] weights, i.e. a 1x1
// convolution (the late-fusion convs, whose 1x1 weight is embedded as the centre tap): 2 k-steps per chunk instead of 18.
// MT = m-tiles (2 image rows x 16 columns) per wave: 2 -> 16x16-pixel workgroup tile, 4 -> 16 wide x 32 high (512
// pixels).  KC = input channels per LDS chunk (32 or 16).  The tall tile with 16-channel chunks (MT 4, KC 16) stages
// 2376 16-byte units per 288 MFMAs instead of 3600 and reads 0.75 fragments per MFMA instead of 1, in 57 KB of LDS
// (two workgroups per CU as before).
template <int NTW, int TAPS = 9, int MT_ = 2, int KC_ = 32>
struct FCfg {
  static constexpr int NT = 256, TW = 16, MT = MT_, TH = 8 * MT, BN = 32 * NTW, KC = KC_, KCP = KC + 8;
  static constexpr int UPR = KC / 8;                                   // 16-byte units per LDS row
  static constexpr int KSTEPS = KC / 16;                               // MFMA k-steps per tap and chunk
  static constexpr int HWd = TW + 2, NHP = (TH + 2) * HWd;
  static constexpr int A_UNITS = NHP * UPR;                            // 16-byte units (8 channels) per chunk
  static constexpr int A_ITERS = (A_UNITS + NT - 1) / NT, A_FULL = A_UNITS / NT, A_REM = A_UNITS % NT;
  static constexpr int W_UNITS = TAPS * BN * UPR;
  static constexpr int W_ITERS = (W_UNITS + NT - 1) / NT, W_FULL = W_UNITS / NT, W_REM = W_UNITS % NT;
  static constexpr int ROWS_PER_IT = NT / UPR;                         // LDS rows (pixels / weight rows) per iteration
  static constexpr int TAPS_PER_IT = ROWS_PER_IT / BN;                 // 1, 2 or 4
  static_assert(ROWS_PER_IT % BN == 0 && (KC == 32 || KC == 16) && (MT == 2 || MT == 4), "unsupported tile");
  static constexpr int AB_FLOATS = 2 * 1024;                           // BN scale / shift of source 0
  static constexpr int SMEM_BYTES = (NHP + TAPS * BN) * KCP * 2 + AB_FLOATS * 4;
  static constexpr int NSTEPS = KSTEPS * TAPS;                         // k-steps (16 channels) per chunk
  static constexpr int TAP0 = TAPS == 1 ? 4 : 0;                       // first tap of the packed weights that is used
  static_assert(W_REM % 64 == 0, "the ragged weight iteration must be wave-uniform");
};

template <int NTW, int TAPS = 9, int MT = 2, int KCH = 32>
__global__ __launch_bounds__(256, MT == 4 ? 2 : 1) void k_conv3x3_bf16_fast(BConvP P) {   // tall tile: <= 256 registers

  using Cfg = FCfg<NTW, TAPS, MT, KCH>;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, KC = Cfg::KC, KCP = Cfg::KCP, NT = Cfg::NT;
  constexpr int UPR = Cfg::UPR;
  constexpr int HWd = Cfg::HWd, NHP = Cfg::NHP, A_ITERS = Cfg::A_ITERS, W_ITERS = Cfg::W_ITERS;
  constexpr int RPI = Cfg::ROWS_PER_IT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);            // [NHP][KCP]
  bf16_t* sW = sA + NHP * KCP;                                 // [TAPS][BN][KCP]
  float* sAB = reinterpret_cast<float*>(sW + TAPS * BN * KCP); // [2][1024] BN scale / shift of source 0

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = 64-pixel slice of the tile (uniform)
  const int l31 = lane & 31, lh = lane >> 5;

  // tile decode: three divisions by launch constants, as multiply-high with host-made reciprocals (a hardware integer
  // division is ~35 instructions through the float unit, and all of this sits in front of the first load)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int coT = fast_div(logical, P.nPix, P.rcp_nPix);
  const int pixT = logical - coT * P.nPix;
  const int t2 = fast_div(pixT, P.tilesX, P.rcp_tilesX);
  const int tx = pixT - t2 * P.tilesX;
  const int bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
  const int ty = t2 - bb * P.tilesY;
  const int x0 = tx * TW, y0 = ty * TH, n0 = coT * BN;
  const bool has_bn = P.a0 != nullptr;
  // tiles whose halo leaves the image need zero padding; the others skip the masks altogether
  const bool border = !(y0 >= 1 && y0 + TH + 1 <= P.H && x0 >= 1 && x0 + TW + 1 <= P.W);

  // ---- staging slots.  Slot `it` of thread t is 16-byte unit u = t + 256 it: halo pixel u >> 2, channel octet t & 3.
  //      a_off = byte offset of that unit inside the CURRENT source at channel 0 (recomputed once, at the switch to
  //      the second source); out-of-image pixels load the nearest image pixel and are zeroed when written to LDS.
  const int aq = tid & (UPR - 1);
  // LDS row of this thread's staging units.  A ds_write_b128 is served in groups of 8 lanes = 2 rows x 4 units; with
  // 80-byte rows, consecutive rows (20 dwords apart) put unit 3 of one row on the banks of unit 0 of the next (2-way
  // conflict on every write: ~20 % of the LDS-active cycles, profiles/r1_pmc_bf16.json).  Rows 4 apart (80 dwords = 16 mod
  // 32) do not collide, so the two rows of a group are r and r + 4: bits 0 and 2 of the row index are swapped.
#if FU_FAST_ROWPERM
  const int srow_lin = tid / UPR;
  // 80-byte rows, 4 units per row: the two rows of an 8-lane write group must be 4 apart (bits 0 and 2 swapped).
  // 48-byte rows, 2 units per row (KC = 16): the four rows of a group must be 2 apart -- rows r, r+2, r+4, r+6 put
  // their 8 units on 8 distinct 16-byte slots of the 128-byte bank window (48 r mod 128 = 0, 96, 64, 32); the low
  // three bits are rotated.
  const int srow = UPR == 4 ? ((srow_lin & ~7) | ((srow_lin & 1) << 2) | ((srow_lin >> 1) & 3))
                            : ((srow_lin & ~7) | ((srow_lin & 3) << 1) | ((srow_lin >> 2) & 1));
#else
  const int srow = tid / UPR;
#endif
  unsigned a_off[A_ITERS];
  unsigned a_ok = 0;
  auto setup_a = [&](int Cs) {
    a_ok = 0;
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int hp = srow + it * RPI;
      const int hy = (hp * 3641) >> 16;                          // hp / 18 (exact for hp < 65536)
      const int hx = hp - hy * HWd;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);   // clamped: always a valid address
      const bool ok = (it < Cfg::A_FULL || hp < NHP) && iy == cy && ix == cx;
      a_ok |= ok ? (1u << it) : 0u;
      a_off[it] = (unsigned)((bb * P.H + cy) * P.W + cx) * (unsigned)(Cs * 2) + 16u * aq;
    });
  };
  setup_a(P.C0);
  // weights [tap][n][Cin]: row r = (t >> 2) + 64 it  ->  tap = r / BN, n = n0 + r % BN; the per-iteration part of the
  // address is uniform.  Rows past N load row 0 (their output columns are never stored).
  const int wrow = srow;
  const int wco = wrow & (BN - 1), wtsub = wrow / BN;
  const bool w_ok = n0 + wco < P.N;
  const unsigned w_off = (w_ok ? (unsigned)(((wtsub + Cfg::TAP0) * P.N + n0 + wco) * P.Cin) * 2u : 0u) + 16u * aq;
  const unsigned w_step = (unsigned)(Cfg::TAPS_PER_IT * P.N * P.Cin) * 2u;

  uint4 ra[A_ITERS];
  uint4 rw[W_ITERS];

  // chunk k0 -> staging registers: uniform bases once (load_begin), then one 16-byte load per slot
  const char* abL = nullptr;
  const char* wbL = nullptr;
  unsigned cmL = 0, woL = 0;
  auto load_begin = [&](int k0) {
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;           // uniform: C0 % 32 == 0 when there is a second source
    abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
             : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    wbL = reinterpret_cast<const char*>(P.wpk) + (size_t)k0 * 2;
    // ragged last chunk (Cin % 32 != 0): octets past Cin load offset 0 and are zeroed on the A side
    cmL = (k0 + 8 * aq < P.Cin) ? 0xffffffffu : 0u;
    woL = w_off & cmL;
  };
  auto load_slot = [&](auto Sc) {
    constexpr int sl = decltype(Sc)::value;
    if constexpr (sl < A_ITERS) {
      ra[sl] = *reinterpret_cast<const uint4*>(abL + (a_off[sl] & cmL));
    } else if constexpr (sl < A_ITERS + W_ITERS) {
      constexpr int it = sl - A_ITERS;
      if (it < Cfg::W_FULL || wm < Cfg::W_REM / 64)
        rw[it] = *reinterpret_cast<const uint4*>(wbL + (woL + (unsigned)it * w_step));
    }
  };
  auto load_chunk = [&](int k0) {
    load_begin(k0);
    static_for<0, A_ITERS + W_ITERS>([&](auto Sc) { load_slot(Sc); });
  };

  auto store_chunk = [&](int k0, auto Mc) {
    constexpr bool MASKED = decltype(Mc)::value;
    const bool bn = has_bn && k0 < P.C0;                        // uniform
    unsigned km = 0;
    if constexpr (MASKED) km = (k0 + 8 * aq < P.Cin) ? a_ok : 0u;
    // coefficient reads are unconditional (stale LDS is harmless when the chunk has no BN): a conditional
    // definition would turn the registers into a scratch array
    const int cc = (bn ? k0 : 0) + 8 * aq;                      // < 1024 + 32: inside sAB even on a ragged chunk
    const float4 a0 = *reinterpret_cast<const float4*>(sAB + cc);
    const float4 a1 = *reinterpret_cast<const float4*>(sAB + cc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sAB + 1024 + cc);
    const float4 b1 = *reinterpret_cast<const float4*>(sAB + 1024 + cc + 4);
    const f32x2 ca0 = {a0.x, a0.y}, ca1 = {a0.z, a0.w}, ca2 = {a1.x, a1.y}, ca3 = {a1.z, a1.w};
    const f32x2 cb0 = {b0.x, b0.y}, cb1 = {b0.z, b0.w}, cb2 = {b1.x, b1.y}, cb3 = {b1.z, b1.w};
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < Cfg::A_FULL || srow + it * RPI < NHP) {
        unsigned x = ra[it].x, y = ra[it].y, z = ra[it].z, w = ra[it].w;
        if (bn) {
          x = bn_relu_pair(x, ca0, cb0); y = bn_relu_pair(y, ca1, cb1);
          z = bn_relu_pair(z, ca2, cb2); w = bn_relu_pair(w, ca3, cb3);
        }
        if constexpr (MASKED) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)km, it, 1);   // bit it -> 0 / 0xffffffff
          x &= m; y &= m; z &= m; w &= m;
        }
        *reinterpret_cast<uint4*>(sA + (srow + it * RPI) * KCP + 8 * aq) = make_uint4(x, y, z, w);
      }
    });
    static_for<0, W_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < Cfg::W_FULL || wm < Cfg::W_REM / 64)
        // component-wise: a whole-struct copy becomes a memcpy from the array and keeps it in scratch
        *reinterpret_cast<uint4*>(sW + (wrow + it * RPI) * KCP + 8 * aq) =
            make_uint4(rw[it].x, rw[it].y, rw[it].z, rw[it].w);
    });
  };

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment base offsets (bf16 elements).  m-tile = 2 image rows x 16 columns; lanes 16..31 (second row) take their
  // columns ROTATED by HWd mod 16 so that the 16 lanes of every ds_read_b128 group hit 16 distinct bank slots.
  int aoff[MT], boff[NTW];
  const int mrow = l31 >> 4;
  const int mcol = mrow ? ((l31 - 16 - (HWd & 15)) & 15) : l31;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) aoff[mt] = (((wm * MT + mt) * 2 + mrow) * HWd + mcol) * KCP + 8 * lh;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) boff[nt] = (nt * 32 + l31) * KCP + 8 * lh;

  const int nChunks = (P.Cin + KC - 1) / KC;
#ifdef FU_CONV_STAMPS
  unsigned long long T0 = __builtin_amdgcn_s_memtime(), T1 = 0, T2 = 0;
#endif
  load_chunk(0);                         // the first chunk's loads go out before anything else
  float biasv[NTW];                      // bias is fetched here: a load in the epilogue would expose a full memory
#pragma unroll                           // latency (and its vmcnt(0) would also wait for the stores before it)
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = n0 + nt * 32 + l31;
    biasv[nt] = (P.bias != nullptr && n < P.N) ? P.bias[n] : 0.f;
  }
  if (has_bn) {                          // BN coefficients of source 0 -> LDS (behind the loads above)
    for (int c = tid; c < P.C0; c += NT) { sAB[c] = P.a0[c]; sAB[1024 + c] = P.b0[c]; }
  }
#ifdef FU_CONV_STAMPS
  unsigned long long Sbar = 0, Swait = 0, Sstore = 0, Smfma = 0;   // (per-phase sums: only in older diagnostic builds)
#endif
  // 18 k-steps (9 taps x 2 halves of the 32-channel chunk), software-pipelined by hand: the fragments of step
  // s+1 are requested from LDS before the MFMAs of step s are issued.
  auto mfma_block = [&](auto Lc) {
    constexpr int LOADS = decltype(Lc)::value;   // 0: none; n: next chunk's loads, n per k-step behind its MFMAs
    // (tall tile: 128 accumulator registers; a second fragment buffer would not fit in 256 registers -- its 8 MFMAs per
    //  k-step and the co-resident wave cover the LDS latency instead of a one-step prefetch)
    constexpr int FD = MT == 4 ? 0 : FU_FAST_FRAG_DIST, NB = FD + 1;
    frag8_t af[NB][MT], bfr[NB][NTW];
    auto load_frags = [&](auto Sc, auto Bc) {
      constexpr int st = decltype(Sc)::value, buf = decltype(Bc)::value;
      constexpr int tap = st / Cfg::KSTEPS, ks = st % Cfg::KSTEPS;    // tap = index inside sW
      constexpr int gtap = tap + Cfg::TAP0;                           // its position in the 3x3 window
      constexpr int toff = ((gtap / 3) * HWd + (gtap % 3)) * KCP + ks * 16;
      if constexpr (MT == 4) {   // no prefetch buffer: the weight fragments first, so that the first MFMAs of the step
#pragma unroll                   // wait for 3 of the 6 reads only (FU_TALL_B_FIRST)
        for (int nt = 0; nt < NTW; ++nt)
          bfr[buf][nt] = *reinterpret_cast<const frag8_t*>(sW + tap * BN * KCP + boff[nt] + ks * 16);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[buf][mt] = *reinterpret_cast<const frag8_t*>(sA + aoff[mt] + toff);
      if constexpr (MT != 4) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          bfr[buf][nt] = *reinterpret_cast<const frag8_t*>(sW + tap * BN * KCP + boff[nt] + ks * 16);
      }
    };
    static_for<0, FD>([&](auto Sc) { load_frags(Sc, std::integral_constant<int, decltype(Sc)::value % NB>{}); });
    static_for<0, Cfg::NSTEPS>([&](auto S) {
      constexpr int st = decltype(S)::value, buf = st % NB;
      if constexpr (st + FD < Cfg::NSTEPS) {
        load_frags(std::integral_constant<int, st + FD>{}, std::integral_constant<int, (st + FD) % NB>{});
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = FU_MFMA32(af[buf][mt], bfr[buf][nt], acc[mt][nt]);
      if constexpr (LOADS > 0)
        static_for<0, LOADS>([&](auto J) {
          constexpr int sl = st * LOADS + decltype(J)::value;
          if constexpr (sl < A_ITERS + W_ITERS) load_slot(std::integral_constant<int, sl>{});
        });
    });
  };
  auto stage = [&](int k0) {
    __syncthreads();            // previous chunk's fragment reads are done (and sAB is visible on the first pass)
    if (border || k0 + KC > P.Cin) store_chunk(k0, std::true_type{});
    else store_chunk(k0, std::false_type{});
    __syncthreads();
  };
  // The last chunk is peeled so that the loop body ALWAYS issues the next chunk's loads.  With the loads under
  // `if (ch + 1 < nChunks)` the staging registers became phis of a loaded and a not-loaded path, hipcc copied two of
  // them right behind the loads and put `s_waitcnt vmcnt(9)` in front of the MFMA block: every chunk then waited for
  // the next chunk's first six loads (a full memory latency) before its first MFMA.
  for (int ch = 0; ch + 1 < nChunks; ++ch) {
    const int k0 = ch * KC;
    stage(k0);
#ifdef FU_CONV_STAMPS
    if (ch == 0) T1 = __builtin_amdgcn_s_memtime();
#endif
    if (P.src1 != nullptr && k0 + KC == P.C0) setup_a(P.C1);   // next chunk starts the second source
#if FU_FAST_LOADS_PER_STEP > 0
    load_begin(k0 + KC);
    // all A_ITERS + W_ITERS loads must fit in the block's k-steps: 2 per step over 18 steps, 4 over the 2 of a 1x1
    constexpr int LPS = Cfg::NSTEPS * FU_FAST_LOADS_PER_STEP >= A_ITERS + W_ITERS
                            ? FU_FAST_LOADS_PER_STEP : (A_ITERS + W_ITERS + Cfg::NSTEPS - 1) / Cfg::NSTEPS;
    static_assert(LPS * Cfg::NSTEPS >= A_ITERS + W_ITERS, "next chunk's loads do not fit behind the k-steps");
    mfma_block(std::integral_constant<int, LPS>{});
#else
    load_chunk(k0 + KC);                                      // raw loads stay in flight under the MFMA block
    mfma_block(std::integral_constant<int, 0>{});
#endif
  }
  stage((nChunks - 1) * KC);
#ifdef FU_CONV_STAMPS
  if (nChunks == 1) T1 = __builtin_amdgcn_s_memtime();
#endif
  mfma_block(std::integral_constant<int, 0>{});
#ifdef FU_CONV_STAMPS
  T2 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue ----------------------------------------------------------------------------------
  // Accumulator layout: lane = channel (l31), registers = 16 pixels.  Values are converted to bf16 pairs first
  // (2 pixels of one channel per dword), then the 4x4 (pixel x channel) block of every lane quad is transposed on the
  // packed data: xor-1 DPP + v_perm (per-lane byte selector), xor-2 DPP + selects.  Afterwards lane j of a quad owns
  // channels 4q..4q+3 of pixel j: one 8-byte store, 8 pixels x 64 contiguous bytes per wave instruction.
  const int qj = lane & 3;
  const bool q_even = !(lane & 1), q_lo = qj < 2;
  const unsigned sel1 = q_even ? 0x05040100u : 0x03020706u;     // even: (own.lo, recv.lo)  odd: (recv.hi, own.hi)
  const bool to0 = n0 < P.D0;                                   // uniform: D0 % BN == 0 with two destinations
  char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + n0 : P.dst1 + (n0 - P.D0));
  const int dstride = to0 ? P.D0 : P.D1;
  float ssum[NTW], ssq[NTW];

  auto epilogue = [&](auto Fc, auto Bc) {
    constexpr bool FULL = decltype(Fc)::value, BIAS = decltype(Bc)::value;
    unsigned sb[MT][4];    // byte offset of the store of (mt, g) from dbase
    unsigned sok = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = qj + 8 * g + 4 * lh;
        const int oy = y0 + (wm * MT + mt) * 2 + (g >> 1);
        const int ox = x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
        sb[mt][g] = ((unsigned)((bb * P.H + oy) * P.W + ox) * (unsigned)dstride + (unsigned)(l31 & ~3)) * 2u;
        if constexpr (!FULL) sok |= (oy < P.H && ox < P.W) ? (1u << (mt * 4 + g)) : 0u;
      }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = n0 + nt * 32 + l31;
      const bool nok = n < P.N;
      const f32x2 bias2 = {biasv[nt], biasv[nt]};
      const bool nqok = (n & ~3) < P.N;
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x2 a01 = {acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1]};
          f32x2 a23 = {acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
          if constexpr (FULL) {
            s2 += a01; s2 += a23;
            q2 = a01 * a01 + q2; q2 = a23 * a23 + q2;
          } else {
            const int oy = y0 + (wm * MT + mt) * 2 + (g >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int p = k + 8 * g + 4 * lh;                   // MFMA row -> pixel (second row rotated, see aoff)
              const int ox = x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
              const float a = acc[mt][nt][4 * g + k];
              if (nok && oy < P.H && ox < P.W) { s2.x += a; q2.x = fmaf(a, a, q2.x); }
            }
          }
          if constexpr (BIAS) { a01 += bias2; a23 += bias2; }
          const unsigned p01 = pack_e2(a01), p23 = pack_e2(a23);
          const unsigned r01 = (unsigned)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xF, 0xF, true);   // quad xor 1
          const unsigned r23 = (unsigned)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xF, 0xF, true);
          const unsigned A = __builtin_amdgcn_perm(r01, p01, sel1);     // pixel (qj & 1),     channel pair
          const unsigned Bq = __builtin_amdgcn_perm(r23, p23, sel1);    // pixel 2 + (qj & 1), channel pair
          const unsigned send = q_lo ? Bq : A;
          const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);  // quad xor 2
          uint2 o;
          o.x = q_lo ? A : recv;
          o.y = q_lo ? recv : Bq;
#if FU_FAST_DBG == 4
          asm volatile("" ::"v"(o.x), "v"(o.y));
#else
          if (FULL || (nqok && ((sok >> (mt * 4 + g)) & 1u)))
            *reinterpret_cast<uint2*>(dbase + sb[mt][g] + nt * 64) = o;
#endif
        }
      }
      ssum[nt] = s2.x + s2.y;
      ssq[nt] = q2.x + q2.y;
    }
  };
#if FU_FAST_STORE16
  // Interior tiles: 16-byte stores.  Three exchange levels inside every group of 8 lanes (xor 1 and xor 2 inside the
  // quads as above, then quad <-> quad through row_shl:4 / row_shr:4 with bank masks) turn 8 accumulator registers
  // (8 pixels x 1 channel per lane) into 8 channels of ONE pixel per lane: half the store instructions of the 8-byte
  // form (the epilogue is store-issue bound on the shallow layers).
  auto epilogue16 = [&](auto Bc) {
    constexpr bool BIAS = decltype(Bc)::value;
    const int li = lane & 7;
    const bool upper = (li & 4) != 0;
    unsigned sb16[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = (li & 3) + 8 * (li >> 2) + 4 * lh;
        const int oy = y0 + (wm * MT + mt) * 2 + h;
        const int ox = x0 + (h ? ((col - (HWd & 15)) & 15) : col);
        sb16[mt][h] = ((unsigned)((bb * P.H + oy) * P.W + ox) * (unsigned)dstride + (unsigned)(l31 & ~7)) * 2u;
      }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const f32x2 bias2 = {biasv[nt], biasv[nt]};
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned E[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            f32x2 a = {acc[mt][nt][8 * h + 2 * t], acc[mt][nt][8 * h + 2 * t + 1]};
            s2 += a;
            q2 = a * a + q2;
            if constexpr (BIAS) a += bias2;
            const unsigned pk = pack_e2(a);
            const unsigned rv = (unsigned)__builtin_amdgcn_mov_dpp((int)pk, 0xB1, 0xF, 0xF, true);   // quad xor 1
            E[t] = __builtin_amdgcn_perm(rv, pk, sel1);      // even lane: pixel 2t, odd lane: pixel 2t + 1 (channel pair)
          }
          unsigned F[2][2];                                  // F[u]: 4 channels of pixel 4u + (lane & 3)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const unsigned send = q_lo ? E[2 * u + 1] : E[2 * u];
            const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);   // quad xor 2
            F[u][0] = q_lo ? E[2 * u] : recv;
            F[u][1] = q_lo ? recv : E[2 * u + 1];
          }
          unsigned R[2];                                     // the partner quad's half of this lane's pixel
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const unsigned send = upper ? F[0][d] : F[1][d];
            const int r1 = __builtin_amdgcn_update_dpp((int)send, (int)send, 0x104, 0xF, 0x5, false);   // row_shl:4 -> lower quads
            R[d] = (unsigned)__builtin_amdgcn_update_dpp(r1, (int)send, 0x114, 0xF, 0xA, false);        // row_shr:4 -> upper quads
          }
          uint4 o;
          o.x = upper ? R[0] : F[0][0];
          o.y = upper ? R[1] : F[0][1];
          o.z = upper ? F[1][0] : R[0];
          o.w = upper ? F[1][1] : R[1];
          *reinterpret_cast<uint4*>(dbase + sb16[mt][h] + nt * 64) = o;
        }
      }
      ssum[nt] = s2.x + s2.y;
      ssq[nt] = q2.x + q2.y;
    }
  };
#endif
  const bool full = (y0 + TH <= P.H) && (x0 + TW <= P.W) && (n0 + BN <= P.N);   // workgroup-uniform
  if (full) {
#if FU_FAST_STORE16
    if (P.bias) epilogue16(std::true_type{});
    else epilogue16(std::false_type{});
#else
    if (P.bias) epilogue(std::true_type{}, std::true_type{});
    else epilogue(std::true_type{}, std::false_type{});
#endif
  } else {
    if (P.bias) epilogue(std::false_type{}, std::true_type{});
    else epilogue(std::false_type{}, std::false_type{});
  }

#ifdef FU_CONV_STAMPS
  const unsigned long long T2b = __builtin_amdgcn_s_memtime();
#endif
  if (P.stats) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [4][BN][2]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      ssum[nt] += __shfl_xor(ssum[nt], 32, 64);
      ssq[nt] += __shfl_xor(ssq[nt], 32, 64);
    }
    __syncthreads();
    if (lh == 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int cn = nt * 32 + l31;
        red[(wm * BN + cn) * 2 + 0] = ssum[nt];
        red[(wm * BN + cn) * 2 + 1] = ssq[nt];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < P.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { s += red[(m * BN + tid) * 2 + 0]; q += red[(m * BN + tid) * 2 + 1]; }
      float* o = P.stats + ((int64_t)pixT * P.N + n0 + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
#ifdef FU_CONV_STAMPS
  const unsigned long long T2c = __builtin_amdgcn_s_memtime();
  if (P.dbg && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long T3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = P.dbg + (size_t)blockIdx.x * 10;
    d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3; d[4] = Sbar; d[5] = Swait; d[6] = Sstore; d[7] = Smfma; d[8] = T2b; d[9] = T2c;
  }
#endif
}

#ifndef FU_TILE_MODE_DEFAULT
#define FU_TILE_MODE_DEFAULT 0
#endif
#if FU_HALF
extern int g_bf16_tile_mode;
#else
int g_bf16_tile_mode = FU_TILE_MODE_DEFAULT;   // 0 = heuristic, 1 = never the tall tile, 2 = tall wherever 64-channel tiles run
#endif

bool conv3x3_bf16_fast_eligible(const BConvP& P) {
  const int64_t px = (int64_t)P.B * P.H * P.W;
  const int64_t lim = (int64_t)1 << 31;
  if (P.src1 && (P.C0 % 32) != 0) return false;
  if (P.dst1 && (P.D0 % 32) != 0) return false;
  if ((P.a0 != nullptr && P.C0 > 1024) || (P.C0 % 8) || (P.C1 % 8) || (P.N % 4)) return false;
  if (px * P.C0 * 2 >= lim || px * P.C1 * 2 >= lim || px * P.D0 * 2 >= lim || px * P.D1 * 2 >= lim) return false;
  if ((int64_t)9 * P.N * P.Cin * 2 >= lim) return false;
  return true;
}

template <int NTW, int TAPS = 9, int MT = 2, int KCH = 32>
static int launch_fast_cfg(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  using Cfg = FCfg<NTW, TAPS, MT, KCH>;
  P.tilesX = ceil_div(P.W, Cfg::TW); P.tilesY = ceil_div(P.H, Cfg::TH);
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = ceil_div(P.N, Cfg::BN);
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  FU_REQUIRE((int64_t)P.nPix * P.nCo * P.nPix < ((int64_t)1 << 32), "conv3x3_bf16_fast: grid too large (%d x %d)",
             P.nPix, P.nCo);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_fast<NTW, TAPS, MT, KCH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL((k_conv3x3_bf16_fast<NTW, TAPS, MT, KCH>), dim3(P.nPix * P.nCo), dim3(Cfg::NT), Cfg::SMEM_BYTES, s,
                     P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

// Tile choice (all tiles are 16x16 = 256 output pixels): 64 output channels per workgroup when that still yields
// >= 512 workgroups (two per CU), else 32.
int launch_conv3x3_bf16_fast(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  const int64_t t256 = (int64_t)P.B * ceil_div(P.H, 16) * ceil_div(P.W, 16);
  const bool wide = P.N >= 64 && t256 * ceil_div(P.N, 64) >= 512 && (!P.dst1 || P.D0 % 64 == 0);
  if (P.center_only) return wide ? launch_fast_cfg<2, 1>(P, o, s) : launch_fast_cfg<1, 1>(P, o, s);
  // 8 input channels (the network's first conv): the K = 72 kernel without LDS staging (fu_conv_rs.hip)
  if (g_bf16_tile_mode == 0 && conv3x3_c8_eligible(P)) return launch_conv3x3_c8(P, o, s);
  // Row-stationary kernel (fu_conv_rs.hip), wherever the shape is eligible and one of its tiles gives every CU two
  // workgroups.  Measured per layer against the kernels below (bench shapes, forward, tools/conv_modes.py): 5-11 % faster
  // on the 128x128, 64x64 and 32x32 layers with N >= 512 channels x tiles, equal on the two-chunk 256x256 layers, slower
  // below 512 workgroups (16x16 level, 512 -> 256 at 32x32).  Tile mode 3 forces it, modes 1 / 2 exclude it.
  // persistent ping-pong kernel (fu_conv_pp.hip): tile mode 4 forces it; by default wherever it is eligible (FU_CONV_PP=0: never,
  // for A/B runs -- bench.py records every FU_* variable of its environment in the line it prints)
  if (conv3x3_pp_eligible(P)) {
    static const int pp_default = [] { const char* e = getenv("FU_CONV_PP"); return e ? atoi(e) : 1; }();
    const bool wants_bnb = o.bnb != nullptr && o.bnb->y != nullptr && P.a0 == nullptr && P.dst1 == nullptr && P.stats == nullptr;
    if (g_bf16_tile_mode == 4 ||
        (g_bf16_tile_mode == 0 && pp_default && (wants_bnb ? conv3x3_pp_preferred_bnb(P) : conv3x3_pp_preferred(P))))
      return launch_conv3x3_pp(P, o, s);
  }
  if (conv3x3_rs_eligible(P)) {
    const int64_t t256 = (int64_t)P.B * (P.H / 16) * (P.W / 16) * (P.N / 64);
    if (g_bf16_tile_mode == 3 || (g_bf16_tile_mode == 0 && t256 >= 512)) return launch_conv3x3_rs(P, o, s);
  }
  // tall tile (16 x 32 pixels, 16-channel chunks).  Measured per layer against the square tile (bench shapes, one
  // stream): 5-9 % faster where it still yields >= 2048 workgroups (the 256x256 layers; the 8-channel first conv 60 ->
  // 46 us), within +-4 % at 1024, 10 % slower at <= 512 -- hence the threshold.
  const int64_t t512 = (int64_t)P.B * ceil_div(P.H, 32) * ceil_div(P.W, 16);
  const bool tall = g_bf16_tile_mode == 2 ? wide
                                          : (g_bf16_tile_mode == 0 && wide && t512 * ceil_div(P.N, 64) >= 2048);
  if (tall) return launch_fast_cfg<2, 9, 4, 16>(P, o, s);
  return wide ? launch_fast_cfg<2>(P, o, s) : launch_fast_cfg<1>(P, o, s);
}

}  // namespace fu

#if !FU_HALF
extern "C" void fu_test_conv_tile_mode(int mode) { fu::g_bf16_tile_mode = mode; }
#endif
