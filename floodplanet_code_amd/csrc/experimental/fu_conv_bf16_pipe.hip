// bf16 3x3 convolution, forward / dgrad: persistent, software-pipelined kernel for gfx950.
//
// Why: with two independent 4-wave workgroups per CU (fu_conv_bf16_fast.hip) the s_memtime stamps show both groups
// in lock step -- LDS staging (MFMA idle, ~3100 cycles per 32-channel chunk), then the MFMA block (pipe shared,
// ~4600 cycles) -- plus a prologue (first-load latency) and an epilogue per 256x64 tile that nothing overlaps.
// Here ONE workgroup per CU (4 waves, one per SIMD) walks a list of output tiles and treats their K chunks as one
// stream with three cursors:
//       L (global loads into registers)  =  S + 1 chunk  =  C (MFMA from LDS) + 2 chunks
// LDS holds two stages (chunk C being read, chunk S being written).  Inside a step every MFMA is followed by a small
// "piece" of the staging work for the next chunk (BN+ReLU on one dword pair, or one ds_write_b128 + the global reload
// of that register), so the VALU / LDS-write / VMEM issue slots sit in the shadow of the 32-cycle MFMAs of the same
// wave; sched_barrier pins that interleave.  Each staging register is reloaded (chunk S + 1) right after it has been
// written to LDS, which gives every global load one full step (>= 2300 cycles) of latency budget and spreads the
// requests over the step instead of issuing them as one burst.  One s_barrier per chunk.  Tile boundaries are
// invisible to the load/stage stream; only the epilogue (same as the fast kernel's) is exposed per tile.
#include "fu_conv_bf16.h"

#ifndef FU_PIPE_DBG
#define FU_PIPE_DBG 0   // experiments: 1 = no global reloads in steady state, 2 = no weight reloads, 3 = no LDS writes
#endif

namespace fu {

template <int NTW>
struct PCfg {
  static constexpr int NT = 256, TW = 16, TH = 16, BN = 32 * NTW, KC = 32, KCP = 40;
  static constexpr int HWd = TW + 2, NHP = (TH + 2) * HWd;
  static constexpr int A_UNITS = NHP * 4;                              // 16-byte units (8 channels) per chunk
  static constexpr int A_ITERS = (A_UNITS + NT - 1) / NT, A_FULL = A_UNITS / NT, A_REM = A_UNITS % NT;
  static constexpr int W_UNITS = 9 * BN * 4;
  static constexpr int W_ITERS = (W_UNITS + NT - 1) / NT, W_FULL = W_UNITS / NT, W_REM = W_UNITS % NT;
  static constexpr int RPI = NT / 4;                                   // LDS rows per staging iteration
  static constexpr int TAPS_PER_IT = RPI / BN;                         // 1 (BN = 64) or 2 (BN = 32)
  static constexpr int STAGE = (NHP + 9 * BN) * KCP;                   // bf16 elements per LDS stage
  static constexpr int AB_FLOATS = 2 * 1024, RED_FLOATS = 4 * BN * 2;
  static constexpr int TRASH_BYTES = NT * 16;                          // sink for the lanes of a ragged iteration
  static constexpr int SMEM_BYTES = 2 * STAGE * 2 + AB_FLOATS * 4 + RED_FLOATS * 4 + TRASH_BYTES;
  static constexpr int MFMAS = 18 * 2 * NTW;                           // per chunk and wave
  static constexpr int PIECES = 5 * A_ITERS + W_ITERS;
  static_assert(PIECES <= MFMAS, "more staging pieces than MFMA slots");
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
};

struct TileCoord { int bb, y0, x0, n0, pixT; };

template <int NTW>
__global__ __launch_bounds__(256) void k_conv3x3_bf16_pipe(BConvP P, int nTiles) {
  using Cfg = PCfg<NTW>;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, KC = Cfg::KC, KCP = Cfg::KCP, NT = Cfg::NT;
  constexpr int HWd = Cfg::HWd, NHP = Cfg::NHP, A_ITERS = Cfg::A_ITERS, W_ITERS = Cfg::W_ITERS, RPI = Cfg::RPI;
  constexpr int STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sS = reinterpret_cast<bf16_t*>(smem_raw);             // 2 stages of { A [NHP][KCP], W [9][BN][KCP] }
  float* sAB = reinterpret_cast<float*>(sS + 2 * STAGE);        // [2][1024] BN scale / shift of source 0
  float* sRed = sAB + Cfg::AB_FLOATS;                           // [4][BN][2] statistics reduction
  unsigned char* sTrash = reinterpret_cast<unsigned char*>(sRed + Cfg::RED_FLOATS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int aq = tid & 3;
  const bool has_bn = P.a0 != nullptr;
  const int nChunks = (P.Cin + KC - 1) / KC;

  // ---- this workgroup's tile list: XCD x owns the logical range [x T/8, (x+1) T/8) (coT-major, so the 32 workgroups
  //      of an XCD share weight tiles in its L2); inside it the workgroups take tiles strided by the group count.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, perXcd = gridDim.x >> 3;
  const int lo = (int)(((int64_t)nTiles * xcd) >> 3), hi = (int)(((int64_t)nTiles * (xcd + 1)) >> 3);
  const int myTiles = (hi - lo - slot + perXcd - 1) / perXcd;   // tiles lo + slot + j perXcd, j < myTiles
  if (myTiles <= 0) return;
  const int nSteps = myTiles * nChunks;

  auto decode = [&](int j) {
    const int jj = min(j, myTiles - 1);                         // past the end: stay on the last tile (harmless)
    const int logical = lo + slot + jj * perXcd;
    TileCoord t;
    const int coT = fast_div(logical, P.nPix, P.rcp_nPix);
    t.pixT = logical - coT * P.nPix;
    const int t2 = fast_div(t.pixT, P.tilesX, P.rcp_tilesX);
    const int tx = t.pixT - t2 * P.tilesX;
    t.bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
    const int ty = t2 - t.bb * P.tilesY;
    t.x0 = tx * TW; t.y0 = ty * TH; t.n0 = coT * BN;
    return t;
  };

  // ---- L cursor state: global addressing of the chunk being loaded ---------------------------------------------
  unsigned a_off[A_ITERS];      // byte offset of staging unit `it` in the current source at channel 0
  unsigned a_okL = 0;           // bit it: halo pixel inside the image
  unsigned w_offL = 0;          // byte offset of this thread's weight row at input channel 0 (tap group 0)
  TileCoord tL;
  int jL = 0, chL = 0;
  const char* abL = nullptr;    // uniform: source base + chunk channel offset
  const char* wbL = nullptr;    // uniform: weights + chunk channel offset
  unsigned cmL = 0;             // 0xffffffff unless this thread's octet is past Cin (ragged last chunk)
  const unsigned w_step = (unsigned)(Cfg::TAPS_PER_IT * P.N * P.Cin) * 2u;
  const int wrow = tid >> 2;

  auto setup_a = [&](int Cs) {
    a_okL = 0;
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int hp = (tid >> 2) + it * RPI;
      const int hy = (hp * 3641) >> 16;                          // hp / 18 (exact for hp < 65536)
      const int hx = hp - hy * HWd;
      const int iy = tL.y0 - 1 + hy, ix = tL.x0 - 1 + hx;
      const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);   // clamped: always a valid address
      const bool ok = (it < Cfg::A_FULL || hp < NHP) && iy == cy && ix == cx;
      a_okL |= ok ? (1u << it) : 0u;
      a_off[it] = (unsigned)((tL.bb * P.H + cy) * P.W + cx) * (unsigned)(Cs * 2) + 16u * aq;
    });
  };
  auto setup_tile_L = [&]() {
    tL = decode(jL);
    setup_a(P.C0);
    const int wco = wrow & (BN - 1), wtsub = wrow / BN;
    const bool w_ok = tL.n0 + wco < P.N;                        // rows past N load row 0: columns never stored
    w_offL = (w_ok ? (unsigned)((wtsub * P.N + tL.n0 + wco) * P.Cin) * 2u : 0u) + 16u * aq;
  };
  auto setup_chunk_L = [&]() {
    const int k0 = chL * KC;
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;            // uniform: C0 % 32 == 0 with two sources
    abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
             : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    wbL = reinterpret_cast<const char*>(P.wpk) + (size_t)k0 * 2;
    cmL = (k0 + 8 * aq < P.Cin) ? 0xffffffffu : 0u;
  };
  auto advance_L = [&]() {
    if (++chL == nChunks) { chL = 0; ++jL; setup_tile_L(); }
    else if (P.src1 != nullptr && chL * KC == P.C0) setup_a(P.C1);
    setup_chunk_L();
  };

  // ---- S cursor state (chunk being written to LDS) and C cursor state (chunk being multiplied) --------------------
  unsigned mk[A_ITERS];         // per staging unit: 0xffffffff keep / 0 zero (padding, ragged channels)
  int k0S = 0; bool bnS = false;
  TileCoord tS, tC;
  int chS = 0, chC = 0;
  float biasv[NTW];
  auto shift_cursors = [&]() {  // C <- S <- L, then L advances
    tC = tS; chC = chS;
    tS = tL; chS = chL; k0S = chL * KC; bnS = has_bn && k0S < P.C0;
    const unsigned km = cmL & a_okL;
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      mk[it] = (unsigned)__builtin_amdgcn_sbfe((int)km, it, 1);
    });
    advance_L();
    if (chC == 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = tC.n0 + nt * 32 + l31;
        biasv[nt] = (P.bias != nullptr && n < P.N) ? P.bias[n] : 0.f;
      }
    }
  };

  // ---- LDS addressing (bf16 element offsets inside a stage) ------------------------------------------------------
  // staging writes: unit `it` -> row (tid >> 2) + 64 it; the lanes a ragged last iteration does not cover write a sink
  const int a_row0 = (tid >> 2) * KCP + 8 * aq;
  const int w_row0 = NHP * KCP + wrow * KCP + 8 * aq;
  unsigned char* const sBase = smem_raw;
  // fragment reads.  m-tile = 2 image rows x 16 columns; lanes 16..31 (second row) take their columns ROTATED by
  // HWd mod 16 so that the 16 lanes of every ds_read_b128 group hit 16 distinct bank slots.
  int aoff[2], boff[NTW];
  const int mrow = l31 >> 4;
  const int mcol = mrow ? ((l31 - 16 - (HWd & 15)) & 15) : l31;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) aoff[mt] = (((wm * 2 + mt) * 2 + mrow) * HWd + mcol) * KCP + 8 * lh;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) boff[nt] = NHP * KCP + (nt * 32 + l31) * KCP + 8 * lh;

  bool fill = true;
  unsigned ra[A_ITERS][4];
  unsigned rw[W_ITERS][4];
  f32x2 ca[4], cb[4];           // BN coefficients of chunk S for this thread's channel octet

  auto reload_a = [&](auto I) {
    constexpr int it = decltype(I)::value;
#if FU_PIPE_DBG == 1
    if (!fill) return;
#endif
    const uint4 t = *reinterpret_cast<const uint4*>(abL + (a_off[it] & cmL));
    ra[it][0] = t.x; ra[it][1] = t.y; ra[it][2] = t.z; ra[it][3] = t.w;
  };
  auto reload_w = [&](auto I) {
    constexpr int it = decltype(I)::value;
#if FU_PIPE_DBG == 1 || FU_PIPE_DBG == 2
    if (!fill) return;
#endif
    const uint4 t = *reinterpret_cast<const uint4*>(wbL + ((w_offL & cmL) + (unsigned)it * w_step));
    rw[it][0] = t.x; rw[it][1] = t.y; rw[it][2] = t.z; rw[it][3] = t.w;
  };
  auto load_coefs = [&]() {     // unconditional: stale LDS is harmless when chunk S has no BN
    const int cc = (bnS ? k0S : 0) + 8 * aq;
    const float4 a0 = *reinterpret_cast<const float4*>(sAB + cc);
    const float4 a1 = *reinterpret_cast<const float4*>(sAB + cc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sAB + 1024 + cc);
    const float4 b1 = *reinterpret_cast<const float4*>(sAB + 1024 + cc + 4);
    ca[0] = f32x2{a0.x, a0.y}; ca[1] = f32x2{a0.z, a0.w}; ca[2] = f32x2{a1.x, a1.y}; ca[3] = f32x2{a1.z, a1.w};
    cb[0] = f32x2{b0.x, b0.y}; cb[1] = f32x2{b0.z, b0.w}; cb[2] = f32x2{b1.x, b1.y}; cb[3] = f32x2{b1.z, b1.w};
  };

  // piece p of "write chunk S (registers) into stage WS, reload the registers with chunk L"
  auto stage_piece = [&](auto Pc, auto WSc, auto BNc) {
    constexpr int p = decltype(Pc)::value, WS = decltype(WSc)::value;
    constexpr bool BN = decltype(BNc)::value;
    if constexpr (p < 5 * A_ITERS) {
      constexpr int it = p / 5, sub = p % 5;
      if constexpr (sub < 4) {
        unsigned v = ra[it][sub];
        if constexpr (BN) v = bn_relu_pair(v, ca[sub], cb[sub]);
        ra[it][sub] = v & mk[it];
      } else {
        bf16_t* dst = sS + WS * STAGE + a_row0 + it * RPI * KCP;
        if constexpr (it >= Cfg::A_FULL)
          if (tid >= Cfg::A_REM) dst = reinterpret_cast<bf16_t*>(sTrash + tid * 16);
#if FU_PIPE_DBG == 3
        if (fill)
#endif
        *reinterpret_cast<uint4*>(dst) = make_uint4(ra[it][0], ra[it][1], ra[it][2], ra[it][3]);
        reload_a(std::integral_constant<int, it>{});
      }
    } else if constexpr (p < Cfg::PIECES) {
      constexpr int it = p - 5 * A_ITERS;
      bf16_t* dst = sS + WS * STAGE + w_row0 + it * RPI * KCP;
      if constexpr (it >= Cfg::W_FULL)
        if (tid >= Cfg::W_REM) dst = reinterpret_cast<bf16_t*>(sTrash + tid * 16);
#if FU_PIPE_DBG == 3
      if (fill)
#endif
      *reinterpret_cast<uint4*>(dst) = make_uint4(rw[it][0], rw[it][1], rw[it][2], rw[it][3]);
      reload_w(std::integral_constant<int, it>{});
    }
  };

  f32x16 acc[2][NTW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // Keeps the accumulators in AGPRs across the step boundaries: without this the register allocator parks them in
  // VGPRs between steps (the epilogue reads them with VALU instructions) and copies all 64 in and out of every step.
  auto pin_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) asm volatile("" : "+a"(acc[i][j]));
  };

  // one step: MFMAs of chunk C from stage RS, interleaved with the staging pieces of chunk S into stage RS ^ 1
  auto step = [&](auto RSc, auto BNc) {
    constexpr int RS = decltype(RSc)::value;
    load_coefs();
    bf16x8 af[3][2], bfr[3][NTW];
    auto load_frag = [&](auto Sc, auto Rc) {      // fragment r (0,1: A m-tiles; 2..: B n-tiles) of k-step s
      constexpr int st = decltype(Sc)::value, r = decltype(Rc)::value, buf = st % 3;
      constexpr int tap = st >> 1, ks = st & 1;
      if constexpr (r < 2) {
        constexpr int toff = RS * STAGE + ((tap / 3) * HWd + (tap % 3)) * KCP + ks * 16;
        af[buf][r] = *reinterpret_cast<const bf16x8*>(sS + aoff[r] + toff);
      } else {
        constexpr int toff = RS * STAGE + tap * BN * KCP + ks * 16;
        bfr[buf][r - 2] = *reinterpret_cast<const bf16x8*>(sS + boff[r - 2] + toff);
      }
    };
    static_for<0, 2>([&](auto Sc) {
      static_for<0, 2 + NTW>([&](auto Rc) { load_frag(Sc, Rc); });
    });
    constexpr int MPS = 2 * NTW;                   // MFMAs per k-step
    static_for<0, Cfg::MFMAS>([&](auto Mc) {
      constexpr int m = decltype(Mc)::value, st = m / MPS, j = m % MPS, mt = j / NTW, nt = j % NTW;
      // fragments of k-step st + 2, spread over this k-step's MFMA slots
      if constexpr (st + 2 < 18) {
        constexpr int nfr = 2 + NTW;               // fragments per k-step
        constexpr int r0 = (j * nfr) / MPS, r1 = ((j + 1) * nfr) / MPS;
        static_for<r0, r1>([&](auto Rc) { load_frag(std::integral_constant<int, st + 2>{}, Rc); });
      }
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st % 3][mt], bfr[st % 3][nt], acc[mt][nt], 0, 0, 0);
      stage_piece(Mc, std::integral_constant<int, RS ^ 1>{}, BNc);
      __builtin_amdgcn_sched_barrier(0);           // pin the interleave: one MFMA, its shadow work, next MFMA
    });
  };

  // ---- epilogue of the tile at cursor C (see fu_conv_bf16_fast.hip for the layout argument) ----------------------
  const int qj = lane & 3;
  const bool q_even = !(lane & 1), q_lo = qj < 2;
  const unsigned sel1 = q_even ? 0x05040100u : 0x03020706u;     // even: (own.lo, recv.lo)  odd: (recv.hi, own.hi)
  float ssum[NTW], ssq[NTW];
  auto epilogue_regs = [&](auto Fc, const TileCoord& t, char* dbase, int dstride) {
    constexpr bool FULL = decltype(Fc)::value;
    unsigned sb[2][4];     // byte offset of the store of (mt, g) from dbase
    unsigned sok = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = qj + 8 * g + 4 * lh;
        const int oy = t.y0 + (wm * 2 + mt) * 2 + (g >> 1);
        const int ox = t.x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
        sb[mt][g] = ((unsigned)((t.bb * P.H + oy) * P.W + ox) * (unsigned)dstride + (unsigned)(l31 & ~3)) * 2u;
        if constexpr (!FULL) sok |= (oy < P.H && ox < P.W) ? (1u << (mt * 4 + g)) : 0u;
      }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = t.n0 + nt * 32 + l31;
      const bool nok = n < P.N;
      const f32x2 bias2 = {biasv[nt], biasv[nt]};
      const bool nqok = (n & ~3) < P.N;
      f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x2 a01 = {acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1]};
          f32x2 a23 = {acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
          if constexpr (FULL) {
            s2 += a01; s2 += a23;
            q2 = a01 * a01 + q2; q2 = a23 * a23 + q2;
          } else {
            const int oy = t.y0 + (wm * 2 + mt) * 2 + (g >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int p = k + 8 * g + 4 * lh;                   // MFMA row -> pixel (second row rotated, see aoff)
              const int ox = t.x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
              const float a = acc[mt][nt][4 * g + k];
              if (nok && oy < P.H && ox < P.W) { s2.x += a; q2.x = fmaf(a, a, q2.x); }
            }
          }
          a01 += bias2; a23 += bias2;
          const unsigned p01 = pack_bf16x2(a01), p23 = pack_bf16x2(a23);
          const unsigned r01 = (unsigned)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xF, 0xF, true);   // quad xor 1
          const unsigned r23 = (unsigned)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xF, 0xF, true);
          const unsigned A = __builtin_amdgcn_perm(r01, p01, sel1);     // pixel (qj & 1),     channel pair
          const unsigned Bq = __builtin_amdgcn_perm(r23, p23, sel1);    // pixel 2 + (qj & 1), channel pair
          const unsigned send = q_lo ? Bq : A;
          const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);  // quad xor 2
          uint2 o;
          o.x = q_lo ? A : recv;
          o.y = q_lo ? recv : Bq;
          if (FULL || (nqok && ((sok >> (mt * 4 + g)) & 1u)))
            *reinterpret_cast<uint2*>(dbase + sb[mt][g] + nt * 64) = o;
        }
      }
      ssum[nt] = s2.x + s2.y;
      ssq[nt] = q2.x + q2.y;
    }
  };
  auto epilogue = [&](const TileCoord& t) {
    const bool to0 = t.n0 < P.D0;                                 // uniform: D0 % BN == 0 with two destinations
    char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + t.n0 : P.dst1 + (t.n0 - P.D0));
    const int dstride = to0 ? P.D0 : P.D1;
    const bool full = (t.y0 + TH <= P.H) && (t.x0 + TW <= P.W) && (t.n0 + BN <= P.N);
    if (full) epilogue_regs(std::true_type{}, t, dbase, dstride);
    else epilogue_regs(std::false_type{}, t, dbase, dstride);
    if (P.stats) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        ssum[nt] += __shfl_xor(ssum[nt], 32, 64);
        ssq[nt] += __shfl_xor(ssq[nt], 32, 64);
      }
      if (lh == 0) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int cn = nt * 32 + l31;
          sRed[(wm * BN + cn) * 2 + 0] = ssum[nt];
          sRed[(wm * BN + cn) * 2 + 1] = ssq[nt];
        }
      }
      __syncthreads();
      if (tid < BN && t.n0 + tid < P.N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) { s += sRed[(m * BN + tid) * 2 + 0]; q += sRed[(m * BN + tid) * 2 + 1]; }
        float* o = P.stats + ((int64_t)t.pixT * P.N + t.n0 + tid) * 2;
        o[0] = s;
        o[1] = q;
      }
      // the next write of sRed is at least one step barrier away
    }
  };

  // ---- pipeline fill -------------------------------------------------------------------------------------------
  setup_tile_L();
  setup_chunk_L();
  static_for<0, A_ITERS>([&](auto I) { reload_a(I); });          // chunk 0 -> registers
  static_for<0, W_ITERS>([&](auto I) { reload_w(I); });
  if (has_bn) {                                                  // BN coefficients of source 0 -> LDS
    for (int c = tid; c < P.C0; c += NT) { sAB[c] = P.a0[c]; sAB[1024 + c] = P.b0[c]; }
  }
  tS = tL; chS = chL;                                            // (overwritten by the shift below)
  shift_cursors();                                               // S = chunk 0, L = chunk 1
  __syncthreads();                                               // sAB visible
  load_coefs();
  if (bnS) static_for<0, Cfg::PIECES>([&](auto Pc) { stage_piece(Pc, std::integral_constant<int, 0>{}, std::true_type{}); });
  else static_for<0, Cfg::PIECES>([&](auto Pc) { stage_piece(Pc, std::integral_constant<int, 0>{}, std::false_type{}); });
  shift_cursors();                                               // C = chunk 0 (stage 0), S = chunk 1, L = chunk 2
  zero_acc();
  __syncthreads();

  fill = false;
  // ---- steady state: step i multiplies chunk i from stage i & 1 -------------------------------------------------
  for (int i = 0; i < nSteps; i += 2) {
    pin_acc();
    if (bnS) step(std::integral_constant<int, 0>{}, std::true_type{});
    else step(std::integral_constant<int, 0>{}, std::false_type{});
    pin_acc();
    __syncthreads();
    if (chC == nChunks - 1) { epilogue(tC); zero_acc(); }
    shift_cursors();
    if (i + 1 >= nSteps) break;
    pin_acc();
    if (bnS) step(std::integral_constant<int, 1>{}, std::true_type{});
    else step(std::integral_constant<int, 1>{}, std::false_type{});
    pin_acc();
    __syncthreads();
    if (chC == nChunks - 1) { epilogue(tC); zero_acc(); }
    shift_cursors();
  }
}

template <int NTW>
static int launch_pipe_cfg(BConvP& P, hipStream_t s) {
  using Cfg = PCfg<NTW>;
  P.tilesX = ceil_div(P.W, Cfg::TW); P.tilesY = ceil_div(P.H, Cfg::TH);
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = ceil_div(P.N, Cfg::BN);
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  FU_REQUIRE((int64_t)P.nPix * P.nCo * P.nPix < ((int64_t)1 << 32), "conv3x3_bf16_pipe: grid too large (%d x %d)",
             P.nPix, P.nCo);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_pipe<NTW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const int nTiles = P.nPix * P.nCo;
  const int grid = nTiles >= 256 ? 256 : ((nTiles + 7) / 8) * 8;   // one persistent workgroup per CU, multiple of 8
  const ProfSlot ps = g_prof_slot;
  g_prof_slot = ProfSlot();
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL((k_conv3x3_bf16_pipe<NTW>), dim3(grid), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P, nTiles);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_conv3x3_bf16_pipe(BConvP& P, hipStream_t s) {
  const int64_t t256 = (int64_t)P.B * ceil_div(P.H, 16) * ceil_div(P.W, 16);
  const bool wide = P.N >= 64 && t256 * ceil_div(P.N, 64) >= 256 && (!P.dst1 || P.D0 % 64 == 0);
  return wide ? launch_pipe_cfg<2>(P, s) : launch_pipe_cfg<1>(P, s);
}

}  // namespace fu
