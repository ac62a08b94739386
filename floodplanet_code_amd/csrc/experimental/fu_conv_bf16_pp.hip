// bf16 3x3 convolution, forward / dgrad: persistent "ping-pong" kernel for gfx950.
//
// Measured on the one-tile-per-workgroup kernel (fu_conv_bf16_fast.hip, two workgroups per CU): both workgroups run
// in lock step -- LDS staging of a chunk (MFMA pipe idle, ~3100 cycles), then the MFMA block (pipe shared by the two,
// ~4600 cycles) -- and nothing overlaps a tile's prologue (first-load latency) and epilogue.  Nothing in the hardware
// keeps two independent workgroups out of phase, so here the two 4-wave groups live in ONE 8-wave workgroup and the
// workgroup barrier does: in every phase one group multiplies its staged chunk while the other group stages its next
// one (BN+ReLU, ds_write, and the global loads of the chunk after that); barrier; roles swap.  Each SIMD holds one
// wave of either group, so MFMA issue of one and VALU / LDS-write / VMEM issue of the other share it.  Each group owns
// an LDS stage (A halo tile + weight slice, same image as the fast kernel), walks its own list of output tiles
// (persistent, one workgroup per CU) and treats the chunks of consecutive tiles as one stream: the loads of the next
// tile's first chunk go out under the current tile's last MFMA block, its epilogue runs in the group's next staging
// phase.  The BatchNorm statistics of a tile leave through a small per-group LDS area one phase later, so an epilogue
// needs no barrier of its own (both groups must execute the same barrier sequence).
//
// group g, chunk k of its stream (K_g chunks in all):   stage S(k) in phase 2k + g,   MFMA M(k) in phase 2k + 1 + g,
// epilogue of a tile whose last chunk is k in phase 2k + 2 + g, statistics written out in phase 2k + 3 + g.
#include "fu_conv_bf16.h"

namespace fu {

template <int NTW>
struct PPCfg {
  static constexpr int NG = 256, NT = 512, TW = 16, TH = 16, BN = 32 * NTW, KC = 32, KCP = 40;
  static constexpr int HWd = TW + 2, NHP = (TH + 2) * HWd;
  static constexpr int A_UNITS = NHP * 4;                              // 16-byte units (8 channels) per chunk
  static constexpr int A_ITERS = (A_UNITS + NG - 1) / NG, A_FULL = A_UNITS / NG, A_REM = A_UNITS % NG;
  static constexpr int W_UNITS = 9 * BN * 4;
  static constexpr int W_ITERS = (W_UNITS + NG - 1) / NG, W_FULL = W_UNITS / NG, W_REM = W_UNITS % NG;
  static constexpr int RPI = NG / 4;                                   // LDS rows per staging iteration
  static constexpr int TAPS_PER_IT = RPI / BN;                         // 1 (BN = 64) or 2 (BN = 32)
  static constexpr int STAGE = (NHP + 9 * BN) * KCP;                   // bf16 elements of one group's stage
  static constexpr int AB_FLOATS = 2 * 1024, RED_FLOATS = 4 * BN * 2;
  static constexpr int SMEM_BYTES = 2 * STAGE * 2 + AB_FLOATS * 4 + 2 * RED_FLOATS * 4;
  static_assert(W_REM % 64 == 0, "the ragged weight iteration must be wave-uniform");
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
};

struct PPTile { int bb, y0, x0, n0, pixT; };

template <int NTW>
__global__ __launch_bounds__(512) void k_conv3x3_bf16_pp(BConvP P, int nTiles) {
  using Cfg = PPCfg<NTW>;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, KC = Cfg::KC, KCP = Cfg::KCP;
  constexpr int HWd = Cfg::HWd, NHP = Cfg::NHP, A_ITERS = Cfg::A_ITERS, W_ITERS = Cfg::W_ITERS, RPI = Cfg::RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);     // ping-pong group (uniform per wave)
  const int wm = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int tg = tid & 255;                                     // thread index inside the group
  const int l31 = lane & 31, lh = lane >> 5;
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw) + grp * Cfg::STAGE;   // [NHP][KCP]
  bf16_t* sW = sA + NHP * KCP;                                           // [9][BN][KCP]
  float* sAB = reinterpret_cast<float*>(smem_raw + 2 * Cfg::STAGE * 2);  // [2][1024] BN scale / shift of source 0
  float* sRed = sAB + Cfg::AB_FLOATS + grp * Cfg::RED_FLOATS;            // [4][BN][2] statistics of the group's tile
  const int aq = tg & 3;
  const bool has_bn = P.a0 != nullptr;
  const int nChunks = (P.Cin + KC - 1) / KC;

  // ---- tile lists: XCD x owns the logical tiles [x T/8, (x+1) T/8) (coT-major: its workgroups share weight slices in
  //      its L2); workgroup `slot` of the XCD and group g take tiles lo + 2 slot + g + j * 2 perXcd
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, perXcd = gridDim.x >> 3;
  const int lo = (int)(((int64_t)nTiles * xcd) >> 3), hi = (int)(((int64_t)nTiles * (xcd + 1)) >> 3);
  const int first = lo + 2 * slot + grp, strideT = 2 * perXcd;
  const int myTiles = first < hi ? (hi - first + strideT - 1) / strideT : 0;
  const int otherFirst = lo + 2 * slot + (grp ^ 1);
  const int otherTiles = otherFirst < hi ? (hi - otherFirst + strideT - 1) / strideT : 0;
  const int K = myTiles * nChunks;                              // chunks in this group's stream
  const int Kmax = (myTiles > otherTiles ? myTiles : otherTiles) * nChunks;
  if (Kmax == 0) return;

  auto decode = [&](int j) {
    const int jj = min(j, max(myTiles - 1, 0));                 // past the end: stay on the last tile (harmless)
    const int logical = min(first + jj * strideT, nTiles - 1);
    PPTile t;
    const int coT = fast_div(logical, P.nPix, P.rcp_nPix);
    t.pixT = logical - coT * P.nPix;
    const int t2 = fast_div(t.pixT, P.tilesX, P.rcp_tilesX);
    const int tx = t.pixT - t2 * P.tilesX;
    t.bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
    const int ty = t2 - t.bb * P.tilesY;
    t.x0 = tx * TW; t.y0 = ty * TH; t.n0 = coT * BN;
    return t;
  };

  // ---- load cursor: tile and chunk whose data is in (or on its way into) the staging registers -------------------
  PPTile tl;                    // tile of the load cursor
  int jL = 0, chL = 0;
  bool borderL = false;
  unsigned a_off[A_ITERS];      // byte offset of staging unit `it` in the current source at channel 0
  unsigned a_ok = 0;            // bit it: halo pixel inside the image
  unsigned w_off = 0;
  const unsigned w_step = (unsigned)(Cfg::TAPS_PER_IT * P.N * P.Cin) * 2u;
  const int wrow = tg >> 2;
  auto setup_a = [&](int Cs) {
    a_ok = 0;
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int hp = (tg >> 2) + it * RPI;
      const int hy = (hp * 3641) >> 16;                          // hp / 18 (exact for hp < 65536)
      const int hx = hp - hy * HWd;
      const int iy = tl.y0 - 1 + hy, ix = tl.x0 - 1 + hx;
      const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);   // clamped: always a valid address
      const bool ok = (it < Cfg::A_FULL || hp < NHP) && iy == cy && ix == cx;
      a_ok |= ok ? (1u << it) : 0u;
      a_off[it] = (unsigned)((tl.bb * P.H + cy) * P.W + cx) * (unsigned)(Cs * 2) + 16u * aq;
    });
  };
  auto setup_tile = [&]() {
    tl = decode(jL);
    borderL = !(tl.y0 >= 1 && tl.y0 + TH + 1 <= P.H && tl.x0 >= 1 && tl.x0 + TW + 1 <= P.W);
    setup_a(P.C0);
    const int wco = wrow & (BN - 1), wtsub = wrow / BN;
    const bool w_ok = tl.n0 + wco < P.N;                        // rows past N load row 0: columns never stored
    w_off = (w_ok ? (unsigned)((wtsub * P.N + tl.n0 + wco) * P.Cin) * 2u : 0u) + 16u * aq;
  };

  uint4 ra[A_ITERS];
  uint4 rw[W_ITERS];
  // chunk chL of tile tl -> staging registers, one 16-byte load per slot (slots 0..A_ITERS-1: halo tile, then weights)
  const char* abL = nullptr;
  const char* wbL = nullptr;
  unsigned cmL = 0, woL = 0;
  auto load_begin = [&]() {
    const int k0 = chL * KC;
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;            // uniform: C0 % 32 == 0 with two sources
    abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
             : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    wbL = reinterpret_cast<const char*>(P.wpk) + (size_t)k0 * 2;
    cmL = (k0 + 8 * aq < P.Cin) ? 0xffffffffu : 0u;             // ragged last chunk: octets past Cin
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
  auto load_chunk = [&]() {
    load_begin();
    static_for<0, A_ITERS + W_ITERS>([&](auto Sc) { load_slot(Sc); });
  };
  // R: the chunk in (or on its way into) the staging registers; M: the chunk in this group's LDS stage
  int jR = 0, chR = 0, jM = 0, chM = 0;
  bool borderR = false;
  unsigned a_okR = 0;
  auto advance_load = [&]() {   // registers now hold the cursor's chunk: remember it, move the cursor on
    jR = jL; chR = chL; borderR = borderL; a_okR = a_ok;
    if (++chL == nChunks) { chL = 0; ++jL; setup_tile(); }
    else if (P.src1 != nullptr && chL * KC == P.C0) setup_a(P.C1);
  };

  auto store_chunk = [&](auto Mc) {   // staging registers (chunk R) -> this group's LDS stage
    constexpr bool MASKED = decltype(Mc)::value;
    const int k0 = chR * KC;
    const bool bn = has_bn && k0 < P.C0;                        // uniform
    unsigned km = 0;
    if constexpr (MASKED) km = (k0 + 8 * aq < P.Cin) ? a_okR : 0u;
    const int cc = (bn ? k0 : 0) + 8 * aq;
    const float4 a0 = *reinterpret_cast<const float4*>(sAB + cc);
    const float4 a1 = *reinterpret_cast<const float4*>(sAB + cc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sAB + 1024 + cc);
    const float4 b1 = *reinterpret_cast<const float4*>(sAB + 1024 + cc + 4);
    const f32x2 ca0 = {a0.x, a0.y}, ca1 = {a0.z, a0.w}, ca2 = {a1.x, a1.y}, ca3 = {a1.z, a1.w};
    const f32x2 cb0 = {b0.x, b0.y}, cb1 = {b0.z, b0.w}, cb2 = {b1.x, b1.y}, cb3 = {b1.z, b1.w};
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < Cfg::A_FULL || tg < Cfg::A_REM) {
        unsigned x = ra[it].x, y = ra[it].y, z = ra[it].z, w = ra[it].w;
        if (bn) {
          x = bn_relu_pair(x, ca0, cb0); y = bn_relu_pair(y, ca1, cb1);
          z = bn_relu_pair(z, ca2, cb2); w = bn_relu_pair(w, ca3, cb3);
        }
        if constexpr (MASKED) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)km, it, 1);   // bit it -> 0 / 0xffffffff
          x &= m; y &= m; z &= m; w &= m;
        }
        *reinterpret_cast<uint4*>(sA + ((tg >> 2) + it * RPI) * KCP + 8 * aq) = make_uint4(x, y, z, w);
      }
    });
    static_for<0, W_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < Cfg::W_FULL || wm < Cfg::W_REM / 64)
        *reinterpret_cast<uint4*>(sW + (wrow + it * RPI) * KCP + 8 * aq) =
            make_uint4(rw[it].x, rw[it].y, rw[it].z, rw[it].w);
    });
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
  // fragment base offsets (bf16 elements); second-row columns rotated by HWd mod 16 (conflict-free ds_read_b128)
  int aoff[2], boff[NTW];
  const int mrow = l31 >> 4;
  const int mcol = mrow ? ((l31 - 16 - (HWd & 15)) & 15) : l31;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) aoff[mt] = (((wm * 2 + mt) * 2 + mrow) * HWd + mcol) * KCP + 8 * lh;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) boff[nt] = (nt * 32 + l31) * KCP + 8 * lh;

  // WITH_LOADS: the next chunk's 15 global loads go out one per k-step behind that step's MFMAs.  Issued as one
  // burst in front of the block they held the wave for 1800-2000 cycles before its first MFMA (stamps): the texture
  // path takes the 60 KB of a group at 64 B/clk.
  auto mfma_block = [&](auto Lc) {
    (void)Lc;
    bf16x8 af[2][2], bfr[2][NTW];
    auto load_frags = [&](auto Sc, auto Bc) {
      constexpr int st = decltype(Sc)::value, buf = decltype(Bc)::value;
      constexpr int tap = st >> 1, ks = st & 1;
      constexpr int toff = ((tap / 3) * HWd + (tap % 3)) * KCP + ks * 16;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *reinterpret_cast<const bf16x8*>(sA + aoff[mt] + toff);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        bfr[buf][nt] = *reinterpret_cast<const bf16x8*>(sW + tap * BN * KCP + boff[nt] + ks * 16);
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
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][mt], bfr[buf][nt], acc[mt][nt], 0, 0, 0);
    });
  };

  // ---- epilogue of the finished tile tc (register part; statistics go to sRed and leave one phase later) -----------
  PPTile tc;                    // tile whose accumulators are complete
  float biasv[NTW];
  const int qj = lane & 3;
  const bool q_even = !(lane & 1), q_lo = qj < 2;
  const unsigned sel1 = q_even ? 0x05040100u : 0x03020706u;     // even: (own.lo, recv.lo)  odd: (recv.hi, own.hi)
  auto epilogue_regs = [&](auto Fc) {
    constexpr bool FULL = decltype(Fc)::value;
    const bool to0 = tc.n0 < P.D0;                              // uniform: D0 % BN == 0 with two destinations
    char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + tc.n0 : P.dst1 + (tc.n0 - P.D0));
    const int dstride = to0 ? P.D0 : P.D1;
    unsigned sb[2][4];     // byte offset of the store of (mt, g) from dbase
    unsigned sok = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = qj + 8 * g + 4 * lh;
        const int oy = tc.y0 + (wm * 2 + mt) * 2 + (g >> 1);
        const int ox = tc.x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
        sb[mt][g] = ((unsigned)((tc.bb * P.H + oy) * P.W + ox) * (unsigned)dstride + (unsigned)(l31 & ~3)) * 2u;
        if constexpr (!FULL) sok |= (oy < P.H && ox < P.W) ? (1u << (mt * 4 + g)) : 0u;
      }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = tc.n0 + nt * 32 + l31;
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
            const int oy = tc.y0 + (wm * 2 + mt) * 2 + (g >> 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int p = k + 8 * g + 4 * lh;                   // MFMA row -> pixel (second row rotated, see aoff)
              const int ox = tc.x0 + ((g >> 1) ? ((p - 16 - (HWd & 15)) & 15) : p);
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
      if (P.stats) {
        float ssum = s2.x + s2.y, ssq = q2.x + q2.y;
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lh == 0) {
          const int cn = nt * 32 + l31;
          sRed[(wm * BN + cn) * 2 + 0] = ssum;
          sRed[(wm * BN + cn) * 2 + 1] = ssq;
        }
      }
    }
  };
  int statPix = -1, statN0 = 0;  // tile whose statistics sit in sRed (-1: none)
  auto flush_stats = [&]() {
    if (statPix >= 0 && tg < BN && statN0 + tg < P.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { s += sRed[(m * BN + tg) * 2 + 0]; q += sRed[(m * BN + tg) * 2 + 1]; }
      float* o = P.stats + ((int64_t)statPix * P.N + statN0 + tg) * 2;
      o[0] = s;
      o[1] = q;
    }
    statPix = -1;
  };

  // ---- prologue: first chunk on its way, BN coefficients into LDS ----------------------------------------------
  if (K > 0) { setup_tile(); load_chunk(); advance_load(); }   // chunk 0 -> registers
  if (has_bn) {
    for (int c = tid; c < P.C0; c += Cfg::NT) { sAB[c] = P.a0[c]; sAB[1024 + c] = P.b0[c]; }
  }
  zero_acc();
  bool epi_pending = false;
  __syncthreads();

  // Both groups run the same loop -- stage chunk k, barrier, multiply chunk k, barrier -- group 1 one barrier late:
  // that one-phase offset is the whole ping-pong (group 0 multiplies while group 1 stages and vice versa).
#ifdef FU_CONV_STAMPS
  unsigned long long tS = 0, tB1 = 0, tM = 0, tB2 = 0, tPre = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
  if (grp == 1) __syncthreads();
  for (int k = 0; k < Kmax + 2; ++k) {
#ifdef FU_CONV_STAMPS
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
    // ---- staging phase: S(k) first (it frees the staging registers), then the epilogue of the tile that M(k - 1)
    //      completed
    if (k < K) {
      if (borderR || chR * KC + KC > P.Cin) store_chunk(std::true_type{});
      else store_chunk(std::false_type{});
      jM = jR; chM = chR;
      // The registers are free again: the NEXT chunk's loads go out now, in this group's staging phase, and have the
      // whole MFMA phase to land (measured: >= 3500 cycles from issue to data under load; issued as a burst in front of
      // the MFMA block they cost the wave 1800-2000 cycles of issue time before its first MFMA, spread inside the block
      // they arrived too late for the next staging phase).  Always issued (past the end: a harmless extra chunk).
      load_chunk();
      advance_load();
    }
    if (epi_pending) {
      const bool full = (tc.y0 + TH <= P.H) && (tc.x0 + TW <= P.W) && (tc.n0 + BN <= P.N);
      if (full) epilogue_regs(std::true_type{});
      else epilogue_regs(std::false_type{});
      if (P.stats) { statPix = tc.pixT; statN0 = tc.n0; }
      zero_acc();
      epi_pending = false;
    }
#ifdef FU_CONV_STAMPS
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- MFMA phase: statistics of the tile whose epilogue just ran, loads of chunk k + 1, M(k)
    flush_stats();
    if (k < K) {
      if (chM == nChunks - 1) {           // M(k) completes a tile: keep its coordinates, fetch its bias
        tc = decode(jM);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int n = tc.n0 + nt * 32 + l31;
          biasv[nt] = (P.bias != nullptr && n < P.N) ? P.bias[n] : 0.f;
        }
        epi_pending = true;
      }
#ifdef FU_CONV_STAMPS
      const unsigned long long tpre = __builtin_amdgcn_s_memtime();
      if (k >= 2 && k < K - 1) tPre += tpre - tc0;
#endif
      mfma_block(std::false_type{});
    }
#ifdef FU_CONV_STAMPS
    const unsigned long long td = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    if (k >= 2 && k < K - 1) { tS += tb - ta; tB1 += tc0 - tb; tM += td - tc0; tB2 += te - td; }
#endif
  }
  if (grp == 0) __syncthreads();
#ifdef FU_CONV_STAMPS
  if (P.dbg && (tid & 255) == 0) {
    unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 2 + grp) * 8;
    d[0] = tS; d[1] = tB1; d[2] = tM; d[3] = tB2; d[4] = (unsigned long long)max(K - 3, 0); d[5] = __builtin_amdgcn_s_memtime() - t0; d[6] = tPre;
  }
#endif
}

template <int NTW>
static int launch_pp_cfg(BConvP& P, hipStream_t s) {
  using Cfg = PPCfg<NTW>;
  P.tilesX = ceil_div(P.W, Cfg::TW); P.tilesY = ceil_div(P.H, Cfg::TH);
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = ceil_div(P.N, Cfg::BN);
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  FU_REQUIRE((int64_t)P.nPix * P.nCo * P.nPix < ((int64_t)1 << 32), "conv3x3_bf16_pp: grid too large (%d x %d)",
             P.nPix, P.nCo);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<NTW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const int nTiles = P.nPix * P.nCo;
  const int pairs = (nTiles + 1) / 2;
  const int grid = pairs >= 256 ? 256 : ((pairs + 7) / 8) * 8;   // one persistent workgroup per CU, multiple of 8
  const ProfSlot ps = g_prof_slot;
  g_prof_slot = ProfSlot();
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL((k_conv3x3_bf16_pp<NTW>), dim3(grid), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P, nTiles);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_conv3x3_bf16_pp(BConvP& P, hipStream_t s) {
  const int64_t t256 = (int64_t)P.B * ceil_div(P.H, 16) * ceil_div(P.W, 16);
  const bool wide = P.N >= 64 && t256 * ceil_div(P.N, 64) >= 512 && (!P.dst1 || P.D0 % 64 == 0);
  return wide ? launch_pp_cfg<2>(P, s) : launch_pp_cfg<1>(P, s);
}

}  // namespace fu
