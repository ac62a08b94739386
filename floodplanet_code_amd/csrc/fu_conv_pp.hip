// 16-bit 3x3 convolution, forward / dgrad: PERSISTENT PING-PONG row-stationary kernel for gfx950 (round 4).
// (compiled for bf16 and, with -DFU_HALF=1, fp16)
//
// Why.  k_conv3x3_bf16_rs (fu_conv_rs.hip) alternates stage() and mfma_block() between two barriers inside a workgroup and
// relies on a second, independent workgroup on the CU to fill the holes.  On a one-round grid the two co-resident workgroups
// start together, so both sit in their prologue (first loads: 12-17k cycles) and both in their epilogue (7-9k cycles) at the
// same time: with 2 or 8 chunks per tile those fixed costs are 50 % / 25 % of a workgroup's life (MFMA busy 0.48).
// Here ONE 8-wave workgroup per CU owns the CU for the whole launch and overlaps its own phases by construction:
//   * the arithmetic is the row-stationary kernel's: MFMA 16x16x32 with the weights as A (16 output channels x 32 input
//     channels) and one image row of 16 pixels as B; a fragment of input row r is read once and feeds the accumulators of
//     output rows r, r-1, r-2.  The workgroup tile is the same 16 columns x 32 rows x 64 channels; a wave owns 8 output rows
//     x 32 channels (64 accumulator registers instead of 128, which is what buys the registers for everything below).
//   * two LDS stages (2 x {39 KB activation halo tile, 36 KB weights}) and the 8 waves in two groups half a step apart
//     (group = channel half): while group 0 multiplies chunk s, group 1 converts / stores its half of chunk s+1 and DMAs
//     the weights group 0 will need; then the roles swap.  A SIMD hosts one wave of each group, so its matrix pipe sees one
//     wave's MFMAs beside the other wave's VALU / LDS-store work all the time (MI355X_MICROARCH.md, "Two waves per SIMD").
//   * persistent: a workgroup walks its tiles (static round-robin over the XCD-aware tile order) as ONE flat sequence of
//     (tile, chunk) steps -- the first chunk of the next tile is fetched and staged while the last chunk of this tile is
//     multiplied, and a group's epilogue (convert, statistics, stores) runs while the other group multiplies.
//   * every weight fragment a group reads was DMA'd by the OTHER group one phase earlier, and the halo rows a wave needs first
//     (rows with bit 2 clear) are staged by group 1: the first fragments of an MFMA phase are complete before the barrier
//     that opens it.
// Accumulation order per output element (chunk, column shift, kernel row) and the statistics' summation order are those of
// k_conv3x3_bf16_rs<8>: outputs, BatchNorm statistics and fused BatchNorm-backward sums are bit-identical to it (tested).
// Shapes (conv3x3_pp_eligible): the row-stationary kernel's, with H % 32 == 0.
#include "fu_conv_bf16.h"

namespace fu {

#if FU_HALF
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define k_conv3x3_bf16_pp k_conv3x3_f16_pp
#else
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PCfg {
  static constexpr int NT = 512, GT = 256, TW = 16, TH = 32, BN = 64, KC = 32;
  static constexpr int HWd = TW + 2, HHt = TH + 2, NHP = HHt * HWd;          // 18 x 34 = 612 halo pixels
  static constexpr int ROWB = 64;                                            // bytes per LDS row (32 channels)
  static constexpr int A_BYTES = NHP * ROWB, W_BYTES = 9 * BN * ROWB;        // 39168 + 36864
  static constexpr int STAGE = A_BYTES + W_BYTES;                            // 76032 per stage
  static constexpr int A_ITERS = 5;                                          // staging slots per thread (16-byte units)
  // group 1 stages the 16 halo rows with bit 2 clear below row 32 + most of rows 32 / 33: 5 full slots (1280 units);
  // group 0 the 16 rows with bit 2 set (1152 units) + the last 4 pixels of row 33 (16 units): 4 full slots + 144 threads
  static constexpr int G0_ROW_UNITS = 16 * HWd * 4, G0_UNITS = G0_ROW_UNITS + 16;
  static constexpr int AB_OFF = 2 * STAGE, AB_FLOATS = 1024;                 // BN scale / shift of source 0
  static constexpr int RED_OFF = AB_OFF + AB_FLOATS * 4, RED_FLOATS = 2 * 8 * 32 * 2;   // [tile parity][wave][32 ch][2]
  static constexpr int SMEM_BYTES = RED_OFF + RED_FLOATS * 4;                // 160256 <= 163840
  static constexpr int M_STEPS = 3 * (8 + 2);                                // (column shift, input row) steps per chunk
};

// BNB: the launch also emits the BatchNorm-backward sums of its destination (BnbFuse; never together with forward
// statistics or a BatchNorm prologue) -- a separate instantiation, because the y rows of that epilogue would otherwise
// set the register budget of every launch (223 registers without them, spills with them).
template <bool BNB>
__global__ __launch_bounds__(512) void k_conv3x3_bf16_pp(BConvP P) {
  using Cfg = PCfg;
  constexpr int HWd = Cfg::HWd, ROWB = Cfg::ROWB, KC = Cfg::KC, A_ITERS = Cfg::A_ITERS, STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sAB = reinterpret_cast<float*>(smem_raw + Cfg::AB_OFF);     // [2][512]
  float* sRed = reinterpret_cast<float*>(smem_raw + Cfg::RED_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, rg = wave & 3;                          // group = channel half; rg = 8-row group of the tile
  const int tg = tid & (Cfg::GT - 1);
  const int lx = lane & 15, lg = lane >> 4;                          // pixel column / channel row m, and k-group (8 channels)
  const int grid = gridDim.x;
  const int nChunks = P.Cin / KC;
  const bool has_bn = P.a0 != nullptr;

  auto decode = [&](int v, int& pixT, int& n0, int& x0, int& y0, int& bb) __attribute__((always_inline)) {
    const int logical = xcd_remap(v, P.nTiles);
    pixT = fast_div(logical, P.nCo, P.rcp_nCo);
    const int coT = logical - pixT * P.nCo;
    const int t2 = fast_div(pixT, P.tilesX, P.rcp_tilesX);
    const int tx = pixT - t2 * P.tilesX;
    bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
    const int ty = t2 - bb * P.tilesY;
    x0 = tx * Cfg::TW; y0 = ty * Cfg::TH; n0 = coT * Cfg::BN;
  };

  // ---- staging slots of this thread: unit ul = tg + 256 it of its group's share of the halo tile (thread constants)
  const int aq = tid & 3;
  unsigned lds_a[A_ITERS];
  unsigned a_live = 0;
  auto slot_yx = [&](int it, int& hy, int& hx) __attribute__((always_inline)) {   // halo row / column of slot it
    const int ul = tg + it * Cfg::GT;
    const int lp = ul >> 2;
    const int lr = (lp * 3641) >> 16;                                // lp / 18
    hx = lp - lr * HWd;
    hy = 8 * (lr >> 2) + (lr & 3) + (grp ? 0 : 4);
    if (!grp && ul >= Cfg::G0_ROW_UNITS) { hy = Cfg::HHt - 1; hx = min(14 + ((ul - Cfg::G0_ROW_UNITS) >> 2), HWd - 1); }
  };
  static_for<0, A_ITERS>([&](auto I) {
    constexpr int it = decltype(I)::value;
    int hy, hx;
    slot_yx(it, hy, hx);
    const bool live = grp || tg + it * Cfg::GT < Cfg::G0_UNITS;
    a_live |= live ? (1u << it) : 0u;
    lds_a[it] = (unsigned)((hy * HWd + hx) * ROWB + ((16 * aq) ^ ((hx & 4) << 3)));
  });

  // ---- cursors over the workgroup's flat sequence of (tile, chunk) steps; both saturate at the last step (the loads / stores
  //      past the end repeat it into a stage nobody reads any more: no conditional loads in the MFMA phase).
  //      LOAD cursor: the step whose activation units are requested next -- two steps ahead of their conversion: a staging
  //      phase converts one register set and refills it at once with the loads of the step after next (with one MFMA
  //      phase of cover the staging phase measured 2700-4200 cycles, most of it waiting for HBM).  The loads sit at the END
  //      of the staging phase: vmcnt counts in order, so behind the weight DMA they are not waited for by its vmcnt(0),
  //      and the MFMA phase carries no vector-memory instruction at all.
  //      STAGE cursor: the step that is converted / stored / whose weights are DMA'd next.
  const int R = (P.nTiles - (int)blockIdx.x + grid - 1) / grid;      // tiles of this workgroup (>= 1)
  const int T = R * nChunks;                                         // steps
  int lv = blockIdx.x, lk = 0, lstep = 0;
  int sv = blockIdx.x, sk = 0, sstep = 0;
  unsigned a_pix[A_ITERS];                                           // clamped pixel index of the slot (< 2^24: eligibility)
  unsigned a_ok = 0;                                                 // bit it: pixel inside the image (and slot live)
  size_t w_tile = 0;                                                 // byte offset of the staged tile's first weight row
  auto load_tile = [&](int v) __attribute__((always_inline)) {
    int pixT, n0, x0, y0, bb;
    decode(v, pixT, n0, x0, y0, bb);
    a_ok = 0;
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      int hy, hx;
      slot_yx(it, hy, hx);                                           // (recomputed per tile: five registers less in the loop)
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
      const bool ok = iy == cy && ix == cx;
      a_ok |= ok ? (1u << it) : 0u;
      a_pix[it] = (unsigned)((bb * P.H + cy) * P.W + cx);
    });
    a_ok &= a_live;
  };
  auto stage_tile = [&](int v) __attribute__((always_inline)) {
    int pixT, n0, x0, y0, bb;
    decode(v, pixT, n0, x0, y0, bb);
    w_tile = (size_t)n0 * (size_t)P.Cin * 2;
  };
  auto advance_load = [&]() __attribute__((always_inline)) {
    if (lstep + 1 < T) {
      ++lstep;
      if (++lk == nChunks) { lk = 0; lv += grid; load_tile(lv); }
    }
  };
  auto advance_stage = [&]() __attribute__((always_inline)) {
    if (sstep + 1 < T) {
      ++sstep;
      if (++sk == nChunks) { sk = 0; sv += grid; stage_tile(sv); }
    }
  };

  // ---- weight DMA.  LDS rows [tap][subtile S][16 rows m][64 B]; row m of subtile S is output channel
  //      32 (S >> 1) + 8 (m >> 2) + 4 (S & 1) + (m & 3): lane (lx, lg) of the wave of channel half h then holds the 8
  //      CONSECUTIVE channels 32 h + 8 lg + 4 q + i of its pixel in its accumulator registers (q = subtile & 1, i = 0..3).
  //      Lane i of a piece lands on row m = i >> 2, physical slot i & 3 = k-group (i & 3) ^ ((m & 4) >> 1).
  //      Group g DMAs the 18 pieces of the OTHER channel half: wave rg takes pieces j = rg + 4 t (tap j >> 1, q = j & 1).
  const int dm = lane >> 2, dgk = (lane & 3) ^ ((dm & 4) >> 1);
  const unsigned w_lane = (unsigned)((8 * (dm >> 2) + (dm & 3)) * P.Cin + 8 * dgk) * 2u;
  const size_t w_tap = (size_t)P.N * (size_t)P.Cin * 2;
  auto dma_weights = [&](int k0, unsigned sb) __attribute__((always_inline)) {
    const int oh = 1 - grp;                                          // the channel half this group stages
    const char* ub = reinterpret_cast<const char*>(P.wpk) + w_tile + (size_t)k0 * 2 + (size_t)(32 * oh) * (size_t)P.Cin * 2;
    unsigned wl = w_lane;
    asm volatile("" : "+v"(wl));                                     // (opaque: see fu_conv_rs.hip, hoisted 64-bit lane addresses)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int j = rg + 4 * t;
      if (j < 18) {                                                  // uniform per wave
        const int tap = j >> 1, q = j & 1;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ub + (size_t)tap * w_tap + (size_t)(4 * q) * (size_t)P.Cin * 2 + (size_t)wl),
            (__attribute__((address_space(3))) void*)(smem_raw + sb + Cfg::A_BYTES + tap * (Cfg::BN * ROWB) + (2 * oh + q) * 1024),
            16, 0, 0);
      }
    }
  };

  // ---- activation loads (registers) and the convert + store into an LDS stage
  uint4 ra[2][A_ITERS];                                              // two steps in flight
  unsigned okset[2] = {0u, 0u};                                      // the zero-padding mask travels with its set
  const char* abL = nullptr;
  unsigned cs2 = 0;                                                  // bytes per pixel of the current source (uniform)
  auto load_begin = [&](auto Set) __attribute__((always_inline)) {
    constexpr int set = decltype(Set)::value;
    okset[set] = a_ok;
    const int k0 = lk * KC;
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;                 // uniform
    abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
             : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    cs2 = (unsigned)(s1 ? P.C1 : P.C0) * 2u;
  };
  auto load_slot = [&](auto Set, auto Sc) __attribute__((always_inline)) {
    constexpr int set = decltype(Set)::value, sl = decltype(Sc)::value;
    ra[set][sl] = *reinterpret_cast<const uint4*>(abL + ((a_pix[sl] & 0xffffffu) * (cs2 & 0xffffffu) + 16u * aq));
  };
  auto store_chunk = [&](unsigned sb, auto Set, auto Bc) __attribute__((always_inline)) {
    constexpr bool BNR = decltype(Bc)::value;
    constexpr int set = decltype(Set)::value;
    unsigned okm = okset[set];
    asm volatile("" : "+v"(okm));
    const int cc = (BNR ? sk * KC : 0) + 8 * aq;                     // < 512: inside sAB
    f32x2 ca0, ca1, ca2, ca3, cb0, cb1, cb2, cb3;
    if constexpr (BNR) {
      const float4 a0 = *reinterpret_cast<const float4*>(sAB + cc);
      const float4 a1 = *reinterpret_cast<const float4*>(sAB + cc + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sAB + 512 + cc);
      const float4 b1 = *reinterpret_cast<const float4*>(sAB + 512 + cc + 4);
      ca0 = f32x2{a0.x, a0.y}; ca1 = f32x2{a0.z, a0.w}; ca2 = f32x2{a1.x, a1.y}; ca3 = f32x2{a1.z, a1.w};
      cb0 = f32x2{b0.x, b0.y}; cb1 = f32x2{b0.z, b0.w}; cb2 = f32x2{b1.x, b1.y}; cb3 = f32x2{b1.z, b1.w};
    }
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      if (it < A_ITERS - 1 || ((a_live >> it) & 1u)) {
        unsigned x = ra[set][it].x, y = ra[set][it].y, z = ra[set][it].z, w = ra[set][it].w;
        if constexpr (BNR) {
          x = bn_relu_pair(x, ca0, cb0); y = bn_relu_pair(y, ca1, cb1);
          z = bn_relu_pair(z, ca2, cb2); w = bn_relu_pair(w, ca3, cb3);
        }
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)okm, it, 1);     // bit it -> 0 / 0xffffffff (zero padding)
        x &= m; y &= m; z &= m; w &= m;
        *reinterpret_cast<uint4*>(smem_raw + sb + lds_a[it]) = make_uint4(x, y, z, w);
      }
    });
  };
  // one staging phase: this group's half of the step under the cursor -> stage `sb`, then the cursor moves on
  auto load_all = [&](auto Set) __attribute__((always_inline)) {
    load_begin(Set);
    static_for<0, A_ITERS>([&](auto Sc) { load_slot(Set, Sc); });
    advance_load();
  };
#ifdef FU_CONV_STAMPS     // diagnostic builds only (tools/stamp_pp.py): s_memtime sums per phase and wave
  unsigned long long tM = 0, tB1 = 0, tS = 0, tE = 0, tB2 = 0, tSd = 0, tSw = 0, tSc = 0, tSv = 0, tSl = 0;
#define FU_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FU_STAMP(v)
#endif
  auto stage_phase = [&](unsigned sb, auto Set) __attribute__((always_inline)) {
    const int k0 = sk * KC;
    FU_STAMP(q0);
    dma_weights(k0, sb);
    FU_STAMP(q1);
#ifdef FU_CONV_STAMPS
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");               // the set's five loads (older than the other set's and the DMA)
#endif
    FU_STAMP(q2);
    if (has_bn && k0 < P.C0) store_chunk(sb, Set, std::true_type{});
    else store_chunk(sb, Set, std::false_type{});
    FU_STAMP(q3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's LDS-DMA pieces have landed (the other set's
    advance_stage();                                                 //  loads too: they are a whole step old)
    FU_STAMP(q4);
    load_all(Set);                                                   // refill the set: the step after next
#ifdef FU_CONV_STAMPS
    const unsigned long long q5 = __builtin_amdgcn_s_memtime();
    tSd += q1 - q0; tSw += q2 - q1; tSc += q3 - q2; tSv += q4 - q3; tSl += q5 - q4;
#endif
  };

  // ---- MFMA phase
  f32x4 acc[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment bases per stage (bytes from smem_raw), opaque so that they stay one register each
  unsigned wfb[2], pfb[2][3];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    wfb[par] = (unsigned)(par * STAGE + Cfg::A_BYTES + (2 * grp) * 1024 + lx * ROWB + ((16 * lg) ^ ((lx & 4) << 3)));
    asm volatile("" : "+v"(wfb[par]));
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      pfb[par][dx] = (unsigned)(par * STAGE + (rg * 8 * HWd + lx + dx) * ROWB + ((16 * lg) ^ (((lx + dx) & 4) << 3)));
      asm volatile("" : "+v"(pfb[par][dx]));
    }
  }
  frag8_t wf[2][3][2];                     // [block parity][kernel row][subtile q]
#ifndef FU_PP_PD
#define FU_PP_PD 3
#endif
  constexpr int PD = FU_PP_PD, NR = PD + 1;
  static_assert(PD <= 4, "the rows requested ahead of the barrier (halo rows 8 rg + [0, PD)) must be group 1's: bit 2 clear");
  frag8_t pf[NR];                          // ring of input-row fragments, PD steps ahead of the MFMAs
  auto ld_w = [&](auto Par, auto Bk, auto Dy, auto Q) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value, bk = decltype(Bk)::value, dy = decltype(Dy)::value, q = decltype(Q)::value;
    wf[bk & 1][dy][q] = *reinterpret_cast<const frag8_t*>(smem_raw + wfb[par] + (dy * 3 + bk) * (Cfg::BN * ROWB) + q * 1024);
  };
  auto ld_p = [&](auto Par, auto Tc) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value, t = decltype(Tc)::value, dx = t / 10, ri = t % 10;
    pf[t % NR] = *reinterpret_cast<const frag8_t*>(smem_raw + pfb[par][dx] + ri * HWd * ROWB);
  };
  // the fragments of the first steps: staged by the other group (weights) / by group 1 (rows with bit 2 clear)
  auto mfma_prefetch = [&](auto Par) __attribute__((always_inline)) {
    static_for<0, 3>([&](auto Dy) {
      ld_w(Par, std::integral_constant<int, 0>{}, Dy, std::integral_constant<int, 0>{});
      ld_w(Par, std::integral_constant<int, 0>{}, Dy, std::integral_constant<int, 1>{});
    });
    static_for<0, PD>([&](auto Tc) { ld_p(Par, Tc); });
  };
  auto mfma_phase = [&](auto Par) __attribute__((always_inline)) {
    static_for<0, Cfg::M_STEPS>([&](auto Tc) {
      constexpr int t = decltype(Tc)::value, dx = t / 10, ri = t % 10;
      if constexpr (t + PD < Cfg::M_STEPS) ld_p(Par, std::integral_constant<int, t + PD>{});
      if constexpr (dx < 2 && ri < 6)      // the next column shift's six weight fragments, one per step
        ld_w(Par, std::integral_constant<int, dx + 1>{}, std::integral_constant<int, ri / 2>{}, std::integral_constant<int, ri % 2>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 3>([&](auto Yc) {
        constexpr int dy = decltype(Yc)::value, ro = ri - dy;
        if constexpr (ro >= 0 && ro < 8) {
          acc[ro][0] = FU_MFMA16(wf[dx & 1][dy][0], pf[t % NR], acc[ro][0]);
          acc[ro][1] = FU_MFMA16(wf[dx & 1][dy][1], pf[t % NR], acc[ro][1]);
        }
      });
    });
  };

  // ---- epilogue of the tile under the MFMA cursor: lane (pixel lx, group lg) holds channels n0 + 32 grp + 8 lg + [0, 8)
  int mv = blockIdx.x, mk = 0, tpar = 0, mstep = 0;
  int pend_pixT = 0, pend_n0 = 0, pend_par = 0;
  bool pend = false;
  auto row_sum = [](float v) __attribute__((always_inline)) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
    return v;
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    int pixT, n0, x0, y0, bb;
    decode(mv, pixT, n0, x0, y0, bb);
    const int nl = n0 + 32 * grp + 8 * lg;
    float* red = sRed + tpar * 512 + (grp * 4 + rg) * 64;            // [32 channels][2]
    uint4 yr[8];                                                     // BnbFuse: the y rows of this lane's pixels; requested first,
    if constexpr (BNB) {                                             // they land while the tile is converted and stored
      const char* ybase = reinterpret_cast<const char*>(P.bnb_y) +
          ((size_t)((bb * P.H + y0 + rg * 8) * P.W + x0 + lx) * (size_t)P.N + (size_t)nl) * 2;
#pragma unroll
      for (int ro = 0; ro < 8; ++ro) yr[ro] = *reinterpret_cast<const uint4*>(ybase + (size_t)ro * (size_t)(P.W * P.N) * 2);
    }
    const bool to0 = n0 < P.D0;                                      // uniform: D0 % 64 == 0 with two destinations
    char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + nl : P.dst1 + (nl - P.D0));
    const int dstride = to0 ? P.D0 : P.D1;
    float biasv[8];
    if (P.bias != nullptr) {
      const float4 b0 = *reinterpret_cast<const float4*>(P.bias + nl), b1 = *reinterpret_cast<const float4*>(P.bias + nl + 4);
      biasv[0] = b0.x; biasv[1] = b0.y; biasv[2] = b0.z; biasv[3] = b0.w;
      biasv[4] = b1.x; biasv[5] = b1.y; biasv[6] = b1.z; biasv[7] = b1.w;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) biasv[c] = 0.f;
    }
    float ssum[8], ssq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) {
      const int oy = y0 + rg * 8 + ro, ox = x0 + lx;
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v0 = acc[ro][q][0], v1 = acc[ro][q][1], v2 = acc[ro][q][2], v3 = acc[ro][q][3];
        if constexpr (!BNB) {
          ssum[4 * q + 0] += v0; ssum[4 * q + 1] += v1; ssum[4 * q + 2] += v2; ssum[4 * q + 3] += v3;
          ssq[4 * q + 0] = fmaf(v0, v0, ssq[4 * q + 0]); ssq[4 * q + 1] = fmaf(v1, v1, ssq[4 * q + 1]);
          ssq[4 * q + 2] = fmaf(v2, v2, ssq[4 * q + 2]); ssq[4 * q + 3] = fmaf(v3, v3, ssq[4 * q + 3]);
        }
        o[2 * q + 0] = pack_e2(f32x2{v0 + biasv[4 * q + 0], v1 + biasv[4 * q + 1]});
        o[2 * q + 1] = pack_e2(f32x2{v2 + biasv[4 * q + 2], v3 + biasv[4 * q + 3]});
        if constexpr (!BNB) acc[ro][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      char* dp = dbase + (size_t)((bb * P.H + oy) * P.W + ox) * (size_t)dstride * 2;
      *reinterpret_cast<uint4*>(dp) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if constexpr (!BNB) {
      if (P.stats) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float r1 = row_sum(ssum[c]), r2 = row_sum(ssq[c]);
          if (lx == 15) { red[(8 * lg + c) * 2 + 0] = r1; red[(8 * lg + c) * 2 + 1] = r2; }
        }
      }
    } else {
      // BatchNorm-backward sums of the destination (BnbFuse, fu_common.h): g = the accumulators (fp32, before their
      // rounding to the element type), y = the BatchNorm's raw input at the same pixels and channels; per channel
      // sum g*m and sum g*m*y over the 8 rows, the 16 pixels of a row (DPP), then the 4 row groups (LDS, by the combine).
      // Four channels at a time (their coefficients + sums = 16 registers beside the accumulators and the y rows).
      static_for<0, 2>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        const float4 a4 = *reinterpret_cast<const float4*>(P.bnb_a + nl + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(P.bnb_b + nl + 4 * q);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ro = 0; ro < 8; ++ro) {
          const unsigned w01 = q ? yr[ro].z : yr[ro].x, w23 = q ? yr[ro].w : yr[ro].y;
          const float yv[4] = {e2f_lo(w01), e2f_hi(w01), e2f_lo(w23), e2f_hi(w23)};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gm = fmaf(av[k], yv[k], bv[k]) > 0.f ? acc[ro][q][k] : 0.f;
            t1[k] += gm;
            t2[k] = fmaf(gm, yv[k], t2[k]);
          }
          acc[ro][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float r1 = row_sum(t1[k]), r2 = row_sum(t2[k]);
          if (lx == 15) { red[(8 * lg + 4 * q + k) * 2 + 0] = r1; red[(8 * lg + 4 * q + k) * 2 + 1] = r2; }
        }
      });
    }
    pend = true; pend_pixT = pixT; pend_n0 = n0; pend_par = tpar;
    tpar ^= 1;
    mv += grid;
  };
  // per-channel sums of a finished tile: the four row groups in a fixed order (wave 0; both groups' partial rows are in LDS)
  auto combine = [&]() __attribute__((always_inline)) {
    const int c = lane, hh = c >> 5, cc = c & 31;
    const float* red = sRed + pend_par * 512;
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) { s += red[((hh * 4 + m) * 32 + cc) * 2 + 0]; t += red[((hh * 4 + m) * 32 + cc) * 2 + 1]; }
    if constexpr (BNB) {
      const float iv = P.bnb_invstd[pend_n0 + c];
      const float nm = -P.bnb_mean[pend_n0 + c] * iv;
      float* o = P.bnb_part + ((int64_t)pend_pixT * P.N + pend_n0 + c) * 2;
      o[0] = s;
      o[1] = fmaf(iv, t, nm * s);                                    // invstd * s2 - mean * invstd * s1
    } else if (P.stats) {
      float* o = P.stats + ((int64_t)pend_pixT * P.N + pend_n0 + c) * 2;
      o[0] = s;
      o[1] = t;
    }
    pend = false;
  };

#ifdef FU_CONV_STAMPS
  const unsigned long long tStart = __builtin_amdgcn_s_memtime();
  const unsigned long long rStart = __builtin_amdgcn_s_memrealtime();
  unsigned long long tLoop = 0;
#endif

  // ---- prologue: coefficient tables, step 0 (both groups), step 1 (group 1, while group 0 multiplies step 0)
  if (has_bn) {
    for (int c = tid; c < P.C0; c += Cfg::NT) { sAB[c] = P.a0[c]; sAB[512 + c] = P.b0[c]; }
  }
  load_tile(lv);
  stage_tile(sv);
  // The loop's body<par> converts and refills set 1 - par.  Group 0 enters it after one staging phase (step 0 from set 0),
  // group 1 after two (step 0 from set 1, step 1 from set 0): steps 0 / 1 go to sets grp / 1 - grp.
  if (grp) { load_all(std::integral_constant<int, 1>{}); load_all(std::integral_constant<int, 0>{}); }
  else { load_all(std::integral_constant<int, 0>{}); load_all(std::integral_constant<int, 1>{}); }
  __syncthreads();                                                   // sAB
  if (grp) {
    stage_phase(0u, std::integral_constant<int, 1>{});
    __syncthreads();
    stage_phase((unsigned)STAGE, std::integral_constant<int, 0>{});
    mfma_prefetch(std::integral_constant<int, 0>{});                 // group 1: step 0 is complete since the last barrier
    __syncthreads();
  } else {
    stage_phase(0u, std::integral_constant<int, 0>{});
    __syncthreads();
  }
  if (!grp) mfma_prefetch(std::integral_constant<int, 0>{});         // group 0: behind the barrier (step 0 was being staged)

  auto body = [&](auto Par) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
#ifdef FU_CONV_STAMPS
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef FU_PP_MPRIO
    __builtin_amdgcn_s_setprio(FU_PP_MPRIO);
#endif
    mfma_phase(Par);
#ifdef FU_PP_MPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef FU_CONV_STAMPS
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
    const bool tile_end = mk + 1 == nChunks && mstep < T;            // uniform (an odd step count ends with a dummy step)
    ++mstep;
    // group 0 stages step s+1 into the other stage; group 1 stages step s+2 into the stage it has just multiplied (group 0
    // left it one phase ago)
    stage_phase((unsigned)((grp ? par : 1 - par) * STAGE), std::integral_constant<int, 1 - par>{});
#ifdef FU_CONV_STAMPS
    const unsigned long long s3 = __builtin_amdgcn_s_memtime();
#endif
    if (wave == 0 && pend) combine();
    if (tile_end) { epilogue(); mk = 0; } else { ++mk; }
    mfma_prefetch(std::integral_constant<int, 1 - par>{});           // first fragments of the next MFMA phase (complete: header)
#ifdef FU_CONV_STAMPS
    const unsigned long long s4 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long s5 = __builtin_amdgcn_s_memtime();
    tM += s1 - s0; tB1 += s2 - s1; tS += s3 - s2; tE += s4 - s3; tB2 += s5 - s4;
#endif
  };
#ifdef FU_CONV_STAMPS
  const unsigned long long tL0 = __builtin_amdgcn_s_memtime();
#endif
  for (int s = 0; s < T; s += 2) {
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
  }
#ifdef FU_CONV_STAMPS
  tLoop = __builtin_amdgcn_s_memtime() - tL0;
#endif
  if (!grp) __syncthreads();                                         // group 1's pre-loop barrier
  if (wave == 0 && pend) combine();
#ifdef FU_CONV_STAMPS
  if (P.dbg && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 8 + wave) * 16;
    d[0] = tM; d[1] = tB1; d[2] = tS; d[3] = tE; d[4] = tB2; d[5] = tLoop; d[6] = __builtin_amdgcn_s_memtime() - tStart;
    d[7] = __builtin_amdgcn_s_memrealtime() - rStart; d[8] = (unsigned long long)T; d[9] = tStart;
    d[10] = tSd; d[11] = tSw; d[12] = tSc; d[13] = tSv; d[14] = tSl;
  }
#endif
}

bool conv3x3_pp_eligible(const BConvP& P) {
  if (!conv3x3_rs_eligible(P)) return false;
  if ((P.H % 32) != 0) return false;
  const int64_t tiles = (int64_t)P.B * (P.H / 32) * (P.W / 16) * (P.N / 64);
  return tiles >= 8 && tiles < ((int64_t)1 << 24);
}

int launch_conv3x3_pp(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  using Cfg = PCfg;
  P.tilesX = P.W / Cfg::TW; P.tilesY = P.H / Cfg::TH;
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = P.N / Cfg::BN;
  P.nTiles = P.nPix * P.nCo;
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  P.rcp_nCo = host_rcp(P.nCo);
  FU_REQUIRE((int64_t)P.nTiles * P.nPix < ((int64_t)1 << 32) && (int64_t)P.nTiles * P.nCo < ((int64_t)1 << 32),
             "conv3x3_pp: grid too large (%d x %d)", P.nPix, P.nCo);
  // BatchNorm-backward sums of the destination, if the API layer asked for them and this launch can give them
  static const BnbFuse none;
  const BnbFuse& f = o.bnb ? *o.bnb : none;
  if (f.y != nullptr && f.tiles_out != nullptr && P.a0 == nullptr && P.dst1 == nullptr && P.stats == nullptr) {
    if ((int64_t)P.nPix * P.N * 2 <= f.max_elems) {
      P.bnb_y = (const bf16_t*)f.y; P.bnb_a = f.a; P.bnb_b = f.b; P.bnb_mean = f.mean; P.bnb_invstd = f.invstd;
      P.bnb_part = f.part;
      *f.tiles_out = P.nPix;
    }
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    FU_HIP_CHECK(hipGetDevice(&dev));
    FU_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
  }
  // one workgroup per CU, a multiple of 8 (the XCD of a workgroup's virtual block ids must not change from tile to tile)
  int grid = P.nTiles < n_cu ? P.nTiles : n_cu;
  grid &= ~7;
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  if (P.bnb_y != nullptr) hipLaunchKernelGGL(k_conv3x3_bf16_pp<true>, dim3(grid), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  else hipLaunchKernelGGL(k_conv3x3_bf16_pp<false>, dim3(grid), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

}  // namespace fu
