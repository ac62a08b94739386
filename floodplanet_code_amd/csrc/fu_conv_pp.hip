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
//     multiplied.  A tile's epilogue (convert, statistics, stores) ran beside the other group's MFMA phase in the first
//     builds; it now has a phase of its own, shared by both groups (see body: it is bound by vector-ALU cycles either way).
//   * every weight fragment a group reads was DMA'd by the OTHER group at least one phase earlier, and the halo rows a wave
//     needs first (rows with bit 2 clear) are converted by group 0, whose conversion of a step ends one phase before group
//     0's and two phases before group 1's MFMA phase of that step: the first fragments of an MFMA phase are complete
//     before the barrier that opens it and are requested ahead of that barrier.
// Accumulation order per output element (chunk, column shift, kernel row) and the statistics' summation order are those of
// k_conv3x3_bf16_rs<8>: outputs, BatchNorm statistics and fused BatchNorm-backward sums are bit-identical to it (tested).
// Shapes (conv3x3_pp_eligible): the row-stationary kernel's, with H % 32 == 0.
#include "fu_conv_bf16.h"
#include <stdlib.h>

namespace fu {

#if FU_HALF
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define FU_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#define FU_ASM_UNPK_LO "v_cvt_f32_f16 %0, %1"
#define FU_ASM_UNPK_HI "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define FU_ASM_PACK "v_cvt_pk_f16_f32 %0, %1, %2"
#define k_conv3x3_bf16_pp k_conv3x3_f16_pp
#else
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FU_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#define FU_ASM_UNPK_LO "v_lshlrev_b32 %0, 16, %1"
#define FU_ASM_UNPK_HI "v_and_b32 %0, 0xffff0000, %1"
#define FU_ASM_PACK "v_cvt_pk_bf16_f32 %0, %1, %2"
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PCfg {
  static constexpr int NT = 512, GT = 256, TW = 16, TH = 32, BN = 64, KC = 32;
  static constexpr int HWd = TW + 2, HHt = TH + 2, NHP = HHt * HWd;          // 18 x 34 = 612 halo pixels
  static constexpr int ROWB = 64;                                            // bytes per LDS row (32 channels)
  static constexpr int A_BYTES = NHP * ROWB, W_BYTES = 9 * BN * ROWB;        // 39168 + 36864
  static constexpr int STAGE = A_BYTES + W_BYTES;                            // 76032 per stage; LDS image: [A0][A1][W0][W1] (the
  static constexpr int W_OFF = 2 * A_BYTES;                                  //  A stages 39168 B apart: a ds_write immediate)
  static constexpr int A_ITERS = 5;                                          // staging slots per thread (16-byte units)
  // group 0 stages the 16 halo rows with bit 2 clear below row 32 + most of rows 32 / 33: 5 full slots (1280 units);
  // group 1 the 16 rows with bit 2 set (1152 units) + the last 4 pixels of row 33 (16 units): 4 full slots + 144 threads
  static constexpr int G1_ROW_UNITS = 16 * HWd * 4, G1_UNITS = G1_ROW_UNITS + 16;
  static constexpr int AB_OFF = 2 * STAGE, AB_FLOATS = 1024;                 // BN scale / shift of source 0
  static constexpr int RED_OFF = AB_OFF + AB_FLOATS * 4, RED_FLOATS = 2 * 8 * 32 * 2;   // [tile parity][wave][32 ch][2]
  static constexpr int BIAS_OFF = RED_OFF + RED_FLOATS * 4, BIAS_FLOATS = 512;   // forward launches: the conv bias (N <= 512)
  static constexpr int SMEM_BYTES = BIAS_OFF + BIAS_FLOATS * 4;              // 162304 <= 163840
  static constexpr int M_STEPS = 3 * (8 + 2);                                // (column shift, input row) steps per chunk
};

// ---- issue-slot plan of the MFMA phase (see mfma_phase): gaps = the 144 MFMAs of a chunk in order; per (column shift dx, input
// row ri) step 2 / 4 / 6 / ... / 6 / 4 / 2 MFMAs (rows 0, 1 and 8, 9 feed fewer output rows)
constexpr int pp_step_mfmas(int ri) { return ri == 0 || ri == 9 ? 2 : ri == 1 || ri == 8 ? 4 : 6; }
constexpr int pp_step_first_gap(int ri) { int g = 0; for (int r = 0; r < ri; ++r) g += pp_step_mfmas(r); return g; }
constexpr int pp_gap_free(int g) {          // free issue slots (of two) behind MFMA g of the phase
  const int dx = g / 48, gg = g % 48;
  int ri = 0;
  while (gg >= pp_step_first_gap(ri) + pp_step_mfmas(ri)) ++ri;
  const int k = gg - pp_step_first_gap(ri), nm = pp_step_mfmas(ri);
  const bool has_w = dx < 2 && ri < 6;
  int used = 0;
  if (k == 0) used += 1;                               // pixel-fragment read
  if (has_w && k == (nm > 2 ? 1 : 0)) used += 1;       // weight-fragment read
  if (k == nm - 1) used += 1;                          // s_waitcnt of the next step's first MFMA
  return used >= 2 ? 0 : 2 - used;
}
constexpr int pp_ops_before(int g) { int n = 0; for (int i = 0; i < g; ++i) n += pp_gap_free(i); return n; }

// BNB: the launch also emits the BatchNorm-backward sums of its destination (BnbFuse; never together with forward
// statistics or a BatchNorm prologue) -- a separate instantiation, because the y rows of that epilogue would otherwise
// set the register budget of every launch (223 registers without them, spills with them).
// BN: source 0 carries a BatchNorm + ReLU prologue (the conversion pieces exist only in this instantiation; a second,
// plain source takes a = 1, b = 0 and a NaN floor, so that the instruction stream of a step does not depend on its source --
// a run-time choice between two MFMA phases made hipcc spill 157 registers at the join).
// FWD: a forward launch -- bias and the per-tile BatchNorm statistics (sum, sum of squares) of the epilogue exist only here; a
// dgrad launch has neither, and its epilogue is a third of the vector instructions (beside the other group's MFMA phase the
// epilogue got the vector issue port only in the gaps that phase leaves: 5000-7500 cycles in the first build; see body).
template <bool BNB, bool BN, bool FWD>
__global__ __launch_bounds__(512) void k_conv3x3_bf16_pp(BConvP P) {
  using Cfg = PCfg;
  constexpr int HWd = Cfg::HWd, ROWB = Cfg::ROWB, KC = Cfg::KC, A_ITERS = Cfg::A_ITERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sAB = reinterpret_cast<float*>(smem_raw + Cfg::AB_OFF);     // [2][512]
  float* sRed = reinterpret_cast<float*>(smem_raw + Cfg::RED_OFF);
  float* sBias = reinterpret_cast<float*>(smem_raw + Cfg::BIAS_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, rg = wave & 3;                          // group = channel half; rg = 8-row group of the tile
  const int tg = tid & (Cfg::GT - 1);
  const int lx = lane & 15, lg = lane >> 4;                          // pixel column / channel row m, and k-group (8 channels)
  const int grid = gridDim.x;
  const int nChunks = P.Cin / KC;

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

  // ---- staging slots of this thread: unit ul = tg + 256 it of its group's share of the halo tile (thread constants).
  //      Group 0 owns the halo rows with bit 2 clear (0-3, 8-11, ..., 32, 33) except the last 4 pixels of row 33: 5 full
  //      slots; group 1 the rows with bit 2 set + those 4 pixels: 4 full slots + 144 threads -- its other 112 threads repeat
  //      their slot 3 in slot 4 (same address, same data: no dead slot, so no exec-masked store into which hipcc would sink
  //      the slot's whole conversion, away from the MFMAs).  The first rows a wave's MFMA phase reads, 8 rg + [0, PD), are
  //      group 0's: see mfma_prefetch.
  const int aq = tid & 3;
  unsigned lds_a[A_ITERS];
  auto slot_yx = [&](int it, int& hy, int& hx) __attribute__((always_inline)) {   // halo row / column of slot it
    int ul = tg + it * Cfg::GT;
    if (grp && ul >= Cfg::G1_UNITS) ul -= Cfg::GT;
    const int lp = ul >> 2;
    const int lr = (lp * 3641) >> 16;                                // lp / 18
    hx = lp - lr * HWd;
    hy = 8 * (lr >> 2) + (lr & 3) + (grp ? 4 : 0);
    if (grp && ul >= Cfg::G1_ROW_UNITS) { hy = Cfg::HHt - 1; hx = min(14 + ((ul - Cfg::G1_ROW_UNITS) >> 2), HWd - 1); }
  };
  static_for<0, A_ITERS>([&](auto I) {
    constexpr int it = decltype(I)::value;
    int hy, hx;
    slot_yx(it, hy, hx);
    lds_a[it] = (unsigned)((hy * HWd + hx) * ROWB + ((16 * aq) ^ ((hx & 4) << 3)));
  });

  // ---- cursors over the workgroup's flat sequence of (tile, chunk) steps; all saturate at the last step (loads, stores and
  //      DMA past the end repeat it into a stage nobody reads any more: nothing in the loop is conditional on the step).
  //      LOAD cursor: the step whose activation units are requested next; three steps ahead of the MFMA cursor: a register
  //      set is converted during one MFMA phase and refilled in the phase after it with the step after next (with a single
  //      phase of cover the conversion measured 2700-4200 cycles, most of it waiting for HBM).
  //      CONVERT cursor: the chunk converted / stored during the current MFMA phase (one step ahead of it).
  //      DMA cursor: the step whose weights (the OTHER group's half) are DMA'd next: one step ahead for group 0, two for
  //      group 1 (whose idle phase comes before the MFMA phase of the same step).
  const int R = (P.nTiles - (int)blockIdx.x + grid - 1) / grid;      // tiles of this workgroup (>= 1)
  const int T = R * nChunks;                                         // steps
  int lv = blockIdx.x, lk = 0, lstep = 0;
  int dv = blockIdx.x, dk = 0, dstep = 0;
  int ck = 0, cstep = 0;
  unsigned a_pix[A_ITERS];                                           // clamped pixel index of the slot (< 2^24: eligibility)
  unsigned a_ok = 0;                                                 // bit it: pixel inside the image
  size_t w_tile = 0;                                                 // byte offset of the DMA cursor's tile's first weight row
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
  };
  auto dma_tile = [&](int v) __attribute__((always_inline)) {
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
  auto advance_dma = [&]() __attribute__((always_inline)) {
    if (dstep + 1 < T) {
      ++dstep;
      if (++dk == nChunks) { dk = 0; dv += grid; dma_tile(dv); }
    }
  };
  auto advance_convert = [&]() __attribute__((always_inline)) {
    if (cstep + 1 < T) {
      ++cstep;
      if (++ck == nChunks) ck = 0;
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
  auto dma_weights = [&](int st) __attribute__((always_inline)) {   // the DMA cursor's step -> stage st; the cursor moves on
    const int oh = 1 - grp;                                          // the channel half this group stages
    const char* ub = reinterpret_cast<const char*>(P.wpk) + w_tile + (size_t)(dk * KC) * 2 + (size_t)(32 * oh) * (size_t)P.Cin * 2;
    unsigned wl = w_lane;
    asm volatile("" : "+v"(wl));                                     // (opaque: see fu_conv_rs.hip, hoisted 64-bit lane addresses)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int j = rg + 4 * t;
      if (j < 18) {                                                  // uniform per wave
        const int tap = j >> 1, q = j & 1;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ub + (size_t)tap * w_tap + (size_t)(4 * q) * (size_t)P.Cin * 2 + (size_t)wl),
            (__attribute__((address_space(3))) void*)(smem_raw + Cfg::W_OFF + st * Cfg::W_BYTES + tap * (Cfg::BN * ROWB) + (2 * oh + q) * 1024),
            16, 0, 0);
      }
    }
    advance_dma();
  };

  // ---- activation loads (registers): two sets = two steps in flight
  uint4 ra[2][A_ITERS];
  unsigned okset[2] = {0u, 0u};                                      // the zero-padding mask travels with its set
  auto load_all = [&](auto Set) __attribute__((always_inline)) {    // the load cursor's step -> set; the cursor moves on
    constexpr int set = decltype(Set)::value;
    okset[set] = a_ok;
    const int k0 = lk * KC;
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;                 // uniform
    const char* abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
                         : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    const unsigned cs2 = (unsigned)(s1 ? P.C1 : P.C0) * 2u;          // bytes per pixel of the source
    // uniform base + 32-bit lane offset = pixel * (channels * 2) + 16 aq: one v_mad_u32_u24 per load
    static_for<0, A_ITERS>([&](auto Sc) {
      constexpr int sl = decltype(Sc)::value;
      ra[set][sl] = *reinterpret_cast<const uint4*>(abL + ((a_pix[sl] & 0xffffffu) * (cs2 & 0xffffffu) + 16u * aq));
    });
    advance_load();
  };
  // ---- convert + store of a set into an LDS stage, in PIECES that the MFMA phase spreads between its MFMAs (one wave
  //      issues an MFMA every 16 cycles and the matrix pipe holds the vector issue port for 8 of them: its own VALU work in
  //      the other 8 is nearly free, while the SAME work issued by the partner wave of the SIMD measured 8-11 cycles per
  //      instruction -- 1900-2100 cycles for the 175 instructions of a BatchNorm + ReLU conversion, the phase's critical path).
  //      Piece (unit it, pair j < 4): relu(a * x + b) of one channel pair; piece (it, 4): zero-padding mask + ds_write_b128.
  f32x2 cva[4], cvb[4];                                              // BatchNorm scale / shift of this lane's 8 channels
  int cfloor = 0;                                                    // ReLU floor on the float's bits: 0, or INT_MIN (identity)
  auto convert_coeffs = [&]() __attribute__((always_inline)) {
    if (ck * KC < P.C0) {                                            // uniform: a chunk of the BatchNorm-activated source
      const int cc = ck * KC + 8 * aq;                               // < 512: inside sAB
      const float4 a0 = *reinterpret_cast<const float4*>(sAB + cc);
      const float4 a1 = *reinterpret_cast<const float4*>(sAB + cc + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sAB + 512 + cc);
      const float4 b1 = *reinterpret_cast<const float4*>(sAB + 512 + cc + 4);
      cva[0] = f32x2{a0.x, a0.y}; cva[1] = f32x2{a0.z, a0.w}; cva[2] = f32x2{a1.x, a1.y}; cva[3] = f32x2{a1.z, a1.w};
      cvb[0] = f32x2{b0.x, b0.y}; cvb[1] = f32x2{b0.z, b0.w}; cvb[2] = f32x2{b1.x, b1.y}; cvb[3] = f32x2{b1.z, b1.w};
      cfloor = 0;
    } else {                                                         // a chunk of the plain second source: identity
#pragma unroll
      for (int j = 0; j < 4; ++j) { cva[j] = f32x2{1.f, 1.f}; cvb[j] = f32x2{0.f, 0.f}; }
      cfloor = (int)0x80000000u;
    }
  };
  // The conversion as a sequence of SINGLE vector instructions, each pinned (an empty asm on its result: pure arithmetic has no
  // chain to a sched_barrier -- unpinned, the DAG linearisation gathers it at its use) so that the MFMA phase can put it into a
  // chosen issue slot.  Per unit: [BN: per channel pair: unpack lo | unpack hi | fma lo | max lo | fma hi | max hi | pack]
  // then [mask bit | and x | and y | and z | and w | ds_write_b128].
  float cvl[2] = {0.f, 0.f}, cvh[2] = {0.f, 0.f};                    // two channel pairs in flight
  unsigned cv_m = 0u;
  constexpr int OPU = BN ? 34 : 6, NOPS = OPU * A_ITERS;             // operations per unit / per step
  // Every vector instruction of the conversion is an inline-asm statement: behind an inline-asm MFMA hipcc pads each of ITS OWN
  // vector instructions with an s_nop (one issue slot each).  It also pads an asm statement that reads a register an asm
  // statement wrote less than ~3 instructions earlier (it has to take it for a transcendental), so two channel pairs A / B run
  // interleaved and every result is used 4 operations after it is made.  Per unit (BN): mask bit | 2 x { unpack loA hiA loB hiB |
  // fma loA hiA loB hiB | max loA hiA loB hiB | pack A B } | and x y z w | ds_write_b128.  ReLU = signed-integer max of the
  // float's bits against 0 (negative floats are negative integers; -0 -> +0 as v_max_f32 gives) or against INT_MIN (identity:
  // a chunk of the plain source).
  auto convert_op = [&](auto Stg, auto Set, auto Kc) __attribute__((always_inline)) {
    constexpr int stg = decltype(Stg)::value, set = decltype(Set)::value, k = decltype(Kc)::value;
    constexpr int it = k / OPU, r = k % OPU;
    constexpr int so = BN ? (r == 0 ? 0 : r >= 29 ? r - 28 : -1) : r;   // >= 0: mask / store part
    if constexpr (so < 0) {
      constexpr int d = (r - 1) / 14, op = (r - 1) % 14;             // double pair d (words 2 d, 2 d + 1), operation
      constexpr int ph = op / 4, ab = op < 12 ? (op % 4) / 2 : op - 12, hl = op % 2;   // phase, pair A / B, lo / hi
      constexpr int j = 2 * d + ab;
      // (component by component: a select between lvalues takes addresses and demotes the register array to scratch)
      if constexpr (op < 4) {
        unsigned w = 0;
        if constexpr (j == 0) w = ra[set][it].x;
        if constexpr (j == 1) w = ra[set][it].y;
        if constexpr (j == 2) w = ra[set][it].z;
        if constexpr (j == 3) w = ra[set][it].w;
        if constexpr (hl == 0) asm volatile(FU_ASM_UNPK_LO : "=v"(cvl[ab]) : "v"(w));
        else asm volatile(FU_ASM_UNPK_HI : "=v"(cvh[ab]) : "v"(w));
      } else if constexpr (ph == 1) {
        if constexpr (hl == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(cvl[ab]) : "v"(cva[j].x), "v"(cvb[j].x));
        else asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(cvh[ab]) : "v"(cva[j].y), "v"(cvb[j].y));
      } else if constexpr (ph == 2) {
        if constexpr (hl == 0) asm volatile("v_max_i32 %0, %1, %0" : "+v"(cvl[ab]) : "s"(cfloor));
        else asm volatile("v_max_i32 %0, %1, %0" : "+v"(cvh[ab]) : "s"(cfloor));
      } else {
        unsigned w;
        asm volatile(FU_ASM_PACK : "=v"(w) : "v"(cvl[ab]), "v"(cvh[ab]));
        if constexpr (j == 0) ra[set][it].x = w;
        if constexpr (j == 1) ra[set][it].y = w;
        if constexpr (j == 2) ra[set][it].z = w;
        if constexpr (j == 3) ra[set][it].w = w;
      }
    } else if constexpr (so == 0) {
      asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(cv_m) : "v"(okset[set]), "n"(it));   // bit it -> 0 / 0xffffffff (zero padding)
    } else if constexpr (so == 1) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(ra[set][it].x) : "v"(cv_m));
    } else if constexpr (so == 2) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(ra[set][it].y) : "v"(cv_m));
    } else if constexpr (so == 3) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(ra[set][it].z) : "v"(cv_m));
    } else if constexpr (so == 4) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(ra[set][it].w) : "v"(cv_m));
    } else {
      *reinterpret_cast<uint4*>(smem_raw + stg * Cfg::A_BYTES + lds_a[it]) = ra[set][it];   // (stage offset: an immediate)
    }
  };
  auto convert_all = [&](auto Stg, auto Set) __attribute__((always_inline)) {   // (prologue: step 0, no MFMAs beside it)
    if constexpr (BN) convert_coeffs();
    static_for<0, NOPS>([&](auto Kc) { convert_op(Stg, Set, Kc); });
    advance_convert();
  };
#ifdef FU_CONV_STAMPS     // diagnostic builds only (tools/stamp_pp.py): s_memtime sums per phase and wave
  unsigned long long tM = 0, tB1 = 0, tS = 0, tE = 0, tB2 = 0, tSd = 0, tSw = 0, tSc = 0, tSv = 0, tSl = 0;
#define FU_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FU_STAMP(v)
#endif

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
    wfb[par] = (unsigned)(Cfg::W_OFF + par * Cfg::W_BYTES + (2 * grp) * 1024 + lx * ROWB + ((16 * lg) ^ ((lx & 4) << 3)));
    asm volatile("" : "+v"(wfb[par]));
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      pfb[par][dx] = (unsigned)(par * Cfg::A_BYTES + (rg * 8 * HWd + lx + dx) * ROWB + ((16 * lg) ^ (((lx + dx) & 4) << 3)));
      asm volatile("" : "+v"(pfb[par][dx]));
    }
  }
  frag8_t wf[2][3][2];                     // [block parity][kernel row][subtile q]
#ifndef FU_PP_PD
#define FU_PP_PD 3
#endif
  constexpr int PD = FU_PP_PD, NR = PD + 2;   // (PD + 2 slots: the slot a read overwrites was last multiplied a whole step ago --
                                              //  with PD + 1 hipcc pads every such read with s_nop wait states behind the MFMA)
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
  // (Measured and dropped, round 4: the zero-padding mask and the ds_write_b128 of a unit -- a 5-dword LDS store holds the wave's
  //  issue for ~50 cycles, three MFMA gaps; five of them cost this phase ~245 cycles -- moved out into the idle phase, with group
  //  1 a whole step ahead so that its rows are complete in time (the first build's schedule).  The MFMA phase of a plain source
  //  went 2665 -> 2485 cycles, but the stores then run beside the OTHER group's MFMA phase and slow that one down: BatchNorm
  //  layers 6480 -> 6730 cycles per step, plain two-chunk layers 10030 -> 9720; in the bench step forward + dgrad got 5 % slower.
  //  Also measured, no change on any layer: each unit as two ds_write_b64 in two different gaps instead of one ds_write_b128.)
  // The MFMA phase of a step (stage Par) + this wave's conversion of the NEXT step's units (set 1 - Par -> stage 1 - Par).
  // Issue rules measured with tools/probes/mfma16_issue_probe.hip: a wave issues one instruction per 4-cycle turn of its SIMD;
  // v_mfma_f32_16x16x32 takes two turns and the matrix pipe 16 cycles, so exactly TWO other instructions of any kind (VALU,
  // ds_read, s_waitcnt, s_nop, SALU) fit behind an MFMA for free -- the third costs 4 cycles, the fourth 5 more -- and a partner
  // wave's VALU takes what the older wave leaves.  Hence: MFMAs in place (inline asm: hipcc renames the accumulators, dst !=
  // src C, and then pads the reuse of the freed registers with s_nop), every (MFMA + its slot instructions) fenced by a
  // sched_barrier, and the slots given out by a fixed plan (pp_gap_free): gap 0 of a (column shift, input row) step holds the
  // pixel-fragment read PD steps ahead, gap 1 the next column shift's weight fragment (steps 0-5), the last gap leaves a slot to
  // the s_waitcnt in front of the next step's first MFMA; everything else takes the conversion, one instruction per slot.
  auto mfma_phase = [&](auto Par) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
    // The fragments requested ahead of the barrier are "used" here, before any new read is issued: hipcc's waitcnt pass loses
    // count of them across the barrier / loop edge and would otherwise wait with lgkmcnt(0) at their first MFMA.
    auto touch = [](const frag8_t& f) __attribute__((always_inline)) { asm volatile("" :: "v"(f)); };
    static_for<0, 3>([&](auto Dy) { touch(wf[0][decltype(Dy)::value][0]); touch(wf[0][decltype(Dy)::value][1]); });
    static_for<0, PD>([&](auto Tc) { touch(pf[decltype(Tc)::value]); });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BN) convert_coeffs();
    static_for<0, Cfg::M_STEPS>([&](auto Tc) {
      constexpr int t = decltype(Tc)::value, dx = t / 10, ri = t % 10;
      constexpr int nm = pp_step_mfmas(ri);
      static_for<0, nm>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value;
        constexpr int dy = (ri < 8 ? 0 : ri - 7) + k / 2, q = k % 2, ro = ri - dy;
        static_assert(ro >= 0 && ro < 8, "kernel row of the k-th MFMA of a step");
        asm volatile(FU_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc[ro][q]) : "v"(wf[dx & 1][dy][q]), "v"(pf[t % NR]));
#ifndef FU_PP_EXP_NOREAD     // (timing experiments of diagnostic builds: wrong results)
        if constexpr (k == 0 && t + PD < Cfg::M_STEPS) ld_p(Par, std::integral_constant<int, (t + PD < Cfg::M_STEPS ? t + PD : 0)>{});
        if constexpr (k == (nm > 2 ? 1 : 0) && dx < 2 && ri < 6)     // the next column shift's six weight fragments, one per step
          ld_w(Par, std::integral_constant<int, dx + 1>{}, std::integral_constant<int, (ri < 6 ? ri / 2 : 0)>{}, std::integral_constant<int, ri % 2>{});
#endif
#ifndef FU_PP_EXP_NOCONV
        constexpr int g = 48 * dx + pp_step_first_gap(ri) + k;       // gap index in the phase
        constexpr int o0 = pp_ops_before(g), o1 = o0 + pp_gap_free(g);
        static_for<o0, (o1 < NOPS ? o1 : NOPS)>([&](auto Oc) {
          convert_op(std::integral_constant<int, 1 - par>{}, std::integral_constant<int, 1 - par>{}, Oc);
        });
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    static_assert(pp_ops_before(144) >= NOPS, "not enough issue slots for the conversion");
    advance_convert();
  };

  int mv = blockIdx.x, mk = 0, tpar = 0, mstep = 0;
  // ---- epilogue of the tile under the MFMA cursor: lane (pixel lx, group lg) holds channels n0 + 32 grp + 8 lg + [0, 8)
  int pend_pixT = 0, pend_n0 = 0, pend_par = 0;
  bool pend = false;
  // Sum over the 16 lanes (pixels) of a row group: row_shr 1, 2, 4, 8, lane 15 of every row ends with the total.  ONE
  // v_add_f32_dpp per step and value (hipcc emits v_mov_b32_dpp + v_add_f32 for the update_dpp form: twice the instructions),
  // the N values of a lane interleaved so that a value's next step is N instructions away (a DPP read needs two wait states
  // behind the VALU write of its operand).  v + shr(v) is the same sum as the two-instruction form: bit-identical results.
  auto row_sums = [](auto& v) __attribute__((always_inline)) {
    constexpr int N = sizeof(v) / sizeof(float);
    static_assert(N >= 4, "interleave too short for the DPP wait states");
    asm volatile("s_nop 1");        // (the last of the values may have been written by the instruction in front: hipcc does not
                                    //  know these are DPP reads and pads nothing)
#define FU_RS_STEP(CTRL) _Pragma("unroll") for (int c = 0; c < N; ++c) \
      asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[c]));
    FU_RS_STEP("row_shr:1") FU_RS_STEP("row_shr:2") FU_RS_STEP("row_shr:4") FU_RS_STEP("row_shr:8")
#undef FU_RS_STEP
  };
  // BnbFuse: the y rows of this lane's pixels, requested at the start of the epilogue: they land while the tile is converted
  // and stored.  (Measured and dropped: requested at the start of the epilogue's phase, the weight-DMA wait behind them stands
  // still for an HBM latency, 125 -> 154 us on 64 -> 64 at 256 x 256; requested a step ahead under a uniform branch, hipcc's
  // waitcnt pass merges the two paths into s_waitcnt vmcnt(0) in front of the next MFMA phase's first ds_write, 149 us.)
  uint4 yr[8];
  auto bnb_request = [&]() __attribute__((always_inline)) {
    int pixT, n0, x0, y0, bb;
    decode(mv, pixT, n0, x0, y0, bb);
    const char* ybase = reinterpret_cast<const char*>(P.bnb_y) +
        ((size_t)((bb * P.H + y0 + rg * 8) * P.W + x0 + lx) * (size_t)P.N + (size_t)(n0 + 32 * grp + 8 * lg)) * 2;
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) yr[ro] = *reinterpret_cast<const uint4*>(ybase + (size_t)ro * (size_t)(P.W * P.N) * 2);
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    FU_STAMP(e0);
    if constexpr (BNB) bnb_request();
    int pixT, n0, x0, y0, bb;
    decode(mv, pixT, n0, x0, y0, bb);
    const int nl = n0 + 32 * grp + 8 * lg;
    float* red = sRed + tpar * 512 + (grp * 4 + rg) * 64;            // [32 channels][2]
    const bool to0 = n0 < P.D0;                                      // uniform: D0 % 64 == 0 with two destinations
    char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + nl : P.dst1 + (nl - P.D0));
    const int dstride = to0 ? P.D0 : P.D1;
    // The epilogue has a phase to itself (see body): no MFMA runs beside it, so (i) the sums, the sums of squares and the bias
    // add are PACKED fp32 operations (v_pk_add_f32 / v_pk_fma_f32: the same IEEE operations per component -- slow only beside
    // a co-resident MFMA stream), and (ii) the accumulators are cleared by the idle matrix pipe itself: 0 x 0 + 0 through one
    // MFMA per accumulator tile (16 issues) instead of 64 v_mov_b32.
    f32x2 bias2[4], s2[4], q2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { bias2[c] = f32x2{0.f, 0.f}; s2[c] = f32x2{0.f, 0.f}; q2[c] = f32x2{0.f, 0.f}; }
    if constexpr (FWD) {
      const float4 b0 = *reinterpret_cast<const float4*>(sBias + nl), b1 = *reinterpret_cast<const float4*>(sBias + nl + 4);
      bias2[0] = f32x2{b0.x, b0.y}; bias2[1] = f32x2{b0.z, b0.w}; bias2[2] = f32x2{b1.x, b1.y}; bias2[3] = f32x2{b1.z, b1.w};
    }
    typedef unsigned u32x4z __attribute__((ext_vector_type(4)));
    u32x4z zbits = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zbits));                                  // (opaque: the clearing MFMAs are not folded into moves)
    const frag8_t zf = __builtin_bit_cast(frag8_t, zbits);
    auto clear = [&](f32x4& a) __attribute__((always_inline)) { a = FU_MFMA16(zf, zf, (f32x4{0.f, 0.f, 0.f, 0.f})); };
    const size_t rowb = (size_t)P.W * (size_t)dstride * 2;           // one address product per tile, one 64-bit add per row
    char* dp = dbase + (size_t)((bb * P.H + y0 + rg * 8) * P.W + x0 + lx) * (size_t)dstride * 2;
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) {
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x2 lo = {acc[ro][q][0], acc[ro][q][1]}, hi = {acc[ro][q][2], acc[ro][q][3]};
        if constexpr (FWD) {
          s2[2 * q + 0] += lo; s2[2 * q + 1] += hi;
          q2[2 * q + 0] = __builtin_elementwise_fma(lo, lo, q2[2 * q + 0]);
          q2[2 * q + 1] = __builtin_elementwise_fma(hi, hi, q2[2 * q + 1]);
          o[2 * q + 0] = pack_e2(lo + bias2[2 * q + 0]);
          o[2 * q + 1] = pack_e2(hi + bias2[2 * q + 1]);
        } else {
          o[2 * q + 0] = pack_e2(lo);
          o[2 * q + 1] = pack_e2(hi);
        }
        if constexpr (!BNB) clear(acc[ro][q]);
      }
      *reinterpret_cast<uint4*>(dp) = make_uint4(o[0], o[1], o[2], o[3]);
      dp += rowb;
    }
    FU_STAMP(e1);
    float st[16];                                                    // [0, 8): sums, [8, 16): sums of squares
#pragma unroll
    for (int c = 0; c < 4; ++c) { st[2 * c] = s2[c].x; st[2 * c + 1] = s2[c].y; st[8 + 2 * c] = q2[c].x; st[8 + 2 * c + 1] = q2[c].y; }
    if constexpr (FWD) {
      if (P.stats) {
        row_sums(st);
        if (lx == 15) {
#pragma unroll
          for (int c = 0; c < 8; ++c) { red[(8 * lg + c) * 2 + 0] = st[c]; red[(8 * lg + c) * 2 + 1] = st[8 + c]; }
        }
      }
    }
    if constexpr (BNB) {
      // BatchNorm-backward sums of the destination (BnbFuse, fu_common.h): g = the accumulators (fp32, before their
      // rounding to the element type), y = the BatchNorm's raw input at the same pixels and channels; per channel
      // sum g*m and sum g*m*y over the 8 rows, the 16 pixels of a row (DPP), then the 4 row groups (LDS, by the combine).
      // Four channels at a time (their coefficients + sums = 16 registers beside the accumulators and the y rows).
      static_for<0, 2>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        const float4 a4 = *reinterpret_cast<const float4*>(P.bnb_a + nl + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(P.bnb_b + nl + 4 * q);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
        float tt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // [0, 4): sum g*m, [4, 8): sum g*m*y
#pragma unroll
        for (int ro = 0; ro < 8; ++ro) {
          const unsigned w01 = q ? yr[ro].z : yr[ro].x, w23 = q ? yr[ro].w : yr[ro].y;
          const float yv[4] = {e2f_lo(w01), e2f_hi(w01), e2f_lo(w23), e2f_hi(w23)};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gm = fmaf(av[k], yv[k], bv[k]) > 0.f ? acc[ro][q][k] : 0.f;
            tt[k] += gm;
            tt[4 + k] = fmaf(gm, yv[k], tt[4 + k]);
          }
          clear(acc[ro][q]);
        }
        row_sums(tt);
        if (lx == 15) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { red[(8 * lg + 4 * q + k) * 2 + 0] = tt[k]; red[(8 * lg + 4 * q + k) * 2 + 1] = tt[4 + k]; }
        }
      });
    }
#ifdef FU_CONV_STAMPS
    { const unsigned long long e2 = __builtin_amdgcn_s_memtime(); tSd += e1 - e0; tSl += e2 - e1; }
#endif
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
    } else if (FWD && P.stats) {
      float* o = P.stats + ((int64_t)pend_pixT * P.N + pend_n0 + c) * 2;
      o[0] = s;
      o[1] = t;
    }
    pend = false;
  };

#ifdef FU_CONV_STAMPS
  const unsigned long long tStart = __builtin_amdgcn_s_memtime();
  unsigned long long tPro = 0;
  const unsigned long long rStart = __builtin_amdgcn_s_memrealtime();
  unsigned long long tLoop = 0;
#endif

  // s_waitcnt vmcnt(5) through the builtin (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14):
  // hipcc's waitcnt pass sees it and knows the LDS-DMA pieces have landed -- behind an asm wait it still holds them pending
  // and puts s_waitcnt vmcnt(0) in front of the first ds_write of the next MFMA phase, i.e. waits for the loads that were
  // issued to stay in flight for two phases.
  auto vm_wait5 = []() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(5 | (7 << 4) | (15 << 8)); };
  // Workgroup barrier of the loop: LDS traffic of this wave done (lgkmcnt), then s_barrier.  __syncthreads() also waits with
  // vmcnt(0) -- for the activation loads that were issued to stay in flight across two phases (and for the epilogue's stores).
  // The LDS-DMA pieces, which do count in vmcnt, are waited for explicitly where they are issued.
  auto wg_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // ---- prologue: coefficient tables, step 0 (both groups convert their halves and DMA the other's weights), the loads of
  //      steps 1 and 2; group 1 then DMAs group 0's weights of step 1 while group 0 runs its first MFMA phase
  if constexpr (BN) {
    for (int c = tid; c < P.C0; c += Cfg::NT) { sAB[c] = P.a0[c]; sAB[512 + c] = P.b0[c]; }
  }
  if constexpr (FWD) {       // the bias from LDS: a global load at the head of the epilogue was an exposed L2 latency per tile
    for (int c = tid; c < P.N; c += Cfg::NT) sBias[c] = P.bias != nullptr ? P.bias[c] : 0.f;
  }
  // (measured, no effect on any layer: s_setprio 1 for the younger half once at kernel start -- MI355X_MICROARCH.md "Two waves per
  //  SIMD" item 4 --, and s_setprio 1 / 2 around the MFMA phase)
  load_tile(lv);
  dma_tile(dv);
  load_all(std::integral_constant<int, 0>{});                        // step 0 -> set 0
  dma_weights(0);                                                    // step 0's weights (LDS is free at kernel start)
  load_all(std::integral_constant<int, 1>{});                        // step 1 -> set 1 (behind the DMA: stays in flight below)
  vm_wait5();                                                        // step 0's loads, the DMA pieces (and the coefficient tables)
  wg_barrier();                                                      // sAB
  convert_all(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});   // step 0 (set 0) -> stage 0
  load_all(std::integral_constant<int, 0>{});                        // step 2 -> set 0
  wg_barrier();                                                      // step 0 complete
  if (grp) {
    dma_weights(1);                                                  // group 0's weights of step 1
    mfma_prefetch(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));           // vmcnt(0) (once: the loads of steps 1, 2 are older than the DMA)
    wg_barrier();
  } else {
    mfma_prefetch(std::integral_constant<int, 0>{});
  }

  // One step of a group: { MFMA phase + conversion of the next step ; barrier ; weight DMA, refill of the converted set,
  // [sums of the previous tile], first fragments of the next MFMA phase ; barrier }.  Group 1 runs one phase behind group 0,
  // so on every SIMD one wave multiplies while the other moves data.
  // The epilogue of a tile gets a phase of its own, the same one for both groups: { ... ; barrier ; data movement ; barrier ;
  // EPILOGUE ; barrier } in group 0, { ... ; barrier ; EPILOGUE ; barrier ; data movement ; barrier } in group 1 (one phase
  // behind).  Its 450 (forward, with statistics) to 900 (BatchNorm-backward sums) vector instructions ran beside the other
  // group's MFMA phase at ~10 cycles each and stretched TWO phases per tile from ~2700 to 5000-7500 cycles (group 0's epilogue
  // beside group 1's last MFMA phase, group 1's beside group 0's first of the next tile); with both groups' epilogues in one
  // phase without MFMAs they issue at the VALU's own rate and the matrix pipe idles once per tile instead of waiting twice.
#ifndef FU_PP_JOINT_EPILOGUE
#define FU_PP_JOINT_EPILOGUE 1      // (0: the epilogue inside the data-movement phase, A/B builds)
#endif
  auto body = [&](auto Par) __attribute__((always_inline)) {
    constexpr int par = decltype(Par)::value;
    FU_STAMP(s0);
#ifdef FU_PP_MPRIO
    __builtin_amdgcn_s_setprio(FU_PP_MPRIO);
#endif
    mfma_phase(Par);
#ifdef FU_PP_MPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    FU_STAMP(s1);
    wg_barrier();
    FU_STAMP(s2);
    const bool tile_end = mk + 1 == nChunks && mstep < T;            // uniform (an odd step count ends with a dummy step)
    ++mstep;
    auto move_data = [&]() __attribute__((always_inline)) {
      // group 0: the weights of step s+1 (stage 1 - par: group 1 left it one phase ago); group 1: of step s+2 (stage par: group
      // 0 multiplied it one phase ago, and group 1's own half of that stage is not touched)
      dma_weights(grp ? par : 1 - par);
      load_all(std::integral_constant<int, 1 - par>{});              // refill the set this phase's conversion has emptied
      vm_wait5();                                                    // the DMA pieces have landed (the five loads stay in flight)
      if (wave == 0 && pend) combine();
    };
#if FU_PP_JOINT_EPILOGUE
    if (tile_end && grp) {                                           // (uniform per wave)
      epilogue(); mk = 0;
      FU_STAMP(s3);
      wg_barrier();
      FU_STAMP(s3a);
      move_data();
      FU_STAMP(s3b);
      mfma_prefetch(std::integral_constant<int, 1 - par>{});
      FU_STAMP(s4);
      wg_barrier();
#ifdef FU_CONV_STAMPS
      const unsigned long long s5 = __builtin_amdgcn_s_memtime();
      tM += s1 - s0; tB1 += s2 - s1; tSw += s3 - s2; tS += s3b - s3a; tE += s4 - s3b; tB2 += s5 - s4; tSv += s3a - s3; tSc += 1;
#endif
    } else if (tile_end) {
      move_data();
      FU_STAMP(s3);
      wg_barrier();
      FU_STAMP(s3a);
      epilogue(); mk = 0;
      FU_STAMP(s3b);
      mfma_prefetch(std::integral_constant<int, 1 - par>{});
      FU_STAMP(s4);
      wg_barrier();
#ifdef FU_CONV_STAMPS
      const unsigned long long s5 = __builtin_amdgcn_s_memtime();
      tM += s1 - s0; tB1 += s2 - s1; tS += s3 - s2; tSw += s3b - s3a; tE += s4 - s3b; tB2 += s5 - s4; tSv += s3a - s3; tSc += 1;
#endif
    } else {
      move_data();
      ++mk;
      FU_STAMP(s3);
      mfma_prefetch(std::integral_constant<int, 1 - par>{});         // first fragments of the next MFMA phase (complete: header)
      FU_STAMP(s4);
      wg_barrier();
#ifdef FU_CONV_STAMPS
      const unsigned long long s5 = __builtin_amdgcn_s_memtime();
      tM += s1 - s0; tB1 += s2 - s1; tS += s3 - s2; tE += s4 - s3; tB2 += s5 - s4;
#endif
    }
#else
    move_data();
    FU_STAMP(s3);
    if (tile_end) { epilogue(); mk = 0; } else { ++mk; }
    FU_STAMP(s3b);
    mfma_prefetch(std::integral_constant<int, 1 - par>{});           // first fragments of the next MFMA phase (complete: header)
    FU_STAMP(s4);
    wg_barrier();
#ifdef FU_CONV_STAMPS
    const unsigned long long s5 = __builtin_amdgcn_s_memtime();
    tM += s1 - s0; tB1 += s2 - s1; tS += s3 - s2; tE += s4 - s3; tB2 += s5 - s4;
    tSw += s3b - s3; tSc += tile_end ? 1 : 0;
#endif
#endif
  };
#ifdef FU_CONV_STAMPS
  const unsigned long long tL0 = __builtin_amdgcn_s_memtime();
  tPro = tL0 - tStart;
#endif
  for (int s = 0; s < T; s += 2) {
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
  }
#ifdef FU_CONV_STAMPS
  tLoop = __builtin_amdgcn_s_memtime() - tL0;
#endif
  if (!grp) wg_barrier();                                            // group 1's pre-loop barrier
  if (wave == 0 && pend) combine();
#ifdef FU_CONV_STAMPS
  if (P.dbg && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* d = P.dbg + ((size_t)blockIdx.x * 8 + wave) * 16;
    d[0] = tM; d[1] = tB1; d[2] = tS; d[3] = tE; d[4] = tB2; d[5] = tLoop; d[6] = __builtin_amdgcn_s_memtime() - tStart;
    d[7] = __builtin_amdgcn_s_memrealtime() - rStart; d[8] = (unsigned long long)T; d[9] = tStart;
    d[10] = tSd; d[11] = tSw; d[12] = tSc; d[13] = tSv; d[14] = tSl; d[15] = tPro;
  }
#endif
}

bool conv3x3_pp_eligible(const BConvP& P) {
  if (!conv3x3_rs_eligible(P)) return false;
  if ((P.H % 32) != 0) return false;
  if ((P.bias != nullptr || P.stats != nullptr) && P.N > PCfg::BIAS_FLOATS) return false;   // (the LDS bias table of the forward launches)
  const int64_t tiles = (int64_t)P.B * (P.H / 32) * (P.W / 16) * (P.N / 64);
  return tiles >= 8 && tiles < ((int64_t)1 << 24);
}

// default dispatch: only where every CU gets at least one tile (one workgroup per CU for the whole launch: 128 tiles leave half
// the chip idle -- 512 -> 256 at 32 x 32 measured 54.8 us against 50.5 on the two-workgroup kernel)
bool conv3x3_pp_preferred(const BConvP& P) {
  const int64_t tiles = (int64_t)P.B * (P.H / 32) * (P.W / 16) * (P.N / 64);
  return conv3x3_pp_eligible(P) && tiles >= 256;
}

// ... of a dgrad launch that is asked for the BatchNorm-backward sums of its destination: from 8 chunks on.  The sums are ~450
// vector instructions per wave and tile in the epilogue, which ran beside the other group's MFMA phase at ~8 cycles per
// instruction; on the two-chunk 256 x 256 layers that is the kernel's critical path (64 -> 64: 125 us against 114 on the
// two-workgroup kernel, whose second workgroup covers it); from 256 input channels on it disappears (512 -> 512 at 32 x 32: 63
// against 74 us).  With the epilogue in a phase of its own (body) the picture is the same -- one-stream trace, pp | rs<8>: 64 -> 64
// at 256 x 256 125 / 117 | 115 / 114 us, 128 -> 64 at 128 x 128 65 | 54, 256 -> 128 at 64 x 64 49 | 44, 128 -> 128 at 128 x 128 92 | 83
// (-DFU_PP_BNB_MIN_CIN=64 builds).
#ifndef FU_PP_BNB_MIN_CIN
#define FU_PP_BNB_MIN_CIN 256
#endif
bool conv3x3_pp_preferred_bnb(const BConvP& P) {
  return conv3x3_pp_preferred(P) && P.Cin >= FU_PP_BNB_MIN_CIN;
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
    const void* ks[5] = {reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<false, false, false>),
                         reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<false, false, true>),
                         reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<false, true, false>),
                         reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<false, true, true>),
                         reinterpret_cast<const void*>(&k_conv3x3_bf16_pp<true, false, false>)};
    for (const void* k : ks) FU_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
  }
  // one workgroup per CU, a multiple of 8 (the XCD of a workgroup's virtual block ids must not change from tile to tile)
  int grid = P.nTiles < n_cu ? P.nTiles : n_cu;
#ifdef FU_EXPERIMENTS     // variant builds only: a smaller grid leaves CUs to the weight-gradient stream
  { static const int cap = [] { const char* e = getenv("FU_PP_GRID"); return e ? atoi(e) : 0; }(); if (cap > 0 && grid > cap && (P.bias == nullptr && P.stats == nullptr)) grid = cap; }
#endif
  grid &= ~7;
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  const bool fwd = P.bias != nullptr || P.stats != nullptr, bn = P.a0 != nullptr;
  const dim3 g(grid), b(Cfg::NT);
  if (P.bnb_y != nullptr) hipLaunchKernelGGL((k_conv3x3_bf16_pp<true, false, false>), g, b, Cfg::SMEM_BYTES, s, P);
  else if (bn && fwd) hipLaunchKernelGGL((k_conv3x3_bf16_pp<false, true, true>), g, b, Cfg::SMEM_BYTES, s, P);
  else if (bn) hipLaunchKernelGGL((k_conv3x3_bf16_pp<false, true, false>), g, b, Cfg::SMEM_BYTES, s, P);
  else if (fwd) hipLaunchKernelGGL((k_conv3x3_bf16_pp<false, false, true>), g, b, Cfg::SMEM_BYTES, s, P);
  else hipLaunchKernelGGL((k_conv3x3_bf16_pp<false, false, false>), g, b, Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

}  // namespace fu
