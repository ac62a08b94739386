// 16-bit 3x3 convolution, forward / dgrad, "row-stationary" kernel for gfx950 (compiled for bf16 and, with -DFU_HALF=1, fp16).
//
// Why another kernel.  k_conv3x3_bf16_fast reads one LDS fragment per MFMA (0.75 with its tall tile): at two workgroups per
// CU that is 84 % of the LDS array's cycles, and its MFMA block measures 4100-4700 cycles for 2304 cycles of MFMA work
// (DESIGN.md section 5).  This kernel restructures the implicit GEMM so that a fragment read from LDS feeds 6 (pixels) or
// 16 (weights) MFMAs:
//   * MFMA shape 16x16x32 with the WEIGHTS as the A operand (16 output channels x 32 input channels) and ONE IMAGE ROW OF
//     16 PIXELS as the B operand (32 input channels x 16 pixels).
//   * A wave owns 8 output rows x 16 columns x 64 channels = 32 accumulator tiles (128 registers).  For a column shift dx
//     it holds the weight fragments of the three kernel rows dy for two 16-channel subtiles (24 registers) and walks the 10
//     input rows of its halo: the fragment of input row r (shifted by dx) is read ONCE and multiplied into the
//     accumulators of output rows r, r-1, r-2 -- the 3x3 window's vertical taps share it.  The vertical shift costs
//     nothing: it is only a choice of accumulator.  Per 32-channel chunk: 288 MFMAs, 36 weight + 60 pixel fragment reads
//     (0.33 per MFMA of half the size, i.e. a third of the LDS read bytes per FLOP of the tall tile).
//   * Fragment addresses are lane-constant bases + immediate offsets (no address arithmetic in the loop): LDS rows are
//     dense 64-byte rows (32 channels), XOR-swizzled by bit 2 of the column / channel-row index, which makes every
//     ds_read_b128 lane group conflict-free for all three column shifts (exhaustive check in DESIGN.md section 3).
//   * Weights go global -> LDS by LDS-DMA (global_load_lds_dwordx4, per-lane source address): no staging registers, no
//     ds_write transfer cycles; the activation tile still passes through registers (BatchNorm + ReLU on the way).
//   * D comes out as (16 channels) x (16 pixels) with lane = pixel: choosing subtile s = channels {16j + 4s + i} makes
//     the four subtiles of a lane 16 CONSECUTIVE channels of its pixel -- the epilogue is convert + two 16-byte stores
//     per output row, no transposes.
// Shapes (conv3x3_rs_eligible): Cin % 32 == 0, N % 64 == 0, H % 32 == 0, W % 16 == 0, BN-activated source <= 512 channels,
// a second source / destination on a 32 / 64 channel boundary.  Everything else stays on k_conv3x3_bf16_fast.
#include "fu_conv_bf16.h"

namespace fu {

#if FU_HALF
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define k_conv3x3_bf16_rs k_conv3x3_f16_rs
#else
#define FU_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ROWS_ = output rows per wave: 8 (16 x 32-pixel workgroup tile) or 4 (16 x 16: the 32x32 and 16x16 levels, where the tall
// tile would leave CUs without work)
template <int ROWS_>
struct RCfg {
  static constexpr int NT = 256, TW = 16, ROWS = ROWS_, TH = 4 * ROWS, BN = 64, KC = 32;
  static constexpr int HWd = TW + 2, HHt = TH + 2, NHP = HHt * HWd;        // 18 x 34 = 612 halo pixels
  static constexpr int ROWB = 64;                                          // bytes per LDS row (32 channels)
  static constexpr int A_BYTES = NHP * ROWB, W_BYTES = 9 * BN * ROWB;      // 39168 + 36864
  static constexpr int A_UNITS = NHP * 4, A_ITERS = (A_UNITS + NT - 1) / NT, A_FULL = A_UNITS / NT;   // 2448: 9 full + 144
  static constexpr int AB_FLOATS = 2 * 512 + 64;                           // BN scale / shift of source 0, this tile's bias
  static constexpr int SMEM_BYTES = A_BYTES + W_BYTES + AB_FLOATS * 4;     // 80384: two workgroups per CU (160 KiB)
  static constexpr int IN_ROWS = ROWS + 2;
};

template <int ROWS_>
__global__ __launch_bounds__(256, 2) void k_conv3x3_bf16_rs(BConvP P) {
  using Cfg = RCfg<ROWS_>;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, BN = Cfg::BN, KC = Cfg::KC, NT = Cfg::NT, HWd = Cfg::HWd, NHP = Cfg::NHP;
  constexpr int A_ITERS = Cfg::A_ITERS, ROWB = Cfg::ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sA = smem_raw;                                    // [612 halo pixels][64 B], slot ^= 2 where (hx & 4)
  unsigned char* sW = smem_raw + Cfg::A_BYTES;                     // [9 taps][64 rows][64 B], slot ^= 2 where (row & 4)
  float* sAB = reinterpret_cast<float*>(sW + Cfg::W_BYTES);        // [2][512] BN scale / shift of source 0

  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave = 8 output rows of the tile
  const int lx = lane & 15, lg = lane >> 4;                        // pixel column / channel row m, and k-group (8 channels)

#ifndef FU_RS_PIXEL_MAJOR
#define FU_RS_PIXEL_MAJOR 1
#endif
  // Workgroup order inside an XCD's contiguous share of the grid (xcd_remap): the CHANNEL tile runs fastest, so the nCo
  // workgroups that read the same activation tile sit on one XCD next to each other in time and the tile is fetched into
  // that L2 once.  (Channel-tile-major order, as in the round-1 kernels, makes every XCD stream the whole input: with
  // stamps the "A convert" phase measured 6000-7000 cycles per chunk waiting for its loads -- 152 KB per CU every 6 us =
  // 6.3 TB/s chip-wide through the fabric, the kernel was memory-system bound on L2 misses, not on its instructions.)
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
#if FU_RS_PIXEL_MAJOR
  const int pixT = fast_div(logical, P.nCo, P.rcp_nCo);
  const int coT = logical - pixT * P.nCo;
#else
  const int coT = fast_div(logical, P.nPix, P.rcp_nPix);
  const int pixT = logical - coT * P.nPix;
#endif
  const int t2 = fast_div(pixT, P.tilesX, P.rcp_tilesX);
  const int tx = pixT - t2 * P.tilesX;
  const int bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
  const int ty = t2 - bb * P.tilesY;
  const int x0 = tx * TW, y0 = ty * TH, n0 = coT * BN;
  const bool has_bn = P.a0 != nullptr;
  const bool border = !(y0 >= 1 && y0 + TH + 1 <= P.H && x0 >= 1 && x0 + TW + 1 <= P.W);

  // ---- activation staging slots: unit u = tid + 256 it = halo pixel (tid >> 2) + 64 it, channel octet tid & 3
  const int aq = tid & 3, srow = tid >> 2;
  unsigned a_pix[A_ITERS];                                         // clamped pixel index of the slot (< 2^24: eligibility)
  unsigned a_ok = 0, a_swz = 0;                                    // bit it: pixel inside the image / LDS slot swizzled
  {
    static_for<0, A_ITERS>([&](auto I) {
      constexpr int it = decltype(I)::value;
      const int hp = srow + it * 64;
      const int hy = (hp * 3641) >> 16;                            // hp / 18
      const int hx = hp - hy * HWd;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
      const bool ok = (it < Cfg::A_FULL || hp < NHP) && iy == cy && ix == cx;
      a_ok |= ok ? (1u << it) : 0u;
      a_swz |= (hx & 4) ? (1u << it) : 0u;
      a_pix[it] = (unsigned)((bb * P.H + cy) * P.W + cx);
    });
  }

  // ---- weight DMA: instruction `tap` of wave wm fills LDS rows [tap*64 + 16 wm, +16) = subtile s = wm of that tap.
  //      Lane i lands on row m = i >> 2, physical slot i & 3, i.e. k-group g = (i & 3) ^ ((m & 4) >> 1); row m of subtile s
  //      is output channel 16 (m >> 2) + 4 s + (m & 3) (the channel order that makes the epilogue transpose-free).
  const int dm = lane >> 2, dgk = (lane & 3) ^ ((dm & 4) >> 1);
  const int dn = 16 * (dm >> 2) + 4 * wm + (dm & 3);
  const unsigned w_src = (unsigned)((n0 + dn) * P.Cin + 8 * dgk) * 2u;          // + (tap * N * Cin + k0) * 2
  const unsigned w_tap = (unsigned)(P.N * P.Cin) * 2u;
  auto dma_weights = [&](int k0) {
    // uniform base (SGPR pair) + one 32-bit lane offset: nine per-tap 64-bit lane addresses would be hoisted out of the
    // chunk loop and cost 18 registers
    // (the lane offset is made opaque per call: as a loop invariant hipcc re-associates the sum into nine hoisted 64-bit
    //  lane addresses + k0 after all -- three of them were spilled, and each reload sat behind an s_waitcnt vmcnt(0) in the
    //  middle of the nine DMA issues, i.e. waited for the pieces already in flight.  Opaque: 256 registers and NO spill at
    //  ROWS 8 (11 before), 158 at ROWS 4; conv class 2.69 -> 2.62 ms per step, one stream, same box)
    const char* ub = reinterpret_cast<const char*>(P.wpk) + (size_t)k0 * 2;
    unsigned ws = w_src;
    asm volatile("" : "+v"(ws));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + (size_t)tap * w_tap + (size_t)ws),
                                       (__attribute__((address_space(3))) void*)(sW + tap * (BN * ROWB) + wm * 1024), 16, 0, 0);
  };

  uint4 ra[A_ITERS];
  const char* abL = nullptr;
  unsigned cs2 = 0;                                                // bytes per pixel of the current source (uniform)
  auto load_begin = [&](int k0) {
    const bool s1 = P.src1 != nullptr && k0 >= P.C0;               // uniform
    abL = s1 ? reinterpret_cast<const char*>(P.src1) + (size_t)(k0 - P.C0) * 2
             : reinterpret_cast<const char*>(P.src0) + (size_t)k0 * 2;
    cs2 = (unsigned)(s1 ? P.C1 : P.C0) * 2u;
  };
  auto load_slot = [&](auto Sc) {
    constexpr int sl = decltype(Sc)::value;
    // uniform base + 32-bit lane offset = pixel * (channels * 2) + 16 aq: one v_mad_u32_u24 per load (pixel < 2^24,
    // bytes per pixel < 2^24; the product < 2^31 by eligibility)
    if constexpr (sl < A_ITERS)
      ra[sl] = *reinterpret_cast<const uint4*>(abL + ((a_pix[sl] & 0xffffffu) * (cs2 & 0xffffffu) + 16u * aq));
  };

  // LDS address of slot it: row (srow + 64 it), byte (16 aq) ^ (32 if the pixel's column has bit 2 set): the two candidate
  // bases differ by +-32, the per-slot bit selects (one v_bfe + one v_mad instead of a table of ten addresses)
  const unsigned lds_a0 = (unsigned)(srow * ROWB + 16 * aq);
  const int lds_ad = (aq < 2) ? 32 : -32;
  auto store_chunk = [&](int k0, auto Mc, auto Bc) {
    constexpr bool MASKED = decltype(Mc)::value, BNR = decltype(Bc)::value;
    // (opaque copies: the per-slot bit extractions below are loop invariant, and LLVM would hoist all twenty of them out of
    //  the chunk loop into registers that then spill)
    unsigned swz = a_swz, okm = a_ok;
    asm volatile("" : "+v"(swz), "+v"(okm));
    const int cc = (BNR ? k0 : 0) + 8 * aq;                        // < 512: inside sAB
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
      if (it < Cfg::A_FULL || srow + it * 64 < NHP) {
        unsigned x = ra[it].x, y = ra[it].y, z = ra[it].z, w = ra[it].w;
        if constexpr (BNR) {
          x = bn_relu_pair(x, ca0, cb0); y = bn_relu_pair(y, ca1, cb1);
          z = bn_relu_pair(z, ca2, cb2); w = bn_relu_pair(w, ca3, cb3);
        }
        if constexpr (MASKED) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)okm, it, 1);   // bit it -> 0 / 0xffffffff
          x &= m; y &= m; z &= m; w &= m;
        }
        const unsigned addr = lds_a0 + (unsigned)((int)__builtin_amdgcn_ubfe(swz, it, 1) * lds_ad);
        *reinterpret_cast<uint4*>(sA + addr + it * (64 * ROWB)) = make_uint4(x, y, z, w);
      }
    });
  };

  f32x4 acc[Cfg::ROWS][4];
#pragma unroll
  for (int r = 0; r < Cfg::ROWS; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[r][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment bases (bytes): weights row m = lx of a subtile; pixels column lx + dx of a halo row
  const unsigned char* wfb = sW + lx * ROWB + ((16 * lg) ^ ((lx & 4) << 3));
  const unsigned char* pfb[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
    pfb[dx] = sA + (wm * Cfg::ROWS * HWd + lx + dx) * ROWB + ((16 * lg) ^ (((lx + dx) & 4) << 3));

  const int nChunks = P.Cin / KC;
#ifdef FU_CONV_STAMPS     // diagnostic builds only (tools/stamp_rs.py): s_memtime sums per phase, wave 0
  const unsigned long long R0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz: cycles / ticks = the clock the chip holds
  unsigned long long T0 = __builtin_amdgcn_s_memtime(), T1 = 0, Sbar = 0, Sstore = 0, Swait = 0, Smfma = 0, Sload = 0, Sdma = 0;
#endif
  // Coefficient tables first: their loads are waited for (vmcnt counts in order) before the LDS writes below, and issued
  // behind the activation loads and the weight DMA that wait would cover those too -- the first stage could then not start
  // converting while the DMA is still in flight.
  if (tid < BN) sAB[1024 + tid] = P.bias != nullptr ? P.bias[n0 + tid] : 0.f;     // (16 registers per lane if kept live)
  if (has_bn) {
    for (int c = tid; c < P.C0; c += NT) { sAB[c] = P.a0[c]; sAB[512 + c] = P.b0[c]; }
  }
  const bool bnb = P.bnb_y != nullptr;             // uniform; the launcher guarantees !has_bn, one destination
  if (bnb && tid < BN) {                           // this tile's 64 channels: a, b, invstd, -mean * invstd
    const int c = n0 + tid;
    const float iv = P.bnb_invstd[c];
    sAB[tid] = P.bnb_a[c]; sAB[64 + tid] = P.bnb_b[c]; sAB[128 + tid] = iv; sAB[192 + tid] = -P.bnb_mean[c] * iv;
  }
  load_begin(0);
  static_for<0, A_ITERS>([&](auto Sc) { load_slot(Sc); });
  dma_weights(0);                                   // LDS is free at kernel start

  // One chunk: for each column shift dx and subtile pair sp, the six weight fragments (3 kernel rows x 2 subtiles) stay
  // in registers while the ten input rows stream past; input row ri feeds output rows ri - dy.
  // (Measured and dropped: the six blocks as ONE software pipeline of 6 x IN_ROWS steps -- pixel ring running on across the
  //  block boundary, the next block's weight fragments read into the registers of the kernel rows that have just died -- so
  //  that no block starts with a burst of ten fragment reads behind an lgkmcnt(0).  Same registers, no spill, bit-identical:
  //  conv class 2624 -> 2629-2636 us per step.  The bubbles are covered by the co-resident workgroup.)
  auto mfma_block = [&](auto Lc) {
    constexpr bool LOADS = decltype(Lc)::value;     // issue the next chunk's activation loads behind the MFMAs
    static_for<0, 6>([&](auto Dc) {
      constexpr int dx = decltype(Dc)::value / 2, sp = decltype(Dc)::value % 2;
      __builtin_amdgcn_sched_barrier(0);            // this block's weight fragments are not hoisted into the previous one
      frag8_t wf[3][2];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          wf[dy][q] = *reinterpret_cast<const frag8_t*>(wfb + (dy * 3 + dx) * (BN * ROWB) + (2 * sp + q) * 1024);
      // pixel fragments: a ring of PD + 1 registers, PD rows ahead of the MFMAs (an LDS read under load returns after
      // 200-300 cycles; one row is only 6 MFMAs = 96 cycles of cover)
      constexpr int PD = 3, NR = PD + 1;
      static_assert(PD < Cfg::IN_ROWS, "prefetch ring deeper than the halo");
      frag8_t pf[NR];
      static_for<0, PD>([&](auto Rc) {
        constexpr int r = decltype(Rc)::value;
        pf[r] = *reinterpret_cast<const frag8_t*>(pfb[dx] + r * HWd * ROWB);
      });
      static_for<0, Cfg::IN_ROWS>([&](auto Rc) {
        constexpr int ri = decltype(Rc)::value;
        if constexpr (ri + PD < Cfg::IN_ROWS) {
          pf[(ri + PD) % NR] = *reinterpret_cast<const frag8_t*>(pfb[dx] + (ri + PD) * HWd * ROWB);
          __builtin_amdgcn_sched_barrier(0);        // keep the prefetch ahead of this row's MFMAs
        }
        static_for<0, 3>([&](auto Yc) {
          constexpr int dy = decltype(Yc)::value, ro = ri - dy;
          if constexpr (ro >= 0 && ro < Cfg::ROWS) {
            acc[ro][2 * sp] = FU_MFMA16(wf[dy][0], pf[ri % NR], acc[ro][2 * sp]);
            acc[ro][2 * sp + 1] = FU_MFMA16(wf[dy][1], pf[ri % NR], acc[ro][2 * sp + 1]);
          }
        });
        if constexpr (LOADS) {
          // 6 * IN_ROWS steps (60 / 36); one load every LS-th step covers the A_ITERS slots (10 / 6)
#ifndef FU_RS_LOAD_STRIDE
#define FU_RS_LOAD_STRIDE 0       // 0: spread over the whole block; n: one load every n-th step from the block's start
#endif
          constexpr int step = (dx * 2 + sp) * Cfg::IN_ROWS + ri,
                        LS = FU_RS_LOAD_STRIDE > 0 ? FU_RS_LOAD_STRIDE : (6 * Cfg::IN_ROWS) / (A_ITERS + 1);
          if constexpr (step % LS == 0 && step / LS < A_ITERS) load_slot(std::integral_constant<int, step / LS>{});
        }
      });
    });
  };
  auto stage = [&](int k0, bool first) {
#ifdef FU_CONV_STAMPS
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                // every wave is done reading the previous chunk's fragments
#ifdef FU_CONV_STAMPS
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef FU_RS_STAGE_PRIO
#define FU_RS_STAGE_PRIO 2
#endif
    // The co-resident workgroup's wave on this SIMD is (mostly) in its MFMA block: at equal priority its stream wins the
    // vector-issue arbitration by age and this conversion crawls (measured: 6000-7000 cycles for ~250 instructions, more
    // than the 5360-cycle MFMA block it should hide under).  An MFMA needs one issue slot per 16 cycles: raising the
    // STAGING wave lets its VALU take the slots in between.
    __builtin_amdgcn_s_setprio(FU_RS_STAGE_PRIO);
    if (!first) dma_weights(k0);    // the weights land while this wave converts its activation units
#ifdef FU_CONV_STAMPS
    const unsigned long long s1a = __builtin_amdgcn_s_memtime();
    if (!first) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // the activation loads (older than the nine DMA pieces)
    const unsigned long long s1b = __builtin_amdgcn_s_memtime();
    Sload += s1b - s1a;
    Sdma += s1a - s1;
#endif
    const bool bn = has_bn && k0 < P.C0;             // uniform
    if (border) { if (bn) store_chunk(k0, std::true_type{}, std::true_type{}); else store_chunk(k0, std::true_type{}, std::false_type{}); }
    else { if (bn) store_chunk(k0, std::false_type{}, std::true_type{}); else store_chunk(k0, std::false_type{}, std::false_type{}); }
#ifdef FU_CONV_STAMPS
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed (nothing else orders them)
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
#ifdef FU_CONV_STAMPS
    const unsigned long long s3 = __builtin_amdgcn_s_memtime();
    Sbar += s1 - s0; Sstore += s2 - s1; Swait += s3 - s2;
#endif
  };
  for (int ch = 0; ch + 1 < nChunks; ++ch) {
    const int k0 = ch * KC;
    stage(k0, ch == 0);
    load_begin(k0 + KC);
#ifdef FU_CONV_STAMPS
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    if (ch == 0) T1 = m0;
#endif
    mfma_block(std::true_type{});
#ifdef FU_CONV_STAMPS
    Smfma += __builtin_amdgcn_s_memtime() - m0;
#endif
  }
  stage((nChunks - 1) * KC, nChunks == 1);
#ifdef FU_CONV_STAMPS
  const unsigned long long m1 = __builtin_amdgcn_s_memtime();
  if (nChunks == 1) T1 = m1;
#endif
  // BatchNorm-backward sums (BnbFuse): the y rows of the tile.  The first NPRE of them are requested here -- the last
  // block issues no activation loads, so they ride in the registers those would occupy and have landed when the block
  // ends -- the rest at the start of the epilogue.
  constexpr int NPRE = A_ITERS / 2;
  uint4 yr[Cfg::ROWS][2];
  const char* ybase = nullptr;
  if (bnb) {
    ybase = reinterpret_cast<const char*>(P.bnb_y) +
            ((size_t)((bb * P.H + y0 + wm * Cfg::ROWS) * P.W + x0 + lx) * (size_t)P.N + (size_t)(n0 + 16 * lg)) * 2;
#pragma unroll
    for (int ro = 0; ro < NPRE; ++ro) {
      const char* yp = ybase + (size_t)ro * (size_t)(P.W * P.N) * 2;
      yr[ro][0] = *reinterpret_cast<const uint4*>(yp);
      yr[ro][1] = *reinterpret_cast<const uint4*>(yp + 16);
    }
  }
  mfma_block(std::false_type{});
#ifdef FU_CONV_STAMPS
  const unsigned long long T2 = __builtin_amdgcn_s_memtime();
  Smfma += T2 - m1;
#endif

  // ---- epilogue: lane (pixel lx, group lg) holds channels n0 + 16 lg + [0, 16) of its pixel in every output row
  const bool to0 = n0 < P.D0;                                   // uniform: D0 % 64 == 0 with two destinations
  char* dbase = reinterpret_cast<char*>(to0 ? P.dst0 + n0 : P.dst1 + (n0 - P.D0));
  const int dstride = to0 ? P.D0 : P.D1;
  // per-channel sums over the tile: 16 lanes (pixels) of a row group, then the 4 waves through LDS; fixed order
  // DPP only (a __shfl_xor is a ds_bpermute: 128 of them were ~4000 cycles of a 2-chunk tile): row_shr 1, 2, 4, 8
  // inside the 16-lane rows; lane 15 of every row ends with the row's total
  auto tile_sums_out = [&](float (&u)[16], float (&q)[16], float* out) __attribute__((always_inline)) {
    auto row_sum = [](float v) {
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
      return v;
    };
#pragma unroll
    for (int c = 0; c < 16; ++c) { u[c] = row_sum(u[c]); q[c] = row_sum(q[c]); }
    float* red = reinterpret_cast<float*>(smem_raw);            // [4 waves][64 channels][2]
    __syncthreads();
    if (lx == 15) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        red[(wm * BN + 16 * lg + c) * 2 + 0] = u[c];
        red[(wm * BN + 16 * lg + c) * 2 + 1] = q[c];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, t = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { s += red[(m * BN + tid) * 2 + 0]; t += red[(m * BN + tid) * 2 + 1]; }
      float* o = out + ((int64_t)pixT * P.N + n0 + tid) * 2;
      o[0] = s;
      o[1] = t;
    }
  };
  if (bnb) {
    // BatchNorm-backward sums of the destination (BnbFuse, fu_common.h): g = the accumulators (fp32, before their rounding
    // to the element type), y = the BatchNorm's raw input at the same pixels and channels.  Channel-group major: the
    // coefficients of four channels stay in registers while the rows stream past, all y rows are resident (2 x 16 bytes
    // per row and lane) -- row major would keep 32 coefficient registers live beside the 128 accumulators.  Six VALU
    // instructions per element: the second sum is accumulated against the raw y and turned into sum g*m*xhat =
    // invstd * (sum g*m*y - mean * sum g*m) once per tile and channel (fp32; the 16-bit modes only).
#pragma unroll
    for (int ro = NPRE; ro < Cfg::ROWS; ++ro) {
      const char* yp = ybase + (size_t)ro * (size_t)(P.W * P.N) * 2;
      yr[ro][0] = *reinterpret_cast<const uint4*>(yp);
      yr[ro][1] = *reinterpret_cast<const uint4*>(yp + 16);
    }
    auto row_sum = [](float v) {
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
      return v;
    };
    float* red = reinterpret_cast<float*>(smem_raw);            // [4 waves][64 channels][2]: the activation tile is done
    __syncthreads();                                            // ... once every wave has left the last block
    static_for<0, 4>([&](auto Sc) {
      constexpr int s = decltype(Sc)::value;
      const float4 ca = *reinterpret_cast<const float4*>(sAB + 16 * lg + 4 * s);
      const float4 cb = *reinterpret_cast<const float4*>(sAB + 64 + 16 * lg + 4 * s);
      const float av[4] = {ca.x, ca.y, ca.z, ca.w}, bv[4] = {cb.x, cb.y, cb.z, cb.w};
      float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ro = 0; ro < Cfg::ROWS; ++ro) {
        const uint4 w4 = yr[ro][s >> 1];
        const unsigned w01 = (s & 1) ? w4.z : w4.x, w23 = (s & 1) ? w4.w : w4.y;    // channels 4s, 4s+1 | 4s+2, 4s+3
        const float yv[4] = {e2f_lo(w01), e2f_hi(w01), e2f_lo(w23), e2f_hi(w23)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gm = fmaf(av[k], yv[k], bv[k]) > 0.f ? acc[ro][s][k] : 0.f;
          t1[k] += gm;
          t2[k] = fmaf(gm, yv[k], t2[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float r1 = row_sum(t1[k]), r2 = row_sum(t2[k]);
        if (lx == 15) {
          red[(wm * BN + 16 * lg + 4 * s + k) * 2 + 0] = r1;
          red[(wm * BN + 16 * lg + 4 * s + k) * 2 + 1] = r2;
        }
      }
    });
    __syncthreads();
    if (tid < BN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { s1 += red[(m * BN + tid) * 2 + 0]; s2 += red[(m * BN + tid) * 2 + 1]; }
      float* o = P.bnb_part + ((int64_t)pixT * P.N + n0 + tid) * 2;
      o[0] = s1;
      o[1] = fmaf(sAB[128 + tid], s2, sAB[192 + tid] * s1);      // invstd * s2 - mean * invstd * s1
    }
  }
  float ssum[16], ssq[16], biasv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; biasv[c] = sAB[1024 + 16 * lg + c]; }
#pragma unroll
  for (int ro = 0; ro < Cfg::ROWS; ++ro) {
    const int oy = y0 + wm * Cfg::ROWS + ro, ox = x0 + lx;
    unsigned o[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v0 = acc[ro][s][0], v1 = acc[ro][s][1], v2 = acc[ro][s][2], v3 = acc[ro][s][3];
      ssum[4 * s + 0] += v0; ssum[4 * s + 1] += v1; ssum[4 * s + 2] += v2; ssum[4 * s + 3] += v3;
      ssq[4 * s + 0] = fmaf(v0, v0, ssq[4 * s + 0]); ssq[4 * s + 1] = fmaf(v1, v1, ssq[4 * s + 1]);
      ssq[4 * s + 2] = fmaf(v2, v2, ssq[4 * s + 2]); ssq[4 * s + 3] = fmaf(v3, v3, ssq[4 * s + 3]);
      const f32x2 p01 = {v0 + biasv[4 * s + 0], v1 + biasv[4 * s + 1]};
      const f32x2 p23 = {v2 + biasv[4 * s + 2], v3 + biasv[4 * s + 3]};
      o[2 * s + 0] = pack_e2(p01);
      o[2 * s + 1] = pack_e2(p23);
    }
    char* dp = dbase + ((size_t)((bb * P.H + oy) * P.W + ox) * (size_t)dstride + (size_t)(16 * lg)) * 2;
    *reinterpret_cast<uint4*>(dp) = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(dp + 16) = make_uint4(o[4], o[5], o[6], o[7]);
  }
  if (P.stats) tile_sums_out(ssum, ssq, P.stats);
#ifdef FU_CONV_STAMPS
  if (P.dbg && tid == 0) {
    const unsigned long long T2c = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long T3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = P.dbg + (size_t)blockIdx.x * 10;
    d[0] = T0; d[1] = T1; d[2] = T2; d[3] = T3; d[4] = Sbar; d[5] = Sstore; d[6] = Swait; d[7] = Smfma; d[8] = T2c; d[9] = Sload + (Sdma << 32);
    P.dbg[(size_t)gridDim.x * 10 + blockIdx.x] = __builtin_amdgcn_s_memrealtime() - R0;
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// 8-input-channel convolution (the network's first conv: 8 bands, unet.py:86 `inc = DoubleConv(n_channels, 64)`), forward.
//
// With 8 channels a pixel is ONE 16-byte vector and the whole 3x3 window is K = 72: the row-stationary idea collapses to a
// kernel without LDS staging.  MFMA 16x16x32 again with the weights as A and a row of 16 pixels as B; the four k-groups of a
// k-step are the three column shifts dx (+ one zero group), the three k-steps are the kernel rows dy:
//     B fragment of lane (pixel x, group dx), step dy  =  the 8 channels of input pixel (row + dy - 1, x + dx - 1)
// i.e. one 16-byte global load per lane and INPUT row, kept in a register ring while the output rows walk down: output row
// r multiplies the fragments of input rows r-1, r, r+1 (steps dy = 0, 1, 2) -- a lane never needs another lane's data.  12
// MFMAs per 16-pixel row and 64 channels (3 steps x 4 subtiles; a quarter of each step multiplies zeros) against 36 on the
// 16-channel-chunk kernel this replaces; the layer is bound by its 134 MB of output (B = 16, 256x256), not by the MFMAs.
// Output / statistics exactly as the row-stationary kernel (lane = pixel with 16 consecutive channels, DPP row sums).
// Shapes: Cin == 8 (one source, no BatchNorm prologue), N % 64 == 0, one destination, H % 16 == 0, W % 16 == 0.
// ------------------------------------------------------------------------------------------------------------------------
#if FU_HALF
#define k_conv3x3_bf16_c8 k_conv3x3_f16_c8
#endif

template <int RW>     // output rows per wave; the workgroup tile is 16 columns x 4 RW rows x 64 channels
__global__ __launch_bounds__(256) void k_conv3x3_bf16_c8(BConvP P) {
  __shared__ float red[4 * 64 * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 15, lg = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int pixT = fast_div(logical, P.nCo, P.rcp_nCo);
  const int coT = logical - pixT * P.nCo;
  const int t2 = fast_div(pixT, P.tilesX, P.rcp_tilesX);
  const int tx = pixT - t2 * P.tilesX;
  const int bb = fast_div(t2, P.tilesY, P.rcp_tilesY);
  const int ty = t2 - bb * P.tilesY;
  const int x0 = tx * 16, y0 = ty * (4 * RW) + wm * RW, n0 = coT * 64;

  // weight fragments: row m = lx of subtile s is output channel 16 (m >> 2) + 4 s + (m & 3); k-group lg = column shift dx
  frag8_t wA[3][4];
  {
    const frag8_t zero = {};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const int co = n0 + 16 * (lx >> 2) + 4 * sb + (lx & 3);
        const int tap = 3 * dy + lg;
        wA[dy][sb] = lg < 3 ? *reinterpret_cast<const frag8_t*>(P.wpk + ((size_t)tap * P.N + co) * 8) : zero;
      }
  }
  float biasv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) biasv[c] = P.bias != nullptr ? P.bias[n0 + 16 * lg + c] : 0.f;

  // this lane's input column and its fragment loader (zero outside the image and for the padding k-group)
  const int ix = x0 + lx + lg - 1;
  const bool colok = lg < 3 && ix >= 0 && ix < P.W;
  const bf16_t* src = P.src0 + ((size_t)bb * P.H * P.W + (colok ? ix : 0)) * 8;
  auto load_row = [&](int iy) {
    const frag8_t zero = {};
    const bool ok = colok && iy >= 0 && iy < P.H;
    return ok ? *reinterpret_cast<const frag8_t*>(src + (size_t)iy * P.W * 8) : zero;
  };

  float ssum[16], ssq[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { ssum[c] = 0.f; ssq[c] = 0.f; }
  char* dbase = reinterpret_cast<char*>(P.dst0 + n0) + (size_t)(16 * lg) * 2;

  // ring of four input-row fragments: rows oy-1, oy, oy+1 are live, oy+2 is in flight.  The row loop is unrolled by the
  // ring size only (every ring index a constant); unrolled completely, hipcc hoists all the loads and the kernel needs
  // 156 registers (two waves per SIMD) -- this store-bound kernel wants waves, not a deep private prefetch
  static_assert(RW % 4 == 0, "rows per wave: a multiple of the ring size");
  frag8_t pf[4];
  pf[0] = load_row(y0 - 1); pf[1] = load_row(y0); pf[2] = load_row(y0 + 1);
#pragma unroll 1
  for (int rb = 0; rb < RW; rb += 4)
  static_for<0, 4>([&](auto Rc) {
    constexpr int k = decltype(Rc)::value;
    const int oy = y0 + rb + k;
    pf[(k + 3) % 4] = load_row(oy + 2);
    f32x4 acc[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) acc[sb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) acc[sb] = FU_MFMA16(wA[dy][sb], pf[(k + dy) % 4], acc[sb]);
    if (oy < P.H) {                                   // (uniform per wave; H % 16 == 0 keeps whole 16-row groups inside)
      unsigned o[8];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const float v0 = acc[sb][0], v1 = acc[sb][1], v2 = acc[sb][2], v3 = acc[sb][3];
        ssum[4 * sb + 0] += v0; ssum[4 * sb + 1] += v1; ssum[4 * sb + 2] += v2; ssum[4 * sb + 3] += v3;
        ssq[4 * sb + 0] = fmaf(v0, v0, ssq[4 * sb + 0]); ssq[4 * sb + 1] = fmaf(v1, v1, ssq[4 * sb + 1]);
        ssq[4 * sb + 2] = fmaf(v2, v2, ssq[4 * sb + 2]); ssq[4 * sb + 3] = fmaf(v3, v3, ssq[4 * sb + 3]);
        o[2 * sb + 0] = pack_e2(f32x2{v0 + biasv[4 * sb + 0], v1 + biasv[4 * sb + 1]});
        o[2 * sb + 1] = pack_e2(f32x2{v2 + biasv[4 * sb + 2], v3 + biasv[4 * sb + 3]});
      }
      char* dp = dbase + ((size_t)((bb * P.H + oy) * P.W + x0 + lx) * (size_t)P.N) * 2;
      *reinterpret_cast<uint4*>(dp) = make_uint4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<uint4*>(dp + 16) = make_uint4(o[4], o[5], o[6], o[7]);
    }
  });
  if (P.stats) {
    auto row_sum = [](float v) {
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
      return v;
    };
#pragma unroll
    for (int c = 0; c < 16; ++c) { ssum[c] = row_sum(ssum[c]); ssq[c] = row_sum(ssq[c]); }
    if (lx == 15) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        red[(wm * 64 + 16 * lg + c) * 2 + 0] = ssum[c];
        red[(wm * 64 + 16 * lg + c) * 2 + 1] = ssq[c];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float sa = 0.f, sq = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { sa += red[(m * 64 + tid) * 2 + 0]; sq += red[(m * 64 + tid) * 2 + 1]; }
      float* op = P.stats + ((int64_t)pixT * P.N + n0 + tid) * 2;
      op[0] = sa;
      op[1] = sq;
    }
  }
}

bool conv3x3_c8_eligible(const BConvP& P) {
  const int64_t px = (int64_t)P.B * P.H * P.W;
  if (P.center_only || P.Cin != 8 || P.src1 || P.a0 != nullptr || P.dst1 || (P.N % 64) || (P.H % 16) || (P.W % 16)) return false;
  if (px * P.N * 2 >= ((int64_t)1 << 31) || px >= ((int64_t)1 << 24)) return false;
  return true;
}

int launch_conv3x3_c8(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  const bool tall = (P.H % 64) == 0;                   // 16 x 64-pixel tiles (fewer statistics rows), else 16 x 16
  P.tilesX = P.W / 16; P.tilesY = P.H / (tall ? 64 : 16);
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = P.N / 64;
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  P.rcp_nCo = host_rcp(P.nCo);
  FU_REQUIRE((int64_t)P.nPix * P.nCo * P.nPix < ((int64_t)1 << 32) && (int64_t)P.nPix * P.nCo * P.nCo < ((int64_t)1 << 32),
             "conv3x3_c8: grid too large (%d x %d)", P.nPix, P.nCo);
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  if (tall) hipLaunchKernelGGL(k_conv3x3_bf16_c8<16>, dim3(P.nPix * P.nCo), dim3(256), 0, s, P);
  else hipLaunchKernelGGL(k_conv3x3_bf16_c8<4>, dim3(P.nPix * P.nCo), dim3(256), 0, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

bool conv3x3_rs_eligible(const BConvP& P) {
  const int64_t px = (int64_t)P.B * P.H * P.W;
  const int64_t lim = (int64_t)1 << 31;
  if (P.center_only) return false;
  if ((P.Cin % 32) || (P.N % 64) || (P.H % 16) || (P.W % 16)) return false;
  if (P.src1 && (P.C0 % 32) != 0) return false;
  if (P.dst1 && (P.D0 % 64) != 0) return false;
  if (P.a0 != nullptr && P.C0 > 512) return false;
  if (px * P.C0 * 2 >= lim || px * P.C1 * 2 >= lim || px * P.D0 * 2 >= lim || px * P.D1 * 2 >= lim) return false;
  if ((int64_t)9 * P.N * P.Cin * 2 >= lim || px >= ((int64_t)1 << 24)) return false;
  return true;
}

template <int ROWS_>
static int launch_rs_cfg(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  using Cfg = RCfg<ROWS_>;
  P.tilesX = P.W / Cfg::TW; P.tilesY = P.H / Cfg::TH;
  P.nPix = P.B * P.tilesX * P.tilesY; P.nCo = P.N / Cfg::BN;
  P.rcp_nPix = host_rcp(P.nPix); P.rcp_tilesX = host_rcp(P.tilesX); P.rcp_tilesY = host_rcp(P.tilesY);
  P.rcp_nCo = host_rcp(P.nCo);
  FU_REQUIRE((int64_t)P.nPix * P.nCo * P.nPix < ((int64_t)1 << 32) && (int64_t)P.nPix * P.nCo * P.nCo < ((int64_t)1 << 32),
             "conv3x3_rs: grid too large (%d x %d)", P.nPix, P.nCo);
  static bool attr_set = false;
  if (!attr_set) {
    FU_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_bf16_rs<ROWS_>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES));
    attr_set = true;
  }
  const ProfSlot ps = o.prof;
  if (ps.start) (void)hipEventRecord(ps.start, s);
  hipLaunchKernelGGL(k_conv3x3_bf16_rs<ROWS_>, dim3(P.nPix * P.nCo), dim3(Cfg::NT), Cfg::SMEM_BYTES, s, P);
  if (ps.stop) (void)hipEventRecord(ps.stop, s);
  FU_LAUNCH_CHECK();
  return 0;
}

// 16 x 32-pixel tiles where they still give every CU two workgroups, 16 x 16 otherwise
int launch_conv3x3_rs(BConvP& P, const LaunchOpts& o, hipStream_t s) {
  const int64_t t512 = (int64_t)P.B * (P.H / 32) * (P.W / 16) * (P.N / 64);
  const bool tall = (P.H % 32) == 0 && t512 >= 512;
  // BatchNorm-backward sums of the destination, if the API layer asked for them and this launch can give them
  static const BnbFuse none;
  const BnbFuse& f = o.bnb ? *o.bnb : none;
  if (f.y != nullptr && f.tiles_out != nullptr && P.a0 == nullptr && P.dst1 == nullptr && P.stats == nullptr) {
    const int64_t tiles = (int64_t)P.B * (P.H / (tall ? 32 : 16)) * (P.W / 16);
    if (tiles * P.N * 2 <= f.max_elems) {
      P.bnb_y = (const bf16_t*)f.y; P.bnb_a = f.a; P.bnb_b = f.b; P.bnb_mean = f.mean; P.bnb_invstd = f.invstd;
      P.bnb_part = f.part;
      *f.tiles_out = (int)tiles;
    }
  }
  if (tall) return launch_rs_cfg<8>(P, o, s);
  return launch_rs_cfg<4>(P, o, s);
}

}  // namespace fu
