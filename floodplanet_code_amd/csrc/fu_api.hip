// libfloodunet: context, static execution plan of the UNet training step and the C ABI (include/floodunet.h).
//
// Data layout in HBM (all owned by the context, one arena allocation):
//   activations  NHWC, element type = precision (fp32 or bf16); for every 3x3 conv only its RAW output y
//                (pre-BatchNorm) is kept -- BN+ReLU is re-applied by whoever reads y (next conv's staging,
//                pool, upsample, head, and the backward kernels), so normalised tensors never touch HBM.
//   pooled[l], up[k]   the only materialised post-activation tensors (pool / bilinear outputs).
//   gradients    one buffer per y (same shape/type): first holds dL/d relu(bn(y)), then, in place, dL/dy.
//   parameters   caller-owned flat fp32 buffers in reference state_dict order (OIHW); the context keeps
//                packed per-tap copies (forward and tap-reversed dgrad layouts) refreshed after each update.
#include "../../include/floodunet.h"
#include "fu_common.h"
#include <stdlib.h>

#include <math.h>
#include <string.h>

#include <string>
#include <vector>

using namespace fu;

namespace fu {

// ---- precision dispatch -------------------------------------------------------------------------
int conv3x3_num_stat_tiles(Prec p, int B, int H, int W) {
  return p == PREC_F32 ? conv3x3_num_stat_tiles_f32(B, H, W)
                       : (p == PREC_BF16 ? conv3x3_num_stat_tiles_bf16(B, H, W) : conv3x3_num_stat_tiles_f16(B, H, W));
}
int launch_conv3x3(Prec p, const ConvIn& in, const void* wpk, const float* bias, void* dst0, int D0, void* dst1,
                   int D1, float* stats, int* n_stat_tiles, int B, int H, int W, hipStream_t s) {
  if (p == PREC_F32)
    return launch_conv3x3_f32(in, (const float*)wpk, bias, (float*)dst0, D0, (float*)dst1, D1, stats, n_stat_tiles, B,
                              H, W, s);
  if (p == PREC_BF16)
    return launch_conv3x3_bf16(in, (const bf16_t*)wpk, bias, (bf16_t*)dst0, D0, (bf16_t*)dst1, D1, stats, n_stat_tiles,
                               B, H, W, s);
  return launch_conv3x3_f16(in, (const bf16_t*)wpk, bias, (bf16_t*)dst0, D0, (bf16_t*)dst1, D1, stats, n_stat_tiles, B,
                            H, W, s);
}
int64_t conv3x3_wgrad_slab_elems(Prec p, int Cin, int Cout, int B, int H, int W) {
  return p == PREC_F32 ? conv3x3_wgrad_slab_elems_f32(Cin, Cout, B, H, W)
                       : (p == PREC_BF16 ? conv3x3_wgrad_slab_elems_bf16(Cin, Cout, B, H, W)
                                         : conv3x3_wgrad_slab_elems_f16(Cin, Cout, B, H, W));
}
int launch_conv3x3_wgrad(Prec p, const ConvIn& in, const void* dy, int Cout, float* slab, float* dw_oihw,
                         int cin_real, const float* db_partials, int n_db_partials, float* db, int B, int H, int W,
                         hipStream_t s) {
  if (p == PREC_F32)
    return launch_conv3x3_wgrad_f32(in, (const float*)dy, Cout, slab, dw_oihw, cin_real, db_partials, n_db_partials,
                                    db, B, H, W, s);
  if (p == PREC_BF16)
    return launch_conv3x3_wgrad_bf16(in, (const bf16_t*)dy, Cout, slab, dw_oihw, cin_real, db_partials, n_db_partials,
                                     db, B, H, W, s);
  return launch_conv3x3_wgrad_f16(in, (const bf16_t*)dy, Cout, slab, dw_oihw, cin_real, db_partials, n_db_partials, db,
                                  B, H, W, s);
}
int64_t conv3x3_pack_elems(Prec p, int cin_pad, int Cout) {
  (void)p;
  return (int64_t)9 * cin_pad * Cout;
}
int launch_pack_conv3x3(Prec p, const float* w_oihw, int Cout, int cin_real, int cin_pad, void* wfwd, void* wdgrad,
                        hipStream_t s) {
  if (p == PREC_F32) return launch_pack_conv3x3_f32(w_oihw, Cout, cin_real, cin_pad, (float*)wfwd, (float*)wdgrad, s);
  if (p == PREC_BF16) return launch_pack_conv3x3_bf16(w_oihw, Cout, cin_real, cin_pad, (bf16_t*)wfwd, (bf16_t*)wdgrad, s);
  return launch_pack_conv3x3_f16(w_oihw, Cout, cin_real, cin_pad, (bf16_t*)wfwd, (bf16_t*)wdgrad, s);
}

}  // namespace fu

// ---- plan structures ------------------------------------------------------------------------------
namespace {

// one launch packs every conv layer: device table of layers, element ranges by prefix sum
struct PackDesc {
  int64_t start;      // first packed element of this layer in the global element numbering
  int64_t w_off;      // offset of the OIHW weight in the flat parameter buffer
  int cout, cin_real, cin_pad, pad_;
  void* wf;
  void* wd;
  int tile_start, tiles_ci;   // bf16 tiled pack: first 32x32 (co x ci) tile of this layer, tiles along ci
  const float* scale;         // eval pack: per-output-channel factor gamma * invstd of the BatchNorm behind the conv
};
constexpr int MAX_PACK = 32;
struct PackTable { PackDesc d[MAX_PACK]; int n; int64_t total; int tiles; };

// fp32 -> raw 16-bit storage of the context's element type
template <bool HALF> __device__ __forceinline__ unsigned short cvt16(float v) { return HALF ? f2h(v) : f2bf(v); }

template <typename T, bool BF16_LAYOUT, bool HALF = false>
__global__ void k_pack_all(const float* __restrict__ params, PackTable tab, int use_scale) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tab.total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int l = 0;
#pragma unroll 1
    for (int k = 1; k < tab.n; ++k) l = idx >= tab.d[k].start ? k : l;
    const PackDesc& D = tab.d[l];
    const int64_t e = idx - D.start;
    const float* w = params + D.w_off;
    int co, ci, tap;
    if (BF16_LAYOUT) {            // element order [tap][co][ci]
      ci = (int)(e % D.cin_pad);
      const int64_t r = e / D.cin_pad;
      co = (int)(r % D.cout);
      tap = (int)(r / D.cout);
    } else {                      // element order [tap][ci][co]
      co = (int)(e % D.cout);
      const int64_t r = e / D.cout;
      ci = (int)(r % D.cin_pad);
      tap = (int)(r / D.cin_pad);
    }
    float v = ci < D.cin_real ? w[((int64_t)co * D.cin_real + ci) * 9 + tap] : 0.f;
    if (use_scale) v *= D.scale[co];
    T* wf = (T*)D.wf;
    T* wd = (T*)D.wd;
    if (BF16_LAYOUT) {
      wf[e] = (T)cvt16<HALF>(v);
      if (wd) wd[((int64_t)(8 - tap) * D.cin_pad + ci) * D.cout + co] = (T)cvt16<HALF>(v);
    } else {
      ElemIO<T>::store1(wf + e, v);
      if (wd) ElemIO<T>::store1(wd + ((int64_t)(8 - tap) * D.cout + co) * D.cin_pad + ci, v);
    }
  }
}

// bf16 layouts, tiled: one workgroup converts a 32 (c_out) x 32 (c_in) x 9 block.  OIHW rows are read as contiguous
// 1152-byte runs, both packed layouts are written as 16-byte vectors along their fastest dimension (wf: c_in,
// wd: c_out); the element-wise kernel above reads with a 36-byte stride and writes 2-byte values 2*cout bytes apart
// (142 us per step for the 17M-parameter UNet, 8x its HBM time).  Needs cout % 8 == 0 and cin_pad % 8 == 0.
template <bool HALF>
__global__ __launch_bounds__(256) void k_pack_tiles_16(const float* __restrict__ params, PackTable tab, int use_scale) {
  constexpr int PITCH = 34;
  __shared__ unsigned short sT[9][32][PITCH];
  int l = 0;
  for (int k = 1; k < tab.n; ++k) l = (int)blockIdx.x >= tab.d[k].tile_start ? k : l;
  const PackDesc& D = tab.d[l];
  const int local = blockIdx.x - D.tile_start;
  const int tco = local / D.tiles_ci, tci = local - tco * D.tiles_ci;
  const int co0 = tco * 32, ci0 = tci * 32;
  const float* w = params + D.w_off;
  // A block's 32 OIHW rows are 32 runs of (up to) 288 contiguous floats.  Where they are 16-byte aligned, a thread fetches its 9
  // float4 pieces back to back (round 4: the scalar loop below issued 36 dependent 4-byte loads per thread, one memory latency
  // each -- 50 us per step at the head of every forward for 138 MB of traffic).
  const int run = min(32, max(D.cin_real - ci0, 0)) * 9;             // valid floats of a row of this tile
  if ((D.w_off & 3) == 0 && (D.cin_real & 3) == 0) {
    float4 v4[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int u = threadIdx.x + k * 256, co_l = u / 72, q = u - co_l * 72;
      const int co = co0 + co_l;
      v4[k] = (co < D.cout && 4 * q < run) ? *reinterpret_cast<const float4*>(w + ((size_t)co * D.cin_real + ci0) * 9 + 4 * q)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);      // (run % 4 == 0: whole pieces)
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int u = threadIdx.x + k * 256, co_l = u / 72, q = u - co_l * 72;
      const float sc = (use_scale && co0 + co_l < D.cout) ? D.scale[co0 + co_l] : 1.f;
      const float vv[4] = {v4[k].x, v4[k].y, v4[k].z, v4[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = 4 * q + j, ci_l = f / 9, tap = f - ci_l * 9;
        sT[tap][co_l][ci_l] = cvt16<HALF>(use_scale ? vv[j] * sc : vv[j]);
      }
    }
  } else {
    for (int e = threadIdx.x; e < 32 * 288; e += 256) {
      const int co_l = e / 288, r = e - co_l * 288;
      const int ci_l = r / 9, tap = r - ci_l * 9;
      const int co = co0 + co_l, ci = ci0 + ci_l;
      float v = (co < D.cout && ci < D.cin_real) ? w[((size_t)co * D.cin_real + ci) * 9 + tap] : 0.f;
      if (use_scale && co < D.cout) v *= D.scale[co];
      sT[tap][co_l][ci_l] = cvt16<HALF>(v);
    }
  }
  __syncthreads();
  bf16_t* wf = (bf16_t*)D.wf;
  bf16_t* wd = (bf16_t*)D.wd;
  for (int it = threadIdx.x; it < 9 * 32 * 4; it += 256) {
    const int oct = it & 3, row = (it >> 2) & 31, tap = it >> 7;
    {   // wf[tap][co][ci]: row = c_out, 8 consecutive c_in
      const int co = co0 + row, ci = ci0 + oct * 8;
      if (co < D.cout && ci < D.cin_pad) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&sT[tap][row][oct * 8]);
        *reinterpret_cast<uint4*>(wf + ((size_t)tap * D.cout + co) * D.cin_pad + ci) =
            make_uint4(src[0], src[1], src[2], src[3]);
      }
    }
    if (wd) {   // wd[8 - tap][ci][co]: row = c_in, 8 consecutive c_out
      const int ci = ci0 + row, co = co0 + oct * 8;
      if (ci < D.cin_pad && co < D.cout) {
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = (unsigned)sT[tap][oct * 8 + 2 * j][row] | ((unsigned)sT[tap][oct * 8 + 2 * j + 1][row] << 16);
        *reinterpret_cast<uint4*>(wd + ((size_t)(8 - tap) * D.cin_pad + ci) * D.cout + co) =
            make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOMENTUM = 0.1f;

// Eval mode (water_seg_model.py:92-96, 138-158: BatchNorm on its running statistics): bn(conv(x)) is affine per output
// channel, so it is folded into the conv once per parameter change -- packed weights times scale = gamma / sqrt(rv + eps),
// bias' = scale * bias + (beta - rm * scale) -- and every consumer's activation prologue becomes relu(1 * y + 0).
__global__ void k_bn_fold_eval(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, const float* __restrict__ bias,
                               float eps, float* __restrict__ scale, float* __restrict__ fbias, float* __restrict__ a,
                               float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  fbias[c] = fmaf(sc, bias[c], beta[c] - rm[c] * sc);
  a[c] = 1.f;
  b[c] = 0.f;
}

struct ParamInfo {
  std::string name;
  int ndim;
  int64_t shape[4];
  int64_t off, numel;
};
struct BnInfo {
  std::string name;
  int C;
  int64_t off;
};

struct Conv {
  int cin_real = 0, cin_pad = 0, cout = 0, level = 0;
  int p_w = -1, p_b = -1, p_g = -1, p_beta = -1, bn = -1;
  void* wf = nullptr;
  void* wd = nullptr;
  float *mean = nullptr, *invstd = nullptr, *a = nullptr, *b = nullptr, *coef = nullptr;
  float *fold_scale = nullptr, *fold_bias = nullptr;   // eval pack (k_bn_fold_eval)
  void* y = nullptr;
  void* gy = nullptr;
  const void* pool_g = nullptr;   // backward: dL/d(maxpool(this output)), to be folded into this conv's BN backward
  int bnb_tiles = 0;              // backward: > 0 = the producer of gy left this many rows of BN-backward sums in bnb_part
  HeadGrad head;                  // backward, last conv only: gy was not stored, the BN-backward apply recomputes it (dl != null)
};

enum BlockKind { BK_INC = 0, BK_DOWN = 1, BK_UP = 2 };

struct Block {
  Conv c[2];
  int kind = BK_INC, level = 0;
  int enc = 0;                // BK_INC / BK_DOWN: encoder this block belongs to
  int role = 0;               // 0..8 = inc, down1..4, up1..4 (names, flops)
  int skip = -1;              // BK_UP: level whose feature is concatenated first
  void* pooled = nullptr;     // BK_DOWN: maxpool output (input of c[0])
  void* g_pooled = nullptr;
  void* up = nullptr;         // BK_UP: upsampled + padded low-resolution input
  void* g_up = nullptr;
  UpTables upt;
  // bilinear=False: ConvTranspose2d(ct_cin, ct_cout, 2, 2) = one 1x1 conv ct_cin -> 4 ct_cout (phase-major) at the low
  // resolution + depth-to-space
  int ct_w = -1, ct_b = -1, ct_cin = 0, ct_cout = 0;
  void* u = nullptr;          // y4: the 1x1 conv's output [B, h, w, 4 ct_cout]
  void* g_u = nullptr;        // g4: its gradient (space-to-depth of dL/d up)
  float* ct_w3 = nullptr;     // embedded OIHW weight [4 ct_cout][ct_cin][3][3] (fp32, centre tap only)
  float* ct_dw3 = nullptr;    // its gradient
  float* ct_b4 = nullptr;     // bias repeated per phase [4 ct_cout]
  void* ct_wf = nullptr;      // packed forward / dgrad copies
  void* ct_wd = nullptr;
  int first_param = 0, num_params = 0;  // contiguous range in the canonical parameter table
};

// Late fusion, one per level (lf_model.py:40-45, 78-90): fused = Conv2d(nE*C, C, 1)(cat_e relu(bn(x_e)))
struct Fuse {
  int p_w = -1, p_b = -1, C = 0;
  void* cat = nullptr;        // [pixels][nE*C]: activated encoder features side by side
  void* gcat = nullptr;       // its gradient
  void* y = nullptr;          // fused feature (plain: no BN / ReLU follows)
  void* gy = nullptr;         // its gradient (written by the decoder's backward)
  float* w3 = nullptr;        // the 1x1 weight as the centre tap of a 3x3 one, OIHW fp32
  float* dw3 = nullptr;
  void* wf = nullptr;         // packed forward / dgrad copies
  void* wd = nullptr;
};

// what the decoder reads at one level: the encoder's own conv output (plain UNet) or the fused feature
struct Feat { void* y; float* a; float* b; void* gy; int C; };

struct ProfRec { int cls; double flops; hipEvent_t e0, e1; };
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> pool;   // pairs
  size_t next = 0;
  std::vector<ProfRec> recs;
  bool overflow = false;
};

struct Arena {
  struct Req { void** slot; size_t bytes; };
  std::vector<Req> reqs;
  char* base = nullptr;
  size_t total = 0;
  template <typename T> void want(T** slot, size_t bytes) {
    reqs.push_back({reinterpret_cast<void**>(slot), bytes});
  }
  int commit() {
    size_t off = 0;
    for (auto& r : reqs) off += (r.bytes + 255) & ~(size_t)255;
    total = off ? off : 256;
    FU_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base), total));
    FU_HIP_CHECK(hipMemset(base, 0, total));
    off = 0;
    for (auto& r : reqs) {
      *r.slot = base + off;
      off += (r.bytes + 255) & ~(size_t)255;
    }
    return 0;
  }
};

}  // namespace

struct Dp;
struct fu_ctx {
  fu_config cfg;
  Prec prec;
  size_t esize;
  int Hs[5], Ws[5], ch[5];
  int nE = 1;                  // encoders (1 for the plain UNet)
  bool fusion = false;         // late fusion: nE encoders -> 5 fusion convs -> decoder
  int nb = 9;                  // blocks: 5 per encoder (inc, down1..4), then up1..4
  int enc_ch[FU_MAX_ENCODERS] = {0}, enc_coff[FU_MAX_ENCODERS] = {0}, cin_pad0[FU_MAX_ENCODERS] = {0};
  void* xin[FU_MAX_ENCODERS] = {nullptr};
  std::vector<ParamInfo> params;
  std::vector<BnInfo> bns;
  int64_t total_params = 0, total_bn = 0;
  std::vector<Block> blk;
  Fuse fuse[5];
  int p_outw = -1, p_outb = -1;
  // bound (caller-owned)
  float* P = nullptr;
  float* G = nullptr;
  float* RM = nullptr;
  float* RV = nullptr;
  int64_t* NBT = nullptr;
  bool packed_dirty = true;
  bool packed_eval = false;       // the packed copies hold the eval-folded weights (BatchNorm inside) rather than the plain ones
  // owned
  Arena arena;
  std::vector<void*> extra_allocs;
  float* logits = nullptr;
  float* dlogits = nullptr;        // dL/dlogits as fu_loss_* (or the caller) left it: never modified by a backward
  float* dlogits_eff = nullptr;    // times the upstream gradient / the fp16 loss scale (launch_loss_grad_eff)
  float* up_scale = nullptr;       // device scalar: upstream gradient of the loss (fu_scale_loss_grad)
  bool have_up_scale = false;
  float* stats = nullptr;
  float* bnb_part = nullptr;
  int64_t bnb_cap = 0;         // floats
  float* db_part = nullptr;
  float* db_part2 = nullptr;      // second bias-gradient partial buffer (side-stream wgrad, alternating per conv)
  hipStream_t side = nullptr;     // side stream for the weight-gradient chain (wgrad + slab reduce + transpose): one of ...
  hipStream_t side_lo = nullptr;  // ... lowest priority (mode 1: nothing but the final join waits for that chain; the main chain
                                  //     conv -> BN backward -> conv is the critical path and gets the CUs first: measured
                                  //     5.66 -> 5.64 ms per step and 0.338 -> 0.350 of peak for the conv launches in the step)
  hipStream_t side_def = nullptr; // ... the default priority (mode 2: an all-reduce bucket waits for its weight gradients)
  hipEvent_t ev_gy = nullptr, ev_wg[2] = {nullptr, nullptr}, ev_blk = nullptr;
  hipEvent_t ev_fence[2] = {nullptr, nullptr};   // fu_backward_fence: compute stream / side stream (created on first use)
  int wg_parity = 0;
  int side_mode = 1;              // fu_set_side_stream: 0 off, 1 on (blocks join), 2 on (the caller joins: fu_backward_join)
  bool wg_pending[2] = {false, false};
  double* dscratch = nullptr;
  fu::SyncDesc sync;           // exact data-parallel mode (fu_set_exact_sync); hook == nullptr: off
  float* slab = nullptr;
  float* ce_part = nullptr;
  float* hb_part = nullptr;
  float* loss_dev = nullptr;
  float* loss_scale = nullptr;    // fp16 mode: {S, 1/S} of the running backward (fu_common.h, launch_loss_grad_eff)
  int* guard = nullptr;           // fp16 mode: non-finite flag / skipped steps / back-off exponent / clean steps (k_guard_book)
  unsigned long long* conf_tmp = nullptr;
  int64_t* n_valid = nullptr;
  float* adam_m = nullptr;        // bound (caller-owned, fu_bind_adam_state): the moments outlive the context
  float* adam_v = nullptr;
  Profiler prof;
  struct Dp* dp = nullptr;            // fu_dp_init: RCCL communicator, communication stream, events
  std::vector<PackTable> pack_tabs;   // <= MAX_PACK layers per launch
  // state
  int last_batch = 0;
  bool fwd_training = false;
  bool have_loss = false;
};

namespace {

std::string dc_prefix(int i) {   // i = block role
  if (i == 0) return "inc.double_conv";
  if (i <= 4) return "down" + std::to_string(i) + ".maxpool_conv.1.double_conv";
  return "up" + std::to_string(i - 4) + ".conv.double_conv";
}

int add_param(fu_ctx* c, const std::string& name, std::initializer_list<int64_t> shape) {
  ParamInfo p;
  p.name = name;
  p.ndim = (int)shape.size();
  p.numel = 1;
  int k = 0;
  for (auto d : shape) { p.shape[k++] = d; p.numel *= d; }
  for (; k < 4; ++k) p.shape[k] = 1;
  p.off = c->total_params;
  c->total_params += p.numel;
  c->params.push_back(p);
  return (int)c->params.size() - 1;
}

void build_axis(int in, std::vector<int>& i0, std::vector<int>& i1, std::vector<float>& w1, std::vector<int>& bo,
                std::vector<float>& bw, bool* ok) {
  const int out = 2 * in;
  // ATen area_pixel_compute_scale<float>(align_corners=True) and compute_source_index_and_lambda
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  i0.resize(out); i1.resize(out); w1.resize(out);
  bo.assign((size_t)in * UP_BWD_MAX, -1);
  bw.assign((size_t)in * UP_BWD_MAX, 0.f);
  std::vector<int> cnt(in, 0);
  auto push = [&](int i, int o, float w) {
    if (w == 0.f) return;
    for (int j = 0; j < cnt[i]; ++j)
      if (bo[(size_t)i * UP_BWD_MAX + j] == o) { bw[(size_t)i * UP_BWD_MAX + j] += w; return; }
    if (cnt[i] >= UP_BWD_MAX) { *ok = false; return; }
    bo[(size_t)i * UP_BWD_MAX + cnt[i]] = o;
    bw[(size_t)i * UP_BWD_MAX + cnt[i]] = w;
    cnt[i]++;
  };
  for (int o = 0; o < out; ++o) {
    const float src = scale * (float)o;
    const int a = (int)src;
    const int off = a < in - 1 ? 1 : 0;
    float l1 = src - (float)a;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    i0[o] = a; i1[o] = a + off; w1[o] = l1;
    push(a, o, 1.f - l1);
    push(a + off, o, l1);
  }
}

template <typename T>
int upload(fu_ctx* c, const std::vector<T>& v, const T** out) {
  void* d = nullptr;
  FU_HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T) + 16));
  FU_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  c->extra_allocs.push_back(d);
  *out = (const T*)d;
  return 0;
}

int build_up_tables(fu_ctx* c, int H, int W, UpTables* t) {
  std::vector<int> yi0, yi1, xi0, xi1, ybo, xbo;
  std::vector<float> yw1, xw1, ybw, xbw;
  bool ok = true;
  build_axis(H, yi0, yi1, yw1, ybo, ybw, &ok);
  build_axis(W, xi0, xi1, xw1, xbo, xbw, &ok);
  FU_REQUIRE(ok, "bilinear backward table overflow (H=%d W=%d)", H, W);
  t->scale_y = 2 * H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
  t->scale_x = 2 * W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  FU_TRY(upload(c, yi0, &t->y_i0)); FU_TRY(upload(c, yi1, &t->y_i1)); FU_TRY(upload(c, yw1, &t->y_w1));
  FU_TRY(upload(c, xi0, &t->x_i0)); FU_TRY(upload(c, xi1, &t->x_i1)); FU_TRY(upload(c, xw1, &t->x_w1));
  FU_TRY(upload(c, ybo, &t->yb_o)); FU_TRY(upload(c, ybw, &t->yb_w));
  FU_TRY(upload(c, xbo, &t->xb_o)); FU_TRY(upload(c, xbw, &t->xb_w));
  return 0;
}

int build_plan(fu_ctx* c) {
  const fu_config& f = c->cfg;
  const int base = f.base_channels;
  const int factor = f.bilinear ? 2 : 1;
  c->ch[0] = base; c->ch[1] = base * 2; c->ch[2] = base * 4; c->ch[3] = base * 8; c->ch[4] = base * 16 / factor;
  c->Hs[0] = f.height; c->Ws[0] = f.width;
  for (int l = 1; l < 5; ++l) { c->Hs[l] = c->Hs[l - 1] / 2; c->Ws[l] = c->Ws[l - 1] / 2; }
  FU_REQUIRE(c->Hs[4] >= 1 && c->Ws[4] >= 1, "tile %dx%d is too small for four 2x poolings", f.height, f.width);
  c->fusion = f.n_encoders >= 1;
  c->nE = c->fusion ? f.n_encoders : 1;
  c->nb = 5 * c->nE + 4;
  c->blk.assign(c->nb, Block());
  for (int e = 0, off = 0; e < c->nE; ++e) {
    c->enc_ch[e] = c->fusion ? f.enc_channels[e] : f.n_channels;
    c->enc_coff[e] = off;
    off += c->enc_ch[e];
    c->cin_pad0[e] = round_up(c->enc_ch[e], c->prec == PREC_F32 ? 4 : 8);
  }
  const int outs[4] = {base * 8 / factor, base * 4 / factor, base * 2 / factor, base};

  int low = c->ch[4];
  for (int i = 0; i < c->nb; ++i) {
    Block& K = c->blk[i];
    int cin, cmid, cout;
    const bool is_enc = i < 5 * c->nE;
    K.enc = is_enc ? i / 5 : 0;
    K.role = is_enc ? i % 5 : 5 + (i - 5 * c->nE);
    const int r = K.role;
    if (r == 0) { K.kind = BK_INC; K.level = 0; cin = c->enc_ch[K.enc]; cmid = cout = c->ch[0]; }
    else if (r <= 4) { K.kind = BK_DOWN; K.level = r; cin = c->ch[r - 1]; cmid = cout = c->ch[r]; }
    else {
      const int k = r - 5;
      K.kind = BK_UP; K.skip = 3 - k; K.level = 3 - k;
      if (f.bilinear) { cin = low + c->ch[3 - k]; cmid = cin / 2; cout = outs[k]; }
      else { K.ct_cin = low; K.ct_cout = low / 2; cin = low / 2 + c->ch[3 - k]; cmid = cout = outs[k]; }
      low = cout;
    }
    if (c->fusion && r == 5) {   // between the encoders and the decoder in the flat buffers (backward order stays adjacent)
      for (int l = 0; l < 5; ++l) {
        Fuse& F = c->fuse[l];
        F.C = c->ch[l];
        const std::string cn = "concat_convs." + std::to_string(l);
        F.p_w = add_param(c, cn + ".weight", {F.C, (int64_t)c->nE * F.C, 1, 1});
        F.p_b = add_param(c, cn + ".bias", {F.C});
      }
    }
    const std::string scope = !c->fusion ? "" : (is_enc ? "encoders." + std::to_string(K.enc) + "." : "decoder.");
    K.first_param = (int)c->params.size();
    if (K.kind == BK_UP && !f.bilinear) {
      const std::string up = scope + "up" + std::to_string(r - 4) + ".up";
      K.ct_w = add_param(c, up + ".weight", {K.ct_cin, K.ct_cout, 2, 2});
      K.ct_b = add_param(c, up + ".bias", {K.ct_cout});
    }
    const std::string pre = scope + dc_prefix(r);
    for (int j = 0; j < 2; ++j) {
      Conv& v = K.c[j];
      v.level = K.level;
      v.cin_real = j == 0 ? cin : cmid;
      v.cin_pad = (r == 0 && j == 0) ? c->cin_pad0[K.enc] : v.cin_real;
      v.cout = j == 0 ? cmid : cout;
      const std::string cn = pre + "." + std::to_string(j == 0 ? 0 : 3);
      const std::string bn = pre + "." + std::to_string(j == 0 ? 1 : 4);
      v.p_w = add_param(c, cn + ".weight", {v.cout, v.cin_real, 3, 3});
      v.p_b = add_param(c, cn + ".bias", {v.cout});
      v.p_g = add_param(c, bn + ".weight", {v.cout});
      v.p_beta = add_param(c, bn + ".bias", {v.cout});
      v.bn = (int)c->bns.size();
      c->bns.push_back({bn, v.cout, c->total_bn});
      c->total_bn += v.cout;
    }
    K.num_params = (int)c->params.size() - K.first_param;
  }
  const std::string dscope = c->fusion ? "decoder." : "";
  c->p_outw = add_param(c, dscope + "outc.conv.weight", {f.n_classes, base, 1, 1});
  c->p_outb = add_param(c, dscope + "outc.conv.bias", {f.n_classes});
  return 0;
}

int alloc_workspace(fu_ctx* c) {
  const fu_config& f = c->cfg;
  const int B = f.max_batch;
  Arena& A = c->arena;
  const size_t es = c->esize;
  auto act = [&](int level, int C) { return (size_t)B * c->Hs[level] * c->Ws[level] * C * es; };
  for (int e = 0; e < c->nE; ++e) A.want(&c->xin[e], act(0, c->cin_pad0[e]));
  int64_t max_stats = 0, max_bnb = 0, max_slab = 0, max_dbp = 0;
  int max_c = 0;
  for (int i = 0; i < c->nb; ++i) {
    Block& K = c->blk[i];
    for (int j = 0; j < 2; ++j) {
      Conv& v = K.c[j];
      const int H = c->Hs[v.level], W = c->Ws[v.level];
      const int64_t npix = (int64_t)B * H * W;
      A.want(&v.y, act(v.level, v.cout));
      A.want(&v.gy, act(v.level, v.cout));
      A.want(&v.mean, v.cout * sizeof(float));
      A.want(&v.invstd, v.cout * sizeof(float));
      A.want(&v.a, v.cout * sizeof(float));
      A.want(&v.b, v.cout * sizeof(float));
      A.want(&v.coef, v.cout * 2 * sizeof(float));
      A.want(&v.fold_scale, v.cout * sizeof(float));
      A.want(&v.fold_bias, v.cout * sizeof(float));
      A.want(&v.wf, conv3x3_pack_elems(c->prec, v.cin_pad, v.cout) * es);
      if (!(K.role == 0 && j == 0)) A.want(&v.wd, conv3x3_pack_elems(c->prec, v.cin_pad, v.cout) * es);
      max_stats = std::max<int64_t>(max_stats, (int64_t)conv3x3_num_stat_tiles(c->prec, B, H, W) * v.cout * 2);
      max_bnb = std::max<int64_t>(max_bnb, bn_bwd_partial_elems(v.cout, npix));
      max_dbp = std::max<int64_t>(max_dbp, bn_bwd_partial_elems(v.cout, npix) / 2);
      max_slab = std::max<int64_t>(max_slab, conv3x3_wgrad_slab_elems(c->prec, v.cin_pad, v.cout, B, H, W));
      max_c = std::max(max_c, v.cout);
    }
    if (K.kind == BK_DOWN) {
      A.want(&K.pooled, act(K.level, K.c[0].cin_real));
      A.want(&K.g_pooled, act(K.level, K.c[0].cin_real));
    } else if (K.kind == BK_UP) {
      const int clow = K.c[0].cin_real - c->ch[K.skip];
      A.want(&K.up, act(K.level, clow));
      A.want(&K.g_up, act(K.level, clow));
      if (!f.bilinear) {
        const int H = c->Hs[K.level], W = c->Ws[K.level];
        A.want(&K.u, act(K.level, K.ct_cout));            // = B h w (4 ct_cout)
        A.want(&K.g_u, act(K.level, K.ct_cout));
        A.want(&K.ct_w3, (size_t)9 * K.ct_cin * 4 * K.ct_cout * sizeof(float));
        A.want(&K.ct_dw3, (size_t)9 * K.ct_cin * 4 * K.ct_cout * sizeof(float));
        A.want(&K.ct_b4, (size_t)4 * K.ct_cout * sizeof(float));
        A.want(&K.ct_wf, conv3x3_pack_elems(c->prec, K.ct_cin, 4 * K.ct_cout) * es);
        A.want(&K.ct_wd, conv3x3_pack_elems(c->prec, K.ct_cin, 4 * K.ct_cout) * es);
        max_slab = std::max<int64_t>(max_slab, conv3x3_wgrad_slab_elems(c->prec, K.ct_cin, 4 * K.ct_cout, B, H / 2, W / 2));
        max_dbp = std::max<int64_t>(max_dbp, (int64_t)2048 * K.ct_cout);
      }
    }
  }
  for (int l = 0; l < 5 && c->fusion; ++l) {
    Fuse& F = c->fuse[l];
    const int Ccat = c->nE * F.C, H = c->Hs[l], W = c->Ws[l];
    A.want(&F.cat, act(l, Ccat));
    A.want(&F.gcat, act(l, Ccat));
    A.want(&F.y, act(l, F.C));
    A.want(&F.gy, act(l, F.C));
    A.want(&F.w3, (size_t)9 * Ccat * F.C * sizeof(float));
    A.want(&F.dw3, (size_t)9 * Ccat * F.C * sizeof(float));
    A.want(&F.wf, conv3x3_pack_elems(c->prec, Ccat, F.C) * es);
    A.want(&F.wd, conv3x3_pack_elems(c->prec, Ccat, F.C) * es);
    max_slab = std::max<int64_t>(max_slab, conv3x3_wgrad_slab_elems(c->prec, Ccat, F.C, B, H, W));
    max_dbp = std::max<int64_t>(max_dbp, (int64_t)2048 * F.C);
  }
  const int64_t npix0 = (int64_t)B * f.height * f.width;
  A.want(&c->logits, npix0 * f.n_classes * sizeof(float));
  A.want(&c->dlogits, npix0 * f.n_classes * sizeof(float));
  A.want(&c->dlogits_eff, npix0 * f.n_classes * sizeof(float));
  A.want(&c->up_scale, 256);
  A.want(&c->stats, max_stats * sizeof(float));
  A.want(&c->bnb_part, max_bnb * sizeof(float));
  c->bnb_cap = max_bnb;
  A.want(&c->db_part, max_dbp * sizeof(float));
  A.want(&c->db_part2, max_dbp * sizeof(float));
  A.want(&c->dscratch, reduce_scratch_elems(std::max(max_c, 64)) * sizeof(double));
  A.want(&c->slab, max_slab * sizeof(float));
  A.want(&c->ce_part, 2 * 1024 * sizeof(float));
  A.want(&c->hb_part, head_bwd_partial_elems(f.base_channels, f.n_classes) * sizeof(float));
  A.want(&c->loss_dev, 256);
  A.want(&c->loss_scale, 256);
  A.want(&c->guard, 256);
  A.want(&c->conf_tmp, 64 * sizeof(unsigned long long));
  A.want(&c->n_valid, 256);
  FU_TRY(A.commit());
  for (int i = 5 * c->nE; i < c->nb && f.bilinear; ++i) {
    Block& K = c->blk[i];
    const int lowlvl = K.level + 1;
    FU_TRY(build_up_tables(c, c->Hs[lowlvl], c->Ws[lowlvl], &K.upt));
  }
  return 0;
}

// the event pair for ONE conv / wgrad launch (empty when profiling is off): goes into that launch's ConvIn::opt.prof
ProfSlot prof_arm(fu_ctx* c, int cls, double flops) {
  Profiler& pr = c->prof;
  ProfSlot ps;
  if (!pr.on) return ps;
  if (pr.next + 2 > pr.pool.size()) { pr.overflow = true; return ps; }
  ProfRec r{cls, flops, pr.pool[pr.next], pr.pool[pr.next + 1]};
  pr.next += 2;
  pr.recs.push_back(r);
  ps.start = r.e0;
  ps.stop = r.e1;
  return ps;
}

inline float* P(fu_ctx* c, int idx) { return c->P + c->params[idx].off; }
inline float* G(fu_ctx* c, int idx) { return c->G + c->params[idx].off; }

int repack(fu_ctx* c, hipStream_t s, bool eval) {
  if (FU_EXP_SKIP(8) && !c->pack_tabs.empty() && c->packed_eval == eval) { c->packed_dirty = false; return 0; }
  const int use_scale = eval ? 1 : 0;
  if (eval) {
    for (int i = 0; i < c->nb; ++i)
      for (int j = 0; j < 2; ++j) {
        Conv& v = c->blk[i].c[j];
        const int64_t off = c->bns[v.bn].off;
        hipLaunchKernelGGL(k_bn_fold_eval, dim3(fu::ceil_div(v.cout, 64)), dim3(64), 0, s, v.cout, P(c, v.p_g),
                           P(c, v.p_beta), c->RM + off, c->RV + off, P(c, v.p_b), BN_EPS, v.fold_scale, v.fold_bias, v.a,
                           v.b);
      }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("eval fold launch failed: %s", hipGetErrorString(e)); return FU_ERR_HIP; }
  }
  if (c->pack_tabs.empty()) {
    bool tiled_ok = true;
    for (int i = 0; i < c->nb; ++i)
      for (int j = 0; j < 2; ++j) {
        const Conv& v = c->blk[i].c[j];
        if (v.cout % 8 != 0 || v.cin_pad % 8 != 0) tiled_ok = false;
      }
    int64_t start = 0;
    for (int i = 0; i < c->nb; ++i)
      for (int j = 0; j < 2; ++j) {
        Conv& v = c->blk[i].c[j];
        if (c->pack_tabs.empty() || c->pack_tabs.back().n == MAX_PACK) {   // one launch per MAX_PACK layers
          PackTable nt;
          nt.n = 0; nt.total = 0; nt.tiles = 0;
          c->pack_tabs.push_back(nt);
          start = 0;
        }
        PackTable& t = c->pack_tabs.back();
        PackDesc& d = t.d[t.n++];
        d.start = start;
        d.w_off = c->params[v.p_w].off;
        d.cout = v.cout; d.cin_real = v.cin_real; d.cin_pad = v.cin_pad; d.pad_ = 0;
        d.wf = v.wf; d.wd = v.wd;
        d.scale = v.fold_scale;
        start += (int64_t)9 * v.cin_pad * v.cout;
        d.tile_start = t.tiles;
        d.tiles_ci = fu::ceil_div(v.cin_pad, 32);
        t.tiles += fu::ceil_div(v.cout, 32) * d.tiles_ci;
        t.total = start;
      }
    if (!tiled_ok) for (PackTable& t : c->pack_tabs) t.tiles = 0;
  }
  const int grid = 2048;
  for (const PackTable& t : c->pack_tabs) {
    if (c->prec == PREC_F32)
      hipLaunchKernelGGL((k_pack_all<float, false>), dim3(grid), dim3(256), 0, s, c->P, t, use_scale);
    else if (t.tiles > 0 && c->prec == PREC_BF16)
      hipLaunchKernelGGL(k_pack_tiles_16<false>, dim3(t.tiles), dim3(256), 0, s, c->P, t, use_scale);
    else if (t.tiles > 0)
      hipLaunchKernelGGL(k_pack_tiles_16<true>, dim3(t.tiles), dim3(256), 0, s, c->P, t, use_scale);
    else if (c->prec == PREC_BF16)
      hipLaunchKernelGGL((k_pack_all<bf16_t, true, false>), dim3(grid), dim3(256), 0, s, c->P, t, use_scale);
    else
      hipLaunchKernelGGL((k_pack_all<bf16_t, true, true>), dim3(grid), dim3(256), 0, s, c->P, t, use_scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pack launch failed: %s", hipGetErrorString(e)); return FU_ERR_HIP; }
  }
  for (int l = 0; l < 5 && c->fusion; ++l) {
    Fuse& F = c->fuse[l];
    const int Ccat = c->nE * F.C;
    FU_TRY(launch_center_to_w3(P(c, F.p_w), (int64_t)F.C * Ccat, F.w3, s));
    FU_TRY(launch_pack_conv3x3(c->prec, F.w3, F.C, Ccat, Ccat, F.wf, F.wd, s));
  }
  if (!c->cfg.bilinear) {
    for (int i = 5 * c->nE; i < c->nb; ++i) {
      Block& K = c->blk[i];
      FU_TRY(launch_convT_to_w3(P(c, K.ct_w), P(c, K.ct_b), K.ct_cin, K.ct_cout, K.ct_w3, K.ct_b4, s));
      FU_TRY(launch_pack_conv3x3(c->prec, K.ct_w3, 4 * K.ct_cout, K.ct_cin, K.ct_cin, K.ct_wf, K.ct_wd, s));
    }
  }
  c->packed_dirty = false;
  c->packed_eval = eval;
  return 0;
}

// decoder inputs: the feature of `level` (skip connection) and the low-resolution input of up block i
Feat level_feat(fu_ctx* c, int level) {
  if (c->fusion) { Fuse& F = c->fuse[level]; return Feat{F.y, nullptr, nullptr, F.gy, F.C}; }
  Conv& v = c->blk[level].c[1];
  return Feat{v.y, v.a, v.b, v.gy, v.cout};
}
Feat low_feat(fu_ctx* c, int i) {
  if (i == 5 * c->nE) return level_feat(c, 4);
  Conv& v = c->blk[i - 1].c[1];
  return Feat{v.y, v.a, v.b, v.gy, v.cout};
}

ConvIn conv_input(fu_ctx* c, int i, int j) {
  Block& K = c->blk[i];
  ConvIn in;
  in.src1 = nullptr; in.C1 = 0; in.a0 = nullptr; in.b0 = nullptr;
  if (j == 1) {
    in.src0 = K.c[0].y; in.C0 = K.c[0].cout; in.a0 = K.c[0].a; in.b0 = K.c[0].b;
  } else if (K.kind == BK_INC) {
    in.src0 = c->xin[K.enc]; in.C0 = c->cin_pad0[K.enc];
  } else if (K.kind == BK_DOWN) {
    in.src0 = K.pooled; in.C0 = K.c[0].cin_real;
  } else {
    const Feat sk = level_feat(c, K.skip);
    in.src0 = sk.y; in.C0 = sk.C; in.a0 = sk.a; in.b0 = sk.b;
    in.src1 = K.up; in.C1 = K.c[0].cin_real - sk.C;
  }
  return in;
}

int conv_fwd(fu_ctx* c, int i, int j, int B, bool training, hipStream_t s) {
  Conv& v = c->blk[i].c[j];
  const int H = c->Hs[v.level], W = c->Ws[v.level];
  ConvIn in = conv_input(c, i, j);
  int nt = 0;
  const double fl = 2.0 * 9 * v.cin_real * v.cout * (double)B * H * W;
  in.opt.prof = prof_arm(c, FU_K_CONV3X3, fl);
  // eval: the packed weights and v.fold_bias already contain the BatchNorm (repack(eval)); y IS bn(conv(x)), v.a / v.b = 1 / 0
  FU_TRY(launch_conv3x3(c->prec, in, v.wf, training ? P(c, v.p_b) : v.fold_bias, v.y, v.cout, nullptr, 0,
                        training ? c->stats : nullptr, &nt, B, H, W, s));
  const int64_t off = c->bns[v.bn].off;
  if (training)
    FU_TRY(launch_bn_finalize(c->stats, nt, v.cout, (int64_t)B * H * W, P(c, v.p_b), P(c, v.p_g), P(c, v.p_beta),
                              BN_EPS, BN_MOMENTUM, v.mean, v.invstd, v.a, v.b, c->RM + off, c->RV + off,
                              c->NBT + v.bn, c->dscratch, s));
  return 0;
}

// Late fusion, all five levels (lf_model.py:78-90).  forward: cat <- [relu(bn(x_e))]_e ; fused = W * cat + bias.
int fuse_forward(fu_ctx* c, int B, hipStream_t s) {
  for (int l = 0; l < 5; ++l) {
    Fuse& F = c->fuse[l];
    const int Ccat = c->nE * F.C, H = c->Hs[l], W = c->Ws[l];
    const int64_t npix = (int64_t)B * H * W;
    for (int e = 0; e < c->nE; ++e) {
      Conv& v = c->blk[5 * e + l].c[1];
      FU_TRY(launch_copy_channels(c->prec, v.y, v.cout, 0, v.a, v.b, F.cat, Ccat, e * F.C, F.C, npix, s));
    }
    ConvIn in{F.cat, Ccat, nullptr, nullptr, nullptr, 0, true};   // 1x1: only the centre tap of wf is non-zero
    FU_TRY(launch_conv3x3(c->prec, in, F.wf, P(c, F.p_b), F.y, F.C, nullptr, 0, nullptr, nullptr, B, H, W, s));
  }
  return 0;
}

// backward of the five fusion convs: needs every F.gy (complete after up1's backward); writes dL/d(activated feature)
// of every encoder level ("=": the encoders' pool backward accumulates into it afterwards, as the skip gradient of
// the plain UNet) and the fusion parameters' gradients
int fuse_backward(fu_ctx* c, int B, hipStream_t s) {
  for (int l = 4; l >= 0; --l) {
    Fuse& F = c->fuse[l];
    const int Ccat = c->nE * F.C, H = c->Hs[l], W = c->Ws[l];
    const int64_t npix = (int64_t)B * H * W;
    int ndbp = 0;
    FU_TRY(launch_channel_partial_sums(c->prec, F.gy, F.C, npix, c->db_part, &ndbp, s));
    ConvIn in{F.cat, Ccat, nullptr, nullptr, nullptr, 0, true};   // only the centre tap of dw3 is computed (and read)
    FU_TRY(launch_conv3x3_wgrad(c->prec, in, F.gy, F.C, c->slab, F.dw3, Ccat, c->db_part, ndbp, G(c, F.p_b), B, H, W,
                                s));
    FU_TRY(launch_center_from_w3(F.dw3, (int64_t)F.C * Ccat, G(c, F.p_w), s));
    ConvIn gin{F.gy, F.C, nullptr, nullptr, nullptr, 0, true};    // the flipped 3x3 of a centre tap is a centre tap
    if (c->nE == 2 && F.C % 64 == 0) {
      // two encoders: the conv kernels' two destinations ARE the encoders' skip gradients (no concat gradient, no split)
      FU_TRY(launch_conv3x3(c->prec, gin, F.wd, nullptr, c->blk[l].c[1].gy, F.C, c->blk[5 + l].c[1].gy, F.C, nullptr,
                            nullptr, B, H, W, s));
    } else {
      FU_TRY(launch_conv3x3(c->prec, gin, F.wd, nullptr, F.gcat, Ccat, nullptr, 0, nullptr, nullptr, B, H, W, s));
      for (int e = 0; e < c->nE; ++e) {
        Conv& v = c->blk[5 * e + l].c[1];
        FU_TRY(launch_copy_channels(c->prec, F.gcat, Ccat, e * F.C, nullptr, nullptr, v.gy, v.cout, 0, F.C, npix, s));
      }
    }
  }
  return 0;
}

int forward_impl(fu_ctx* c, const float* x, const SrcList* srcs, int B, bool training, float* logits_out,
                 hipStream_t s) {
  const fu_config& f = c->cfg;
  if (c->packed_dirty || c->packed_eval != !training) FU_TRY(repack(c, s, !training));
  for (int e = 0; e < c->nE; ++e) {
    if (srcs)     // several input tensors side by side (fu_forward_srcs): the concat happens inside the layout conversion
      FU_TRY(launch_gather_nchw_to_nhwc(c->prec, *srcs, c->xin[e], B, c->enc_ch[e], f.height, f.width, c->cin_pad0[e],
                                        c->enc_coff[e], s));
    else
      FU_TRY(launch_nchw_to_nhwc(c->prec, x, c->xin[e], B, c->enc_ch[e], f.height, f.width, c->cin_pad0[e], s,
                                 f.n_channels, c->enc_coff[e]));
  }
  for (int i = 0; i < c->nb; ++i) {
    Block& K = c->blk[i];
    if (c->fusion && i == 5 * c->nE) FU_TRY(fuse_forward(c, B, s));
    if (K.kind == BK_DOWN) {
      Conv& pv = c->blk[i - 1].c[1];
      FU_TRY(launch_maxpool2(c->prec, pv.y, pv.a, pv.b, K.pooled, B, c->Hs[pv.level], c->Ws[pv.level], pv.cout, s));
    } else if (K.kind == BK_UP) {
      const Feat pv = low_feat(c, i);
      const int h = c->Hs[K.level + 1], w = c->Ws[K.level + 1], H = c->Hs[K.level], W = c->Ws[K.level];
      if (c->cfg.bilinear) {
        FU_TRY(launch_upsample2(c->prec, pv.y, pv.a, pv.b, K.up, B, h, w, pv.C, H, W, K.upt, s));
      } else {
        // ConvTranspose2d(k2,s2) (unet.py:48-51): the four phase GEMMs as one 1x1 conv of relu(bn(low)) with 4 ct_cout
        // output channels at the low resolution, then depth-to-space + F.pad (unet.py:57-62)
        ConvIn lin{pv.y, pv.C, pv.a, pv.b, nullptr, 0, true};
        FU_TRY(launch_conv3x3(c->prec, lin, K.ct_wf, K.ct_b4, K.u, 4 * K.ct_cout, nullptr, 0, nullptr, nullptr, B, h, w, s));
        FU_TRY(launch_depth_to_space(c->prec, K.u, K.up, B, h, w, K.ct_cout, H, W, s));
      }
    }
    FU_TRY(conv_fwd(c, i, 0, B, training, s));
    FU_TRY(conv_fwd(c, i, 1, B, training, s));
  }
  Conv& last = c->blk[c->nb - 1].c[1];
  FU_TRY(launch_head_fwd(c->prec, last.y, last.a, last.b, P(c, c->p_outw), P(c, c->p_outb), f.base_channels,
                         f.n_classes, B, f.height, f.width, c->logits, logits_out, s));
  c->last_batch = B;
  c->fwd_training = training;
  c->have_loss = false;
  c->have_up_scale = false;
  return 0;
}

// testing hook (fu_test_bnb_separate) and A/B switch (environment FU_BNB_SEPARATE): 1 = BatchNorm-backward sums always by
// their own reduce pass, never from the producer of the gradient (BnbFuse, fu_common.h)
static int g_bnb_separate = getenv("FU_BNB_SEPARATE") != nullptr ? 1 : 0;
static int g_head_store_g = getenv("FU_HEAD_STORE_G") != nullptr ? 1 : 0;   // testing hook / A-B (fu_test_head_store_g): the head backward stores its data gradient even where the apply pass could recompute it
// testing hook (fu_test_perturb_bnb_sums): the fused sums are multiplied by this factor after the kernel that emitted
// them -- the negative control of the parity tests (a wrong fused sum must make them fail); 1 = off, no launch
static float g_test_perturb_bnb = 1.f;
__global__ void k_scale_floats(float* x, int64_t n, float f) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= f;
}
int perturb_bnb(float* part, int tiles, int C, hipStream_t s) {
  if (g_test_perturb_bnb == 1.f || tiles <= 0) return 0;
  hipLaunchKernelGGL(k_scale_floats, dim3(256), dim3(256), 0, s, part, (int64_t)tiles * C * 2, g_test_perturb_bnb);
  return hipGetLastError() == hipSuccess ? 0 : FU_ERR_HIP;
}

int backward_conv(fu_ctx* c, int i, int j, int B, hipStream_t s) {
  Block& K = c->blk[i];
  Conv& v = K.c[j];
  const int H = c->Hs[v.level], W = c->Ws[v.level];
  const int64_t npix = (int64_t)B * H * W;
  int ndb = 0;
  // BN + ReLU backward: gy <- dL/dy ; dgamma, dbeta
  // The weight-gradient chain of this conv (wgrad, slab reduce, transpose) depends only on gy and on saved activations
  // and nothing in the rest of backward depends on it: it runs on a side stream, concurrently with this conv's dgrad
  // and the next BN backward (its 8-wave workgroups spend more than half of every stage staging with the MFMA pipe
  // idle, tools/stamp_wgrad.py; the dgrad workgroups that fit beside them on a CU use it).  db partials alternate
  // between two buffers so that the main stream only has to wait for the wgrad of two convs ago.
  const bool side = c->side != nullptr && c->side_mode != 0;
  const int par = c->wg_parity;
  float* dbp = (side && par) ? c->db_part2 : c->db_part;
  if (side && c->wg_pending[par]) FU_HIP_CHECK(hipStreamWaitEvent(s, c->ev_wg[par], 0));   // buffer free again
  FU_TRY(launch_bn_bwd(c->prec, v.gy, v.y, v.cout, npix, v.a, v.b, v.mean, v.invstd, P(c, v.p_g), G(c, v.p_g),
                       G(c, v.p_beta), c->bnb_part, v.coef, dbp, &ndb, c->dscratch, s, v.pool_g, B, H, W, v.bnb_tiles,
                       v.head.dl ? &v.head : nullptr));
  v.pool_g = nullptr;
  v.bnb_tiles = 0;
  v.head = HeadGrad{};
  // weight (and bias) gradient
  ConvIn in = conv_input(c, i, j);
  const double fl = 2.0 * 9 * v.cin_real * v.cout * (double)B * H * W;
  hipStream_t ws = s;
  if (side) {
    FU_HIP_CHECK(hipEventRecord(c->ev_gy, s));
    FU_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_gy, 0));
    ws = c->side;
  }
  in.opt.prof = prof_arm(c, FU_K_WGRAD, fl);
  FU_TRY(launch_conv3x3_wgrad(c->prec, in, v.gy, v.cout, c->slab, G(c, v.p_w), v.cin_real, dbp, ndb,
                              G(c, v.p_b), B, H, W, ws));
  if (side) {
    FU_HIP_CHECK(hipEventRecord(c->ev_wg[par], c->side));
    c->wg_pending[par] = true;
    c->wg_parity ^= 1;
  }
  // data gradient
  ConvIn din{v.gy, v.cout, nullptr, nullptr, nullptr, 0};
  if (!(K.role == 0 && j == 0)) din.opt.prof = prof_arm(c, FU_K_CONV3X3, fl);
  if (j == 1) {
    // this dgrad's destination is dL/d relu(bn(y)) of the block's first conv: a kernel that can (the row-stationary 16-bit
    // one) also leaves that BatchNorm's backward sums in bnb_part, consumed by the very next launch_bn_bwd on this stream
    Conv& v0 = K.c[0];
    const bool separate = g_bnb_separate != 0;      // testing hook / FU_BNB_SEPARATE: always the separate reduce pass
    int tiles = 0;
    BnbFuse f;
    if (c->prec != PREC_F32 && !c->sync.hook && !separate) {
      f.y = v0.y; f.a = v0.a; f.b = v0.b; f.mean = v0.mean; f.invstd = v0.invstd;
      f.part = c->bnb_part; f.max_elems = c->bnb_cap; f.tiles_out = &tiles;
      din.opt.bnb = &f;
    }
    FU_TRY(launch_conv3x3(c->prec, din, v.wd, nullptr, v0.gy, v0.cout, nullptr, 0, nullptr, nullptr, B, H, W, s));
    v0.bnb_tiles = tiles;
    FU_TRY(perturb_bnb(c->bnb_part, tiles, v0.cout, s));
  } else if (K.kind == BK_DOWN) {
    FU_TRY(launch_conv3x3(c->prec, din, v.wd, nullptr, K.g_pooled, v.cin_real, nullptr, 0, nullptr, nullptr, B, H, W,
                          s));
    // the pool's backward (route g_pooled to the first argmax of every window, add to the skip gradient) is folded into
    // the BN backward of the pooled tensor: the next backward block on this stream (k_bn_bwd_pool)
    Conv& pv = c->blk[i - 1].c[1];
    pv.pool_g = K.g_pooled;
  } else if (K.kind == BK_UP) {
    const Feat sk = level_feat(c, K.skip);
    const Feat pv = low_feat(c, i);
    FU_TRY(launch_conv3x3(c->prec, din, v.wd, nullptr, sk.gy, sk.C, K.g_up, v.cin_real - sk.C, nullptr, nullptr,
                          B, H, W, s));
    const int h = c->Hs[K.level + 1], w = c->Ws[K.level + 1];
    if (c->cfg.bilinear) {
      FU_TRY(launch_upsample2_bwd(c->prec, K.g_up, pv.gy, B, h, w, pv.C, H, W, K.upt, s));
    } else {
      // g4 = space-to-depth of dL/d(up) (the F.pad region carries no gradient: it is simply not gathered); the 1x1 conv's
      // bias gradient is the sum of g4 over pixels and phases, its weight gradient a one-tap wgrad, its data gradient a
      // 1x1 conv with the transposed weights
      FU_TRY(launch_space_to_depth(c->prec, K.g_up, K.g_u, B, h, w, K.ct_cout, H, W, s));
      int ndbp = 0;
      FU_TRY(launch_channel_partial_sums(c->prec, K.g_u, K.ct_cout, (int64_t)B * h * w * 4, c->db_part, &ndbp, s));
      FU_TRY(launch_colsum_partials(c->db_part, ndbp, K.ct_cout, G(c, K.ct_b), s));
      ConvIn lin{pv.y, pv.C, pv.a, pv.b, nullptr, 0, true};
      FU_TRY(launch_conv3x3_wgrad(c->prec, lin, K.g_u, 4 * K.ct_cout, c->slab, K.ct_dw3, K.ct_cin, nullptr, 0, nullptr, B, h,
                                  w, s));
      FU_TRY(launch_convT_grad_from_w3(K.ct_dw3, K.ct_cin, K.ct_cout, G(c, K.ct_w), s));
      ConvIn gin{K.g_u, 4 * K.ct_cout, nullptr, nullptr, nullptr, 0, true};
      FU_TRY(launch_conv3x3(c->prec, gin, K.ct_wd, nullptr, pv.gy, K.ct_cin, nullptr, 0, nullptr, nullptr, B, h, w, s));
    }
  }
  return 0;
}

// backward order: 0 = head, 1..4 = up4..up1, [5 = the fusion convs], then down4..inc of the last encoder ... the first
int num_backward_blocks(const fu_ctx* c) { return 5 + (c->fusion ? 1 : 0) + 5 * c->nE; }
int backward_block_index(const fu_ctx* c, int block) {
  if (block <= 4) return c->nb - block;                       // nb-1 (up4) ... nb-4 (up1)
  const int k = block - 5 - (c->fusion ? 1 : 0);              // 0 .. 5*nE-1 over the encoders, last encoder first
  return 5 * c->nE - 1 - k;
}

int backward_block_impl(fu_ctx* c, int block, const float* dlogits_ext, hipStream_t s, bool join) {
  const fu_config& f = c->cfg;
  const int B = c->last_batch;
  if (block == 0) {
    if (dlogits_ext) {
      FU_TRY(launch_dlogits_from_nchw(dlogits_ext, c->dlogits, f.n_classes, B, f.height, f.width, s));
      c->have_up_scale = false;     // the caller's dlogits IS the whole upstream gradient
    } else {
      FU_REQUIRE(c->have_loss, "fu_backward: no dlogits given and no fu_loss_* call since the last forward");
    }
    // What the head backward reads: the stored gradient itself, or -- out of place, so that a repeated backward of the same
    // loss starts from the same input -- times the upstream gradient of loss.backward() and, for fp16 gradient maps, the
    // power-of-two loss scale chosen from max|dL/dlogits| (UnscaleScope removes it where parameter gradients are written)
    const float* dl = c->dlogits;
    if (c->prec == PREC_F16 || c->have_up_scale) {
      FU_TRY(launch_loss_grad_eff(c->dlogits, c->dlogits_eff, (int64_t)B * f.height * f.width * f.n_classes,
                                  c->have_up_scale ? c->up_scale : nullptr, c->ce_part,
                                  c->prec == PREC_F16 ? c->loss_scale : nullptr, s,
                                  c->prec == PREC_F16 ? c->guard : nullptr));
      dl = c->dlogits_eff;
    }
    Conv& last = c->blk[c->nb - 1].c[1];
    // the head's data gradient is dL/d relu(bn(y)) of the last conv: it can leave that BatchNorm's backward sums behind
    BnbFuse fz;
    int tiles = 0;
    const bool want = c->prec != PREC_F32 && !c->sync.hook && !g_bnb_separate;
    if (want) {
      fz.y = last.y; fz.a = last.a; fz.b = last.b; fz.mean = last.mean; fz.invstd = last.invstd;
      fz.part = c->bnb_part; fz.max_elems = c->bnb_cap; fz.tiles_out = &tiles;
      // ... and then g = dl . w need not be stored at all: the apply pass of that BatchNorm recomputes it (HeadGrad)
      fz.skip_g = last.cout % 8 == 0 && 2048 % last.cout == 0 && !g_head_store_g;
    }
    FU_TRY(launch_head_bwd(c->prec, dl, last.y, last.a, last.b, P(c, c->p_outw), f.base_channels, f.n_classes,
                           (int64_t)B * f.height * f.width, last.gy, c->hb_part, G(c, c->p_outw), G(c, c->p_outb), s,
                           want ? &fz : nullptr));
    last.bnb_tiles = tiles;
    FU_REQUIRE(!(want && fz.skip_g) || tiles > 0, "head backward: the fused BatchNorm sums were refused (partials %lld floats)",
               (long long)c->bnb_cap);
    last.head = HeadGrad{};
    if (want && fz.skip_g) { last.head.dl = dl; last.head.w = P(c, c->p_outw); last.head.ncls = f.n_classes; }
    FU_TRY(perturb_bnb(c->bnb_part, tiles, last.cout, s));
    return 0;
  }
  if (c->fusion && block == 5) {
    // shares the slab and the bias-gradient partials with the side stream's weight-gradient chain: join it first (the
    // encoders' chains that follow are ordered behind this block by their ev_gy events)
    if (c->side && c->side_mode != 0) {
      FU_HIP_CHECK(hipEventRecord(c->ev_blk, c->side));
      FU_HIP_CHECK(hipStreamWaitEvent(s, c->ev_blk, 0));
      c->wg_pending[0] = c->wg_pending[1] = false;
    }
    return fuse_backward(c, B, s);
  }
  const int i = backward_block_index(c, block);
  FU_TRY(backward_conv(c, i, 1, B, s));
  FU_TRY(backward_conv(c, i, 0, B, s));
  if (c->side && c->side_mode != 0 && join) {   // the block's gradients are complete (for the caller's all-reduce / Adam) once the side stream is
    FU_HIP_CHECK(hipEventRecord(c->ev_blk, c->side));
    FU_HIP_CHECK(hipStreamWaitEvent(s, c->ev_blk, 0));
    c->wg_pending[0] = c->wg_pending[1] = false;
  }
  return 0;
}

int check_fwd_args(fu_ctx* c, const float* x, int batch) {
  FU_REQUIRE(c != nullptr, "null context");
  FU_REQUIRE(c->P && c->RM && c->RV && c->NBT, "fu_forward: buffers not bound (fu_bind_buffers)");
  FU_REQUIRE(x != nullptr, "fu_forward: null input");
  FU_REQUIRE(batch >= 1 && batch <= c->cfg.max_batch, "fu_forward: batch %d outside 1..%d", batch, c->cfg.max_batch);
  return 0;
}

double conv_flops(fu_ctx* c, bool train) {
  double fwd = 0.0, first = 0.0;
  for (int i = 0; i < c->nb; ++i)
    for (int j = 0; j < 2; ++j) {
      const Conv& v = c->blk[i].c[j];
      const double fl = 2.0 * 9 * v.cin_real * v.cout * c->Hs[v.level] * c->Ws[v.level];
      if (c->blk[i].role == 0 && j == 0) first += fl;      // the encoders' first convs have no data gradient
      fwd += fl;
    }
  for (int l = 0; l < 5 && c->fusion; ++l)                   // the 1x1 fusion convs (the MACs they need, not the 3x3 run)
    fwd += 2.0 * c->nE * c->fuse[l].C * c->fuse[l].C * c->Hs[l] * c->Ws[l];
  if (!c->cfg.bilinear)
    for (int i = 5 * c->nE; i < c->nb; ++i) {
      const Block& K = c->blk[i];
      fwd += 2.0 * 4 * K.ct_cin * K.ct_cout * c->Hs[K.level + 1] * c->Ws[K.level + 1];
    }
  fwd += 2.0 * c->cfg.base_channels * c->cfg.n_classes * c->cfg.height * c->cfg.width;
  return train ? 3.0 * fwd - first : fwd;
}

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

int fu_abi_version(void) { return FU_ABI_VERSION; }
const char* fu_last_error(void) { return get_error(); }

int fu_create(const fu_config* cfg, fu_ctx** out) {
  FU_REQUIRE(cfg && out, "fu_create: null argument");
  FU_REQUIRE(cfg->struct_size == (int32_t)sizeof(fu_config), "fu_create: fu_config size mismatch (%d vs %zu)",
             cfg->struct_size, sizeof(fu_config));
  FU_REQUIRE(cfg->n_channels >= 1 && cfg->n_channels <= 64, "n_channels must be 1..64");
  FU_REQUIRE(cfg->n_classes >= 1 && cfg->n_classes <= HEAD_MAX_CLS, "n_classes must be 1..%d", HEAD_MAX_CLS);
  const int b = cfg->base_channels;
  FU_REQUIRE(b == 4 || b == 8 || b == 16 || b == 32 || b == 64, "base_channels must be 4, 8, 16, 32 or 64");
  FU_REQUIRE(cfg->max_batch >= 1, "max_batch must be >= 1");
  FU_REQUIRE(cfg->height >= 16 && cfg->width >= 16, "tile must be at least 16x16");
  FU_REQUIRE(cfg->precision == FU_F32 || cfg->precision == FU_BF16 || cfg->precision == FU_F16, "unknown precision %d",
             cfg->precision);
  FU_REQUIRE(cfg->bilinear || b == 64, "bilinear=0 exists only at base_channels 64 (the reference's UNetDecoder "
             "channel plan is inconsistent for bilinear=False, unet.py:176-183)");
  FU_REQUIRE(cfg->precision == FU_F32 || b >= 8, "bf16 / fp16 precision needs base_channels >= 8");
  FU_REQUIRE(cfg->n_encoders >= 0 && cfg->n_encoders <= FU_MAX_ENCODERS, "n_encoders must be 0..%d", FU_MAX_ENCODERS);
  if (cfg->n_encoders >= 1) {
    FU_REQUIRE(cfg->bilinear, "late fusion exists only with bilinear upsampling (lf_model.py:38)");
    int sum = 0;
    for (int e = 0; e < cfg->n_encoders; ++e) {
      FU_REQUIRE(cfg->enc_channels[e] >= 1, "enc_channels[%d] must be >= 1", e);
      sum += cfg->enc_channels[e];
    }
    FU_REQUIRE(sum == cfg->n_channels, "enc_channels sum to %d, n_channels is %d", sum, cfg->n_channels);
  }
  FU_HIP_CHECK(hipSetDevice(cfg->device));
  fu_ctx* c = new (std::nothrow) fu_ctx();
  FU_REQUIRE(c, "out of host memory");
  c->cfg = *cfg;
  c->prec = cfg->precision == FU_F32 ? PREC_F32 : (cfg->precision == FU_BF16 ? PREC_BF16 : PREC_F16);
  c->esize = c->prec == PREC_F32 ? 4 : 2;
  int st = build_plan(c);
  if (st == 0) st = alloc_workspace(c);
  // side stream for the weight-gradient chain (bilinear nets; the ConvTranspose variant shares the slab with its own
  // wgrad on the main stream).  FU_NO_SIDE_STREAM=1 keeps everything on the caller's stream.
  // (bf16 only: in fp32 mode the concurrency changes nothing in the step time -- both kernels are MFMA bound at 0.7 of
  //  the fp32 peak -- and only inflates the per-launch durations of the parity build)
  if (st == 0 && c->cfg.bilinear && c->prec != fu::PREC_F32 && !getenv("FU_NO_SIDE_STREAM")) {
    int plo = 0, phi = 0;
    (void)hipDeviceGetStreamPriorityRange(&plo, &phi);      // plo = the lowest priority (numerically largest)
#ifdef FU_EXPERIMENTS   // A/B knob: FU_SIDE_CU_RESERVE=k keeps k of every 8 CUs (mask bits i with i % 8 < k) out of the side stream
    if (const char* e = getenv("FU_SIDE_CU_RESERVE")) {
      const int k = atoi(e);
      uint32_t mask[8];
      for (int w = 0; w < 8; ++w) {
        mask[w] = 0;
        for (int b = 0; b < 32; ++b) if (((w * 32 + b) % 8) >= k) mask[w] |= 1u << b;
      }
      if (hipExtStreamCreateWithCUMask(&c->side_def, 8, mask) != hipSuccess) c->side_def = nullptr;
      if (c->side_def) setenv("FU_SIDE_PRIO_DEFAULT", "1", 1);   // one masked stream serves both modes
    } else
#endif
    if (hipStreamCreateWithFlags(&c->side_def, hipStreamNonBlocking) != hipSuccess) c->side_def = nullptr;
    if (c->side_def && !getenv("FU_SIDE_PRIO_DEFAULT") &&
        hipStreamCreateWithPriority(&c->side_lo, hipStreamNonBlocking, plo) != hipSuccess)
      c->side_lo = nullptr;
    c->side = c->side_lo ? c->side_lo : c->side_def;        // side_mode starts at 1
    if (c->side) {
      bool ok = hipEventCreateWithFlags(&c->ev_gy, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_wg[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_wg[1], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_blk, hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        if (c->side_lo) (void)hipStreamDestroy(c->side_lo);
        (void)hipStreamDestroy(c->side_def);
        c->side = c->side_lo = c->side_def = nullptr;
      }
    }
  }
  if (st != 0) { fu_destroy(c); return st; }
  *out = c;
  return FU_OK;
}

int fu_destroy(fu_ctx* c) {
  if (!c) return FU_OK;
  (void)hipSetDevice(c->cfg.device);
  (void)hipDeviceSynchronize();
  (void)fu_dp_destroy(c);
  for (hipEvent_t e : c->prof.pool) (void)hipEventDestroy(e);
  if (c->side) {
    if (c->side_lo) (void)hipStreamDestroy(c->side_lo);
    if (c->side_def) (void)hipStreamDestroy(c->side_def);
    for (hipEvent_t e : {c->ev_gy, c->ev_wg[0], c->ev_wg[1], c->ev_blk}) if (e) (void)hipEventDestroy(e);
  }
  for (hipEvent_t e : c->ev_fence) if (e) (void)hipEventDestroy(e);   // (fu_backward_fence creates them without a side stream too)
  if (c->arena.base) (void)hipFree(c->arena.base);
  for (void* p : c->extra_allocs) (void)hipFree(p);
  delete c;
  return FU_OK;
}

int fu_num_params(const fu_ctx* c) { return c ? (int)c->params.size() : 0; }
int64_t fu_total_param_elems(const fu_ctx* c) { return c ? c->total_params : 0; }
int fu_param_info(const fu_ctx* c, int index, const char** name, int32_t* ndim, int64_t shape[4], int64_t* flat_offset) {
  FU_REQUIRE(c && index >= 0 && index < (int)c->params.size(), "fu_param_info: bad index %d", index);
  const ParamInfo& p = c->params[index];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = p.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
  if (flat_offset) *flat_offset = p.off;
  return FU_OK;
}
int fu_num_bn(const fu_ctx* c) { return c ? (int)c->bns.size() : 0; }
int64_t fu_total_bn_channels(const fu_ctx* c) { return c ? c->total_bn : 0; }
int fu_bn_info(const fu_ctx* c, int index, const char** name, int32_t* channels, int64_t* flat_offset) {
  FU_REQUIRE(c && index >= 0 && index < (int)c->bns.size(), "fu_bn_info: bad index %d", index);
  if (name) *name = c->bns[index].name.c_str();
  if (channels) *channels = c->bns[index].C;
  if (flat_offset) *flat_offset = c->bns[index].off;
  return FU_OK;
}

int fu_bind_buffers(fu_ctx* c, float* params, float* grads, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked) {
  FU_REQUIRE(c && params && running_mean && running_var && num_batches_tracked, "fu_bind_buffers: null buffer");
  c->P = params; c->G = grads; c->RM = running_mean; c->RV = running_var; c->NBT = num_batches_tracked;
  c->packed_dirty = true;
  c->pack_tabs.clear();
  return FU_OK;
}
int fu_bind_adam_state(fu_ctx* c, float* exp_avg, float* exp_avg_sq) {
  FU_REQUIRE(c && exp_avg && exp_avg_sq, "fu_bind_adam_state: null argument");
  c->adam_m = exp_avg; c->adam_v = exp_avg_sq;
  return FU_OK;
}
int fu_params_changed(fu_ctx* c) {
  FU_REQUIRE(c, "null context");
  c->packed_dirty = true;
  return FU_OK;
}

namespace {
struct SyncScope {   // makes the context's exact-sync descriptor visible to the launchers for one API call
  explicit SyncScope(const fu_ctx* c, bool on) { fu::g_sync = (on && c && c->sync.hook) ? &c->sync : nullptr; }
  ~SyncScope() { fu::g_sync = nullptr; }
};
struct UnscaleScope {   // backward calls in fp16 mode: parameter gradients are written times 1 / loss scale
  explicit UnscaleScope(const fu_ctx* c) { fu::g_grad_unscale = (c && c->prec == PREC_F16) ? c->loss_scale + 1 : nullptr; }
  ~UnscaleScope() { fu::g_grad_unscale = nullptr; }
};
}  // namespace

int fu_set_exact_sync(fu_ctx* c, fu_sync_hook hook, void* user, int world, void* exchange, int64_t exchange_bytes) {
  FU_REQUIRE(c, "null context");
  if (!hook || world <= 1) { c->sync = fu::SyncDesc(); return FU_OK; }
  if (c->prec == PREC_F16) {   // every rank picks its own loss scale: the summed BN-backward statistics would mix scales
    set_error("fu_set_exact_sync: the exact (SyncBN) mode is not available in fp16 precision; use bf16 or fp32");
    return FU_ERR_UNSUPPORTED;
  }
  FU_REQUIRE(exchange && exchange_bytes >= fu_exact_sync_bytes(c), "fu_set_exact_sync: exchange buffer of at least %lld bytes needed",
             (long long)fu_exact_sync_bytes(c));
  c->sync.hook = hook; c->sync.user = user; c->sync.world = world; c->sync.xbuf = exchange; c->sync.xbytes = exchange_bytes;
  return FU_OK;
}
int64_t fu_exact_sync_bytes(const fu_ctx* c) {
  if (!c) return 0;
  int max_c = 64;
  for (const BnInfo& b : c->bns) max_c = std::max(max_c, b.C);
  return std::max<int64_t>(fu::reduce_scratch_elems(max_c) * (int64_t)sizeof(double), 2 * 1024 * (int64_t)sizeof(float));
}

int fu_forward(fu_ctx* c, const float* x, int batch, int training, float* logits_out, fu_stream stream) {
  FU_TRY(check_fwd_args(c, x, batch));
  SyncScope sc(c, training != 0);   // eval-mode BN uses the running statistics: nothing to exchange
  return forward_impl(c, x, nullptr, batch, training != 0, logits_out, (hipStream_t)stream);
}

int fu_forward_srcs(fu_ctx* c, const float* const* srcs, const int32_t* src_channels, int n_src, int batch, int training,
                    float* logits_out, fu_stream stream) {
  FU_REQUIRE(srcs && src_channels && n_src >= 1 && n_src <= 8, "fu_forward_srcs: 1..8 sources");
  FU_TRY(check_fwd_args(c, srcs[0], batch));
  SrcList S;
  S.n = n_src;
  int off = 0;
  for (int k = 0; k < n_src; ++k) {
    FU_REQUIRE(srcs[k] && src_channels[k] >= 1, "fu_forward_srcs: bad source %d", k);
    S.p[k] = srcs[k]; S.c[k] = src_channels[k]; S.coff[k] = off; off += src_channels[k];
  }
  S.coff[n_src] = off;
  FU_REQUIRE(off == c->cfg.n_channels, "fu_forward_srcs: the sources have %d channels in all, the model takes %d", off,
             c->cfg.n_channels);
  SyncScope sc(c, training != 0);
  return forward_impl(c, nullptr, &S, batch, training != 0, logits_out, (hipStream_t)stream);
}

int fu_loss_ce(fu_ctx* c, const int64_t* target, int ignore_index, float* loss_out, int64_t* confusion_out,
               int64_t* n_valid_out, fu_stream stream) {
  FU_REQUIRE(c && target, "fu_loss_ce: null argument");
  FU_REQUIRE(c->last_batch > 0, "fu_loss_ce: no forward pass yet");
  hipStream_t s = (hipStream_t)stream;
  SyncScope sc(c, c->fwd_training);
  const int64_t npix = (int64_t)c->last_batch * c->cfg.height * c->cfg.width;
  FU_TRY(launch_ce_loss(c->logits, target, c->cfg.n_classes, ignore_index, npix, c->ce_part,
                        loss_out ? loss_out : c->loss_dev, c->n_valid, confusion_out, n_valid_out, c->conf_tmp, s));
  if (c->fwd_training) {
    FU_TRY(launch_ce_grad(c->logits, target, c->cfg.n_classes, ignore_index, npix, c->n_valid, c->dlogits, s));
    c->have_loss = true;
    c->have_up_scale = false;
  }
  return FU_OK;
}

int fu_loss_bce_dice(fu_ctx* c, const int64_t* target, int ignore_index, float dice_weight, float* loss_out,
                     fu_stream stream) {
  FU_REQUIRE(c && target, "fu_loss_bce_dice: null argument");
  FU_REQUIRE(c->last_batch > 0, "fu_loss_bce_dice: no forward pass yet");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)c->last_batch * c->cfg.height * c->cfg.width;
  FU_TRY(launch_bce_dice(c->logits, target, c->cfg.n_classes, ignore_index, npix, dice_weight, c->ce_part,
                         c->loss_dev + 8, loss_out ? loss_out : c->loss_dev, c->n_valid,
                         c->fwd_training ? c->dlogits : nullptr, s));
  if (c->fwd_training) { c->have_loss = true; c->have_up_scale = false; }
  return FU_OK;
}

int fu_scale_loss_grad(fu_ctx* c, const float* scale_dev, fu_stream stream) {
  FU_REQUIRE(c && scale_dev, "fu_scale_loss_grad: null argument");
  if (!(c->last_batch > 0 && c->fwd_training && c->have_loss)) {
    set_error("fu_scale_loss_grad: no loss gradient stored (training fu_forward + fu_loss_* first)");
    return FU_ERR_STATE;
  }
  // kept as a device scalar and applied when the backward forms the head's input (backward_block_impl, block 0): the stored
  // gradient is not touched, calling this twice for one loss replaces the factor instead of compounding it
  FU_HIP_CHECK(hipMemcpyAsync(c->up_scale, scale_dev, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  c->have_up_scale = true;
  return FU_OK;
}

int fu_num_blocks(const fu_ctx* c) { return c ? num_backward_blocks(c) : 0; }

int fu_backward_block(fu_ctx* c, int block, const float* dlogits, fu_stream stream) {
  FU_REQUIRE(c, "null context");
  FU_REQUIRE(block >= 0 && block < num_backward_blocks(c), "fu_backward_block: block %d outside 0..%d", block,
             num_backward_blocks(c) - 1);
  if (!(c->last_batch > 0 && c->fwd_training)) {
    set_error("fu_backward: the last fu_forward was not a training forward");
    return FU_ERR_STATE;
  }
  FU_REQUIRE(c->G, "fu_backward: no gradient buffer bound");
  SyncScope sc(c, true);
  UnscaleScope us(c);
  return backward_block_impl(c, block, dlogits, (hipStream_t)stream, c->side_mode != 2);
}

int fu_set_side_stream(fu_ctx* c, int mode) {
  FU_REQUIRE(c && mode >= 0 && mode <= 2, "fu_set_side_stream: mode must be 0, 1 or 2");
  c->side_mode = mode;
  if (c->side) c->side = (mode == 2 || !c->side_lo) ? c->side_def : c->side_lo;
  return FU_OK;
}

int fu_backward_join(fu_ctx* c, fu_stream stream) {
  FU_REQUIRE(c, "null context");
  if (c->side && c->side_mode != 0) {
    FU_HIP_CHECK(hipEventRecord(c->ev_blk, c->side));
    FU_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, c->ev_blk, 0));
    c->wg_pending[0] = c->wg_pending[1] = false;
  }
  return FU_OK;
}

int fu_backward_fence(fu_ctx* c, fu_stream stream, fu_stream waiter) {
  FU_REQUIRE(c, "null context");
  FU_REQUIRE(stream != waiter, "fu_backward_fence: the waiting stream must not be the compute stream (use fu_backward_join)");
  if (!c->ev_fence[0]) FU_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fence[0], hipEventDisableTiming));
  if (!c->ev_fence[1]) FU_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fence[1], hipEventDisableTiming));
  FU_HIP_CHECK(hipEventRecord(c->ev_fence[0], (hipStream_t)stream));
  FU_HIP_CHECK(hipStreamWaitEvent((hipStream_t)waiter, c->ev_fence[0], 0));
  if (c->side && c->side_mode != 0) {     // (nothing is pending there in mode 1: every block has joined already)
    FU_HIP_CHECK(hipEventRecord(c->ev_fence[1], c->side));
    FU_HIP_CHECK(hipStreamWaitEvent((hipStream_t)waiter, c->ev_fence[1], 0));
  }
  return FU_OK;
}

int fu_backward(fu_ctx* c, const float* dlogits, fu_stream stream) {
  FU_REQUIRE(c, "null context");
  if (!(c->last_batch > 0 && c->fwd_training)) {
    set_error("fu_backward: the last fu_forward was not a training forward");
    return FU_ERR_STATE;
  }
  FU_REQUIRE(c->G, "fu_backward: no gradient buffer bound");
  SyncScope sc(c, true);
  UnscaleScope us(c);
  // whole backward: the side stream (weight gradients) is joined once, after the last block
  const int nblk = num_backward_blocks(c);
  for (int b = 0; b < nblk; ++b) FU_TRY(backward_block_impl(c, b, dlogits, (hipStream_t)stream, b == nblk - 1));
  return FU_OK;
}

int fu_block_param_range(const fu_ctx* c, int block, int64_t* flat_offset, int64_t* numel) {
  FU_REQUIRE(c && block >= 0 && block < num_backward_blocks(c), "fu_block_param_range: bad block %d", block);
  int first, count;
  if (block == 0) { first = c->p_outw; count = 2; }
  else if (c->fusion && block == 5) { first = c->fuse[0].p_w; count = 10; }
  else { const Block& K = c->blk[backward_block_index(c, block)]; first = K.first_param; count = K.num_params; }
  const ParamInfo& a = c->params[first];
  const ParamInfo& z = c->params[first + count - 1];
  if (flat_offset) *flat_offset = a.off;
  if (numel) *numel = z.off + z.numel - a.off;
  return FU_OK;
}

int fu_adam_step(fu_ctx* c, double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale,
                 fu_stream stream) {
  FU_REQUIRE(c && c->P && c->G, "fu_adam_step: parameter / gradient buffers not bound");
  FU_REQUIRE(step >= 1, "fu_adam_step: step is 1-based");
  if (!c->adam_m || !c->adam_v) {
    set_error("fu_adam_step: no moment buffers bound (fu_bind_adam_state)");
    return FU_ERR_STATE;
  }
  // fp16: an overflowed gradient map leaves inf / NaN in the gradient buffer -- such a step is skipped and the loss scale
  // backs off (the guard kernels in fu_elementwise.hip); 13 us of the 69 MB gradient read per step, fp16 mode only
  const int* skip = nullptr;
  if (c->prec == PREC_F16) {
    FU_TRY(launch_grad_finite_check(c->G, c->total_params, c->guard, (hipStream_t)stream));
    skip = c->guard;
  }
  FU_TRY(launch_adam(c->P, c->G, c->adam_m, c->adam_v, c->total_params, lr, beta1, beta2, eps, step, grad_scale,
                     (hipStream_t)stream, skip));
  if (skip) FU_TRY(launch_guard_book(c->guard, (hipStream_t)stream));
  c->packed_dirty = true;
  return FU_OK;
}

int fu_fp16_guard_state(fu_ctx* c, int64_t* skipped_steps, int32_t* backoff_exponent) {
  FU_REQUIRE(c, "null context");
  int h[4] = {0, 0, 0, 0};
  if (c->prec == PREC_F16) FU_HIP_CHECK(hipMemcpy(h, c->guard, sizeof(h), hipMemcpyDeviceToHost));   // synchronises
  if (skipped_steps) *skipped_steps = h[1];
  if (backoff_exponent) *backoff_exponent = h[2];
  return FU_OK;
}

int fu_adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale, float out[7]) {
  FU_REQUIRE(out && step >= 1, "fu_adam_scalars: bad argument");
  adam_scalars(lr, beta1, beta2, eps, step, grad_scale, out);
  return FU_OK;
}

int fu_adam_step_dev(fu_ctx* c, const float* scalars_dev, fu_stream stream) {
  FU_REQUIRE(c && c->P && c->G && scalars_dev, "fu_adam_step_dev: parameter / gradient buffers not bound, or null scalars");
  if (!c->adam_m || !c->adam_v) {
    set_error("fu_adam_step_dev: no moment buffers bound (fu_bind_adam_state)");
    return FU_ERR_STATE;
  }
  const int* skip = nullptr;
  if (c->prec == PREC_F16) {
    FU_TRY(launch_grad_finite_check(c->G, c->total_params, c->guard, (hipStream_t)stream));
    skip = c->guard;
  }
  FU_TRY(launch_adam_dev(c->P, c->G, c->adam_m, c->adam_v, c->total_params, scalars_dev, (hipStream_t)stream, skip));
  if (skip) FU_TRY(launch_guard_book(c->guard, (hipStream_t)stream));
  c->packed_dirty = true;
  return FU_OK;
}

int fu_adam_state(fu_ctx* c, float** exp_avg, float** exp_avg_sq) {
  FU_REQUIRE(c, "null context");
  if (exp_avg) *exp_avg = c->adam_m;
  if (exp_avg_sq) *exp_avg_sq = c->adam_v;
  return FU_OK;
}

int fu_zero_grads(fu_ctx* c, fu_stream stream) {
  FU_REQUIRE(c && c->G, "fu_zero_grads: no gradient buffer bound");
  FU_HIP_CHECK(hipMemsetAsync(c->G, 0, c->total_params * sizeof(float), (hipStream_t)stream));
  return FU_OK;
}

int64_t fu_workspace_bytes(const fu_ctx* c) { return c ? (int64_t)c->arena.total : 0; }

int fu_flops_per_tile(const fu_ctx* c, double* fwd, double* train) {
  FU_REQUIRE(c, "null context");
  if (fwd) *fwd = conv_flops(const_cast<fu_ctx*>(c), false);
  if (train) *train = conv_flops(const_cast<fu_ctx*>(c), true);
  return FU_OK;
}

int fu_profile_enable(fu_ctx* c, int enable) {
  FU_REQUIRE(c, "null context");
  Profiler& pr = c->prof;
  if (enable) {
    const size_t want = 2 * 8192;
    while (pr.pool.size() < want) {
      hipEvent_t e;
      FU_HIP_CHECK(hipEventCreate(&e));
      pr.pool.push_back(e);
    }
    if (enable != 2) {          // 2 = resume: keep the records of the earlier enabled stretches
      pr.next = 0;
      pr.recs.clear();
      pr.overflow = false;
    }
    pr.on = true;
  } else {
    pr.on = false;
  }
  return FU_OK;
}

int fu_profile_read(fu_ctx* c, int kernel_class, int64_t* launches, double* total_ms, double* total_flops,
                    const char** kernel_name) {
  FU_REQUIRE(c && kernel_class >= 0 && kernel_class < FU_K_NUM, "fu_profile_read: bad class %d", kernel_class);
  FU_HIP_CHECK(hipDeviceSynchronize());
  int64_t n = 0;
  double ms = 0.0, fl = 0.0;
  for (const ProfRec& r : c->prof.recs) {
    if (r.cls != kernel_class) continue;
    float t = 0.f;
    FU_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (kernel_name)
    *kernel_name = kernel_class == FU_K_CONV3X3
                       // 16-bit: the class = every forward / dgrad launch, i.e. k_conv3x3_bf16_rs<8|4> and the
                       // k_conv3x3_bf16_fast<...> instantiations (the common prefix matches them all in a kernel trace)
                       ? (c->prec == PREC_F32 ? "k_conv3x3_f32" : (c->prec == PREC_BF16 ? "k_conv3x3_bf16" : "k_conv3x3_f16"))
                       : (c->prec == PREC_F32 ? "k_wgrad_f32" : (c->prec == PREC_BF16 ? "k_wgrad_bf16" : "k_wgrad_f16"));
  return FU_OK;
}

int fu_stitch_add(fu_ctx* c, int sample, float* canvas, float* weight, int canvas_h, int canvas_w, int h0, int w0,
                  int hE, int wE, fu_stream stream) {
  FU_REQUIRE(c && canvas && weight, "fu_stitch_add: null argument");
  FU_REQUIRE(c->last_batch > 0 && sample >= 0 && sample < c->last_batch, "fu_stitch_add: sample %d not in the last batch",
             sample);
  const int H = c->cfg.height, W = c->cfg.width;
  const int dh = hE - h0, dw = wE - w0;
  FU_REQUIRE(h0 >= 0 && w0 >= 0 && dh >= 0 && dw >= 0 && hE <= canvas_h && wE <= canvas_w && dh <= H && dw <= W,
             "fu_stitch_add: crop [%d:%d, %d:%d] does not fit canvas %dx%d / tile %dx%d", h0, hE, w0, wE, canvas_h,
             canvas_w, H, W);
  if (dh == 0 || dw == 0) return FU_OK;
  const float* lg = c->logits + (int64_t)sample * H * W * c->cfg.n_classes;
  return launch_stitch_add(lg, c->cfg.n_classes, W, canvas, weight, canvas_w, h0, w0, dh, dw, (hipStream_t)stream);
}

int fu_stitch_finalize(float* canvas, const float* weight, int n_classes, int canvas_h, int canvas_w,
                       int64_t* argmax_out, fu_stream stream) {
  FU_REQUIRE(canvas && weight && n_classes >= 1 && n_classes <= HEAD_MAX_CLS, "fu_stitch_finalize: bad argument");
  return launch_stitch_finalize(canvas, weight, n_classes, (int64_t)canvas_h * canvas_w, argmax_out,
                                (hipStream_t)stream);
}

int fu_augment(const float* image, const int64_t* target, float* image_out, int64_t* target_out, const int32_t* flags,
               const float* angles_deg, int B, int C, int H, int W, int64_t target_fill, fu_stream stream) {
  FU_REQUIRE(image && image_out && flags && angles_deg, "fu_augment: null argument");
  FU_REQUIRE((target == nullptr) == (target_out == nullptr), "fu_augment: target and target_out go together");
  FU_REQUIRE(image != image_out && (!target || target != target_out), "fu_augment: in-place operation is not supported");
  return launch_augment(image, target, image_out, target_out, flags, angles_deg, B, C, H, W, target_fill,
                        (hipStream_t)stream);
}

int fu_assemble_tiles(const float* const* srcs, const int32_t* src_channels, int n_src, int B, int H, int W,
                      const int32_t* valid_h, const int32_t* valid_w, int norm_mode, const float* global_mean,
                      const float* global_std, float pad_value, float* out, float* mean_out, float* std_out,
                      fu_stream stream) {
  FU_REQUIRE(srcs && src_channels && out && B >= 1 && H >= 1 && W >= 1, "fu_assemble_tiles: bad argument");
  return launch_assemble_tiles(srcs, src_channels, n_src, B, H, W, valid_h, valid_w, norm_mode, global_mean, global_std,
                               pad_value, out, mean_out, std_out, (hipStream_t)stream);
}

int fu_resize_lanczos4_tiles(const float* windows, int B, int C, int win_h, int win_w, const int32_t* iy, const float* wy,
                             const int32_t* ix, const float* wx, int tile_h, int tile_w, int scale_mode, float* out,
                             fu_stream stream) {
  FU_REQUIRE(windows && iy && wy && ix && wx && out && B >= 1 && C >= 1, "fu_resize_lanczos4_tiles: bad argument");
  return launch_resize_lanczos4_tiles(windows, B, C, win_h, win_w, iy, wy, ix, wx, tile_h, tile_w, scale_mode, out,
                                      (hipStream_t)stream);
}

// ---- data-parallel collective behind the C ABI (fu_allreduce_begin / wait) ----------------------------------
// RCCL is resolved at run time (dlopen): the library has no link-time dependency on it, a process that never calls
// fu_dp_init never loads it, and inside a torch process the copy torch already loaded is the one that is used.
}  // extern "C"
#include <dlfcn.h>
struct Dp {
  void* lib = nullptr;
  void* comm = nullptr;
  hipStream_t stream = nullptr;       // communication stream: all-reduces run here, beside the backward kernels
  hipEvent_t ev_ready = nullptr, ev_done = nullptr, ev_side = nullptr;
  int rank = 0, world = 1;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, fu_dp_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
namespace {
constexpr int kNcclFloat32 = 7, kNcclInt64 = 4, kNcclSum = 0;      // ncclDataType_t / ncclRedOp_t values of rccl.h
void* rccl_handle() {
  static void* h = nullptr;
  if (h) return h;
  for (const char* n : {"librccl.so", "librccl.so.1"}) {           // a copy that is already in the process (torch's) first
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) return h;
  }
  for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) return h;
  }
  return nullptr;
}
template <typename F> bool sym(void* lib, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(lib, name));
  return *out != nullptr;
}
int dp_load(Dp* d) {
  d->lib = rccl_handle();
  if (!d->lib) { set_error("fu_dp: librccl.so not found (%s)", dlerror()); return FU_ERR_UNSUPPORTED; }
  const bool ok = sym(d->lib, "ncclGetUniqueId", &d->GetUniqueId) && sym(d->lib, "ncclCommInitRank", &d->CommInitRank) &&
                  sym(d->lib, "ncclCommDestroy", &d->CommDestroy) && sym(d->lib, "ncclAllReduce", &d->AllReduce) &&
                  sym(d->lib, "ncclBroadcast", &d->Broadcast) && sym(d->lib, "ncclGetErrorString", &d->GetErrorString);
  if (!ok) { set_error("fu_dp: librccl.so lacks an expected symbol"); return FU_ERR_UNSUPPORTED; }
  return 0;
}
#define FU_NCCL(d, expr)                                                                      \
  do {                                                                                        \
    const int _r = (expr);                                                                    \
    if (_r != 0) { set_error("%s failed: %s", #expr, (d)->GetErrorString(_r)); return FU_ERR_HIP; } \
  } while (0)
}  // namespace
extern "C" {

int fu_dp_unique_id(fu_dp_id* id) {
  FU_REQUIRE(id, "fu_dp_unique_id: null argument");
  Dp d;
  FU_TRY(dp_load(&d));
  FU_NCCL(&d, d.GetUniqueId(id));
  return FU_OK;
}

int fu_dp_init(fu_ctx* c, const fu_dp_id* id, int rank, int world) {
  FU_REQUIRE(c && id && world >= 1 && rank >= 0 && rank < world, "fu_dp_init: bad argument (rank %d of %d)", rank, world);
  FU_REQUIRE(c->dp == nullptr, "fu_dp_init: the context already has a communicator");
  FU_HIP_CHECK(hipSetDevice(c->cfg.device));
  Dp* d = new (std::nothrow) Dp();
  FU_REQUIRE(d, "out of host memory");
  int st = dp_load(d);
  if (st == 0) {
    const int r = d->CommInitRank(&d->comm, world, *id, rank);
    if (r != 0) { set_error("ncclCommInitRank failed: %s", d->GetErrorString(r)); st = FU_ERR_HIP; }
  }
  if (st == 0 && (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess ||
                  hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&d->ev_side, hipEventDisableTiming) != hipSuccess)) {
    set_error("fu_dp_init: stream / event creation failed");
    st = FU_ERR_HIP;
  }
  if (st != 0) {
    if (d->comm) (void)d->CommDestroy(d->comm);
    delete d;
    return st;
  }
  d->rank = rank; d->world = world;
  c->dp = d;
  return FU_OK;
}

int fu_dp_destroy(fu_ctx* c) {
  if (!c || !c->dp) return FU_OK;
  Dp* d = c->dp;
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->comm) (void)d->CommDestroy(d->comm);
  if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
  if (d->ev_done) (void)hipEventDestroy(d->ev_done);
  if (d->ev_side) (void)hipEventDestroy(d->ev_side);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  c->dp = nullptr;
  return FU_OK;
}

int fu_dp_broadcast_state(fu_ctx* c, fu_stream stream) {
  FU_REQUIRE(c && c->dp, "fu_dp_broadcast_state: no communicator (fu_dp_init)");
  FU_REQUIRE(c->P && c->RM && c->RV && c->NBT, "fu_dp_broadcast_state: buffers not bound (fu_bind_buffers)");
  Dp* d = c->dp;
  hipStream_t s = (hipStream_t)stream;
  FU_NCCL(d, d->Broadcast(c->P, c->P, (size_t)c->total_params, kNcclFloat32, 0, d->comm, s));
  FU_NCCL(d, d->Broadcast(c->RM, c->RM, (size_t)c->total_bn, kNcclFloat32, 0, d->comm, s));
  FU_NCCL(d, d->Broadcast(c->RV, c->RV, (size_t)c->total_bn, kNcclFloat32, 0, d->comm, s));
  FU_NCCL(d, d->Broadcast(c->NBT, c->NBT, c->bns.size(), kNcclInt64, 0, d->comm, s));
  c->packed_dirty = true;
  return FU_OK;
}

int fu_allreduce_begin(fu_ctx* c, int64_t flat_offset, int64_t numel, fu_stream stream) {
  FU_REQUIRE(c && c->dp, "fu_allreduce_begin: no communicator (fu_dp_init)");
  FU_REQUIRE(c->G, "fu_allreduce_begin: no gradient buffer bound");
  FU_REQUIRE(flat_offset >= 0 && numel >= 0 && flat_offset + numel <= c->total_params,
             "fu_allreduce_begin: range [%lld, +%lld) outside the %lld gradient elements", (long long)flat_offset,
             (long long)numel, (long long)c->total_params);
  if (numel == 0) return FU_OK;
  Dp* d = c->dp;
  // the bucket's gradients are final in `stream` order (the caller joined the side stream: fu_backward_join / a joining
  // fu_backward_block); the all-reduce starts behind them on the communication stream, the caller's stream moves on
  FU_HIP_CHECK(hipEventRecord(d->ev_ready, (hipStream_t)stream));
  FU_HIP_CHECK(hipStreamWaitEvent(d->stream, d->ev_ready, 0));
  if (c->side && c->side_mode == 2) {      // caller-join mode: the weight-gradient chain may still be behind; only the
    FU_HIP_CHECK(hipEventRecord(d->ev_side, c->side));             // communication stream waits for it, `stream` moves on
    FU_HIP_CHECK(hipStreamWaitEvent(d->stream, d->ev_side, 0));
  }
  float* g = c->G + flat_offset;
  FU_NCCL(d, d->AllReduce(g, g, (size_t)numel, kNcclFloat32, kNcclSum, d->comm, d->stream));
  return FU_OK;
}

int fu_allreduce_wait(fu_ctx* c, fu_stream stream) {
  FU_REQUIRE(c && c->dp, "fu_allreduce_wait: no communicator (fu_dp_init)");
  Dp* d = c->dp;
  FU_HIP_CHECK(hipEventRecord(d->ev_done, d->stream));
  FU_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, d->ev_done, 0));
  return FU_OK;
}

// ---- single operators --------------------------------------------------------------------------------
int fu_elem_size(int precision) { return precision == FU_F32 ? 4 : 2; }   /* FU_BF16 and FU_F16: 2 */

static int prec_of(int precision, Prec* p) {
  FU_REQUIRE(precision == FU_F32 || precision == FU_BF16 || precision == FU_F16, "unknown precision %d", precision);
  *p = precision == FU_F32 ? PREC_F32 : (precision == FU_BF16 ? PREC_BF16 : PREC_F16);
  return 0;
}

int fu_op_nchw_to_nhwc(int precision, const float* src, void* dst, int B, int C, int H, int W, int c_pad,
                       fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  return launch_nchw_to_nhwc(p, src, dst, B, C, H, W, c_pad, (hipStream_t)stream);
}
int fu_op_nhwc_to_nchw(int precision, const void* src, float* dst, int B, int C, int H, int W, int c_pad,
                       fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  return launch_nhwc_to_nchw(p, src, dst, B, C, H, W, c_pad, (hipStream_t)stream);
}

namespace {
struct TmpBuf {
  void* p = nullptr;
  ~TmpBuf() { if (p) (void)hipFree(p); }
  int get(size_t bytes) { FU_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16)); return 0; }
};
__global__ void k_stats_collapse(const float* part, int nTiles, int C, float* sum, float* sq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0, q = 0;
  for (int t = 0; t < nTiles; ++t) { s += part[((int64_t)t * C + c) * 2]; q += part[((int64_t)t * C + c) * 2 + 1]; }
  sum[c] = (float)s; sq[c] = (float)q;
}
}  // namespace

int fu_op_conv3x3_fwd(int precision, const void* src0, int C0, const float* bn_a0, const float* bn_b0,
                      const void* src1, int C1, const float* w_oihw, const float* bias, void* y, int Cout, int B, int H,
                      int W, float* stats_sum, float* stats_sqsum, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C0 + (src1 ? C1 : 0);
  TmpBuf wf, st;
  FU_TRY(wf.get(conv3x3_pack_elems(p, Cin, Cout) * fu_elem_size(precision)));
  FU_TRY(launch_pack_conv3x3(p, w_oihw, Cout, Cin, Cin, wf.p, nullptr, s));
  const bool want_stats = stats_sum && stats_sqsum;
  if (want_stats) FU_TRY(st.get((size_t)conv3x3_num_stat_tiles(p, B, H, W) * Cout * 2 * sizeof(float)));
  ConvIn in{src0, C0, bn_a0, bn_b0, src1, src1 ? C1 : 0};
  int nt = 0;
  FU_TRY(launch_conv3x3(p, in, wf.p, bias, y, Cout, nullptr, 0, want_stats ? (float*)st.p : nullptr, &nt, B, H, W, s));
  if (want_stats)
    hipLaunchKernelGGL(k_stats_collapse, dim3(ceil_div(Cout, 64)), dim3(64), 0, s, (const float*)st.p, nt, Cout,
                       stats_sum, stats_sqsum);
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_op_conv3x3_dgrad(int precision, const void* dy, int Cout, const float* w_oihw, void* dx0, int C0, void* dx1,
                        int C1, int B, int H, int W, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C0 + (dx1 ? C1 : 0);
  TmpBuf wd;
  FU_TRY(wd.get(conv3x3_pack_elems(p, Cin, Cout) * fu_elem_size(precision)));
  FU_TRY(launch_pack_conv3x3(p, w_oihw, Cout, Cin, Cin, nullptr, wd.p, s));
  ConvIn in{dy, Cout, nullptr, nullptr, nullptr, 0};
  FU_TRY(launch_conv3x3(p, in, wd.p, nullptr, dx0, C0, dx1, dx1 ? C1 : 0, nullptr, nullptr, B, H, W, s));
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_op_conv3x3_wgrad(int precision, const void* src0, int C0, const float* bn_a0, const float* bn_b0,
                        const void* src1, int C1, const void* dy, int Cout, float* dw_oihw, int B, int H, int W,
                        fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C0 + (src1 ? C1 : 0);
  TmpBuf slab;
  FU_TRY(slab.get(conv3x3_wgrad_slab_elems(p, Cin, Cout, B, H, W) * sizeof(float)));
  ConvIn in{src0, C0, bn_a0, bn_b0, src1, src1 ? C1 : 0};
  FU_TRY(launch_conv3x3_wgrad(p, in, dy, Cout, (float*)slab.p, dw_oihw, Cin, nullptr, 0, nullptr, B, H, W, s));
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_op_maxpool2(int precision, const void* src, const float* bn_a, const float* bn_b, void* dst, int B, int H,
                   int W, int C, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  return launch_maxpool2(p, src, bn_a, bn_b, dst, B, H, W, C, (hipStream_t)stream);
}

int fu_op_upsample2(int precision, const void* src, const float* bn_a, const float* bn_b, void* dst, int B, int H,
                    int W, int C, int outH, int outW, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  fu_ctx tmp;  // only used as an allocation list for the tables
  UpTables t;
  int st = build_up_tables(&tmp, H, W, &t);
  if (st == 0) st = launch_upsample2(p, src, bn_a, bn_b, dst, B, H, W, C, outH, outW, t, (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  for (void* q : tmp.extra_allocs) (void)hipFree(q);
  return st;
}

// ---- op-level test hooks for the code that only runs in the benched dispatch (fused BatchNorm-backward sums) ------------
namespace {
// [nTiles][C][2] partial rows -> per-channel sums, fp64 accumulation in tile order
int collapse_partials(const float* part, int nTiles, int C, float* s1, float* s2, hipStream_t s) {
  hipLaunchKernelGGL(k_stats_collapse, dim3(ceil_div(C, 64)), dim3(64), 0, s, part, nTiles, C, s1, s2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("collapse launch failed: %s", hipGetErrorString(e)); return FU_ERR_HIP; }
  return 0;
}
}  // namespace

int fu_op_conv3x3_dgrad_bnsums(int precision, const void* dy, int Cout, const float* w_oihw, void* dx, int C0,
                               const void* y, const float* bn_a, const float* bn_b, const float* mean,
                               const float* invstd, float* sum_gm, float* sum_gmx, int B, int H, int W,
                               fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  FU_REQUIRE(p != PREC_F32, "fu_op_conv3x3_dgrad_bnsums: 16-bit precisions only (fp32 keeps the separate reduce pass)");
  FU_REQUIRE(dy && w_oihw && dx && y && bn_a && bn_b && mean && invstd && sum_gm && sum_gmx, "fu_op_conv3x3_dgrad_bnsums: null argument");
  hipStream_t s = (hipStream_t)stream;
  TmpBuf wd, part;
  FU_TRY(wd.get(conv3x3_pack_elems(p, C0, Cout) * fu_elem_size(precision)));
  FU_TRY(launch_pack_conv3x3(p, w_oihw, Cout, C0, C0, nullptr, wd.p, s));
  const int64_t cap = (int64_t)B * ceil_div(H, 16) * ceil_div(W, 16) * C0 * 2;
  FU_TRY(part.get((size_t)cap * sizeof(float)));
  int tiles = 0;
  BnbFuse f;
  f.y = y; f.a = bn_a; f.b = bn_b; f.mean = mean; f.invstd = invstd;
  f.part = (float*)part.p; f.max_elems = cap; f.tiles_out = &tiles;
  ConvIn in{dy, Cout, nullptr, nullptr, nullptr, 0};
  in.opt.bnb = &f;
  FU_TRY(launch_conv3x3(p, in, wd.p, nullptr, dx, C0, nullptr, 0, nullptr, nullptr, B, H, W, s));
  if (tiles <= 0) {
    set_error("fu_op_conv3x3_dgrad_bnsums: the kernel that ran does not emit the sums for this shape / dispatch");
    return FU_ERR_UNSUPPORTED;
  }
  FU_TRY(perturb_bnb((float*)part.p, tiles, C0, s));
  FU_TRY(collapse_partials((const float*)part.p, tiles, C0, sum_gm, sum_gmx, s));
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_op_head_bwd(int precision, const float* dlogits_nhwc, const void* y, const float* bn_a, const float* bn_b,
                   const float* w, int C, int ncls, int64_t npix, void* g, float* dw, float* db, const float* mean,
                   const float* invstd, float* sum_gm, float* sum_gmx, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  FU_REQUIRE(dlogits_nhwc && y && w && g && dw && db, "fu_op_head_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  TmpBuf part, bpart;
  FU_TRY(part.get((size_t)head_bwd_partial_elems(C, ncls) * sizeof(float)));
  const bool want = mean && invstd && sum_gm && sum_gmx;
  const int64_t cap = (int64_t)2048 * C * 2;
  int tiles = 0;
  BnbFuse f;
  if (want) {
    FU_TRY(bpart.get((size_t)cap * sizeof(float)));
    f.y = y; f.a = bn_a; f.b = bn_b; f.mean = mean; f.invstd = invstd;
    f.part = (float*)bpart.p; f.max_elems = cap; f.tiles_out = &tiles;
  }
  FU_TRY(launch_head_bwd(p, dlogits_nhwc, y, bn_a, bn_b, w, C, ncls, npix, g, (float*)part.p, dw, db, s,
                         want ? &f : nullptr));
  if (want) {
    if (tiles <= 0) {
      set_error("fu_op_head_bwd: the head-backward kernel does not emit the sums in this precision");
      return FU_ERR_UNSUPPORTED;
    }
    FU_TRY(perturb_bnb((float*)bpart.p, tiles, C, s));
    FU_TRY(collapse_partials((const float*)bpart.p, tiles, C, sum_gm, sum_gmx, s));
  }
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_op_bn_bwd(int precision, void* g, const void* y, int C, int B, int H, int W, const float* bn_a,
                 const float* bn_b, const float* mean, const float* invstd, const void* g_pool, float* dgamma,
                 float* dbeta, fu_stream stream) {
  Prec p; FU_TRY(prec_of(precision, &p));
  FU_REQUIRE(g && y && bn_a && bn_b && mean && invstd && dgamma && dbeta, "fu_op_bn_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * H * W;
  TmpBuf part, coef, dbp, scr;
  FU_TRY(part.get((size_t)bn_bwd_partial_elems(C, npix) * sizeof(float)));
  FU_TRY(coef.get((size_t)C * 2 * sizeof(float)));
  FU_TRY(dbp.get((size_t)bn_bwd_partial_elems(C, npix) * sizeof(float)));
  FU_TRY(scr.get((size_t)reduce_scratch_elems(std::max(C, 64)) * sizeof(double)));
  int ndb = 0;
  FU_TRY(launch_bn_bwd(p, g, y, C, npix, bn_a, bn_b, mean, invstd, nullptr, dgamma, dbeta, (float*)part.p,
                       (float*)coef.p, (float*)dbp.p, &ndb, (double*)scr.p, s, g_pool, B, H, W, 0));
  FU_HIP_CHECK(hipStreamSynchronize(s));
  return FU_OK;
}

int fu_test_get_buffer(fu_ctx* c, int block, int which, void** ptr, int64_t* elems) {
  FU_REQUIRE(c && ptr && elems, "fu_test_get_buffer: null argument");
  FU_REQUIRE(block >= 0 && block < c->nb, "fu_test_get_buffer: block %d outside 0..%d", block, c->nb - 1);
  const Block& K = c->blk[block];
  const int B = c->cfg.max_batch;
  auto act = [&](int level, int C) { return (int64_t)B * c->Hs[level] * c->Ws[level] * C; };
  switch (which) {
    case 0: *ptr = K.c[0].y; *elems = act(K.c[0].level, K.c[0].cout); break;
    case 1: *ptr = K.c[0].gy; *elems = act(K.c[0].level, K.c[0].cout); break;
    case 2: *ptr = K.c[1].y; *elems = act(K.c[1].level, K.c[1].cout); break;
    case 3: *ptr = K.c[1].gy; *elems = act(K.c[1].level, K.c[1].cout); break;
    case 4: *ptr = K.g_pooled; *elems = K.kind == BK_DOWN ? act(K.level, K.c[0].cin_real) : 0; break;
    case 5: *ptr = K.g_up; *elems = K.kind == BK_UP ? act(K.level, K.c[0].cin_real - c->ch[K.skip]) : 0; break;
    default: set_error("fu_test_get_buffer: which must be 0..5"); return FU_ERR_INVALID;
  }
  return FU_OK;
}

}  // extern "C"

extern "C" void fu_test_bnb_separate(int on) { g_bnb_separate = on ? 1 : 0; }
extern "C" void fu_test_head_store_g(int on) { g_head_store_g = on ? 1 : 0; }
extern "C" void fu_test_perturb_bnb_sums(float factor) { g_test_perturb_bnb = factor; }
