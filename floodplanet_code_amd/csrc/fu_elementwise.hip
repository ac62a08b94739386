// Memory-bound kernels of the UNet training path (gfx950): layout conversion, BatchNorm statistics
// finalisation and backward, max-pool, bilinear x2 resize, 1x1 head + cross entropy, Adam.
// All of them are HBM-bound: 16-byte vector accesses along the NHWC channel dimension, fp32 math,
// deterministic two-level reductions (per-block partials -> fixed-order finalisation), wave64 shuffles.
#include "fu_common.h"

#include <stdarg.h>

namespace fu {

thread_local const SyncDesc* g_sync = nullptr;
thread_local const float* g_grad_unscale = nullptr;

int sync_sum_over_ranks(void* payload, int64_t n_elems, bool is_double, hipStream_t s) {
  const SyncDesc* d = g_sync;
  if (!d || !d->hook || d->world <= 1) return 0;
  const size_t bytes = (size_t)n_elems * (is_double ? 8 : 4);
  FU_REQUIRE((int64_t)bytes <= d->xbytes, "exact sync: exchange buffer too small (%zu > %lld bytes)", bytes,
             (long long)d->xbytes);
  FU_HIP_CHECK(hipMemcpyAsync(d->xbuf, payload, bytes, hipMemcpyDeviceToDevice, s));
  if (d->hook(d->user, n_elems, is_double ? 1 : 0) != 0) {
    set_error("exact sync: the all-reduce hook failed");
    return 3;   // FU_ERR_STATE
  }
  FU_HIP_CHECK(hipMemcpyAsync(payload, d->xbuf, bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// error message storage
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

#define FU_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

static inline int grid_for(int64_t work, int block, int cap = 8192) {
  int64_t g = ceil_div64(work, block);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// NCHW fp32 <-> NHWC T
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, T* __restrict__ dst, int C, int HW, int cpad,
                               int64_t total, int srcC) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cpad);
    const int64_t bp = idx / cpad;
    const int p = (int)(bp % HW);
    const int64_t b = bp / HW;
    const float v = (c < C) ? src[(b * srcC + c) * HW + p] : 0.f;   // srcC: channels per sample of the source
    ElemIO<T>::store1(dst + idx, v);
  }
}

// 16-bit destinations with c_pad % 8 == 0 (the training path): one thread per (pixel, channel octet) -- eight plane reads
// that are coalesced across the threads of a wave, one 16-byte store; 32-bit indices.  (The flat kernel above decodes a
// 64-bit index twice per ELEMENT and reads with a stride of H*W floats between neighbouring threads.)
template <typename T>
__global__ __launch_bounds__(256) void k_nchw_to_nhwc_v8(const float* __restrict__ src, T* __restrict__ dst, int C, int HW,
                                                          int cpad, int srcC) {
  const int p = blockIdx.x * 256 + threadIdx.x;        // pixel inside the sample
  const int o = blockIdx.y, b = blockIdx.z;            // channel octet, sample
  if (p >= HW) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j;
    v[j] = c < C ? src[((size_t)b * srcC + c) * HW + p] : 0.f;
  }
  VecIO<T>::store(dst + ((size_t)b * HW + p) * cpad + o * 8, v);
}

template <typename T>
__global__ void k_nhwc_to_nchw(const T* __restrict__ src, float* __restrict__ dst, int C, int HW, int cpad,
                               int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(idx % HW);
    const int64_t bc = idx / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    dst[idx] = ElemIO<T>::load1(src + (b * HW + p) * cpad + c);
  }
}

int launch_nchw_to_nhwc(Prec p, const float* src, void* dst, int B, int C, int H, int W, int c_pad, hipStream_t s,
                        int src_channels, int src_channel_offset) {
  // C channels starting at src_channel_offset of an NCHW tensor with src_channels per sample (0 = C: the whole tensor)
  const int64_t total = (int64_t)B * H * W * c_pad;
  const int g = grid_for(total, 256);
  const int srcC = src_channels > 0 ? src_channels : C;
  src += (int64_t)src_channel_offset * H * W;
  if (p != PREC_F32 && c_pad % 8 == 0 && B <= 65535 && c_pad / 8 <= 65535) {
    const dim3 g8((unsigned)ceil_div(H * W, 256), (unsigned)(c_pad / 8), (unsigned)B);
    if (p == PREC_BF16)
      hipLaunchKernelGGL(k_nchw_to_nhwc_v8<bf16_t>, g8, dim3(256), 0, s, src, (bf16_t*)dst, C, H * W, c_pad, srcC);
    else
      hipLaunchKernelGGL(k_nchw_to_nhwc_v8<f16_t>, g8, dim3(256), 0, s, src, (f16_t*)dst, C, H * W, c_pad, srcC);
    FU_LAUNCH_CHECK();
    return 0;
  }
  if (p == PREC_F32)
    hipLaunchKernelGGL(k_nchw_to_nhwc<float>, dim3(g), dim3(256), 0, s, src, (float*)dst, C, H * W, c_pad, total, srcC);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_nchw_to_nhwc<bf16_t>, dim3(g), dim3(256), 0, s, src, (bf16_t*)dst, C, H * W, c_pad, total,
                       srcC);
  else
    hipLaunchKernelGGL(k_nchw_to_nhwc<f16_t>, dim3(g), dim3(256), 0, s, src, (f16_t*)dst, C, H * W, c_pad, total,
                       srcC);
  FU_LAUNCH_CHECK();
  return 0;
}

// The same conversion from SEVERAL NCHW sources taken side by side along the channel axis (ef_model.py:28-44: the image and
// the auxiliary maps; stacked sensors): destination channel c is channel ch_off + c of the virtual concatenation.  The
// torch.concat copy of the reference disappears into the layout conversion the first conv needs anyway.  One thread per
// (pixel, vector of V channels); plane reads are coalesced across the threads of a wave, one 16-byte store.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_nchw_to_nhwc(SrcList S, T* __restrict__ dst, int C, int HW, int cpad,
                                                              int ch_off) {
  constexpr int V = VecIO<T>::V;
  const int p = blockIdx.x * 256 + threadIdx.x;        // pixel inside the sample
  const int o = blockIdx.y, b = blockIdx.z;            // channel vector, sample
  if (p >= HW) return;
  float v[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = o * V + j;
    v[j] = 0.f;
    if (c < C) {
      const int cg = ch_off + c;
      int si = 0;
      for (int k = 1; k < S.n; ++k) si = cg >= S.coff[k] ? k : si;
      v[j] = S.p[si][((size_t)b * S.c[si] + (cg - S.coff[si])) * HW + p];
    }
  }
  VecIO<T>::store(dst + ((size_t)b * HW + p) * cpad + o * V, v);
}

int launch_gather_nchw_to_nhwc(Prec p, const SrcList& S, void* dst, int B, int C, int H, int W, int c_pad, int ch_off,
                               hipStream_t s) {
  const int V = p == PREC_F32 ? 4 : 8;
  FU_REQUIRE(c_pad % V == 0 && B <= 65535 && c_pad / V <= 65535, "gather_nchw_to_nhwc: bad geometry (c_pad %d, batch %d)", c_pad, B);
  FU_REQUIRE(ch_off >= 0 && ch_off + C <= S.coff[S.n], "gather_nchw_to_nhwc: channels [%d, %d) outside the %d source channels",
             ch_off, ch_off + C, S.coff[S.n]);
  const dim3 g((unsigned)ceil_div(H * W, 256), (unsigned)(c_pad / V), (unsigned)B);
  if (p == PREC_F32) hipLaunchKernelGGL(k_gather_nchw_to_nhwc<float>, g, dim3(256), 0, s, S, (float*)dst, C, H * W, c_pad, ch_off);
  else if (p == PREC_BF16) hipLaunchKernelGGL(k_gather_nchw_to_nhwc<bf16_t>, g, dim3(256), 0, s, S, (bf16_t*)dst, C, H * W, c_pad, ch_off);
  else hipLaunchKernelGGL(k_gather_nchw_to_nhwc<f16_t>, g, dim3(256), 0, s, S, (f16_t*)dst, C, H * W, c_pad, ch_off);
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_nhwc_to_nchw(Prec p, const void* src, float* dst, int B, int C, int H, int W, int c_pad, hipStream_t s) {
  const int64_t total = (int64_t)B * C * H * W;
  const int g = grid_for(total, 256);
  if (p == PREC_F32)
    hipLaunchKernelGGL(k_nhwc_to_nchw<float>, dim3(g), dim3(256), 0, s, (const float*)src, dst, C, H * W, c_pad, total);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_nhwc_to_nchw<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)src, dst, C, H * W, c_pad,
                       total);
  else
    hipLaunchKernelGGL(k_nhwc_to_nchw<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)src, dst, C, H * W, c_pad,
                       total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// two-level per-channel reduction of partials [nPart][C][NV] (fp32) -> [G][C][NV] (fp64)
// ------------------------------------------------------------------------------------------------
static constexpr int RED_GROUPS = 32;

template <int NV>
__global__ void k_partials_reduce(const float* __restrict__ part, double* __restrict__ out, int nPart, int C,
                                  int perGroup) {
  // block: 32 channels x 8 partial lanes; grid (ceil(C/32), G)
  __shared__ double sm[8][32][NV];
  const int cl = threadIdx.x & 31, j = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int g = blockIdx.y;
  const int t0 = g * perGroup;
  const int t1 = min(nPart, t0 + perGroup);
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int t = t0 + j; t < t1; t += 8) {
      const float* q = part + ((int64_t)t * C + c) * NV;
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += (double)q[v];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) sm[j][cl][v] = acc[v];
  __syncthreads();
  if (j == 0 && c < C) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      double s = 0.0;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) s += sm[jj][cl][v];
      out[((int64_t)g * C + c) * NV + v] = s;
    }
  }
}

// scratch for the fp64 second level lives in a static device buffer per call site (passed in)
template <int NV>
static int reduce_partials(const float* part, double* out, int nPart, int C, hipStream_t s, int* groups_out) {
  int G = nPart < RED_GROUPS ? nPart : RED_GROUPS;
  if (G < 1) G = 1;
  const int perGroup = ceil_div(nPart, G);
  G = ceil_div(nPart, perGroup);
  if (G < 1) G = 1;
  hipLaunchKernelGGL(k_partials_reduce<NV>, dim3(ceil_div(C, 32), G), dim3(256), 0, s, part, out, nPart, C, perGroup);
  FU_LAUNCH_CHECK();
  *groups_out = G;
  return 0;
}

// fixed-shape xor tree over the 32 lanes of a half wave (deterministic); every lane ends with the total
__device__ __forceinline__ double half_wave_sum(double v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm forward statistics -> coefficients
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_finalize(
    const double* __restrict__ dpart, int G, int C, double count, const float* __restrict__ conv_bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ a_o, float* __restrict__ b_o,
    float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  // 8 channels per block, 32 lanes per channel: lane g fetches group g's partial (one memory latency instead of a
  // G-long dependent chain), then a fixed xor tree
  const int g = threadIdx.x & 31;
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool ok = c < C && g < G;
  double S = ok ? dpart[((int64_t)g * C + c) * 2 + 0] : 0.0;
  double Q = ok ? dpart[((int64_t)g * C + c) * 2 + 1] : 0.0;
  S = half_wave_sum(S);
  Q = half_wave_sum(Q);
  if (c >= C || g != 0) return;
  const double m0 = S / count;
  double var = Q / count - m0 * m0;
  if (var < 0.0) var = 0.0;
  const double mean = m0 + (conv_bias ? (double)conv_bias[c] : 0.0);
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  const float a = gamma[c] * invstd;
  mean_o[c] = meanf;
  invstd_o[c] = invstd;
  a_o[c] = a;
  b_o[c] = beta[c] - meanf * a;
  if (rmean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * meanf;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
  if (nbt && c == 0) *nbt += 1;
}

// ------------------------------------------------------------------------------------------------
// One launch per BatchNorm instead of two (k_partials_reduce + finalize): per-CHANNEL parallelism.  Channels are
// independent, so a block that owns 4 channels can reduce ALL their tile partials ([nPart][C][2] fp32, 32 contiguous
// bytes per tile and block) and finalise them itself -- no second level across blocks, no inter-block hand-off.  128 tile
// lanes x 2 float4 columns; fp64 accumulation; lanes meet in LDS and are summed in a fixed order (deterministic).
// MODE 0: forward statistics -> mean / invstd / a / b (+ running statistics); MODE 1: backward sums -> dgamma / dbeta / coef.
// (The exact data-parallel mode exchanges the partials between the two levels and keeps the two-launch form.)
// ------------------------------------------------------------------------------------------------
struct BnFwdOut {
  const float* conv_bias; const float* gamma; const float* beta; float eps, momentum;
  float *mean, *invstd, *a, *b, *rmean, *rvar; int64_t* nbt;
};
struct BnBwdOut { const float* unscale; float *dgamma, *dbeta, *coef; };

template <int MODE, typename OUT>
__global__ __launch_bounds__(256) void k_bn_stats_fused(const float* __restrict__ part, int nPart, int C, double count,
                                                        OUT o) {
  __shared__ double sm[128][8];
  const int col = threadIdx.x & 1, tl = threadIdx.x >> 1;
  const int c0 = blockIdx.x * 4;                        // C % 4 == 0
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const float* base = part + (size_t)c0 * 2 + col * 4;
  int t = tl;
  for (; t + 3 * 128 < nPart; t += 4 * 128) {           // four independent 16-byte loads in flight per thread
    const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)t * C * 2);
    const float4 v1 = *reinterpret_cast<const float4*>(base + (size_t)(t + 128) * C * 2);
    const float4 v2 = *reinterpret_cast<const float4*>(base + (size_t)(t + 256) * C * 2);
    const float4 v3 = *reinterpret_cast<const float4*>(base + (size_t)(t + 384) * C * 2);
    a0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
    a1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    a2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
    a3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
  }
  for (; t < nPart; t += 128) {
    const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)t * C * 2);
    a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
  }
  sm[tl][col * 4 + 0] = a0; sm[tl][col * 4 + 1] = a1; sm[tl][col * 4 + 2] = a2; sm[tl][col * 4 + 3] = a3;
  __syncthreads();
  // 8 values (4 channels x 2) x 16 lanes each: lane j sums tile lanes j, j+16, ... (8 adds), then a fixed xor tree
  const int v = threadIdx.x >> 4, j = threadIdx.x & 15;
  double s = 0.0;
  if (v < 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s += sm[j + 16 * k][v];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __syncthreads();
  if (v < 8 && j == 0) sm[0][v] = s;
  __syncthreads();
  if (threadIdx.x >= 4) return;
  const int c = c0 + threadIdx.x;
  const double S = sm[0][threadIdx.x * 2 + 0], Q = sm[0][threadIdx.x * 2 + 1];
  if constexpr (MODE == 0) {
    const double m0 = S / count;
    double var = Q / count - m0 * m0;
    if (var < 0.0) var = 0.0;
    const double mean = m0 + (o.conv_bias ? (double)o.conv_bias[c] : 0.0);
    const float invstd = (float)(1.0 / sqrt(var + (double)o.eps));
    const float meanf = (float)mean;
    const float a = o.gamma[c] * invstd;
    o.mean[c] = meanf;
    o.invstd[c] = invstd;
    o.a[c] = a;
    o.b[c] = o.beta[c] - meanf * a;
    if (o.rmean) {
      const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
      o.rmean[c] = (1.f - o.momentum) * o.rmean[c] + o.momentum * meanf;
      o.rvar[c] = (1.f - o.momentum) * o.rvar[c] + o.momentum * (float)unbiased;
    }
    if (o.nbt && c == 0) *o.nbt += 1;
  } else {
    const double us = o.unscale ? (double)*o.unscale : 1.0;
    if (o.dbeta) o.dbeta[c] = (float)(S * us);
    if (o.dgamma) o.dgamma[c] = (float)(Q * us);
    o.coef[c * 2 + 0] = (float)(S / count);
    o.coef[c * 2 + 1] = (float)(Q / count);
  }
}

int launch_bn_finalize(const float* partials, int nTiles, int C, int64_t count, const float* conv_bias,
                       const float* gamma, const float* beta, float eps, float momentum, float* mean, float* invstd,
                       float* a, float* b, float* running_mean, float* running_var, int64_t* nbt, double* dscratch,
                       hipStream_t s) {
  if (FU_EXP_SKIP(1)) return 0;
  if (sync_world() <= 1 && C % 4 == 0) {
    BnFwdOut o{conv_bias, gamma, beta, eps, momentum, mean, invstd, a, b, running_mean, running_var, nbt};
    hipLaunchKernelGGL((k_bn_stats_fused<0, BnFwdOut>), dim3(C / 4), dim3(256), 0, s, partials, nTiles, C, (double)count, o);
    FU_LAUNCH_CHECK();
    return 0;
  }
  int G = 0;
  FU_TRY(reduce_partials<2>(partials, dscratch, nTiles, C, s, &G));
  FU_TRY(sync_sum_over_ranks(dscratch, (int64_t)G * C * 2, true, s));     // exact DP: global batch statistics
  hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(C, 8)), dim3(256), 0, s, dscratch, G, C,
                     (double)count * sync_world(),
                     conv_bias, gamma, beta, eps, momentum, mean, invstd, a, b, running_mean, running_var, nbt);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward.  g = dL/d relu(bn(y)) in place -> dL/dy.
//   m = [a*y+b > 0], xh = (y-mean)*invstd
//   s1 = sum g*m, s2 = sum g*m*xh   (dbeta, dgamma)
//   dy = a * (g*m - s1/N - xh*s2/N)
// ------------------------------------------------------------------------------------------------
static constexpr int BNB_THREADS = 256;

template <typename T>
__global__ void k_bn_bwd_reduce(const T* __restrict__ g, const T* __restrict__ y, int C, int64_t npix,
                                const float* __restrict__ a, const float* __restrict__ b,
                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                float* __restrict__ partials) {
  extern __shared__ float sm[];  // [rows][C][2]
  const int CV = C >> 2;
  const int rows = BNB_THREADS / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const bool active = row < rows;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (active) {
    float av[4], bv[4], mv[4], iv[4];
    ElemIO<float>::load4(a + cv * 4, av);
    ElemIO<float>::load4(b + cv * 4, bv);
    ElemIO<float>::load4(mean + cv * 4, mv);
    ElemIO<float>::load4(invstd + cv * 4, iv);
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += (int64_t)gridDim.x * rows) {
      float gv[4], yv[4];
      ElemIO<T>::load4(g + p * C + cv * 4, gv);
      ElemIO<T>::load4(y + p * C + cv * 4, yv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = bn_act_pre(av[j], yv[j], bv[j]);
        const float gm = z > 0.f ? gv[j] : 0.f;
        s1[j] += gm;
        s2[j] += gm * ((yv[j] - mv[j]) * iv[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sm[((row * C) + cv * 4 + j) * 2 + 0] = s1[j];
      sm[((row * C) + cv * 4 + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += BNB_THREADS) {
    float t1 = 0.f, t2 = 0.f;
    for (int r = 0; r < rows; ++r) {
      t1 += sm[(r * C + c) * 2 + 0];
      t2 += sm[(r * C + c) * 2 + 1];
    }
    partials[((int64_t)blockIdx.x * C + c) * 2 + 0] = t1;
    partials[((int64_t)blockIdx.x * C + c) * 2 + 1] = t2;
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const double* __restrict__ dpart, int G, int C,
                                                         double count, double grad_share,
                                                         const float* __restrict__ unscale,
                                                         float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ coef) {
  const int g = threadIdx.x & 31;
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool ok = c < C && g < G;
  double S1 = ok ? dpart[((int64_t)g * C + c) * 2 + 0] : 0.0;
  double S2 = ok ? dpart[((int64_t)g * C + c) * 2 + 1] : 0.0;
  S1 = half_wave_sum(S1);
  S2 = half_wave_sum(S2);
  if (c >= C || g != 0) return;
  // (exact DP: S1, S2 are sums over all ranks; the parameter gradients are summed over the ranks afterwards, so each
  //  rank contributes 1/world of them)
  const double us = unscale ? (double)*unscale : 1.0;      // fp16 mode: g carries the loss scale, the parameters' gradients do not
  if (dbeta) dbeta[c] = (float)(S1 * grad_share * us);
  if (dgamma) dgamma[c] = (float)(S2 * grad_share * us);
  coef[c * 2 + 0] = (float)(S1 / count);
  coef[c * 2 + 1] = (float)(S2 / count);
}

template <typename T>
__global__ void k_bn_bwd_apply(T* __restrict__ g, const T* __restrict__ y, int C, int64_t npix,
                               const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ mean, const float* __restrict__ invstd,
                               const float* __restrict__ coef, float* __restrict__ db_partials) {
  extern __shared__ float sm[];  // [rows][C]
  const int CV = C >> 2;
  const int rows = BNB_THREADS / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const bool active = row < rows;
  float sd[4] = {0, 0, 0, 0};
  if (active) {
    float av[4], bv[4], mv[4], iv[4], c1[4], c2[4];
    ElemIO<float>::load4(a + cv * 4, av);
    ElemIO<float>::load4(b + cv * 4, bv);
    ElemIO<float>::load4(mean + cv * 4, mv);
    ElemIO<float>::load4(invstd + cv * 4, iv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c1[j] = coef[(cv * 4 + j) * 2 + 0];
      c2[j] = coef[(cv * 4 + j) * 2 + 1];
    }
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += (int64_t)gridDim.x * rows) {
      float gv[4], yv[4], o[4];
      ElemIO<T>::load4(g + p * C + cv * 4, gv);
      ElemIO<T>::load4(y + p * C + cv * 4, yv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = bn_act_pre(av[j], yv[j], bv[j]);
        const float gm = z > 0.f ? gv[j] : 0.f;
        const float xh = (yv[j] - mv[j]) * iv[j];
        o[j] = av[j] * (gm - c1[j] - xh * c2[j]);
        sd[j] += o[j];
      }
      ElemIO<T>::store4(g + p * C + cv * 4, o);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[row * C + cv * 4 + j] = sd[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += BNB_THREADS) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * C + c];
    db_partials[(int64_t)blockIdx.x * C + c] = t;
  }
}

template <int V>
__device__ __forceinline__ void load_coef(const float* a, const float* b, int c0, float (&av)[V], float (&bv)[V]);
// The same apply pass for the BatchNorm in front of the 1x1 head, with g = dl . w recomputed per element (HeadGrad,
// fu_common.h) instead of read: g stays in fp32 (the stored copy was rounded to the element type), the sums in `coef`
// were taken by k_head_bwd from the same fp32 values.  16-byte vectors, one pixel's channels on C / V lanes.
template <typename T, int NC>
__global__ __launch_bounds__(BNB_THREADS) void k_bn_bwd_apply_head(const float* __restrict__ dl, const float* __restrict__ w,
                                                                    int ncls_rt, T* __restrict__ g, const T* __restrict__ y,
                                                                    int C, int64_t npix, const float* __restrict__ a,
                                                                    const float* __restrict__ b,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ coef,
                                                                    float* __restrict__ db_partials) {
  constexpr int V = VecIO<T>::V;
  constexpr int KMAX = NC ? NC : HEAD_MAX_CLS;
  const int ncls = NC ? NC : ncls_rt;
  extern __shared__ float sm[];  // [rows][C]
  const int CV = C / V;
  const int rows = BNB_THREADS / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  float sd[V];
#pragma unroll
  for (int j = 0; j < V; ++j) sd[j] = 0.f;
  if (row < rows) {
    float av[V], bv[V], mv[V], iv[V], c1[V], c2[V], wv[KMAX][V];
    load_coef<V>(a, b, cv * V, av, bv);
    load_coef<V>(mean, invstd, cv * V, mv, iv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      c1[j] = coef[(cv * V + j) * 2 + 0];
      c2[j] = coef[(cv * V + j) * 2 + 1];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) wv[k][j] = k < ncls ? w[k * C + cv * V + j] : 0.f;
    }
    constexpr int U = 2;                                         // pixels in flight per thread
    const int64_t step = (int64_t)gridDim.x * rows;
    for (int64_t p0 = (int64_t)blockIdx.x * rows + row; p0 < npix; p0 += U * step) {
      float yv[U][V], d[U][KMAX];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * step;
        const bool ok = p < npix;
        VecIO<T>::load(y + (ok ? p : p0) * C + cv * V, yv[u]);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) d[u][k] = (ok && k < ncls) ? dl[p * ncls + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * step;
        if (p < npix) {
          float o[V];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            float gv = 0.f;                                      // same expression and order as k_head_bwd
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
              if (k < ncls) gv += d[u][k] * wv[k][j];
            const float z = bn_act_pre(av[j], yv[u][j], bv[j]);
            const float gm = z > 0.f ? gv : 0.f;
            const float xh = (yv[u][j] - mv[j]) * iv[j];
            o[j] = av[j] * (gm - c1[j] - xh * c2[j]);
            sd[j] += o[j];
          }
          VecIO<T>::store(g + p * C + cv * V, o);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) sm[row * C + cv * V + j] = sd[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += BNB_THREADS) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * C + c];
    db_partials[(int64_t)blockIdx.x * C + c] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward with the max-pool backward folded in (round 2).  For the four encoder outputs that feed a pool,
// dL/d relu(bn(y)) = g (the skip gradient written by the decoder's dgrad) + the pooled gradient routed to the first argmax
// of every 2x2 window.  k_maxpool2_bwd used to add that into g in a read-modify-write pass of its own (a y-sized read, a
// g-sized read and write); here one thread owns a whole 2x2 window x 4 channels, recomputes the window's activations (it
// needs them for the ReLU mask anyway), routes the pooled gradient in registers and does the BN-backward reduction /
// apply on the four pixels.  Same tie rule (first maximum in row-major order, as ATen), same fixed-order block partials.
// ------------------------------------------------------------------------------------------------
template <typename T, bool APPLY>
__global__ void k_bn_bwd_pool(T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ gpool, int C, int B,
                              int H, int W, const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ mean, const float* __restrict__ invstd,
                              const float* __restrict__ coef, float* __restrict__ partials, unsigned rcpWw,
                              unsigned rcpHw) {
  extern __shared__ float sm[];  // APPLY: [rows][C] (sum dy);  reduce: [rows][C][2]
  const int CV = C >> 2;
  const int rows = BNB_THREADS / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const int Ho = H >> 1, Wo = W >> 1, Hw = (H + 1) >> 1, Ww = (W + 1) >> 1;     // pool outputs; windows incl. odd edges
  const int nwin = B * Hw * Ww;             // pixel and window counts < 2^31 (launch_bn_bwd checks): 32-bit window decode by
                                            // host reciprocals (it was two 64-bit divisions per window); 64-bit element offsets
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (row < rows) {
    float av[4], bv[4], mv[4], iv[4], c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    ElemIO<float>::load4(a + cv * 4, av);
    ElemIO<float>::load4(b + cv * 4, bv);
    ElemIO<float>::load4(mean + cv * 4, mv);
    ElemIO<float>::load4(invstd + cv * 4, iv);
    if (APPLY) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { c1[j] = coef[(cv * 4 + j) * 2 + 0]; c2[j] = coef[(cv * 4 + j) * 2 + 1]; }
    }
    for (int wi = blockIdx.x * rows + row; wi < nwin; wi += gridDim.x * rows) {
      const int r = fast_div(wi, Ww, rcpWw);
      const int wx = wi - r * Ww;
      const int bb = fast_div(r, Hw, rcpHw);
      const int wy = r - bb * Hw;
      const bool pooled = wy < Ho && wx < Wo;                     // complete window: has a pool output
      float yv[4][4], gv[4][4], z[4][4], gp[4] = {0, 0, 0, 0};
      bool ok[4];
      int64_t off[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int py = 2 * wy + (q >> 1), px = 2 * wx + (q & 1);
        ok[q] = py < H && px < W;
        off[q] = (int64_t)((bb * H + (ok[q] ? py : 0)) * W + (ok[q] ? px : 0)) * C + cv * 4;
        ElemIO<T>::load4(y + off[q], yv[q]);
        ElemIO<T>::load4(g + off[q], gv[q]);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[q][j] = bn_act_pre(av[j], yv[q][j], bv[j]);
      }
      if (pooled) ElemIO<T>::load4(gpool + (int64_t)((bb * Ho + wy) * Wo + wx) * C + cv * 4, gp);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int am = 0;
        float m = fmaxf(z[0][j], 0.f);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const float zq = fmaxf(z[q][j], 0.f);
          if (zq > m) { m = zq; am = q; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q][j] += (am == q) ? gp[j] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gm = (z[q][j] > 0.f && ok[q]) ? gv[q][j] : 0.f;
          const float xh = (yv[q][j] - mv[j]) * iv[j];
          if (APPLY) {
            o[j] = av[j] * (gm - c1[j] - xh * c2[j]);
            s1[j] += ok[q] ? o[j] : 0.f;
          } else {
            s1[j] += gm;
            s2[j] += gm * xh;
          }
        }
        if (APPLY && ok[q]) ElemIO<T>::store4(g + off[q], o);
      }
    }
    if (APPLY) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sm[row * C + cv * 4 + j] = s1[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sm[((row * C) + cv * 4 + j) * 2 + 0] = s1[j];
        sm[((row * C) + cv * 4 + j) * 2 + 1] = s2[j];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += BNB_THREADS) {
    if (APPLY) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += sm[r * C + c];
      partials[(int64_t)blockIdx.x * C + c] = t;
    } else {
      float t1 = 0.f, t2 = 0.f;
      for (int r = 0; r < rows; ++r) { t1 += sm[(r * C + c) * 2 + 0]; t2 += sm[(r * C + c) * 2 + 1]; }
      partials[((int64_t)blockIdx.x * C + c) * 2 + 0] = t1;
      partials[((int64_t)blockIdx.x * C + c) * 2 + 1] = t2;
    }
  }
}

static int bn_bwd_blocks(int C, int64_t npix) {
  const int rows = BNB_THREADS / (C >> 2);
  int64_t nb = ceil_div64(npix, (int64_t)rows * 8);  // ~8 pixels per thread
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}
int64_t bn_bwd_partial_elems(int C, int64_t npix) { return (int64_t)bn_bwd_blocks(C, npix) * C * 2; }

template <typename T>
static void launch_bn_bwd_pool_t(bool apply, int nb, size_t sh, hipStream_t s, void* g, const void* y, const void* gpool,
                                 int C, int B, int H, int W, const float* a, const float* b, const float* mean,
                                 const float* invstd, const float* coef, float* partials) {
  const unsigned rW = host_rcp((W + 1) >> 1), rH = host_rcp((H + 1) >> 1);
  if (apply)
    hipLaunchKernelGGL((k_bn_bwd_pool<T, true>), dim3(nb), dim3(BNB_THREADS), sh, s, (T*)g, (const T*)y, (const T*)gpool, C,
                       B, H, W, a, b, mean, invstd, coef, partials, rW, rH);
  else
    hipLaunchKernelGGL((k_bn_bwd_pool<T, false>), dim3(nb), dim3(BNB_THREADS), sh, s, (T*)g, (const T*)y, (const T*)gpool,
                       C, B, H, W, a, b, mean, invstd, coef, partials, rW, rH);
}

int launch_bn_bwd(Prec p, void* g, const void* y, int C, int64_t npix, const float* a, const float* b,
                  const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                  float* partials, float* coef, float* db_partials, int* n_db_partials, double* dscratch,
                  hipStream_t s, const void* g_pool, int B, int H, int W, int ext_partials, const HeadGrad* head) {
  (void)gamma;
  FU_REQUIRE(head == nullptr || (ext_partials > 0 && g_pool == nullptr && p != PREC_F32 && C % 8 == 0 && BNB_THREADS % (C / 8) == 0),
             "bn_bwd: a recomputed head gradient needs the producer's sums, a 16-bit element type and C | 2048");
  FU_REQUIRE(C % 4 == 0 && C <= 1024, "bn_bwd: channels must be a multiple of 4 and <= 1024 (got %d)", C);
  // ext_partials > 0: `partials` already holds that many [C][2] rows of the two sums (written by the producer of g, see
  // BnbFuse in fu_common.h) -- the reduce pass is skipped, the apply pass keeps its own grid
  FU_REQUIRE(ext_partials == 0 || (g_pool == nullptr && sync_world() <= 1), "bn_bwd: external partial sums with a pooled source / exact sync");
  const int nb = bn_bwd_blocks(C, npix);
  const int rows = BNB_THREADS / (C >> 2);
  const size_t sh1 = (size_t)rows * C * 2 * sizeof(float);
  const bool pool = g_pool != nullptr;     // the max-pool backward of this tensor is folded into the two passes
  if (pool && FU_EXP_SKIP(64)) {
  } else if (pool) {
    FU_REQUIRE((int64_t)B * H * W == npix && BNB_THREADS % (C >> 2) == 0, "bn_bwd (pooled): bad geometry");
    // both fast_div decodes of k_bn_bwd_pool need n * d < 2^32: windows / Ww and (windows / Ww) / Hw
    FU_REQUIRE(npix < ((int64_t)1 << 31) && (int64_t)B * ((H + 1) / 2) * (int64_t)((W + 1) / 2) * ((W + 1) / 2) < ((int64_t)1 << 32) &&
                   (int64_t)B * ((H + 1) / 2) * (int64_t)((H + 1) / 2) < ((int64_t)1 << 32),
               "bn_bwd (pooled): tensor too large for the 32-bit window decode");
    if (p == PREC_F32) launch_bn_bwd_pool_t<float>(false, nb, sh1, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, partials);
    else if (p == PREC_BF16) launch_bn_bwd_pool_t<bf16_t>(false, nb, sh1, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, partials);
    else launch_bn_bwd_pool_t<f16_t>(false, nb, sh1, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, partials);
  } else if (ext_partials > 0) {
    // (nothing to launch)
  } else if (p == PREC_F32)
    hipLaunchKernelGGL(k_bn_bwd_reduce<float>, dim3(nb), dim3(BNB_THREADS), sh1, s, (const float*)g, (const float*)y,
                       C, npix, a, b, mean, invstd, partials);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_bn_bwd_reduce<bf16_t>, dim3(nb), dim3(BNB_THREADS), sh1, s, (const bf16_t*)g,
                       (const bf16_t*)y, C, npix, a, b, mean, invstd, partials);
  else
    hipLaunchKernelGGL(k_bn_bwd_reduce<f16_t>, dim3(nb), dim3(BNB_THREADS), sh1, s, (const f16_t*)g,
                       (const f16_t*)y, C, npix, a, b, mean, invstd, partials);
  FU_LAUNCH_CHECK();
  if (FU_EXP_SKIP(2)) {
  } else if (sync_world() <= 1) {
    BnBwdOut o{g_grad_unscale, dgamma, dbeta, coef};
    hipLaunchKernelGGL((k_bn_stats_fused<1, BnBwdOut>), dim3(C / 4), dim3(256), 0, s, partials,
                       ext_partials > 0 ? ext_partials : nb, C, (double)npix, o);
    FU_LAUNCH_CHECK();
  } else {
    int G = 0;
    FU_TRY(reduce_partials<2>(partials, dscratch, nb, C, s, &G));
    FU_TRY(sync_sum_over_ranks(dscratch, (int64_t)G * C * 2, true, s));     // exact DP: global sums of g and g*xhat
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ceil_div(C, 8)), dim3(256), 0, s, dscratch, G, C,
                       (double)npix * sync_world(), 1.0 / sync_world(), g_grad_unscale, dgamma, dbeta, coef);
    FU_LAUNCH_CHECK();
  }
  const size_t sh2 = (size_t)rows * C * sizeof(float);
  if ((pool && FU_EXP_SKIP(64)) || (!pool && FU_EXP_SKIP(16))) {
  } else if (pool) {
    if (p == PREC_F32) launch_bn_bwd_pool_t<float>(true, nb, sh2, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, db_partials);
    else if (p == PREC_BF16) launch_bn_bwd_pool_t<bf16_t>(true, nb, sh2, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, db_partials);
    else launch_bn_bwd_pool_t<f16_t>(true, nb, sh2, s, g, y, g_pool, C, B, H, W, a, b, mean, invstd, coef, db_partials);
  } else if (head) {
    const size_t shh = (size_t)(BNB_THREADS / (C / 8)) * C * sizeof(float);
#define FU_APPLY_HEAD(TT, NC) \
    hipLaunchKernelGGL((k_bn_bwd_apply_head<TT, NC>), dim3(nb), dim3(BNB_THREADS), shh, s, head->dl, head->w, head->ncls, \
                       (TT*)g, (const TT*)y, C, npix, a, b, mean, invstd, coef, db_partials)
    if (p == PREC_BF16) {
      switch (head->ncls) {
        case 2: FU_APPLY_HEAD(bf16_t, 2); break;
        case 3: FU_APPLY_HEAD(bf16_t, 3); break;
        default: FU_APPLY_HEAD(bf16_t, 0); break;
      }
    } else {
      switch (head->ncls) {
        case 2: FU_APPLY_HEAD(f16_t, 2); break;
        case 3: FU_APPLY_HEAD(f16_t, 3); break;
        default: FU_APPLY_HEAD(f16_t, 0); break;
      }
    }
#undef FU_APPLY_HEAD
  } else if (p == PREC_F32)
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(nb), dim3(BNB_THREADS), sh2, s, (float*)g, (const float*)y, C, npix,
                       a, b, mean, invstd, coef, db_partials);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_bn_bwd_apply<bf16_t>, dim3(nb), dim3(BNB_THREADS), sh2, s, (bf16_t*)g, (const bf16_t*)y, C,
                       npix, a, b, mean, invstd, coef, db_partials);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<f16_t>, dim3(nb), dim3(BNB_THREADS), sh2, s, (f16_t*)g, (const f16_t*)y, C,
                       npix, a, b, mean, invstd, coef, db_partials);
  FU_LAUNCH_CHECK();
  *n_db_partials = nb;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2) on relu(a*y+b) (or on y as is when a == null)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_act4(const T* p, const float (&av)[4], const float (&bv)[4], bool bn,
                                          float (&z)[4]) {
  ElemIO<T>::load4(p, z);
  if (bn) {
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = bn_act(av[j], z[j], bv[j]);
  }
}

// Elementwise NHWC kernels below share one indexing scheme: grid = (items of one output row / 256, rows, batch), one
// 16-byte channel vector (VecIO: 4 fp32 / 8 bf16) per thread, 32-bit index math.  (The first versions decoded a flat
// 64-bit index with four 64-bit divisions per 8-byte vector and ran 2-5x off the HBM roofline on index math alone.)
template <typename T, int V>
__device__ __forceinline__ void load_act(const T* p, const float (&av)[V], const float (&bv)[V], bool bn, float (&z)[V]) {
  VecIO<T>::load(p, z);
  if (bn) {
#pragma unroll
    for (int j = 0; j < V; ++j) z[j] = bn_act(av[j], z[j], bv[j]);
  }
}
template <int V>
__device__ __forceinline__ void load_coef(const float* a, const float* b, int c0, float (&av)[V], float (&bv)[V]) {
#pragma unroll
  for (int j = 0; j < V; j += 4) {
    const float4 x = *reinterpret_cast<const float4*>(a + c0 + j);
    const float4 y = *reinterpret_cast<const float4*>(b + c0 + j);
    av[j] = x.x; av[j + 1] = x.y; av[j + 2] = x.z; av[j + 3] = x.w;
    bv[j] = y.x; bv[j + 1] = y.y; bv[j + 2] = y.z; bv[j + 3] = y.w;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxpool2(const T* __restrict__ src, const float* __restrict__ a,
                                                  const float* __restrict__ b, T* __restrict__ dst, int H, int W, int C,
                                                  int Ho, int Wo, int CV, unsigned rcpCV) {
  constexpr int V = VecIO<T>::V;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= Wo * CV) return;
  const int ox = fast_div(item, CV, rcpCV), cv = item - ox * CV;
  const int oy = blockIdx.y, bb = blockIdx.z;
  float av[V], bv[V];
  const bool bn = a != nullptr;
  if (bn) load_coef<V>(a, b, cv * V, av, bv);
  const T* base = src + ((size_t)(bb * H + oy * 2) * W + ox * 2) * C + cv * V;
  float z00[V], z01[V], z10[V], z11[V], o[V];
  load_act<T, V>(base, av, bv, bn, z00);
  load_act<T, V>(base + C, av, bv, bn, z01);
  load_act<T, V>(base + (size_t)W * C, av, bv, bn, z10);
  load_act<T, V>(base + (size_t)W * C + C, av, bv, bn, z11);
#pragma unroll
  for (int j = 0; j < V; ++j) o[j] = fmaxf(fmaxf(z00[j], z01[j]), fmaxf(z10[j], z11[j]));
  VecIO<T>::store(dst + ((size_t)(bb * Ho + oy) * Wo + ox) * C + cv * V, o);
}

// g_src[first argmax of the window] += g_dst   (ties -> first in row-major order, as ATen's max_pool2d)
// launch geometry of the row kernels above
template <typename T>
static bool row_grid(int C, int items_w, int rows, int B, dim3* grid, int* CV, unsigned* rcp) {
  constexpr int V = 16 / (int)sizeof(T);
  if (C % V != 0 || rows > 65535 || B > 65535) return false;
  *CV = C / V;
  *rcp = host_rcp(*CV);
  *grid = dim3(ceil_div(items_w * *CV, 256), rows, B);
  return true;
}

int launch_maxpool2(Prec p, const void* src, const float* a, const float* b, void* dst, int B, int H, int W, int C,
                    hipStream_t s) {
  const int Ho = H / 2, Wo = W / 2;
  dim3 g; int CV; unsigned rcp;
  if (p == PREC_F32) {
    FU_REQUIRE(row_grid<float>(C, Wo, Ho, B, &g, &CV, &rcp), "maxpool: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_maxpool2<float>, g, dim3(256), 0, s, (const float*)src, a, b, (float*)dst, H, W, C, Ho, Wo, CV,
                       rcp);
  } else if (p == PREC_BF16) {
    FU_REQUIRE(row_grid<bf16_t>(C, Wo, Ho, B, &g, &CV, &rcp), "maxpool: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_maxpool2<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)src, a, b, (bf16_t*)dst, H, W, C, Ho, Wo,
                       CV, rcp);
  } else {
    FU_REQUIRE(row_grid<f16_t>(C, Wo, Ho, B, &g, &CV, &rcp), "maxpool: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_maxpool2<f16_t>, g, dim3(256), 0, s, (const f16_t*)src, a, b, (f16_t*)dst, H, W, C, Ho, Wo,
                       CV, rcp);
  }
  FU_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------
// bilinear x2 (align_corners=True) of relu(a*y+b), zero-padded to outH x outW (F.pad of unet.py:57-62)
// ------------------------------------------------------------------------------------------------
// R consecutive output rows per thread: one row per thread was latency-bound (a wave lived ~2.5 us for one 16-byte store
// per lane: 3.0 TB/s on the 256x256 level); the 4R loads of a thread are independent and issued ahead of the arithmetic
template <typename T, int R>
__global__ __launch_bounds__(256) void k_upsample2(const T* __restrict__ src, const float* __restrict__ a,
                                                   const float* __restrict__ b, T* __restrict__ dst, int H, int W, int C,
                                                   int outH, int outW, int py0, int px0, UpTables t, int CV,
                                                   unsigned rcpCV) {
  constexpr int V = VecIO<T>::V;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= outW * CV) return;
  const int ox = fast_div(item, CV, rcpCV), cv = item - ox * CV;
  const int oy0 = blockIdx.y * R, bb = blockIdx.z;
  const int ux = ox - px0;
  const bool in_x = ux >= 0 && ux < 2 * W;
  float av[V], bv[V];
  const bool bn = a != nullptr;
  if (bn) load_coef<V>(a, b, cv * V, av, bv);
  // source index and weight as ATen computes them (area_pixel_compute_scale<float>, align_corners=True): the same
  // float expressions as the host tables of the backward pass (fu_api.hip build_axis), evaluated here so that no
  // load depends on a table load
  const float sx = t.scale_x * (float)ux;
  const int x0 = in_x ? (int)sx : 0;
  const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float wx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f), wx0 = 1.f - wx1;
  const T* base = src + (size_t)bb * H * W * C + cv * V;
  float z00[R][V], z01[R][V], z10[R][V], z11[R][V], wy1[R];
  bool in[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int uy = oy0 + r - py0;                               // block-uniform
    in[r] = in_x && uy >= 0 && uy < 2 * H && oy0 + r < outH;
    const float sy = t.scale_y * (float)uy;
    const int y0 = in[r] ? (int)sy : 0;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
    wy1[r] = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
    VecIO<T>::load(base + ((size_t)y0 * W + x0) * C, z00[r]);
    VecIO<T>::load(base + ((size_t)y0 * W + x1) * C, z01[r]);
    VecIO<T>::load(base + ((size_t)y1 * W + x0) * C, z10[r]);
    VecIO<T>::load(base + ((size_t)y1 * W + x1) * C, z11[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (oy0 + r >= outH) break;
    float o[V];
    const float wy0 = 1.f - wy1[r];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float p00 = z00[r][j], p01 = z01[r][j], p10 = z10[r][j], p11 = z11[r][j];
      if (bn) {
        p00 = bn_act(av[j], p00, bv[j]); p01 = bn_act(av[j], p01, bv[j]);
        p10 = bn_act(av[j], p10, bv[j]); p11 = bn_act(av[j], p11, bv[j]);
      }
      o[j] = in[r] ? wy0 * (wx0 * p00 + wx1 * p01) + wy1[r] * (wx0 * p10 + wx1 * p11) : 0.f;
    }
    VecIO<T>::store(dst + ((size_t)(bb * outH + oy0 + r) * outW + ox) * C + cv * V, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_upsample2_bwd(const T* __restrict__ gdst, T* __restrict__ gsrc, int H, int W,
                                                       int C, int outH, int outW, int py0, int px0, UpTables t, int CV,
                                                       unsigned rcpCV) {
  constexpr int V = VecIO<T>::V;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= W * CV) return;
  const int ix = fast_div(item, CV, rcpCV), cv = item - ix * CV;
  const int iy = blockIdx.y, bb = blockIdx.z;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  const T* base = gdst + (size_t)bb * outH * outW * C + cv * V;
  // the column list of this lane once, ahead of the row loop (it was re-read from the table inside it, a dependent load
  // and a data-dependent break per tap); same taps in the same order, so the sums are unchanged
  int xo[UP_BWD_MAX];
  float xw[UP_BWD_MAX];
#pragma unroll
  for (int jx = 0; jx < UP_BWD_MAX; ++jx) { xo[jx] = t.xb_o[ix * UP_BWD_MAX + jx]; xw[jx] = t.xb_w[ix * UP_BWD_MAX + jx]; }
  for (int jy = 0; jy < UP_BWD_MAX; ++jy) {
    const int oy = t.yb_o[iy * UP_BWD_MAX + jy];                // block-uniform
    if (oy < 0) break;
    const float wy = t.yb_w[iy * UP_BWD_MAX + jy];
    const T* rowp = base + (size_t)(oy + py0) * outW * C;
#pragma unroll
    for (int jx = 0; jx < UP_BWD_MAX; ++jx) {
      if (__builtin_amdgcn_ballot_w64(xo[jx] >= 0) == 0) break;  // wave-uniform: lists are filled front to back
      if (xo[jx] >= 0) {
        const float w = wy * xw[jx];
        float gv[V];
        VecIO<T>::load(rowp + (size_t)(xo[jx] + px0) * C, gv);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += w * gv[j];
      }
    }
  }
  VecIO<T>::store(gsrc + ((size_t)(bb * H + iy) * W + ix) * C + cv * V, acc);
}

int launch_upsample2(Prec p, const void* src, const float* a, const float* b, void* dst, int B, int H, int W, int C,
                     int outH, int outW, const UpTables& t, hipStream_t s) {
  FU_REQUIRE(outH >= 2 * H && outW >= 2 * W, "upsample: target smaller than 2x source");
  const int py0 = (outH - 2 * H) / 2, px0 = (outW - 2 * W) / 2;
  dim3 g; int CV; unsigned rcp;
  // four rows per thread where that still leaves >= 2048 workgroups, else one
  auto go = [&](auto tag) {
    using T = decltype(tag);
    FU_REQUIRE(row_grid<T>(C, outW, outH, B, &g, &CV, &rcp), "upsample: unsupported shape (C=%d H=%d B=%d)", C, outH, B);
    if ((int64_t)g.x * ceil_div(outH, 4) * B >= 2048) {
      g.y = ceil_div(outH, 4);
      hipLaunchKernelGGL((k_upsample2<T, 4>), g, dim3(256), 0, s, (const T*)src, a, b, (T*)dst, H, W, C, outH, outW, py0, px0,
                         t, CV, rcp);
    } else {
      hipLaunchKernelGGL((k_upsample2<T, 1>), g, dim3(256), 0, s, (const T*)src, a, b, (T*)dst, H, W, C, outH, outW, py0, px0,
                         t, CV, rcp);
    }
    return 0;
  };
  if (p == PREC_F32) FU_TRY(go(float{}));
  else if (p == PREC_BF16) FU_TRY(go(bf16_t{}));
  else FU_TRY(go(f16_t{}));
  FU_LAUNCH_CHECK();
  return 0;
}

int launch_upsample2_bwd(Prec p, const void* g_dst, void* g_src, int B, int H, int W, int C, int outH, int outW,
                         const UpTables& t, hipStream_t s) {
  if (FU_EXP_SKIP(32)) return 0;
  const int py0 = (outH - 2 * H) / 2, px0 = (outW - 2 * W) / 2;
  dim3 g; int CV; unsigned rcp;
  if (p == PREC_F32) {
    FU_REQUIRE(row_grid<float>(C, W, H, B, &g, &CV, &rcp), "upsample_bwd: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_upsample2_bwd<float>, g, dim3(256), 0, s, (const float*)g_dst, (float*)g_src, H, W, C, outH,
                       outW, py0, px0, t, CV, rcp);
  } else if (p == PREC_BF16) {
    FU_REQUIRE(row_grid<bf16_t>(C, W, H, B, &g, &CV, &rcp), "upsample_bwd: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_upsample2_bwd<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)g_dst, (bf16_t*)g_src, H, W, C, outH,
                       outW, py0, px0, t, CV, rcp);
  } else {
    FU_REQUIRE(row_grid<f16_t>(C, W, H, B, &g, &CV, &rcp), "upsample_bwd: unsupported shape (C=%d H=%d B=%d)", C, H, B);
    hipLaunchKernelGGL(k_upsample2_bwd<f16_t>, g, dim3(256), 0, s, (const f16_t*)g_dst, (f16_t*)g_src, H, W, C, outH,
                       outW, py0, px0, t, CV, rcp);
  }
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(k=2, s=2) support (bilinear=False variant, unet.py:48-51).  out[2y+ky][2x+kx][co] =
// sum_ci in[y][x][ci] * w[ci][co][ky][kx] + b[co]: four independent 1x1 GEMMs, one per output phase p = 2 ky + kx.  They
// run as ONE 1x1 convolution at the LOW resolution with N = 4 cout output channels ordered (p, co) -- on the MFMA conv
// kernels with the weight embedded as a centre tap, exactly the MACs the operator needs (the first version convolved the
// zero-stuffed input with a 3x3 kernel: 9/4 of the MACs and three extra passes) -- followed by a depth-to-space pass that
// interleaves the phases and applies F.pad (unet.py:57-62).  Backward: space-to-depth of dL/d(up) (which crops the pad),
// then the 1x1 conv's dgrad and its one-tap wgrad.
// ------------------------------------------------------------------------------------------------
// up[b][py0 + 2y + ky][px0 + 2x + kx][co] = y4[b][y][x][(2 ky + kx) cout + co]; zero outside the 2h x 2w window
template <typename T>
__global__ void k_depth_to_space(const T* __restrict__ y4, T* __restrict__ up, int h, int w, int C, int outH, int outW,
                                 int py0, int px0, int64_t total) {
  const int CV = C >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    int64_t r = idx / CV;
    const int ox = (int)(r % outW); r /= outW;
    const int oy = (int)(r % outH);
    const int64_t bb = r / outH;
    float o[4] = {0, 0, 0, 0};
    const int uy = oy - py0, ux = ox - px0;
    if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w) {
      const int ph = (uy & 1) * 2 + (ux & 1);
      ElemIO<T>::load4(y4 + (((bb * h + (uy >> 1)) * w + (ux >> 1)) * 4 + ph) * (int64_t)C + cv * 4, o);
    }
    ElemIO<T>::store4(up + idx * 4, o);
  }
}
// g4[b][y][x][(2 ky + kx) cout + co] = g_up[b][py0 + 2y + ky][px0 + 2x + kx][co]  (the pad region is dropped)
template <typename T>
__global__ void k_space_to_depth(const T* __restrict__ gup, T* __restrict__ g4, int h, int w, int C, int outH, int outW,
                                 int py0, int px0, int64_t total) {
  const int CV = C >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    int64_t r = idx / CV;
    const int ph = (int)(r & 3); r >>= 2;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int64_t bb = r / h;
    float v[4];
    ElemIO<T>::load4(gup + ((bb * outH + py0 + 2 * y + (ph >> 1)) * outW + px0 + 2 * x + (ph & 1)) * (int64_t)C + cv * 4, v);
    ElemIO<T>::store4(g4 + idx * 4, v);
  }
}

// zero everything outside the [py0, py0+2H) x [px0, px0+2W) window of an outH x outW map (the F.pad region)
template <typename T>
__global__ void k_channel_partial_sums(const T* __restrict__ g, int C, int64_t npix, float* __restrict__ partials) {
  extern __shared__ float sm[];  // [rows][C]
  const int CV = C >> 2;
  const int rows = BNB_THREADS / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  if (row < rows) {
    float sd[4] = {0, 0, 0, 0};
    for (int64_t p = (int64_t)blockIdx.x * rows + row; p < npix; p += (int64_t)gridDim.x * rows) {
      float gv[4];
      ElemIO<T>::load4(g + p * C + cv * 4, gv);
#pragma unroll
      for (int j = 0; j < 4; ++j) sd[j] += gv[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[row * C + cv * 4 + j] = sd[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += BNB_THREADS) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += sm[r * C + c];
    partials[(int64_t)blockIdx.x * C + c] = t;
  }
}

// convT weight [Cin][Cout][2][2] -> the embedded 1x1: OIHW 3x3 [(p, co)][Cin][3][3] with w3[...][centre] = w[ci][co][ky][kx]
// (p = 2 ky + kx), zeros elsewhere; bias4[(p, co)] = b[co]
__global__ void k_convT_to_w3(const float* __restrict__ w, const float* __restrict__ b, int Cin, int Cout,
                              float* __restrict__ w3, float* __restrict__ bias4, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % 9);
    const int64_t r = idx / 9;
    const int ci = (int)(r % Cin);
    const int n4 = (int)(r / Cin);                 // (p, co)
    const int ph = n4 / Cout, co = n4 - ph * Cout;
    w3[idx] = tap == 4 ? w[(((int64_t)ci * Cout + co) * 2 + (ph >> 1)) * 2 + (ph & 1)] : 0.f;
    if (tap == 4 && ci == 0) bias4[n4] = b[co];
  }
}
__global__ void k_convT_grad_from_w3(const float* __restrict__ dw3, int Cin, int Cout, float* __restrict__ dw,
                                     int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int kx = (int)(idx & 1), ky = (int)((idx >> 1) & 1);
    const int64_t r = idx >> 2;
    const int co = (int)(r % Cout);
    const int ci = (int)(r / Cout);
    dw[idx] = dw3[((int64_t)((ky * 2 + kx) * Cout + co) * Cin + ci) * 9 + 4];
  }
}
// out[c] = unscale * sum_i partials[i][c]  (bias gradient of the transposed conv; fixed order)
__global__ void k_colsum_partials(const float* __restrict__ partials, int n, int C, const float* __restrict__ unscale,
                                  float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)partials[(int64_t)i * C + c];
  if (unscale) s *= (double)*unscale;
  out[c] = (float)s;
}

int launch_depth_to_space(Prec p, const void* y4, void* up, int B, int h, int w, int C, int outH, int outW,
                          hipStream_t s) {
  const int py0 = (outH - 2 * h) / 2, px0 = (outW - 2 * w) / 2;
  const int64_t total = (int64_t)B * outH * outW * (C / 4);
  const int g = grid_for(total, 256);
  if (p == PREC_F32)
    hipLaunchKernelGGL(k_depth_to_space<float>, dim3(g), dim3(256), 0, s, (const float*)y4, (float*)up, h, w, C, outH,
                       outW, py0, px0, total);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_depth_to_space<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)y4, (bf16_t*)up, h, w, C, outH,
                       outW, py0, px0, total);
  else
    hipLaunchKernelGGL(k_depth_to_space<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)y4, (f16_t*)up, h, w, C, outH,
                       outW, py0, px0, total);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_space_to_depth(Prec p, const void* gup, void* g4, int B, int h, int w, int C, int outH, int outW,
                          hipStream_t s) {
  const int py0 = (outH - 2 * h) / 2, px0 = (outW - 2 * w) / 2;
  const int64_t total = (int64_t)B * h * w * 4 * (C / 4);
  const int g = grid_for(total, 256);
  if (p == PREC_F32)
    hipLaunchKernelGGL(k_space_to_depth<float>, dim3(g), dim3(256), 0, s, (const float*)gup, (float*)g4, h, w, C, outH,
                       outW, py0, px0, total);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_space_to_depth<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)gup, (bf16_t*)g4, h, w, C, outH,
                       outW, py0, px0, total);
  else
    hipLaunchKernelGGL(k_space_to_depth<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)gup, (f16_t*)g4, h, w, C, outH,
                       outW, py0, px0, total);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_colsum_partials(const float* partials, int n, int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum_partials, dim3(ceil_div(C, 64)), dim3(64), 0, s, partials, n, C, g_grad_unscale, out);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_channel_partial_sums(Prec p, const void* g, int C, int64_t npix, float* partials, int* n_partials,
                                hipStream_t s) {
  FU_REQUIRE(C % 4 == 0 && C <= 1024, "channel_sum: channels must be a multiple of 4 and <= 1024");
  const int rows = BNB_THREADS / (C >> 2);
  int64_t nb = ceil_div64(npix, (int64_t)rows * 8);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  const size_t sh = (size_t)rows * C * sizeof(float);
  if (p == PREC_F32)
    hipLaunchKernelGGL(k_channel_partial_sums<float>, dim3((unsigned)nb), dim3(BNB_THREADS), sh, s, (const float*)g, C,
                       npix, partials);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL(k_channel_partial_sums<bf16_t>, dim3((unsigned)nb), dim3(BNB_THREADS), sh, s, (const bf16_t*)g,
                       C, npix, partials);
  else
    hipLaunchKernelGGL(k_channel_partial_sums<f16_t>, dim3((unsigned)nb), dim3(BNB_THREADS), sh, s, (const f16_t*)g,
                       C, npix, partials);
  FU_LAUNCH_CHECK();
  *n_partials = (int)nb;
  return 0;
}
int launch_convT_to_w3(const float* w, const float* b, int Cin, int Cout, float* w3, float* bias4, hipStream_t s) {
  const int64_t total = (int64_t)4 * Cout * Cin * 9;
  hipLaunchKernelGGL(k_convT_to_w3, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, w, b, Cin, Cout, w3, bias4, total);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_convT_grad_from_w3(const float* dw3, int Cin, int Cout, float* dw, hipStream_t s) {
  const int64_t total = (int64_t)Cin * Cout * 4;
  hipLaunchKernelGGL(k_convT_grad_from_w3, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, dw3, Cin, Cout, dw, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Late fusion (lf_model.py:78-90): channel-window copies between NHWC tensors, with the producer's BN + ReLU applied on
// the way in (concat of the encoders' features) or plain (split of the concat's gradient), and the 1x1 fusion weight
// embedded as the centre tap of a 3x3 one (the fusion conv runs on the 3x3 kernels for now: 9x the MACs it needs).
// ------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ void k_copy_channels(const T* __restrict__ src, int srcC, int src_off, const float* __restrict__ a,
                                const float* __restrict__ b, T* __restrict__ dst, int dstC, int dst_off, int C,
                                int64_t npix) {
  const int vec = C / V;
  const int64_t total = npix * vec;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % vec);
    const int64_t p = idx / vec;
    float v[V];
    VecIO<T>::load(src + p * srcC + src_off + cv * V, v);
    if (a != nullptr) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = bn_act(a[cv * V + k], v[k], b[cv * V + k]);
    }
    VecIO<T>::store(dst + p * dstC + dst_off + cv * V, v);
  }
}
int launch_copy_channels(Prec p, const void* src, int srcC, int src_off, const float* a, const float* b, void* dst,
                         int dstC, int dst_off, int C, int64_t npix, hipStream_t s) {
  const int V = p == PREC_F32 ? 4 : 8;
  FU_REQUIRE(C % V == 0 && srcC % V == 0 && dstC % V == 0 && src_off % V == 0 && dst_off % V == 0,
             "copy_channels: channel counts / offsets must be multiples of %d", V);
  const int g = grid_for(npix * (C / V), 256);
  if (p == PREC_F32)
    hipLaunchKernelGGL((k_copy_channels<float, 4>), dim3(g), dim3(256), 0, s, (const float*)src, srcC, src_off, a, b,
                       (float*)dst, dstC, dst_off, C, npix);
  else if (p == PREC_BF16)
    hipLaunchKernelGGL((k_copy_channels<bf16_t, 8>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, srcC, src_off, a, b,
                       (bf16_t*)dst, dstC, dst_off, C, npix);
  else
    hipLaunchKernelGGL((k_copy_channels<f16_t, 8>), dim3(g), dim3(256), 0, s, (const f16_t*)src, srcC, src_off, a, b,
                       (f16_t*)dst, dstC, dst_off, C, npix);
  FU_LAUNCH_CHECK();
  return 0;
}
__global__ void k_center_to_w3(const float* __restrict__ w, float* __restrict__ w3, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 9; i += (int64_t)gridDim.x * blockDim.x)
    w3[i] = (i % 9 == 4) ? w[i / 9] : 0.f;
}
__global__ void k_center_from_w3(const float* __restrict__ dw3, float* __restrict__ dw, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dw[i] = dw3[i * 9 + 4];
}
int launch_center_to_w3(const float* w, int64_t n, float* w3, hipStream_t s) {
  hipLaunchKernelGGL(k_center_to_w3, dim3(grid_for(n * 9, 256, 4096)), dim3(256), 0, s, w, w3, n);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_center_from_w3(const float* dw3, int64_t n, float* dw, hipStream_t s) {
  hipLaunchKernelGGL(k_center_from_w3, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, dw3, dw, n);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// head: logits[p][k] = bias[k] + sum_c relu(a*y+b)[p][c] * w[k][c]      (OutConv, unet.py:74-77)
// LPP = C/4 lanes cooperate on one pixel (16 for C = 64); partial dot products meet through wave shuffles.
// ------------------------------------------------------------------------------------------------
// sum over the LPP (power of two <= 16) consecutive lanes of a group; the result is valid in the LAST lane of the
// group.  DPP only: quad_perm for xor 1 / 2, row_shr for the quad-to-quad steps (a __shfl_xor is a ds_bpermute).
__device__ __forceinline__ float dpp_add(float v, const int ctrl_sel) {
  int r;
  const int x = __float_as_int(v);
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); break;   // row_shr:4
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); break;  // row_shr:8
  }
  return v + __int_as_float(r);
}
__device__ __forceinline__ float group_sum_last(float v, int LPP) {
  // largest distance first: the same association as the xor-shuffle tree this replaces (bit-identical fp32 logits)
  if (LPP >= 16) v = dpp_add(v, 3);
  if (LPP >= 8) v = dpp_add(v, 2);
  if (LPP >= 4) v = dpp_add(v, 1);
  if (LPP >= 2) v = dpp_add(v, 0);
  return v;
}

// NC: compile-time class count (register arrays sized for it); NC == 0: any count up to HEAD_MAX_CLS.
// Every thread keeps U pixels in flight per iteration (the loop is latency bound otherwise: one 16-byte load per
// thread and ~200 VGPRs for 8 classes gave 89 us for 134 MB).
template <typename T, int NC, int U>
__global__ __launch_bounds__(256) void k_head_fwd(const T* __restrict__ y, const float* __restrict__ a,
                                                  const float* __restrict__ b, const float* __restrict__ w,
                                                  const float* __restrict__ bias, int C, int ncls_rt, int HW, int LPP,
                                                  float* __restrict__ logits_nhwc, float* __restrict__ logits_nchw) {
  constexpr int V = VecIO<T>::V;
  constexpr int KMAX = NC ? NC : HEAD_MAX_CLS;
  const int ncls = NC ? NC : ncls_rt;
  const int lane_in = threadIdx.x & (LPP - 1);
  const int ppb = 256 / LPP;                 // pixels per block and unroll slot
  const int grp = threadIdx.x / LPP;
  const int bb = blockIdx.y;
  float av[V], bv[V];
  const bool bn = a != nullptr;
  if (bn) load_coef<V>(a, b, lane_in * V, av, bv);
  float wv[KMAX][V];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
#pragma unroll
    for (int j = 0; j < V; ++j) wv[k][j] = (k < ncls) ? w[k * C + lane_in * V + j] : 0.f;
  }
  const T* yb = y + (size_t)bb * HW * C + lane_in * V;
  for (int p0 = blockIdx.x * ppb * U; p0 < HW; p0 += gridDim.x * ppb * U) {
    float z[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * ppb + grp;
#pragma unroll
      for (int j = 0; j < V; ++j) z[u][j] = 0.f;
      if (p < HW) load_act<T, V>(yb + (size_t)p * C, av, bv, bn, z[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * ppb + grp;
      float acc[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        float d = 0.f;
        if (k < ncls) {   // uniform
          d = z[u][0] * wv[k][0] + z[u][1] * wv[k][1];   // same expression (and contraction) as the 4-wide original
#pragma unroll
          for (int j = 2; j < V; ++j) d = d + z[u][j] * wv[k][j];
          d = group_sum_last(d, LPP);
        }
        acc[k] = d;
      }
      if (p < HW && lane_in == LPP - 1) {
        const size_t pg = (size_t)bb * HW + p;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < ncls) {
            const float v = acc[k] + bias[k];
            logits_nhwc[pg * ncls + k] = v;
            if (logits_nchw) logits_nchw[((size_t)bb * ncls + k) * HW + p] = v;
          }
        }
      }
    }
  }
}

template <typename T>
static bool head_geometry(int C, int* LPP) {
  constexpr int V = 16 / (int)sizeof(T);
  if (C % V != 0) return false;
  *LPP = C / V;
  return *LPP >= 1 && *LPP <= 16 && (*LPP & (*LPP - 1)) == 0;
}

template <typename T>
static void launch_head_fwd_t(const T* y, const float* a, const float* b, const float* w, const float* bias, int C, int ncls,
                              int B, int HW, int LPP, float* nhwc, float* nchw, hipStream_t s) {
  constexpr int U = 4;
  const int g = ceil_div(ceil_div(HW, (256 / LPP) * U), 2);
#define FU_HEAD_FWD(NC)                                                                                              \
  hipLaunchKernelGGL((k_head_fwd<T, NC, U>), dim3(g, B), dim3(256), 0, s, y, a, b, w, bias, C, ncls, HW, LPP, nhwc, nchw)
  switch (ncls) {
    case 1: FU_HEAD_FWD(1); break;
    case 2: FU_HEAD_FWD(2); break;
    case 3: FU_HEAD_FWD(3); break;
    case 4: FU_HEAD_FWD(4); break;
    default: FU_HEAD_FWD(0); break;
  }
#undef FU_HEAD_FWD
}

int launch_head_fwd(Prec p, const void* y, const float* a, const float* b, const float* w, const float* bias, int C,
                    int ncls, int B, int H, int W, float* logits_nhwc, float* logits_nchw, hipStream_t s) {
  FU_REQUIRE(ncls >= 1 && ncls <= HEAD_MAX_CLS, "head: n_classes must be 1..%d", HEAD_MAX_CLS);
  FU_REQUIRE(B <= 65535, "head: batch too large (%d)", B);
  const int HW = H * W;
  int LPP;
  if (p == PREC_F32) {
    FU_REQUIRE(head_geometry<float>(C, &LPP), "head: base channels must be 4, 8, 16, 32 or 64 in fp32 (got %d)", C);
    launch_head_fwd_t<float>((const float*)y, a, b, w, bias, C, ncls, B, HW, LPP, logits_nhwc, logits_nchw, s);
  } else if (p == PREC_BF16) {
    FU_REQUIRE(head_geometry<bf16_t>(C, &LPP), "head: base channels must be 8, 16, 32, 64 or 128 in bf16 (got %d)", C);
    launch_head_fwd_t<bf16_t>((const bf16_t*)y, a, b, w, bias, C, ncls, B, HW, LPP, logits_nhwc, logits_nchw, s);
  } else {
    FU_REQUIRE(head_geometry<f16_t>(C, &LPP), "head: base channels must be 8, 16, 32, 64 or 128 in fp16 (got %d)", C);
    launch_head_fwd_t<f16_t>((const f16_t*)y, a, b, w, bias, C, ncls, B, HW, LPP, logits_nhwc, logits_nchw, s);
  }
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// softmax cross entropy with ignore_index (water_seg_model.py:40,103-107), argmax, confusion counts
// ------------------------------------------------------------------------------------------------
static constexpr int CE_BLOCK = 256;
static constexpr int CE_MAX_BLOCKS = 1024;

__global__ void k_ce_loss(const float* __restrict__ logits, const int64_t* __restrict__ target, int ncls,
                          int ignore_index, int64_t npix, float* __restrict__ partials,
                          unsigned long long* __restrict__ conf_tmp) {
  __shared__ unsigned int hist[HEAD_MAX_CLS * HEAD_MAX_CLS];
  __shared__ float wsum[CE_BLOCK / 64][2];
  for (int i = threadIdx.x; i < ncls * ncls; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  float lsum = 0.f, cnt = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[p];
    float z[HEAD_MAX_CLS];
    float m = -INFINITY;
    int am = 0;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) {
      if (k < ncls) {
        z[k] = logits[p * ncls + k];
        if (z[k] > m) { m = z[k]; am = k; }
      }
    }
    if (t != (int64_t)ignore_index && t >= 0 && t < ncls) {
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_MAX_CLS; ++k)
        if (k < ncls) se += expf(z[k] - m);
      const float lse = m + logf(se);
      lsum += lse - z[(int)t];
      cnt += 1.f;
      atomicAdd(&hist[(int)t * ncls + am], 1u);
    }
  }
  lsum = wave_sum(lsum);
  cnt = wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { wsum[wave][0] = lsum; wsum[wave][1] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int wv = 0; wv < CE_BLOCK / 64; ++wv) { s0 += wsum[wv][0]; s1 += wsum[wv][1]; }
    partials[blockIdx.x * 2 + 0] = s0;
    partials[blockIdx.x * 2 + 1] = s1;
  }
  for (int i = threadIdx.x; i < ncls * ncls; i += blockDim.x)
    if (hist[i]) atomicAdd(&conf_tmp[i], (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(256) void k_ce_finalize(const float* __restrict__ partials, int nblk, int ncls,
                                                     float* __restrict__ loss_out, int64_t* __restrict__ n_valid_dev,
                                                     unsigned long long* __restrict__ conf_tmp,
                                                     int64_t* __restrict__ conf_accum,
                                                     int64_t* __restrict__ n_valid_out) {
  __shared__ double sm[2][256];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { s += (double)partials[i * 2]; c += (double)partials[i * 2 + 1]; }
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {   // fixed tree: deterministic
    if ((int)threadIdx.x < w) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + w];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s = sm[0][0]; c = sm[1][0];
    // CrossEntropyLoss mean over non-ignored pixels; 0/0 = NaN -> nan_to_num -> 0  (water_seg_model.py:104-106)
    const float loss = c > 0.0 ? (float)(s / c) : 0.f;
    if (loss_out) *loss_out = loss;
    *n_valid_dev = (int64_t)(c + 0.5);
    if (n_valid_out) *n_valid_out = (int64_t)(c + 0.5);
  }
  for (int i = threadIdx.x; i < ncls * ncls; i += blockDim.x) {
    if (conf_accum) conf_accum[i] += (int64_t)conf_tmp[i];
    conf_tmp[i] = 0ull;
  }
}

int launch_ce_loss(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                   float* partials, float* loss_out, int64_t* n_valid_dev, int64_t* confusion_accum,
                   int64_t* n_valid_out, unsigned long long* conf_tmp, hipStream_t s) {
  const int nblk = grid_for(npix, CE_BLOCK, CE_MAX_BLOCKS);
  hipLaunchKernelGGL(k_ce_loss, dim3(nblk), dim3(CE_BLOCK), 0, s, logits_nhwc, target, ncls, ignore_index, npix,
                     partials, conf_tmp);
  FU_LAUNCH_CHECK();
  FU_TRY(sync_sum_over_ranks(partials, (int64_t)nblk * 2, false, s));      // exact DP: global loss sum and N_valid
  hipLaunchKernelGGL(k_ce_finalize, dim3(1), dim3(256), 0, s, partials, nblk, ncls, loss_out, n_valid_dev, conf_tmp,
                     confusion_accum, n_valid_out);
  FU_LAUNCH_CHECK();
  return 0;
}

__global__ void k_ce_grad(const float* __restrict__ logits, const int64_t* __restrict__ target, int ncls,
                          int ignore_index, int64_t npix, const int64_t* __restrict__ n_valid,
                          float* __restrict__ dl) {
  const int64_t nv = *n_valid;
  const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[p];
    const bool valid = (t != (int64_t)ignore_index && t >= 0 && t < ncls) && nv > 0;
    float z[HEAD_MAX_CLS];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k)
      if (k < ncls) { z[k] = logits[p * ncls + k]; m = fmaxf(m, z[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k)
      if (k < ncls) { z[k] = expf(z[k] - m); se += z[k]; }
    const float r = 1.f / se;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k)
      if (k < ncls) dl[p * ncls + k] = valid ? (z[k] * r - ((int)t == k ? 1.f : 0.f)) * inv : 0.f;
  }
}

int launch_ce_grad(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                   const int64_t* n_valid_dev, float* dlogits_nhwc, hipStream_t s) {
  const int g = grid_for(npix, 256, 4096);
  hipLaunchKernelGGL(k_ce_grad, dim3(g), dim3(256), 0, s, logits_nhwc, target, ncls, ignore_index, npix, n_valid_dev,
                     dlogits_nhwc);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BCE + soft Dice on p = softmax(z)[1] (north-star extension; the reference has no such loss -> parity is pinned
// only by oracle/unet_oracle.py:bce_dice_loss).  All spatial reductions in fp32 registers + wave shuffles, fp64 finalize.
//   BCE  = -(1/N) sum_valid [ t log p + (1-t) log(1-p) ],  Dice = 1 - (2 sum p t + 1) / (sum p + sum t + 1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bd_pixel(const float* z, int ncls, float& p, float& logp, float& log1mp, float* s,
                                         float* s1) {
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) m = fmaxf(m, z[k]);
  float se = 0.f, se1 = 0.f;
  float e[HEAD_MAX_CLS];
#pragma unroll
  for (int k = 0; k < HEAD_MAX_CLS; ++k) {
    e[k] = k < ncls ? expf(z[k] - m) : 0.f;
    se += e[k];
    if (k != 1) se1 += e[k];
  }
  const float inv = 1.f / se, inv1 = se1 > 0.f ? 1.f / se1 : 0.f;
#pragma unroll
  for (int k = 0; k < HEAD_MAX_CLS; ++k) { s[k] = e[k] * inv; s1[k] = (k != 1) ? e[k] * inv1 : 0.f; }
  p = s[1];
  const float lse = logf(se);
  logp = (z[1] - m) - lse;
  log1mp = logf(se1) - lse;
}

__global__ void k_bd_loss(const float* __restrict__ logits, const int64_t* __restrict__ target, int ncls,
                          int ignore_index, int64_t npix, float* __restrict__ partials) {
  __shared__ float wsum[CE_BLOCK / 64][5];
  float acc[5] = {0, 0, 0, 0, 0};  // bce, p*t, p, t, n
  for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tg = target[px];
    if (tg != (int64_t)ignore_index && tg >= 0 && tg < ncls) {
      float z[HEAD_MAX_CLS], s[HEAD_MAX_CLS], s1[HEAD_MAX_CLS];
#pragma unroll
      for (int k = 0; k < HEAD_MAX_CLS; ++k) z[k] = k < ncls ? logits[px * ncls + k] : -INFINITY;
      float p, lp, l1p;
      bd_pixel(z, ncls, p, lp, l1p, s, s1);
      const float t = tg == 1 ? 1.f : 0.f;
      acc[0] -= t > 0.f ? lp : l1p;
      acc[1] += p * t; acc[2] += p; acc[3] += t; acc[4] += 1.f;
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float v = wave_sum(acc[j]);
    if ((threadIdx.x & 63) == 0) wsum[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
    for (int wv = 0; wv < CE_BLOCK / 64; ++wv) t += wsum[wv][threadIdx.x];
    partials[blockIdx.x * 5 + threadIdx.x] = t;
  }
}

// coef: [0] = 1/N (0 if N == 0), [1] = D, [2] = 2I+1, [3] = dice weight
__global__ __launch_bounds__(256) void k_bd_finalize(const float* __restrict__ partials, int nblk, float dice_w,
                                                     float* __restrict__ loss_out, float* __restrict__ coef,
                                                     int64_t* __restrict__ n_valid_dev) {
  __shared__ double sm[5][256];
  double a[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblk; i += 256)
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] += (double)partials[i * 5 + j];
#pragma unroll
  for (int j = 0; j < 5; ++j) sm[j][threadIdx.x] = a[j];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
#pragma unroll
      for (int j = 0; j < 5; ++j) sm[j][threadIdx.x] += sm[j][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double bce = sm[0][0], I = sm[1][0], Sp = sm[2][0], St = sm[3][0], N = sm[4][0];
    const double D = Sp + St + 1.0, Nn = 2.0 * I + 1.0;
    const double loss = N > 0.0 ? bce / N + (double)dice_w * (1.0 - Nn / D) : 0.0;
    if (loss_out) *loss_out = (float)loss;
    coef[0] = N > 0.0 ? (float)(1.0 / N) : 0.f;
    coef[1] = (float)D; coef[2] = (float)Nn; coef[3] = N > 0.0 ? dice_w : 0.f;
    *n_valid_dev = (int64_t)(N + 0.5);
  }
}

__global__ void k_bd_grad(const float* __restrict__ logits, const int64_t* __restrict__ target, int ncls,
                          int ignore_index, int64_t npix, const float* __restrict__ coef, float* __restrict__ dl) {
  const float invN = coef[0], D = coef[1], Nn = coef[2], w = coef[3];
  for (int64_t px = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tg = target[px];
    const bool valid = tg != (int64_t)ignore_index && tg >= 0 && tg < ncls && invN > 0.f;
    float z[HEAD_MAX_CLS], s[HEAD_MAX_CLS], s1[HEAD_MAX_CLS];
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) z[k] = k < ncls ? logits[px * ncls + k] : -INFINITY;
    float p, lp, l1p;
    bd_pixel(z, ncls, p, lp, l1p, s, s1);
    const float t = tg == 1 ? 1.f : 0.f;
    const float ddice = -(2.f * t * D - Nn) / (D * D);   // d Dice / d p
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) {
      if (k < ncls) {
        const float d1 = k == 1 ? 1.f : 0.f;
        const float gb = (s[k] - t * d1 - (1.f - t) * s1[k]) * invN;
        const float gd = w * ddice * p * (d1 - s[k]);
        dl[px * ncls + k] = valid ? gb + gd : 0.f;
      }
    }
  }
}

int launch_bce_dice(const float* logits_nhwc, const int64_t* target, int ncls, int ignore_index, int64_t npix,
                    float dice_w, float* partials, float* coef, float* loss_out, int64_t* n_valid_dev,
                    float* dlogits_nhwc, hipStream_t s) {
  FU_REQUIRE(ncls >= 2, "bce_dice needs n_classes >= 2 (class 1 = flood)");
  const int nblk = grid_for(npix, CE_BLOCK, 400);   // 5 floats per block must fit the 2*1024-float partial buffer
  hipLaunchKernelGGL(k_bd_loss, dim3(nblk), dim3(CE_BLOCK), 0, s, logits_nhwc, target, ncls, ignore_index, npix,
                     partials);
  FU_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bd_finalize, dim3(1), dim3(256), 0, s, partials, nblk, dice_w, loss_out, coef, n_valid_dev);
  FU_LAUNCH_CHECK();
  if (dlogits_nhwc) {
    hipLaunchKernelGGL(k_bd_grad, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, logits_nhwc, target, ncls,
                       ignore_index, npix, coef, dlogits_nhwc);
    FU_LAUNCH_CHECK();
  }
  return 0;
}

__global__ void k_dlogits_from_nchw(const float* __restrict__ src, float* __restrict__ dst, int ncls, int HW,
                                    int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % ncls);
    const int64_t p = idx / ncls;
    const int64_t bb = p / HW;
    const int pp = (int)(p % HW);
    dst[idx] = src[(bb * ncls + k) * HW + pp];
  }
}

int launch_dlogits_from_nchw(const float* dlogits_nchw, float* dlogits_nhwc, int ncls, int B, int H, int W,
                             hipStream_t s) {
  const int64_t total = (int64_t)B * H * W * ncls;
  hipLaunchKernelGGL(k_dlogits_from_nchw, dim3(grid_for(total, 256)), dim3(256), 0, s, dlogits_nchw, dlogits_nhwc, ncls,
                     H * W, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// head backward: G[p][c] = sum_k dl[p][k] w[k][c];  dW[k][c] = sum_p dl[p][k] z[p][c];  db[k] = sum_p dl[p][k]
// ------------------------------------------------------------------------------------------------
static constexpr int HB_BLOCKS = 2048;

// BNB: also emit the BatchNorm-backward sums of g (sum g*m, sum g*m*xhat per channel, BnbFuse in fu_common.h) -- y and the
// mask are in registers here anyway; bnpart[block][C][2], one row per block.
template <typename T, int NC, int U, bool BNB, bool STORE = true>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dl, const T* __restrict__ y,
                                                  const float* __restrict__ a, const float* __restrict__ b,
                                                  const float* __restrict__ w, int C, int ncls_rt, int npix, int LPP,
                                                  T* __restrict__ g, float* __restrict__ partials,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  float* __restrict__ bnpart) {
  constexpr int V = VecIO<T>::V;
  constexpr int KMAX = NC ? NC : HEAD_MAX_CLS;
  const int ncls = NC ? NC : ncls_rt;
  extern __shared__ float sm[];  // [groups][ncls*C + ncls]
  const int lane_in = threadIdx.x & (LPP - 1);
  const int grp = threadIdx.x / LPP;
  const int ppb = 256 / LPP;
  const int stride = ncls * C + ncls;
  float av[V], bv[V];
  const bool bn = a != nullptr;
  if (bn) load_coef<V>(a, b, lane_in * V, av, bv);
  float wv[KMAX][V], dw[KMAX][V], db[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
#pragma unroll
    for (int j = 0; j < V; ++j) { wv[k][j] = (k < ncls) ? w[k * C + lane_in * V + j] : 0.f; dw[k][j] = 0.f; }
    db[k] = 0.f;
  }
  float iv[V], mi[V], s1[V], s2[V];                   // BNB: invstd, -mean * invstd, the two sums
#pragma unroll
  for (int j = 0; j < V; ++j) {
    iv[j] = BNB ? invstd[lane_in * V + j] : 0.f;
    mi[j] = BNB ? -mean[lane_in * V + j] * iv[j] : 0.f;
    s1[j] = 0.f; s2[j] = 0.f;
  }
  // U pixels per thread in flight per iteration; dW / db are summed per thread in visiting order, then per block in
  // LDS and over the blocks in k_head_bwd_finalize (fixed order, deterministic)
  for (int p0 = blockIdx.x * ppb * U; p0 < npix; p0 += gridDim.x * ppb * U) {
    float z[U][V], xh[BNB ? U : 1][V], d[U][KMAX];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * ppb + grp;
      const bool ok = p < npix;
#pragma unroll
      for (int j = 0; j < V; ++j) z[u][j] = 0.f;
      if constexpr (BNB) {
        // (z = 0 marks the masked elements: the mask of the BatchNorm backward is a*y + b > 0, and z = max(a*y + b, 0))
#pragma unroll
        for (int j = 0; j < V; ++j) xh[u][j] = 0.f;
        if (ok) {
          float yv[V];
          VecIO<T>::load(y + (size_t)p * C + lane_in * V, yv);
#pragma unroll
          for (int j = 0; j < V; ++j) { z[u][j] = bn_act(av[j], yv[j], bv[j]); xh[u][j] = fmaf(yv[j], iv[j], mi[j]); }
        }
      } else if (ok) load_act<T, V>(y + (size_t)p * C + lane_in * V, av, bv, bn, z[u]);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) d[u][k] = (ok && k < ncls) ? dl[(size_t)p * ncls + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * ppb + grp;
      if (p < npix) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < ncls) {
#pragma unroll
            for (int j = 0; j < V; ++j) { o[j] += d[u][k] * wv[k][j]; dw[k][j] += d[u][k] * z[u][j]; }
            db[k] += d[u][k];
          }
        }
        if constexpr (STORE) VecIO<T>::store(g + (size_t)p * C + lane_in * V, o);
        if constexpr (BNB) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float gm = z[u][j] > 0.f ? o[j] : 0.f;
            s1[j] += gm;
            s2[j] = fmaf(gm, xh[u][j], s2[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < ncls) {
#pragma unroll
      for (int j = 0; j < V; ++j) sm[grp * stride + k * C + lane_in * V + j] = dw[k][j];
      if (lane_in == 0) sm[grp * stride + ncls * C + k] = db[k];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < stride; e += blockDim.x) {
    float t = 0.f;
    for (int gq = 0; gq < ppb; ++gq) t += sm[gq * stride + e];
    partials[(size_t)blockIdx.x * stride + e] = t;
  }
  if constexpr (BNB) {
    __syncthreads();                                   // sm: now [groups][C][2]
#pragma unroll
    for (int j = 0; j < V; ++j) {
      sm[(grp * C + lane_in * V + j) * 2 + 0] = s1[j];
      sm[(grp * C + lane_in * V + j) * 2 + 1] = s2[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
      float t = 0.f;
      for (int gq = 0; gq < ppb; ++gq) t += sm[gq * 2 * C + e];
      bnpart[(size_t)blockIdx.x * 2 * C + e] = t;
    }
  }
}

__global__ __launch_bounds__(256) void k_head_bwd_finalize(const float* __restrict__ partials, int nblk, int C,
                                                           int ncls, const float* __restrict__ unscale,
                                                           float* __restrict__ dw, float* __restrict__ db) {
  // 8 elements per block, 32 lanes per element; each lane sums every 32nd block partial, fixed xor tree at the end
  const int stride = ncls * C + ncls;
  const int g = threadIdx.x & 31;
  const int e = blockIdx.x * 8 + (threadIdx.x >> 5);
  double s = 0.0;
  if (e < stride) {
#pragma unroll 4
    for (int i = g; i < nblk; i += 32) s += (double)partials[(int64_t)i * stride + e];
  }
  s = half_wave_sum(s);
  if (e >= stride || g != 0) return;
  if (unscale) s *= (double)*unscale;       // fp16 mode: dlogits carried the loss scale
  if (e < ncls * C) dw[e] = (float)s;
  else db[e - ncls * C] = (float)s;
}

int64_t head_bwd_partial_elems(int C, int ncls) { return (int64_t)HB_BLOCKS * (ncls * C + ncls); }

int launch_head_bwd(Prec p, const float* dlogits_nhwc, const void* y, const float* a, const float* b, const float* w,
                    int C, int ncls, int64_t npix, void* g, float* partials, float* dw, float* db, hipStream_t s,
                    const BnbFuse* fuse) {
  FU_REQUIRE(npix < ((int64_t)1 << 31), "head_bwd: too many pixels");
  int LPP;
  if (p == PREC_F32) FU_REQUIRE(head_geometry<float>(C, &LPP), "head_bwd: unsupported channel count %d", C);
  else if (p == PREC_BF16) FU_REQUIRE(head_geometry<bf16_t>(C, &LPP), "head_bwd: unsupported channel count %d", C);
  else FU_REQUIRE(head_geometry<f16_t>(C, &LPP), "head_bwd: unsupported channel count %d", C);
  constexpr int U = 4;
  const int ppb = 256 / LPP;
  int nblk = (int)ceil_div64(npix, (int64_t)ppb * U);
  if (nblk > HB_BLOCKS) nblk = HB_BLOCKS;
  const int stride = ncls * C + ncls;
  // the BatchNorm-backward sums of g, if asked for (16-bit modes with BatchNorm coefficients: the bench path)
  const bool bnb = fuse && fuse->y == y && fuse->tiles_out && a != nullptr && p != PREC_F32 &&
                   (int64_t)nblk * C * 2 <= fuse->max_elems;
  size_t sh = (size_t)ppb * stride * sizeof(float);
  if (bnb && (size_t)ppb * C * 2 * sizeof(float) > sh) sh = (size_t)ppb * C * 2 * sizeof(float);
  FU_REQUIRE(sh <= 64 * 1024, "head_bwd: LDS request too large (%zu)", sh);
  const float* bmean = bnb ? fuse->mean : nullptr;
  const float* binv = bnb ? fuse->invstd : nullptr;
  float* bpart = bnb ? fuse->part : nullptr;
#define FU_HEAD_BWD(TT, NC)                                                                                     \
  do {                                                                                                          \
    if (bnb && fuse->skip_g)                                                                                    \
      hipLaunchKernelGGL((k_head_bwd<TT, NC, U, true, false>), dim3(nblk), dim3(256), sh, s, dlogits_nhwc, (const TT*)y, a, b, \
                         w, C, ncls, (int)npix, LPP, (TT*)g, partials, bmean, binv, bpart);                     \
    else if (bnb)                                                                                               \
      hipLaunchKernelGGL((k_head_bwd<TT, NC, U, true>), dim3(nblk), dim3(256), sh, s, dlogits_nhwc, (const TT*)y, a, b, w, C, \
                         ncls, (int)npix, LPP, (TT*)g, partials, bmean, binv, bpart);                           \
    else                                                                                                        \
      hipLaunchKernelGGL((k_head_bwd<TT, NC, U, false>), dim3(nblk), dim3(256), sh, s, dlogits_nhwc, (const TT*)y, a, b, w, \
                         C, ncls, (int)npix, LPP, (TT*)g, partials, bmean, binv, bpart);                         \
  } while (0)
  if (p == PREC_F32) {
    switch (ncls) {
      case 1: FU_HEAD_BWD(float, 1); break;
      case 2: FU_HEAD_BWD(float, 2); break;
      case 3: FU_HEAD_BWD(float, 3); break;
      case 4: FU_HEAD_BWD(float, 4); break;
      default: FU_HEAD_BWD(float, 0); break;
    }
  } else if (p == PREC_BF16) {
    switch (ncls) {
      case 1: FU_HEAD_BWD(bf16_t, 1); break;
      case 2: FU_HEAD_BWD(bf16_t, 2); break;
      case 3: FU_HEAD_BWD(bf16_t, 3); break;
      case 4: FU_HEAD_BWD(bf16_t, 4); break;
      default: FU_HEAD_BWD(bf16_t, 0); break;
    }
  } else {
    switch (ncls) {
      case 1: FU_HEAD_BWD(f16_t, 1); break;
      case 2: FU_HEAD_BWD(f16_t, 2); break;
      case 3: FU_HEAD_BWD(f16_t, 3); break;
      case 4: FU_HEAD_BWD(f16_t, 4); break;
      default: FU_HEAD_BWD(f16_t, 0); break;
    }
  }
#undef FU_HEAD_BWD
  FU_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_head_bwd_finalize, dim3(ceil_div(stride, 8)), dim3(256), 0, s, partials, nblk, C, ncls,
                     g_grad_unscale, dw, db);
  FU_LAUNCH_CHECK();
  if (bnb) *fuse->tiles_out = nblk;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// On-GPU tile augmentation (datasets/base_dataset.py:494-555): per sample hflip -> vflip -> rotate(angle) with
// torchvision's tensor semantics (nearest, expand=False, centre = image centre, zero fill), applied identically to
// the image [B,C,H,W] fp32 and the target [B,H,W] int64.  One gather pass: the three transforms are composed into a
// single source coordinate per output pixel.
// ------------------------------------------------------------------------------------------------
__global__ void k_augment(const float* __restrict__ img, const int64_t* __restrict__ tgt, float* __restrict__ img_o,
                          int64_t* __restrict__ tgt_o, const int* __restrict__ flags, const float* __restrict__ angle,
                          int C, int H, int W, int64_t target_fill, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % W);
    const int64_t r = idx / W;
    const int oy = (int)(r % H);
    const int b = (int)(r / H);
    const int f = flags[b];
    int sx = ox, sy = oy;
    bool inside = true;
    if (f & 4) {
      // torchvision F.rotate -> affine_grid + grid_sample(nearest, align_corners=False).  Index arithmetic: bit-exact
      // with oracle/unet_oracle.py:augment -- the same fp32 operations in the same order, each rounded on its own (no
      // fma contraction), cos / sin evaluated in double and rounded once to float, round-half-even.
#pragma clang fp contract(off)
      const float th = angle[b] * 0.017453292519943295f;
      const float cs = (float)cos((double)th), sn = (float)sin((double)th);
      const float xc = ((float)ox + 0.5f) - 0.5f * (float)W, yc = ((float)oy + 0.5f) - 0.5f * (float)H;
      const float px = cs * xc, qx = sn * yc, py = sn * xc, qy = cs * yc;
      const float xs = ((px - qx) + 0.5f * (float)W) - 0.5f;
      const float ys = ((py + qy) + 0.5f * (float)H) - 0.5f;
      sx = (int)nearbyintf(xs);
      sy = (int)nearbyintf(ys);
      inside = sx >= 0 && sx < W && sy >= 0 && sy < H;
    }
    if (f & 2) sy = H - 1 - sy;   // the rotate input is the v-flipped, h-flipped tile
    if (f & 1) sx = W - 1 - sx;
    if (tgt_o) tgt_o[idx] = inside ? tgt[((int64_t)b * H + sy) * W + sx] : target_fill;
    for (int c = 0; c < C; ++c) {
      const int64_t o = (((int64_t)b * C + c) * H + oy) * W + ox;
      img_o[o] = inside ? img[(((int64_t)b * C + c) * H + sy) * W + sx] : 0.f;
    }
  }
}

int launch_augment(const float* img, const int64_t* tgt, float* img_o, int64_t* tgt_o, const int* flags,
                   const float* angle, int B, int C, int H, int W, int64_t target_fill, hipStream_t s) {
  const int64_t total = (int64_t)B * H * W;
  hipLaunchKernelGGL(k_augment, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, img, tgt, img_o, tgt_o, flags, angle,
                     C, H, W, target_fill, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Tile assembly: per-tile normalisation (base_dataset.py:77-113), edge-crop buffer
// (base_dataset.py:271-325) and multi-sensor channel concatenation (ef_model.py:28-44 / stacked sensors) of a whole batch
// that is already in HBM, instead of per item in DataLoader workers.
// ------------------------------------------------------------------------------------------------

// one block per (sample, channel): mean and POPULATION std (numpy .mean / .std, ddof = 0) over the valid crop, two passes
// (the plane stays in L2), fp64 accumulation, fixed-order block reduction
__global__ __launch_bounds__(256) void k_tile_stats(SrcList S, int Ctot, int H, int W, const int* __restrict__ vh,
                                                    const int* __restrict__ vw, float* __restrict__ mean_o,
                                                    float* __restrict__ std_o) {
  __shared__ double sm[256];
  const int b = blockIdx.x / Ctot, c = blockIdx.x - b * Ctot;
  int si = 0;
  for (int k = 1; k < S.n; ++k) si = c >= S.coff[k] ? k : si;
  const float* plane = S.p[si] + ((int64_t)b * S.c[si] + (c - S.coff[si])) * H * W;
  // (the crop sizes come from device memory: clamp, an oversized or negative entry must not read past the plane)
  const int h = vh ? min(max(vh[b], 0), H) : H, w = vw ? min(max(vw[b], 0), W) : W;
  const int n = h * w;
  auto block_sum = [&](double v) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
  };
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)plane[(i / w) * W + (i % w)];
  const double mean = n > 0 ? block_sum(a) / n : 0.0;
  double q = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { const double dlt = (double)plane[(i / w) * W + (i % w)] - mean; q += dlt * dlt; }
  const double var = n > 0 ? block_sum(q) / n : 1.0;
  if (threadIdx.x == 0) { mean_o[blockIdx.x] = (float)mean; std_o[blockIdx.x] = (float)sqrt(var); }
}

// out[b][c][y][x] = inside the valid crop ? (src - mean[b][c]) / std[b][c] : pad_value
__global__ void k_assemble_tiles(SrcList S, int Ctot, int H, int W, const int* __restrict__ vh, const int* __restrict__ vw,
                                 const float* __restrict__ mean, const float* __restrict__ stdv, int per_sample,
                                 float pad_value, float* __restrict__ out, int64_t total) {
#pragma clang fp contract(off)      // image -= mean; image /= std: two roundings, as numpy does them
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    int64_t r = idx / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % Ctot);
    const int b = (int)(r / Ctot);
    int si = 0;
    for (int k = 1; k < S.n; ++k) si = c >= S.coff[k] ? k : si;
    const bool inside = y < (vh ? min(max(vh[b], 0), H) : H) && x < (vw ? min(max(vw[b], 0), W) : W);
    float v = pad_value;
    if (inside) {
      v = S.p[si][(((int64_t)b * S.c[si] + (c - S.coff[si])) * H + y) * W + x];
      if (mean) {
        const int mi = per_sample ? b * Ctot + c : c;
        const float d = v - mean[mi];
        v = d / stdv[mi];
      }
    }
    out[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Lanczos-4 resampling of a batch of tiles (round 4).  The reference resamples the WHOLE raster to the
// label raster's size for every item it loads (floodplanet.py:338-340 -> utils_image.py:11-54, cv2.INTER_LANCZOS4) and then cuts
// the tile out.  Lanczos is local: tile rows [Y0, Y0 + TH) of the resampled raster depend on a window of ~TH * scale + 8 source
// rows, so the host ships that window and two 8-tap tables per tile axis (index into the window, weight) and the device computes
//     out[b][c][y][x] = sum_kx wx[b][x][kx] * ( sum_ky wy[b][y][ky] * win[b][c][iy[b][y][ky]][ix[b][x][kx]] )
// in fp32 with the taps accumulated in order and multiply / add rounded separately -- the arithmetic of the numpy restatement
// (datasets/resize.py: rows first, then columns), so the tile equals the crop of the host's whole-raster result bit for bit.
// scale_mode = the sensor scaling that follows the crop in the reference (floodplanet.py:347 / :406 / :467 / :525).
// One thread per output pixel; the window (a few KB per tile and band) is served from L2.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_lanczos4_tiles(const float* __restrict__ win, int C, int win_h, int win_w,
                                                               const int* __restrict__ iy, const float* __restrict__ wy,
                                                               const int* __restrict__ ix, const float* __restrict__ wx,
                                                               int TH, int TW, int scale_mode, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int bc = blockIdx.z, b = bc / C;
  if (x >= TW || y >= TH) return;
  const float* w = win + (size_t)bc * win_h * win_w;
  const int* iyb = iy + ((size_t)b * TH + y) * 8;
  const float* wyb = wy + ((size_t)b * TH + y) * 8;
  const int* ixb = ix + ((size_t)b * TW + x) * 8;
  const float* wxb = wx + ((size_t)b * TW + x) * 8;
  int ry[8], cx[8];
  float fy[8], fx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { ry[k] = iyb[k] * win_w; fy[k] = wyb[k]; cx[k] = ixb[k]; fx[k] = wxb[k]; }
  float o = 0.f;
#pragma unroll
  for (int kx = 0; kx < 8; ++kx) {
    float t = 0.f;
#pragma unroll
    for (int ky = 0; ky < 8; ++ky) {
      const float p = fy[ky] * w[ry[ky] + cx[kx]];      // (two roundings: numpy multiplies, then adds)
      t = t + p;
    }
    const float q = fx[kx] * t;
    o = o + q;
  }
  if (scale_mode == 1) o = fminf(fmaxf((o + 50.f) / 100.f, 0.f), 1.f);            // S1 (dB): clip((x + 50) / 100, 0, 1), NaN -> 0
  else if (scale_mode == 2) o = fminf(fmaxf(o / 4096.f, 0.f), 1.f);                // S2: clip(x / 2^12, 0, 1)
  else if (scale_mode == 3) o = fminf(fmaxf(o, 0.f), 18607.72f) / 18607.72f;       // L8: clip(x, 0, 18607.72) / 18607.72
  else if (scale_mode == 4) o = o / 65536.f;                                       // PS stored as uint16: x / 2^16
  out[((size_t)bc * TH + y) * TW + x] = o;
}

int launch_resize_lanczos4_tiles(const float* win, int B, int C, int win_h, int win_w, const int* iy, const float* wy,
                                 const int* ix, const float* wx, int TH, int TW, int scale_mode, float* out, hipStream_t s) {
  FU_REQUIRE((int64_t)B * C <= 65535 && TH >= 1 && TW >= 1 && win_h >= 1 && win_w >= 1, "resize_lanczos4_tiles: bad shape");
  FU_REQUIRE(scale_mode >= 0 && scale_mode <= 4, "resize_lanczos4_tiles: scale_mode %d", scale_mode);
  hipLaunchKernelGGL(k_resize_lanczos4_tiles, dim3(ceil_div(TW, 64), ceil_div(TH, 4), B * C), dim3(256), 0, s, win, C, win_h,
                     win_w, iy, wy, ix, wx, TH, TW, scale_mode, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("k_resize_lanczos4_tiles launch failed: %s", hipGetErrorString(e)); return 2; }
  return 0;
}

int launch_assemble_tiles(const float* const* srcs, const int* src_channels, int n_src, int B, int H, int W, const int* vh,
                          const int* vw, int norm_mode, const float* gmean, const float* gstd, float pad_value, float* out,
                          float* mean_out, float* std_out, hipStream_t s) {
  FU_REQUIRE(n_src >= 1 && n_src <= 8, "assemble: 1..8 sources (got %d)", n_src);
  SrcList S;
  S.n = n_src;
  int off = 0;
  for (int k = 0; k < n_src; ++k) {
    FU_REQUIRE(srcs[k] && src_channels[k] >= 1, "assemble: bad source %d", k);
    S.p[k] = srcs[k]; S.c[k] = src_channels[k]; S.coff[k] = off; off += src_channels[k];
  }
  S.coff[n_src] = off;
  const int Ctot = off;
  const float *mean = nullptr, *stdv = nullptr;
  int per_sample = 0;
  if (norm_mode == 1) {        // 'local'
    FU_REQUIRE(mean_out && std_out, "assemble: norm_mode 'local' needs mean_out / std_out [B, sum C]");
    hipLaunchKernelGGL(k_tile_stats, dim3(B * Ctot), dim3(256), 0, s, S, Ctot, H, W, vh, vw, mean_out, std_out);
    FU_LAUNCH_CHECK();
    mean = mean_out; stdv = std_out; per_sample = 1;
  } else if (norm_mode == 2) { // 'global'
    FU_REQUIRE(gmean && gstd, "assemble: norm_mode 'global' needs the per-channel parameters");
    mean = gmean; stdv = gstd;
  } else {
    FU_REQUIRE(norm_mode == 0, "assemble: norm_mode must be 0 (None), 1 ('local') or 2 ('global')");
  }
  const int64_t total = (int64_t)B * Ctot * H * W;
  hipLaunchKernelGGL(k_assemble_tiles, dim3(grid_for(total, 256)), dim3(256), 0, s, S, Ctot, H, W, vh, vw, mean, stdv,
                     per_sample, pad_value, out, total);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Inference stitching (utils/utils_image.py:410-494, predict.py:329-347): softmax of a crop's logits is added into an
// overlap-averaging canvas, canvas[h0:hE, w0:wE, :] += p[:dh, :dw, :], weight += 1; finalisation divides by
// (weight + 1e-5) and emits the argmax map.
// ------------------------------------------------------------------------------------------------
__global__ void k_stitch_add(const float* __restrict__ logits_nhwc, int ncls, int cropW, float* __restrict__ canvas,
                             float* __restrict__ weight, int canvasW, int h0, int w0, int dh, int dw) {
  const int64_t total = (int64_t)dh * dw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % dw), y = (int)(idx / dw);
    const float* z = logits_nhwc + ((int64_t)y * cropW + x) * ncls;
    float m = -INFINITY, e[HEAD_MAX_CLS], se = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) m = fmaxf(m, z[k]);
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) { e[k] = k < ncls ? expf(z[k] - m) : 0.f; se += e[k]; }
    const float inv = 1.f / se;
    const int64_t o = (int64_t)(h0 + y) * canvasW + (w0 + x);
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) canvas[o * ncls + k] += e[k] * inv;
    weight[o] += 1.f;
  }
}

__global__ void k_stitch_finalize(float* __restrict__ canvas, const float* __restrict__ weight, int ncls,
                                  int64_t npix, int64_t* __restrict__ argmax_out) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const float inv = 1.f / (weight[p] + 1e-5f);
    float best = -INFINITY;
    int am = 0;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) {
      if (k < ncls) {
        const float v = canvas[p * ncls + k] * inv;
        canvas[p * ncls + k] = v;
        if (v > best) { best = v; am = k; }
      }
    }
    if (argmax_out) argmax_out[p] = am;
  }
}

int launch_stitch_add(const float* logits_nhwc, int ncls, int cropW, float* canvas, float* weight, int canvasW, int h0,
                      int w0, int dh, int dw, hipStream_t s) {
  hipLaunchKernelGGL(k_stitch_add, dim3(grid_for((int64_t)dh * dw, 256, 2048)), dim3(256), 0, s, logits_nhwc, ncls,
                     cropW, canvas, weight, canvasW, h0, w0, dh, dw);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_stitch_finalize(float* canvas, const float* weight, int ncls, int64_t npix, int64_t* argmax_out,
                           hipStream_t s) {
  hipLaunchKernelGGL(k_stitch_finalize, dim3(grid_for(npix, 256, 2048)), dim3(256), 0, s, canvas, weight, ncls, npix,
                     argmax_out);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor update order; water_seg_model.py:200)
// ------------------------------------------------------------------------------------------------
// skip (optional, fp16 mode): device flag set by k_grad_finite_check when a gradient of this step is not finite -- the whole
// update is then left out (parameters and moments untouched), as a GradScaler skips such a step
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float w1, float beta2, float omb2, float bc2_sqrt,
                       float eps, float neg_step, float gscale, const int* __restrict__ skip) {
  // every operation rounds on its own, in ATen's order (no fma contraction): with identical inputs the update is the
  // same float sequence as torch's CPU Adam (lerp_ / mul_ / addcmul_ / sqrt / div / add_ / addcdiv_)
#pragma clang fp contract(off)
  if (skip && *skip) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    const float dm = gi - mi;
    mi = fmaf(w1, dm, mi);                    // exp_avg.lerp_(grad, 1-beta1): the weight < 0.5 branch of ATen's vectorised lerp, fmadd(weight, end - self, self)
    const float vb = vi * beta2;
    const float og = omb2 * gi;
    vi = vb + og * gi;                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    const float num = neg_step * mi;
    p[i] = p[i] + num / denom;                // param.addcdiv_(exp_avg, denom, value=-step_size): self + value * t1 / t2
    m[i] = mi;
    v[i] = vi;
  }
}

// the seven float scalars of k_adam: formed in double as torch.optim.Adam forms them in Python, then rounded once to float
// (the cast ATen applies to a Python scalar operand of a float tensor op)
void adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale, float out[7]) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const double step_size = lr / bc1;
  const double bc2_sqrt = sqrt(bc2);
  out[0] = (float)(1.0 - beta1); out[1] = (float)beta2; out[2] = (float)(1.0 - beta2); out[3] = (float)bc2_sqrt;
  out[4] = (float)eps; out[5] = (float)(-step_size); out[6] = (float)grad_scale;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                double eps, int64_t step, double grad_scale, hipStream_t s, const int* skip) {
  float sc[7];
  adam_scalars(lr, beta1, beta2, eps, step, grad_scale, sc);
  hipLaunchKernelGGL(k_adam, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, p, g, m, v, n, sc[0], sc[1], sc[2], sc[3],
                     sc[4], sc[5], sc[6], skip);
  FU_LAUNCH_CHECK();
  return 0;
}

// fp16 guard.  The loss scale is chosen once per backward from max|dL/dlogits|; what the chain multiplies on top (a
// BatchNorm with a tiny variance: gamma * invstd in the hundreds) can still push an fp16 gradient map past 65504.  The inf /
// NaN then reaches the flat gradient buffer; guard[0] flags it, the Adam launch of that step does nothing, and the next
// backward's scale is halved once more (guard[2] = back-off exponent, taken back by one every 64 clean steps).
//   guard[0] non-finite flag of the running step, [1] steps skipped so far, [2] back-off exponent, [3] clean steps since
__global__ __launch_bounds__(256) void k_grad_finite_check(const float* __restrict__ g, int64_t n, int* __restrict__ guard) {
  bool bad = false;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      bad = bad || !(fabsf(v.x) <= 3.0e38f) || !(fabsf(v.y) <= 3.0e38f) || !(fabsf(v.z) <= 3.0e38f) || !(fabsf(v.w) <= 3.0e38f);
    } else {
      for (int64_t k = i; k < n; ++k) bad = bad || !(fabsf(g[k]) <= 3.0e38f);
    }
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(guard, 1);
}
__global__ void k_guard_book(int* __restrict__ guard) {
  if (guard[0]) { guard[1] += 1; guard[2] = min(guard[2] + 1, 14); guard[3] = 0; guard[0] = 0; }
  else if (++guard[3] >= 64) { guard[3] = 0; guard[2] = max(guard[2] - 1, 0); }
}
int launch_grad_finite_check(const float* g, int64_t n, int* guard, hipStream_t s) {
  hipLaunchKernelGGL(k_grad_finite_check, dim3(grid_for(n, 1024 * 4, 2048)), dim3(256), 0, s, g, n, guard);
  FU_LAUNCH_CHECK();
  return 0;
}
int launch_guard_book(int* guard, hipStream_t s) {
  hipLaunchKernelGGL(k_guard_book, dim3(1), dim3(1), 0, s, guard);
  FU_LAUNCH_CHECK();
  return 0;
}

// The same update with its scalars read from DEVICE memory: a captured (hipGraph) step replays this launch unchanged while
// the step count -- and with it the bias corrections -- moves on; the caller refreshes the seven floats before each replay.
__global__ void k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, int64_t n, const float* __restrict__ sc, const int* __restrict__ skip) {
#pragma clang fp contract(off)
  if (skip && *skip) return;
  const float w1 = sc[0], beta2 = sc[1], omb2 = sc[2], bc2_sqrt = sc[3], eps = sc[4], neg_step = sc[5], gscale = sc[6];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    const float dm = gi - mi;
    mi = fmaf(w1, dm, mi);
    const float vb = vi * beta2;
    const float og = omb2 * gi;
    vi = vb + og * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float num = neg_step * mi;
    p[i] = p[i] + num / denom;
    m[i] = mi;
    v[i] = vi;
  }
}
int launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* scalars_dev, hipStream_t s,
                    const int* skip) {
  hipLaunchKernelGGL(k_adam_dev, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, p, g, m, v, n, scalars_dev, skip);
  FU_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The gradient the head backward consumes: eff = dlogits * up * S, written OUT OF PLACE (the stored loss gradient stays as
// fu_loss_* left it, so a second backward of the same loss -- retain_graph, fu_backward_block(0) twice -- sees the same
// input; in place, the second call would have found max|dl| already in [32, 64), chosen S = 1 and unscaled by 1).
//   up: optional device scalar, the upstream gradient autograd hands to loss.backward() (fu_scale_loss_grad);
//   S:  fp16 mode only (scale != null): 2^k with max|dl * up| * S in [2^5, 2^6) -- three decades of headroom to fp16's
//       65504 for what the backward chain multiplies on top, while the bulk of the gradient maps stays in fp16's normal
//       range; chosen from the data on the device (no host read, any loss, any upstream scale).  scale[0] = S,
//       scale[1] = 1/S (fu_common.h, g_grad_unscale).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_absmax_partial(const float* __restrict__ x, int64_t n, float* __restrict__ partials) {
  __shared__ float sm[4];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = fabsf(x[i]);
    m = (v <= 3.0e38f && v > m) ? v : m;        // (non-finite entries do not define the scale)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void k_loss_grad_eff(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                       const float* __restrict__ partials, int nPart,
                                                       const float* __restrict__ up, float* __restrict__ scale,
                                                       const int* __restrict__ guard) {
  __shared__ float sm[4];
  const float upv = up ? *up : 1.f;
  float f = upv;
  if (scale) {                                                        // uniform
    float m = threadIdx.x < nPart ? partials[threadIdx.x] : 0.f;      // nPart <= 256
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) * fabsf(upv);
    int e = 0;
    if (m > 0.f && m <= 3.0e38f) { (void)frexpf(m, &e); e = 6 - e; }  // m = f * 2^e', f in [0.5, 1)  ->  m * 2^(6 - e') in [32, 64)
    if (guard) e -= guard[2];                                        // back-off after overflowed steps (k_guard_book)
    e = min(max(e, -60), 60);
    const float S = ldexpf(1.f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = S; scale[1] = ldexpf(1.f, -e); }
    f = upv * S;                                                      // a power of two: no extra rounding
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = x[i] * f;
}
int launch_loss_grad_eff(const float* dlogits, float* out, int64_t n, const float* up_scale_dev, float* partials,
                         float* scale, hipStream_t s, const int* guard) {
  int g = 0;
  if (scale) {
    g = grid_for(n, 256 * 16, 256);
    hipLaunchKernelGGL(k_absmax_partial, dim3(g), dim3(256), 0, s, dlogits, n, partials);
    FU_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_loss_grad_eff, dim3(grid_for(n, 256 * 4, 2048)), dim3(256), 0, s, dlogits, out, n, partials, g,
                     up_scale_dev, scale, guard);
  FU_LAUNCH_CHECK();
  return 0;
}

}  // namespace fu
