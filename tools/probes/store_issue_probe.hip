// Store-issue probe (round 4): what does one global_store_dwordx4 of a wave cost the CU, by address pattern?  The conv
// epilogue stores 16 pixels x 64 bytes per wave instruction (the wave owns 32 of a pixel's channels): 16 half lines at a
// pixel stride of 128 bytes .. 1 KB.  Patterns (64 lanes x 16 bytes each, 8 stores per wave back to back, 8 waves per CU,
// 256 workgroups, every workgroup its own 1 MB region):
//   0  contiguous 1 KB                        (lane i -> 16 i)
//   1  8 full 128-byte lines, stride 256 B    (lanes 8p..8p+7 -> pixel p)
//   2  16 half lines of 64 B, stride 128 B    (the 64-channel layers' epilogue)
//   3  16 half lines of 64 B, stride 1 KB     (the 512-channel layers' epilogue)
//   4  64 separate 16-byte pieces, stride 128 B
// Reported: cycles from the first store's issue to the last store's issue (issue cost), and to s_waitcnt vmcnt(0)
// (completion), per store instruction, median over the waves.   hipcc --offload-arch=gfx950 -O2 store_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(512) void k_store(unsigned char* buf, unsigned long long* out, int pattern, int rowstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t off;
  switch (pattern) {
    case 0: off = (size_t)lane * 16; break;
    case 1: off = (size_t)(lane >> 3) * 256 + (lane & 7) * 16; break;
    case 2: off = (size_t)(lane >> 2) * 128 + (lane & 3) * 16; break;
    case 3: off = (size_t)(lane >> 2) * 1024 + (lane & 3) * 16; break;
    default: off = (size_t)lane * 128; break;
  }
  unsigned char* p = buf + (size_t)blockIdx.x * (1u << 20) + (size_t)wave * (64u << 10) + off;
  uint4 v = make_uint4(lane, wave, blockIdx.x, pattern);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int r = 0; r < 8; ++r) *reinterpret_cast<uint4*>(p + (size_t)r * rowstride) = v;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

int main() {
  unsigned char* buf; unsigned long long* out;
  hipMalloc(&buf, (size_t)256 << 20);
  hipMalloc(&out, 256 * 8 * 2 * sizeof(unsigned long long));
  const char* names[5] = {"contiguous 1 KB", "8 full 128-B lines", "16 half lines, stride 128 B", "16 half lines, stride 1 KB",
                          "64 pieces of 16 B, stride 128 B"};
  for (int pat = 0; pat < 5; ++pat) {
    std::vector<unsigned long long> h(256 * 8 * 2);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_store, dim3(256), dim3(512), 0, 0, buf, out, pat, 8192);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int i = 0; i < 256 * 8; ++i) { a.push_back(h[2 * i] / 8.0); b.push_back(h[2 * i + 1] / 8.0); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-34s issue %7.1f cycles per store (p90 %7.1f)   to completion %7.1f (p90 %7.1f)\n", names[pat], a[a.size() / 2],
           a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10]);
  }
  return 0;
}
