// Standalone micro-benchmark of the front-end stage-A kernel variants (1 GiB of IQ in HBM).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/bench_decim tools/bench_decim.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../airspy-fmradion_amd/csrc/design.hpp"
#include "../airspy-fmradion_amd/csrc/kernels.hpp"
using namespace fmr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class F>
static float time_it(const char *name, double bytes, F &&launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) launch();
  hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  printf("%-28s %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12 * 100);
  return ms;
}

__global__ void k_copy(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x == 12345.f) out[0] = acc;
}

int main() {
  const size_t N = (size_t)1 << 27;
  ResamplerDesign rs;
  rs.design(10e6, 384e3, 140.0);
  std::vector<float> fa(rs.hA.begin(), rs.hA.end());
  const int D = rs.D, NA = rs.NA, Q = 16;
  std::vector<float> hp((size_t)D * Q, 0.f);
  for (int k = 0; k < NA; k++) hp[(size_t)(k % D) * Q + k / D] = fa[k];
  float2 *d_iq, *d_mid, *d_halo; float *d_hA, *d_hp;
  CK(hipMalloc(&d_iq, N * 8)); CK(hipMalloc(&d_mid, (N / D + 16) * 8)); CK(hipMalloc(&d_halo, 4096 * 8));
  CK(hipMalloc(&d_hA, NA * 4)); CK(hipMalloc(&d_hp, hp.size() * 4));
  CK(hipMemset(d_iq, 0x3c, N * 8)); CK(hipMemset(d_halo, 0, 4096 * 8));
  CK(hipMemcpy(d_hA, fa.data(), NA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_hp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  const int count = (int)((N - 1 - rs.ca()) / D + 1);
  const int H = NA - 1 + D;
  const double bytes = 8.0 * N;
  time_it("read-only float4 sum", bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)d_iq, (float4 *)d_mid, N / 2); });
  time_it("v1 (1 out/lane)", bytes, [&] {
    hipLaunchKernelGGL(k_ifr_decim<256>, dim3((count + 255) / 256), dim3(256), sizeof(float2) * (256 * D + NA - 1), 0, d_iq, (long long)N, (long long)N,
                       d_halo, H, d_hA, NA, D, (long long)rs.ca(), count, d_mid, (long long)0, 0, 0u, 0);
  });
  auto v2 = [&](auto bl, auto abl, const char *name, auto cvt) {
    constexpr int BL = decltype(bl)::value, ABL = decltype(abl)::value, T = 2 * BL, CV = decltype(cvt)::value;
    int s_pad = T + Q; while ((s_pad & 15) != 2) s_pad++;
    const unsigned magic = (1u << 24) / D + 1;
    time_it(name, bytes, [&] {
      hipLaunchKernelGGL((k_ifr_decim2<BL, 16, ABL, false, CV>), dim3((count + T - 1) / T), dim3(BL), sizeof(float2) * (D * s_pad + 2), 0, d_iq, (long long)N, (long long)N,
                         d_halo, H, d_hp, D, rs.ca(), (long long)0, count, d_mid, (long long)0, 0, 0u, 0, s_pad, magic);
    });
  };
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=256" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=256" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=128" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=128" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 64>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=64" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 64>{}, std::integral_constant<int, 0>{}, "v2 BLOCK=64" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 1>{}, "v2 B256 no-compute" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 1>{}, "v2 B256 no-compute" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 2>{}, "v2 B256 no-loads" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 256>{}, std::integral_constant<int, 2>{}, "v2 B256 no-loads" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 1>{}, "v2 B128 no-compute" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 1>{}, "v2 B128 no-compute" " cv0", std::integral_constant<int, 0>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 2>{}, "v2 B128 no-loads" " cv1", std::integral_constant<int, 1>{});
  v2(std::integral_constant<int, 128>{}, std::integral_constant<int, 2>{}, "v2 B128 no-loads" " cv0", std::integral_constant<int, 0>{});
  CK(hipDeviceSynchronize());
  return 0;
}
