// Ablation: the stage-A sums of the IF resampler (151 taps, decimation 10) carried in fp32 (product) and in fp64.
// Both kernels read the same 1 GiB of IQ and round their outputs to fp32; the error of each against an fp64 host
// evaluation (taps in fp64, as the oracle holds them) is reported for the first 2^18 outputs, next to the launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/bench_acc64.bin tools/bench_acc64.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../airspy-fmradion_amd/csrc/design.hpp"
#include "../airspy-fmradion_amd/csrc/kernels.hpp"
using namespace fmr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t N = (size_t)1 << 27;
  ResamplerDesign rs;
  rs.design(10e6, 384e3, 140.0);
  const int D = rs.D, NA = rs.NA, Q = 16;
  std::vector<float> hp((size_t)D * Q, 0.f);
  for (int k = 0; k < NA; k++) hp[(size_t)(k % D) * Q + k / D] = (float)rs.hA[k];
  // an FM-like unit-amplitude carrier with noise (the values matter for rounding, not for the timing)
  std::vector<float2> x(N);
  { std::mt19937_64 g(1); std::normal_distribution<float> nd(0.f, 0.05f);
    double ph = 0.;
    for (size_t n = 0; n < N; n++) { ph += 0.047 * std::sin(2e-3 * (double)n) + 0.01; x[n] = make_float2((float)std::cos(ph) + nd(g), (float)std::sin(ph) + nd(g)); if (n == (1u << 22)) n = N - 1; }
    for (size_t n = (1u << 22) + 1; n < N; n++) x[n] = x[n & ((1u << 22) - 1)]; }
  float2 *d_iq, *d_mid, *d_halo; float *d_hp;
  CK(hipMalloc(&d_iq, N * 8)); CK(hipMalloc(&d_mid, (N / D + 16) * 8)); CK(hipMalloc(&d_halo, 4096 * 8)); CK(hipMalloc(&d_hp, hp.size() * 4));
  CK(hipMemcpy(d_iq, x.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemset(d_halo, 0, 4096 * 8));
  CK(hipMemcpy(d_hp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  const int count = (int)((N - 1 - rs.ca()) / D + 1);
  const int H = NA - 1 + D;
  constexpr int BL = 256, T = 2 * BL;
  int s_pad = T + Q; while ((s_pad & 15) != 2) s_pad++;
  const unsigned magic = (1u << 24) / D + 1;
  const int M = 1 << 18;
  // host fp64: y[m] = sum_k hA[k] x[D m + ca - k], zero history (the halo is zero)
  std::vector<double> ref(2 * (size_t)M);
  double ref_pow = 0.;
  for (int m = 0; m < M; m++) {
    double ar = 0., ai = 0.;
    for (int k = 0; k < NA; k++) {
      const long long n = (long long)D * m + rs.ca() - k;
      if (n < 0) continue;
      ar += rs.hA[k] * (double)x[n].x; ai += rs.hA[k] * (double)x[n].y;
    }
    ref[2 * m] = ar; ref[2 * m + 1] = ai; ref_pow += ar * ar + ai * ai;
  }
  std::vector<float2> got(M);
  auto run = [&](auto cvt, const char *name) -> int {
    constexpr int CV = decltype(cvt)::value;
    auto launch = [&] {
      hipLaunchKernelGGL((k_ifr_decim2<BL, 16, 0, false, CV>), dim3((count + T - 1) / T), dim3(BL), sizeof(float2) * (D * s_pad + 2), 0, d_iq, (long long)N, (long long)N,
                         d_halo, H, d_hp, D, rs.ca(), (long long)0, count, d_mid, (long long)0, 0, 0u, 0, s_pad, magic);
    };
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; i++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    CK(hipMemcpy(got.data(), d_mid, (size_t)M * 8, hipMemcpyDeviceToHost));
    double e2 = 0., emax = 0.;
    for (int m = 0; m < M; m++) {
      const double er = got[m].x - ref[2 * m], ei = got[m].y - ref[2 * m + 1];
      e2 += er * er + ei * ei; emax = std::fmax(emax, std::fmax(std::fabs(er), std::fabs(ei)));
    }
    printf("%-44s %8.1f us  %7.1f GB/s   rel. rms error vs fp64 host %.3e  (max abs %.3e)\n", name, ms * 1e3, 8.0 * N / (ms * 1e-3) / 1e9, std::sqrt(e2 / ref_pow), emax);
    return 0;
  };
  if (run(std::integral_constant<int, 1>{}, "stage A, fp32 accumulate (v_pk_fma_f32, product)")) return 1;
  if (run(std::integral_constant<int, 0>{}, "stage A, fp32 accumulate (scalar fmaf form)")) return 1;
  if (run(std::integral_constant<int, 2>{}, "stage A, fp64 accumulate (v_fma_f64)")) return 1;
  // the floor of the comparison: the fp64 host result itself rounded to fp32
  { double e2 = 0.; for (int m = 0; m < M; m++) { const double er = (double)(float)ref[2 * m] - ref[2 * m], ei = (double)(float)ref[2 * m + 1] - ref[2 * m + 1]; e2 += er * er + ei * ei; }
    printf("%-44s %8s     %7s        rel. rms error vs fp64 host %.3e\n", "fp64 host result rounded to fp32", "-", "-", std::sqrt(e2 / ref_pow)); }
  printf("(taps are fp32 in all three kernels: the product's table.  fp64 taps change the result by the tap rounding, rel. 6e-8 per tap.)\n");
  return 0;
}
