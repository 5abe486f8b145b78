// VALU issue-rate probe for gfx950: packed fp32 FMA vs scalar-operand fp32 FMA vs fp64 FMA/MUL/ADD.
// hipcc --offload-arch=gfx950 -O3 -o tools/bench_valu.bin tools/bench_valu.hip && tools/bench_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (MODE == 0) {          // v_pk_fma_f32, 8 independent chains
    v2f a[8];
    for (int i = 0; i < 8; i++) a[i] = (v2f){(float)t, (float)i};
    v2f m = {s, s}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(m, a[i], c);
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    out[t] = r;
  } else if (MODE == 1) {   // v_fma_f32
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)(t + i);
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = fmaf(s, a[i], 0.5f);
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    out[t] = r;
  } else if (MODE == 2) {   // v_fma_f64
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = (double)(t + i);
    const double sd = s;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = fma(sd, a[i], 0.5);
    double r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    out[t] = (float)r;
  } else {                  // v_mul_f64 + v_add_f64 (no contraction)
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = (double)(t + i);
    const double sd = s;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int i = 0; i < 8; i++) { double p = __dmul_rn(sd, a[i]); a[i] = __dadd_rn(p, 0.5); }
    double r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    out[t] = (float)r;
  }
}

template <int MODE>
int run(const char *name, int per_iter_instr) {
  float *out;
  const int blocks = 256 * 8, iters = 4096;
  CHK(hipMalloc(&out, sizeof(float) * blocks * 256));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 16, 0.999f);
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double winstr = (double)blocks * 4 * iters * per_iter_instr;       // wave-instructions
  const double per_simd = winstr / (256.0 * 4);
  printf("%-22s %8.3f ms   %.2f ns per wave-instr per SIMD  (at 2.4 GHz: %.2f clk)\n", name, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  CHK(hipFree(out));
  return 0;
}

int main() {
  run<0>("v_pk_fma_f32", 64);
  run<1>("v_fma_f32", 64);
  run<2>("v_fma_f64", 64);
  run<3>("v_mul_f64+v_add_f64", 128);
  return 0;
}
