// Micro-benchmark: f32 VALU throughput of ONE SIMD by instruction form, waves per SIMD and independent chains per wave.
// Answers what the stage-A loop of the fused front end can hope for: is v_pk_fma_f32 twice a v_fma_f32, how many
// independent accumulator chains a wave needs, and what a second / third wave on the SIMD adds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bench_valu2.bin tools/bench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// FORM 0: v_fma_f32   1: v_pk_fma_f32   2: v_pk_fma_f32 with op_sel broadcast of the low half of the tap pair
template <int FORM, int CH>
__global__ __launch_bounds__(768) void k(float *out, int iters, unsigned long long *cyc) {
  v2f acc[CH], x[CH];
  for (int i = 0; i < CH; i++) { acc[i] = (v2f){0.f, 0.f}; x[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f - i}; asm volatile("" : "+v"(x[i]), "+v"(acc[i])); }
  v2f t = {1.0001f, 0.9999f};
  asm volatile("" : "+v"(t));
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 48 / CH; u++)
#pragma unroll
      for (int o = 0; o < CH; o++) {
        if (FORM == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[o].x) : "v"(t.x), "v"(x[o].x));
        if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[o]) : "v"(t), "v"(x[o]));
        if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[o]) : "v"(t), "v"(x[o]));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < CH; i++) r += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int FORM, int CH>
static int run(const char *name, float *out, unsigned long long *cyc) {
  const int iters = 2000;
  for (int wps = 1; wps <= 3; wps++) {
    const int threads = 256 * wps;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<FORM, CH>), dim3(256), dim3(threads), 0, 0, out, iters, cyc); CK(hipDeviceSynchronize()); }
    unsigned long long c[12];
    CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int w = 0; w < 4 * wps; w++) mx = c[w] > mx ? (double)c[w] : mx;
    const double per_wave = mx / (iters * 48.0);
    printf("%-28s chains %d  waves/SIMD %d : %6.2f cycles per instruction and wave, %5.2f per instruction of the SIMD\n", name, CH, wps, per_wave, per_wave / wps);
  }
  return 0;
}

int main() {
  float *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 768 * 4)); CK(hipMalloc(&cyc, 12 * 8));
  run<0, 2>("v_fma_f32", out, cyc); run<0, 4>("v_fma_f32", out, cyc); run<0, 8>("v_fma_f32", out, cyc); run<0, 16>("v_fma_f32", out, cyc);
  run<1, 2>("v_pk_fma_f32", out, cyc); run<1, 4>("v_pk_fma_f32", out, cyc); run<1, 8>("v_pk_fma_f32", out, cyc); run<1, 16>("v_pk_fma_f32", out, cyc);
  run<2, 2>("v_pk_fma_f32 op_sel", out, cyc); run<2, 8>("v_pk_fma_f32 op_sel", out, cyc);
  return 0;
}
