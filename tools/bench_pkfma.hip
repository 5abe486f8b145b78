// Micro-benchmark: issue rate of v_pk_fma_f32 / v_fma_f32 from ONE or TWO waves per SIMD, tap operand in SGPRs or VGPRs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bench_pkfma.bin tools/bench_pkfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, const float *taps, int iters, unsigned long long *cyc) {
  v2f acc[6], x[6];
  for (int i = 0; i < 6; i++) { acc[i] = (v2f){0.f, 0.f}; x[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f - i}; asm volatile("" : "+v"(x[i])); }
  v2f tv[8];
  float ts[8];
  for (int i = 0; i < 8; i++) { ts[i] = taps[i]; tv[i] = (v2f){taps[i], taps[i + 8]}; asm volatile("" : "+v"(tv[i])); }
  float sa[6];
  for (int i = 0; i < 6; i++) sa[i] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int o = 0; o < 6; o++) {
        if (MODE == 0) acc[o] = __builtin_elementwise_fma((v2f){ts[u], ts[u]}, x[o], acc[o]);            // SGPR tap
        if (MODE == 1) acc[o] = __builtin_elementwise_fma((v2f){tv[u].x, tv[u].x}, x[o], acc[o]);        // VGPR tap (op_sel)
        if (MODE == 2) { sa[o] = fmaf(ts[u], x[o].x, sa[o]); }                                           // scalar fma, SGPR tap
        if (MODE == 3) acc[o] = __builtin_elementwise_fma(tv[u], x[o], acc[o]);                          // full VGPR pair
      }
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]));
    asm volatile("" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(sa[3]), "+v"(sa[4]), "+v"(sa[5]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 6; i++) r += acc[i].x + acc[i].y + sa[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *out, *taps; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&taps, 64)); CK(hipMalloc(&cyc, 8));
  CK(hipMemset(taps, 0, 64));
  const int iters = 2000;
  const char *names[4] = {"v_pk_fma_f32, SGPR tap", "v_pk_fma_f32, VGPR tap op_sel", "v_fma_f32, SGPR tap", "v_pk_fma_f32, full VGPR pair"};
  for (int threads : {256, 512}) {
    for (int m = 0; m < 4; m++) {
      for (int rep = 0; rep < 2; rep++) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, taps, iters, cyc);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, taps, iters, cyc);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, taps, iters, cyc);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, taps, iters, cyc);
        CK(hipDeviceSynchronize());
      }
      unsigned long long c;
      CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("%d waves/SIMD  %-32s %.2f cycles per instruction (per wave)\n", threads / 256, names[m], (double)c / (iters * 48.0));
    }
  }
  return 0;
}
