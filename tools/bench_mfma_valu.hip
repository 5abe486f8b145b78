// Micro-benchmark: does an f32 MFMA stream on one wave slow a packed-FMA stream on another wave of the SAME SIMD?
// 8 waves per workgroup (two per SIMD): waves 0-3 run v_pk_fma_f32, waves 4-7 run, per mode, nothing / v_pk_fma_f32 /
// v_mfma_f32_16x16x4_f32 (one dependent chain, as stage B of the fused front end) / four independent MFMA chains.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bench_mfma_valu.bin tools/bench_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *cyc) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    v2f acc[6], x[6];
    for (int i = 0; i < 6; i++) { acc[i] = (v2f){0.f, 0.f}; x[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f - i}; asm volatile("" : "+v"(x[i])); }
    v2f t = {1.0001f, 0.9999f};
    asm volatile("" : "+v"(t));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int o = 0; o < 6; o++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[o]) : "v"(t), "v"(x[o]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 6; i++) r += acc[i].x + acc[i].y;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
  } else if (MODE == 1) {
    v2f acc[6], x[6];
    for (int i = 0; i < 6; i++) { acc[i] = (v2f){0.f, 0.f}; x[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f - i}; asm volatile("" : "+v"(x[i])); }
    v2f t = {1.0001f, 0.9999f};
    asm volatile("" : "+v"(t));
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int o = 0; o < 6; o++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[o]) : "v"(t), "v"(x[o]));
    for (int i = 0; i < 6; i++) r += acc[i].x + acc[i].y;
  } else if (MODE == 2 || MODE == 3) {
    v4f acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x * 1e-3f, b = 1.f;
    asm volatile("" : "+v"(a), "+v"(b));
    // 48 packed FMAs of the other wave take ~48 x 6 cycles; an MFMA 16x16x4 f32 is 32 cycles: 9 per iteration keep pace
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 9; u++) {
        const int c = (MODE == 3) ? (u & 3) : 0;
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      }
    for (int i = 0; i < 4; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  }
  else if (MODE == 4 || MODE == 5) {
    v4f acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)1.f; }
    asm volatile("" : "+v"(a), "+v"(b));
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 9; u++) {
        const int c = (MODE == 5) ? (u % 3) : 0;
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
      }
    for (int i = 0; i < 4; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 2000;
  const char *names[6] = {"other wave of the SIMD idle", "other wave: v_pk_fma_f32 too", "other wave: f32 MFMA, one dependent chain", "other wave: f32 MFMA, four independent chains",
                          "other wave: 9 x mfma_f32_16x16x32_f16, one chain", "other wave: 9 x mfma_f32_16x16x32_f16, three chains"};
  for (int m = 0; m < 6; m++) {
    for (int rep = 0; rep < 2; rep++) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      if (m == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
      CK(hipDeviceSynchronize());
    }
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-48s %.2f cycles per v_pk_fma_f32 of the measured wave\n", names[m], (double)c / (iters * 48.0));
  }
  return 0;
}
