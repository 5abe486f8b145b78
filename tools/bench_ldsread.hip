// Micro-benchmark: ds_read_b128 / ds_read_b64 rate per CU as a function of the lane stride (in 16-byte words).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NREAD>
__global__ __launch_bounds__(256) void k(float *out, int stride_words, int iters, unsigned long long *cyc, int active) {
  extern __shared__ __attribute__((aligned(16))) v4f lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (v4f){1.f, 2.f, 3.f, (float)i};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // stride_words < 0: the quad pattern of the fused kernel's stage A (lane 4 g + q reads word 20 g + 19 q)
  const v4f *p = lds + (stride_words < 0 ? 20 * (lane >> 2) + 19 * (lane & 3) + wave * 320 : (lane * stride_words + wave * 7) % 4096);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters && lane < active; it++) {
    v4f x[NREAD];
#pragma unroll
    for (int i = 0; i < NREAD; i++) x[i] = p[i];
#pragma unroll
    for (int i = 0; i < NREAD; i++) acc += x[i];
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 500;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
  for (int stride : {1, 2, 3, 4, 5, 7, 8, 15, 16, 25}) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 8192 * 16, 0, out, stride, iters, cyc, 64); CK(hipDeviceSynchronize()); }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("ds_read_b128, lane stride %2d words, 4 waves/CU: %.2f cycles per wave-instruction per wave => %.2f per CU-instr\n", stride, (double)c / (iters * 32.0), (double)c / (iters * 32.0) / 4);
  }
  // active lanes per wave (the rest of the wave is masked off): does a partly filled ds_read_b128 cost less?
  for (int stride : {-1, 15, 10, 20, 21})
    for (int active : {64, 63, 48, 42, 32, 16}) {
      for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 8192 * 16, 0, out, stride, iters, cyc, active); CK(hipDeviceSynchronize()); }
      unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("ds_read_b128, lane stride %2d words, %2d active lanes: %.2f cycles per CU-instr\n", stride, active, (double)c / (iters * 32.0) / 4);
    }
  return 0;
}
