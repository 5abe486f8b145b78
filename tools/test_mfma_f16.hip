// Checks on the GPU before stage B is rebuilt on fp16 MFMAs:
//  1. fragment layout of v_mfma_f32_16x16x32_f16 (A: row = lane & 15, k = 8 (lane >> 4) + e; B: col = lane & 15, same k;
//     D: col = lane & 15, row = 4 (lane >> 4) + reg) against a host product of an asymmetric pair;
//  2. accuracy of the two-term fp16 split (x = h + l, three products hh + hl + lh) against fp64;
//  3. whether ds_read_b128 / ds_read_b64 work at addresses that are only 2-byte aligned, and what they cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/test_mfma_f16.bin tools/test_mfma_f16.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mfma(const float *A /*16x32*/, const float *B /*32x16*/, float *D /*16x16*/, int split) {
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  h8 ah, al, bh, bl;
  for (int e = 0; e < 8; e++) {
    const float a = A[i * 32 + 8 * kq + e], b = B[(8 * kq + e) * 16 + i];
    const _Float16 a1 = (_Float16)a, b1 = (_Float16)b;
    ah[e] = a1; al[e] = (_Float16)(a - (float)a1);
    bh[e] = b1; bl[e] = (_Float16)(b - (float)b1);
  }
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  if (split) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  }
  for (int v = 0; v < 4; v++) D[(4 * kq + v) * 16 + i] = acc[v];
}

__global__ void k_unaligned(const unsigned short *src, unsigned short *dst, int shift, int iters, unsigned long long *cyc, int wide) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = src[i];
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds) + 2u * (unsigned)(threadIdx.x * 24 + shift);   // lane stride 48 B
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u acc = {0, 0, 0, 0}, v;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (wide) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + 0u * it));
    else {
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      v2u a, b;
      asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(addr));
      v = (v4u){a.x, a.y, b.x, b.y};
    }
    acc ^= v;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  (void)r0; (void)r1; (void)r2; (void)r3;
  for (int e = 0; e < 4; e++) { dst[threadIdx.x * 8 + 2 * e] = (unsigned short)(v[e] & 0xffff); dst[threadIdx.x * 8 + 2 * e + 1] = (unsigned short)(v[e] >> 16); }
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = acc.x ^ acc.y ^ acc.z ^ acc.w; }
}

int main() {
  std::mt19937 g(3);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<float> A(16 * 32), B(32 * 16), D(256);
  for (auto &v : A) v = u(g) * 0.2f;
  for (auto &v : B) v = u(g);
  float *dA, *dB, *dD;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  for (int split = 0; split < 2; split++) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD, split);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double e2 = 0, r2 = 0, emax = 0;
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        double ref = 0;
        for (int k = 0; k < 32; k++) ref += (double)A[i * 32 + k] * B[k * 16 + j];
        const double e = D[i * 16 + j] - ref;
        e2 += e * e; r2 += ref * ref; emax = fmax(emax, fabs(e));
      }
    printf("mfma_f32_16x16x32_f16 %s: rel rms error vs fp64 %.3e (max abs %.3e)  -> layout %s\n", split ? "two-term split (hh + hl + lh)" : "fp16 operands as they are",
           sqrt(e2 / r2), emax, sqrt(e2 / r2) < (split ? 1e-6 : 2e-3) ? "as assumed" : "WRONG");
  }
  // fp32 fmaf chain for comparison
  { double e2 = 0, r2 = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double ref = 0; float acc = 0.f; for (int k = 0; k < 32; k++) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; acc = fmaf(A[i * 32 + k], B[k * 16 + j], acc); } e2 += (acc - ref) * (acc - ref); r2 += ref * ref; }
    printf("fp32 fmaf chain (what the f32 MFMA computes): rel rms error vs fp64 %.3e\n", sqrt(e2 / r2)); }
  // unaligned LDS reads
  std::vector<unsigned short> src(8192), dst(512);
  for (int i = 0; i < 8192; i++) src[i] = (unsigned short)i;
  unsigned short *dS, *dT; unsigned long long *cyc;
  CK(hipMalloc(&dS, 8192 * 2)); CK(hipMalloc(&dT, 512 * 2)); CK(hipMalloc(&cyc, 16));
  CK(hipMemcpy(dS, src.data(), 8192 * 2, hipMemcpyHostToDevice));
  for (int wide = 1; wide >= 0; wide--)
    for (int shift = 0; shift < 8; shift++) {
      hipLaunchKernelGGL(k_unaligned, dim3(1), dim3(64), 0, 0, dS, dT, shift, 1000, cyc, wide);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("%s, misaligned by %d elements: FAULT (%s)\n", wide ? "ds_read_b128" : "2 x ds_read_b64", shift, hipGetErrorString(e)); return 0; }
      unsigned long long c[2];
      CK(hipMemcpy(dst.data(), dT, 512 * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
      bool ok = true;
      for (int l = 0; l < 64; l++) for (int e2 = 0; e2 < 8; e2++) ok = ok && dst[l * 8 + e2] == (unsigned short)(l * 24 + shift + e2);
      printf("%s at +%d fp16 elements (%2d bytes) from 16-byte alignment: data %s, %.1f cycles per read (dependent, one wave)\n",
             wide ? "ds_read_b128    " : "2 x ds_read_b64 ", shift, 2 * shift, ok ? "correct" : "WRONG", (double)c[0] / 1000.0);
    }
  return 0;
}
