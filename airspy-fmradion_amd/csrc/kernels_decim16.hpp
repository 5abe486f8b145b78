// kernels_decim16.hpp -- stage A of IfResampler alone (sfmbase/IfResampler.cpp:37-78, first half) in the fused front end's
// matrix-core form, for the shapes whose stage B is a kernel of its own: the R8B resampler class at 10 MS/s (D = 10, NA = 195:
// the specification of r8b::CDSPResampler24, IfResampler.cpp:25-29).  Round 6: k_ifr_decim2<.., 24> -- the vector-ALU form
// with the tile de-interleaved mod D in LDS -- took 0.247 ms per 2^27 input samples (0.54 of the HBM peak).
//
// One 576-lane workgroup per CU owns a contiguous run of EPOCHS (500 mid samples = 5000 input samples) of one stream:
//   wave 0      loader : the fused kernel's LDS-DMA ring (fused_fill_sh: three slots, two epochs ahead, a 16-byte hole after
//                        every 160 samples, the call's ends through the same instructions with a per-lane source)
//   waves 1..8  stage A: Decim16A -- mid[J] = sum_i c[i] x[10 J + i] as the banded product of FusedMfmaA on
//                        v_mfma_f32_16x16x32_f16 (taps and samples as two fp16 terms, three products, fp32 accumulate), with
//                        eleven k-tiles of 32 inputs for the 16 outputs x (150 + 195 + 1) inputs of a column; wave (unit,
//                        component, half) owns 256 consecutive outputs of one component over the k-tiles 0-5 or 6-10 -- TWO
//                        waves per SIMD: a lone wave issues its conversions and its MFMAs one after the other (300 cycles per
//                        k-tile measured in the fused kernel, tools/bench_fused.hip; with four waves and eleven k-tiles each
//                        this kernel ran at 0.229-0.237 ms, under the vector-ALU kernel's 0.247 but far from the ring's 0.2).
//                        The two partial sums of an output cross to the other layout -- interleaved re / im, 16 bytes per lane
//                        -- through a double-buffered staging area, are added there and leave for HBM one epoch later.
// One `s_waitcnt lgkmcnt(0); s_barrier` per epoch.  A column tile that holds a non-finite sample or one beyond fp16's range is
// recomputed with plain fp32 tap loops (exact tap support), as in the fused kernel.
#pragma once
#include "kernels_fused.hpp"

namespace fmr {

template <int D, int NA>
struct Decim16Shape {
  static constexpr int DEC = D;
  static constexpr int ME = 500;
  static constexpr int NKT = (15 * D + NA + 1 + 31) / 32;          // k-tiles of a column: 16 outputs reach over 15 D + NA (+ 1: either parity) inputs
  static constexpr int RS = D * ME + 32 * NKT - 4 * D;            // the last column tile starts at output 496
  static constexpr int NPIECE = RS / 2;
  static constexpr int PRE = (RS - D * ME) / 2;
  static constexpr int PADP = 80;
  static constexpr int NPOS = NPIECE + (NPIECE - 1) / PADP;
  static constexpr int PREPOS = PRE + PRE / PADP;
  static constexpr int CSKIP = PREPOS / 64;
  static constexpr int SLOT_BYTES = NPOS * 16;
  static constexpr int NDMA = (NPOS + 63) / 64;
  static constexpr int NSLOT = 3, AHEAD = NSLOT - 1;
  static constexpr int KSPLIT = (NKT + 1) / 2;                    // k-tiles [0, KSPLIT) and [KSPLIT, NKT)
  static constexpr int STAGE_BYTES = ME * 8;                      // one epoch of partial results, interleaved
  static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + 4 * STAGE_BYTES + 64;
  static_assert(D == 10 && (RS % 2) == 0 && (NA & 1) == 1, "the 10 MS/s shapes, type-I stage A");
  static_assert(NKT <= 11 && 32 * NKT < 3 * 160, "hole arithmetic of Decim16A::run: a column crosses at most two holes");
  static_assert(NDMA <= 63 && NDMA - CSKIP >= 32, "vmcnt is a 6-bit counter; decim16_wait");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  __host__ __device__ static constexpr int pos_of_piece(int p) { return p + p / PADP; }
};

// wait until at most the batch just issued (NDMA - CSKIP instructions, or none) is outstanding
template <class SH>
__device__ __forceinline__ void decim16_wait(int young) {
  if (young >= SH::NDMA - SH::CSKIP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SH::NDMA - SH::CSKIP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D, int NA, int KT0, int KT1>
struct Decim16A {
  using SH = Decim16Shape<D, NA>;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  static constexpr int NKT = SH::NKT, NK = KT1 - KT0;
  h8 ah[NK], al[NK];
  __device__ __forceinline__ void load(const uint4 *afragA, int par, int lane) {
#pragma unroll
    for (int kt = KT0; kt < KT1; kt++) {
      const uint4 h = afragA[((par * NKT + kt) * 2 + 0) * 64 + lane], l = afragA[((par * NKT + kt) * 2 + 1) * 64 + lane];
      __builtin_memcpy(&ah[kt - KT0], &h, 16); __builtin_memcpy(&al[kt - KT0], &l, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (FusedB16::load: not "in flight" inside the epoch loop)
#pragma unroll
    for (int kt = 0; kt < NK; kt++) { asm volatile("" : "+v"(ah[kt])); asm volatile("" : "+v"(al[kt])); }
  }
  // one epoch: this wave's share (k-tiles KT0 .. KT1 - 1) of the unit's 256 outputs of component C, from the slot into the
  // staging area of its half (float index 2 j + C of the epoch)
  template <int C, int PARITY>
  __device__ __forceinline__ void run(const FusedArgs &a, const unsigned char *slot, float *stage, int unit, int lane) const {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int n = lane & 15, kg = lane >> 4;
    const int jl0 = 256 * unit + 16 * n + 4 * kg;            // this lane's outputs: jl0 .. jl0 + 3 (D rows 4 kg + v of column n)
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    const unsigned addr = (unsigned)(size_t)slot + (unsigned)(unit * (8 * 2560 + 16 * 16) + 8 * (160 * n + 8 * kg) + 16 * n);
    float m2048 = -2048.0f;
    asm volatile("" : "+s"(m2048));
    v4f w[2][4];
#define DECIM16_READ(KT, BUF)                                                                                              \
  {                                                                                                                        \
    constexpr int off_ = 256 * (KT) + ((KT) >= 5 ? 16 : 0) + ((KT) >= 10 ? 16 : 0);                                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][0]) : "v"(addr), "n"(off_) : "memory");                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][1]) : "v"(addr), "n"(off_ + 16) : "memory");                \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][2]) : "v"(addr), "n"(off_ + 32) : "memory");                \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][3]) : "v"(addr), "n"(off_ + 48) : "memory");                \
  }
    DECIM16_READ(KT0, KT0 & 1)
    auto step = [&](auto kt_tag) {
      constexpr int kt = decltype(kt_tag)::value, cur = kt & 1;
      if constexpr (kt >= KT0 && kt < KT1) {
        if constexpr (kt + 1 < KT1) { DECIM16_READ(kt + 1, cur ^ 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(w[cur][j]));
        v4u xh, xl;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const v4f xx = w[cur][j];
          unsigned h, l;
          FusedMfmaA<kFusedD, kFusedNA>::split2(C ? xx.y : xx.x, C ? xx.w : xx.z, m2048, h, l);
          xh[j] = h; xl[j] = l;
        }
        h8 bh, bl;
        __builtin_memcpy(&bh, &xh, 16); __builtin_memcpy(&bl, &xl, 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt - KT0], bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt - KT0], bl, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kt - KT0], bh, acc2, 0, 0, 0);
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{});
#undef DECIM16_READ
    if (jl0 >= SH::ME) return;
    v4f yo = (acc0 + (acc1 + acc2) * (1.0f / 2048.0f)) * (1.0f / 8192.0f);
    // exact-support repair (FusedMfmaA::run): a non-finite input sample, or one beyond fp16's range, has made every output of
    // its 16 x 32 (KT1 - KT0) column tile NaN; rare and wave-uniform.  This wave's share of an output: the taps that meet the
    // inputs of its k-tiles -- tap t of the column's row rr sits at column input D rr + PARITY + t
    if (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(yo.x + yo.y + yo.z + yo.w)) != 0) {
#pragma unroll 1
      for (int v = 0; v < 4; v++) {
        float acc = 0.f;
        const int sb = D * (jl0 + v) + PARITY, rr0 = D * (4 * kg + v) + PARITY;
        const int t0 = max(0, 32 * KT0 - rr0), t1 = min(NA, 32 * KT1 - rr0);
#pragma unroll 1
        for (int t = t0; t < t1; t++) {
          const int sm = sb + t;
          acc = fmaf(a.hA[t], *reinterpret_cast<const float *>(slot + 8 * sm + 16 * (sm / 160) + 4 * C), acc);
        }
        yo.x = v == 0 ? acc : yo.x; yo.y = v == 1 ? acc : yo.y; yo.z = v == 2 ? acc : yo.z; yo.w = v == 3 ? acc : yo.w;
      }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) stage[2 * (jl0 + v) + C] = yo[v];
  }
};

// Host side: the tap fragments of both parities, [par][kt][high | low][lane][8 halves] (fused_make_afragA with NKT k-tiles)
template <int D, int NA>
inline void decim16_make_afragA(const float *hA, unsigned short *out /* 2 * NKT * 2 * 64 * 8 */) {
  constexpr int NKT = Decim16Shape<D, NA>::NKT;
  for (int par = 0; par < 2; par++)
    for (int kt = 0; kt < NKT; kt++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 8; e++) {
          const int r = lane & 15, kg = lane >> 4, t = 32 * kt + 8 * kg + e - D * r - par;
          const float c = (t >= 0 && t < NA) ? hA[t] * 8192.0f : 0.f;
          const _Float16 h = (_Float16)c, l = (_Float16)((c - (float)h) * 2048.0f);
          unsigned short hb, lb;
          __builtin_memcpy(&hb, &h, 2); __builtin_memcpy(&lb, &l, 2);
          out[((((size_t)par * NKT + kt) * 2 + 0) * 64 + lane) * 8 + e] = hb;
          out[((((size_t)par * NKT + kt) * 2 + 1) * 64 + lane) * 8 + e] = lb;
        }
}

// FusedArgs as this kernel reads it: iq / iq_stride / n_valid, in_halo / H_in, afragA, hA, zero16, nbase (region start of the
// epoch whose first output is j = 0), count_mid, mid / mid_stride / H_mid (d_mid = [H_mid halo | data]: outputs j = 0 ..
// count_mid - 1 at mid[H_mid + j]), n_tiles = epochs of the call, tiles_per_wg = epochs per workgroup.
#define DECIM16_THREADS 576
#ifndef DECIM16_ABL
#define DECIM16_ABL 0          // diagnostic builds: 1 no stage-A arithmetic, 2 no input DMA
#endif
template <int D, int NA, int PAR>
__global__ __launch_bounds__(DECIM16_THREADS) void k_ifr_decim16(FusedArgs a) {
  using SH = Decim16Shape<D, NA>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_d16[];
  float *stage = reinterpret_cast<float *>(lds_d16 + SH::NSLOT * SH::SLOT_BYTES);       // [epoch parity][half][2 ME]
  const int s = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int e0 = (int)blockIdx.x * a.tiles_per_wg, e1 = min(e0 + a.tiles_per_wg, a.n_tiles);
  if (e0 >= e1) return;
  const int ne = e1 - e0, NE = ne + 1;                      // (the last iteration only stores the last epoch's results)
  const int jE0 = SH::ME * e0;
  const float2 *xs = a.iq + (long long)s * a.iq_stride;
  const float2 *hs = a.in_halo + (long long)s * a.H_in;
  if (wave == 0) {
    // ------------------------------------------------------------------ loader (k_ifr_fused's, AHEAD = 2)
    if (!(DECIM16_ABL & 2)) {
      fused_fill_sh<SH>(a, xs, hs, jE0, lds_d16, lane, true);
      if (ne > 1) fused_fill_sh<SH>(a, xs, hs, jE0 + SH::ME, lds_d16 + SH::SLOT_BYTES, lane, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the slots of epochs 0 and 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = SH::AHEAD;
    for (int e = 0; e < NE; e++) {
      int cn = 0;
      if (!(DECIM16_ABL & 2) && e + SH::AHEAD < ne) cn = fused_fill_sh<SH>(a, xs, hs, jE0 + SH::ME * (e + SH::AHEAD), lds_d16 + (size_t)slot * SH::SLOT_BYTES, lane, false);
      slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
      decim16_wait<SH>(cn);                                 // the slot of epoch e + 1 has landed: everything but the batch just issued
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    // ------------------------------------------------------------------ stage A on the fp16 matrix cores
    // wave 1 + aw: component aw & 1, unit (aw >> 1) & 1, k-tile half aw >> 2 -- the two halves of a (unit, component) share a SIMD
    const int aw = wave - 1, unit = (aw >> 1) & 1;
    float2 *ms = a.mid + (long long)s * a.mid_stride + a.H_mid;
    auto role = [&](auto &qa, auto half_tag) {
      constexpr int half = decltype(half_tag)::value;
      qa.load(a.afragA, PAR, lane);
      fused_barrier();
      int slot = 0;
      for (int e = 0; e < NE; e++) {
        if (half == 1 && e >= 1) {
          // the previous epoch's results: 250 pieces of two samples, a lane each; the two partial sums are added here
          const int p = (aw - 4) * 64 + lane;
          if (p < SH::ME / 2) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef float v2f __attribute__((ext_vector_type(2)));
            const float *sp = stage + ((e - 1) & 1) * 4 * SH::ME;
            const v4f v = reinterpret_cast<const v4f *>(sp)[p] + reinterpret_cast<const v4f *>(sp + 2 * SH::ME)[p];
            const int j = jE0 + SH::ME * (e - 1) + 2 * p;
            if (j < a.count_mid) *FUSED_GPTR(v2f, ms + j) = (v2f){v.x, v.y};
            if (j + 1 < a.count_mid) *FUSED_GPTR(v2f, ms + j + 1) = (v2f){v.z, v.w};
          }
        }
        if (e < ne) {
          const unsigned char *sl = lds_d16 + (size_t)slot * SH::SLOT_BYTES;
          float *st = stage + ((e & 1) * 2 + half) * 2 * SH::ME;
          if (DECIM16_ABL & 1) {}
          else if (aw & 1) qa.template run<1, PAR>(a, sl, st, unit, lane);
          else qa.template run<0, PAR>(a, sl, st, unit, lane);
          slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
          if (aw == 7 && e + 1 < ne) fused_copy_preroll_sh<SH>(sl, lds_d16 + (size_t)slot * SH::SLOT_BYTES, lane);
        }
        fused_barrier();
      }
    };
    if (aw < 4) { Decim16A<D, NA, 0, SH::KSPLIT> qa; role(qa, std::integral_constant<int, 0>{}); }
    else { Decim16A<D, NA, SH::KSPLIT, SH::NKT> qa; role(qa, std::integral_constant<int, 1>{}); }
  }
}

}  // namespace fmr
