// kernels_fused.hpp -- the fused front end: IfResampler stage A -> stage B -> PhaseDiscriminator in ONE
// persistent, wave-specialised kernel, so that the 1 MHz `mid` signal (and, with the discriminator epilogue, the
// 384 kHz IF's second read) never goes through HBM.  Rows a1 + a7 (+ a3/a8 partial sums) of SURVEY.md 8a:
// sfmbase/IfResampler.cpp:37-78, sfmbase/PhaseDiscriminator.cpp:33-46, sfmbase/FmDecode.cpp:95,141-150.
//
// Shape: the 10 MS/s class -- stage A D = 10, NA = 103 (the equiripple design, design.hpp) onto 1 MHz, followed by the
// LB/MB = 48/125, TB = 210 polyphase stage (the k_ifr_poly4 shape).
//
// One 768-lane workgroup per CU owns a CONTIGUOUS run of "macro tiles" (8 periods of stage B = 384 IF samples
// = 1000 mid samples = 10 000 input samples) of one stream and walks it in EPOCHS of half a macro tile
// (500 mid samples, 5 000 input samples, 40 KB).  Wave roles (one s_barrier per epoch, nothing else synchronises):
//   wave 0       loader  : LDS-DMA (global_load_lds_dwordx4) of the input region of epoch e+2 into a 3-slot ring,
//                          80 KB in flight per CU; it never reads LDS, so the compiler puts no wait in its path and
//                          its loads stay in flight across the barriers (s_waitcnt vmcnt(40) by hand)
//   waves 4..11  stage A : quad form (FusedQuad): four lanes share four consecutive outputs, a quarter of the tap
//                          window each, out of the natural-order slot; results go to a 3-window `mid` ring in LDS
//   waves 1..3   stage B : the 48 x 332 banded polyphase matrix as v_mfma_f32_16x16x4_f32 (rows = 16 positions,
//                          columns = 8 periods x (re, im)), one row tile per wave, the 63 live k-steps of a macro tile
//                          spread over the two epochs that follow its last input; then a third each of the
//                          EPILOGUE: IF samples of the finished macro tile -> atan2 / wrapped difference (the
//                          discriminator), float -> double widening, per-block partial sums, coalesced stores
// Arithmetic: every stage-A output is fp32 FMAs in a fixed order (quarter of the window, word, even / odd sample, then
// the quad's reduction tree); stage B is bit-identical to k_ifr_poly4 (an f32 MFMA is a k-ordered fmaf chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmr {

// Everything is call-relative and 32-bit on the device: the host folds the absolute stream positions into a few
// reference values of the call's first epoch (E_ref = 4 T_first - 1).
struct FusedArgs {
  const float2 *iq; long long iq_stride; long long n_valid;   // this call's input (per stream: iq + s * iq_stride)
  const float2 *in_halo; int H_in;                            // last H_in input samples of the previous call
  const float *taps;                                          // device copy of FusedTaps::h (read through scalar loads)
  long long nbase;            // region start (local input index, even) of an epoch whose first output is j = 0:
                              //   nb(E) = nbase + D * jE(E),  nbase = n0 + ca - (NA - 1) - par
  int j_ref;                  // jE(E_ref): call-relative index (m - mA_prev) of the first mid sample of epoch E_ref
  int pos_ref;                // mid-ring position of that sample: (m + 104) mod 3000
  int t3_ref;                 // T_first mod 3 (which of the three ring windows the first macro tile reads)
  int kb_ref;                 // 384 T_first - kB_prev: call-relative IF index of the first sample of the first macro tile
  int count_mid;              // this call produces mid samples j = 0 .. count_mid-1
  float2 *mid; long long mid_stride; int H_mid;               // d_mid = [H_mid halo | data]; only the next call's halo is written
  const float *afrag;                                         // stage-B A fragments (layout of k_ifr_poly4)
  int n_if;                                                   // this call produces IF samples 0 .. n_if-1
  float2 *out; long long out_stride; int out_off;             // IF buffer ([halo | data])
  int n_tiles; int tiles_per_wg;                              // macro tiles of the call and their split over workgroups
  // discriminator epilogue (base == nullptr: IF only)
  double *base; long long base_stride; int base_off;          // MPX as doubles ([halo | data]), FmDecode.cpp:143
  float *dec; long long dec_stride;                           // float copy of the discriminator output (debug tap), may be null
  float nf, bound;                                            // PhaseDiscriminator.cpp:28-30
  StreamState *st;                                            // disc_save in, disc_save_next / disc_save_valid out
  const float *hB_last;                                       // stage-B tap row of position 47 (TB taps): the IF sample before a run
  FusedPart *part;                                            // [S][3 n_tiles] partial sums per 128-sample third of a macro tile
  const int *if_off; const int *if_len; int nb;               // block table (IF index space, this call)
  int part_from;                                              // partial sums are needed from this IF index on (k_stats walks the last ~400 blocks)
  const int *wg_blk0;                                         // [gridDim.x]: block that holds the first IF sample of each workgroup's run
  unsigned long long *dbg;                                    // tools/bench_fused.hip: per wave {busy, total} shader cycles of workgroup 0
};

#ifndef FUSED_DMA_AUX
#define FUSED_DMA_AUX 2      // cache policy bits of the LDS-DMA loads: nt, the input is streamed once (168 vs 190 us for the
                             // bare DMA ring on 1 GiB, tools/bench_fused.hip)
#endif
// cycle counters only in the instrumented ablation builds (s_memtime costs ~100 cycles of latency per read)
#define FUSED_CLK() (DBG ? __builtin_readcyclecounter() : 0ull)
constexpr int kFusedD = 10, kFusedNA = 103;      // the shape the product instantiates (fmradion_amd.hip)
#define FUSED_TAP_PAD 32
#define FUSED_TAP_LEN 232
// Stage-A taps as the quad form reads them (FusedArgs::taps points at a device copy): h[FUSED_TAP_PAD + k] = hA[k], zeros elsewhere.
struct FusedTaps { float h[FUSED_TAP_LEN]; };

template <int D, int NA>
struct FusedShape {
  static constexpr int ME = 500;                 // mid samples per epoch
  static constexpr int EPT = 1000 / ME;          // epochs per macro tile
  static constexpr int RS = D * ME + 144;        // input samples per ring slot (pre-roll NA - D + parity + slack), even
  static constexpr int NPIECE = RS / 2;          // 16-byte pieces per slot
  static constexpr int PRE = (RS - D * ME) / 2;  // pieces a region shares with the one before it
  static_assert(PRE > 64 && PRE <= 128, "fused_fill / fused_copy_preroll handle the shared pieces in DMA instructions 0 and 1");
  static constexpr int NDMA = (NPIECE + 63) / 64;
  static constexpr int NSLOT = 3, AHEAD = NSLOT - 1;   // ring slots; the loader runs AHEAD epochs in front of stage A
  static constexpr int MIDR = 3000, MIDM = 207;  // mid ring: three macro-tile windows + mirror of the first 207
  static constexpr int LDS_BYTES = NSLOT * RS * 8 + (MIDR + MIDM + 1) * 8 + 384 * 8 + 64;
  static_assert(NA - D + 1 + D * ME <= RS, "slot too small");
  static_assert((NA & 1) == 1 && NA + 2 * FUSED_TAP_PAD <= FUSED_TAP_LEN, "tap table");
  static_assert((AHEAD - 1) * NDMA <= 63, "vmcnt is a 6-bit counter");
};

// one barrier per epoch.  LDS traffic only: no wave waits here for its global stores, and the loader's DMA stays
// in flight (a __syncthreads() would drain vmcnt)
__device__ __forceinline__ void fused_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- role: loader ---------------------------------------------------------------------------------------
template <int D, int NA>
// The first 144 samples of a region are the last 144 of the region before it (regions advance by D * ME): only the
// first region of a workgroup is read whole; later ones skip those 72 pieces and the stage-A waves copy them from
// the previous slot (fused_copy_preroll) -- the input crosses HBM once.
__device__ __forceinline__ int fused_fill(const FusedArgs &a, const float2 *xs, const float2 *hs, int jE,
                                          unsigned char *slot, int lane, bool whole) {
  using SH = FusedShape<D, NA>;
  const long long nb = a.nbase + (long long)D * jE;
  if (nb >= 0 && nb + SH::RS <= a.n_valid) {
    const float2 *src = xs + nb + 2 * lane;
    if (whole)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)slot, 16, 0, FUSED_DMA_AUX);
#pragma unroll
    for (int c = 1; c < SH::NDMA; c++) {
      if ((c < SH::NDMA - 1 || 64 * c + lane < SH::NPIECE) && (c > 1 || whole || 64 + lane >= SH::PRE))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 128 * c),
                                         (__attribute__((address_space(3))) void *)(slot + 1024 * c), 16, 0, FUSED_DMA_AUX);
    }
    return whole ? SH::NDMA : SH::NDMA - 1;
  }
  // edge region (start / end of the call): guarded element loads, previous call's tail from in_halo, zeros elsewhere
  float4 *dst = reinterpret_cast<float4 *>(slot);
  for (int p = lane; p < SH::NPIECE; p += 64) {
    float2 v[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const long long n = nb + 2 * p + e;
      v[e] = make_float2(0.f, 0.f);
      if (n < 0) { if (n >= -(long long)a.H_in) v[e] = hs[a.H_in + n]; }
      else if (n < a.n_valid) v[e] = xs[n];
    }
    dst[p] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  return 0;
}

// stage-A side of the above: pieces [D*ME/2, D*ME/2 + PRE) of the slot of this epoch -> pieces [0, PRE) of the next
template <int D, int NA>
__device__ __forceinline__ void fused_copy_preroll(const unsigned char *cur, unsigned char *nxt, int lane) {
  using SH = FusedShape<D, NA>;
  const float4 *src = reinterpret_cast<const float4 *>(cur) + D * SH::ME / 2;
  float4 *dst = reinterpret_cast<float4 *>(nxt);
  dst[lane] = src[lane];
  if (64 + lane < SH::PRE) dst[64 + lane] = src[64 + lane];
}

// wait until at most `young` DMA instructions are outstanding (young = those issued for later epochs; a batch is
// NDMA instructions -- the template argument -- or one more for a whole region)
template <int NDMA>
__device__ __forceinline__ void fused_wait_dma(int young) {
  if (3 * NDMA <= 63 && young >= 3 * NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA <= 63 ? 3 * NDMA : 0) : "memory");
  else if (2 * NDMA <= 63 && young >= 2 * NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA <= 63 ? 2 * NDMA : 0) : "memory");
  else if (young >= NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- role: stage A, quad form (waves 4 .. 11) -----------------------------------------------------------------
// Four neighbouring lanes share four consecutive outputs: lane 4 g + q runs quarter q of the tap window (QS = 13 of
// the 52 word steps for NA = 103; 19 of 76 for the 151-tap Kaiser design of round 2) for outputs 4 g .. 4 g + 3, the quad adds its partial sums with two DPP steps and lane q stores
// output 4 g + q -- a wave stores 64 consecutive mid samples.  Why this shape (tools/bench_ldsread.hip,
// tools/bench_pkfma.hip): a ds_read_b128 costs ~5.4 cycles of the CU's LDS pipe whether 16 or 64 lanes are active, and
// one wave issues a packed FMA only every ~6 cycles.  Against three outputs on 42 lanes of a wave this form reads
// 8 x 34 instead of 8 x 48 words per epoch and issues 152 instead of 228 packed FMAs per wave, with all 64 lanes
// busy, no partial sums in LDS and no extra epoch of latency.  Lane addresses 20 g + 19 q (16-byte words) are distinct
// mod 16 over any 16 consecutive lanes: conflict-free.  Words are consumed in load order (word w feeds output o at
// step w - 5 o), so a word's registers die after its eight FMAs; the quarter's 38 taps stay in VGPR pairs.
template <int D, int NA, int PAR>
struct FusedQuad {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  static constexpr int HD = D / 2, OQ = 4, NSTEP = (PAR + NA - 1) / 2 + 1, QS = NSTEP / 4, NW = QS + (OQ - 1) * HD;
  static_assert(NSTEP % 4 == 0 && D == 10 && (QS & 1) == 1, "quarter windows of equal length; lane addresses 20 g + QS q are distinct mod 16 over 16 lanes for odd QS");
  v2f tp[QS];                               // tap pair of step t of this lane's quarter: (even sample, odd sample) of the word
  __device__ __forceinline__ void load(const float *h, int q) {     // h = a.taps + FUSED_TAP_PAD (zero padded both sides)
#pragma unroll
    for (int t = 0; t < QS; t++) {
      const int k0 = PAR + NA - 1 - 2 * (QS * q + t);
      tp[t] = (v2f){h[k0], h[k0 - 1]};
      asm volatile("" : "+v"(tp[t]));
    }
  }
  template <int W>
  __device__ __forceinline__ void word(v2f (&acc)[OQ][2], const v4f xx) const {
#pragma unroll
    for (int o = 0; o < OQ; o++) {
      const int t = W - HD * o;
      if (t >= 0 && t < QS) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[o][0]) : "v"(tp[t]), "v"((v2f){xx.x, xx.y}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[o][1]) : "v"(tp[t]), "v"((v2f){xx.z, xx.w}));
      }
    }
  }
  // ROT: the i-th word a wave handles is word (i + ROT) mod NW.  A word feeds 1, 2, 3, 4, 3, 2, 1 outputs along the
  // window, so a wave is LDS-bound at both ends and issue-bound in the middle; the two stage-A waves of a SIMD run
  // half a window apart (ROT = 0 / NW / 2) and the FMA density of the pair is flat.
  template <int ROT, int I0, int I1>
  __device__ __forceinline__ void words(v2f (&acc)[OQ][2], const v4f *x) const {
    if constexpr (I0 < I1) { word<(I0 + ROT) % NW>(acc, x[I0]); words<ROT, I0 + 1, I1>(acc, x); }
  }
  template <int ROT, int I0, int I1>
  __device__ __forceinline__ void words(v2f (&acc)[OQ][2], const v4f *, const v4f z) const {     // ablation form
    if constexpr (I0 < I1) { word<(I0 + ROT) % NW>(acc, z); words<ROT, I0 + 1, I1>(acc, nullptr, z); }
  }
  template <int ROT, int G, int PF, int GI>
  __device__ __forceinline__ void groups(v2f (&acc)[OQ][2], const v4f *w, v4f *x) const {
    constexpr int NGRP = (NW + G - 1) / G;
    if constexpr (GI < NGRP) {
#pragma unroll
      for (int t = 0; t < G; t++) { const int i = PF + G * GI + t; if (i < NW) x[i] = w[(i + ROT) % NW]; }
      words<ROT, G * GI, (G * GI + G < NW ? G * GI + G : NW)>(acc, x);
      __builtin_amdgcn_sched_barrier(0);
      groups<ROT, G, PF, GI + 1>(acc, w, x);
    }
  }
  // one epoch: ME outputs from the slot; aw = 0..7
  template <int ROT, int ABL = 0>
  __device__ __forceinline__ void run(const FusedArgs &a, int s, int jE, int pos0, const unsigned char *slot, float2 *midr,
                                      int aw, int lane, bool no_math) const {
    using SH = FusedShape<D, NA>;
    const int jl = 64 * aw + lane;                      // = 4 g + q: the output this lane stores
    if ((jl & ~3) >= SH::ME) return;                    // whole quads only (ME is a multiple of 4)
    const int g = jl >> 2, q = lane & 3;
    const v4f *w = reinterpret_cast<const v4f *>(__builtin_assume_aligned(slot, 16)) + (OQ * HD) * g + QS * q;
    v2f acc[OQ][2];
#pragma unroll
    for (int o = 0; o < OQ; o++) acc[o][0] = acc[o][1] = (v2f){0.f, 0.f};
    if (!no_math) {
      constexpr int G = 4, PF = 8;
      v4f x[NW + G + PF];
      if (ABL & 8) {            // ablation: the FMAs alone (operands from registers)
        v4f z = {1.f, 2.f, 3.f, 4.f};
        asm volatile("" : "+v"(z));
        words<ROT, 0, NW>(acc, &z - 0, z);
      } else if (ABL & 16) {    // ablation: the LDS reads alone
        v4f sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NW; i++) { const v4f t = w[i]; sum += t; }
        acc[0][0] = (v2f){sum.x + sum.z, sum.y + sum.w};
      } else {
#pragma unroll
        for (int i = 0; i < PF && i < NW; i++) x[i] = w[(i + ROT) % NW];
        groups<ROT, G, PF, 0>(acc, w, x);
      }
    }
    // quad reduce-scatter: lane q ends with the total of output q.  Step 1 pairs q with q ^ 2 (each keeps two outputs,
    // hands over its partial sums of the other two), step 2 pairs q with q ^ 1: ((q) + (q^2)) + ((q^1) + (q^3)).
    auto xq = [](v2f v, auto ctrl) {
      constexpr int C = decltype(ctrl)::value;
      return (v2f){__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), C, 0xF, 0xF, true)),
                   __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), C, 0xF, 0xF, true))};
    };
    v2f y[OQ];
#pragma unroll
    for (int o = 0; o < OQ; o++) y[o] = acc[o][0] + acc[o][1];
    const bool hi = (q & 2) != 0, od = (q & 1) != 0;
    v2f ka = hi ? y[2] : y[0], kb2 = hi ? y[3] : y[1];
    const v2f sa = hi ? y[0] : y[2], sb = hi ? y[1] : y[3];
    ka += xq(sa, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    kb2 += xq(sb, std::integral_constant<int, 0x4E>{});
    v2f ke = od ? kb2 : ka;
    const v2f se = od ? ka : kb2;
    ke += xq(se, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    float2 yo = make_float2(ke.x, ke.y);
    const int j = jE + jl;
    if (jE < 0) {                    // (wave-uniform test first: only a call's first epochs reach back)
      if (j < 0) {                   // produced by an earlier call: its tail is the prefix halo of d_mid, older samples are never used
        const int h = j + a.H_mid;
        yo = (h >= 0) ? a.mid[(long long)s * a.mid_stride + h] : make_float2(0.f, 0.f);
      }
    }
    int pos = pos0 + jl;
    if (pos >= SH::MIDR) pos -= SH::MIDR;
    midr[pos] = yo;
    if (pos < SH::MIDM) midr[pos + SH::MIDR] = yo;
    // the next call's stage-B history: the last H_mid mid samples of this call, at their d_mid positions (k_shift_halo re-seats them)
    if (jE + SH::ME > a.count_mid - a.H_mid)
      if (j >= a.count_mid - a.H_mid && j < a.count_mid && j >= 0) a.mid[(long long)s * a.mid_stride + a.H_mid + j] = yo;
  }
};

// ---- role: stage B (a quarter of the k-steps of a macro tile per epoch) -----------------------------------
template <int MT0, int NMT>
struct FusedB {
  typedef float v4f __attribute__((ext_vector_type(4)));
  using SHB = Poly4Shape<48, 125, 210>;
  float afr[NMT][SHB::NK];
  v4f acc[NMT];
  __device__ __forceinline__ void load(const float *afrag, int lane) {
#pragma unroll
    for (int t = 0; t < NMT; t++)
#pragma unroll
      for (int i = 0; i < SHB::NK; i++) afr[t][i] = afrag[((MT0 + t) * SHB::NK + i) * 64 + lane];
  }
  template <int KS0, int KS1>
  __device__ __forceinline__ void run(const float *xb) {
#pragma unroll
    for (int ks = KS0; ks < KS1; ks++) {
      const float b = xb[8 * ks];
#pragma unroll
      for (int t = 0; t < NMT; t++)
        if (ks >= SHB::ks_lo(MT0 + t) && ks <= SHB::ks_hi(MT0 + t))
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t][ks - SHB::ks_lo(MT0 + t)], b, acc[t], 0, 0, 0);
    }
  }
  // phase q = 0 .. NPH-1 of the macro tile whose window starts at ring position p (NPH = epochs per tile).  The first
  // phase shares its epoch with the epilogue of the previous tile, so it gets the fewest of the row tile's 63 live
  // k-steps.  One accumulator, k ascending: bit-identical to k_ifr_poly4.
  template <int NPH>
  __device__ __forceinline__ void epoch(int q, int p, const float2 *midr, float2 *stage, int lane) {
    static_assert(NMT == 1 && (NPH == 2 || NPH == 4), "one row tile per wave");
    constexpr int LO = SHB::ks_lo(MT0), HI = SHB::ks_hi(MT0) + 1;
    constexpr int C1 = LO + (NPH == 4 ? 6 : 16), C2 = LO + 25, C3 = LO + 44;
    const int n = lane & 15, kq = lane >> 4;
    const float *xb = reinterpret_cast<const float *>(midr) + 2 * (p + (n >> 1) * 125 + kq) + (n & 1);
    if (q == 0) {
      acc[0] = (v4f){0.f, 0.f, 0.f, 0.f};
      run<LO, C1>(xb);
    } else if (NPH == 4 && q == 1) {
      run<C1, C2>(xb);
    } else if (NPH == 4 && q == 2) {
      run<C2, C3>(xb);
    } else {
      run<(NPH == 4 ? C3 : C1), HI>(xb);
      float *sf = reinterpret_cast<float *>(stage);
#pragma unroll
      for (int v = 0; v < 4; v++) sf[2 * ((n >> 1) * 48 + 16 * MT0 + 4 * kq + v) + (n & 1)] = acc[0][v];
    }
  }
};

// A 64-entry window of the block table in registers (lane l: block base + l), so that the epilogue's walk along the
// blocks costs v_readlane, not a global load: under the input stream a load takes microseconds.
struct FusedBlkWin {
  int base, end_l, len_l;        // per lane: end = if_off + if_len of block base + lane (INT_MAX past the table)
  __device__ __forceinline__ void load(const FusedArgs &a, int b0, int lane) {
    base = b0;
    const int b = b0 + lane;
    len_l = (b < a.nb) ? a.if_len[b] : 0;
    end_l = (b < a.nb) ? a.if_off[b] + len_l : 0x7fffffff;
  }
  __device__ __forceinline__ int end(int blk) const { return __builtin_amdgcn_readlane(end_l, __builtin_amdgcn_readfirstlane(blk - base)); }
  __device__ __forceinline__ int len(int blk) const { return __builtin_amdgcn_readlane(len_l, __builtin_amdgcn_readfirstlane(blk - base)); }
};

// atan2 for the discriminator: |error| <= 2.9e-7 rad over the plane (rms 7.4e-8, the same as atan2f -- the fp32
// rounding of the quotient dominates; tests/test_gpu_parity.py holds the discriminator to the oracle).  22 VALU
// instructions against ~70 of the library call: the epilogue shares a SIMD with two stage-A waves, and what bounds the
// kernel beside the input stream is VALU issue (~5 cycles per instruction and SIMD, tools/bench_fused.hip).
// atan(t) = t + t s P(s), s = t^2, t in [0, 1]: degree-7 minimax fit of (atan(t) / t - 1) / s.
__device__ __forceinline__ float fused_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx);
  if (mx == 0.f) t = 0.f;                                     // atan2(+-0, +-0)
  const float sq = t * t;
  float p = 0.0026222190354019403f;
  p = fmaf(p, sq, -0.015132431872189045f);
  p = fmaf(p, sq, 0.04112168401479721f);
  p = fmaf(p, sq, -0.07366690784692764f);
  p = fmaf(p, sq, 0.10573924332857132f);
  p = fmaf(p, sq, -0.1418597251176834f);
  p = fmaf(p, sq, 0.1999039649963379f);
  p = fmaf(p, sq, -0.33332985639572144f);
  float r = fmaf(t, p * sq, t);
  if (ay > ax) r = 1.57079637f - r;
  if (__float_as_int(x) < 0) r = 3.14159274f - r;
  r = copysignf(r, y);
  if (__builtin_isunordered(x, y)) r = __builtin_nanf("");   // NaN in -> NaN out (the caller zeroes the difference, Utility.h:336-343)
  return r;
}

// The discriminator of one staged third (PhaseDiscriminator.cpp:33-46, FmDecode.cpp:141-150): lane l owns the CONSECUTIVE
// samples idx0 + 2 l and idx0 + 2 l + 1, so that the IF pair and the MPX pair leave as one 16-byte non-temporal store each
// -- two store instructions per wave and third.  Under the input stream every store queues behind the loader's DMA in
// the CU's memory pipeline: the stores, not the arithmetic, are what the epilogue costs (tools/bench_fused.hip, "no
// global stores": 25 of 250 us; one sample per lane and store: 4 us more, plain instead of nt stores: 3 us more).
// prev0 = normalised phase of the sample before idx0 (wave-uniform); save0 = the previous call's last phase
// (m_save_value), which precedes the call's sample 0.
template <int MT0, int ABL = 0>
__device__ __forceinline__ void fused_epilogue(const FusedArgs &a, int s, const float2 *stage, int kb, int tile_g, int &blk,
                                                FusedBlkWin &win, float prev0, float save0, float2 *os, int lane) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  typedef v4f __attribute__((aligned(8))) v4f_u;
  typedef v2d __attribute__((aligned(8))) v2d_u;
  const int idx0 = 128 * MT0, k0 = kb + idx0;
  if (k0 >= a.n_if || k0 + 128 <= 0) return;
  const v4f xx = reinterpret_cast<const v4f *>(stage + idx0)[lane];
  const float2 x0 = make_float2(xx.x, xx.y), x1 = make_float2(xx.z, xx.w);
  const float inv_nf = 1.0f / a.nf;
  const float ph0 = (ABL & 64) ? x0.x : fused_atan2(x0.y, x0.x) * inv_nf, ph1 = (ABL & 64) ? x1.x : fused_atan2(x1.y, x1.x) * inv_nf;     // V4
  // phase of the sample before the lane's first one: lane l - 1's second sample (DPP wave_shr:1), lane 0 <- prev0
  float pv0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(prev0), __float_as_int(ph1), 0x138, 0xF, 0xF, false));
  float pv1 = ph0;
  const int ka = k0 + 2 * lane, kc = ka + 1;
  if (ka == 0) pv0 = save0;
  if (kc == 0) pv1 = save0;
  auto diff = [&](float ph, float pv) {
    float d = ph - pv;                                                              // V5
    if (d > a.bound) d -= 2 * a.bound;
    if (d < -a.bound) d += 2 * a.bound;
    if (isnan(d)) d = 0.f;                                                          // Utility.h:336-343
    return d;
  };
  const float d0 = diff(ph0, pv0), d1 = diff(ph1, pv1);
  const bool va = ka >= 0 && ka < a.n_if, vc = kc >= 0 && kc < a.n_if;
  double *bs = a.base + (long long)s * a.base_stride + a.base_off;
  if (!((ABL & 128) && d0 != 12345.f)) {
    if (va && vc) {
      v4f_u *po = reinterpret_cast<v4f_u *>(os + ka);
      v2d_u *pb = reinterpret_cast<v2d_u *>(bs + ka);
      const v2d dd = {(double)d0, (double)d1};
      __builtin_nontemporal_store(xx, po);
      __builtin_nontemporal_store(dd, pb);
      if (a.dec) { float *pd = a.dec + (long long)s * a.dec_stride + ka; pd[0] = d0; pd[1] = d1; }
    } else {
      if (va) { os[ka] = x0; bs[ka] = (double)d0; if (a.dec) a.dec[(long long)s * a.dec_stride + ka] = d0; }
      if (vc) { os[kc] = x1; bs[kc] = (double)d1; if (a.dec) a.dec[(long long)s * a.dec_stride + kc] = d1; }
    }
  }
  if (ka == a.n_if - 1) { a.st[s].disc_save_next = ph0; a.st[s].disc_save_valid = 1; }
  if (kc == a.n_if - 1) { a.st[s].disc_save_next = ph1; a.st[s].disc_save_valid = 1; }
  if ((ABL & 256) || k0 + 128 <= a.part_from) return;
  // ---- per-block partial sums: walk to the block of the first valid sample, cut at its end
  const int kf = k0 < 0 ? 0 : k0;
  for (;;) {
    if (blk - win.base >= 64) win.load(a, blk, lane);       // rare: the window is exhausted
    if (blk >= a.nb || win.end(blk) > kf) break;
    blk++;
  }
  int cut = 0x7fffffff, blk1 = -1;
  if (blk < a.nb) {
    cut = win.end(blk);
    if (cut < k0 + 128 && cut < a.n_if) {
      blk1 = blk + 1;
      for (;;) {
        if (blk1 >= a.nb) { blk1 = -1; break; }
        if (blk1 - win.base >= 64) { win.load(a, blk, lane); if (blk1 - win.base >= 64) { while (blk1 < a.nb && a.if_len[blk1] == 0) blk1++; if (blk1 >= a.nb) blk1 = -1; break; } }
        if (win.len(blk1) != 0) break;
        blk1++;
      }
    }
  }
  float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
  auto add = [&](bool valid, int k, float d, float2 x) {
    if (!valid) return;
    const float e = x.x * x.x + x.y * x.y;
    if (k < cut) { sa[0] += d; sa[1] += d * d; sa[2] += e; } else { sb[0] += d; sb[1] += d * d; sb[2] += e; }
  };
  add(va, ka, d0, x0);
  add(vc, kc, d1, x1);
#pragma unroll
  for (int c = 0; c < 3; c++) { sa[c] = wave_sum_dpp(sa[c]); sb[c] = (blk1 >= 0) ? wave_sum_dpp(sb[c]) : 0.f; }
  if (lane == 0) {
    FusedPart pt;
    pt.blk[0] = blk < a.nb ? blk : -1; pt.blk[1] = blk1;
#pragma unroll
    for (int c = 0; c < 3; c++) { pt.sum[0][c] = sa[c]; pt.sum[1][c] = sb[c]; }
    a.part[((long long)s * a.n_tiles + tile_g) * 3 + MT0] = pt;
  }
}

// normalised phase of IF sample k = (first sample of macro tile at ring window p) - 1: position 47 of the period before
// the tile, a plain k-ordered fmaf chain over the TB taps of its row (bit-equal to the MFMA form), on lane 0
__device__ __forceinline__ float fused_prev_phase(const FusedArgs &a, const float2 *midr, int p) {
  float re = 0.f, im = 0.f;
  int pos = p - 3; if (pos < 0) pos += 3000;           // mid sample 1000 T - 107 (the window starts at 1000 T - 104)
  for (int j = 0; j < 210; j++) {
    const float h = a.hB_last[j];
    const float2 x = midr[pos];
    re = fmaf(h, x.x, re); im = fmaf(h, x.y, im);
    if (++pos == 3000) pos = 0;
  }
  return fused_atan2(im, re) * (1.0f / a.nf);
}

template <int EPT, int LAG, int MT0, bool OFF, bool DBG, int ABL = 0>
__device__ __forceinline__ void fused_role_b(const FusedArgs &a, int s, int i0, int t3, int nt, int NE, const float2 *midr, float2 *stage, int lane, int wave) {
  FusedB<MT0, 1> b;
  b.load(a.afrag, lane);
  float2 *os = a.out + (long long)s * a.out_stride + a.out_off;
  fused_barrier();
  int p = 1000 * t3, q = 0;
  int kb = a.kb_ref + 384 * i0;                      // call-relative IF index of the first staged sample
  int tile_g = i0, blk = 0;
  float prev_tile = 0.f;                              // phase of the last sample of the previous tile (wave 1's first sample needs it)
  float save0 = 0.f;
  FusedBlkWin win{};
  if (a.base) { save0 = a.st[s].disc_save; blk = a.wg_blk0[blockIdx.x]; win.load(a, blk, lane); }
  unsigned long long busy = 0, t_begin = FUSED_CLK();
  for (int e = 0; e < NE; e++) {
    const unsigned long long tb = FUSED_CLK();
    if (e == EPT + 1 + LAG && a.base && MT0 == 0) {
      // the sample before this workgroup's first one: previous call (disc_save), or recomputed from the warm-up mid samples
      if (kb <= 0) prev_tile = save0;
      else { float v = 0.f; if (lane == 0) v = fused_prev_phase(a, midr, p); prev_tile = __shfl(v, 0, 64); }
    }
    if (e >= 2 * EPT + 1 + LAG && ((e - 1 - LAG) % EPT) == 0) {  // epilogue of the tile staged at the end of the previous epoch
      if (a.base) {
        // phase of the sample before this wave's third: the previous tile's last sample (wave 0) or staged sample 128 MT0 - 1
        float prev0;
        if (MT0 == 0) prev0 = prev_tile;
        else { const float2 xp = stage[128 * MT0 - 1]; prev0 = fused_atan2(xp.y, xp.x) * (1.0f / a.nf); }
        if (MT0 == 0) { const float2 xl = stage[383]; prev_tile = fused_atan2(xl.y, xl.x) * (1.0f / a.nf); }
        fused_epilogue<MT0, ABL>(a, s, stage, kb, tile_g, blk, win, prev0, save0, os, lane);
      } else {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int idx = 128 * MT0 + t * 64 + lane;
          const int k = kb + idx;
          if (k >= 0 && k < a.n_if) os[k] = stage[idx];
        }
      }
      kb += 384; tile_g++;
    }
    if (!OFF && e >= EPT + 1 + LAG && e <= EPT * nt + EPT + LAG) {
      b.template epoch<EPT>(q, p, midr, stage, lane);
      if (++q == EPT) { q = 0; p = (p == 2000) ? 0 : p + 1000; }
    }
    if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    busy += FUSED_CLK() - tb;
    fused_barrier();
  }
  if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { a.dbg[2 * wave] = busy; a.dbg[2 * wave + 1] = FUSED_CLK() - t_begin; }
}

// Sum the pieces of every block (index order: deterministic) into the per-block statistics k_disc used to write:
// mean / rms of the discriminator output (Utility.h:135-152) and the IF RMS (Utility.h:118-132, FmDecode.cpp:95).
__global__ void k_fused_blk_reduce(const FusedPart *__restrict__ part, int n_tiles, int kb_ref, const int *__restrict__ if_off,
                                   const int *__restrict__ if_len, int nb, float *__restrict__ bb_mean_blk,
                                   float *__restrict__ bb_rms_blk, float *__restrict__ if_rms_blk) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (b >= nb) return;
  const int n = if_len[b];
  if (n == 0) return;
  const int lo = if_off[b], hi = lo + n - 1;
  int g0 = (lo - kb_ref) / 128, g1 = (hi - kb_ref) / 128;        // thirds that hold samples of the block (kb_ref <= 0 <= lo)
  const int ng = 3 * n_tiles;
  if (g1 >= ng) g1 = ng - 1;
  float sd = 0.f, sq = 0.f, se = 0.f;
  for (int g = g0; g <= g1; g++) {
    const FusedPart pt = part[(long long)s * ng + g];
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (pt.blk[h] == b) { sd += pt.sum[h][0]; sq += pt.sum[h][1]; se += pt.sum[h][2]; }
  }
  const float fn = (float)(unsigned)n;
  bb_mean_blk[(long long)s * nb + b] = sd / fn;
  bb_rms_blk[(long long)s * nb + b] = sqrtf(sq / fn);
  if_rms_blk[(long long)s * nb + b] = sqrtf(se / fn);
}

// ABL: ablation mask for tools/bench_fused.hip (0 = product; 1 no stage-A arithmetic, 2 no stage-B MFMAs, 4 no input DMA)
#define FUSED_THREADS 768     // twelve waves: loader, three stage-B waves, eight stage-A waves
template <int D, int NA, int PAR, int ABL = 0>
__global__ __launch_bounds__(FUSED_THREADS) void k_ifr_fused(FusedArgs a) {
  using SH = FusedShape<D, NA>;
  constexpr bool DBG = (ABL & 32) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_f[];
  float2 *midr = reinterpret_cast<float2 *>(lds_f + SH::NSLOT * SH::RS * 8);
  float2 *stage = midr + (SH::MIDR + SH::MIDM + 1);
  const int s = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i0 = blockIdx.x * a.tiles_per_wg;
  const int i1 = min(i0 + a.tiles_per_wg, a.n_tiles);
  if (i0 >= i1) return;
  const int nt = i1 - i0, NE = SH::EPT * nt + SH::EPT + 2, EA = SH::EPT * nt;       // EA = last stage-A epoch
  const int jE0 = a.j_ref + 1000 * i0;                         // first mid sample of epoch 0 (a macro tile = 1000 mid samples)
  const int pos00 = (a.pos_ref + 1000 * (i0 % 3)) % SH::MIDR;
  const int t3 = (a.t3_ref + i0) % 3;
  const float2 *xs = a.iq + (long long)s * a.iq_stride;
  const float2 *hs = a.in_halo + (long long)s * a.H_in;

  if (wave == 0) {
    // ------------------------------------------------------------------ loader
    // cy[k]: DMA instructions of the batch issued k+1 epochs ago ... the batches younger than the one needed next
    int cy[SH::AHEAD];
#pragma unroll
    for (int k = 0; k < SH::AHEAD; k++) cy[k] = 0;
    if (!(ABL & 4)) {
#pragma unroll
      for (int k = 0; k < SH::AHEAD; k++)
        if (k <= EA) { const int c = fused_fill<D, NA>(a, xs, hs, jE0 + SH::ME * k, lds_f + (size_t)k * SH::RS * 8, lane, k == 0); if (k >= 1) cy[k - 1] = c; }
    }
    { int young = 0;
#pragma unroll
      for (int k = 0; k < SH::AHEAD - 1; k++) young += cy[k];
      fused_wait_dma<SH::NDMA - 1>(young); }               // the slot of epoch 0 has landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = SH::AHEAD;
    unsigned long long busy = 0, t_begin = FUSED_CLK();
    for (int e = 0; e < NE; e++) {
      const unsigned long long tb = FUSED_CLK();
      int cn = 0;
      if (!(ABL & 4) && e + SH::AHEAD <= EA)
        cn = fused_fill<D, NA>(a, xs, hs, jE0 + SH::ME * (e + SH::AHEAD), lds_f + (size_t)slot * SH::RS * 8, lane, false);
      slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
      // the slot of epoch e+1 must have landed: everything but the AHEAD-1 younger batches (cy[0] is epoch e+1 itself)
      int young = cn;
#pragma unroll
      for (int k = 1; k < SH::AHEAD - 1; k++) young += cy[k];
      fused_wait_dma<SH::NDMA - 1>(young);
#pragma unroll
      for (int k = 0; k + 1 < SH::AHEAD - 1; k++) cy[k] = cy[k + 1];
      if (SH::AHEAD >= 2) cy[SH::AHEAD - 2] = cn;
      busy += FUSED_CLK() - tb;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { a.dbg[0] = busy; a.dbg[1] = FUSED_CLK() - t_begin; }
  } else if (wave == 1) {
    // ------------------------------------------------------------------ stage B (one row tile per wave) + a third of the epilogue
    fused_role_b<SH::EPT, 0, 0, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else if (wave == 2) {
    fused_role_b<SH::EPT, 0, 1, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else if (wave == 3) {
    fused_role_b<SH::EPT, 0, 2, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else {
    // ------------------------------------------------------------------ stage A, quad form
    const int aw = wave - 4;
    FusedQuad<D, NA, PAR> qa;
    qa.load(a.taps + FUSED_TAP_PAD, lane & 3);
    fused_barrier();
    int slot = 0, pos0 = pos00, jE = jE0;
    unsigned long long busy = 0, t_begin = FUSED_CLK();
    for (int e = 0; e < NE; e++) {
      const unsigned long long tb = FUSED_CLK();
      if (e <= EA) {
        const unsigned char *sl = lds_f + (size_t)slot * SH::RS * 8;
        if (aw < 4) qa.template run<0, (ABL & 24)>(a, s, jE, pos0, sl, midr, aw, lane, (ABL & 1) != 0);
        else qa.template run<FusedQuad<D, NA, PAR>::NW / 2, (ABL & 24)>(a, s, jE, pos0, sl, midr, aw, lane, (ABL & 1) != 0);
        slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
        if (aw == 7 && e < EA) fused_copy_preroll<D, NA>(sl, lds_f + (size_t)slot * SH::RS * 8, lane);
        pos0 += SH::ME; if (pos0 >= SH::MIDR) pos0 -= SH::MIDR;
        jE += SH::ME;
      }
      if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      busy += FUSED_CLK() - tb;
      fused_barrier();
    }
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { a.dbg[2 * wave] = busy; a.dbg[2 * wave + 1] = FUSED_CLK() - t_begin; }
  }
}

}  // namespace fmr
