// kernels_fused.hpp -- the fused front end: IfResampler stage A -> stage B -> PhaseDiscriminator in ONE
// persistent, wave-specialised kernel, so that the 1 MHz `mid` signal (and, with the discriminator epilogue, the
// 384 kHz IF's second read) never goes through HBM.  Rows a1 + a7 (+ a3/a8 partial sums) of SURVEY.md 8a:
// sfmbase/IfResampler.cpp:37-78, sfmbase/PhaseDiscriminator.cpp:33-46, sfmbase/FmDecode.cpp:95,141-150.
//
// Shape: the 10 MS/s class -- stage A D = 10, NA = 103 (the equiripple design, design.hpp) onto 1 MHz, followed by the
// LB/MB = 48/125, TB = 210 polyphase stage (the k_ifr_poly4 shape).
//
// One 512-lane workgroup per CU (eight waves, two per SIMD; 152 KB of the CU's 160 KB of LDS) owns a CONTIGUOUS run of
// "macro tiles" (8 periods of stage B = 384 IF samples = 1000 mid samples = 10 000 input samples) of one stream and walks
// it in EPOCHS of half a macro tile (500 mid samples, 5 000 input samples, 40 KB).  The runs are cut by weight on the host
// (a tile whose epilogue writes per-block partial sums counts 1.10: fmradion_amd.hip, run_tables).  Wave roles -- one
// `s_waitcnt lgkmcnt(0); s_barrier` per epoch (fused_barrier), nothing else synchronises:
//   wave 0       loader  : LDS-DMA (global_load_lds_dwordx4, nt) of the input region two epochs ahead into a 3-slot ring
//                          (FusedShape: a 16-byte hole after every 160 samples keeps stage A's sixteen column windows on
//                          sixteen bank quads); it never reads LDS, its loads stay in flight across the barriers and a slot
//                          is released by `s_waitcnt vmcnt(n)` by hand (fused_wait_upto).  The epochs at the two ends of a
//                          call take the same instructions with a per-lane source: samples, in_halo or a zero block (fused_fill)
//   waves 4..7   stage A : FusedMfmaA -- mid[J] = sum_i c[i] x[10 J + i] as a banded product on v_mfma_f32_16x16x32_f16:
//                          A = the taps (16 outputs x 256 inputs, fragments from the host), B = sixteen column tiles of 256
//                          input samples of one component; taps and samples as two fp16 terms (x = xh + xl / 2048, the low
//                          term formed inside v_fma_mixlo/hi_f16), three products, fp32 accumulate; wave (unit, component)
//                          owns 256 consecutive outputs, k-tile kt + 1 read while kt is converted and multiplied.  Results
//                          go to the `mid` ring: three macro-tile windows in LDS as four fp16 planes (high / low x re / im)
//   waves 1..3   stage B : FusedB16 -- the 48 x 250 banded polyphase matrix on the same instruction (rows = 16 positions of
//                          the wave's row tile, columns = 8 periods x (re, im), nine k-tiles of 32 ring positions), then a
//                          third each of the EPILOGUE (fused_epilogue): IF samples of the finished macro tile -> 22-instruction
//                          atan2, wrapped difference through DPP (the discriminator), per-block partial sums for the
//                          statistics, and 8 bytes per IF sample of stores: the MPX as the float it is and |x|^2 for the
//                          IF AGC's state solve (an IF FIR or the equaliser behind the kernel: the IF pair instead)
// Arithmetic: fp32 accumulation inside the MFMA (its internal summation order); 3.1e-7 relative RMS from the fp64 oracle at
// the IF (tolerance 2e-6).  A tile that holds a non-finite value or a sample beyond fp16's range is recomputed with plain
// fp32 tap loops (exact tap support, tests/test_gpu_fused_levels.py).  Results are a function of the absolute sample index:
// neither the cut into calls nor into workgroups changes a bit (tests/test_gpu_pipeline.py).
// History of the forms this replaced (768 lanes, the quad form of stage A, stage B on the f32 MFMA): NOTEBOOK.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmr {

// Everything is call-relative and 32-bit on the device: the host folds the absolute stream positions into a few
// reference values of the call's first epoch (E_ref = 4 T_first - 1).
struct FusedArgs {
  const float2 *iq; long long iq_stride; long long n_valid;   // this call's input (per stream: iq + s * iq_stride)
  const float2 *in_halo; int H_in;                            // last H_in input samples of the previous call
  const uint4 *afragA;                                        // stage-A tap fragments of both parities (fused_make_afragA)
  const float *hA, *hB;                                       // the plain tap tables (hA[NA], hB[48][210]): the exact-support repair of tiles that hold a non-finite value
  long long nbase;            // region start (local input index, even) of an epoch whose first output is j = 0:
                              //   nb(E) = nbase + D * jE(E),  nbase = n0 + ca - (NA - 1) - par
  int j_ref;                  // jE(E_ref): call-relative index (m - mA_prev) of the first mid sample of epoch E_ref
  int pos_ref;                // mid-ring position of that sample: (m + 104) mod 3000
  int t3_ref;                 // T_first mod 3 (which of the three ring windows the first macro tile reads)
  int kb_ref;                 // 384 T_first - kB_prev: call-relative IF index of the first sample of the first macro tile
  int count_mid;              // this call produces mid samples j = 0 .. count_mid-1
  float2 *mid; long long mid_stride; int H_mid;               // d_mid = [H_mid halo | data]; only the next call's halo is written
  const uint4 *afragB; float hB_inv_scale;                     // stage-B tap fragments (fused_make_afragB) and the inverse of their scale
  int n_if;                                                   // this call produces IF samples 0 .. n_if-1
  float2 *out; long long out_stride; int out_off;             // IF buffer ([halo | data])
  int n_tiles; int tiles_per_wg;                              // macro tiles of the call and their split over workgroups
  // discriminator epilogue (base == nullptr: IF only)
  fm_mpx_t *base; long long base_stride; int base_off;        // MPX ([halo | data]), FmDecode.cpp:143
  float *nrm; long long nrm_stride; int nrm_off;              // |x|^2 of the IF samples for the AGC's state solve (then `out` may be null: nobody reads the IF)
  float *dec; long long dec_stride;                           // float copy of the discriminator output (debug tap), may be null
  float nf, bound;                                            // PhaseDiscriminator.cpp:28-30
  StreamState *st;                                            // disc_save in, disc_save_next / disc_save_valid out
  const float *hB_last;                                       // stage-B tap row of position 47 (TB taps): the IF sample before a run
  FusedPart *part;                                            // [S][3 n_tiles] partial sums per 128-sample third of a macro tile
  const int *if_off; const int *if_len; int nb;               // block table (IF index space, this call)
  int part_from;                                              // partial sums are needed from this IF index on (k_stats walks the last ~400 blocks)
  const float2 *zero16;                                       // sixteen zero bytes in device memory (the loader's source beyond the call's ends)
  const int *wg_tile0;                                        // [gridDim.x + 1]: first macro tile of each workgroup's run (null: tiles_per_wg for all)
  const int *wg_blk0;                                         // [gridDim.x]: block that holds the first IF sample of each workgroup's run
  unsigned long long *dbg;                                    // tools/bench_fused.hip: per wave {busy, total} shader cycles of workgroup 0
  float *mid32;                                               // [workgroup][2][3000]: fp32 copies of mid-ring samples fp16 cannot hold (FusedRing::at32), written and read on the repair paths only
  unsigned long long *stamps;                                 // FMR_FE_STAMPS=1 (diagnostics): per workgroup {start, end} of the constant 100 MHz clock and the hardware id
  float fir_c0; int fir_order;                                // Poly4FirDiscEpi: the IF filter's lag-0 tap and its order (the head outputs of a block have no lag 0, Filter.cpp:57-68)
};

#ifndef FUSED_DMA_AUX
#define FUSED_DMA_AUX 2      // cache policy bits of the LDS-DMA loads: nt, the input is streamed once (168 vs 190 us for the
                             // bare DMA ring on 1 GiB, tools/bench_fused.hip)
#endif
// cycle counters only in the instrumented ablation builds (s_memtime costs ~100 cycles of latency per read)
#ifndef FUSED_DMA_CAP
#define FUSED_DMA_CAP 0      // > 0: the loader keeps at most this many DMA instructions outstanding (it then waits at s_waitcnt, not in the issue queue)
#endif
// The epilogue's stores are plain (write-back through L2), not non-temporal: with 8 bytes per IF sample left, letting L2 gather
// the lines costs the input stream less than streaming them out in 512-byte pieces (tools/bench_fused.hip: 223 against 231 us).
// ... and GLOBAL stores, said so: behind the reinterpret_cast to an under-aligned vector type the compiler no longer knew the
// address space and emitted FLAT stores (rounds 2-5).  A flat instruction counts in lgkmcnt as well as in vmcnt, so the
// `s_waitcnt lgkmcnt(0)` in front of every epoch's barrier (fused_barrier) was also a wait for the epilogue's stores to pass
// the address check of the memory pipeline -- where they queue behind the loader's DMA.
#define FUSED_GPTR(T, p) ((__attribute__((address_space(1))) T *)(p))
#ifdef FUSED_NT_STORES
#define FUSED_STORE(T, v, p) __builtin_nontemporal_store((v), FUSED_GPTR(T, p))
#else
#define FUSED_STORE(T, v, p) (*FUSED_GPTR(T, p) = (v))
#endif
#ifndef FUSED_B_PRIO
#define FUSED_B_PRIO 0       // s_setprio of the stage-B / epilogue waves
#endif
#define FUSED_CLK() (DBG ? __builtin_readcyclecounter() : 0ull)
constexpr int kFusedD = 10, kFusedNA = 103;      // the shape the product instantiates (fmradion_amd.hip)

template <int D, int NA>
struct FusedShape {
  static constexpr int DEC = D;
  static constexpr int ME = 500;                 // mid samples per epoch
  static constexpr int EPT = 1000 / ME;          // epochs per macro tile
  // Input samples per ring slot.  Stage A runs on the matrix cores in column tiles of 16 outputs x 256 inputs (FusedMfmaA): the
  // last column of an epoch starts at output 496 and reads the 256 samples from 4960 on -- real samples of the next
  // region, so that the structural zeros of the banded tap matrix never meet stale LDS contents.
  static constexpr int RS = D * ME + 216;
  static constexpr int NPIECE = RS / 2;          // 16-byte pieces (two samples) per slot
  static constexpr int PRE = (RS - D * ME) / 2;  // pieces a region shares with the one before it
  // LDS layout of a slot: a 16-byte hole after every PADP pieces (160 samples = the distance between two column tiles),
  // piece p at position p + p / PADP.  The sixteen columns a quarter-wave reads at once then sit 81 positions apart --
  // sixteen different bank quads -- where 80 positions (1280 bytes) put all of them on the same one.
  static constexpr int PADP = 80;
  static constexpr int NPOS = NPIECE + (NPIECE - 1) / PADP;
  static constexpr int PREPOS = PRE + PRE / PADP;  // position of the first piece a region does not share (below it: shared pieces and their holes)
  static constexpr int SLOT_BYTES = NPOS * 16;
  static constexpr int CSKIP = PREPOS / 64;        // DMA instructions that hold shared pieces only (fused_fill skips them, fused_copy_preroll fills their positions)
  static_assert(CSKIP == 1, "the 103-tap shape");
  static constexpr int NDMA = (NPOS + 63) / 64;
  static constexpr int NSLOT = 3, AHEAD = NSLOT - 1;   // ring slots; the loader runs AHEAD epochs in front of stage A
  // The mid signal between the stages: a ring of three macro-tile windows (3000 samples) in LDS as FOUR fp16 planes -- high
  // and (2048 x) low term of the real and of the imaginary part, in that order -- because both stages run on the fp16 matrix
  // cores (FusedMfmaA / FusedB16).  A stage-B column is a period of 125 mid samples; in the ring a period takes PSTR = 136
  // positions (11 unused, zeroed once), so that every column starts on a 16-byte boundary of its plane AND the sixteen
  // columns a quarter-wave reads at once (8 periods x 2 components, the planes 408 sixteen-byte units apart) fall on sixteen
  // different bank quads: sample s of the ring sits at position s + 11 (s / 125).
  static constexpr int MIDR = 3000, PSTR = 136, TILEP = 8 * PSTR, RINGP = 3 * TILEP, PLANE_BYTES = 2 * RINGP;
  static constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + 4 * PLANE_BYTES + 384 * 8 + 64;
  static_assert((PLANE_BYTES / 16) % 16 == 8 && (RINGP % 8) == 0 && (MIDR % 4) == 0 && (ME % 4) == 0, "mid ring");
  static_assert(D == 10 && (RS % 2) == 0 && RS >= D * 496 + 256, "slot too small for the last column tile");
  static_assert((NA & 1) == 1, "type-I stage A");
  static_assert((AHEAD - 1) * NDMA <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  __host__ __device__ static constexpr int pos_of_piece(int p) { return p + p / PADP; }
};

// the constant 100 MHz clock (wall_clock64() of the HIP headers costs the kernel a private segment)
__device__ __forceinline__ unsigned long long fused_realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// FMR_FE_STAMPS=1: a one-thread kernel in front of and behind the fused launch on its stream: where on the constant clock the
// stream reached them (the launch's own dispatch time stamps are not on that clock's epoch)
__global__ void k_fused_stamp(unsigned long long *p, int cyc_off) {
  // cyc_off != 0: the shader clock at this instant too -- cycles of this compute unit's counter over ~1 us of the constant
  // clock (the counters of different XCDs are not comparable, so both reads are taken here), in kHz
  const unsigned long long t = fused_realtime();
  if (cyc_off) {
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long t1 = t;
    while (t1 - t < 100) t1 = fused_realtime();
    const unsigned long long c1 = __builtin_readcyclecounter();
    p[cyc_off] = (c1 - c0) * 100000ull / (t1 - t);
  }
  *p = t;
}

// fmr_probe_shader_clock: cycles of this compute unit's counter over `ticks` ticks of the constant 100 MHz clock
__global__ void k_probe_clock(unsigned long long *out, int ticks) {
  const unsigned long long t0 = fused_realtime(), c0 = __builtin_readcyclecounter();
  unsigned long long t1 = t0;
  while (t1 - t0 < (unsigned long long)ticks) t1 = fused_realtime();
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
}

// one barrier per epoch.  LDS traffic only: no wave waits here for its global stores, and the loader's DMA stays
// in flight (a __syncthreads() would drain vmcnt)
__device__ __forceinline__ void fused_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the mid ring (FusedShape) --------------------------------------------------------------------------
struct FusedRing {
  using SH = FusedShape<kFusedD, kFusedNA>;
  __device__ __forceinline__ static int pos_of(int s) { return s + 11 * (s / 125); }          // ring sample 0 .. 2999 -> position
  // the value of ring sample s (any integer: wrapped) of component c as the matrix cores see it: high + low / 2048
  __device__ __forceinline__ static float at(const unsigned char *ring, int c, int s) {
    s %= SH::MIDR; if (s < 0) s += SH::MIDR;
    const int p = pos_of(s);
    const _Float16 *h = reinterpret_cast<const _Float16 *>(ring + c * SH::PLANE_BYTES), *l = reinterpret_cast<const _Float16 *>(ring + (2 + c) * SH::PLANE_BYTES);
    return (float)h[p] + (float)l[p] * (1.0f / 2048.0f);
  }
  // ... on the repair paths: a mid sample beyond fp16's range (|y| > 65504: its high term is inf) was also stored as the float
  // it is, in this workgroup's part of FusedArgs::mid32, by the stage-A wave that produced it (FusedMfmaA::run) -- an epoch
  // or more ago, behind a vmcnt(0) wait and a barrier; read past the CU's vector cache, which may hold an older line.
  __device__ __forceinline__ static float at32(const unsigned char *ring, const float *mid32, int c, int s) {
    s %= SH::MIDR; if (s < 0) s += SH::MIDR;
    const float v = at(ring, c, s);
    if (__builtin_isfinite(v)) return v;
    return __hip_atomic_load(mid32 + c * SH::MIDR + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ static float *mid32_of(const FusedArgs &a) { return a.mid32 + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * SH::MIDR); }
};

// ---- role: loader ---------------------------------------------------------------------------------------
// The first PRE pieces of a region are the last PRE of the region before it (regions advance by D * ME samples): only the
// first region of a workgroup is read whole; later ones skip them and a stage-A wave copies them from the previous slot
// (fused_copy_preroll) -- the input crosses HBM once.  Lane l of DMA instruction c fills position 64 c + l; the lane
// that lands on a hole fetches its neighbour's piece again (same cache line, never read from LDS).
// (SH: FusedShape, or Decim16Shape of kernels_decim16.hpp -- the same ring under the stage-A-only kernel of the R8B class)
template <class SH>
__device__ __forceinline__ int fused_fill_sh(const FusedArgs &a, const float2 *xs, const float2 *hs, int jE,
                                             unsigned char *slot, int lane, bool whole) {
  constexpr int D = SH::DEC;
  const long long nb = a.nbase + (long long)D * jE;
  if (nb >= 0 && nb + SH::RS <= a.n_valid) {
    const float2 *src = xs + nb;
    int issued = 0;
#pragma unroll
    for (int c = 0; c < SH::NDMA; c++) {
      if (c < SH::CSKIP && !whole) continue;
      issued++;
      const int pos = 64 * c + lane;
      const int q = pos / (SH::PADP + 1), hole = (pos % (SH::PADP + 1)) == SH::PADP;
      const int piece = pos - q - hole;
      if (FUSED_DMA_CAP > 0 && !whole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FUSED_DMA_CAP > 0 ? FUSED_DMA_CAP - 1 : 0) : "memory");
      if ((c < SH::NDMA - 1 || pos < SH::NPOS) && (c > SH::CSKIP || whole || pos >= SH::PREPOS))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * piece),
                                         (__attribute__((address_space(3))) void *)(slot + 1024 * c), 16, 0, FUSED_DMA_AUX);
    }
    return issued;
  }
  // edge region (start / end of the call): the previous call's tail from in_halo, zeros beyond the call.  The same DMA
  // instructions with a per-lane source -- the call's samples, in_halo, or sixteen zero bytes in device memory -- as long
  // as no 16-byte piece straddles one of the two borders (the region starts on an even sample: true when H_in and the call's
  // length are even).  Round 6: as guarded element loads with a wait behind them (below), one such epoch took 7-20 us under
  // the other workgroups' input streams, and the first and the last run of every call set the launch's end.
  if (a.zero16 && !(a.H_in & 1) && !(a.n_valid & 1) && !(nb & 1)) {
    int issued = 0;
#pragma unroll
    for (int c = 0; c < SH::NDMA; c++) {
      if (c < SH::CSKIP && !whole) continue;
      issued++;
      const int pos = 64 * c + lane;
      const int q = pos / (SH::PADP + 1), hole = (pos % (SH::PADP + 1)) == SH::PADP;
      const int piece = pos - q - hole;
      const long long n0 = nb + 2 * piece;
      const float2 *gp = (n0 >= 0 && n0 + 2 <= a.n_valid) ? xs + n0 : (n0 < 0 && n0 >= -(long long)a.H_in) ? hs + (a.H_in + n0) : a.zero16;
      if ((c < SH::NDMA - 1 || pos < SH::NPOS) && (c > SH::CSKIP || whole || pos >= SH::PREPOS))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                         (__attribute__((address_space(3))) void *)(slot + 1024 * c), 16, 0, FUSED_DMA_AUX);
    }
    return issued;
  }
  // (odd lengths: element loads, eight pieces per lane in flight at a time, from clamped addresses)
  float4 *dst = reinterpret_cast<float4 *>(slot);
  constexpr int NIT = (SH::NPIECE + 63) / 64, UB = 8;
#pragma unroll 1
  for (int b0 = 0; b0 < NIT; b0 += UB) {
    float2 v[UB][2];
#pragma unroll
    for (int u = 0; u < UB; u++) {
      const int p = lane + 64 * (b0 + u);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const long long n = nb + 2 * p + e;
        const bool in_h = n < 0 && n >= -(long long)a.H_in, in_x = n >= 0 && n < a.n_valid;
        const float2 *src = in_h ? hs + (a.H_in + n) : xs + (in_x ? n : 0);
        const float2 t = *src;
        v[u][e] = (in_h || in_x) ? t : make_float2(0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; u++) {
      const int p = lane + 64 * (b0 + u);
      if (p < SH::NPIECE) dst[SH::pos_of_piece(p)] = make_float4(v[u][0].x, v[u][0].y, v[u][1].x, v[u][1].y);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  return 0;
}
template <int D, int NA>
__device__ __forceinline__ int fused_fill(const FusedArgs &a, const float2 *xs, const float2 *hs, int jE, unsigned char *slot, int lane, bool whole) {
  return fused_fill_sh<FusedShape<D, NA>>(a, xs, hs, jE, slot, lane, whole);
}

// stage-A side of the above: pieces [D*ME/2, D*ME/2 + PRE) of the slot of this epoch -> pieces [0, PRE) of the next
template <class SH>
__device__ __forceinline__ void fused_copy_preroll_sh(const unsigned char *cur, unsigned char *nxt, int lane) {
  const float4 *src = reinterpret_cast<const float4 *>(cur);
  float4 *dst = reinterpret_cast<float4 *>(nxt);
  constexpr int P0 = SH::DEC * SH::ME / 2;
#pragma unroll
  for (int g = 0; 64 * g < SH::PRE; g++)
    if (64 * (g + 1) <= SH::PRE || 64 * g + lane < SH::PRE) dst[SH::pos_of_piece(64 * g + lane)] = src[SH::pos_of_piece(P0 + 64 * g + lane)];
}
template <int D, int NA>
__device__ __forceinline__ void fused_copy_preroll(const unsigned char *cur, unsigned char *nxt, int lane) {
  fused_copy_preroll_sh<FusedShape<D, NA>>(cur, nxt, lane);
}

// wait until at most `young` DMA instructions are outstanding (young = those issued for later epochs), for the few batch sizes
// the loader issues (a smaller constant is merely conservative)
__device__ __forceinline__ void fused_wait_upto(int young) {
  if (young >= 42) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
  else if (young >= 41) asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
  else if (young >= 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else if (young >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- role: stage A on the fp16 matrix cores (waves 4 .. 7) --------------------------------------------------------
// mid[J] = sum_i c[i] x[10 J + i], c[i] = hA[i - PAR] (103 taps), as a BANDED product on v_mfma_f32_16x16x32_f16:
//   D[r][n] = sum_k A[r][k] B[k][n],   A[r][k] = c[k - 10 r]  (16 outputs x 256 inputs, taps in 40 % of the entries),
//   B[k][n] = x[10 (J0 + 16 n) + k]    (column n = the 16 outputs from J0 + 16 n on, one component of the samples).
// A wave owns a UNIT of 256 consecutive outputs (16 columns) of one component: waves 4 / 5 the outputs 0 .. 255 (re / im),
// waves 6 / 7 the outputs 256 .. 499; eight k-tiles of 32 inputs each.  Both operands are split in two fp16 terms
// (tools/test_mfma_f16.hip, tools/stageA_f16_split.py): x = xh + xl / 2048, c 2^13 = ch + cl / 2048, and
//   mid = (ch xh  +  (ch xl + cl xh) / 2048) / 2^13      -- three products, two fp32 accumulators;
// the low terms are carried times 2048 so that they stay normal fp16 numbers wherever the high term is one (|x| >= 6.1e-5;
// below that the error is bounded by 1.5e-11 absolute; |x| > 65504 becomes inf -> NaN like a NaN sample).  The tap fragments
// come ready-made from the host (FusedArgs::afragA, 64 VGPRs); a B fragment is eight consecutive samples of the slot:
// four ds_read_b128 (the other component rides along), converted in registers -- 20 VALU instructions per k-tile against
// the 104 packed FMAs + 28 reads a wave of the quad form (rounds 2-4) spent on 64 outputs.
// Lane (n = lane & 15, kg = lane >> 4) reads bytes 8 (160 n + 8 kg) + 16 n + [256 kt + 16 (kt >= 5)] + 16 j of the unit: the
// hole after every 160 samples (FusedShape) makes the sixteen columns of a quarter-wave hit sixteen different bank quads.
template <int D, int NA>
struct FusedMfmaA {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  static constexpr int NKT = 8;
  static_assert(D == 10 && NA <= 256 - 150 - 1, "16 outputs x 256 inputs hold the band for either parity");
  h8 ah[NKT], al[NKT];
  __device__ __forceinline__ void load(const uint4 *afragA, int par, int lane) {
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
      const uint4 h = afragA[((par * NKT + kt) * 2 + 0) * 64 + lane], l = afragA[((par * NKT + kt) * 2 + 1) * 64 + lane];
      __builtin_memcpy(&ah[kt], &h, 16); __builtin_memcpy(&al[kt], &l, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (see FusedB16::load)
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) { asm volatile("" : "+v"(ah[kt])); asm volatile("" : "+v"(al[kt])); }
  }
  // Two samples of one component -> packed (high, 2048 x low) fp16 terms: high = rne(x), low = rne(2048 x - 2048 high) -- the
  // difference is formed inside the mixed-precision FMA, one rounding.  Four instructions per pair.
  __device__ __forceinline__ static void split2(float x0, float x1, float m2048, unsigned &hi, unsigned &lo) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_convertvector((v2f){x0, x1}, h2);
    __builtin_memcpy(&hi, &h, 4);
    const v2f xs = (v2f){x0, x1} * (v2f){2048.0f, 2048.0f};
    unsigned l;         // (mixlo leaves the upper half alone, mixhi then writes it)
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "s"(m2048), "v"(xs.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "s"(m2048), "v"(xs.y));
    lo = l;
  }
  // one epoch: the unit's 256 outputs of component C from the slot.  ABL: 1 no arithmetic at all, 8 no LDS reads, 16 reads only
  template <int C, int PARITY, int ABL = 0>
  __device__ __forceinline__ void run(const FusedArgs &a, int s, int jE, int pos0, const unsigned char *slot, unsigned char *ring,
                                      int unit, int lane) const {
    using SH = FusedShape<D, NA>;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int n = lane & 15, kg = lane >> 4;
    const int jl0 = 256 * unit + 16 * n + 4 * kg;            // this lane's outputs: jl0 .. jl0 + 3 (D rows 4 kg + v of column n)
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    if (!(ABL & 1)) {
      const unsigned addr = (unsigned)(size_t)slot + (unsigned)(unit * (8 * 2560 + 16 * 16) + 8 * (160 * n + 8 * kg) + 16 * n);
      float m2048 = -2048.0f;
      asm volatile("" : "+s"(m2048));
      // k-tile kt + 1 is read while k-tile kt is converted and multiplied: one wave per SIMD runs this role, nothing else
      // hides the LDS latency (reads return in order: lgkmcnt(4) = the older four have landed)
      v4f w[2][4];
      v4f sum = {0.f, 0.f, 0.f, 0.f};
#define FUSED_A_READ(KT, BUF)                                                                                              \
  {                                                                                                                        \
    constexpr int off_ = 256 * (KT) + ((KT) >= 5 ? 16 : 0);                                                                \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][0]) : "v"(addr), "n"(off_) : "memory");                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][1]) : "v"(addr), "n"(off_ + 16) : "memory");                \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][2]) : "v"(addr), "n"(off_ + 32) : "memory");                \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[BUF][3]) : "v"(addr), "n"(off_ + 48) : "memory");                \
  }
      if (!(ABL & 8)) FUSED_A_READ(0, 0)
      auto step = [&](auto kt_tag) {
        constexpr int kt = decltype(kt_tag)::value, cur = kt & 1;
        if (!(ABL & 8)) {
          if constexpr (kt + 1 < NKT) { FUSED_A_READ(kt + 1, cur ^ 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < 4; j++) asm volatile("" : "+v"(w[cur][j]));
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) { w[cur][j] = (v4f){(float)(kt + j), 0.5f, 0.25f, 2.f}; asm volatile("" : "+v"(w[cur][j])); }
        }
        if (ABL & 16) {
#pragma unroll
          for (int j = 0; j < 4; j++) sum += w[cur][j];
          return;
        }
        v4u xh, xl;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const v4f xx = w[cur][j];
          unsigned h, l;
          split2(C ? xx.y : xx.x, C ? xx.w : xx.z, m2048, h, l);
          xh[j] = h; xl[j] = l;
        }
        h8 bh, bl;
        __builtin_memcpy(&bh, &xh, 16); __builtin_memcpy(&bl, &xl, 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bl, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kt], bh, acc2, 0, 0, 0);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
#undef FUSED_A_READ
      if (ABL & 16) acc0 = sum;
    }
    if (jl0 >= SH::ME) return;                               // (ME is a multiple of 4: a lane's four outputs are all in or all out)
    v4f yo = (acc0 + (acc1 + acc2) * (1.0f / 2048.0f)) * (1.0f / 8192.0f);
    // A non-finite input sample (or one beyond fp16's range) has made every output of its 16 x 256 column tile NaN: zeros of
    // the banded tap matrix times NaN.  The reference's footprint is the tap support (Utility.h:336-343 only sees the
    // discriminator's output), so a wave that finds a non-finite output -- rare, wave-uniform -- recomputes its unit with
    // plain fp32 FMAs over the NA taps of every output: outputs whose support holds the bad sample stay non-finite, the
    // others are what the quad form of rounds 2-4 produced (1e-7 from the matrix-core result).
    // Samples beyond fp16's range (|x| > 65504: un-normalised FLOAT files, FileSource.cpp:514-528) take the same path: their
    // high term is inf, the products NaN; the fp32 loop below is exact for them.
    if (!(ABL & 1) && __builtin_amdgcn_ballot_w64(!__builtin_isfinite(yo.x + yo.y + yo.z + yo.w)) != 0) {
#pragma unroll 1
      for (int v = 0; v < 4; v++) {
        float acc = 0.f;
        const int sb = D * (jl0 + v) + PARITY;                 // slot-relative sample of tap 0
#pragma unroll 1
        for (int t = 0; t < NA; t++) {
          const int sm = sb + t;
          acc = fmaf(a.hA[t], *reinterpret_cast<const float *>(slot + 8 * sm + 16 * (sm / 160) + 4 * C), acc);
        }
        yo.x = v == 0 ? acc : yo.x; yo.y = v == 1 ? acc : yo.y; yo.z = v == 2 ? acc : yo.z; yo.w = v == 3 ? acc : yo.w;
      }
    }
    if (jE < 0) {                    // (wave-uniform test first: only a call's first epochs reach back)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int j = jE + jl0 + v;
        if (j < 0) {                 // produced by an earlier call: its tail is the prefix halo of d_mid, older samples are never used
          const int h = j + a.H_mid;
          yo[v] = (h >= 0) ? reinterpret_cast<const float *>(a.mid + (long long)s * a.mid_stride + h)[C] : 0.f;
        }
      }
    }
    int pos = pos0 + jl0;            // a multiple of 4, like MIDR: the four samples wrap together
    if (pos >= SH::MIDR) pos -= SH::MIDR;
    // A mid sample the ring's fp16 terms cannot hold (|y| > 65504, or not finite): the wave leaves fp32 copies of its whole
    // unit for stage B's repair path (FusedRing::at32), which is what such a sample sends every tile that reads it to.  Rare
    // and wave-uniform; the stores are acknowledged before the epoch's barrier.
    if (!(ABL & 1) && a.mid32 && __builtin_amdgcn_ballot_w64(!(fmaxf(fmaxf(fabsf(yo.x), fabsf(yo.y)), fmaxf(fabsf(yo.z), fabsf(yo.w))) < 65000.f)) != 0) {
      float *m32 = FusedRing::mid32_of(a) + C * SH::MIDR + pos;
#pragma unroll
      for (int v = 0; v < 4; v++) __hip_atomic_store(m32 + v, yo[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    {
      // into the ring as fp16 terms (high, 2048 x low): what stage B multiplies.  The four samples may straddle the end of a
      // period (125 is odd): four 2-byte stores per plane.
      const int blk = pos / 125, r0 = pos - 125 * blk;
      _Float16 *hp = reinterpret_cast<_Float16 *>(ring + C * SH::PLANE_BYTES), *lp = reinterpret_cast<_Float16 *>(ring + (2 + C) * SH::PLANE_BYTES);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int pi = pos + v + 11 * blk + ((r0 + v >= 125) ? 11 : 0);
        const _Float16 h = (_Float16)yo[v];
        hp[pi] = h;
        lp[pi] = (_Float16)((yo[v] - (float)h) * 2048.0f);
      }
    }
    // the next call's stage-B history: the last H_mid mid samples of this call, at their d_mid positions (k_shift_halo re-seats them)
    if (jE + SH::ME > a.count_mid - a.H_mid) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int j = jE + jl0 + v;
        if (j >= a.count_mid - a.H_mid && j < a.count_mid && j >= 0) reinterpret_cast<float *>(a.mid + (long long)s * a.mid_stride + a.H_mid + j)[C] = yo[v];
      }
    }
  }
};

// Host side of FusedMfmaA: the tap fragments of both parities, [par][kt][high | low][lane][8 halves] (32 KB).
// hA: the NA stage-A taps (symmetric); the A operand of lane (r = lane & 15, kg = lane >> 4) in k-tile kt holds
// c[32 kt + 8 kg + e - 10 r], e = 0 .. 7, c[i] = hA[i - par].
template <int D, int NA>
inline void fused_make_afragA(const float *hA, unsigned short *out /* 2 * 8 * 2 * 64 * 8 */) {
  for (int par = 0; par < 2; par++)
    for (int kt = 0; kt < 8; kt++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 8; e++) {
          const int r = lane & 15, kg = lane >> 4, t = 32 * kt + 8 * kg + e - D * r - par;
          const float c = (t >= 0 && t < NA) ? hA[t] * 8192.0f : 0.f;
          const _Float16 h = (_Float16)c, l = (_Float16)((c - (float)h) * 2048.0f);
          unsigned short hb, lb;
          __builtin_memcpy(&hb, &h, 2); __builtin_memcpy(&lb, &l, 2);
          out[((((size_t)par * 8 + kt) * 2 + 0) * 64 + lane) * 8 + e] = hb;
          out[((((size_t)par * 8 + kt) * 2 + 1) * 64 + lane) * 8 + e] = lb;
        }
}

// ---- role: stage B on the fp16 matrix cores (waves 1 .. 3: one row tile of 16 positions each) ---------------------------
// IF sample (period P, position pp) = sum_j hB[phi(pp)][j] mid[window + 125 P + off(pp) + j]: rows = the 16 positions of the
// wave's row tile, columns = 8 periods x (re, im), k = the POSITIONS of the ring from the row tile's first tap on (K0 = 0,
// 40, 80: multiples of eight) -- nine k-tiles of 32 cover the 250 + 22 positions a row tile's band spans, the unused
// positions between two periods meet zero taps.  Both operands in two fp16 terms, three products, three independent
// accumulators (FusedMfmaA; the taps scaled by a power of two, FusedArgs::hB_inv_scale undoes it).  A B fragment is ONE
// aligned 16-byte read per term: 18 reads and 27 MFMAs of 16 cycles per macro tile and wave, where the f32 form of rounds
// 2-4 (v_mfma_f32_16x16x4_f32, one dependent chain, one 4-byte read per k-step) spent 63 reads and 63 MFMAs of 32 cycles --
// measured ~80 cycles per k-step, 2500 cycles per epoch on the waves that also carry the discriminator epilogue.
template <int MT0>
struct FusedB16 {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  using SH = FusedShape<kFusedD, kFusedNA>;
  static constexpr int NKT = 9;
  static constexpr int off(int pp) { return (pp * 125) / 48; }
  static constexpr int K0 = (off(16 * MT0) / 8) * 8;
  static_assert(off(16 * MT0 + 15) + 209 + 11 * ((off(16 * MT0 + 15) + 209) / 125) < K0 + 32 * NKT, "nine k-tiles hold the row tile's band");
  h8 ah[NKT], al[NKT];
  v4f acc0, acc1, acc2;
  __device__ __forceinline__ void load(const uint4 *afragB, int lane) {
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
      const uint4 h = afragB[((MT0 * NKT + kt) * 2 + 0) * 64 + lane], l = afragB[((MT0 * NKT + kt) * 2 + 1) * 64 + lane];
      __builtin_memcpy(&ah[kt], &h, 16); __builtin_memcpy(&al[kt], &l, 16);
    }
    // The fragments must not stay "results of loads in flight" in the compiler's books: inside the epoch loop it cannot
    // tell them from the epilogue's stores (one counter, vmcnt) and would wait for vmcnt(0) -- i.e. for every store of the
    // previous tile to be acknowledged by memory, microseconds under the input stream -- in front of the first MFMA of
    // every phase (measured: 35 us of a 235 us launch).  Wait here, once, and hand the registers over through an empty asm.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) { asm volatile("" : "+v"(ah[kt])); asm volatile("" : "+v"(al[kt])); }
  }
  template <int KT0, int KT1>
  __device__ __forceinline__ void run(const unsigned char *ring, int base) {
    h8 bh[KT1 - KT0], bl[KT1 - KT0];
#pragma unroll
    for (int kt = KT0; kt < KT1; kt++) {
      int pos = base + 32 * kt;
      if (pos >= SH::RINGP) pos -= SH::RINGP;                       // (the last window runs over the end of the ring)
      bh[kt - KT0] = *reinterpret_cast<const h8 *>(ring + 2 * pos);
      bl[kt - KT0] = *reinterpret_cast<const h8 *>(ring + 2 * pos + 2 * SH::PLANE_BYTES);
    }
#pragma unroll
    for (int kt = KT0; kt < KT1; kt++) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bh[kt - KT0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bl[kt - KT0], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kt], bh[kt - KT0], acc2, 0, 0, 0);
    }
  }
  // The IF sample BEFORE the macro tile whose window starts at ring sample p -- position 47 of the period before it -- with
  // exactly the arithmetic a tile computes it with (row tile 2, the column of period "-1"): a workgroup's first phase
  // difference then does not depend on where the call was cut into workgroups (a plain tap loop differs from the matrix
  // cores' summation by rounding).  The lanes that hold row 15 of column 0 / 1 store re / im into `out`.
  __device__ __forceinline__ void sample_before(int p, const unsigned char *ring, float *out, int lane, const FusedArgs &a) {
    static_assert(MT0 == 2, "position 47 is row 15 of row tile 2");
    const int n = lane & 15, kq = lane >> 4;
    int base = (p / 1000) * SH::TILEP + ((n >> 1) - 1) * SH::PSTR + K0 + 8 * kq;      // every column one period early
    if (base < 0) base += SH::RINGP;
    const unsigned char *plane = ring + (n & 1) * SH::PLANE_BYTES;
    acc0 = acc1 = acc2 = (v4f){0.f, 0.f, 0.f, 0.f};
    run<0, 4>(plane, base);
    run<4, NKT>(plane, base);
    const v4f y = (acc0 + (acc1 + acc2) * (1.0f / 2048.0f)) * a.hB_inv_scale;
    float r = y[3];
    if ((n >> 1) == 0 && kq == 3 && !__builtin_isfinite(r)) {          // (the tile's own repair path, for this one sample)
      r = 0.f;
      const float *m32 = FusedRing::mid32_of(a);
      for (int j = 0; j < 210; j++) r = fmaf(a.hB_last[j], FusedRing::at32(ring, m32, n & 1, p - 3 + j), r);
    }
    if ((n >> 1) == 0 && kq == 3) out[n & 1] = r;
  }
  // phase q = 0 / 1 of the macro tile whose window starts at ring sample p (0, 1000 or 2000).  The first phase shares its
  // epoch with the epilogue of the previous tile and takes four of the nine k-tiles.
  template <int NPH>
  __device__ __forceinline__ void epoch(int q, int p, const unsigned char *ring, float2 *stage, int lane, const FusedArgs &a) {
    static_assert(NPH == 2, "two epochs per macro tile");
    const int n = lane & 15, kq = lane >> 4;
    const int base = (p / 1000) * SH::TILEP + (n >> 1) * SH::PSTR + K0 + 8 * kq;
    const unsigned char *plane = ring + (n & 1) * SH::PLANE_BYTES;
    if (q == 0) {
      acc0 = acc1 = acc2 = (v4f){0.f, 0.f, 0.f, 0.f};
      run<0, 4>(plane, base);
    } else {
      run<4, NKT>(plane, base);
      v4f y = (acc0 + (acc1 + acc2) * (1.0f / 2048.0f)) * a.hB_inv_scale;
      // exact-support repair, as in stage A: a non-finite mid sample has poisoned every row of the banded tile whose
      // k-range holds it.  Rare and wave-uniform: position pp of the period = the TB taps of its row over the mid ring.
      if (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(y[0] + y[1] + y[2] + y[3])) != 0) {
#pragma unroll 1
        for (int v = 0; v < 4; v++) {
          const int pp = 16 * MT0 + 4 * kq + v, phi = (pp * 125) % 48, s0 = p + (n >> 1) * 125 + off(pp);
          const float *hr = a.hB + phi * 210, *m32 = FusedRing::mid32_of(a);
          float r = 0.f;
#pragma unroll 1
          for (int j = 0; j < 210; j++) r = fmaf(hr[j], FusedRing::at32(ring, m32, n & 1, s0 + j), r);
          y.x = v == 0 ? r : y.x; y.y = v == 1 ? r : y.y; y.z = v == 2 ? r : y.z; y.w = v == 3 ? r : y.w;       // (y[v] with a run-time v went through scratch memory)
        }
      }
      float *sf = reinterpret_cast<float *>(stage);
#pragma unroll
      for (int v = 0; v < 4; v++) sf[2 * ((n >> 1) * 48 + 16 * MT0 + 4 * kq + v) + (n & 1)] = y[v];
    }
  }
};

// Host side of FusedB16: the tap fragments [row tile][k-tile][high | low][lane][8 halves] (55 KB) and the power of two the
// taps were scaled by.  hB: [48 phases][210 taps]; position pi of a period's window = K0 + 32 kt + 8 kq + e holds mid sample
// m = 125 (pi / 136) + pi % 136 of the window (positions 125 .. 135 of every 136: none).
inline float fused_make_afragB(const float *hB, unsigned short *out /* 3 * 9 * 2 * 64 * 8 */) {
  float tmax = 0.f;
  for (int i = 0; i < 48 * 210; i++) tmax = std::fmax(tmax, std::fabs(hB[i]));
  int e2 = 0;
  if (tmax > 0.f) (void)std::frexp(tmax, &e2);
  const float sc = std::ldexp(1.0f, 10 - e2);               // tmax * sc in [512, 1024)
  for (int mt = 0; mt < 3; mt++) {
    const int K0 = (((16 * mt) * 125 / 48) / 8) * 8;
    for (int kt = 0; kt < 9; kt++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 8; e++) {
          const int pp = 16 * mt + (lane & 15), pi = K0 + 32 * kt + 8 * (lane >> 4) + e, rp = pi % 136;
          const int m = 125 * (pi / 136) + rp, j = m - (pp * 125) / 48;
          const float c = (rp < 125 && j >= 0 && j < 210) ? hB[((pp * 125) % 48) * 210 + j] * sc : 0.f;
          const _Float16 h = (_Float16)c, l = (_Float16)((c - (float)h) * 2048.0f);
          unsigned short hb, lb;
          __builtin_memcpy(&hb, &h, 2); __builtin_memcpy(&lb, &l, 2);
          out[((((size_t)mt * 9 + kt) * 2 + 0) * 64 + lane) * 8 + e] = hb;
          out[((((size_t)mt * 9 + kt) * 2 + 1) * 64 + lane) * 8 + e] = lb;
        }
  }
  return 1.0f / sc;
}

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
// XE: the |x|^2 of the block sums (the IF level, FmDecode.cpp:95) is taken from xe[0 .. 127] instead of the staged samples --
// behind the IF filter the level is that of the filter's INPUT.
template <int MT0, int ABL = 0, bool XE = false>
__device__ __forceinline__ void fused_epilogue(const FusedArgs &a, int s, const float2 *stage, int kb, int tile_g, int &blk,
                                                FusedBlkWin &win, float prev0, float save0, float2 *os, int lane,
                                                const float2 *xe = nullptr) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef v4f __attribute__((aligned(8))) v4f_u;
  typedef v2f __attribute__((aligned(4))) v2f_u;
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
  fm_mpx_t *bs = a.base + (long long)s * a.base_stride + a.base_off;
  float *ns = a.nrm ? a.nrm + (long long)s * a.nrm_stride + a.nrm_off : nullptr;
  const float e0 = x0.x * x0.x + x0.y * x0.y, e1 = x1.x * x1.x + x1.y * x1.y;
  if (!((ABL & 128) && d0 != 12345.f)) {
    if (va && vc) {
      // 8 bytes per IF sample in the product configuration (MPX + |x|^2): what the stores cost is their bytes -- every one
      // queues behind the loader's DMA (tools/bench_fused.hip: 16 B per sample 27 us of the launch, 8 B 10 us, 4 B 4 us)
      if (ABL & 1024) __builtin_nontemporal_store((v4f){d0, d1, e0, e1}, FUSED_GPTR(v4f_u, ns + 2 * ka));      // (harness: what ONE 16-byte store costs)
      else {
      FUSED_STORE(v2f_u, ((v2f){d0, d1}), bs + ka);
      if (ns && !(ABL & 512)) FUSED_STORE(v2f_u, ((v2f){e0, e1}), ns + ka);
      }
      if (os) __builtin_nontemporal_store(xx, FUSED_GPTR(v4f_u, os + ka));
      if (a.dec) { float *pd = a.dec + (long long)s * a.dec_stride + ka; pd[0] = d0; pd[1] = d1; }
    } else {
      if (va) { bs[ka] = d0; if (ns) ns[ka] = e0; if (os) *FUSED_GPTR(v2f, os + ka) = (v2f){x0.x, x0.y}; if (a.dec) a.dec[(long long)s * a.dec_stride + ka] = d0; }
      if (vc) { bs[kc] = d1; if (ns) ns[kc] = e1; if (os) *FUSED_GPTR(v2f, os + kc) = (v2f){x1.x, x1.y}; if (a.dec) a.dec[(long long)s * a.dec_stride + kc] = d1; }
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
  if constexpr (XE) {
    const v4f xi = reinterpret_cast<const v4f *>(xe)[lane];
    add(va, ka, d0, make_float2(xi.x, xi.y));
    add(vc, kc, d1, make_float2(xi.z, xi.w));
  } else {
  add(va, ka, d0, x0);
  add(vc, kc, d1, x1);
  }
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

template <int EPT, int LAG, int MT0, bool OFF, bool DBG, int ABL = 0>
__device__ __forceinline__ void fused_role_b(const FusedArgs &a, int s, int i0, int t3, int nt, int NE, const unsigned char *midr, float2 *stage, int lane, int wave) {
  FusedB16<MT0> b;
  b.load(a.afragB, lane);
  if (FUSED_B_PRIO) __builtin_amdgcn_s_setprio(FUSED_B_PRIO);
  float2 *os = a.out ? a.out + (long long)s * a.out_stride + a.out_off : nullptr;     // (null: MPX + |x|^2 only)
  fused_barrier();
  int p = 1000 * t3, q = 0;
  int kb = a.kb_ref + 384 * i0;                      // call-relative IF index of the first staged sample
  int tile_g = i0, blk = 0;
  float prev_tile = 0.f;                              // phase of the last sample of the previous tile (wave 1's first sample needs it)
  float save0 = 0.f;
  FusedBlkWin win{};
  if (a.base) { save0 = a.st[s].disc_save; blk = a.wg_blk0[(int)blockIdx.x]; win.load(a, blk, lane); }
  // (as in FusedB16::load: nothing loaded before the loop may still count as "in flight" inside it, or its first use in
  // every epoch waits for the epilogue's stores)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(save0), "+v"(win.end_l), "+v"(win.len_l));
  blk = __builtin_amdgcn_readfirstlane(blk);
  unsigned long long busy = 0, t_begin = FUSED_CLK();
  for (int e = 0; e < NE; e++) {
    const unsigned long long tb = FUSED_CLK();
    // the sample before this workgroup's first one: the previous call's (disc_save), or recomputed from the warm-up mid
    // samples -- by the wave that owns position 47, one epoch before the wave that needs its phase reads it
    if constexpr (MT0 == 2) { if (e == EPT + LAG && a.base && kb > 0) b.sample_before(p, midr, reinterpret_cast<float *>(stage + 384), lane, a); }
    if (e == EPT + 1 + LAG && a.base && MT0 == 0) {
      if (kb <= 0) prev_tile = save0;
      else { const float2 xb = stage[384]; prev_tile = fused_atan2(xb.y, xb.x) * (1.0f / a.nf); }
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
          // (a plain vector type: float2's assignment operator takes a generic `this` and would bring the flat store back)
          typedef float v2f_ __attribute__((ext_vector_type(2)));
          if (k >= 0 && k < a.n_if) *FUSED_GPTR(v2f_, os + k) = *reinterpret_cast<const v2f_ *>(stage + idx);
        }
      }
      kb += 384; tile_g++;
    }
    if (!OFF && e >= EPT + 1 + LAG && e <= EPT * nt + EPT + LAG) {
      b.template epoch<EPT>(q, p, midr, stage, lane, a);
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

// ---- the same epilogue behind the dense stage B of the R8B resampler class (k_ifr_poly5h, kernels.hpp; round 6) ----------
// A tile of that kernel is 3072 consecutive IF samples staged in LDS in sample order: wave w takes samples 384 w .. 384 w + 383
// -- a "macro tile" of the epilogue above, three thirds of 128 -- with the same partial-sum records (FusedPart, macro tile
// 8 tile + w), so that everything behind the front end (k_stats, the IF AGC's state solve on |x|^2, the PLL stage) is what it
// is behind the fused kernel.  A workgroup walks a contiguous run of tiles: the phase of the sample before a tile is the
// previous tile's last one (s_carry), and the first sample of a RUN is left to k_poly5h_heads -- the phases on either side
// of a run boundary are each computed by the tile that owns the sample, so the result does not depend on how the call is
// cut into runs.  Of FusedArgs this uses: n_if, nf, bound, base*, nrm*, out*, dec*, st, part, n_tiles (= 8 x tiles),
// part_from, if_off / if_len / nb, wg_blk0.
struct Poly5hDiscEpi {
  using Args = FusedArgs;
  static constexpr bool kOn = true;
  int blk, it;
  FusedBlkWin win;
  float save0;
  __device__ __forceinline__ void begin(const Args &a, int s, int lane) {
    save0 = a.st[s].disc_save;
    blk = a.wg_blk0[(int)blockIdx.x];
    win.load(a, blk, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(save0), "+v"(win.end_l), "+v"(win.len_l));
    blk = __builtin_amdgcn_readfirstlane(blk);
    it = 0;
  }
  // kb: call-relative IF index of the tile's first sample; run_ph = a.mid32 reused: [stream][workgroup][2] phases of the
  // run's first and last sample
  __device__ __forceinline__ void tile(const Args &a, int s, const float2 *stage, int kb, int tile, bool first, bool last, int lane,
                                       int wave, float *s_carry) {
    const float inv_nf = 1.0f / a.nf;
    const float2 *mst = stage + 384 * wave;
    const int kbm = kb + 384 * wave, tile_g = 8 * tile + wave;
    float2 *os = a.out ? a.out + (long long)s * a.out_stride + a.out_off : nullptr;
    float *run_ph = a.mid32 + ((size_t)s * gridDim.x + blockIdx.x) * 2;
    float prev0;
    if (wave == 0) {
      if (kb <= 0) prev0 = save0;                             // (the call's sample 0 takes save0 by its own test)
      else if (first) prev0 = __builtin_nanf("");             // unknown here: difference 0, k_poly5h_heads fills it in
      else prev0 = s_carry[it & 1];
    } else { const float2 xp = stage[384 * wave - 1]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
#ifndef FMR_P5H_EPI_ABL
#define FMR_P5H_EPI_ABL 0        // (diagnostic builds: fused_epilogue's ablation mask -- 128 no global stores, 256 no block sums)
#endif
    fused_epilogue<0, FMR_P5H_EPI_ABL>(a, s, mst, kbm, tile_g, blk, win, prev0, save0, os, lane);
    { const float2 xp = mst[127]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
    fused_epilogue<1, FMR_P5H_EPI_ABL>(a, s, mst, kbm, tile_g, blk, win, prev0, save0, os, lane);
    { const float2 xp = mst[255]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
    fused_epilogue<2, FMR_P5H_EPI_ABL>(a, s, mst, kbm, tile_g, blk, win, prev0, save0, os, lane);
    if (wave == 7 && lane == 0) {
      const float2 xl = stage[3071];
      const float ph = fused_atan2(xl.y, xl.x) * inv_nf;
      s_carry[(it + 1) & 1] = ph;
      if (last) run_ph[1] = ph;
    }
    if (wave == 0 && lane == 0 && first) { const float2 xf = stage[0]; run_ph[0] = fused_atan2(xf.y, xf.x) * inv_nf; }
    it++;
  }
};

// ---- ... and behind the IF filter of FM (main.cpp -f: LowPassFilterFirIQ, Filter.cpp:37-96) on the matrix cores: k_ifr_poly4 in
// its 1 : 1 shape (48 / 48, TB = the filter's taps, the lag-0 tap left out of the banded matrix) with this epilogue, round 6.
// A wave's pass ends with 384 consecutive filter outputs staged in LDS.  (1) The outputs behind a block's head get their lag-0
// term c[0] x[i] (the reference's head path sums the lags 1 .. order only), in place; (2) behind a barrier the discriminator
// epilogue above runs on the corrected samples -- the filtered IF samples go to `out` (the IF AGC reads them), the block sums'
// |x|^2 is the filter's INPUT (FmDecode.cpp:95).  k_fm_block3<.., true> + k_disc_heads, which this replaces for calls of whole
// tiles, took 0.145 ms per 5.2 M IF samples with the reference's add / multiply / add rounding on the vector ALUs; here the
// filter is one fmaf chain in lag order per output (1e-7 relative from that, inside the 2e-6 the front end is held to).
struct Poly4FirDiscEpi {
  using Args = FusedArgs;
  static constexpr bool kOn = true;
  int blk, blk_c, it;
  FusedBlkWin win, win_c;
  float save0;
  __device__ __forceinline__ void begin(const Args &a, int s, int lane) {
    save0 = a.st[s].disc_save;
    blk = a.wg_blk0[(int)blockIdx.x];
    win.load(a, blk, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(save0), "+v"(win.end_l), "+v"(win.len_l));
    blk = __builtin_amdgcn_readfirstlane(blk);
    blk_c = blk; win_c = win;
    it = 0;
  }
  // lag 0 for the outputs of one third that lie behind their block's head; stage: the macro tile's 384 samples, xw: their inputs
  template <int MT0>
  __device__ __forceinline__ void lag0(const Args &a, float2 *stage, const float2 *xw, int kb, int lane) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int k0 = kb + 128 * MT0;
    if (k0 >= a.n_if || k0 + 128 <= 0) return;
    const int kf = k0 < 0 ? 0 : k0;
    for (;;) {
      if (blk_c - win_c.base >= 64) win_c.load(a, blk_c, lane);
      if (blk_c >= a.nb || win_c.end(blk_c) > kf) break;
      blk_c++;
    }
    if (blk_c >= a.nb) return;
    const int cut = win_c.end(blk_c), st0 = cut - win_c.len(blk_c);
    const int ka = k0 + 2 * lane;
    v4f y = reinterpret_cast<v4f *>(stage + 128 * MT0)[lane];
    const v4f x = reinterpret_cast<const v4f *>(xw + 128 * MT0)[lane];
    // exact-support repair, as in the fused front end: a non-finite IF sample has made every output of the banded tile whose
    // k-range holds it NaN (zero taps times NaN).  Rare, wave-uniform: the lane's two outputs again as plain tap loops over the
    // window -- an output whose lags 1 .. order hold no such sample comes out finite, as in the reference
    bool fixed = false;
    if (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(y.x + y.y + y.z + y.w)) != 0) {
      const float2 *xp = xw + 128 * MT0 + 2 * lane;            // the input of this lane's first output
      float2 r0 = make_float2(0.f, 0.f), r1 = make_float2(0.f, 0.f);
#pragma unroll 1
      for (int j = a.fir_order; j >= 1; j--) {
        const float c = a.hA[j];
        const float2 u = xp[-j], w = xp[1 - j];
        r0.x = fmaf(c, u.x, r0.x); r0.y = fmaf(c, u.y, r0.y);
        r1.x = fmaf(c, w.x, r1.x); r1.y = fmaf(c, w.y, r1.y);
      }
      y = (v4f){r0.x, r0.y, r1.x, r1.y};
      fixed = true;
    }
    const bool body_a = ka - (ka < cut ? st0 : cut) >= a.fir_order, body_c = ka + 1 - (ka + 1 < cut ? st0 : cut) >= a.fir_order;
    if (body_a) { y.x = fmaf(x.x, a.fir_c0, y.x); y.y = fmaf(x.y, a.fir_c0, y.y); }
    if (body_c) { y.z = fmaf(x.z, a.fir_c0, y.z); y.w = fmaf(x.w, a.fir_c0, y.w); }
    if (body_a || body_c || fixed) reinterpret_cast<v4f *>(stage + 128 * MT0)[lane] = y;
  }
  // kb: call-relative index of the wave's first staged sample; stage_all: the four waves' staging areas, 384 samples each
  __device__ __forceinline__ void pass(const Args &a, int s, float2 *stage_all, int wave, int h, int kb, int tile_g, bool first, bool last,
                                       int lane, float *s_carry, const float2 *xw) {
    const float inv_nf = 1.0f / a.nf;
    float2 *mst = stage_all + 384 * wave;
    lag0<0>(a, mst, xw, kb, lane); lag0<1>(a, mst, xw, kb, lane); lag0<2>(a, mst, xw, kb, lane);
    __syncthreads();                                          // every wave's samples are final: the phase before a wave's first one is its neighbour's
    float2 *os = a.out ? a.out + (long long)s * a.out_stride + a.out_off : nullptr;
    float *run_ph = a.mid32 + ((size_t)s * gridDim.x + blockIdx.x) * 2;
    float prev0;
    if (wave == 0) {
      if (kb <= 0) prev0 = save0;
      else if (first) prev0 = __builtin_nanf("");             // k_poly5h_heads fills it in
      else prev0 = s_carry[it & 1];
    } else { const float2 xp = stage_all[384 * wave - 1]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
    fused_epilogue<0, 0, true>(a, s, mst, kb, tile_g, blk, win, prev0, save0, os, lane, xw);
    { const float2 xp = mst[127]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
    fused_epilogue<1, 0, true>(a, s, mst, kb, tile_g, blk, win, prev0, save0, os, lane, xw + 128);
    { const float2 xp = mst[255]; prev0 = fused_atan2(xp.y, xp.x) * inv_nf; }
    fused_epilogue<2, 0, true>(a, s, mst, kb, tile_g, blk, win, prev0, save0, os, lane, xw + 256);
    if (wave == 3 && lane == 0) {
      const float2 xl = mst[383];
      const float ph = fused_atan2(xl.y, xl.x) * inv_nf;
      s_carry[(it + 1) & 1] = ph;
      if (last) run_ph[1] = ph;
    }
    if (first && lane == 0) { const float2 xf = mst[0]; run_ph[0] = fused_atan2(xf.y, xf.x) * inv_nf; }
    it++;
  }
};

// the first sample of every run but the call's first: phase difference across the run boundary, and its share of the block sums
// (runs as the kernel in front cut them: the first run_rem of tiles_per_wg + 1 tiles, the others of tiles_per_wg)
__global__ void k_poly5h_heads(FusedArgs a, int grid, int tiles_per_wg, int run_rem = 0) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x + 1, s = blockIdx.y;
  if (w >= grid) return;
  const int t0 = w * tiles_per_wg + min(w, run_rem);
  const int kb = a.kb_ref + 3072 * t0;
  if (kb <= 0 || kb >= a.n_if) return;
  const float *run_ph = a.mid32 + (size_t)s * grid * 2;
  float d = run_ph[2 * w] - run_ph[2 * (w - 1) + 1];                                 // V5, as fused_epilogue
  if (d > a.bound) d -= 2 * a.bound;
  if (d < -a.bound) d += 2 * a.bound;
  if (isnan(d)) d = 0.f;
  a.base[(long long)s * a.base_stride + a.base_off + kb] = d;
  if (a.dec) a.dec[(long long)s * a.dec_stride + kb] = d;
  if (kb + 128 > a.part_from) {
    FusedPart *pt = a.part + ((long long)s * a.n_tiles + 8ll * t0) * 3;
    pt->sum[0][0] += d; pt->sum[0][1] += d * d;
  }
}

// ABL: ablation mask for tools/bench_fused.hip (0 = product; 1 no stage-A arithmetic, 2 no stage-B MFMAs, 4 no input DMA)
#define FUSED_THREADS 512     // eight waves, two per SIMD: loader, three stage-B waves, four stage-A waves
template <int D, int NA, int PAR, int ABL = 0>
__global__ __launch_bounds__(FUSED_THREADS) void k_ifr_fused(FusedArgs a) {
  using SH = FusedShape<D, NA>;
  constexpr bool DBG = (ABL & 32) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_f[];
  unsigned char *midr = lds_f + SH::NSLOT * SH::SLOT_BYTES;
  float2 *stage = reinterpret_cast<float2 *>(midr + 4 * SH::PLANE_BYTES);
  const int s = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i0 = a.wg_tile0 ? a.wg_tile0[(int)blockIdx.x] : (int)blockIdx.x * a.tiles_per_wg;
  const int i1 = a.wg_tile0 ? a.wg_tile0[(int)blockIdx.x + 1] : min(i0 + a.tiles_per_wg, a.n_tiles);
  if (i0 >= i1) return;
  // (the unused positions of the mid ring meet zero taps: they must hold finite numbers)
  for (int i = threadIdx.x; i < 4 * SH::PLANE_BYTES / 16; i += FUSED_THREADS) reinterpret_cast<uint4 *>(midr)[i] = make_uint4(0u, 0u, 0u, 0u);
  const int nt = i1 - i0, NE = SH::EPT * nt + SH::EPT + 2, EA = SH::EPT * nt;       // EA = last stage-A epoch
  const int jE0 = a.j_ref + 1000 * i0;                         // first mid sample of epoch 0 (a macro tile = 1000 mid samples)
  const int pos00 = (a.pos_ref + 1000 * (i0 % 3)) % SH::MIDR;
  const int t3 = (a.t3_ref + i0) % 3;
  const float2 *xs = a.iq + (long long)s * a.iq_stride;
  const float2 *hs = a.in_halo + (long long)s * a.H_in;

  if (a.stamps && threadIdx.x == 0) {
    const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.stamps[3 * w] = fused_realtime();
    a.stamps[3 * w + 2] = ((unsigned long long)xcc << 32) | hw;
  }
  if (wave == 0) {
    // ------------------------------------------------------------------ loader
    // cy[k]: DMA instructions of the batch issued k+1 epochs ago ... the batches younger than the one needed next
    int cy[SH::AHEAD];
#pragma unroll
    for (int k = 0; k < SH::AHEAD; k++) cy[k] = 0;
    if (!(ABL & 4)) {
#pragma unroll
      for (int k = 0; k < SH::AHEAD; k++)
        if (k <= EA) { const int c = fused_fill<D, NA>(a, xs, hs, jE0 + SH::ME * k, lds_f + (size_t)k * SH::SLOT_BYTES, lane, k == 0); if (k >= 1) cy[k - 1] = c; }
    }
    { int young = 0;
#pragma unroll
      for (int k = 0; k < SH::AHEAD - 1; k++) young += cy[k];
      fused_wait_upto(young); }               // the slot of epoch 0 has landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = SH::AHEAD;
    unsigned long long busy = 0, t_begin = FUSED_CLK();
    for (int e = 0; e < NE; e++) {
      const unsigned long long tb = FUSED_CLK();
      int cn = 0;
      if (!(ABL & 4) && e + SH::AHEAD <= EA)
        cn = fused_fill<D, NA>(a, xs, hs, jE0 + SH::ME * (e + SH::AHEAD), lds_f + (size_t)slot * SH::SLOT_BYTES, lane, false);
      slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
      // the slot of epoch e+1 must have landed: everything but the AHEAD-1 younger batches (cy[0] is epoch e+1 itself)
      int young = cn;
#pragma unroll
      for (int k = 1; k < SH::AHEAD - 1; k++) young += cy[k];
      fused_wait_upto(young);
#pragma unroll
      for (int k = 0; k + 1 < SH::AHEAD - 1; k++) cy[k] = cy[k + 1];
      if (SH::AHEAD >= 2) cy[SH::AHEAD - 2] = cn;
      busy += FUSED_CLK() - tb;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) { a.dbg[0] = busy; a.dbg[1] = FUSED_CLK() - t_begin; }
    if (a.stamps && lane == 0) a.stamps[3 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = fused_realtime();
  } else if (wave == 1) {
    // ------------------------------------------------------------------ stage B (one row tile per wave) + a third of the epilogue
    fused_role_b<SH::EPT, 0, 0, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else if (wave == 2) {
    fused_role_b<SH::EPT, 0, 1, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else if (wave == 3) {
    fused_role_b<SH::EPT, 0, 2, (ABL & 2) != 0, (ABL & 32) != 0, ABL>(a, s, i0, t3, nt, NE, midr, stage, lane, wave);
  } else {
    // ------------------------------------------------------------------ stage A on the fp16 matrix cores
    const int aw = wave - 4, unit = aw >> 1;
    FusedMfmaA<D, NA> qa;
    qa.load(a.afragA, PAR, lane);
    fused_barrier();
    int slot = 0, pos0 = pos00, jE = jE0;
    unsigned long long busy = 0, t_begin = FUSED_CLK();
    for (int e = 0; e < NE; e++) {
      const unsigned long long tb = FUSED_CLK();
      if (e <= EA) {
        const unsigned char *sl = lds_f + (size_t)slot * SH::SLOT_BYTES;
        if (aw & 1) qa.template run<1, PAR, (ABL & 25)>(a, s, jE, pos0, sl, midr, unit, lane);
        else qa.template run<0, PAR, (ABL & 25)>(a, s, jE, pos0, sl, midr, unit, lane);
        slot = (slot == SH::NSLOT - 1) ? 0 : slot + 1;
        if (aw == 3 && e < EA) fused_copy_preroll<D, NA>(sl, lds_f + (size_t)slot * SH::SLOT_BYTES, lane);
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
