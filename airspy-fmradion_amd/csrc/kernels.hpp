// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the FM/AM chain.
//
// Compiled with -ffp-contract=off: every place that wants a fused multiply-add
// says fmaf()/fma() explicitly; everything else keeps the reference's
// separate-rounding arithmetic (x86-64 baseline build, CMakeLists.txt:193-199),
// which is what makes most stages bit-comparable with the oracle.
//
// Buffer convention ("prefix halo"): a stage buffer is [halo H | data of this
// call]; reads at local index -1..-H reach the samples of the previous call.
// k_shift_halo re-seats the halos at the end of a call.
//
// Stream dimension: blockIdx.y (parallel kernels) or the thread index (serial
// recurrence kernels, one lane per stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmr {
// The MPX signal (discriminator output) in HBM for the FM decoder: the float the discriminator produces; FmDecode.cpp:143
// widens it to double, which every consumer does when it loads it (bit-identical, half the bytes).
typedef float fm_mpx_t;


// ---------------------------------------------------------------------------
// Per-stream state carried across launches ("per-stream PLL state carried
// across launches" of the north star); initial values: SURVEY.md 8a table.
// ---------------------------------------------------------------------------
struct PpsEventDev {
  unsigned long long pps_index, sample_index;
  double block_position;
  unsigned block, pad;
};

#define FMR_MAX_PPS 64

struct StreamState {
  // IfSimpleAgc (IfSimpleAgc.cpp:26-34)
  float agc_gain;
  // PhaseDiscriminator m_save_value (PhaseDiscriminator.cpp:30)
  float disc_save, disc_save_next;
  int disc_save_valid;
  // FmDecoder / AmDecoder statistics (FmDecode.cpp:95,149-150)
  float if_rms, baseband_mean, baseband_level;
  int stereo_detected;
  // PilotPhaseLock (PilotPhaseLock.cpp:37-51)
  double pll_phase, pll_freq, pll_freq_err, pll_level;
  double bq_i_x1, bq_i_x2, bq_q_x1, bq_q_x2, lf_x1;
  int lock_cnt, pilot_periods;
  unsigned long long pps_cnt, sample_cnt;
  int n_pps, pad0;
  PpsEventDev pps[FMR_MAX_PPS];
  // LowPassFilterRC x2, HighPassFilterIir x2 (Filter.cpp:169,240)
  double de_mono_x1, de_stereo_x1;
  double dc_mono_x1, dc_mono_x2, dc_st_x1, dc_st_x2;
  // MultipathFilter (MultipathFilter.cpp:59-75)
  double mpf_error;
  unsigned mpf_resets;
  unsigned agc_sync_timeouts;    // k_mpf4 gave up waiting for the AGC kernel beside it (protocol error: reported, fmr_status)
  // AmDecoder: AfSimpleAgc gain, dc block, de-emphasis
  double af_gain, am_dc_x1, am_dc_x2, am_de_x1;
  double am_de_x1_next, am_dc_x1_next, am_dc_x2_next;    // staged by the time-parallel AM tail, committed once it has converged
  // mean PLL phase increment over the previous call (= pilot frequency estimate);
  // seeds the initial trajectory guess of the time-parallel PLL (kernels_par.hpp)
  double pll_favg;
  int pll_favg_valid, pad2;
};

struct PllConst {
  double minfreq, maxfreq;
  double bq_b0, bq_a1, bq_a2;   // PilotPhaseLock.cpp:48-49
  double lf_b0, lf_b1;          // :51
  int lock_delay;               // :43
  int pilot_frequency;          // PilotPhaseLock.h:31
  double minsignal;             // PilotPhaseLock.h:37
};

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Serial per-lane loops over global memory: issue U independent loads first, then
// run the U dependent steps, so one memory latency is paid per U samples instead of
// per sample (the recurrences themselves cannot be reordered).
template <int U, class T, class F>
__device__ __forceinline__ void serial_prefetch(const T *__restrict__ p, int i0, int i1, F &&f) {
  int i = i0;
  for (; i + U <= i1; i += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = p[i + u];
#pragma unroll
    for (int u = 0; u < U; u++) f(i + u, v[u]);
  }
  for (; i < i1; i++) f(i, p[i]);
}
// The same with the loads of batch k+1 issued BEFORE the steps of batch k.  A body that stores to global memory makes
// the difference: gfx9 counts loads and stores in one in-order counter (vmcnt), so waiting for loads issued after a
// batch's stores waits for the stores' acknowledgements too -- a full memory round trip plus the load latency in
// front of every batch.  Issued ahead of the stores, the loads are waited for with the stores still in flight.
template <int U, class T, class F>
__device__ __forceinline__ void serial_prefetch_ahead(const T *__restrict__ p, int i0, int i1, F &&f) {
  int i = i0;
  T cur[U], nxt[U];
  if (i + U <= i1) {
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = p[i + u];
  }
  for (; i + U <= i1; i += U) {
#pragma unroll
    for (int u = 0; u < U; u++) nxt[u] = p[min(i + U + u, i1 - 1)];   // (clamped, not predicated: a branch around the
                                                                      // loads makes the compiler wait with vmcnt(0))
#pragma unroll
    for (int u = 0; u < U; u++) f(i + u, cur[u]);
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
  }
  for (; i < i1; i++) f(i, p[i]);
}
template <int U, class T, class T2, class F>
__device__ __forceinline__ void serial_prefetch2(const T *__restrict__ p, const T2 *__restrict__ q, int i0, int i1, F &&f) {
  int i = i0;
  for (; i + U <= i1; i += U) {
    T v[U];
    T2 w[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = p[i + u]; w[u] = q[i + u]; }
#pragma unroll
    for (int u = 0; u < U; u++) f(i + u, v[u], w[u]);
  }
  for (; i < i1; i++) f(i, p[i], q[i]);
}

template <int BLOCK>
__device__ __forceinline__ float block_sum(float v, float *scratch /* BLOCK/64 */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  float r = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / 64; i++) r += scratch[i];
  __syncthreads();
  return r;
}

// FourthConverterIQ (include/FourthConverterIQ.h:45-79), down-conversion:
// table index cycles 0,1,2,3 with the absolute sample index.
__device__ __forceinline__ float2 fourth_rot(float2 v, unsigned idx) {
  switch (idx & 3u) {
  case 0: return v;
  case 1: return make_float2(v.y, -v.x);
  case 2: return make_float2(-v.x, -v.y);
  default: return make_float2(-v.y, v.x);
  }
}

// FourthConverterIQ::process as a stand-alone kernel (FourthConverterIQ.h:45-79): the table index advances by one
// per sample (down: 0,1,2,3 -> x1, x(-j)..., up: 0,3,2,1); idx0 = the object's m_index at the call.
__global__ void k_fourth_shift(const float2 *__restrict__ in, float2 *__restrict__ out, long long n, unsigned idx0, int up) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned step = (unsigned)(i & 3);
    const unsigned idx = up ? (idx0 + 4u - step) & 3u : (idx0 + step) & 3u;
    out[i] = fourth_rot(in[i], idx);
  }
}

// ---------------------------------------------------------------------------
// K_A  ifr_decim : front-end stage A, integer decimation by D with an NA-tap
// linear-phase FIR.  The only kernel that touches every input IQ sample:
// HBM-read bound (8 B per input sample), see DESIGN.md.
//   y[m] = sum_k hA[k] * x[D*m + ca - k]
// v1 layout: one output per lane, input span staged in LDS with coalesced
// float2 loads, taps through the scalar cache (wave-uniform index).
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ifr_decim(
    const float2 *__restrict__ iq, long long iq_stride, long long n_valid,
    const float2 *__restrict__ halo, int H, const float *__restrict__ hA, int NA, int D,
    long long top0, int count, float2 *__restrict__ mid, long long mid_stride, int mid_off,
    unsigned rot_base, int fourth) {
  extern __shared__ float2 lds_a[];
  const int s = blockIdx.y;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * BLOCK;
  const int span = BLOCK * D + NA - 1;
  const long long lo = top0 + (long long)m0 * D - (NA - 1);
  const float2 *xs = iq + (long long)s * iq_stride;
  const float2 *hs = halo + (long long)s * H;
  for (int i = tid; i < span; i += BLOCK) {
    const long long n = lo + i;
    float2 v = make_float2(0.f, 0.f);
    if (n < 0) {
      if (n >= -(long long)H) v = hs[H + n];
    } else if (n < n_valid) {
      v = xs[n];
    }
    if (fourth) v = fourth_rot(v, (unsigned)((long long)rot_base + n));
    lds_a[i] = v;
  }
  __syncthreads();
  const int m = m0 + tid;
  if (m < count) {
    const float2 *xp = lds_a + tid * D + (NA - 1);
    float ax = 0.f, ay = 0.f;
#pragma unroll 8
    for (int k = 0; k < NA; k++) {
      const float h = hA[k];
      const float2 x = xp[-k];
      ax = fmaf(h, x.x, ax);
      ay = fmaf(h, x.y, ay);
    }
    mid[(long long)s * mid_stride + mid_off + m] = make_float2(ax, ay);
  }
}

// ---------------------------------------------------------------------------
// Source sample formats the front end reads directly (fused ingest: the conversion the reference does on the
// host -- RtlSdrSource.cpp:359-365 for offset-binary u8, libsndfile's sf_read_float for the FileSource formats
// of FileSource.cpp:120-128 -- happens while the tile is staged, so HBM carries 4 or 2 bytes per IQ sample
// instead of 8).  All scalings are powers of two: exact in float.
//   0 cf32 (I, Q float)   1 s16 LE / 32768   2 u8 (b - 128) / 128   3 s8 / 128
// ---------------------------------------------------------------------------
template <int FMT> struct IqFmt;
template <> struct IqFmt<0> { static constexpr int BPS = 8, G = 2; };
template <> struct IqFmt<1> { static constexpr int BPS = 4, G = 4; };
template <> struct IqFmt<2> { static constexpr int BPS = 2, G = 8; };
template <> struct IqFmt<3> { static constexpr int BPS = 2, G = 8; };
template <int FMT>
__device__ __forceinline__ float2 iq_load1(const void *base, long long n) {      // sample n of a raw stream
  if (FMT == 0) return reinterpret_cast<const float2 *>(base)[n];
  if (FMT == 1) { const short2 v = reinterpret_cast<const short2 *>(base)[n]; return make_float2(v.x * (1.0f / 32768.0f), v.y * (1.0f / 32768.0f)); }
  if (FMT == 2) { const uchar2 v = reinterpret_cast<const uchar2 *>(base)[n]; return make_float2(((int)v.x - 128) * (1.0f / 128.0f), ((int)v.y - 128) * (1.0f / 128.0f)); }
  const char2 v = reinterpret_cast<const char2 *>(base)[n];
  return make_float2(v.x * (1.0f / 128.0f), v.y * (1.0f / 128.0f));
}
// sample j (0 .. G-1) of one 16-byte group
template <int FMT>
__device__ __forceinline__ float2 iq_unpack(const uint4 &w, int j) {
  const unsigned wd[4] = {w.x, w.y, w.z, w.w};
  if (FMT == 0) return make_float2(__uint_as_float(wd[2 * j]), __uint_as_float(wd[2 * j + 1]));
  if (FMT == 1) {
    const unsigned u = wd[j];
    return make_float2((float)(short)(u & 0xffffu) * (1.0f / 32768.0f), (float)(short)(u >> 16) * (1.0f / 32768.0f));
  }
  const unsigned u = wd[j >> 1] >> ((j & 1) * 16);
  if (FMT == 2) return make_float2(((int)(u & 0xffu) - 128) * (1.0f / 128.0f), ((int)((u >> 8) & 0xffu) - 128) * (1.0f / 128.0f));
  return make_float2((float)(signed char)(u & 0xffu) * (1.0f / 128.0f), (float)(signed char)((u >> 8) & 0xffu) * (1.0f / 128.0f));
}

// ---------------------------------------------------------------------------
// K_A v2  ifr_decim2 : the HBM-bound front-end kernel, LDS traffic cut 4x.
//   y[m] = sum_p sum_q hp[p][q] * X_p[m - q],   X_p[i] = x[D i + ca - p]   (polyphase form)
// * the input tile is staged with coalesced 16-byte loads (all issued before the first
//   LDS write) and written to LDS DE-INTERLEAVED by phase (row p holds X_p), so that
//   lanes that own adjacent outputs read adjacent LDS words: conflict-free ds_read_b128;
// * every lane produces TWO adjacent outputs: one 16-byte LDS read (two samples)
//   feeds four complex MACs;
// * taps are wave-uniform: Q per phase through the scalar cache into SGPRs.
// Q = taps per phase (even, zero padded), hp = [D][Q] on the host side.
// ABL: ablation switch for tools/bench_decim.hip (0 = product, 1 = no compute, 2 = no loads).
// ---------------------------------------------------------------------------
template <int BLOCK, int Q, int ABL = 0, bool FOURTH = false, int CV = 1, int FMT = 0>
__global__ __launch_bounds__(BLOCK) void k_ifr_decim2(
    const float2 *__restrict__ iq, long long iq_stride, long long n_valid,
    const float2 *__restrict__ halo, int H, const float *__restrict__ hp, int D, int ca,
    long long n0 /* D*mA_prev - call_start: local input index of x[D*m] for output m = 0 */,
    int count, float2 *__restrict__ mid, long long mid_stride, int mid_off, unsigned rot_base, int fourth,
    int S_pad, unsigned div_magic) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_a2[];
  constexpr int T = 2 * BLOCK;            // outputs per workgroup
  const int s = blockIdx.y;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * T;
  // X_p[i], i = i0 .. i0 + T + Q - 1, i0 = m0 - Q;  input index n = n0 + D*i + ca - p
  const long long n_base = n0 + (long long)D * (m0 - Q) + ca - (D - 1);   // smallest input index of the tile
  const int span = D * (T + Q);
  constexpr int G = IqFmt<FMT>::G;                   // samples per 16-byte group
  const char *xs = reinterpret_cast<const char *>(iq) + (long long)s * iq_stride * IqFmt<FMT>::BPS;
  const float2 *hs = halo + (long long)s * H;
  const int par = (int)(((n_base % G) + G) % G);     // make the group address 16-byte aligned
  const long long n_al = n_base - par;
  const int npairs = (span + par + G - 1) / G;       // 16-byte groups of the tile
  const unsigned dummy = (unsigned)(D * S_pad);      // one spare LDS slot swallows out-of-tile elements
  // branch-free scatter of one staged sample: element index ee -> (phase row p, column u)
  auto put = [&](int ee, float2 v, long long n) {
    if (FOURTH) v = fourth_rot(v, (unsigned)((long long)rot_base + n));
    const unsigned u = (unsigned)(((unsigned long long)(unsigned)ee * div_magic) >> 24);   // ee / D
    const unsigned p = (unsigned)(D - 1) - ((unsigned)ee - u * (unsigned)D);
    const unsigned addr = ((unsigned)ee < (unsigned)span) ? p * (unsigned)S_pad + u : dummy;
    lds_a2[addr] = v;
  };
  // ---- stage.  Interior tiles issue ALL their 16-byte loads before the first LDS write
  // (one HBM latency per tile); edge tiles (halo / end of the input) go element-wise.
  if (n_al >= 0 && n_al + (long long)G * npairs <= n_valid) {
    const uint4 *src = reinterpret_cast<const uint4 *>(xs + n_al * IqFmt<FMT>::BPS);
    const int full = npairs / BLOCK;                 // trips in which every lane has a group
    constexpr int MAXP = 32 / G;                     // 16 groups of cf32 pairs, 8 of s16, 4 of 8-bit samples
    uint4 w[MAXP];
#pragma unroll
    for (int t = 0; t < MAXP; t++) {
      const int g = tid + t * BLOCK;
      if (ABL == 2) w[t] = make_uint4(0x3f800000u, 0x40000000u, 0x40400000u, 0x40800000u);
      else if (t < full || g < npairs) w[t] = src[g];
    }
#pragma unroll
    for (int t = 0; t < MAXP; t++) {
      const int g = tid + t * BLOCK;
      if (t < full || g < npairs) {
        const int e0 = G * g - par;
#pragma unroll
        for (int j = 0; j < G; j++) put(e0 + j, iq_unpack<FMT>(w[t], j), n_al + (long long)G * g + j);
      }
    }
  } else {
    for (int g = tid; g < npairs; g += BLOCK) {
#pragma unroll
      for (int j = 0; j < G; j++) {
        const long long n = n_al + (long long)G * g + j;
        float2 v = make_float2(0.f, 0.f);
        if (n < 0) { if (n >= -(long long)H) v = hs[H + n]; } else if (n < n_valid) v = iq_load1<FMT>(xs, n);
        put(G * g - par + j, v, n);
      }
    }
  }
  __syncthreads();
  // ---- compute: outputs m0 + 2*tid, m0 + 2*tid + 1
  float y0r = 0.f, y0i = 0.f, y1r = 0.f, y1i = 0.f;
  if (ABL == 1) {
    const float2 t0 = lds_a2[Q + 2 * tid];
    y0r = t0.x; y0i = t0.y;
  } else {
    if (CV == 2) {
    // ablation (tools/bench_acc64.hip): the same sums carried in fp64, rounded to fp32 once at the end
    double d0r = 0., d0i = 0., d1r = 0., d1i = 0.;
    for (int p = 0; p < D; p++) {
      const float *h = hp + p * Q;
      const float4 *row = reinterpret_cast<const float4 *>(lds_a2 + p * S_pad + Q + 2 * tid);
#pragma unroll
      for (int j = 0; j <= Q / 2; j++) {
        const float4 ab = row[-j];
        if (j < Q / 2) {
          const double hb = h[2 * j], ha = h[2 * j + 1];
          d1r = fma(hb, (double)ab.z, d1r); d1i = fma(hb, (double)ab.w, d1i);
          d1r = fma(ha, (double)ab.x, d1r); d1i = fma(ha, (double)ab.y, d1i);
          d0r = fma(hb, (double)ab.x, d0r); d0i = fma(hb, (double)ab.y, d0i);
        }
        if (j >= 1) {
          const double h1 = h[2 * j - 1];
          d0r = fma(h1, (double)ab.z, d0r); d0i = fma(h1, (double)ab.w, d0i);
        }
      }
    }
    y0r = (float)d0r; y0i = (float)d0i; y1r = (float)d1r; y1i = (float)d1i;
    } else if (CV == 0) {
    for (int p = 0; p < D; p++) {
      const float *h = hp + p * Q;
      const float4 *row = reinterpret_cast<const float4 *>(lds_a2 + p * S_pad + Q + 2 * tid);
#pragma unroll
      for (int j = 0; j <= Q / 2; j++) {
        const float4 ab = row[-j];
        if (j < Q / 2) {
          const float hb = h[2 * j], ha = h[2 * j + 1];
          y1r = fmaf(hb, ab.z, y1r); y1i = fmaf(hb, ab.w, y1i);
          y1r = fmaf(ha, ab.x, y1r); y1i = fmaf(ha, ab.y, y1i);
          y0r = fmaf(hb, ab.x, y0r); y0i = fmaf(hb, ab.y, y0i);
        }
        if (j >= 1) {
          const float h1 = h[2 * j - 1];
          y0r = fmaf(h1, ab.z, y0r); y0i = fmaf(h1, ab.w, y0i);
        }
      }
    }
    } else {
    // Explicit 2-wide vectors: (re,im) x broadcast tap -> v_pk_fma_f32; four independent
    // accumulators per lane so that the packed FMAs do not queue behind each other.
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    v2f a0 = {0.f, 0.f}, b0 = a0, a1 = a0, b1 = a0;
    for (int p = 0; p < D; p++) {
      const float *h = hp + p * Q;                    // wave-uniform: scalar loads
      const v4f *row = reinterpret_cast<const v4f *>(
          __builtin_assume_aligned(lds_a2 + p * S_pad + Q + 2 * tid, 16));
      v4f ab[Q / 2 + 1];
#pragma unroll
      for (int j = 0; j <= Q / 2; j++) ab[j] = row[-j];   // a = X_p[2l-2j] (xy), b = X_p[2l-2j+1] (zw)
#pragma unroll
      for (int j = 0; j < Q / 2; j++) {
        const v2f hb = {h[2 * j], h[2 * j]}, ha = {h[2 * j + 1], h[2 * j + 1]};
        const v2f aj = {ab[j].x, ab[j].y}, bj = {ab[j].z, ab[j].w}, bn = {ab[j + 1].z, ab[j + 1].w};
        b1 = __builtin_elementwise_fma(hb, bj, b1);   // output 2l+1: q=2j   -> b_j
        a1 = __builtin_elementwise_fma(ha, aj, a1);   //               q=2j+1 -> a_j
        a0 = __builtin_elementwise_fma(hb, aj, a0);   // output 2l:   q=2j   -> a_j
        b0 = __builtin_elementwise_fma(ha, bn, b0);   //               q=2j+1 -> b_{j+1}
      }
    }
    y0r = a0.x + b0.x; y0i = a0.y + b0.y;
    y1r = a1.x + b1.x; y1i = a1.y + b1.y;
    }
  }
  const int m = m0 + 2 * tid;
  float2 *o = mid + (long long)s * mid_stride + mid_off + m;
  if (m + 1 < count) {
    o[0] = make_float2(y0r, y0i);
    o[1] = make_float2(y1r, y1i);
  } else if (m < count) {
    o[0] = make_float2(y0r, y0i);
  }
}

// ---------------------------------------------------------------------------
// K_B  ifr_poly : front-end stage B, rational LB/MB polyphase resampler.
//   y[k] = sum_j hB[p_k][j] * mid[n_k - W + 1 + j],  t = k*MB, n_k = t / LB, p_k = t % LB
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ifr_poly(
    const float2 *__restrict__ mid, long long mid_stride, long long mid_abs0, int mid_valid,
    const float *__restrict__ hB, int TB, unsigned LB, unsigned MB, unsigned long long t0,
    int count, float2 *__restrict__ out, long long out_stride, int out_off) {
  extern __shared__ float2 lds_b[];
  const int s = blockIdx.y;
  const int tid = threadIdx.x;
  const int W = TB >> 1;
  const unsigned long long tk0 = t0 + (unsigned long long)(blockIdx.x * BLOCK) * MB;
  const long long x_lo = (long long)(tk0 / LB) - W + 1 - mid_abs0;
  const int span = (int)(((unsigned long long)(BLOCK - 1) * MB) / LB) + TB + 2;
  const float2 *ms = mid + (long long)s * mid_stride;
  for (int i = tid; i < span; i += BLOCK) {
    const long long idx = x_lo + i;
    float2 v = make_float2(0.f, 0.f);
    if (idx >= 0 && idx < mid_valid) v = ms[idx];
    lds_b[i] = v;
  }
  __syncthreads();
  const int k = blockIdx.x * BLOCK + tid;
  if (k < count) {
    const unsigned long long t = t0 + (unsigned long long)k * MB;
    const long long nk = (long long)(t / LB);
    const unsigned p = (unsigned)(t % LB);
    const int xi = (int)(nk - W + 1 - mid_abs0 - x_lo);
    const float *h = hB + (size_t)p * TB;
    const float2 *xp = lds_b + xi;
    float ax = 0.f, ay = 0.f;
#pragma unroll 4
    for (int j = 0; j < TB; j++) {
      const float c = h[j];
      const float2 x = xp[j];
      ax = fmaf(c, x.x, ax);
      ay = fmaf(c, x.y, ay);
    }
    out[(long long)s * out_stride + out_off + k] = make_float2(ax, ay);
  }
}

// ---------------------------------------------------------------------------
// K_B frac  ifr_poly_frac : stage B for ratios whose phase table would not fit -- ppm-corrected source rates
// (main.cpp:708-711: LB of the order 1e5 .. 1e8).  The table holds the prototype at mu = p / LT, p = 0 .. LT; the taps of
// an output are interpolated linearly between the two rows around its exact phase (design.hpp).  Positions are exact
// integers, call-relative: t = rem0 + k MB in units of 1 / LB mid sample, n_k = nk0 + t / LB, mu = (t mod LB) / LB.
// x0: index in the stream's mid buffer of the first sample output 0 reads (nk0 - W + 1); TBP: row pitch (TB rounded up
// to a multiple of 4, zero padded).
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ifr_poly_frac(
    const float2 *__restrict__ mid, long long mid_stride, long long x0, int mid_valid,
    const float *__restrict__ tab, int TB, int TBP, int LT, unsigned long long LB, unsigned long long MB,
    unsigned long long rem0, int count, float2 *__restrict__ out, long long out_stride, int out_off) {
  extern __shared__ float2 lds_b[];
  const int s = blockIdx.y;
  const int tid = threadIdx.x;
  const unsigned long long tk0 = rem0 + (unsigned long long)(blockIdx.x * BLOCK) * MB;
  const long long n_lo = (long long)(tk0 / LB);
  const long long x_lo = x0 + n_lo;
  const int span = (int)(((unsigned long long)(BLOCK - 1) * MB) / LB) + TB + 2 + 4;
  const float2 *ms = mid + (long long)s * mid_stride;
  for (int i = tid; i < span; i += BLOCK) {
    const long long idx = x_lo + i;
    float2 v = make_float2(0.f, 0.f);
    if (idx >= 0 && idx < mid_valid) v = ms[idx];
    lds_b[i] = v;
  }
  __syncthreads();
  const int k = blockIdx.x * BLOCK + tid;
  if (k < count) {
    const unsigned long long t = rem0 + (unsigned long long)k * MB;
    const long long nk = (long long)(t / LB);
    const unsigned long long xx = (t - (unsigned long long)nk * LB) * (unsigned long long)LT;   // (t mod LB) LT < 2^46
    const unsigned long long row = xx / LB;
    const float a = (float)(xx - row * LB) / (float)LB;
    const float4 *h0 = reinterpret_cast<const float4 *>(tab + (size_t)row * TBP);
    const float4 *h1 = reinterpret_cast<const float4 *>(tab + (size_t)(row + 1) * TBP);
    const float2 *xp = lds_b + (int)(nk - n_lo);
    float ax = 0.f, ay = 0.f;
    for (int j4 = 0; j4 < TBP / 4; j4++) {
      const float4 c0 = h0[j4], c1 = h1[j4];
      const float c[4] = {fmaf(a, c1.x - c0.x, c0.x), fmaf(a, c1.y - c0.y, c0.y), fmaf(a, c1.z - c0.z, c0.z), fmaf(a, c1.w - c0.w, c0.w)};
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float2 x = xp[4 * j4 + u];
        ax = fmaf(c[u], x.x, ax);
        ay = fmaf(c[u], x.y, ay);
      }
    }
    out[(long long)s * out_stride + out_off + k] = make_float2(ax, ay);
  }
}

// ---------------------------------------------------------------------------
// K_B v2  ifr_poly2 : rational polyphase stage with WAVE-UNIFORM taps.
// Output k = P*LB + p (P = period, p = position in the period) uses tap phase
// phi[p] = (p*MB) % LB at mid sample P*MB + off[p], off[p] = (p*MB) / LB.  Lanes own
// 64 consecutive PERIODS and walk the positions p together, so every lane of a wave
// needs the same tap row: taps come through the scalar cache (SGPR operands of packed
// FMAs), and lane l reads LDS at l*MB + off[p] + j -- stride MB samples, conflict-free
// for odd MB.  One workgroup = 64 periods x LB outputs; its 4 waves share the positions.
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ifr_poly2(
    const float2 *__restrict__ mid, long long mid_stride, long long mid_abs0, int mid_valid,
    const float *__restrict__ hB, int TB, int LB, int MB, const int *__restrict__ phi, const int *__restrict__ off,
    long long k0 /* absolute index of the first output of this call */, int count,
    float2 *__restrict__ out, long long out_stride, int out_off, int tile_len) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_b2[];
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tap rows go through SGPRs
  constexpr int NW = BLOCK / 64;
  const int W = TB >> 1;
  const long long P0 = k0 / LB + (long long)blockIdx.x * 64;       // first period of this tile
  const long long a0 = P0 * MB - W + 1;                            // absolute mid index of lds[0]
  const float2 *ms = mid + (long long)s * mid_stride;
  // ---- stage the mid-rate tile (coalesced, 8 loads in flight per lane)
  for (int i0 = 0; i0 < tile_len; i0 += 8 * BLOCK) {
    float2 v[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int i = i0 + t * BLOCK + tid;
      const long long idx = a0 + i - mid_abs0;
      v[t] = (i < tile_len && idx >= 0 && idx < mid_valid) ? ms[idx] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int i = i0 + t * BLOCK + tid;
      if (i < tile_len) lds_b2[i] = v[t];
    }
  }
  __syncthreads();
  float2 *os = out + (long long)s * out_stride + out_off;
  const long long kbase = (P0 + lane) * LB - k0;                   // local output index of position 0 of my period
  const float2 *xl = lds_b2 + lane * MB;
  // two positions per trip (independent accumulators) and 16 taps per inner step keep
  // enough loads in flight to cover the LDS / scalar-cache latency
  for (int p = wave; p < LB; p += 2 * NW) {
    const int p2 = (p + NW < LB) ? p + NW : p;
    const float *h0 = hB + (size_t)phi[p] * TB, *h1 = hB + (size_t)phi[p2] * TB;   // wave-uniform tap rows
    const float2 *x0 = xl + off[p], *x1 = xl + off[p2];
    v2f a0 = {0.f, 0.f}, a1 = a0, c0 = a0, c1 = a0;
    int j = 0;
    for (; j + 16 <= TB; j += 16) {
      float2 xa[16], xb[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { xa[u] = x0[j + u]; xb[u] = x1[j + u]; }
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        a0 = __builtin_elementwise_fma((v2f){h0[j + u], h0[j + u]}, (v2f){xa[u].x, xa[u].y}, a0);
        a1 = __builtin_elementwise_fma((v2f){h0[j + u + 1], h0[j + u + 1]}, (v2f){xa[u + 1].x, xa[u + 1].y}, a1);
        c0 = __builtin_elementwise_fma((v2f){h1[j + u], h1[j + u]}, (v2f){xb[u].x, xb[u].y}, c0);
        c1 = __builtin_elementwise_fma((v2f){h1[j + u + 1], h1[j + u + 1]}, (v2f){xb[u + 1].x, xb[u + 1].y}, c1);
      }
    }
    for (; j < TB; j++) {
      const float2 xa = x0[j], xb = x1[j];
      a0 = __builtin_elementwise_fma((v2f){h0[j], h0[j]}, (v2f){xa.x, xa.y}, a0);
      c0 = __builtin_elementwise_fma((v2f){h1[j], h1[j]}, (v2f){xb.x, xb.y}, c0);
    }
    const long long k = kbase + p;
    if (k >= 0 && k < count) os[k] = make_float2(a0.x + a1.x, a0.y + a1.y);
    const long long k2 = kbase + p2;
    if (p2 != p && k2 >= 0 && k2 < count) os[k2] = make_float2(c0.x + c1.x, c0.y + c1.y);
  }
}

// ---------------------------------------------------------------------------
// K_B v3  ifr_poly3 : as v2 (lanes own periods, wave-uniform taps), but a wave takes Q CONSECUTIVE
// positions p0..p0+Q-1 at once and walks the union of their windows (TB + off[p0+Q-1] - off[p0]
// samples) a single time: every mid sample read from LDS feeds Q packed FMAs, one per position,
// with tap hq[i] = h[phi[p0+q]][i - (off[p0+q] - off[p0])] taken from a ZERO-PADDED tap row (PADZ
// zeros either side) -- so the shift costs a pointer offset, not a branch.  v2 issues one
// ds_read_b64 per packed FMA and is LDS-bandwidth bound at a quarter of the FMA rate; v3 reads Q
// times less.  Per-position accumulation is sequential in tap order.
// ---------------------------------------------------------------------------
#define FMR_POLY_PADZ 32
template <int BLOCK, int Q>
__global__ __launch_bounds__(BLOCK) void k_ifr_poly3(
    const float2 *__restrict__ mid, long long mid_stride, long long mid_abs0, int mid_valid,
    const float *__restrict__ hBp, int TB, int LB, int MB, const int *__restrict__ phi, const int *__restrict__ off,
    long long k0, int count, float2 *__restrict__ out, long long out_stride, int out_off, int tile_len) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_b3[];
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = BLOCK / 64;
  const int W = TB >> 1;
  const int TBP = TB + 2 * FMR_POLY_PADZ;
  const long long P0 = k0 / LB + (long long)blockIdx.x * 64;
  const long long a0 = P0 * MB - W + 1;
  const float2 *ms = mid + (long long)s * mid_stride;
  for (int i0 = 0; i0 < tile_len; i0 += 8 * BLOCK) {
    float2 v[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int i = i0 + t * BLOCK + tid;
      const long long idx = a0 + i - mid_abs0;
      v[t] = (i < tile_len && idx >= 0 && idx < mid_valid) ? ms[idx] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int i = i0 + t * BLOCK + tid;
      if (i < tile_len) lds_b3[i] = v[t];
    }
  }
  __syncthreads();
  float2 *os = out + (long long)s * out_stride + out_off;
  const long long kbase = (P0 + lane) * LB - k0;
  const float2 *xl = lds_b3 + lane * MB;
  const int ngroups = (LB + Q - 1) / Q;
  for (int g = wave; g < ngroups; g += NW) {
    const int p0 = g * Q;
    const int o0 = __builtin_amdgcn_readfirstlane(off[p0]);
    const float *hq[Q];
    int dmax = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int pq = min(p0 + q, LB - 1);
      const int dq = __builtin_amdgcn_readfirstlane(off[pq]) - o0;
      const int ph = __builtin_amdgcn_readfirstlane(phi[pq]);
      hq[q] = hBp + (size_t)ph * TBP + FMR_POLY_PADZ - dq;
      dmax = dq;
    }
    const int span = TB + dmax;
    const float2 *x0 = xl + o0;
    v2f acc[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) acc[q] = (v2f){0.f, 0.f};
    for (int i = 0; i < span; i += 8) {
      float2 xa[8];
#pragma unroll
      for (int u = 0; u < 8; u++) xa[u] = x0[i + u];
#pragma unroll
      for (int u = 0; u < 8; u++) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
          const float h = hq[q][i + u];
          acc[q] = __builtin_elementwise_fma((v2f){h, h}, (v2f){xa[u].x, xa[u].y}, acc[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const long long k = kbase + p0 + q;
      if (p0 + q < LB && k >= 0 && k < count) os[k] = make_float2(acc[q].x, acc[q].y);
    }
  }
}

// ---------------------------------------------------------------------------
// K_B v4  ifr_poly4 : the rational polyphase stage as an f32 MFMA product, for the LB/MB = 48/125,
// TB = 210 shape (every source rate whose stage A lands on 1 MHz: 10, 6, 3 MS/s ...).
// For one period P the LB outputs are  Y[p] = sum_m A[p][m] X[m],  X[m] = mid[P*MB - W + 1 + m],
// A[p][m] = h[phi[p]][m - off[p]] (0 outside the TB taps): a constant (48 x 332) banded matrix.
// v_mfma_f32_16x16x4_f32 tiles: rows = 16 positions (3 row tiles), columns = 8 periods x (re, im),
// k = 4 consecutive m.  The A fragments of all 3 x 63 live (non-zero) k-steps stay in registers for
// the life of the (persistent) workgroup, so the taps are never re-read; every B fragment (one
// ds_read_b32 per lane) feeds up to three MFMAs.  v2/v3 stream the taps through the scalar cache
// (52 KB table, 16 KB cache) and stall on its misses at ~25 % VALU utilisation.
// An MFMA is bit-for-bit a k-ordered fmaf chain, i.e. the same sequential tap-order accumulation
// as v2/v3 (plus exact zero terms).
// ---------------------------------------------------------------------------
// What happens to a finished tile (64 periods = 3072 IF samples, staged in LDS): Poly5hStoreIf writes the IF samples out (the
// tiles of a call interleaved over the workgroups); Poly5hDiscEpi (kernels_fused.hpp, round 6) runs the phase discriminator
// and the block statistics there -- the fused front end's epilogue, a wave per 384 samples -- over CONTIGUOUS runs of
// tiles_per_wg tiles per workgroup, and the IF samples never go to HBM (k_disc was a separate 52-57 us pass over them).
struct Poly5hStoreIf {
  struct Args { int unused; };
  static constexpr bool kOn = false;
};

template <int LB, int MB, int TB>
struct Poly4Shape {
  static constexpr int off(int p) { return (p * MB) / LB; }
  static constexpr int ks_lo(int mt) { return off(16 * mt) / 4; }
  static constexpr int ks_hi(int mt) { return (off(16 * mt + 15) + TB + 3) / 4 - 1; }
  static constexpr int MT = LB / 16;
  static constexpr int nks(int mt) { return ks_hi(mt) - ks_lo(mt) + 1; }
  static constexpr int NK = nks(0) > nks(MT - 1) ? (nks(0) > nks(1) ? nks(0) : nks(1)) : (nks(MT - 1) > nks(1) ? nks(MT - 1) : nks(1));
  static constexpr int KS_ALL = ks_hi(MT - 1) + 1;          // k-steps of the union window
  static constexpr int XLEN = 4 * KS_ALL;                   // mid samples one period touches
};

// (EPI: what happens to a wave's 384 staged outputs -- Poly5hStoreIf: stored as they are, tiles interleaved over the
// workgroups; Poly4FirDiscEpi of kernels_fused.hpp, round 6: the IF filter's discriminator epilogue, contiguous runs of tiles)
template <int LB, int MB, int TB, int MINB = 2, class EPI = Poly5hStoreIf>
__global__ __launch_bounds__(256, MINB) void k_ifr_poly4(
    const float2 *__restrict__ mid, long long mid_stride, long long mid_abs0, int mid_valid,
    const float *__restrict__ afrag, long long k0, int count, float2 *__restrict__ out, long long out_stride,
    int out_off, int tile_len, int n_tiles, typename EPI::Args ea = typename EPI::Args{}, int tiles_per_wg = 0, int run_rem = 0) {
  using SH = Poly4Shape<LB, MB, TB>;
  static_assert(LB == 48, "three 16-row tiles");
  typedef float v4f __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float2 lds_b4[];
  __shared__ float s_carry4[2];
  const int x_len = ((tile_len + 127) / 128) * 128;          // the x region holds whole 1 KB wave chunks
  float2 *stage = lds_b4 + x_len;                            // 4 waves x (8 periods x LB) float2
  const int s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  constexpr int W = TB >> 1;
  // (EPI) runs: the first run_rem workgroups take tiles_per_wg + 1 tiles, the others tiles_per_wg
  const int tile_lo = EPI::kOn ? (int)blockIdx.x * tiles_per_wg + min((int)blockIdx.x, run_rem) : (int)blockIdx.x;
  const int tile_hi = EPI::kOn ? min(tile_lo + tiles_per_wg + ((int)blockIdx.x < run_rem ? 1 : 0), n_tiles) : n_tiles;
  const int tile_step = EPI::kOn ? 1 : (int)gridDim.x;
  EPI epi;
  if constexpr (EPI::kOn) {
    static_assert(MB == 48, "the 1 : 1 shape: an FIR");
    // (on the decoder stream's critical path, beside the audio tail of the call before: issue priority, as the PLL's passes)
    __builtin_amdgcn_s_setprio(3);
    if (tile_lo < tile_hi) epi.begin(ea, s, lane);
  }
  // ---- the constant A fragments: a[mt][i] = A[16 mt + n][4 (ks_lo(mt) + i) + kq]
  float a[SH::MT][SH::NK];
#pragma unroll
  for (int mt = 0; mt < SH::MT; mt++)
#pragma unroll
    for (int i = 0; i < SH::NK; i++) a[mt][i] = afrag[(mt * SH::NK + i) * 64 + lane];
  const float2 *ms = mid + (long long)s * mid_stride;
  float2 *os = out + (long long)s * out_stride + out_off;
  float2 *mystage = stage + wave * (8 * LB);
  for (int tile = tile_lo; tile < tile_hi; tile += tile_step) {
    const long long P0 = k0 / LB + (long long)tile * 64;
    const long long a0 = P0 * MB - W + 1;
    __syncthreads();                                          // previous tile fully consumed
    const long long src0 = a0 - mid_abs0;
    if (src0 >= 0 && src0 + x_len <= mid_valid) {
      // interior tile: direct global -> LDS copies (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16 B), all
      // of a wave's ~17 chunks in flight at once and no staging registers; edge tiles need the zero fill below
      const float2 *src = ms + src0;
      for (int c = wave; c < x_len / 128; c += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 128 + lane * 2),
                                         (__attribute__((address_space(3))) void *)(lds_b4 + c * 128), 16, 0, 0);
    } else
    for (int i0 = 0; i0 < tile_len; i0 += 8 * 256) {
      float2 v[8];
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int i = i0 + t * 256 + tid;
        const long long idx = a0 + i - mid_abs0;
        v[t] = (i < tile_len && idx >= 0 && idx < mid_valid) ? ms[idx] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int i = i0 + t * 256 + tid;
        if (i < tile_len) lds_b4[i] = v[t];
      }
    }
    __syncthreads();
    const float *xf = reinterpret_cast<const float *>(lds_b4);
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
      const int q0 = (wave + 4 * h) * 8;                      // first period of this column tile
      const float *xb = xf + 2 * ((q0 + (n >> 1)) * MB + kq) + (n & 1);
      v4f acc[SH::MT];
#pragma unroll
      for (int mt = 0; mt < SH::MT; mt++) acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < SH::KS_ALL; ks++) {
        const float b = xb[8 * ks];
#pragma unroll
        for (int mt = 0; mt < SH::MT; mt++)
          if (ks >= SH::ks_lo(mt) && ks <= SH::ks_hi(mt))
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][ks - SH::ks_lo(mt)], b, acc[mt], 0, 0, 0);
      }
      // D[row = 4 kq + v][col = n] -> position p = 16 mt + 4 kq + v of period q0 + n/2, component n & 1
      float *sf = reinterpret_cast<float *>(mystage);
#pragma unroll
      for (int mt = 0; mt < SH::MT; mt++)
#pragma unroll
        for (int v = 0; v < 4; v++) sf[2 * ((n >> 1) * LB + 16 * mt + 4 * kq + v) + (n & 1)] = acc[mt][v];
      __syncthreads();
      const long long kb = (P0 + q0) * LB - k0;               // local output index of the staged run
      if constexpr (EPI::kOn) {
        // the wave's 384 outputs are consecutive samples (eight periods of 48); the input sample of output t of the tile is
        // window sample t + TB - 1 (MB == LB: the window starts W - 1 samples before the period, the buffer's index 0 is
        // TB - 1 - (W - 1) samples before sample 0: fmradion_amd.hip)
        epi.pass(ea, s, stage, wave, h, (int)kb, 8 * tile + wave + 4 * h, tile == tile_lo && h == 0 && wave == 0, tile + 1 == tile_hi && h == 1,
                 lane, s_carry4, lds_b4 + LB * q0 + (TB - 1));
      } else {
#pragma unroll
      for (int t = 0; t < (8 * LB) / 64; t++) {
        const int idx = t * 64 + lane;
        const long long k = kb + idx;
        if (k >= 0 && k < count) os[k] = mystage[idx];
      }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// k_ifr_poly5h : the same dense product on the fp16 matrix cores -- v_mfma_f32_16x16x32_f16, 16 x the f32 MFMA rate --
// with BOTH operands split in two fp16 terms, x = h + l, and three products per tile (hh + hl + lh; ll is 2^-22 of the
// result): 16 / 3 of the f32 rate with fp32-class accuracy.  What makes the split safe for any signal level:
//   * taps: scaled at design time by a power of two so that the largest is in [512, 1024) -- the smallest taps of a
//     180 dB design then sit in fp16's subnormal range, whose absolute step (6e-8) is 2^-34 of the largest tap;
//   * mid samples: every tile finds the maximum of its own window and scales by the power of two that brings it into
//     [512, 1024) before the split (h keeps 11 bits, l the next 11: 2^-22 of the window's maximum);
//   * the MFMA accumulates in fp32; as in the f32 kernel the accumulators are flushed into a second set every 128 taps;
//   * the result is multiplied by the inverse of the two scales (powers of two: exact).
// (SQ counters of round 5's form, one wave per SIMD: matrix pipe busy 28 %, VALU 29 %, parked at s_waitcnt or the barrier
// 31 % of the wave cycles, 36 % of the LDS cycles bank conflicts: ds_read_b128 is served in groups of sixteen NON-adjacent
// lanes -- {0-3, 12-15, 20-27} ... -- not the half-waves the layout below was made for.)
// Layouts: the tile's window as four fp16 planes in LDS (re_h, re_l, im_h, im_l); a B fragment is eight consecutive
// elements of a plane from (period n/2) 125 + 32 kb + 8 kq -- a 16-byte read on a 2-byte boundary (LDS takes it,
// tools/test_mfma_f16.hip).  A fragments [k-block][row tile][h | l][lane][8], streamed through LDS in chunks of four
// k-blocks (128 taps) by LDS-DMA, double buffered, fetched once per workgroup.  Wave w: column tiles w and w + 4 x three
// row tiles = six accumulators, 18 MFMAs per k-block on ten 16-byte reads.
// ---------------------------------------------------------------------------
#define FMR_POLY5H_KCH 4                       // k-blocks (of 32 taps) per A chunk
// Waves per workgroup: 8 = two per SIMD with one column tile each (round 6: the shifts and LDS reads of one wave run under the
// MFMAs of the other; 0.34 -> 0.30 ms), 4 = one per SIMD with two tiles each (rounds 4-5).  Measured and dropped in round 6
// (NOTEBOOK.md): a deeper ring of A chunks, a third register set with the shifts between the MFMAs, reads in flight across
// the chunk barrier -- every one of them slower (more barriers, or spills at the 256 registers two waves per SIMD leave).
#ifndef FMR_POLY5H_WAVES
#define FMR_POLY5H_WAVES 8
#endif
typedef _Float16 fmr_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ fmr_h8 lds_read_h8(const void *p, int byte_off) {     // 16 bytes from any 2-byte boundary
  fmr_h8 v;
  // ("memory": the read must stay behind the barrier that publishes the planes / the chunk, which the compiler only
  // orders against memory operations it knows about)
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p + (unsigned)byte_off) : "memory");
  return v;
}
// Eight consecutive fp16 elements that start A elements (2 A bytes) past a 16-byte boundary: two aligned reads and a
// funnel shift by a compile-time amount (a read on a 2-byte boundary works but costs ~20 aligned ones, measured).  The
// reads are issued by issue(), the fragment is formed by get() once they have landed.
template <int A>
struct FmrH8Shifted {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u lo, hi;
  __device__ __forceinline__ void issue(const void *p, int byte_off) {       // byte_off: where the fragment starts (= 2 A mod 16)
    const unsigned addr = (unsigned)(size_t)p + (unsigned)(byte_off - 2 * A);
    asm volatile("ds_read_b128 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    if (A != 0) asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(hi) : "v"(addr) : "memory");
  }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(lo)); if (A != 0) asm volatile("" : "+v"(hi)); }
  __device__ __forceinline__ fmr_h8 get() const {
    constexpr int Q = (2 * A) / 4, R = (2 * A) % 4;
    const unsigned d[8] = {lo.x, lo.y, lo.z, lo.w, A ? hi.x : 0u, A ? hi.y : 0u, A ? hi.z : 0u, A ? hi.w : 0u};
    v4u o;
    if (R == 0) o = (v4u){d[Q], d[Q + 1], d[Q + 2], d[Q + 3]};
    else o = (v4u){__builtin_amdgcn_alignbyte(d[Q + 1], d[Q], R), __builtin_amdgcn_alignbyte(d[Q + 2], d[Q + 1], R),
                   __builtin_amdgcn_alignbyte(d[Q + 3], d[Q + 2], R), __builtin_amdgcn_alignbyte(d[(Q + 4) & 7], d[Q + 3], R)};
    fmr_h8 r;
    __builtin_memcpy(&r, &o, 16);
    return r;
  }
  static constexpr int kReads = A ? 2 : 1;
};

template <int LB, int MB, class EPI = Poly5hStoreIf>
__global__ __launch_bounds__(64 * FMR_POLY5H_WAVES) void k_ifr_poly5h(
    const float2 *__restrict__ mid, long long mid_stride, long long mid_abs0, int mid_valid,
    const _Float16 *__restrict__ afrag, int n_kb, float inv_tap_scale, int TB, long long k0, int count,
    float2 *__restrict__ out, long long out_stride, int out_off, int tile_len, int n_tiles,
    typename EPI::Args ea = typename EPI::Args{}, int tiles_per_wg = 0, int run_rem = 0) {
  static_assert(LB == 48, "three 16-row tiles");
  constexpr int KCH = FMR_POLY5H_KCH, MT = LB / 16, NWV = FMR_POLY5H_WAVES, NT = 64 * NWV, NH = 8 / NWV;
  static_assert(NWV == 4 || NWV == 8, "eight column tiles per tile: two per wave or one");
  constexpr int CHB = KCH * MT * 2 * 64 * 16;                  // bytes per A chunk
  typedef float v4f __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_5h[];
  // elements per plane (slack: the last k-block of the last period); = 32 mod 64, so that the im planes start 32 banks
  // away from the re planes
  const int x_len = ((tile_len + 127) / 128) * 128 + 96;
  _Float16 *pl = reinterpret_cast<_Float16 *>(lds_5h);        // [4][x_len]: re_h, re_l, im_h, im_l
  unsigned char *abuf = lds_5h + (size_t)4 * x_len * 2;        // [2][CHB]
  // NWV x (8 periods x LB): the results' staging area lies over the A buffers, which are idle by then (a barrier in between)
  float2 *stage = reinterpret_cast<float2 *>(abuf);
  static_assert(2 * CHB >= NWV * 8 * LB * (int)sizeof(float2), "the staging area fits over the A buffers");
  __shared__ float s_max[NWV];
  __shared__ float s_carry[2];                                 // (EPI: phase of the previous tile's last sample, double buffered)
  const int s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int W = TB >> 1;
  const float2 *ms = mid + (long long)s * mid_stride;
  float2 *os = out + (long long)s * out_stride + out_off;
  float2 *mystage = stage + wave * (8 * LB);
  const int n_chunks = n_kb / KCH;
  // the workgroup's tiles: blockIdx.x, + gridDim.x, ... or (EPI) the contiguous run [tile_lo, tile_hi): the first run_rem
  // workgroups take tiles_per_wg + 1 tiles, the others tiles_per_wg -- the longer runs lie at the head of the call, the runs
  // whose tiles write the per-block partial sums (the last ~400 blocks: 2 us more per tile) at its end
  const int tile_lo = EPI::kOn ? (int)blockIdx.x * tiles_per_wg + min((int)blockIdx.x, run_rem) : (int)blockIdx.x;
  const int tile_hi = EPI::kOn ? min(tile_lo + tiles_per_wg + ((int)blockIdx.x < run_rem ? 1 : 0), n_tiles) : n_tiles;
  const int tile_step = EPI::kOn ? 1 : (int)gridDim.x;
  EPI epi;
  if constexpr (EPI::kOn) { static_assert(NWV == 8 && LB == 48, "a wave per 384 staged samples"); if (tile_lo < tile_hi) epi.begin(ea, s, lane); }
  auto fetch_a = [&](int c) {       // chunk c -> buffer c & 1: CHB / 1024 one-KB pieces, wave w issues pieces w, w + 4, ...
    const unsigned char *src = reinterpret_cast<const unsigned char *>(afrag) + (size_t)c * CHB;
    unsigned char *dst = abuf + (c & 1) * CHB;
    for (int p = wave; p < CHB / 1024; p += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
  };
  for (int tile = tile_lo; tile < tile_hi; tile += tile_step) {
    const long long P0 = k0 / LB + (long long)tile * 64;
    const long long a0 = P0 * MB - W + 1;
    __syncthreads();                                          // previous tile fully consumed
    fetch_a(0);
    // ---- the window: maximum, scale, split into the four planes
    auto ld = [&](int i) -> float2 {
      const long long idx = a0 + i - mid_abs0;
      return (i < tile_len && idx >= 0 && idx < mid_valid) ? ms[idx] : make_float2(0.f, 0.f);
    };
    // one pass over the window: every lane keeps its (up to 48) samples in registers across the maximum -- all its loads
    // in flight at once, nothing read twice
    constexpr int NPL = 48 * 256 / NT;                         // x_len <= NPL * NT (checked by the host)
    float2 wv[NPL];
    float mx = 0.f;
#pragma unroll
    for (int u = 0; u < NPL; u++) wv[u] = (tid + u * NT < x_len) ? ld(tid + u * NT) : make_float2(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NPL; u++) mx = fmaxf(mx, fmaxf(fabsf(wv[u].x), fabsf(wv[u].y)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = s_max[0];
#pragma unroll
    for (int w = 1; w < NWV; w++) mx = fmaxf(mx, s_max[w]);
    // 2^e with mx 2^e in [512, 1024); a window of zeros (or one that holds no finite maximum) is not scaled
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) ex = 9 - (int)((__float_as_uint(mx) >> 23) & 0xff) + 127;
    ex = max(-100, min(100, ex));
    const float sc = __uint_as_float((unsigned)(127 + ex) << 23);
    const float inv = __uint_as_float((unsigned)(127 - ex) << 23) * inv_tap_scale;
#pragma unroll
    for (int u = 0; u < NPL; u++) {
      const int i = tid + u * NT;
      if (i < x_len) {
        const float xr = wv[u].x * sc, xi = wv[u].y * sc;
        const _Float16 rh = (_Float16)xr, ih = (_Float16)xi;
        pl[i] = rh; pl[x_len + i] = (_Float16)(xr - (float)rh);
        pl[2 * x_len + i] = ih; pl[3 * x_len + i] = (_Float16)(xi - (float)ih);
      }
    }
    // this lane's B columns: column tile t = wave + 4 h holds the periods t, t + 8, ... t + 56 (column n: period
    // 8 (n / 2) + t, plane pair re / im by n & 1).  Eight periods apart the windows start 1000 elements = 500 dwords apart,
    // 52 banks: the sixteen 16-byte reads of a half-wave fall on sixteen different bank quads (8 consecutive periods,
    // 62.5 dwords apart, landed on top of each other: ~6 cycles per read instead of 1).
    int bofs[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) bofs[h] = 2 * ((n & 1) * 2 * x_len + (8 * (n >> 1) + (wave + NWV * h)) * MB + 8 * kq);
    v4f acc[NH][MT], tot[NH][MT];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) tot[h][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
    // Fragments of k-block k live in register set k & 1: the reads of the next k-block are issued BEFORE the 18 MFMAs of
    // the current one (one wave per SIMD: nothing else hides the LDS latency).  Column tile t starts 125 t elements into
    // the planes: 5 t mod 8 elements past a 16-byte boundary, the same for every lane and k-block -- a compile-time
    // constant per wave (FmrH8Shifted).
    auto run_chunks = [&](auto a0_tag, auto a1_tag) {
      constexpr int A0 = decltype(a0_tag)::value, A1 = decltype(a1_tag)::value;        // (one column tile per wave: A1 unused)
      constexpr int NRD = 2 * FmrH8Shifted<A0>::kReads + (NH == 2 ? 2 * FmrH8Shifted<A1>::kReads : 0) + 2 * MT;    // LDS reads per k-block
      FmrH8Shifted<A0> b0h[2], b0l[2];
      FmrH8Shifted<A1> b1h[2], b1l[2];
      fmr_h8 ah[2][MT], al[2][MT];
      auto read_b = [&](int set, int kb) {
        b0h[set].issue(pl, bofs[0] + 64 * kb); b0l[set].issue(pl, bofs[0] + 64 * kb + 2 * x_len);
        if constexpr (NH == 2) { b1h[set].issue(pl, bofs[NH - 1] + 64 * kb); b1l[set].issue(pl, bofs[NH - 1] + 64 * kb + 2 * x_len); }
      };
      auto read_a = [&](int set, const unsigned char *ab, int k) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          ah[set][mt] = lds_read_h8(ab, ((k * MT + mt) * 2 + 0) * 1024);
          al[set][mt] = lds_read_h8(ab, ((k * MT + mt) * 2 + 1) * 1024);
        }
      };
      for (int c = 0; c < n_chunks; c++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of chunk c have landed
        __syncthreads();                                        // ... and everybody's (and the planes, c = 0); chunk c - 1's buffer is free
        if (c + 1 < n_chunks) fetch_a(c + 1);
        const unsigned char *ab = abuf + (c & 1) * CHB + lane * 16;
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) acc[h][mt] = (v4f){0.f, 0.f, 0.f, 0.f};
        read_b(0, c * KCH);
        read_a(0, ab, 0);
#pragma unroll
        for (int k = 0; k < KCH; k++) {
          const int cur = k & 1, nxt = cur ^ 1;
          if (k + 1 < KCH) {
            // the current set must have landed, the next one may stay in flight: LDS returns in order
            read_b(nxt, c * KCH + k + 1);
            read_a(nxt, ab, k + 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          // (the MFMAs must depend on something behind the wait: every fragment through an empty asm)
          b0h[cur].pin(); b0l[cur].pin();
          if constexpr (NH == 2) { b1h[cur].pin(); b1l[cur].pin(); }
#pragma unroll
          for (int mt = 0; mt < MT; mt++) { asm volatile("" : "+v"(ah[cur][mt])); asm volatile("" : "+v"(al[cur][mt])); }
          fmr_h8 bh[NH], bl[NH];
          bh[0] = b0h[cur].get(); bl[0] = b0l[cur].get();
          if constexpr (NH == 2) { bh[NH - 1] = b1h[cur].get(); bl[NH - 1] = b1l[cur].get(); }
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int h = 0; h < NH; h++) acc[h][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][mt], bh[h], acc[h][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int h = 0; h < NH; h++) acc[h][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][mt], bl[h], acc[h][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int h = 0; h < NH; h++) acc[h][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][mt], bh[h], acc[h][mt], 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) tot[h][mt] += acc[h][mt];
      }
    };
    // column tile t: 125 t elements in, i.e. (5 t) mod 8 past a 16-byte boundary (four waves: tiles w and w + 4; eight: tile w)
    static_assert(MB == 125, "the misalignment table below is 125 t mod 8");
    using I0 = std::integral_constant<int, 0>;
    if constexpr (NWV == 4) {
      switch (wave) {
      case 0: run_chunks(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}); break;
      case 1: run_chunks(std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{}); break;
      case 2: run_chunks(std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{}); break;
      default: run_chunks(std::integral_constant<int, 7>{}, std::integral_constant<int, 3>{}); break;
      }
    } else {
      switch (wave) {
      case 0: run_chunks(std::integral_constant<int, 0>{}, I0{}); break;
      case 1: run_chunks(std::integral_constant<int, 5>{}, I0{}); break;
      case 2: run_chunks(std::integral_constant<int, 2>{}, I0{}); break;
      case 3: run_chunks(std::integral_constant<int, 7>{}, I0{}); break;
      case 4: run_chunks(std::integral_constant<int, 4>{}, I0{}); break;
      case 5: run_chunks(std::integral_constant<int, 1>{}, I0{}); break;
      case 6: run_chunks(std::integral_constant<int, 6>{}, I0{}); break;
      default: run_chunks(std::integral_constant<int, 3>{}, I0{}); break;
      }
    }
    __syncthreads();                                          // everybody has read its last A fragments: the buffers become the staging area
    if constexpr (EPI::kOn) {
      // the tile in sample order: period 8 (n / 2) + wave of the tile, position 16 mt + 4 kq + v, component n & 1
      float *sf = reinterpret_cast<float *>(stage);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int v = 0; v < 4; v++) sf[2 * ((8 * (n >> 1) + wave) * LB + 16 * mt + 4 * kq + v) + (n & 1)] = tot[0][mt][v] * inv;
      __syncthreads();
#ifndef FMR_P5H_NO_EPI_CALL
      epi.tile(ea, s, stage, (int)(P0 * LB - k0), tile, tile == tile_lo, tile + 1 == tile_hi, lane, wave, s_carry);
#endif
      continue;
    }
    // D[row = 4 kq + v][col = n] -> position p = 16 mt + 4 kq + v of period 8 (n / 2) + t, component n & 1
#pragma unroll
    for (int h = 0; h < NH; h++) {
      const int t = wave + NWV * h;
      float *sf = reinterpret_cast<float *>(mystage);
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int v = 0; v < 4; v++) sf[2 * ((n >> 1) * LB + 16 * mt + 4 * kq + v) + (n & 1)] = tot[h][mt][v] * inv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave writes and reads its own staging area
      const long long kb = P0 * LB - k0;                      // local output index of the tile's first sample
#pragma unroll
      for (int u = 0; u < (8 * LB) / 64; u++) {
        const int idx = u * 64 + lane, j = idx / LB, pos = idx - j * LB;      // staged period j = period 8 j + t of the tile
        const long long k = kb + (long long)(8 * j + t) * LB + pos;
        if (k >= 0 && k < count) os[k] = mystage[idx];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------
// k_shift_halo : re-seat prefix halos at the end of a call.  All elements are
// 8 bytes (float2 or double).  newhalo[i] = concat(halo,data)[i + N].
// ---------------------------------------------------------------------------
struct HaloDesc {           // (in 32-bit words: the host doubles stride, H and N for 8-byte elements)
  unsigned *buf;             // start of [halo | data] of stream 0
  long long stride;          // words between streams
  int H;                     // halo length
  int N;                     // data words appended in this call
};
#define FMR_MAX_HALO 12
struct HaloTable { HaloDesc d[FMR_MAX_HALO]; int n; };

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_shift_halo(HaloTable tab) {
  const HaloDesc d = tab.d[blockIdx.x];
  unsigned *b = d.buf + (long long)blockIdx.y * d.stride;
  if (d.N <= 0) return;
  for (int c = 0; c < d.H; c += BLOCK) {
    const int i = c + threadIdx.x;
    unsigned v = 0;
    if (i < d.H) v = b[i + d.N];
    __syncthreads();
    if (i < d.H) b[i] = v;
    __syncthreads();
  }
}

// Pipelined chain: the halo of a ring slot comes from the slot of the call before it -- dst[i] = concat(halo, data)_src[i + N],
// i < H, src != dst (N = samples the previous call appended to src).
struct CarryDesc {          // (in 32-bit words, as HaloDesc)
  const unsigned *src;
  unsigned *dst;
  long long stride;
  int H, N;
};
struct CarryTable { CarryDesc d[4]; int n; };
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_carry_halo(CarryTable tab) {
  const CarryDesc d = tab.d[blockIdx.x];
  const unsigned *a = d.src + (long long)blockIdx.y * d.stride;
  unsigned *b = d.dst + (long long)blockIdx.y * d.stride;
  for (int i = threadIdx.x; i < d.H; i += BLOCK) b[i] = a[i + d.N];
}

// K_A's input halo lives in its own buffer because the caller owns the input:
// newhalo[i] = concat(halo, iq[0..N))[i + N].
template <int BLOCK, int FMT = 0>
__global__ __launch_bounds__(BLOCK) void k_update_in_halo(
    float2 *__restrict__ halo, int H, const float2 *__restrict__ iq, long long iq_stride, long long N) {
  float2 *hs = halo + (long long)blockIdx.y * H;
  const char *xs = reinterpret_cast<const char *>(iq) + (long long)blockIdx.y * iq_stride * IqFmt<FMT>::BPS;
  for (int c = 0; c < H; c += BLOCK) {
    const int i = c + threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (i < H) {
      const long long j = (long long)i + N;  // index into concat(halo, iq)
      v = (j < H) ? hs[j] : iq_load1<FMT>(xs, j - H);
    }
    __syncthreads();
    if (i < H) hs[i] = v;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Per-block tables (host-computed, identical for all streams)
// ---------------------------------------------------------------------------
struct BlockTab {
  const int *if_off;     // [nb] offset of the block inside this call's IF samples
  const int *if_len;     // [nb]
  const int *au_off;     // [nb] offset at the audio rate (FM: 48 kHz; AM: = if_off)
  const int *au_len;     // [nb]
  const int *mpf_active; // [nb] 1 when the equaliser runs on this block
  int nb;
};

// ---------------------------------------------------------------------------
// K_blk  fm_block : per decoder block -- IF RMS (Utility.h:118-132) and the
// optional LowPassFilterFirIQ (Filter.cpp:37-96) incl. the block-head path
// (hazard H1).  FM: RMS of the block entering the decoder (FmDecode.cpp:95);
// AM: RMS after the FIR (AmDecode.cpp:101,154).
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fm_block(
    const float2 *__restrict__ ifb, long long if_stride, int if_halo, BlockTab bt,
    const float *__restrict__ coeff, int ntaps, int fir_enable, int rms_after_fir,
    float2 *__restrict__ firb, long long fir_stride, float *__restrict__ if_rms_blk) {
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const float2 *x = ifb + (long long)s * if_stride + if_halo + bt.if_off[b];
  float2 *y = firb + (long long)s * fir_stride + bt.if_off[b];
  const int order = ntaps - 1;
  const int half_order = (order - 1) / 2;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    float2 v = x[i];
    if (fir_enable) {
      float yr = 0.f, yi = 0.f;
      if (i < order) {
        // head: lags 1..order, state part first then in-block part (Filter.cpp:59-68)
        for (int j = i + 1; j <= order; j++) {
          const float2 t = x[i - j];
          const float c = coeff[j];
          yr += t.x * c; yi += t.y * c;
        }
        for (int j = 1; j <= i; j++) {
          const float2 t = x[i - j];
          const float c = coeff[j];
          yr += t.x * c; yi += t.y * c;
        }
      } else {
        // body: folded symmetric form incl. lag 0 (Filter.cpp:73-82)
        for (int k = 0; k <= half_order; k++) {
          const float2 a = x[i - k], bb = x[i - (order - k)];
          const float c = coeff[k];
          yr += (a.x + bb.x) * c; yi += (a.y + bb.y) * c;
        }
        if ((order % 2) == 0) {
          const float2 t = x[i - order / 2];
          const float c = coeff[order / 2];
          yr += t.x * c; yi += t.y * c;
        }
      }
      const float2 o = make_float2(yr, yi);
      y[i] = o;
      if (rms_after_fir) v = o;
    }
    acc += v.x * v.x + v.y * v.y;
  }
  const float tot = block_sum<BLOCK>(acc, scratch);
  if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tot / (float)(unsigned)n);
}

// K_blk v2 (round 2): the same arithmetic (identical operation order, bit-identical results) out of LDS, one output per
// lane -- the kernel of the 48 kHz modes (AM, SSB, CW: 256-sample blocks, 255 / 2049 taps), where almost every output is a
// head output and k_fm_block3 below measured 0.168 against this kernel's 0.145 ms per 2 M IF samples.  The block is walked in
// tiles of TL outputs; a tile's window (order history samples + TL) and the coefficients are staged once, so a tap costs
// two LDS reads instead of two cached global loads -- the AM / SSB filters have 255 / 2049 taps, and with the 48 kHz
// modes' short blocks almost every output takes the sequential block-head path (hazard H1).
template <int BLOCK, int TL>
__global__ __launch_bounds__(BLOCK) void k_fm_block2(
    const float2 *__restrict__ ifb, long long if_stride, int if_halo, BlockTab bt,
    const float *__restrict__ coeff, int ntaps, int rms_after_fir,
    float2 *__restrict__ firb, long long fir_stride, float *__restrict__ if_rms_blk) {
  extern __shared__ float2 lds_fb[];
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const int order = ntaps - 1;
  const int half_order = (order - 1) / 2;
  float2 *xs = lds_fb;                                              // [order + TL]: xs[order + t] = x[i0 + t]
  float *cs = reinterpret_cast<float *>(lds_fb + order + TL);       // [ntaps]
  const float2 *x = ifb + (long long)s * if_stride + if_halo + bt.if_off[b];
  float2 *y = firb + (long long)s * fir_stride + bt.if_off[b];
  for (int k = threadIdx.x; k < ntaps; k += BLOCK) cs[k] = coeff[k];
  float acc = 0.f;
  for (int i0 = 0; i0 < n; i0 += TL) {
    const int tn = min(TL, n - i0);
    __syncthreads();
    for (int k = threadIdx.x; k < order + tn; k += BLOCK) xs[k] = x[i0 - order + k];    // (reaches into the prefix halo)
    __syncthreads();
    for (int t = threadIdx.x; t < tn; t += BLOCK) {
      const int i = i0 + t;
      const float2 *xl = xs + order + t;                            // xl[-j] = x[i - j]
      float yr = 0.f, yi = 0.f;
      if (i < order) {
        // head: lags 1..order, state part first then in-block part (Filter.cpp:59-68)
        for (int j = i + 1; j <= order; j++) {
          const float2 tt = xl[-j];
          const float c = cs[j];
          yr += tt.x * c; yi += tt.y * c;
        }
        for (int j = 1; j <= i; j++) {
          const float2 tt = xl[-j];
          const float c = cs[j];
          yr += tt.x * c; yi += tt.y * c;
        }
      } else {
        // body: folded symmetric form incl. lag 0 (Filter.cpp:73-82)
        for (int k = 0; k <= half_order; k++) {
          const float2 a = xl[-k], bb = xl[-(order - k)];
          const float c = cs[k];
          yr += (a.x + bb.x) * c; yi += (a.y + bb.y) * c;
        }
        if ((order % 2) == 0) {
          const float2 tt = xl[-(order / 2)];
          const float c = cs[order / 2];
          yr += tt.x * c; yi += tt.y * c;
        }
      }
      const float2 o = make_float2(yr, yi);
      y[i] = o;
      const float2 v = rms_after_fir ? o : xl[0];
      acc += v.x * v.x + v.y * v.y;
    }
  }
  const float tot = block_sum<BLOCK>(acc, scratch);
  if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tot / (float)(unsigned)n);
}

// ---------------------------------------------------------------------------
// k_fir_finish (round 6): behind the IF FIR of the 48 kHz modes on the matrix cores.  LowPassFilterFirIQ::process
// (Filter.cpp:37-96) sums the lags 1 .. order for the first `order` outputs of a block (the head path has no lag 0) and the
// lags 0 .. order for the others: k_ifr_poly4<48, 48, 255> runs the lags 1 .. order for every output, one fmaf chain in lag
// order over the stream (no block structure), this kernel adds c[0] x[i] to the outputs behind a block's head and takes the
// block's RMS (Utility.h:118-132, AmDecode.cpp:107 / NbfmDecode.cpp) in k_fm_block2's summation order.
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fir_finish(
    const float2 *__restrict__ ifb, long long if_stride, int if_halo, BlockTab bt, const float *__restrict__ coeff, int ntaps,
    float2 *__restrict__ firb, long long fir_stride, float *__restrict__ if_rms_blk) {
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const int order = ntaps - 1;
  const float2 *x = ifb + (long long)s * if_stride + if_halo + bt.if_off[b];
  float2 *y = firb + (long long)s * fir_stride + bt.if_off[b];
  const float c0 = coeff[0];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    float2 o = y[i];
    bool store = false;
    // exact-support repair: a non-finite input sample has made every output of its banded tile NaN (zero taps times NaN); an
    // output whose lags 1 .. order hold no such sample is computed again with the plain tap loop (x reaches into the halo)
    if (!__builtin_isfinite(o.x + o.y)) {
      float2 r = make_float2(0.f, 0.f);
      for (int j = order; j >= 1; j--) { const float c = coeff[j]; const float2 u = x[i - j]; r.x = fmaf(c, u.x, r.x); r.y = fmaf(c, u.y, r.y); }
      o = r; store = true;
    }
    if (i >= order) {
      const float2 xi = x[i];
      o.x = fmaf(xi.x, c0, o.x); o.y = fmaf(xi.y, c0, o.y);
      store = true;
    }
    if (store) y[i] = o;
    acc += o.x * o.x + o.y * o.y;
  }
  const float tot = block_sum<BLOCK>(acc, scratch);
  if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tot / (float)(unsigned)n);
}

// ---------------------------------------------------------------------------
// K_fm_block3 (round 4): k_fm_block's arithmetic (identical operation order, bit-identical results) out of LDS -- the block
// is walked in tiles, a tile's window (order history samples + the tile) and the coefficients are staged once; the AM / SSB
// filters have 255 / 2049 taps, and with the 48 kHz modes' short blocks almost every output takes the sequential block-head
// path (hazard H1) -- with FOUR consecutive outputs per lane and, for FM, the phase discriminator
// (k_disc: PhaseDiscriminator.cpp:33-46, the fp64 widening FmDecode.cpp:143, the block statistics Utility.h:135-152)
// as its epilogue.  The folded body (Filter.cpp:73-82) of output i is sum_k (x[i - k] + x[i - order + k]) c[k]: as k
// advances one window slides down and the other up, by one sample each -- a lane that owns four outputs keeps both
// windows in registers and reads ONE new sample per window and step (0.5 LDS reads per output and tap pair instead of
// 2, the coefficients four at a time), and the three operations of a pair (add, multiply, add -- the reference's
// rounding, no FMA) are packed over re / im.  Every output is still one accumulator chain over k in the reference's
// order: bit-identical to k_fm_block.  Head outputs (i < order, Filter.cpp:59-68, hazard H1) take the one-output
// code.  The discriminator's neighbour phase crosses lanes through LDS; a block's first difference needs the last
// output of the block before it, which belongs to another workgroup: the kernel leaves that one sample out, stores the
// phases of the block's first and last output and the block's sums without it, and k_disc_heads (one lane per block)
// finishes the block -- recomputing the neighbour's output on one lane, 127 dependent loads, made the kernel twice as
// slow as the two it replaces.
// Round 3's one-output-per-lane form (k_fm_block2, 94 us) + k_disc (32 us) per 5.2 M IF samples -> one kernel.
// ---------------------------------------------------------------------------
template <class XF>
__device__ __forceinline__ float2 fir_one(XF X /* X(j) = x[i - j] */, int i, int order, const float *cs) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int half_order = (order - 1) / 2;
  v2f y = {0.f, 0.f};                     // (packed over re / im: the same multiply, then add, per component)
  auto ldv = [&](int j) { const float2 t = X(j); return v2f{t.x, t.y}; };
  if (i < order) {
#pragma unroll 8
    for (int j = i + 1; j <= order; j++) { const v2f pr = ldv(j) * cs[j]; y = y + pr; }
#pragma unroll 8
    for (int j = 1; j <= i; j++) { const v2f pr = ldv(j) * cs[j]; y = y + pr; }
  } else {
#pragma unroll 4
    for (int k = 0; k <= half_order; k++) {
      const v2f sum = ldv(k) + ldv(order - k);
      const v2f pr = sum * cs[k];
      y = y + pr;
    }
    if ((order % 2) == 0) { const v2f pr = ldv(order / 2) * cs[order / 2]; y = y + pr; }
  }
  return make_float2(y.x, y.y);
}
// LDS layout of a tile's window for k_fm_block3: sample m in plane m & 3 at position m >> 2.  A lane that owns four
// consecutive outputs reads sample 4 lane + c: with the samples in order that is a stride of 32 bytes, an eight-way bank
// conflict; in planes the lanes of one instruction read consecutive positions of one plane.  A plane's length is 8 mod 32
// float2, so that lanes reading consecutive samples (the staging stores, the one-output code) spread over all banks too.
__host__ __device__ inline int fm_block3_plane(int order, int tl) {
  int p = (order + tl + 3) / 4;
  p += (8 - p % 32 + 32) % 32;
  return p;
}
template <int BLOCK, bool DISC>
// (waves_per_eu: 94 registers and five waves per SIMD instead of 112 and four)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_fm_block3(
    const float2 *__restrict__ ifb, long long if_stride, int if_halo, BlockTab bt,
    const float *__restrict__ coeff, int ntaps, int rms_after_fir,
    float2 *__restrict__ firb, long long fir_stride, float *__restrict__ if_rms_blk,
    float nf, float bound, float *__restrict__ dec, long long dec_stride,
    fm_mpx_t *__restrict__ base, long long base_stride, int base_off,
    float *__restrict__ bb_mean_blk /* DISC: the block's sum of d without its first sample */,
    float *__restrict__ bb_rms_blk /* DISC: the sum of d^2 likewise */, float *__restrict__ blk_ph /* [S][nb][2] */,
    int tl /* tile length: a multiple of 4, <= 4 BLOCK; the host sizes the dynamic LDS for it (short blocks -- the 48 kHz
              modes' 256 samples -- need a quarter of the 1024-output tile's LDS, i.e. four times the workgroups per CU) */) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int R = 4, TL = BLOCK * R;
  extern __shared__ float2 lds_fb[];
  __shared__ float scratch[BLOCK / 64];
  __shared__ float ph[DISC ? TL + 1 : 1];                           // ph[1 + t] = phase of the tile's output t, ph[0] the one before
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const int order = ntaps - 1;
  const int half_order = (order - 1) / 2, npairs = half_order + 1;
  const int PL = fm_block3_plane(order, tl);
  float2 *xs = lds_fb;                                              // [4 PL]: sample m of the window (x[i0 - order + m]) at XS(m)
  auto XS = [&](int m) -> float2 & { return xs[(m & 3) * PL + (m >> 2)]; };
  float *cs = reinterpret_cast<float *>(lds_fb + 4 * PL);           // [ntaps + 3 (+1)], 16-byte aligned
  float2 *xlin = lds_fb + 4 * PL + ((ntaps + 3 + 1) / 2);           // [order + tl]: the same window in order, for the one-output code
                                                                    // (lanes read consecutive samples there; in planes every read needs its own address arithmetic)
  const float2 *x = ifb + (long long)s * if_stride + if_halo + bt.if_off[b];
  float2 *y = firb + (long long)s * fir_stride + bt.if_off[b];
  for (int k = threadIdx.x; k < ntaps + 3; k += BLOCK) cs[k] = k < ntaps ? coeff[k] : 0.f;
  float acc = 0.f, vsum = 0.f, vsq = 0.f;
  float carry = 0.f;                                                // (lane BLOCK - 1: phase of the last output of the tile before)
  for (int i0 = 0; i0 < n; i0 += tl) {
    const int tn = min(tl, n - i0);
    __syncthreads();
    if (DISC && i0 > 0 && threadIdx.x == BLOCK - 1) ph[0] = carry;
    for (int k = threadIdx.x; k < order + tn; k += BLOCK) { const float2 v = x[i0 - order + k]; XS(k) = v; xlin[k] = v; }    // (reaches into the prefix halo)
    __syncthreads();
    const int t0 = R * threadIdx.x;
    auto emit = [&](int t, float2 o) {       // output t of the tile: the filtered sample, the level sum, the phase
      y[i0 + t] = o;
      const float2 v = rms_after_fir ? o : xlin[order + t];
      acc += v.x * v.x + v.y * v.y;
      if (DISC) ph[1 + t] = atan2f(o.y, o.x) / nf;                         // V4
    };
    // one output per lane: the head outputs (i < order: 2 x order dependent steps each -- four of them on one lane were most
    // of the block's time), the up to three body outputs that complete a group of four behind them, and the tile's last
    // outputs that do not fill a group
    const int hb = min(tn, max(0, (order - i0 + R - 1) & ~(R - 1))), tb = hb + ((tn - hb) & ~(R - 1));
    for (int t = threadIdx.x; t < hb + (tn - tb); t += BLOCK) {
      const int tt = t < hb ? t : tb + (t - hb);
      const float2 *xl = xlin + order + tt;
      emit(tt, fir_one([&](int j) { return xl[-j]; }, i0 + tt, order, cs));
    }
    if (t0 >= hb && t0 < tb) {
      // (t0 is a multiple of four: sample order + t0 + idx sits in plane (order + idx) & 3 at position t0 / 4 + ((order + idx) >> 2))
      const float2 *xq = xs + (t0 >> 2);
      {
        // ---- four body outputs: windows A (x[i - k], slides down) and B (x[i - order + k], slides up) in registers
        v2f ac[R];
#pragma unroll
        for (int r = 0; r < R; r++) ac[r] = v2f{0.f, 0.f};
        auto ldf = [&](int idx) { const int m = order + idx; return xq[(m & 3) * PL + (m >> 2)]; };
        auto ld = [&](int idx) { const float2 v = ldf(idx); return v2f{v.x, v.y}; };
        v2f A[4], B[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { A[r] = ld(r); B[r] = ld(r - order); }
        auto four_steps = [&](int k, int nsteps) {      // steps k .. k + nsteps - 1 (nsteps uniform, 1..4)
          const float4 c4 = *reinterpret_cast<const float4 *>(cs + k);
          const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (u < nsteps) {
              // step k + u: output r takes A sample (r - u) -> slot (r - u) & 3 (slots hold samples -k-u .. -k-u+3),
              //                             B sample (r + u) -> slot (r + u) & 3
#pragma unroll
              for (int r = 0; r < R; r++) {
                const v2f sum = A[(r - u) & 3] + B[(r + u) & 3];
                const v2f pr = sum * cc[u];
                ac[r] = ac[r] + pr;
              }
              // next step: A gains sample -(k + u) - 1 in the slot of the one it drops, B gains -order + k + u + 4
              A[(3 - u) & 3] = ld(-(k + u) - 1);
              B[u & 3] = ld(-order + k + u + 4);
            }
          }
        };
        int k = 0;
        for (; k + 4 <= npairs; k += 4) four_steps(k, 4);
        if (k < npairs) four_steps(k, npairs - k);
        if ((order % 2) == 0) {
          const float cm = cs[order / 2];
#pragma unroll
          for (int r = 0; r < R; r++) { const v2f pr = ld(r - order / 2) * cm; ac[r] = ac[r] + pr; }
        }
#pragma unroll
        for (int r = 0; r < R; r++) emit(t0 + r, make_float2(ac[r].x, ac[r].y));
      }
    }
    if (DISC) {
      __syncthreads();
      // the tile's last phase moves to lane BLOCK - 1 for the next tile
      const float last = ph[tn];
      if (threadIdx.x == BLOCK - 1) carry = last;
      if (t0 < tn) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int t = t0 + r;
          if (t < tn && i0 + t == 0) blk_ph[((long long)s * bt.nb + b) * 2] = ph[1];
          if (t < tn && i0 + t > 0) {
            float d = ph[1 + t] - ph[t];                               // V5
            if (d > bound) d -= 2 * bound;
            if (d < -bound) d += 2 * bound;
            if (isnan(d)) d = 0.f;                                     // Utility.h:336-343
            const int i = i0 + t;
            dec[(long long)s * dec_stride + bt.if_off[b] + i] = d;
            base[(long long)s * base_stride + base_off + bt.if_off[b] + i] = d;
            vsum += d;
            vsq += d * d;
          }
          if (t < tn && i0 + t == n - 1) blk_ph[((long long)s * bt.nb + b) * 2 + 1] = ph[1 + t];
        }
      }
    }
  }
  const float tot = block_sum<BLOCK>(acc, scratch);
  if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tot / (float)(unsigned)n);
  if (DISC) {
    const float ts = block_sum<BLOCK>(vsum, scratch);
    const float tq = block_sum<BLOCK>(vsq, scratch);
    if (threadIdx.x == 0) { bb_mean_blk[(long long)s * bt.nb + b] = ts; bb_rms_blk[(long long)s * bt.nb + b] = tq; }
  }
}
// the first sample of every block behind k_fm_block3<.., true>: its difference against the last phase of the block before
// (or the carried phase, PhaseDiscriminator.cpp:33-46), the block's statistics, the phase the call leaves behind
__global__ void k_disc_heads(BlockTab bt, const float *__restrict__ blk_ph, float bound,
                             float *__restrict__ dec, long long dec_stride, fm_mpx_t *__restrict__ base, long long base_stride,
                             int base_off, float *__restrict__ bb_mean_blk, float *__restrict__ bb_rms_blk, StreamState *st) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
  if (b >= bt.nb) return;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const float *pp = blk_ph + (long long)s * bt.nb * 2;
  int pb = b - 1;
  while (pb >= 0 && bt.if_len[pb] == 0) pb--;
  const float prev = pb < 0 ? st[s].disc_save : pp[2 * pb + 1];
  float d = pp[2 * b] - prev;                                          // V5
  if (d > bound) d -= 2 * bound;
  if (d < -bound) d += 2 * bound;
  if (isnan(d)) d = 0.f;                                               // Utility.h:336-343
  dec[(long long)s * dec_stride + bt.if_off[b]] = d;
  base[(long long)s * base_stride + base_off + bt.if_off[b]] = d;
  const long long bi = (long long)s * bt.nb + b;
  bb_mean_blk[bi] = (bb_mean_blk[bi] + d) / (float)(unsigned)n;
  bb_rms_blk[bi] = sqrtf((bb_rms_blk[bi] + d * d) / (float)(unsigned)n);
  int nb2 = b + 1;
  while (nb2 < bt.nb && bt.if_len[nb2] == 0) nb2++;
  if (nb2 >= bt.nb) { st[s].disc_save_next = pp[2 * b + 1]; st[s].disc_save_valid = 1; }
}

// ---------------------------------------------------------------------------
// K_finetune : FineTuner::process (FineTuner.cpp:55-73), the table-driven mixer of the SSB / CW / WSPR modes
// (AmDecode.cpp:107-136), in place, one workgroup per block.  idx0 = table index of the call's first sample
// (the reference's m_index).  With if_rms_blk it also takes the block RMS of its output (AmDecode.cpp:154 measures
// the IF level after the last mixer), same lane order as k_fm_block.
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_finetune(float2 *__restrict__ buf, long long stride, int off0, BlockTab bt,
                                                    const float2 *__restrict__ table, int table_size, unsigned idx0,
                                                    float *__restrict__ if_rms_blk) {
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  float2 *x = buf + (long long)s * stride + off0 + bt.if_off[b];
  const unsigned base = (idx0 + (unsigned)bt.if_off[b]) % (unsigned)table_size;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    const float2 v = x[i];
    const float2 t = table[(base + (unsigned)i) % (unsigned)table_size];
    const float re = v.x * t.x - v.y * t.y;       // std::complex<float> operator*
    const float im = v.x * t.y + v.y * t.x;
    x[i] = make_float2(re, im);
    acc += re * re + im * im;
  }
  if (if_rms_blk) {
    const float tot = block_sum<BLOCK>(acc, scratch);
    if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tot / (float)(unsigned)n);
  }
}

// ---------------------------------------------------------------------------
// K_agc : IfSimpleAgc (IfSimpleAgc.cpp:37-57), nonlinear serial recurrence,
// one lane per stream.  Emits the gain applied to each sample; consumers form
// x*g themselves (same two float multiplies as the reference).
// ---------------------------------------------------------------------------
// progress != nullptr (FM with the equaliser, round 3): the equaliser kernel runs BESIDE this one on another stream and
// consumes the gains as they appear -- every gain is stored write-through (agent scope) and, every 256 samples, the
// count of finished samples of THIS call is published in progress[s] (zeroed at the head of the call) behind a vmcnt(0)
// wait.  k_mpf4 polls it before it loads a chunk.  A serial recurrence of 80 ns per sample in front of a serial
// recurrence of 240 ns per sample was a quarter of the call (12.9 of 52 ms per 161 k IF samples).
__global__ void k_if_agc(const float2 *__restrict__ x, long long x_stride, int x_off, int n,
                         float *__restrict__ gain, long long g_stride, StreamState *st, int n_streams,
                         float initial_gain, float max_gain, float rate,
                         unsigned long long *__restrict__ progress = nullptr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_streams) return;
  const float2 *xs = x + (long long)s * x_stride + x_off;
  float *gs = gain + (long long)s * g_stride;
  float g = st[s].agc_gain;
  const double r = (double)rate;
  auto put = [&](int i, float v) {
    if (progress) __hip_atomic_store(gs + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else gs[i] = v;
  };
  auto publish = [&](int done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(progress + s, (unsigned long long)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const float2 v0 = xs[i], v1 = xs[i + 1], v2 = xs[i + 2], v3 = xs[i + 3];
    const float2 vv[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      put(i + u, g);
      const float xr = vv[u].x * g, xi = vv[u].y * g;
      const float nrm = xr * xr + xi * xi;
      const float z = (float)(1.0 + (r * (1.0 - (double)nrm)));
      g *= z;
      if (!isfinite(g)) g = initial_gain;
      else if (g > max_gain) g = max_gain;
    }
    if (progress && ((i + 4) & 255) == 0) publish(i + 4);
  }
  for (; i < n; i++) {
    const float2 v = xs[i];
    put(i, g);
    const float xr = v.x * g, xi = v.y * g;
    const float nrm = xr * xr + xi * xi;
    const float z = (float)(1.0 + (r * (1.0 - (double)nrm)));
    g *= z;
    if (!isfinite(g)) g = initial_gain;
    else if (g > max_gain) g = max_gain;
  }
  st[s].agc_gain = g;
  if (progress) publish(n);
}

// The same recurrence with one WAVE per stream, for the equaliser's chain (k_mpf4 runs beside it and consumes the gains as
// they are published).  The recurrence is serial whatever runs it; what one lane per stream pays on top is a memory
// instruction per sample (a 4-byte write-through store, an 8-byte load) -- 168 ns per sample beside the equaliser.  Here
// the wave loads 64 samples with one instruction, every lane runs the same chain on v_readlane'd samples, lane u keeps
// the gain of sample u, and one 256-byte store writes 64 gains: ~17 instructions per sample, none of them memory.
// Bit-identical to k_if_agc (same operations in the same order on the same values).
__global__ __launch_bounds__(64) void k_if_agc_wave(const float2 *__restrict__ x, long long x_stride, int x_off, int n,
                                                    float *__restrict__ gain, long long g_stride, StreamState *st,
                                                    float initial_gain, float max_gain, float rate,
                                                    unsigned long long *__restrict__ progress) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const float2 *xs = x + (long long)s * x_stride + x_off;
  float *gs = gain + (long long)s * g_stride;
  float g = st[s].agc_gain;
  const double r = (double)rate;
  auto ld = [&](int i0) { const int i = i0 + lane; return i < n ? xs[i] : make_float2(0.f, 0.f); };
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  float2 v = ld(0);
  for (int i0 = 0; i0 < n; i0 += 64) {
    const float2 vn = ld(i0 + 64);
    const int cnt = min(64, n - i0);
    float gv = 0.f;
    auto chain = [&](float sx, float sy) {
      const float xr = sx * g, xi = sy * g;
      const float nrm = xr * xr + xi * xi;
      const float z = (float)(1.0 + (r * (1.0 - (double)nrm)));
      const float g2 = g * z;
      const float gm = (g2 > max_gain) ? max_gain : g2;       // (selects, not branches: the values are uniform and the
      g = isfinite(g2) ? gm : initial_gain;                    //  compiler would branch on them, 64 times per pass)
    };
    if (cnt == 64) {
      auto body = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const int sg = __builtin_amdgcn_readfirstlane(__float_as_int(g));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(gv) : "s"(sg), "n"(u));
        chain(rl(v.x, u), rl(v.y, u));
      };
      auto all = [&](auto... uc) { (body(uc), ...); };
      auto run = [&](auto base) {
        constexpr int B = decltype(base)::value;
        all(std::integral_constant<int, B>{}, std::integral_constant<int, B + 1>{}, std::integral_constant<int, B + 2>{},
            std::integral_constant<int, B + 3>{}, std::integral_constant<int, B + 4>{}, std::integral_constant<int, B + 5>{},
            std::integral_constant<int, B + 6>{}, std::integral_constant<int, B + 7>{});
      };
      run(std::integral_constant<int, 0>{}); run(std::integral_constant<int, 8>{}); run(std::integral_constant<int, 16>{});
      run(std::integral_constant<int, 24>{}); run(std::integral_constant<int, 32>{}); run(std::integral_constant<int, 40>{});
      run(std::integral_constant<int, 48>{}); run(std::integral_constant<int, 56>{});
    } else {
      for (int u = 0; u < cnt; u++) {
        gv = (lane == u) ? g : gv;
        chain(rl(v.x, u), rl(v.y, u));
      }
    }
    if (lane < cnt) {
      if (progress) __hip_atomic_store(gs + i0 + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else gs[i0 + lane] = gv;
    }
    const int done = i0 + cnt;
    if (progress && ((done & 255) == 0 || done == n)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(progress + s, (unsigned long long)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    v = vn;
  }
  if (lane == 0) st[s].agc_gain = g;
  if (progress && n == 0 && lane == 0) __hip_atomic_store(progress + s, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef FMR_AB_PARTNERS
// test hook (FMR_TEST_AGC_LATE): keeps a stream busy for `ticks` of the 100 MHz clock, so that the AGC kernel behind it
// starts late and the equaliser beside it has to wait (tests/test_gpu_configs.py)
__global__ void k_hold_stream(unsigned long long ticks) {
  const unsigned long long t_lim = wall_clock64() + ticks;
  while (wall_clock64() < t_lim) __builtin_amdgcn_s_sleep(64);
}
#endif

// ---------------------------------------------------------------------------
// K_mpf : MultipathFilter (MultipathFilter.cpp:92-197), constant-modulus NLMS.  Serial over samples (taps depend on
// previous outputs); the update cadence restarts in every block (hazard H2).  The product form is k_mpf4 below
// (a chain wave and three helpers); the earlier forms and their measurements are in NOTEBOOK.md.
// ---------------------------------------------------------------------------
#define FMR_MPF_CH 2048
// Wave sum without the LDS crossbar: four DPP steps sum each row of 16 lanes (quad_perm xor 1, xor 2, then the
// row_half_mirror / row_mirror pairings), four v_readlane add the rows.  Every lane returns the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp_add = [](float x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xF, 0xF, true));
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
  auto rl = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
  const float r0 = rl(v, 0), r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
  return (r0 + r1) + (r2 + r3);
}





// Complex multiply-accumulates are two v_pk_fma_f32 with op_sel / neg_lo (same rounding as four fmaf).
__device__ __forceinline__ void mpf_cmac(float __attribute__((ext_vector_type(2))) &acc, float2 sv, float2 cv) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f s = {sv.x, sv.y}, c = {cv.x, cv.y};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(s), "v"(c));                                    // += s.x * (c.x, c.y)
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(s), "v"(c));      // += s.y * (-c.y, c.x)
}
__device__ __forceinline__ float row_sum_dpp(float v) {        // every lane of a 16-lane row gets the row's sum
  auto dpp_add = [](float x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xF, 0xF, true));
  };
  v = dpp_add(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

// ---------------------------------------------------------------------------
// K_mpf v4 (round 4): a CHAIN wave and three HELPER waves, no barrier inside a chunk.
// The cycle account of round 3's kernel (four waves that met in every group; profiles/r04_mpf_account.txt: 2100 cycles per group of four samples, of which the ten LDS reads
// and the complex MACs are 150; the exchange between the four waves through LDS and a barrier, the error / factor chain
// and the coefficient update the rest) says what a group costs is not its arithmetic but that four waves meet in every
// group -- and that a single wave issues one instruction every four to five cycles whatever its kind, so the chain's
// INSTRUCTION COUNT is its time.  The coefficients only change after every fourth sample (MultipathFilter.cpp:176,186),
// and only the output of THAT sample feeds the update, so:
//   * wave 0, the chain: holds all N taps (TPL per lane), computes the dot product of the group's update sample alone
//     (2 TPL packed FMAs per lane, then ONE reduction for both components: v_permlane32_swap folds the real parts
//     into lanes 0-31 and the imaginary parts into lanes 32-63, four DPP steps and a row_bcast:15 finish it, two
//     v_readlane fetch the result), error -> factor (:115-135), updates its taps (:139; 2 TPL packed FMAs) and writes them
//     into a ring of coefficient snapshots in LDS;
//   * waves 1..3, the helpers: each computes ONE of the group's other three outputs from the snapshot of that group --
//     behind the chain, which never waits for them (it looks at their progress every RING/2-th group: the ring must not
//     lap them; they look at the chain's progress only when they have caught up with what they last saw).
// What keeps the chain short:
//   * tap (j, lane) is coefficient (ref + 64 j + lane) mod 64 TPL: the reference tap (:158) is lane 0 of j = 0 for every
//     equaliser length, its reset two v_cndmask; a lane without a tap (index >= N) reads a zero cell instead of the
//     window (address = position * stride + base with stride 0), so its coefficient stays zero without a select;
//   * nobody looks for non-finite values (:182-184,190-192) inside the chunk: once an output or an error is not finite
//     everything behind it is not either (the taps are), so the chunk's epilogue scans the outputs and the errors for the
//     FIRST bad one, restores what the reference would have left (the state up to that sample, the error of the last
//     update before it) and ends the block;
//   * mu of every update position (:130) comes from a prefix sum of |x|^2 over the chunk (fp64), not from a window sum per
//     position;
//   * flags and snapshots are read and written through address-space-3 pointers: a volatile access through a generic
//     pointer is a FLAT instruction, 400 cycles each on LDS.
// Summation order of a dot product: per lane two chains over its taps (j even / odd), then the wave sum; the update is
// fused multiply-adds where the reference rounds the products first -- as far from the reference's VOLK kernels as
// round 3's order was, and within the same tolerances (hazard H7).
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) int fmr_lds_int;
typedef float fmr_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) fmr_v2f fmr_lds_v2f;
__device__ __forceinline__ int lds_ld_volatile(const int *p) { return *(const volatile fmr_lds_int *)(fmr_lds_int *)(p); }
__device__ __forceinline__ void lds_st_volatile(int *p, int v) { *(volatile fmr_lds_int *)(fmr_lds_int *)(p) = v; }
__device__ __forceinline__ float2 lds_ld_volatile2(const float2 *p) {
  const fmr_v2f v = *(const volatile fmr_lds_v2f *)(fmr_lds_v2f *)(p);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ unsigned lds_offset(const void *p) { return (unsigned)(size_t)(const fmr_lds_int *)(p); }
__device__ __forceinline__ float2 lds_ld2_at(unsigned byte_off) {
  const fmr_v2f v = *(const fmr_lds_v2f *)(size_t)byte_off;
  return make_float2(v.x, v.y);
}
// c += conj(s) * f   (:139: c.x += s.x f.x + s.y f.y,  c.y += s.x f.y - s.y f.x)
__device__ __forceinline__ void mpf_cupd(fmr_v2f &c, float2 sv, fmr_v2f f) {
  const fmr_v2f s = {sv.x, sv.y};
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "v"(s), "v"(f));                                   // += s.x * (f.x, f.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(c) : "v"(s), "v"(f));      // += s.y * (f.y, -f.x)
}
__device__ __forceinline__ void mpf_cmac2(fmr_v2f &acc, float2 sv, fmr_v2f c) {     // acc += s * c
  const fmr_v2f s = {sv.x, sv.y};
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(s), "v"(c));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(s), "v"(c));
}
// sum of a complex value over the wave, as two scalars
__device__ __forceinline__ void wave_sum_c(fmr_v2f a, float &sx, float &sy) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(a.y), false, false);
  float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);     // lanes 0-31: real parts of lanes l and l + 32; lanes 32-63: imaginary
  auto dpp_add = [](float x, auto ctrl, auto rows) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, true));
  };
  t = dpp_add(t, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});    // quad_perm [1,0,3,2]
  t = dpp_add(t, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});    // quad_perm [2,3,0,1]
  t = dpp_add(t, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});   // row_half_mirror
  t = dpp_add(t, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});   // row_mirror: every lane of a row = the row's sum
  t = dpp_add(t, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast:15 into rows 1 and 3
  sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16));
  sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 48));
}
template <int TPL>
__global__ __launch_bounds__(256) void k_mpf4(
    const float2 *__restrict__ xin, long long x_stride, int x_off,
    const float *__restrict__ gain, long long g_stride, BlockTab bt,
    float2 *__restrict__ out, long long out_stride, float2 *__restrict__ coeff_g,
    float2 *__restrict__ state_g, int N, int ref, int *__restrict__ mpf_ok, StreamState *st,
    const unsigned long long *__restrict__ progress, unsigned wait_ticks) {
  constexpr int NT = 256, CH = FMR_MPF_CH, NG = CH / 4 + 2, M = 64 * TPL;
  constexpr int RING = (TPL <= 10) ? 16 : 8, LOOK = RING / 2;           // (the ring is the largest array: 160 KB of LDS hold 8 slots of 1280 taps)
  constexpr int NOBAD = 0x7fffffff;
  extern __shared__ float2 lds_m[];
  float2 *xw = lds_m;                                                     // [N + CH + 8]
  float *smu = reinterpret_cast<float *>(xw + N + CH + 8);                // [NG]
  float2 *yo = reinterpret_cast<float2 *>(smu + NG);                      // [CH] outputs of the chunk
  double *errs = reinterpret_cast<double *>(yo + CH);                     // [NG] error after update k of the chunk
  float2 *snap = reinterpret_cast<float2 *>(errs + NG);                   // [RING][M] coefficients of group g in slot g % RING
  int *ctl = reinterpret_cast<int *>(snap + RING * M);                    // [0] updates published by the chain, [1..3] groups done
                                                                          // by helper h, [4] first bad event of the chunk
  float2 *zcell = reinterpret_cast<float2 *>(ctl + 16);                   // [1] zero: what a lane without a tap reads
  double *psum = reinterpret_cast<double *>(snap + M);                    // [N + CH + 1] prefix sums of |x|^2 (slots 1.. of the ring,
  double *ptot = psum + (N + CH + 2);                                     //  free until the chain starts) and [NT] segment totals
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float2 *xs = xin + (long long)s * x_stride + x_off;
  const float *gs = gain + (long long)s * g_stride;
  float2 *os = out + (long long)s * out_stride;
  float2 *cg = coeff_g + (long long)s * N;
  float2 *sg = state_g + (long long)s * N;
  fmr_v2f c[TPL];
  unsigned wstr[TPL], wbas[TPL];          // window address of tap j at chunk position p: p * wstr + wbas (bytes)
  int tapi[TPL];
#pragma unroll
  for (int j = 0; j < TPL; j++) {
    const int i = (ref + 64 * j + lane) % M;
    const bool v = i < N;
    tapi[j] = v ? i : -1;
    const float2 cv = v ? cg[i] : make_float2(0.f, 0.f);
    c[j] = fmr_v2f{cv.x, cv.y};
    wstr[j] = v ? 8u : 0u;
    wbas[j] = v ? lds_offset(xw + 1 + i) : lds_offset(zcell);
  }
  if (tid == 0) zcell[0] = make_float2(0.f, 0.f);
  double err_last = st[s].mpf_error;
  unsigned resets = st[s].mpf_resets;
  bool gave_up = false;
  auto ldwin = [&](float2 (&d)[TPL], int p) {
#pragma unroll
    for (int j = 0; j < TPL; j++) d[j] = lds_ld2_at(__umul24((unsigned)p, wstr[j]) + wbas[j]);
  };
  auto dot = [&](const float2 (&sv)[TPL], float &yx, float &yy) {
    fmr_v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TPL; j++) mpf_cmac2((j & 1) ? a1 : a0, sv[j], c[j]);
    wave_sum_c(a0 + a1, yx, yy);
  };
  for (int b = 0; b < bt.nb; b++) {
    const int n = bt.if_len[b];
    int ok = 1;
    if (n == 0 || !bt.mpf_active[b]) {
      if (tid == 0) mpf_ok[(long long)s * bt.nb + b] = 0;
      continue;
    }
    const int off = bt.if_off[b];
    for (int i = tid; i < N; i += NT) xw[i] = sg[i];
    __syncthreads();
    for (int c0 = 0; c0 < n && ok; c0 += CH) {
      const int cn = min(CH, n - c0);
      if (progress && !gave_up) {       // (the AGC kernel beside this one: k_if_agc_wave)
        if (tid == 0) {
          const unsigned long long need = (unsigned long long)(off + c0 + cn);
          const unsigned long long t_lim = wall_clock64() + wait_ticks;
          bool there = false;
          for (;;) {
            if (__hip_atomic_load(progress + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { there = true; break; }
            if (wall_clock64() > t_lim) break;
            __builtin_amdgcn_s_sleep(8);
          }
          if (!there) { st[s].agc_sync_timeouts++; ctl[0] = -1; } else ctl[0] = 0;
        }
        __syncthreads();
        gave_up = ctl[0] < 0;
        __syncthreads();
      }
      for (int i = tid; i < cn; i += NT) {
        const float2 v = xs[off + c0 + i];
        const float g = progress ? __hip_atomic_load(gs + off + c0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gs[off + c0 + i];
        xw[N + i] = make_float2(v.x * g, v.y * g);
      }
      if (tid < 8) xw[N + cn + tid] = make_float2(0.f, 0.f);
      if (tid < 16) ctl[tid] = (tid == 4) ? NOBAD : 0;
      if (w == 0) {                                                  // the coefficients the chunk starts with: snapshot 0
#pragma unroll
        for (int j = 0; j < TPL; j++) snap[64 * j + lane] = make_float2(c[j].x, c[j].y);
      }
      __syncthreads();
      // mu of every update position of the chunk: p = q0 + 4 k, window xw[p + 1 .. p + N] = psum[p + 1 + N] - psum[p + 1]
      const int q0 = (4 - (c0 & 3)) & 3;
      const int nu = (q0 < cn) ? (cn - q0 + 3) / 4 : 0;
      {
        const int L = N + cn, seg = (L + NT - 1) / NT, i0 = tid * seg, i1 = min(L, i0 + seg);
        double acc = 0.0;
        for (int i = i0; i < i1; i++) { const float2 v = xw[i]; acc += (double)(v.x * v.x + v.y * v.y); }
        ptot[tid] = acc;
        __syncthreads();
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int t = 0;
        for (; t + 4 <= tid; t += 4) { a0 += ptot[t]; a1 += ptot[t + 1]; a2 += ptot[t + 2]; a3 += ptot[t + 3]; }
        for (; t < tid; t++) a0 += ptot[t];
        double run = (a0 + a1) + (a2 + a3);
        for (int i = i0; i < i1; i++) { psum[i] = run; const float2 v = xw[i]; run += (double)(v.x * v.x + v.y * v.y); }
        if (i1 == L && i0 < i1) psum[L] = run;
        __syncthreads();
        for (int k = tid; k < nu; k += NT) {
          const int p = q0 + 4 * k;
          const double e = psum[p + 1 + N] - psum[p + 1];
          smu[k] = (float)(0.1 / ((double)(float)e + 1e-10));        // :130
        }
        __syncthreads();
      }
      // ---- the groups of the chunk: group k = the outputs behind update k - 1 up to and including update position
      // q0 + 4 k; the last group (k = nu) is what follows the last update position, without an update
      const int u_last = q0 + 4 * (nu - 1);          // (nu = 0: -4 or less, every sample is in the last group)
      const bool has_tail = cn - 1 > u_last;
      if (w == 0) {
        // ================================================================ chain
        float2 sA[TPL], sB[TPL];
        auto step = [&](const float2 (&sl)[TPL], float2 (&sn)[TPL], int k) {
          const int p = q0 + 4 * k;
          const float mu = smu[k];
          ldwin(sn, (k + 1 < nu) ? p + 4 : cn - 1);        // the next group's window: issued now, used after this group's update
          float yx, yy;
          dot(sl, yx, yy);
          yo[p] = make_float2(yx, yy);
          const double env = (double)(yx * yx + yy * yy);
          const double error = 1.0 - env;
          const float factor = (float)(error * (double)mu);         // :133
          const fmr_v2f f = {factor * yx, factor * yy};
#pragma unroll
          for (int j = 0; j < TPL; j++) mpf_cupd(c[j], sl[j], f);
          if (lane == 0) c[0] = fmr_v2f{1.f, 0.f};                   // :158
          float2 *sp = snap + ((k + 1) & (RING - 1)) * M;            // the coefficients of group k + 1
#pragma unroll
          for (int j = 0; j < TPL; j++) sp[64 * j + lane] = make_float2(c[j].x, c[j].y);
          errs[k] = error;
          asm volatile("" ::: "memory");     // (LDS executes a wave's operations in order: the flag lands behind the snapshot)
          lds_st_volatile(ctl, k + 1);
          if (((k + 1) & (LOOK - 1)) == 0) {           // the ring must not lap the helpers
            for (;;) {
              const int h1 = lds_ld_volatile(ctl + 1), h2 = lds_ld_volatile(ctl + 2), h3 = lds_ld_volatile(ctl + 3);
              if (min(h1, min(h2, h3)) + (RING - LOOK - 1) >= k + 1) break;   // helpers read slot >= k + 1 - (LOOK - 1) - ..., the chain writes <= k + LOOK
              __builtin_amdgcn_s_sleep(1);
            }
          }
        };
        ldwin(sA, nu > 0 ? q0 : cn - 1);
        int k = 0;
        for (; k + 2 <= nu; k += 2) { step(sA, sB, k); step(sB, sA, k + 1); }
        if (k < nu) {
          step(sA, sB, k);
#pragma unroll
          for (int j = 0; j < TPL; j++) sA[j] = sB[j];
        }
        if (has_tail) { float yx, yy; dot(sA, yx, yy); yo[cn - 1] = make_float2(yx, yy); }
      } else {
        // ================================================================ helper w: the output w samples before the group's last
        int known = 0;          // updates the chain had published when last looked at
        fmr_v2f keep[TPL];
#pragma unroll
        for (int j = 0; j < TPL; j++) keep[j] = c[j];
        for (int g = 0; g <= nu; g++) {
          if (g == nu && !has_tail) break;
          const int p_end = (g < nu) ? q0 + 4 * g : cn - 1, p_lo = (g > 0) ? q0 + 4 * (g - 1) : -1;
          const int p = p_end - w;                                  // my sample of this group (if it belongs to it)
          if (p > p_lo) {
            float2 sv[TPL];
            ldwin(sv, p);
            if (g > known) {                                        // caught up with what I last saw of the chain: look again
              do { known = lds_ld_volatile(ctl); if (known < g) __builtin_amdgcn_s_sleep(1); } while (known < g);
            }
            const float2 *sp = snap + (g & (RING - 1)) * M;
#pragma unroll
            for (int j = 0; j < TPL; j++) { const float2 cv = lds_ld_volatile2(sp + 64 * j + lane); c[j] = fmr_v2f{cv.x, cv.y}; }
            float yx, yy;
            dot(sv, yx, yy);
            yo[p] = make_float2(yx, yy);
          }
          lds_st_volatile(ctl + w, g + 1);
        }
#pragma unroll
        for (int j = 0; j < TPL; j++) c[j] = keep[j];
      }
      __syncthreads();
      // ---- what the reference would have left behind: the earliest bad sample ends the block (:182-184,190-192).
      // Event key = 2 * position + (1 if it is the error of that position's update that is not finite, its output being so)
      {
        int key = NOBAD;
        for (int i = tid; i < cn; i += NT) {
          const float2 y = yo[i];
          if (!(isfinite(y.x) && isfinite(y.y))) { key = 2 * i; break; }
        }
        for (int k = tid; k < nu; k += NT)
          if (!isfinite(errs[k])) { key = min(key, 2 * (q0 + 4 * k) + 1); break; }
        if (key != NOBAD) atomicMin(ctl + 4, key);
      }
      __syncthreads();
      const int key = ctl[4];
      int pushed = cn, stored = cn;
      if (key != NOBAD) {
        const int bad = key >> 1;
        ok = 0;
        pushed = bad + 1;
        stored = bad;                                                // (the block falls back to the AGC output anyway)
        if (key & 1) err_last = errs[(bad - q0) >> 2];               // :190-192: the error of THIS update stands
        else {
          const int gp = (bad <= q0) ? 0 : (bad - q0 + 3) >> 2;      // the group of the bad output: updates 0 .. gp - 1 came before it
          if (gp > 0) err_last = errs[gp - 1];
        }
      } else if (nu > 0) err_last = errs[nu - 1];
      // new state = last N entries pushed so far
      for (int i = tid; i < stored; i += NT) os[off + c0 + i] = yo[i];
      constexpr int TPS = (M + NT - 1) / NT + 1;
      float2 tmp[TPS];
#pragma unroll
      for (int j = 0; j < TPS; j++) { const int i = tid + NT * j; tmp[j] = (i < N) ? xw[pushed + i] : make_float2(0.f, 0.f); }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TPS; j++) { const int i = tid + NT * j; if (i < N) xw[i] = tmp[j]; }
      __syncthreads();
    }
    for (int i = tid; i < N; i += NT) sg[i] = xw[i];
    if (!ok) {
      // FmDecode.cpp:117-123: re-initialise the taps, block falls back to the AGC output
#pragma unroll
      for (int j = 0; j < TPL; j++) c[j] = fmr_v2f{tapi[j] == ref ? 1.f : 0.f, 0.f};
      resets++;
    }
    if (tid == 0) mpf_ok[(long long)s * bt.nb + b] = ok;
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < TPL; j++) if (tapi[j] >= 0) cg[tapi[j]] = make_float2(c[j].x, c[j].y);
  }
  if (tid == 0) { st[s].mpf_error = err_last; st[s].mpf_resets = resets; }
}

// ---------------------------------------------------------------------------
// K_disc : PhaseDiscriminator (PhaseDiscriminator.cpp:33-46) per decoder block,
// fused with the float->double widening (FmDecode.cpp:143) and the block
// mean / rms of the MPX signal (Utility.h:135-152).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float2 disc_src(const float2 *xs, const float *gs, const float2 *ms, int use_mpf, int idx) {
  if (use_mpf) return ms[idx];
  const float2 v = xs[idx];
  if (!gs) return v;               // gain applied elsewhere / not needed (atan2 is scale invariant)
  const float g = gs[idx];
  return make_float2(v.x * g, v.y * g);
}

template <int BLOCK, class MPX /* fm_mpx_t for FM, double for NBFM (whose audio path reads it as it is) */>
__global__ __launch_bounds__(BLOCK) void k_disc(
    const float2 *__restrict__ xin, long long x_stride, int x_off,
    const float *__restrict__ gain, long long g_stride,
    const float2 *__restrict__ mpfb, long long m_stride, const int *__restrict__ mpf_ok,
    BlockTab bt, float nf, float bound, float *__restrict__ dec, long long dec_stride,
    MPX *__restrict__ base, long long base_stride, int base_off,
    float *__restrict__ bb_mean_blk, float *__restrict__ bb_rms_blk, StreamState *st,
    float *__restrict__ if_rms_blk /* non-null: also the IF RMS of the block (k_fm_block's job when no IF FIR runs) */) {
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const int off = bt.if_off[b];
  const float2 *xs = xin + (long long)s * x_stride + x_off;
  const float *gs = gain ? gain + (long long)s * g_stride : nullptr;
  const float2 *ms = mpfb ? mpfb + (long long)s * m_stride : nullptr;
  const int use_mpf = mpfb ? mpf_ok[(long long)s * bt.nb + b] : 0;
  float vsum = 0.f, vsq = 0.f, rsq = 0.f;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    const float2 v = disc_src(xs, gs, ms, use_mpf, off + i);
    if (if_rms_blk) { const float2 raw = xs[off + i]; rsq += raw.x * raw.x + raw.y * raw.y; }   // same lane order as k_fm_block
    const float ph = atan2f(v.y, v.x) / nf;                      // V4
    float prev;
    if (i > 0) {
      const float2 p = disc_src(xs, gs, ms, use_mpf, off + i - 1);
      prev = atan2f(p.y, p.x) / nf;
    } else {
      int pb = b - 1;
      while (pb >= 0 && bt.if_len[pb] == 0) pb--;
      if (pb < 0) {
        prev = st[s].disc_save;
      } else {
        const int pu = mpfb ? mpf_ok[(long long)s * bt.nb + pb] : 0;
        const float2 p = disc_src(xs, gs, ms, pu, bt.if_off[pb] + bt.if_len[pb] - 1);
        prev = atan2f(p.y, p.x) / nf;
      }
    }
    float d = ph - prev;                                          // V5
    if (d > bound) d -= 2 * bound;
    if (d < -bound) d += 2 * bound;
    if (isnan(d)) d = 0.f;                                        // Utility.h:336-343
    dec[(long long)s * dec_stride + off + i] = d;
    base[(long long)s * base_stride + base_off + off + i] = (MPX)d;
    vsum += d;
    vsq += d * d;
    if (i == n - 1) {
      // is this the last non-empty block of the call?
      int nb2 = b + 1;
      while (nb2 < bt.nb && bt.if_len[nb2] == 0) nb2++;
      if (nb2 >= bt.nb) { st[s].disc_save_next = ph; st[s].disc_save_valid = 1; }
    }
  }
  const float ts = block_sum<BLOCK>(vsum, scratch);
  const float tq = block_sum<BLOCK>(vsq, scratch);
  if (if_rms_blk) {
    const float tr = block_sum<BLOCK>(rsq, scratch);
    if (threadIdx.x == 0) if_rms_blk[(long long)s * bt.nb + b] = sqrtf(tr / (float)(unsigned)n);
  }
  if (threadIdx.x == 0) {
    bb_mean_blk[(long long)s * bt.nb + b] = ts / (float)(unsigned)n;
    bb_rms_blk[(long long)s * bt.nb + b] = sqrtf(tq / (float)(unsigned)n);
  }
}

// AM: demodulate_am / demodulate_dsb (AmDecode.cpp:221-234) + widening (:190)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_am_demod(
    const float2 *__restrict__ xin, long long x_stride, int x_off,
    const float *__restrict__ gain, long long g_stride, BlockTab bt, int dsb,
    float *__restrict__ dec, long long dec_stride, double *__restrict__ demod, long long demod_stride,
    float *__restrict__ bb_mean_blk, float *__restrict__ bb_rms_blk) {
  __shared__ float scratch[BLOCK / 64];
  const int b = blockIdx.x, s = blockIdx.y;
  const int n = bt.if_len[b];
  if (n == 0) return;
  const int off = bt.if_off[b];
  const float2 *xs = xin + (long long)s * x_stride + x_off;
  const float *gs = gain + (long long)s * g_stride;
  float vsum = 0.f, vsq = 0.f;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    const float2 v = disc_src(xs, gs, nullptr, 0, off + i);
    const float d = dsb ? v.x : sqrtf(v.x * v.x + v.y * v.y);    // V12 / V11
    dec[(long long)s * dec_stride + off + i] = d;
    demod[(long long)s * demod_stride + off + i] = (double)d;
    vsum += d;
    vsq += d * d;
  }
  const float ts = block_sum<BLOCK>(vsum, scratch);
  const float tq = block_sum<BLOCK>(vsq, scratch);
  if (threadIdx.x == 0) {
    bb_mean_blk[(long long)s * bt.nb + b] = ts / (float)(unsigned)n;
    bb_rms_blk[(long long)s * bt.nb + b] = sqrtf(tq / (float)(unsigned)n);
  }
}

// ---------------------------------------------------------------------------
// K_stats : per-block scalar bookkeeping, one lane per stream: if_rms of the
// last block, the 0.95/0.05 EMAs (FmDecode.cpp:149-150, AmDecode.cpp:208-209),
// commit of the discriminator's carried phase.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Per-block statistics of the fused front end (kernels_fused.hpp) travel as partial sums over the 128 IF samples a
// stage-B wave finishes at a time, cut at the (at most one) block boundary inside them.
struct FusedPart {
  int blk[2];            // block of the samples before / after the cut (-1: none)
  float sum[2][3];       // sum d, sum d^2 (discriminator output), sum |x|^2 (IF) of each piece
};

// Pipelined chain, everything the front-end stage carries into its next call in ONE launch (a launch on the critical stream
// costs a few microseconds whatever it does): blockIdx.x = 0 the input history (k_update_in_halo, cf32), 1 the stage-B
// history (k_shift_halo of d_mid), 2 the discriminator's phase (k_disc_commit; commit = 0: the statistics kernel does it).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fe_post(float2 *__restrict__ in_halo, int H_in, const float2 *__restrict__ iq,
                                                    long long iq_stride, long long N_in, float2 *__restrict__ mid,
                                                    long long mid_stride, int H_mid, int N_mid, StreamState *st, int commit) {
  const int s = blockIdx.y;
  if (blockIdx.x == 2) {
    if (commit && threadIdx.x == 0 && st[s].disc_save_valid) { st[s].disc_save = st[s].disc_save_next; st[s].disc_save_valid = 0; }
    return;
  }
  float2 *h = blockIdx.x == 0 ? in_halo + (long long)s * H_in : mid + (long long)s * mid_stride;
  const int H = blockIdx.x == 0 ? H_in : H_mid;
  const long long N = blockIdx.x == 0 ? N_in : (long long)N_mid;
  if (N <= 0) return;
  const float2 *xs = iq + (long long)s * iq_stride;
  for (int c = 0; c < H; c += BLOCK) {       // newhalo[i] = concat(halo, data)[i + N]
    const int i = c + threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (i < H) {
      const long long j = (long long)i + N;
      v = (blockIdx.x == 1 || j < H) ? h[j] : xs[j - H];
    }
    __syncthreads();
    if (i < H) h[i] = v;
    __syncthreads();
  }
}

// one wave per stream: 64 block results per load, the EMA chain runs on SGPR broadcasts.  part != nullptr: the block
// values are summed here from the fused front end's pieces (index order: deterministic) instead of read from the
// per-block arrays k_disc writes -- mean / rms of the discriminator output (Utility.h:135-152) and the IF RMS
// (Utility.h:118-132, FmDecode.cpp:95).
#define FMR_STATS_THREADS 384
__global__ __launch_bounds__(FMR_STATS_THREADS) void k_stats(BlockTab bt, const float *__restrict__ if_rms_blk,
                                              const float *__restrict__ bb_mean_blk,
                                              const float *__restrict__ bb_rms_blk, StreamState *st, int n_streams,
                                              int has_disc, const FusedPart *__restrict__ part = nullptr, int n_tiles = 0,
                                              int kb_ref = 0) {
  // One workgroup per stream.  Phase 1, all waves: a lane per block fetches (or, behind the fused front end, sums from the
  // partial sums, index order: deterministic) the block's three values -- every load of up to 512 blocks in flight at
  // once; with one wave doing 64 blocks at a time this was 0.15-0.2 ms of memory latency.  Phase 2, wave 0: the EMA
  // chain over the blocks in order, from LDS (6 KB: the kernel fits beside the front end's workgroup).
  constexpr int NT = FMR_STATS_THREADS;
  __shared__ float s_r[NT], s_m[NT], s_l[NT];
  __shared__ int s_n[NT];
  const int s = blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (s >= n_streams) return;
  float m = st[s].baseband_mean, l = st[s].baseband_level, r = st[s].if_rms;
  // The EMAs forget: 0.95^400 = 1.2e-9 is below half a float ulp of the result, so only the last ~400 non-empty blocks
  // of a long call can be seen in the float state -- the walk starts there (it was 0.2 ms of one wave for 2048 blocks).
  int b_first = 0;
  {
    int seen = 0;
    for (int b0 = ((bt.nb - 1) / 64) * 64; b0 > 0; b0 -= 64) {
      const int b = b0 + lane;
      seen += __popcll(__ballot(b < bt.nb && bt.if_len[b] != 0));
      if (seen >= 400) { b_first = b0; break; }
    }
  }
  for (int b0 = b_first; b0 < bt.nb; b0 += NT) {
    const int b = min(b0 + (int)threadIdx.x, bt.nb - 1);
    const int my_n = (b0 + (int)threadIdx.x < bt.nb) ? bt.if_len[b] : 0;
    float my_r = 0.f, my_m = 0.f, my_l = 0.f;
    if (part) {
      float sd = 0.f, sq = 0.f, se = 0.f;
      if (my_n) {
        const int lo = bt.if_off[b], hi = lo + my_n - 1, ng = 3 * n_tiles;
        const int g0 = (lo - kb_ref) / 128, g1 = min((hi - kb_ref) / 128, ng - 1);   // thirds that hold samples of the block
        // (eight entries in flight per lane: a block spans ~20 thirds, and one at a time that was 20 memory round trips
        // in a row -- 0.13 ms on average beside the PLL's first pass.  Summed in the same index order.)
        for (int g = g0; g <= g1; g += 8) {
          FusedPart pt[8];
#pragma unroll
          for (int u = 0; u < 8; u++) pt[u] = part[(long long)s * ng + min(g + u, g1)];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (g + u > g1) break;
#pragma unroll
            for (int h = 0; h < 2; h++)
              if (pt[u].blk[h] == b) { sd += pt[u].sum[h][0]; sq += pt[u].sum[h][1]; se += pt[u].sum[h][2]; }
          }
        }
      }
      const float fn = (float)(unsigned)(my_n ? my_n : 1);
      my_m = sd / fn; my_l = sqrtf(sq / fn); my_r = sqrtf(se / fn);
    } else if (my_n) {
      my_r = if_rms_blk[(long long)s * bt.nb + b];
      my_m = bb_mean_blk[(long long)s * bt.nb + b];
      my_l = bb_rms_blk[(long long)s * bt.nb + b];
    }
    __syncthreads();                     // (the previous stage's chain has read its values)
    s_n[threadIdx.x] = my_n; s_r[threadIdx.x] = my_r; s_m[threadIdx.x] = my_m; s_l[threadIdx.x] = my_l;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int cnt = min(NT, bt.nb - b0);
      for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int vn = s_n[j0 + lane];
        const float vr = s_r[j0 + lane], vm = s_m[j0 + lane], vl = s_l[j0 + lane];
        const int c2 = min(64, cnt - j0);
        for (int j = 0; j < c2; j++) {
          if (__builtin_amdgcn_readlane(vn, j) == 0) continue;
          r = readlane_f(vr, j);
          m = (float)(0.95 * (double)m + 0.05 * (double)readlane_f(vm, j));
          l = (float)(0.95 * (double)l + 0.05 * (double)readlane_f(vl, j));
        }
      }
    }
  }
  if (threadIdx.x == 0) {
    st[s].baseband_mean = m; st[s].baseband_level = l; st[s].if_rms = r;
    if (has_disc && st[s].disc_save_valid) { st[s].disc_save = st[s].disc_save_next; st[s].disc_save_valid = 0; }
  }
}

// ---------------------------------------------------------------------------
// K_pll : PilotPhaseLock (PilotPhaseLock.cpp:56-171) + demod_stereo
// (FmDecode.cpp:224-239).  Nonlinear feedback loop: strictly serial per
// stream, one lane per stream; FP64 with the float table-driven fast_atan2f.
// ---------------------------------------------------------------------------
// Branch-free: the function sits on the loop-carried chain of every PLL sample step, and its eleven-way control flow
// compiled to exec-mask branches around every arm.  Same comparisons, same float operations in the same order on the
// arm that counts (one division, one table interpolation, one add / subtract from 0, pi or pi/2, one negation), so
// the result is bit-identical to the branchy form: a - b == a + (-b) and b - a == -(a - b) exactly in IEEE arithmetic,
// and 0 + b == b because b >= +0.
__device__ __forceinline__ float fast_atan2f_dev(float y, float x, const float *tab) {
  const float y_abs = fabsf(y), x_abs = fabsf(x);
  const bool any = (y_abs > 0.0f) || (x_abs > 0.0f);
  const bool ylt = y_abs < x_abs;
  const float z = (ylt ? y_abs : x_abs) / (ylt ? x_abs : y_abs);
  float alpha = z * 255.0f;
  const int index = ((int)alpha) & 0xff;
  alpha -= (float)index;
  const float t0 = tab[index], t1 = tab[index + 1];
  float base_angle = t0;
  base_angle += (t1 - t0) * alpha;
  // (double)z < 0.003921569 in the reference: for a float z that is z < 0x1.010104p-8f, the smallest float above the constant
  base_angle = (z < 0x1.010104p-8f) ? z : base_angle;
  const bool xgt = x_abs > y_abs, xpos = x >= 0.0f, ypos = y >= 0.0f;
  const float c = xgt ? (xpos ? 0.0f : 3.14159265358979323846f) : 1.57079632679489661923f;
  const bool minus = xgt ? !xpos : xpos;
  const float r = c + (minus ? -base_angle : base_angle);
  const float angle = ypos ? r : -r;
  return any ? angle : 0.0f;
}

__global__ __launch_bounds__(64) void k_pll(
    const fm_mpx_t *__restrict__ base, long long base_stride, int base_off, BlockTab bt,
    double *__restrict__ raw, long long raw_stride, int raw_off, const float *__restrict__ atan_tab,
    PllConst pc, int pilot_shift, int *__restrict__ stereo_blk, StreamState *st, int n_streams) {
  __shared__ float tab[257];
  for (int i = threadIdx.x; i < 257; i += blockDim.x) tab[i] = atan_tab[i];
  __syncthreads();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_streams) return;
  StreamState &S = st[s];
  const fm_mpx_t *xin = base + (long long)s * base_stride + base_off;
  double *out = raw + (long long)s * raw_stride + raw_off;
  double phase = S.pll_phase, freq = S.pll_freq, freq_err = S.pll_freq_err, level = S.pll_level;
  double i1 = S.bq_i_x1, i2 = S.bq_i_x2, q1 = S.bq_q_x1, q2 = S.bq_q_x2, lf1 = S.lf_x1;
  int lock_cnt = S.lock_cnt, pilot_periods = S.pilot_periods;
  unsigned long long pps_cnt = S.pps_cnt, sample_cnt = S.sample_cnt;
  int n_pps = 0;
  const double two_pi = 2.0 * 3.14159265358979323846;
  for (int b = 0; b < bt.nb; b++) {
    const int n = bt.if_len[b];
    if (n == 0) { stereo_blk[(long long)s * bt.nb + b] = (lock_cnt >= pc.lock_delay); continue; }
    const int off = bt.if_off[b];
    const bool was_locked = (lock_cnt >= pc.lock_delay);
    const int pps_blk_start = n_pps;
    for (int i = 0; i < n; i++) {
      double psin, pcos;
      sincos(phase, &psin, &pcos);
      const double x = xin[off + i];
      const double carrier = pilot_shift ? (2 * pcos * pcos - 1) : (2 * psin * pcos);
      out[off + i] = (carrier * x) * 2.0;                 // demod_stereo: V7 then adjust_gain
      const double phasor_i = psin * x, phasor_q = pcos * x;
      double w = phasor_i - (pc.bq_a1 * i1 + pc.bq_a2 * i2);
      const double new_i = pc.bq_b0 * w; i2 = i1; i1 = w;
      w = phasor_q - (pc.bq_a1 * q1 + pc.bq_a2 * q2);
      const double new_q = pc.bq_b0 * w; q2 = q1; q1 = w;
      const double phase_err = (double)fast_atan2f_dev((float)new_q, (float)new_i, tab);
      level = sqrt((new_i * new_i) + (new_q * new_q));
      const double new_err = pc.lf_b0 * phase_err + pc.lf_b1 * lf1;   // a1 == 0
      lf1 = phase_err;
      freq_err = new_err;
      freq += freq_err;
      freq = fmax(pc.minfreq, fmin(pc.maxfreq, freq));
      phase += freq;
      if (phase > two_pi) {
        phase -= two_pi;
        pilot_periods++;
        if (pilot_periods == pc.pilot_frequency) {
          pilot_periods = 0;
          if (was_locked) {
            if (n_pps < FMR_MAX_PPS) {
              PpsEventDev &ev = S.pps[n_pps];
              ev.pps_index = pps_cnt;
              ev.sample_index = sample_cnt + (unsigned long long)i;
              ev.block_position = (double)i / (double)n;
              ev.block = (unsigned)b;
            }
            n_pps++;
            pps_cnt++;
          }
        }
      }
    }
    if (2 * level > pc.minsignal) {                       // :154-160
      if (lock_cnt < pc.lock_delay) lock_cnt += n;
    } else {
      lock_cnt = 0;
    }
    if (lock_cnt < pc.lock_delay) {                       // :163-167
      pilot_periods = 0;
      pps_cnt = 0;
      n_pps = pps_blk_start;  // m_pps_events.clear(): this block's events only
    }
    sample_cnt += (unsigned long long)n;
    stereo_blk[(long long)s * bt.nb + b] = (lock_cnt >= pc.lock_delay);
  }
  S.pll_phase = phase; S.pll_freq = freq; S.pll_freq_err = freq_err; S.pll_level = level;
  S.bq_i_x1 = i1; S.bq_i_x2 = i2; S.bq_q_x1 = q1; S.bq_q_x2 = q2; S.lf_x1 = lf1;
  S.lock_cnt = lock_cnt; S.pilot_periods = pilot_periods; S.pps_cnt = pps_cnt; S.sample_cnt = sample_cnt;
  S.n_pps = n_pps < FMR_MAX_PPS ? n_pps : FMR_MAX_PPS;
  S.stereo_detected = (lock_cnt >= pc.lock_delay);
}



// ---------------------------------------------------------------------------
// Audio resampler (AudioResampler.cpp:37-61 stand-in), FP64, two channels:
// stage A integer decimation, stage B polyphase.  Sequential accumulation in
// tap order (bit-comparable with the oracle).
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_aud_decim(
    const double *__restrict__ x0, const double *__restrict__ x1, long long x_stride, int x_off,
    const double *__restrict__ hA, int NA, int D, long long top0, int count,
    double *__restrict__ y0, double *__restrict__ y1, long long y_stride, int y_off) {
  const int s = blockIdx.y, ch = blockIdx.z;
  const int m = blockIdx.x * BLOCK + threadIdx.x;
  if (m >= count) return;
  const double *x = (ch ? x1 : x0) + (long long)s * x_stride + x_off;
  double *y = (ch ? y1 : y0) + (long long)s * y_stride + y_off;
  const double *xp = x + top0 + (long long)m * D - (NA - 1);   // oldest tap first in memory
  double acc = 0.0;
  // acc += hA[k] * x[top - k], k ascending: walk memory backwards, 8 loads in flight
  int k = 0;
  for (; k + 8 <= NA; k += 8) {
    double xv[8], hv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { xv[u] = xp[(NA - 1) - (k + u)]; hv[u] = hA[k + u]; }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += hv[u] * xv[u];
  }
  for (; k < NA; k++) acc += hA[k] * xp[(NA - 1) - k];
  y[m] = acc;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_aud_poly(
    const double *__restrict__ m0, const double *__restrict__ m1, long long m_stride, long long mid_abs0,
    const double *__restrict__ hB, int TB, unsigned LB, unsigned MB, unsigned long long t0, int count,
    double *__restrict__ y0, double *__restrict__ y1, long long y_stride, int y_off) {
  const int s = blockIdx.y, ch = blockIdx.z;
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= count) return;
  const double *mid = (ch ? m1 : m0) + (long long)s * m_stride;
  double *y = (ch ? y1 : y0) + (long long)s * y_stride + y_off;
  const unsigned long long t = t0 + (unsigned long long)k * MB;
  const long long nk = (long long)(t / LB);
  const unsigned p = (unsigned)(t % LB);
  const double *h = hB + (size_t)p * TB;
  const double *xp = mid + (nk - (TB >> 1) + 1 - mid_abs0);
  double acc = 0.0;
  serial_prefetch2<8>(h, xp, 0, TB, [&](int, double hv, double xv) { acc += hv * xv; });
  y[k] = acc;
}

// Audio stage B, period form (small LB): one lane per PERIOD computes its LB outputs as LB
// interleaved accumulation chains (each still sequential in tap order -> bit-comparable with
// the oracle), taps are wave-uniform (scalar loads), and the mid-rate tile is staged in LDS
// de-interleaved by residue mod MB so that lanes read consecutive words (conflict-free).
template <int BLOCK, int LBT, int MB>
__global__ __launch_bounds__(BLOCK) void k_aud_poly2(
    const double *__restrict__ m0, const double *__restrict__ m1, long long m_stride, long long mid_abs0,
    const double *__restrict__ hB, int TB, long long k0, int count,
    double *__restrict__ y0, double *__restrict__ y1, long long y_stride, int y_off, int ni_pad, int mid_valid,
    int ch_base) {
  extern __shared__ __attribute__((aligned(16))) double lds_ap[];
  const int s = blockIdx.y, ch = blockIdx.z + ch_base;
  const int tid = threadIdx.x;
  const double *mid = (ch ? m1 : m0) + (long long)s * m_stride;
  double *y = (ch ? y1 : y0) + (long long)s * y_stride + y_off;
  const int W = TB >> 1;
  const long long Pt = k0 / LBT + (long long)blockIdx.x * BLOCK;       // first period of the tile
  const long long a0 = Pt * MB - W + 1;                                // absolute mid index of staged element 0
  int off[LBT], phi[LBT];
#pragma unroll
  for (int p = 0; p < LBT; p++) { off[p] = (p * MB) / LBT; phi[p] = (p * MB) % LBT; }
  const int lx = (BLOCK - 1) * MB + off[LBT - 1] + TB;                 // staged elements
  // ---- stage (coalesced) with de-interleave: element e -> row e % MB, column e / MB
  for (int e0 = 0; e0 < lx; e0 += 8 * BLOCK) {
    double v[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int e = e0 + t * BLOCK + tid;
      const long long idx = a0 + e - mid_abs0;      // tiles are period-aligned: they may stick out of the valid data
      v[t] = (e < lx && idx >= 0 && idx < mid_valid) ? mid[idx] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int e = e0 + t * BLOCK + tid;
      if (e < lx) lds_ap[(e % MB) * ni_pad + e / MB] = v[t];
    }
  }
  __syncthreads();
  double acc[LBT];
#pragma unroll
  for (int p = 0; p < LBT; p++) acc[p] = 0.0;
  // batches of 8 taps: 8 scalar tap loads and 8 LDS reads per chain are in flight before the
  // (ordered) multiply-adds consume them
  int j = 0;
  for (; j + 8 <= TB; j += 8) {
    double hv[LBT][8], xv[LBT][8];
#pragma unroll
    for (int p = 0; p < LBT; p++)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int jj = off[p] + j + u;                                  // wave-uniform
        hv[p][u] = hB[(size_t)phi[p] * TB + j + u];
        xv[p][u] = lds_ap[(jj % MB) * ni_pad + tid + jj / MB];
      }
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int p = 0; p < LBT; p++) acc[p] += hv[p][u] * xv[p][u];      // same order as the oracle
  }
  for (; j < TB; j++) {
#pragma unroll
    for (int p = 0; p < LBT; p++) {
      const int jj = off[p] + j;
      acc[p] += hB[(size_t)phi[p] * TB + j] * lds_ap[(jj % MB) * ni_pad + tid + jj / MB];
    }
  }
  const long long kb = (Pt + tid) * LBT - k0;
#pragma unroll
  for (int p = 0; p < LBT; p++) {
    const long long k = kb + p;
    if (k >= 0 && k < count) y[k] = acc[p];
  }
}

// ---------------------------------------------------------------------------
// K_pcut : LowPassFilterFirAudio (Filter.cpp:107-163), the 19 kHz pilot-cut
// FIR at 48 kHz, per audio block incl. the block-head path (hazard H1).
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pilotcut(
    const double *__restrict__ a0, const double *__restrict__ a1, long long a_stride, int a_halo,
    BlockTab bt, const double *__restrict__ coeff, int ntaps,
    double *__restrict__ p0, double *__restrict__ p1, long long p_stride) {
  const int b = blockIdx.x, s = blockIdx.y, ch = blockIdx.z;
  const int n = bt.au_len[b];
  if (n == 0) return;
  const double *x = (ch ? a1 : a0) + (long long)s * a_stride + a_halo + bt.au_off[b];
  double *y = (ch ? p1 : p0) + (long long)s * p_stride + bt.au_off[b];
  const int order = ntaps - 1, half_order = (order - 1) / 2;
  for (int i = threadIdx.x; i < n; i += BLOCK) {
    double acc = 0.0;
    if (i < order) {
      for (int j = i + 1; j <= order; j++) acc += x[i - j] * coeff[j];
      for (int j = 1; j <= i; j++) acc += x[i - j] * coeff[j];
    } else {
      for (int k = 0; k <= half_order; k++) acc += (x[i - k] + x[i - (order - k)]) * coeff[k];
      if ((order % 2) == 0) acc += x[i - order / 2] * coeff[order / 2];
    }
    y[i] = acc;
  }
}

// LDS-staged form of k_pilotcut: the block's samples (plus the `order` samples before it) and the taps
// sit in LDS, adjacent lanes own adjacent outputs (conflict-free ds_read_b64).  Same summation order.
#define FMR_PCUT_MAXTAPS 128
template <int BLOCK, int TILE>
__global__ __launch_bounds__(BLOCK) void k_pilotcut2(
    const double *__restrict__ a0, const double *__restrict__ a1, long long a_stride, int a_halo,
    BlockTab bt, const double *__restrict__ coeff, int ntaps,
    double *__restrict__ p0, double *__restrict__ p1, long long p_stride, double out_gain, int ch_base) {
  __shared__ double cs[FMR_PCUT_MAXTAPS];
  __shared__ double xs[TILE + FMR_PCUT_MAXTAPS];
  const int b = blockIdx.x, s = blockIdx.y, ch = blockIdx.z + ch_base;
  const int n = bt.au_len[b];
  if (n == 0) return;
  const double *x = (ch ? a1 : a0) + (long long)s * a_stride + a_halo + bt.au_off[b];
  double *y = (ch ? p1 : p0) + (long long)s * p_stride + bt.au_off[b];
  const int order = ntaps - 1, half_order = (order - 1) / 2;
  for (int j = threadIdx.x; j < ntaps; j += BLOCK) cs[j] = coeff[j];
  for (int t0 = 0; t0 < n; t0 += TILE) {
    const int tn = min(TILE, n - t0);
    __syncthreads();
    for (int m = threadIdx.x; m < tn + order; m += BLOCK) xs[m] = x[t0 + m - order];
    __syncthreads();
    for (int r = threadIdx.x; r < tn; r += BLOCK) {
      const int i = t0 + r;
      const double *xc = xs + r + order;      // xc[-j] = x[i - j]
      double acc = 0.0;
      if (i < order) {
        for (int j = i + 1; j <= order; j++) acc += xc[-j] * cs[j];
        for (int j = 1; j <= i; j++) acc += xc[-j] * cs[j];
      } else {
        // body taps are wave-uniform: scalar loads, not LDS traffic (the kernel is LDS-bandwidth bound)
#pragma unroll 8
        for (int k = 0; k <= half_order; k++) acc += (xc[-k] + xc[-(order - k)]) * coeff[k];
        if ((order % 2) == 0) acc += xc[-(order / 2)] * coeff[order / 2];
      }
      y[i] = acc * out_gain;     // 1.0 for the FM pilot cut (exact); NbfmDecoder's -3 dB (Utility.h:307-312)
    }
  }
}

// ---------------------------------------------------------------------------
// K_out : DC block (HighPassFilterIir, Filter.cpp:243-250,301-311) on mono and
// L-R -- two serial lanes per stream -- then the output mux of
// FmDecode.cpp:194-220,242-283 on all lanes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fm_out(
    double *__restrict__ p0, double *__restrict__ p1, long long p_stride, BlockTab bt, int n_audio,
    double b0, double b1, double b2, double a1, double a2, int stereo, int pilot_shift,
    const int *__restrict__ stereo_blk, double *__restrict__ audio, long long audio_stride, StreamState *st) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  double *m = p0 + (long long)s * p_stride;
  double *d = p1 + (long long)s * p_stride;
  if (lane < (stereo ? 2 : 1)) {
    double *p = lane ? d : m;
    double x1 = lane ? st[s].dc_st_x1 : st[s].dc_mono_x1;
    double x2 = lane ? st[s].dc_st_x2 : st[s].dc_mono_x2;
    for (int i = 0; i < n_audio; i++) {
      const double x0 = p[i] - (a1 * x1 + a2 * x2);
      p[i] = b0 * x0 + b1 * x1 + b2 * x2;
      x2 = x1; x1 = x0;
    }
    if (lane) { st[s].dc_st_x1 = x1; st[s].dc_st_x2 = x2; }
    else { st[s].dc_mono_x1 = x1; st[s].dc_mono_x2 = x2; }
  }
  __syncthreads();
  double *out = audio + (long long)s * audio_stride;
  if (!stereo) {
    for (int i = lane; i < n_audio; i += 64) out[i] = m[i];
    return;
  }
  for (int b = 0; b < bt.nb; b++) {
    const int n = bt.au_len[b], off = bt.au_off[b];
    const int locked = stereo_blk[(long long)s * bt.nb + b];
    for (int i = lane; i < n; i += 64) {
      double l, r;
      if (locked) {
        if (pilot_shift) { l = r = d[off + i]; }
        else {
          const double mm = m[off + i];
          const double ss = 1.017 * d[off + i];
          l = mm + ss; r = mm - ss;
        }
      } else {
        if (pilot_shift) { l = r = 0.0; }
        else { l = r = m[off + i]; }
      }
      out[2 * (off + i)] = l;
      out[2 * (off + i) + 1] = r;
    }
  }
}

// ---------------------------------------------------------------------------
// K_am_tail : AmDecoder audio tail, serial per stream: DC block
// (AmDecode.cpp:194) -> AfSimpleAgc (AfSimpleAgc.cpp:36-56) -> de-emphasis
// (AmDecode.cpp:212-214, AM mode only).
// ---------------------------------------------------------------------------
__global__ void k_am_tail(const double *__restrict__ demod, long long d_stride, int n,
                          double hb0, double hb1, double hb2, double ha1, double ha2,
                          double af_init, double af_max, double af_ref, double af_rate,
                          double de_b0, double de_a1, int do_deemph,
                          double *__restrict__ audio, long long audio_stride, StreamState *st, int n_streams,
                          const int *__restrict__ skip = nullptr, int skip_stride = 0, int *__restrict__ ran = nullptr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_streams) return;
  if (skip && *reinterpret_cast<const int *>(reinterpret_cast<const char *>(skip) + (size_t)s * skip_stride)) return;   // the time-parallel form converged
  if (ran) *reinterpret_cast<int *>(reinterpret_cast<char *>(ran) + (size_t)s * skip_stride) = 1;
  const double *x = demod + (long long)s * d_stride;
  double *out = audio + (long long)s * audio_stride;
  double x1 = st[s].am_dc_x1, x2 = st[s].am_dc_x2, g = st[s].af_gain, e1 = st[s].am_de_x1;
  for (int i = 0; i < n; i++) {
    const double x0 = x[i] - (ha1 * x1 + ha2 * x2);
    const double v = hb0 * x0 + hb1 * x1 + hb2 * x2;
    x2 = x1; x1 = x0;
    const double xg = v * g;
    double o = xg * af_ref;
    const double z = 1.0 + (af_rate * (1.0 - (xg * xg)));
    g *= z;
    if (!isfinite(g)) g = af_init;
    else if (g > af_max) g = af_max;
    if (do_deemph) {
      const double w = o - de_a1 * e1;
      o = de_b0 * w;
      e1 = w;
    }
    out[i] = o;
  }
  st[s].am_dc_x1 = x1; st[s].am_dc_x2 = x2; st[s].af_gain = g; st[s].am_de_x1 = e1;
}

}  // namespace fmr
