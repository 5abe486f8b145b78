// Standalone check + micro-benchmark of the fused front-end kernel (kernels_fused.hpp) against the
// three-kernel path it replaces (k_ifr_decim2 -> k_ifr_poly4), on HBM-resident IQ.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/bench_fused tools/bench_fused.hip
//   /tmp/bench_fused [log2 N for the timing run, default 27]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../airspy-fmradion_amd/csrc/design.hpp"
#include "../airspy-fmradion_amd/csrc/kernels.hpp"
#include "../airspy-fmradion_amd/csrc/kernels_fused.hpp"
using namespace fmr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_fill(float2 *x, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned a = (unsigned)(i * 2654435761u) ^ seed, b = (unsigned)(i * 40503u + 12345u) ^ (seed * 7u);
    a ^= a >> 15; a *= 2246822519u; a ^= a >> 13; b ^= b >> 16; b *= 3266489917u; b ^= b >> 13;
    x[i] = make_float2((float)(a & 0xffffff) / 8388608.0f - 1.0f, (float)(b & 0xffffff) / 8388608.0f - 1.0f);
  }
}

struct Ctx {
  ResamplerDesign rs;
  ResamplerCounter rsc;
  int H_in, H_mid, H_if = 1;
  size_t max_in, max_mid, max_if;
  float2 *d_in_halo, *d_mid, *d_if_old, *d_if_new;
  float *d_hpA, *d_afrag;
  unsigned long long *d_dbg = nullptr; uint4 *d_afragA = nullptr, *d_afragB = nullptr; float hB_inv_scale = 1.f; float *d_hAf = nullptr, *d_hBf = nullptr;
  long long wg_key = -1;
  float *d_base = nullptr, *d_nrm = nullptr; float *d_dec = nullptr, *d_hBlast = nullptr, *d_stats = nullptr; StreamState *d_st = nullptr; FusedPart *d_part = nullptr; int *d_tab = nullptr;
  bool epi = false, lean = false; int h_off_dev400 = 0; int nb = 0; std::vector<int> h_off, h_len; int *d_wgblk = nullptr;
  int poly2_tile;
};

static void setup(Ctx &c, size_t max_in) {
  c.rs.design(10e6, 384e3, 140.0);
  const auto &rs = c.rs;
  c.H_in = rs.NA - 1 + rs.D; c.H_mid = rs.TB;
  c.max_in = max_in; c.max_mid = max_in / rs.D + 2; c.max_if = (size_t)((double)max_in * rs.L / rs.M) + 4;
  std::vector<float> fa(rs.hA.begin(), rs.hA.end()), fb(rs.hB.begin(), rs.hB.end());
  const int qa = 16;
  std::vector<float> hp((size_t)rs.D * qa, 0.f);
  for (int k = 0; k < rs.NA; k++) hp[(size_t)(k % rs.D) * qa + k / rs.D] = fa[k];
  CK(hipMalloc(&c.d_hpA, hp.size() * 4)); CK(hipMemcpy(c.d_hpA, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  for (int k = 0; k < rs.NA; k++)
    if (fa[k] != fa[rs.NA - 1 - k]) { printf("stage-A taps are not symmetric at %d\n", k); exit(1); }
  std::vector<int> phi(rs.LB), off(rs.LB);
  for (long long q = 0; q < rs.LB; q++) { phi[q] = (int)((q * rs.MB) % rs.LB); off[q] = (int)((q * rs.MB) / rs.LB); }
  c.poly2_tile = (int)(64 * rs.MB + off[rs.LB - 1] + rs.TB) + 64;
  using SH = Poly4Shape<48, 125, 210>;
  std::vector<float> af((size_t)SH::MT * SH::NK * 64, 0.f);
  for (int mt = 0; mt < SH::MT; mt++)
    for (int i = 0; i < SH::nks(mt); i++)
      for (int l = 0; l < 64; l++) {
        const int pp = 16 * mt + (l & 15), m = 4 * (SH::ks_lo(mt) + i) + (l >> 4), j = m - off[pp];
        if (j >= 0 && j < rs.TB) af[((size_t)mt * SH::NK + i) * 64 + l] = fb[(size_t)phi[pp] * rs.TB + j];
      }
  CK(hipMalloc(&c.d_afrag, af.size() * 4)); CK(hipMemcpy(c.d_afrag, af.data(), af.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&c.d_in_halo, c.H_in * 8)); CK(hipMemset(c.d_in_halo, 0, c.H_in * 8));
  CK(hipMalloc(&c.d_mid, (c.H_mid + c.max_mid) * 8)); CK(hipMemset(c.d_mid, 0, (c.H_mid + c.max_mid) * 8));
  CK(hipMalloc(&c.d_if_old, (c.H_if + c.max_if) * 8)); CK(hipMalloc(&c.d_if_new, (c.H_if + c.max_if) * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 125, 210>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  constexpr int kL = FusedShape<kFusedD, kFusedNA>::LDS_BYTES;
#define SETATTR(P, A) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_fused<kFusedD, kFusedNA, P, A>), hipFuncAttributeMaxDynamicSharedMemorySize, kL))
  SETATTR(0, 0); SETATTR(1, 0); SETATTR(0, 512); SETATTR(1, 512); SETATTR(0, 1024); SETATTR(1, 1024); SETATTR(0, 64); SETATTR(1, 64); SETATTR(0, 128); SETATTR(1, 128); SETATTR(0, 256); SETATTR(1, 256); SETATTR(0, 448); SETATTR(1, 448); SETATTR(0, 46); SETATTR(1, 46); SETATTR(0, 54); SETATTR(1, 54); SETATTR(0, 39); SETATTR(1, 39); SETATTR(0, 1); SETATTR(1, 1); SETATTR(0, 2); SETATTR(1, 2); SETATTR(0, 3); SETATTR(1, 3);
  SETATTR(0, 4); SETATTR(1, 4); SETATTR(0, 7); SETATTR(1, 7); SETATTR(0, 5); SETATTR(1, 5); SETATTR(0, 6); SETATTR(1, 6);
  SETATTR(0, 14); SETATTR(1, 14); SETATTR(0, 22); SETATTR(1, 22); SETATTR(0, 32); SETATTR(1, 32); SETATTR(0, 36); SETATTR(1, 36); SETATTR(0, 38); SETATTR(1, 38); SETATTR(0, 37); SETATTR(1, 37); SETATTR(0, 35); SETATTR(1, 35);
  CK(hipMalloc(&c.d_hAf, fa.size() * 4)); CK(hipMemcpy(c.d_hAf, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&c.d_hBf, fb.size() * 4)); CK(hipMemcpy(c.d_hBf, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
  { std::vector<unsigned short> fr(2 * 8 * 2 * 64 * 8);
    fused_make_afragA<kFusedD, kFusedNA>(fa.data(), fr.data());
    CK(hipMalloc(&c.d_afragA, fr.size() * 2)); CK(hipMemcpy(c.d_afragA, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
    std::vector<unsigned short> frb((size_t)3 * 9 * 2 * 64 * 8);
    c.hB_inv_scale = fused_make_afragB(fb.data(), frb.data());
    CK(hipMalloc(&c.d_afragB, frb.size() * 2)); CK(hipMemcpy(c.d_afragB, frb.data(), frb.size() * 2, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&c.d_dbg, 32 * 8)); CK(hipMemset(c.d_dbg, 0, 32 * 8));
  CK(hipMalloc(&c.d_wgblk, 1024 * 4));
  CK(hipMalloc(&c.d_base, c.max_if * 4)); CK(hipMalloc(&c.d_nrm, c.max_if * 8 + 64)); CK(hipMalloc(&c.d_dec, c.max_if * 4)); CK(hipMalloc(&c.d_st, sizeof(StreamState)));
  CK(hipMemset(c.d_st, 0, sizeof(StreamState)));
  CK(hipMalloc(&c.d_part, (c.max_if / 128 + 16) * sizeof(FusedPart))); CK(hipMalloc(&c.d_tab, 2 * 4096 * 4)); CK(hipMalloc(&c.d_stats, 3 * 4096 * 4));
  { std::vector<float> row(fb.begin() + (size_t)phi[47] * rs.TB, fb.begin() + (size_t)(phi[47] + 1) * rs.TB);
    CK(hipMalloc(&c.d_hBlast, row.size() * 4)); CK(hipMemcpy(c.d_hBlast, row.data(), row.size() * 4, hipMemcpyHostToDevice)); }
}

struct CallGeom { long long mA_prev, kB_prev, n_prev, N_in; int count_mid; long long N_if; };

static CallGeom advance(Ctx &c, long long N_in) {
  CallGeom g{c.rsc.mA, c.rsc.kB, c.rsc.n_in, N_in, 0, 0};
  g.N_if = c.rsc.advance(c.rs, N_in);
  g.count_mid = (int)(c.rsc.mA - g.mA_prev);
  return g;
}

static void launch_old(Ctx &c, const CallGeom &g, const float2 *d_iq, float2 *ifbuf, bool with_decim = true, bool with_poly = true) {
  const auto &rs = c.rs;
  const long long top0 = (long long)rs.D * g.mA_prev + rs.ca() - g.n_prev;
  constexpr int BL2 = 128, T2 = 256;
  int s_pad = T2 + 16;
  while ((s_pad & 15) != 2) s_pad++;
  const size_t lds2 = sizeof(float2) * ((size_t)rs.D * s_pad + 2);
  const unsigned magic = (unsigned)((1u << 24) / (unsigned)rs.D + 1);
  if (with_decim && g.count_mid > 0)
    hipLaunchKernelGGL((k_ifr_decim2<BL2, 16, 0, false>), dim3((g.count_mid + T2 - 1) / T2, 1), dim3(BL2), lds2, 0, d_iq, (long long)c.max_in,
                       g.N_in, c.d_in_halo, c.H_in, c.d_hpA, rs.D, rs.ca(), top0 - rs.ca(), g.count_mid, c.d_mid,
                       (long long)(c.H_mid + c.max_mid), c.H_mid, 0u, 0, s_pad, magic);
  if (with_poly && g.N_if > 0) {
    const long long P_first = g.kB_prev / rs.LB, P_last = (g.kB_prev + g.N_if - 1) / rs.LB;
    const int tiles = (int)((P_last - P_first) / 64 + 1);
    hipLaunchKernelGGL((k_ifr_poly4<48, 125, 210>), dim3(std::min(tiles, 512), 1), dim3(256),
                       sizeof(float2) * (size_t)(((c.poly2_tile + 127) / 128) * 128 + 4 * 8 * 48), 0, c.d_mid,
                       (long long)(c.H_mid + c.max_mid), g.mA_prev - c.H_mid, c.H_mid + g.count_mid, c.d_afrag, g.kB_prev,
                       (int)g.N_if, ifbuf, (long long)(c.H_if + c.max_if), c.H_if, c.poly2_tile, tiles);
  }
}

template <int ABL = 0>
static void launch_new(Ctx &c, const CallGeom &g, const float2 *d_iq, float2 *ifbuf, int n_wg) {
  const auto &rs = c.rs;
  constexpr int D = kFusedD, NA = kFusedNA;
  FusedArgs a{};
  a.iq = d_iq; a.iq_stride = (long long)c.max_in; a.n_valid = g.N_in;
  a.in_halo = c.d_in_halo; a.H_in = c.H_in; a.afragA = c.d_afragA; a.hA = c.d_hAf; a.hB = c.d_hBf;
  const long long n0 = (long long)rs.D * g.mA_prev - g.n_prev;
  const long long lo0 = n0 + rs.ca() - (NA - 1);
  const int par = (int)(((lo0 % 2) + 2) % 2);
  a.nbase = lo0 - par;
  const long long P_first = g.kB_prev / 48, P_last = (g.kB_prev + g.N_if - 1) / 48;
  constexpr int kME = FusedShape<kFusedD, kFusedNA>::ME, kEPT = FusedShape<kFusedD, kFusedNA>::EPT;
  const long long T_first = P_first / 8, E_ref = kEPT * T_first - 1;
  a.j_ref = (int)(kME * E_ref + 104 - g.mA_prev);
  a.pos_ref = (int)((((kME * E_ref + 208) % 3000) + 3000) % 3000);
  a.t3_ref = (int)(T_first % 3);
  a.kb_ref = (int)(384 * T_first - g.kB_prev);
  a.count_mid = g.count_mid;
  a.mid = c.d_mid; a.mid_stride = (long long)(c.H_mid + c.max_mid); a.H_mid = c.H_mid;
  a.afragB = c.d_afragB; a.hB_inv_scale = c.hB_inv_scale; a.n_if = (int)g.N_if;
  a.out = ifbuf; a.out_stride = (long long)(c.H_if + c.max_if); a.out_off = c.H_if;
  a.dbg = c.d_dbg;
  if (c.epi) {
    a.base = c.d_base; a.base_stride = (long long)c.max_if; a.base_off = 0; a.dec = c.lean ? nullptr : c.d_dec; a.dec_stride = (long long)c.max_if;
    if (c.lean) { a.out = nullptr; a.nrm = c.d_nrm; a.nrm_stride = (long long)c.max_if; a.nrm_off = 0; }      // the product configuration: MPX + |x|^2, no IF samples
    if (c.lean && c.nb > 400) a.part_from = c.h_off_dev400;     // as in the chain: no debug copy, block sums only where k_stats reads them
    a.nf = (float)((75000.0 / 384000.0) * 2.0 * M_PI); a.bound = (float)(1.0 / ((75000.0 / 384000.0) * 2.0));
    a.st = c.d_st; a.hB_last = c.d_hBlast; a.part = c.d_part; a.if_off = c.d_tab; a.if_len = c.d_tab + 4096; a.nb = c.nb;
  }
  a.n_tiles = (int)(P_last / 8 - T_first + 1);
  a.tiles_per_wg = (a.n_tiles + n_wg - 1) / n_wg;
  const int grid = (a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  constexpr size_t kLds = FusedShape<D, NA>::LDS_BYTES;
  const long long wkey = ((long long)grid << 40) ^ ((long long)a.tiles_per_wg << 20) ^ (long long)(a.kb_ref + 4096) ^ ((long long)g.N_if << 8);
  if (c.epi && wkey == c.wg_key) a.wg_blk0 = c.d_wgblk;      // same geometry as the last launch (timing loops): no host round trip
  else if (c.epi) {
    c.wg_key = wkey;
    // block of the first IF sample of every workgroup (host side of the block walk)
    std::vector<int> tab(2 * 4096), wb(grid);
    CK(hipMemcpy(tab.data(), c.d_tab, tab.size() * 4, hipMemcpyDeviceToHost));
    int b = 0;
    for (int w = 0; w < grid; w++) {
      const long long kf = std::max<long long>(0, (long long)a.kb_ref + 384ll * w * a.tiles_per_wg);
      while (b < c.nb && tab[b] + tab[4096 + b] <= kf) b++;
      wb[w] = b;
    }
    CK(hipMemcpy(c.d_wgblk, wb.data(), grid * 4, hipMemcpyHostToDevice));
    a.wg_blk0 = c.d_wgblk;
  }
  if (par) hipLaunchKernelGGL((k_ifr_fused<D, NA, 1, ABL>), dim3(grid, 1), dim3(FUSED_THREADS), kLds, 0, a);
  else hipLaunchKernelGGL((k_ifr_fused<D, NA, 0, ABL>), dim3(grid, 1), dim3(FUSED_THREADS), kLds, 0, a);
}

static void halo_updates(Ctx &c, const CallGeom &g, const float2 *d_iq) {
  hipLaunchKernelGGL((k_update_in_halo<256, 0>), dim3(1, 1), dim3(256), 0, 0, c.d_in_halo, c.H_in, d_iq, (long long)c.max_in, g.N_in);
  HaloTable ht{};
  ht.d[0] = HaloDesc{(unsigned *)c.d_mid, 2 * (c.H_mid + (long long)c.max_mid), 2 * c.H_mid, 2 * g.count_mid};
  ht.n = 1;
  hipLaunchKernelGGL(k_shift_halo<256>, dim3(1, 1), dim3(256), 0, 0, ht);
}

static double compare(const char *what, const float2 *d_a, const float2 *d_b, size_t n) {
  std::vector<float2> a(n), b(n);
  CK(hipMemcpy(a.data(), d_a, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_b, n * 8, hipMemcpyDeviceToHost));
  double se = 0, sr = 0, mx = 0; size_t bad = 0, first_bad = (size_t)-1;
  for (size_t i = 0; i < n; i++) {
    const double dx = (double)a[i].x - b[i].x, dy = (double)a[i].y - b[i].y;
    const double e = dx * dx + dy * dy;
    se += e; sr += (double)a[i].x * a[i].x + (double)a[i].y * a[i].y;
    if (!(e == e)) { bad++; if (first_bad == (size_t)-1) first_bad = i; }
    if (e > mx) mx = e;
  }
  const double rel = std::sqrt(se / (sr > 0 ? sr : 1));
  printf("%-40s n=%zu rel_rms=%.3e max_abs=%.3e nan=%zu first_nan=%zd ref_rms=%.4f\n", what, n, rel, std::sqrt(mx), bad, (ssize_t)first_bad, std::sqrt(sr / n));
  return rel;
}

template <class F>
static float time_it(const char *name, double bytes, F &&launch, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) launch();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  printf("%-44s %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12 * 100);
  return ms;
}

int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 27;
  const size_t N = (size_t)1 << lg;
  Ctx c;
  setup(c, N);
  float2 *d_iq;
  CK(hipMalloc(&d_iq, N * 8));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d_iq, N, 12345u);
  CK(hipDeviceSynchronize());
  int fails = 0;
  // ---- correctness: call 1 (cold: zero halos) of odd length, then call 2 with carried state
  const long long N1 = 1000003, N2 = (long long)std::min<size_t>(N - N1 - 16, (size_t)6000011);
  {
    CallGeom g1 = advance(c, N1);
    CK(hipMemset(c.d_if_old, 0, (c.H_if + c.max_if) * 8)); CK(hipMemset(c.d_if_new, 0xff, (c.H_if + c.max_if) * 8));
    launch_old(c, g1, d_iq, c.d_if_old);
    CK(hipDeviceSynchronize());
    // keep the old path's mid tail for the halo check
    std::vector<float2> mid_old(c.H_mid + g1.count_mid);
    CK(hipMemcpy(mid_old.data(), c.d_mid, mid_old.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(c.d_mid + c.H_mid, 0, (size_t)g1.count_mid * 8));     // the fused kernel writes only the tail
    launch_new(c, g1, d_iq, c.d_if_new, 7);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    if (compare("call 1 (cold, 1000003 samples, 7 WGs)", c.d_if_old + c.H_if, c.d_if_new + c.H_if, (size_t)g1.N_if) > 2e-6) fails++;
    std::vector<float2> mid_new(c.H_mid + g1.count_mid);
    CK(hipMemcpy(mid_new.data(), c.d_mid, mid_new.size() * 8, hipMemcpyDeviceToHost));
    double se = 0, sr = 0;
    for (int i = g1.count_mid - c.H_mid; i < g1.count_mid; i++) {
      const float2 p = mid_old[c.H_mid + i], q = mid_new[c.H_mid + i];
      se += (double)(p.x - q.x) * (p.x - q.x) + (double)(p.y - q.y) * (p.y - q.y); sr += (double)p.x * p.x + (double)p.y * p.y;
    }
    printf("mid tail written by the fused kernel: rel_rms=%.3e\n", std::sqrt(se / sr));
    if (!(std::sqrt(se / sr) < 2e-6)) fails++;
    // carry the state with the OLD path's mid (bit-equal histories for both paths of call 2)
    CK(hipMemcpy(c.d_mid, mid_old.data(), mid_old.size() * 8, hipMemcpyHostToDevice));
    halo_updates(c, g1, d_iq);
    CK(hipDeviceSynchronize());
    CallGeom g2 = advance(c, N2);
    std::vector<float2> halo_in(c.H_in), halo_mid(c.H_mid);
    CK(hipMemcpy(halo_in.data(), c.d_in_halo, c.H_in * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(halo_mid.data(), c.d_mid, c.H_mid * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(c.d_if_old, 0, (c.H_if + c.max_if) * 8)); CK(hipMemset(c.d_if_new, 0xff, (c.H_if + c.max_if) * 8));
    launch_old(c, g2, d_iq + N1, c.d_if_old);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(c.d_mid, halo_mid.data(), c.H_mid * 8, hipMemcpyHostToDevice));
    launch_new(c, g2, d_iq + N1, c.d_if_new, 256);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    printf("call 2: mA_prev=%lld kB_prev=%lld n_prev=%lld N_in=%lld count_mid=%d N_if=%lld\n", g2.mA_prev, g2.kB_prev, g2.n_prev, g2.N_in, g2.count_mid, g2.N_if);
    if (compare("call 2 (carried state, odd offset, 256 WGs)", c.d_if_old + c.H_if, c.d_if_new + c.H_if, (size_t)g2.N_if) > 2e-6) fails++;
    CK(hipMemcpy(c.d_mid, halo_mid.data(), c.H_mid * 8, hipMemcpyHostToDevice));
    CK(hipMemset(c.d_if_new, 0xff, (c.H_if + c.max_if) * 8));
    launch_new(c, g2, d_iq + N1, c.d_if_new, 3);
    CK(hipDeviceSynchronize());
    if (compare("call 2 again with 3 WGs", c.d_if_old + c.H_if, c.d_if_new + c.H_if, (size_t)g2.N_if) > 2e-6) fails++;
  }
  // ---- discriminator epilogue + block statistics: a third call (carried disc_save), blocks of 65536 input samples
  {
    halo_updates(c, CallGeom{0, 0, 0, N2, (int)(c.rsc.mA - 99993), 0}, d_iq + N1);   // state after call 2 (old path's mid is in d_mid)
    CK(hipDeviceSynchronize());
    const long long N3 = 40 * 65536 + 12345;
    // per-block IF lengths from the count law
    ResamplerCounter rc2 = c.rsc;
    c.h_off.clear(); c.h_len.clear();
    long long acc_if = 0, left = N3;
    while (left > 0) { const long long bl = std::min<long long>(65536, left); const long long k = rc2.advance(c.rs, bl); c.h_off.push_back((int)acc_if); c.h_len.push_back((int)k); acc_if += k; left -= bl; }
    c.nb = (int)c.h_off.size(); c.wg_key = -1;
    CallGeom g3 = advance(c, N3);
    std::vector<int> tab(2 * 4096, 0);
    for (int b = 0; b < c.nb; b++) { tab[b] = c.h_off[b]; tab[4096 + b] = c.h_len[b]; }
    CK(hipMemcpy(c.d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    StreamState st{}; st.disc_save = 0.4321f;
    CK(hipMemcpy(c.d_st, &st, sizeof st, hipMemcpyHostToDevice));
    c.epi = true;
    CK(hipMemset(c.d_if_new, 0xff, (c.H_if + c.max_if) * 8));
    launch_new(c, g3, d_iq + N1 + N2, c.d_if_new, 5);
    hipLaunchKernelGGL(k_fused_blk_reduce, dim3((c.nb + 63) / 64, 1), dim3(64), 0, 0, c.d_part, (int)((g3.kB_prev + g3.N_if - 1) / 48 / 8 - (g3.kB_prev / 48) / 8 + 1),
                       (int)(384 * ((g3.kB_prev / 48) / 8) - g3.kB_prev), c.d_tab, c.d_tab + 4096, c.nb, c.d_stats, c.d_stats + 4096, c.d_stats + 8192);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    c.epi = false;
    std::vector<float2> xif(g3.N_if); std::vector<float> bse(g3.N_if); std::vector<float> dc(g3.N_if), stats(3 * 4096);
    CK(hipMemcpy(xif.data(), c.d_if_new + c.H_if, g3.N_if * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(bse.data(), c.d_base, g3.N_if * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dc.data(), c.d_dec, g3.N_if * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(stats.data(), c.d_stats, stats.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&st, c.d_st, sizeof st, hipMemcpyDeviceToHost));
    const float nf = (float)((75000.0 / 384000.0) * 2.0 * M_PI), bound = (float)(1.0 / ((75000.0 / 384000.0) * 2.0));
    double se = 0, sr = 0; float prev = 0.4321f; size_t nbad = 0;
    std::vector<float> dref(g3.N_if);
    for (long long k = 0; k < g3.N_if; k++) {
      const float ph = atan2f(xif[k].y, xif[k].x) / nf;
      float d = ph - prev; if (d > bound) d -= 2 * bound; if (d < -bound) d += 2 * bound;
      prev = ph; dref[k] = d;
      double e = fabs((double)d - bse[k]); if (e > 2 * bound - 1e-3) e = fabs(e - 2 * bound);   // a wrap decided the other way by one ulp
      se += e * e; sr += (double)d * d;
      if (e > 1e-5 || (float)bse[k] != dc[k]) { if (nbad < 5) printf("  disc mismatch at %lld: ref %.7f got %.7f dec %.7f\n", k, d, bse[k], dc[k]); nbad++; }
    }
    printf("discriminator epilogue: n=%lld rms_err=%.3e (signal rms %.3f) mismatches=%zu  disc_save_next=%.6f (ref %.6f) valid=%d\n", g3.N_if,
           std::sqrt(se / g3.N_if), std::sqrt(sr / g3.N_if), nbad, st.disc_save_next, prev, st.disc_save_valid);
    if (nbad || st.disc_save_valid != 1 || fabsf(st.disc_save_next - prev) > 1e-6f) fails++;
    double worst = 0;
    for (int b = 0; b < c.nb; b++) {
      if (!c.h_len[b]) continue;
      double sd = 0, sq = 0, sx = 0;
      for (int i = 0; i < c.h_len[b]; i++) { const long long k = c.h_off[b] + i; sd += dref[k]; sq += (double)dref[k] * dref[k]; sx += (double)xif[k].x * xif[k].x + (double)xif[k].y * xif[k].y; }
      const double m = sd / c.h_len[b], r = std::sqrt(sq / c.h_len[b]), x = std::sqrt(sx / c.h_len[b]);
      worst = std::max(worst, std::max(fabs(stats[b] - m), std::max(fabs(stats[4096 + b] - r) / r, fabs(stats[8192 + b] - x) / x)));
    }
    printf("block statistics (%d blocks): worst deviation %.3e\n", c.nb, worst);
    if (!(worst < 1e-4)) fails++;
  }
  printf(fails ? "CORRECTNESS: %d FAILURES\n" : "CORRECTNESS: ok\n", fails);
  // ---- timing on the full buffer (steady-state geometry: a call in the middle of a stream)
  c.rsc.reset();
  advance(c, 12345678);
  CallGeom g = advance(c, (long long)N);
  const double bytes = 8.0 * N;
  time_it("old: ifr_decim2 (stage A)", bytes, [&] { launch_old(c, g, d_iq, c.d_if_old, true, false); });
  time_it("old: ifr_poly4 (stage B)", bytes, [&] { launch_old(c, g, d_iq, c.d_if_old, false, true); });
  time_it("old: stage A + stage B", bytes, [&] { launch_old(c, g, d_iq, c.d_if_old); });
  {
    // block table of the timing geometry (65536-sample blocks) for the runs with the discriminator epilogue
    ResamplerCounter rc2; rc2.advance(c.rs, 12345678);
    std::vector<int> tab(2 * 4096, 0);
    long long acc_if = 0, left = (long long)N; int nb = 0;
    while (left > 0 && nb < 4096) { const long long bl = std::min<long long>(65536, left); const long long k = rc2.advance(c.rs, bl); tab[nb] = (int)acc_if; tab[4096 + nb] = (int)k; acc_if += k; left -= bl; nb++; }
    c.nb = nb; c.wg_key = -1; c.h_off_dev400 = nb > 400 ? tab[nb - 400] : 0;
    CK(hipMemcpy(c.d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }
  c.epi = true;
  c.lean = true;
  time_it("fused A+B+discriminator AS IN THE CHAIN (lean)", bytes, [&] { launch_new(c, g, d_iq, c.d_if_new, 256); });
  time_it("lean, no global stores", bytes, [&] { launch_new<128>(c, g, d_iq, c.d_if_new, 256); });
  time_it("lean, no atan2", bytes, [&] { launch_new<64>(c, g, d_iq, c.d_if_new, 256); });
  time_it("lean, MPX store only (one 8-byte store per lane)", bytes, [&] { launch_new<512>(c, g, d_iq, c.d_if_new, 256); });
  time_it("lean, one 16-byte store per lane (MPX + |x|^2 together)", bytes, [&] { launch_new<1024>(c, g, d_iq, c.d_if_new, 256); });
  for (int i = 0; i < 3; i++) time_it("fused A+B+discriminator AS IN THE CHAIN (lean)", bytes, [&] { launch_new(c, g, d_iq, c.d_if_new, 256); });
  c.lean = false;
  time_it("fused A+B+discriminator, 256 workgroups", bytes, [&] { launch_new(c, g, d_iq, c.d_if_new, 256); });
  time_it("fused A+B+discriminator, 248 workgroups", bytes, [&] { launch_new(c, g, d_iq, c.d_if_new, 248); });
  time_it("epilogue ablation: no atan2", bytes, [&] { launch_new<64>(c, g, d_iq, c.d_if_new, 256); });
  time_it("epilogue ablation: no global stores", bytes, [&] { launch_new<128>(c, g, d_iq, c.d_if_new, 256); });
  time_it("epilogue ablation: no block sums", bytes, [&] { launch_new<256>(c, g, d_iq, c.d_if_new, 256); });
  time_it("epilogue ablation: none of the three", bytes, [&] { launch_new<448>(c, g, d_iq, c.d_if_new, 256); });
  c.epi = false;
  for (int nwg : {256, 512, 248})
    { char nm[64]; snprintf(nm, sizeof nm, "fused A+B, %d workgroups", nwg); time_it(nm, bytes, [&] { launch_new(c, g, d_iq, c.d_if_new, nwg); }); }
  time_it("ablation: no stage-A math", bytes, [&] { launch_new<1>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: no stage-B MFMA", bytes, [&] { launch_new<2>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: no A math, no B MFMA (DMA only)", bytes, [&] { launch_new<3>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: no DMA (A + B on stale LDS)", bytes, [&] { launch_new<4>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: nothing (barriers + epilogue)", bytes, [&] { launch_new<7>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: B only, no DMA", bytes, [&] { launch_new<5>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: A only, no DMA", bytes, [&] { launch_new<6>(c, g, d_iq, c.d_if_new, 256); });
  auto dump = [&](const char *what) {
    unsigned long long h[32];
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, c.d_dbg, sizeof h, hipMemcpyDeviceToHost));
    printf("%-28s cycles busy/total per wave:", what);
    for (int w = 0; w < FUSED_THREADS / 64; w++) printf("  w%d %llu/%llu", w, h[2 * w], h[2 * w + 1]);
    printf("\n");
  };
  time_it("ablation: A FMAs only (no LDS reads), no B, no DMA", bytes, [&] { launch_new<14>(c, g, d_iq, c.d_if_new, 256); });
  time_it("ablation: A LDS reads only (no FMAs), no B, no DMA", bytes, [&] { launch_new<22>(c, g, d_iq, c.d_if_new, 256); });
  c.epi = true; launch_new<32>(c, g, d_iq, c.d_if_new, 256); dump("product with epilogue"); c.epi = false;
  launch_new<32>(c, g, d_iq, c.d_if_new, 256); dump("product");
  launch_new<36>(c, g, d_iq, c.d_if_new, 256); dump("no DMA");
  launch_new<38>(c, g, d_iq, c.d_if_new, 256); dump("A only, no DMA");
  launch_new<37>(c, g, d_iq, c.d_if_new, 256); dump("B only, no DMA");
  launch_new<35>(c, g, d_iq, c.d_if_new, 256); dump("DMA only");
  launch_new<46>(c, g, d_iq, c.d_if_new, 256); dump("A FMAs only");
  launch_new<54>(c, g, d_iq, c.d_if_new, 256); dump("A LDS reads only");
  launch_new<39>(c, g, d_iq, c.d_if_new, 256); dump("nothing");
  {
    // fixed cost of a launch vs cost per epoch: the same geometry at half and a quarter of the length
    for (int sh = 1; sh <= 2; sh++) {
      c.rsc.reset(); advance(c, 12345678);
      CallGeom gh = advance(c, (long long)(N >> sh));
      char nm[96];
      snprintf(nm, sizeof nm, "1/%d length: fused A+B", 1 << sh);
      time_it(nm, bytes / (1 << sh), [&] { launch_new(c, gh, d_iq, c.d_if_new, 256); });
      snprintf(nm, sizeof nm, "1/%d length: nothing (barriers)", 1 << sh);
      time_it(nm, bytes / (1 << sh), [&] { launch_new<7>(c, gh, d_iq, c.d_if_new, 256); });
      snprintf(nm, sizeof nm, "1/%d length: DMA only", 1 << sh);
      time_it(nm, bytes / (1 << sh), [&] { launch_new<3>(c, gh, d_iq, c.d_if_new, 256); });
    }
    c.rsc.reset(); advance(c, 12345678); advance(c, (long long)N);
  }
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  launch_new(c, g, d_iq, c.d_if_new, 256);
  CK(hipDeviceSynchronize());
  if (compare("timing geometry: fused vs old", c.d_if_old + c.H_if, c.d_if_new + c.H_if, (size_t)g.N_if) > 2e-6) fails++;
  // ---- does the placement of the input buffer matter?  The same kernels over six more allocations of the input (all
  // kept until the end), and over the first one again
  if (argc > 2) {
    std::vector<float2 *> bufs;
    for (int a = 0; a < 6; a++) {
      float2 *p = nullptr;
      if (hipMalloc(&p, N * 8 + (size_t)a * 4096 * 17) != hipSuccess) break;
      CK(hipMemcpy(p, d_iq, N * 8, hipMemcpyDeviceToDevice));
      bufs.push_back(p);
    }
    bufs.push_back(d_iq);
    for (size_t a = 0; a < bufs.size(); a++) {
      char nm[96];
      snprintf(nm, sizeof nm, "allocation %zu (%p): DMA only", a, (void *)bufs[a]);
      time_it(nm, bytes, [&] { launch_new<3>(c, g, bufs[a], c.d_if_new, 256); });
      snprintf(nm, sizeof nm, "allocation %zu: product", a);
      time_it(nm, bytes, [&] { launch_new(c, g, bufs[a], c.d_if_new, 256); });
    }
  }
  return fails ? 1 : 0;
}
