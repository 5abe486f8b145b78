/*
 * fmradion_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See fmradion_oracle.h for the parity status ("parity unpinned" for the two
 * resamplers; the rest pinned by SURVEY.md section 8c known answers).
 *
 * Build: see oracle/Makefile (-O3 -ftree-vectorize, no -ffast-math, as the
 * reference's CMakeLists.txt:193-199; -ffp-contract=off so that no FMA is
 * formed, matching the reference's x86-64 baseline build).
 */
#include "fmradion_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================== */
/* Resampler specification (ours).  DESIGN.md "Resampler specification".     */
/* Stands in for r8b::CDSPResampler / CDSPResampler24 which the reference    */
/* uses at IfResampler.cpp:26-29,56-59 and AudioResampler.cpp:28-29,48.      */
/* ======================================================================== */

static double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  double q = x * x * 0.25;
  for (int k = 1; k < 1000; k++) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-21 * sum) break;
  }
  return sum;
}

static double sinc_pi(double x) { /* sin(pi x)/(pi x) */
  if (fabs(x) < 1e-12) return 1.0;
  return sin(M_PI * x) / (M_PI * x);
}

static long long gcd_ll(long long a, long long b) {
  while (b) { long long t = a % b; a = b; b = t; }
  return a < 0 ? -a : a;
}

/* ---- equiripple stage A ("Resampler specification"): Parks-McClellan exchange for a type-I (odd length,
 * symmetric) multiband filter.  Ours, like the rest of the resampler specification (r8brain is absent); the procedure
 * is the published one (McClellan, Parks, Rabiner 1973) on the customary grid: band b contributes
 * round((hi - lo) / delf) points from lo in steps of delf = 0.5 / (16 (M + 1)), the last one moved onto hi;
 * barycentric Lagrange interpolation in x = cos(2 pi f); stop when the extremal errors agree to 1e-4 relative.
 * edges: 2 nb band edges in cycles per sample; returns the number of exchange steps, -1 if it did not settle. */
typedef struct { int r, gs; double *grid, *D, *W, *E, *x, *y, *ad; int *ext; } pm_t;

static void pm_params(pm_t *p) {
  const int r = p->r;
  for (int i = 0; i <= r; i++) p->x[i] = cos(2.0 * M_PI * p->grid[p->ext[i]]);
  const int ld = (r - 1) / 15 + 1;           /* strided products keep the barycentric weights in range */
  for (int i = 0; i <= r; i++) {
    double denom = 1.0;
    const double xi = p->x[i];
    for (int j = 0; j < ld; j++)
      for (int k = j; k <= r; k += ld)
        if (k != i) denom *= 2.0 * (xi - p->x[k]);
    if (fabs(denom) < 1e-5) denom = 1e-5;
    p->ad[i] = 1.0 / denom;
  }
  double numer = 0.0, denom = 0.0, sign = 1.0;
  for (int i = 0; i <= r; i++) {
    numer += p->ad[i] * p->D[p->ext[i]];
    denom += sign * p->ad[i] / p->W[p->ext[i]];
    sign = -sign;
  }
  const double delta = numer / denom;
  sign = 1.0;
  for (int i = 0; i <= r; i++) {
    p->y[i] = p->D[p->ext[i]] - sign * delta / p->W[p->ext[i]];
    sign = -sign;
  }
}

static double pm_response(const pm_t *p, double freq) {
  double numer = 0.0, denom = 0.0;
  const double xc = cos(2.0 * M_PI * freq);
  for (int i = 0; i <= p->r; i++) {
    double c = xc - p->x[i];
    if (fabs(c) < 1.0e-7) return p->y[i];
    c = p->ad[i] / c;
    denom += c;
    numer += c * p->y[i];
  }
  return numer / denom;
}

static void pm_search(pm_t *p, int *found) {
  const int r = p->r, gs = p->gs;
  const double *E = p->E;
  int k = 0;
  if ((E[0] > 0.0 && E[0] > E[1]) || (E[0] < 0.0 && E[0] < E[1])) found[k++] = 0;
  for (int i = 1; i < gs - 1; i++)
    if ((E[i] >= E[i - 1] && E[i] > E[i + 1] && E[i] > 0.0) || (E[i] <= E[i - 1] && E[i] < E[i + 1] && E[i] < 0.0)) found[k++] = i;
  { const int j = gs - 1;
    if ((E[j] > 0.0 && E[j] > E[j - 1]) || (E[j] < 0.0 && E[j] < E[j - 1])) found[k++] = j; }
  int extra = k - (r + 1);
  while (extra > 0) {
    int up = E[found[0]] > 0.0, l = 0, alt = 1;
    for (int j = 1; j < k; j++) {
      if (fabs(E[found[j]]) < fabs(E[found[l]])) l = j;
      if (up && E[found[j]] < 0.0) up = 0;
      else if (!up && E[found[j]] > 0.0) up = 1;
      else { alt = 0; break; }             /* two neighbours of one sign: drop the smaller seen so far */
    }
    if (alt && extra == 1) l = (fabs(E[found[k - 1]]) < fabs(E[found[0]])) ? k - 1 : 0;
    for (int j = l; j < k - 1; j++) found[j] = found[j + 1];
    k--; extra--;
  }
  for (int i = 0; i <= r && i < k; i++) p->ext[i] = found[i];
}

static int pm_design(int N, int nb, const double *edges, const double *des, const double *wt, double *h) {
  const int M = (N - 1) / 2, r = M + 1, dens = 16;
  const double delf = 0.5 / (dens * r);
  pm_t p; p.r = r;
  int gs = 0;
  for (int b = 0; b < nb; b++) gs += (int)((edges[2 * b + 1] - edges[2 * b]) / delf + 0.5);
  p.gs = gs;
  if (gs < r + 2) return -1;
  p.grid = (double *)malloc(sizeof(double) * gs * 4); p.D = p.grid + gs; p.W = p.D + gs; p.E = p.W + gs;
  p.x = (double *)malloc(sizeof(double) * (r + 1) * 3); p.y = p.x + (r + 1); p.ad = p.y + (r + 1);
  p.ext = (int *)malloc(sizeof(int) * (r + 1));
  int *found = (int *)malloc(sizeof(int) * 2 * gs);
  { int j = 0;
    for (int b = 0; b < nb; b++) {
      double lowf = edges[2 * b];
      const int k = (int)((edges[2 * b + 1] - edges[2 * b]) / delf + 0.5);
      for (int i = 0; i < k; i++) { p.D[j] = des[b]; p.W[j] = wt[b]; p.grid[j] = lowf; lowf += delf; j++; }
      p.grid[j - 1] = edges[2 * b + 1];
    } }
  for (int i = 0; i <= r; i++) p.ext[i] = i * (gs - 1) / r;
  int it, ok = 0;
  for (it = 0; it < 100; it++) {
    pm_params(&p);
    for (int i = 0; i < gs; i++) p.E[i] = p.W[i] * (p.D[i] - pm_response(&p, p.grid[i]));
    pm_search(&p, found);
    double mn = fabs(p.E[p.ext[0]]), mx = mn;
    for (int i = 1; i <= r; i++) { const double c = fabs(p.E[p.ext[i]]); if (c < mn) mn = c; if (c > mx) mx = c; }
    if ((mx - mn) / mx < 0.0001) { ok = 1; break; }
  }
  pm_params(&p);
  /* frequency sampling: the response at i / N, i = 0 .. M, then the inverse cosine sum */
  double *A = (double *)malloc(sizeof(double) * (M + 1));
  for (int i = 0; i <= M; i++) A[i] = pm_response(&p, (double)i / N);
  for (int n = 0; n <= M; n++) {
    double val = A[0];
    const double xx = 2.0 * M_PI * (n - M) / N;
    for (int k = 1; k <= M; k++) val += 2.0 * A[k] * cos(xx * k);
    h[n] = val / N;
    h[N - 1 - n] = h[n];
  }
  free(A); free(found); free(p.ext); free(p.x); free(p.grid);
  return ok ? it + 1 : -1;
}

/* Stage A of the 140 dB (float32 data) class: N = 0.68 x the Kaiser estimate (made odd), pass band weight 1, stop bands
 * k mid -+ fstop (k = 1 .. D / 2, cut at in / 2) weight 800, the bands between them free -- what falls there is removed
 * by stage B.  Measured for D = 2 .. 20: ripple <= 0.0010 dB peak to peak, aliases of the pass band <= -142 dB
 * (tests/test_resampler_independent.py repeats the check against scipy.signal.remez).  Returns 0 if the exchange did
 * not settle (the caller keeps the Kaiser design). */
static int rs_design_equiripple(double in_rate, double mid, int D, double fpass, double fstop, int N, double *h) {
  double edges[2 * 40], des[40], wt[40];
  int nb = 0;
  edges[0] = 0.0; edges[1] = fpass / in_rate; des[0] = 1.0; wt[0] = 1.0; nb = 1;
  for (int k = 1; k <= D / 2 && nb < 40; k++) {
    const double lo = (k * mid - fstop) / in_rate;
    double hi = (k * mid + fstop) / in_rate;
    if (lo >= 0.5) break;
    if (hi > 0.5) hi = 0.5;
    edges[2 * nb] = lo; edges[2 * nb + 1] = hi; des[nb] = 0.0; wt[nb] = 800.0; nb++;
  }
  if (pm_design(N, nb, edges, des, wt, h) < 0) return 0;
  double sum = 0;
  for (int k = 0; k < N; k++) sum += h[k];
  for (int k = 0; k < N; k++) h[k] /= sum;
  /* The exchange's own stopping rule says the extremal errors agree, not how large they are: the response is checked
   * against the class (every stop band <= -140 dB, pass band ripple <= 0.0012 dB peak to peak) on 64 points per band plus the
   * edges; a design that misses -- 0.68 N does at D = 28, by 0.2 dB -- is not used (the caller tries a longer one). */
  const double c = 0.5 * (N - 1);
  double stop_max = 0.0, pass_lo = 1.0, pass_hi = 1.0;
  for (int b = 0; b < nb; b++)
    for (int g = 0; g <= 64; g++) {
      const double f = edges[2 * b] + (edges[2 * b + 1] - edges[2 * b]) * (double)g / 64.0;
      double re = 0.0;
      for (int k = 0; k < N; k++) re += h[k] * cos(2.0 * M_PI * f * ((double)k - c));
      if (b == 0) { if (re < pass_lo) pass_lo = re; if (re > pass_hi) pass_hi = re; }
      else if (fabs(re) > stop_max) stop_max = fabs(re);
    }
  if (stop_max > 1.0e-7 || pass_hi - pass_lo > 1.4e-4) return 0;      /* -140 dB; 0.0012 dB peak to peak */
  return 1;
}

struct ora_resampler {
  double in_rate, out_rate, atten;
  double pass_frac;  /* pass band edge as a fraction of out_rate/2 (0.885: the product's specification) */
  int stop_nyquist;  /* 1: stop band from out_rate/2 (nothing aliases at all, r8brain-class); 0: from out_rate - fpass */
  long long L, M;   /* out/in = L/M */
  int D;            /* stage A integer decimation (1 = bypass) */
  int NA;           /* stage A taps (odd), 0 when D == 1 */
  double *hA;
  long long LB, MB; /* stage B ratio out/mid = LB/MB */
  int TB;           /* stage B taps per phase (even) */
  double *hB;       /* [LB][TB]; fractional-phase form: [LT + 1][TB] */
  int LT;           /* 0: one table row per phase.  > 0 (fractional-phase form, ratios whose LB rows would not fit, e.g.
                     * ppm-corrected rates, main.cpp:708-711): rows are the prototype at mu = p / LT, p = 0 .. LT, and the
                     * taps of an output at mu = (k MB mod LB) / LB are interpolated linearly between rows
                     * floor(mu LT) and floor(mu LT) + 1 */
  /* streaming state */
  long long n_in;   /* inputs consumed so far */
  long long mA;     /* stage A outputs produced so far */
  long long kB;     /* outputs produced so far */
  double *xa; long long xa_base; int xa_len, xa_cap; /* input history */
  double *xm; long long xm_base; int xm_len, xm_cap; /* mid-rate history */
};

static void rs_design(ora_resampler *rs) {
  const double A = rs->atten;
  const double beta = 0.1102 * (A - 8.7);
  const double i0b = bessel_i0(beta);
  const double fpass = rs->pass_frac * rs->out_rate * 0.5;
  const double fstop = rs->stop_nyquist ? rs->out_rate * 0.5 : rs->out_rate - fpass;
  long long in_i = llround(rs->in_rate), out_i = llround(rs->out_rate);
  if (fabs(rs->in_rate - (double)in_i) > 1e-6 || fabs(rs->out_rate - (double)out_i) > 1e-6) {
    /* rates that are not whole hertz are taken to the millihertz: the ratio stays an exact rational */
    in_i = llround(rs->in_rate * 1000.0); out_i = llround(rs->out_rate * 1000.0);
  }
  long long g = gcd_ll(in_i, out_i);
  rs->L = out_i / g;
  rs->M = in_i / g;
  int D = (int)floor(rs->in_rate / (2.6 * rs->out_rate));
  if (D < 1) D = 1;
  rs->D = D;
  const double mid = rs->in_rate / D;
  if (D > 1) {
    const double f1 = fpass, f2 = mid - fstop;
    const double dw = 2.0 * M_PI * (f2 - f1) / rs->in_rate;
    int N = (int)ceil((A - 7.95) / (2.285 * dw)) + 1;
    if ((N & 1) == 0) N++;
    rs->NA = N;
    rs->hA = (double *)malloc(sizeof(double) * N);
    const double fc = 0.5 * (f1 + f2) / rs->in_rate; /* cycles/sample */
    const double c = 0.5 * (N - 1);
    double sum = 0;
    for (int k = 0; k < N; k++) {
      double t = (k - c);
      double r = t / c;
      double w = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) / i0b;
      rs->hA[k] = 2.0 * fc * sinc_pi(2.0 * fc * t) * w;
      sum += rs->hA[k];
    }
    for (int k = 0; k < N; k++) rs->hA[k] /= sum;
    if (A <= 150.0 && !rs->stop_nyquist && D <= 78) {
      /* the IF class: an equiripple stage A of 0.68 x the Kaiser length N if the exchange settles AND its response meets
       * the class (rs_design_equiripple checks it); else 4 % of N longer, up to 0.96 N; else the Kaiser window stays */
      for (int pc = 68; pc <= 96; pc += 4) {
        int NE = (int)((N * (long long)pc + 99) / 100);
        if ((NE & 1) == 0) NE++;
        double *he = (double *)malloc(sizeof(double) * NE);
        if (rs_design_equiripple(rs->in_rate, mid, D, fpass, fstop, NE, he)) { free(rs->hA); rs->hA = he; rs->NA = NE; break; }
        free(he);
      }
    }
  } else {
    rs->NA = 0;
    rs->hA = NULL;
  }
  {
    long long num = rs->L * D, den = rs->M;
    long long g2 = gcd_ll(num, den);
    rs->LB = num / g2;
    rs->MB = den / g2;
    const double dw = 2.0 * M_PI * (fstop - fpass) / ((double)rs->LB * mid);
    double nproto = (A - 7.95) / (2.285 * dw) + 1.0;
    int T = (int)ceil(nproto / (double)rs->LB);
    if (T & 1) T++;
    if (T < 2) T = 2;
    rs->TB = T;
    rs->LT = (rs->LB * (long long)T > (1ll << 22)) ? 1024 : 0;
    const long long rows = rs->LT ? rs->LT + 1 : rs->LB, prow = rs->LT ? rs->LT : rs->LB;
    rs->hB = (double *)malloc(sizeof(double) * (size_t)rows * T);
    const double W = 0.5 * T;
    const double fc = 0.5 * rs->out_rate / mid; /* cycles per mid sample */
    double sum = 0;
    for (long long p = 0; p < rows; p++) {
      for (int j = 0; j < T; j++) {
        double t = (double)p / (double)prow + W - 1.0 - j;
        double r = t / W;
        double w = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) / i0b;
        double v = 2.0 * fc * sinc_pi(2.0 * fc * t) * w;
        rs->hB[p * T + j] = v;
        if (p < prow) sum += v;
      }
    }
    const double scale = (double)prow / sum;
    for (long long i = 0; i < rows * T; i++) rs->hB[i] *= scale;
  }
}

/* ora_rs_create2: the same two-stage structure with another specification -- used by the tests to build an
 * "r8brain-class" resampler (2 % transition band, >= 180 dB, the defaults of r8b::CDSPResampler24 the reference
 * constructs at IfResampler.cpp:26-29) beside the product's, and measure what the difference does to the audio. */
ora_resampler *ora_rs_create2(double in_rate, double out_rate, double atten_db, double pass_frac, int stop_nyquist);
ora_resampler *ora_rs_create(double in_rate, double out_rate, double atten_db) {
  return ora_rs_create2(in_rate, out_rate, atten_db, 0.885, 0);
}
ora_resampler *ora_rs_create2(double in_rate, double out_rate, double atten_db, double pass_frac, int stop_nyquist) {
  ora_resampler *rs = (ora_resampler *)calloc(1, sizeof(*rs));
  rs->in_rate = in_rate;
  rs->out_rate = out_rate;
  rs->atten = atten_db;
  rs->pass_frac = pass_frac;
  rs->stop_nyquist = stop_nyquist;
  rs_design(rs);
  rs->xa_cap = 1 << 12; rs->xa = (double *)malloc(sizeof(double) * rs->xa_cap);
  rs->xm_cap = 1 << 12; rs->xm = (double *)malloc(sizeof(double) * rs->xm_cap);
  /* zero history: x[n<0] = 0 (latency-compensated start) */
  int ca = rs->NA ? (rs->NA - 1) / 2 : 0;
  rs->xa_base = -ca; rs->xa_len = ca;
  memset(rs->xa, 0, sizeof(double) * ca);
  int W = rs->TB / 2;
  rs->xm_base = -(W - 1); rs->xm_len = W - 1;
  memset(rs->xm, 0, sizeof(double) * (W - 1));
  return rs;
}

void ora_rs_destroy(ora_resampler *rs) {
  if (!rs) return;
  free(rs->hA); free(rs->hB); free(rs->xa); free(rs->xm); free(rs);
}

long long ora_rs_info(const ora_resampler *rs, int which) {
  switch (which) {
  case 0: return rs->D;
  case 1: return rs->NA;
  case 2: return rs->LB;
  case 3: return rs->MB;
  case 4: return rs->TB;
  case 5: return rs->L;
  case 6: return rs->M;
  case 7: return rs->LT;
  }
  return -1;
}
const double *ora_rs_taps_a(const ora_resampler *rs) { return rs->hA; }
const double *ora_rs_taps_b(const ora_resampler *rs) { return rs->hB; }

static void buf_append(double **buf, int *len, int *cap, const double *src, int n) {
  if (*len + n > *cap) {
    while (*len + n > *cap) *cap *= 2;
    *buf = (double *)realloc(*buf, sizeof(double) * (size_t)*cap);
  }
  memcpy(*buf + *len, src, sizeof(double) * (size_t)n);
  *len += n;
}

int ora_rs_process(ora_resampler *rs, const double *in, int n, double *out, int cap) {
  if (n <= 0) return 0;
  const int D = rs->D;
  const int ca = rs->NA ? (rs->NA - 1) / 2 : 0;
  const int W = rs->TB / 2;
  /* ---- stage A: y_A[m] = sum_k hA[k] x[D m + ca - k] ---- */
  rs->n_in += n;
  long long mA_avail;
  if (D == 1) {
    buf_append(&rs->xm, &rs->xm_len, &rs->xm_cap, in, n);
    mA_avail = rs->n_in;
  } else {
    buf_append(&rs->xa, &rs->xa_len, &rs->xa_cap, in, n);
    mA_avail = (rs->n_in >= ca + 1) ? (rs->n_in - 1 - ca) / D + 1 : 0;
    for (long long m = rs->mA; m < mA_avail; m++) {
      long long top = (long long)D * m + ca; /* newest input index used */
      const double *xp = rs->xa + (top - rs->xa_base);
      double acc = 0.0;
      for (int k = 0; k < rs->NA; k++) acc += rs->hA[k] * xp[-k];
      buf_append(&rs->xm, &rs->xm_len, &rs->xm_cap, &acc, 1);
    }
    /* drop inputs older than D*mA_avail - ca */
    long long keep_from = (long long)D * mA_avail - ca;
    if (keep_from > rs->xa_base) {
      int drop = (int)(keep_from - rs->xa_base);
      if (drop > rs->xa_len) drop = rs->xa_len;
      memmove(rs->xa, rs->xa + drop, sizeof(double) * (size_t)(rs->xa_len - drop));
      rs->xa_len -= drop; rs->xa_base += drop;
    }
  }
  rs->mA = mA_avail;
  /* ---- stage B: y[k] = sum_j hB[p][j] mid[n_k - W + 1 + j] ---- */
  long long kB_avail = 0;
  if (rs->mA >= W + 1) {
    long long q1 = rs->mA - W; /* Q + 1 */
    kB_avail = (long long)(((__int128)q1 * rs->LB + rs->MB - 1) / rs->MB);
  }
  int nout = (int)(kB_avail - rs->kB);
  if (nout > cap) return -1;
  for (long long k = rs->kB; k < kB_avail; k++) {
    const __int128 t = (__int128)k * rs->MB;
    long long nk = (long long)(t / rs->LB);
    long long p = (long long)(t % rs->LB);
    const double *xp = rs->xm + (nk - W + 1 - rs->xm_base);
    double acc = 0.0;
    if (rs->LT) {
      const __int128 x = (__int128)p * rs->LT;
      const long long row = (long long)(x / rs->LB);
      const double a = (double)(long long)(x % rs->LB) / (double)rs->LB;
      const double *h0 = rs->hB + row * rs->TB, *h1 = h0 + rs->TB;
      for (int j = 0; j < rs->TB; j++) acc += (h0[j] + a * (h1[j] - h0[j])) * xp[j];
    } else {
      const double *h = rs->hB + p * rs->TB;
      for (int j = 0; j < rs->TB; j++) acc += h[j] * xp[j];
    }
    out[k - rs->kB] = acc;
  }
  rs->kB = kB_avail;
  {
    long long nk = (long long)(((__int128)rs->kB * rs->MB) / rs->LB);
    long long keep_from = nk - W + 1;
    if (keep_from > rs->xm_base) {
      int drop = (int)(keep_from - rs->xm_base);
      if (drop > rs->xm_len) drop = rs->xm_len;
      memmove(rs->xm, rs->xm + drop, sizeof(double) * (size_t)(rs->xm_len - drop));
      rs->xm_len -= drop; rs->xm_base += drop;
    }
  }
  return nout;
}

/* IfResampler: sfmbase/IfResampler.cpp:37-78.  Deinterleave to two double
 * arrays (V8, :50), run two real resamplers in lock-step (:56-59), narrow the
 * double outputs to complex float (:69-72). */
struct ora_ifr { ora_resampler *re, *im; double *bre, *bim, *ore, *oim; int cap; };

#define ORA_IF_ATTEN_DB 140.0
#define ORA_AUDIO_ATTEN_DB 180.0

ora_ifr *ora_ifr_create(double in_rate, double out_rate) {
  ora_ifr *h = (ora_ifr *)calloc(1, sizeof(*h));
  h->re = ora_rs_create(in_rate, out_rate, ORA_IF_ATTEN_DB);
  h->im = ora_rs_create(in_rate, out_rate, ORA_IF_ATTEN_DB);
  return h;
}
ora_ifr *ora_ifr_create2(double in_rate, double out_rate, double atten_db, double pass_frac, int stop_nyquist) {
  ora_ifr *h = (ora_ifr *)calloc(1, sizeof(*h));
  h->re = ora_rs_create2(in_rate, out_rate, atten_db, pass_frac, stop_nyquist);
  h->im = ora_rs_create2(in_rate, out_rate, atten_db, pass_frac, stop_nyquist);
  return h;
}
void ora_ifr_destroy(ora_ifr *h) {
  if (!h) return;
  ora_rs_destroy(h->re); ora_rs_destroy(h->im);
  free(h->bre); free(h->bim); free(h->ore); free(h->oim); free(h);
}
int ora_ifr_process(ora_ifr *h, const float *iq, int n, float *out_iq, int cap) {
  if (n > h->cap) {
    h->cap = n;
    h->bre = (double *)realloc(h->bre, sizeof(double) * n);
    h->bim = (double *)realloc(h->bim, sizeof(double) * n);
    h->ore = (double *)realloc(h->ore, sizeof(double) * (n + 16));
    h->oim = (double *)realloc(h->oim, sizeof(double) * (n + 16));
  }
  for (int i = 0; i < n; i++) { h->bre[i] = iq[2 * i]; h->bim[i] = iq[2 * i + 1]; }
  int ocap = n + 16;
  int nr = ora_rs_process(h->re, h->bre, n, h->ore, ocap);
  int ni = ora_rs_process(h->im, h->bim, n, h->oim, ocap);
  if (nr < 0 || nr != ni || nr > cap) return -1;
  for (int i = 0; i < nr; i++) {
    out_iq[2 * i] = (float)h->ore[i];
    out_iq[2 * i + 1] = (float)h->oim[i];
  }
  return nr;
}

/* ======================================================================== */
/* LowPassFilterFirIQ -- sfmbase/Filter.cpp:27-96                             */
/* ======================================================================== */
struct ora_firiq { float *coeff; float *state; /* order complex */ unsigned order, downsample, pos; };

ora_firiq *ora_firiq_create(const float *coeff, int ntaps, int downsample) {
  ora_firiq *f = (ora_firiq *)calloc(1, sizeof(*f));
  f->coeff = (float *)malloc(sizeof(float) * ntaps);
  memcpy(f->coeff, coeff, sizeof(float) * ntaps);
  f->order = (unsigned)(ntaps - 1);                       /* :29 */
  f->downsample = (unsigned)downsample;
  f->pos = 0;
  f->state = (float *)calloc(2 * (size_t)f->order + 2, sizeof(float)); /* :33 */
  return f;
}
void ora_firiq_destroy(ora_firiq *f) { if (f) { free(f->coeff); free(f->state); free(f); } }

int ora_firiq_process(ora_firiq *f, const float *in, int n_, float *out) {
  const unsigned order = f->order, n = (unsigned)n_, pstep = f->downsample;
  unsigned p = f->pos;
  if (n == 0) return 0;                                     /* :48-51 */
  const unsigned nout = (n - p + pstep - 1) / pstep;        /* :54 */
  const float *c = f->coeff, *st = f->state;
  unsigned i = 0;
  /* head: lags 1..order only, un-folded (Filter.cpp:59-68; hazard H1) */
  for (; p < n && p < order; p += pstep, i++) {
    float yr = 0, yi = 0;
    for (unsigned j = p + 1; j <= order; j++) {
      unsigned s = order + p - j;
      yr += st[2 * s] * c[j]; yi += st[2 * s + 1] * c[j];
    }
    for (unsigned j = 1; j <= p; j++) {
      yr += in[2 * (p - j)] * c[j]; yi += in[2 * (p - j) + 1] * c[j];
    }
    out[2 * i] = yr; out[2 * i + 1] = yi;
  }
  /* body: folded symmetric form incl. lag 0 (Filter.cpp:73-82) */
  const unsigned half_order = (order - 1) / 2;
  for (; p < n; p += pstep, i++) {
    float yr = 0, yi = 0;
    for (unsigned k = 0; k <= half_order; k++) {
      float sr = in[2 * (p - k)] + in[2 * (p - (order - k))];
      float si = in[2 * (p - k) + 1] + in[2 * (p - (order - k)) + 1];
      yr += sr * c[k]; yi += si * c[k];
    }
    if ((order % 2) == 0) {
      yr += in[2 * (p - order / 2)] * c[order / 2];
      yi += in[2 * (p - order / 2) + 1] * c[order / 2];
    }
    out[2 * i] = yr; out[2 * i + 1] = yi;
  }
  f->pos = p - n;                                           /* :87 */
  if (n < order) {                                          /* :90-95 */
    memmove(f->state, f->state + 2 * n, sizeof(float) * 2 * (order - n));
    memcpy(f->state + 2 * (order - n), in, sizeof(float) * 2 * n);
  } else {
    memcpy(f->state, in + 2 * (n - order), sizeof(float) * 2 * order);
  }
  return (int)nout;
}

/* LowPassFilterFirAudio -- sfmbase/Filter.cpp:101-163 (same shape, double) */
struct ora_firaudio { double *coeff; double *state; unsigned order, pos; };

ora_firaudio *ora_firaudio_create(const double *coeff, int ntaps) {
  ora_firaudio *f = (ora_firaudio *)calloc(1, sizeof(*f));
  f->coeff = (double *)malloc(sizeof(double) * ntaps);
  memcpy(f->coeff, coeff, sizeof(double) * ntaps);
  f->order = (unsigned)(ntaps - 1);
  f->state = (double *)calloc((size_t)f->order + 1, sizeof(double));
  return f;
}
void ora_firaudio_destroy(ora_firaudio *f) { if (f) { free(f->coeff); free(f->state); free(f); } }

int ora_firaudio_process(ora_firaudio *f, const double *in, int n_, double *out) {
  const unsigned order = f->order, n = (unsigned)n_;
  unsigned p = f->pos;
  if (n == 0) return 0;
  const unsigned nout = n - p;
  const double *c = f->coeff, *st = f->state;
  unsigned i = 0;
  for (; p < n && p < order; p++, i++) {                    /* :126-135 */
    double y = 0;
    for (unsigned j = p + 1; j <= order; j++) y += st[order + p - j] * c[j];
    for (unsigned j = 1; j <= p; j++) y += in[p - j] * c[j];
    out[i] = y;
  }
  const unsigned half_order = (order - 1) / 2;
  for (; p < n; p++, i++) {                                 /* :140-149 */
    double y = 0;
    for (unsigned k = 0; k <= half_order; k++) y += (in[p - k] + in[p - (order - k)]) * c[k];
    if ((order % 2) == 0) y += in[p - order / 2] * c[order / 2];
    out[i] = y;
  }
  f->pos = p - n;
  if (n < order) {
    memmove(f->state, f->state + n, sizeof(double) * (order - n));
    memcpy(f->state + (order - n), in, sizeof(double) * n);
  } else {
    memcpy(f->state, in + (n - order), sizeof(double) * order);
  }
  return (int)nout;
}

/* ======================================================================== */
/* IIR sections -- sfmbase/Filter.cpp:167-290                                 */
/* ======================================================================== */
void ora_iir1_init(ora_iir1 *f, double b0, double b1, double a1) {
  f->b0 = b0; f->b1 = b1; f->a1 = a1; f->x1 = 0;          /* :167-169 */
}
double ora_iir1_step(ora_iir1 *f, double x) {               /* :172-178 */
  double x0 = x;
  x0 -= f->a1 * f->x1;
  double y = f->b0 * x0 + f->b1 * f->x1;
  f->x1 = x0;
  return y;
}
void ora_biquad_init(ora_biquad *f, double b0, double b1, double b2, double a1, double a2) {
  f->b0 = b0; f->b1 = b1; f->b2 = b2; f->a1 = a1; f->a2 = a2; f->x1 = 0; f->x2 = 0;
}
double ora_biquad_step(ora_biquad *f, double x) {           /* :243-250 */
  double x0 = x;
  x0 -= f->a1 * f->x1 + f->a2 * f->x2;
  double y = f->b0 * x0 + f->b1 * f->x1 + f->b2 * f->x2;
  f->x2 = f->x1;
  f->x1 = x0;
  return y;
}
void ora_lowpass_rc_init(ora_iir1 *f, double timeconst) {   /* :186-188 */
  double a1 = -exp(-1 / timeconst);
  double b0 = 1 + a1;
  ora_iir1_init(f, b0, 0, a1);
}
void ora_highpass_init(ora_biquad *f, double cutoff) {      /* :254-290 */
  /* p1s = w / exp(j*3pi/4) = w * exp(-j*3pi/4);  p1z = exp(p1s) */
  double w = 2 * M_PI * cutoff;
  double ang = (2 * 1 + 2 - 1) / (double)(2 * 2) * M_PI;
  /* w / (cos a + j sin a) = w (cos a - j sin a) */
  double sr = w * cos(ang), si = -w * sin(ang);
  double er = exp(sr);
  double pr = er * cos(si), pi_ = er * sin(si);
  double b0 = 1, b1 = -2, b2 = 1;
  double a1 = -2 * pr;
  /* abs(p1z*p1z) */
  double qr = pr * pr - pi_ * pi_, qi = 2 * pr * pi_;
  double a2 = hypot(qr, qi);
  double g = (b0 - b1 + b2) / (1 - a1 + a2);
  ora_biquad_init(f, b0 / g, b1 / g, b2 / g, a1, a2);
}

/* ======================================================================== */
/* Utility.h                                                                  */
/* ======================================================================== */
float ora_rms_level(const float *iq, int n) {               /* :118-132 */
  if (n == 0) return 0.0f;
  float level = 0;
  for (int i = 0; i < n; i++) {
    float m = iq[2 * i] * iq[2 * i] + iq[2 * i + 1] * iq[2 * i + 1]; /* V1 */
    level += m;                                                       /* V2 */
  }
  return sqrtf(level / (float)(unsigned)n);
}
void ora_mean_rms(const float *x, int n, float *mean, float *rms) { /* :135-152 */
  if (n == 0) { *mean = 0; *rms = 0; return; }
  float vsum = 0, vsumsq = 0;
  for (int i = 0; i < n; i++) vsum += x[i];          /* V2 */
  for (int i = 0; i < n; i++) vsumsq += x[i] * x[i]; /* V3 */
  *mean = vsum / (float)(unsigned)n;
  *rms = sqrtf(vsumsq / (float)(unsigned)n);
}

/* fast_atan_table: Utility.h:165-217.  Regenerated, not transcribed:
 * entry i == float("%.6e" % atan(i/255)), entry 256 repeats entry 255
 * (SURVEY.md 8a "Constants that can be regenerated"; checked against the
 * reference text by tests/test_reference_pins.py). */
static float g_atan_table[257];
static int g_atan_ready = 0;
static void atan_table_init(void) {
  if (g_atan_ready) return;
  char buf[64];
  for (int i = 0; i < 256; i++) {
    snprintf(buf, sizeof buf, "%.6e", atan((double)i / 255.0));
    g_atan_table[i] = (float)strtod(buf, NULL);
  }
  g_atan_table[256] = g_atan_table[255];
  g_atan_ready = 1;
}
const float *ora_fast_atan_table(void) { atan_table_init(); return g_atan_table; }

float ora_fast_atan2f(float y, float x) {                   /* :236-304 */
  atan_table_init();
  float x_abs, y_abs, z, alpha, angle, base_angle;
  int index;
  y_abs = fabsf(y);
  x_abs = fabsf(x);
  if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
  if (y_abs < x_abs) z = y_abs / x_abs; else z = x_abs / y_abs;
  if (z < 0.003921569) {            /* TAN_MAP_RES, compared in double */
    base_angle = z;
  } else {
    alpha = z * (float)255;         /* TAN_MAP_SIZE */
    index = ((int)alpha) & 0xff;
    alpha -= (float)index;
    base_angle = g_atan_table[index];
    base_angle += (g_atan_table[index + 1] - g_atan_table[index]) * alpha;
  }
  if (x_abs > y_abs) {
    if (x >= 0.0) {
      angle = (y >= 0.0) ? base_angle : -base_angle;
    } else {
      angle = 3.14159265358979323846;
      if (y >= 0.0) angle -= base_angle; else angle = base_angle - angle;
    }
  } else {
    if (y >= 0.0) {
      angle = 1.57079632679489661923;
      if (x >= 0.0) angle -= base_angle; else angle += base_angle;
    } else {
      angle = -1.57079632679489661923;
      if (x >= 0.0) angle += base_angle; else angle -= base_angle;
    }
  }
  return angle;
}

/* ======================================================================== */
/* AGCs                                                                       */
/* ======================================================================== */
void ora_ifagc_init(ora_ifagc *a, float initial, float max_gain, float rate) {
  a->initial_gain = initial; a->max_gain = max_gain; a->rate = rate;
  a->current_gain = initial;                                /* IfSimpleAgc.cpp:26-34 */
}
void ora_ifagc_process(ora_ifagc *a, const float *in, int n, float *out) { /* :37-57 */
  for (int i = 0; i < n; i++) {
    float xr = in[2 * i] * a->current_gain;
    float xi = in[2 * i + 1] * a->current_gain;
    out[2 * i] = xr; out[2 * i + 1] = xi;
    float nrm = xr * xr + xi * xi;                                   /* std::norm */
    float z = (float)(1.0 + ((double)a->rate * (1.0 - (double)nrm))); /* :46, H6 */
    a->current_gain *= z;
    if (!isfinite(a->current_gain)) {
      a->current_gain = a->initial_gain;
    } else if (a->current_gain > a->max_gain) {
      a->current_gain = a->max_gain;
    }
  }
}
void ora_afagc_init(ora_afagc *a, double initial, double max_gain, double reference, double rate) {
  a->initial_gain = initial; a->max_gain = max_gain; a->reference = reference;
  a->rate = rate; a->current_gain = initial;                /* AfSimpleAgc.cpp:26-34 */
}
void ora_afagc_process(ora_afagc *a, const double *in, int n, double *out) { /* :36-56 */
  for (int i = 0; i < n; i++) {
    double x2 = in[i] * a->current_gain;
    out[i] = x2 * a->reference;
    double z = 1.0 + (a->rate * (1.0 - (x2 * x2)));
    a->current_gain *= z;
    if (!isfinite(a->current_gain)) {
      a->current_gain = a->initial_gain;
    } else if (a->current_gain > a->max_gain) {
      a->current_gain = a->max_gain;
    }
  }
}

/* ======================================================================== */
/* PhaseDiscriminator -- sfmbase/PhaseDiscriminator.cpp:27-46                 */
/* ======================================================================== */
void ora_disc_init(ora_disc *d, double max_freq_dev) {
  d->normalize_factor = (float)(max_freq_dev * 2.0 * M_PI); /* double passed as float, H6 */
  d->boundary = (float)(1.0 / (max_freq_dev * 2.0));
  d->save_value = 0;
}
void ora_disc_process(ora_disc *d, const float *iq, int n, float *out) {
  float prev = d->save_value;
  const float bound = d->boundary;
  for (int i = 0; i < n; i++) {
    /* V4: atan2f(im, re) / normalizeFactor */
    float ph = atan2f(iq[2 * i + 1], iq[2 * i]) / d->normalize_factor;
    /* V5: wrapped first difference with carried last phase */
    float v = ph - prev;
    if (v > bound) v -= 2 * bound;
    if (v < -bound) v += 2 * bound;
    prev = ph;
    if (isnan(v)) v = 0;                                    /* Utility.h:336-343 */
    out[i] = v;
  }
  if (n > 0) d->save_value = prev;
}

/* ======================================================================== */
/* PilotPhaseLock -- sfmbase/PilotPhaseLock.cpp:35-171                        */
/* ======================================================================== */
struct ora_pll {
  double minfreq, maxfreq, freq, phase, pilot_level, freq_err;
  int lock_delay, lock_cnt, pilot_periods;
  uint64_t pps_cnt, sample_cnt;
  ora_pps_event *events; int n_events, cap_events;
  ora_biquad bq_i, bq_q;
  ora_iir1 lf;
};
#define PLL_SAMPLE_RATE_IF 384000.0
#define PLL_BANDWIDTH (30 / PLL_SAMPLE_RATE_IF)  /* PilotPhaseLock.h:35 */
#define PLL_MINSIGNAL 0.001                      /* PilotPhaseLock.h:37 */
#define PLL_PILOT_FREQUENCY 19000                /* PilotPhaseLock.h:31 */

ora_pll *ora_pll_create(double freq) {
  ora_pll *p = (ora_pll *)calloc(1, sizeof(*p));
  p->minfreq = (freq - PLL_BANDWIDTH) * 2.0 * M_PI;         /* :37 */
  p->maxfreq = (freq + PLL_BANDWIDTH) * 2.0 * M_PI;
  p->freq = freq * 2.0 * M_PI;
  p->phase = 0;
  p->pilot_level = 0;
  p->lock_delay = (int)(15.0 / PLL_BANDWIDTH);              /* :43 */
  p->lock_cnt = 0;
  ora_biquad_init(&p->bq_i, 1.46974784e-06, 0, 0, -1.99682419, 0.996825659); /* :48 */
  ora_biquad_init(&p->bq_q, 1.46974784e-06, 0, 0, -1.99682419, 0.996825659); /* :49 */
  ora_iir1_init(&p->lf, 0.000304341788, -0.000304324564, 0);                 /* :51 */
  p->cap_events = 8;
  p->events = (ora_pps_event *)malloc(sizeof(ora_pps_event) * p->cap_events);
  return p;
}
void ora_pll_destroy(ora_pll *p) { if (p) { free(p->events); free(p); } }

void ora_pll_process(ora_pll *p, const double *in, int n_, double *out, int pilot_shift) {
  unsigned n = (unsigned)n_;
  int was_locked = (p->lock_cnt >= p->lock_delay);          /* :62 */
  p->n_events = 0;
  if (n > 0) p->pilot_level = 1000.0; else return;          /* :65-71 */
  for (unsigned i = 0; i < n; i++) {
    double psin = sin(p->phase);
    double pcos = cos(p->phase);
    if (pilot_shift) out[i] = 2 * pcos * pcos - 1;          /* :80-88 */
    else out[i] = 2 * psin * pcos;
    double x = in[i];
    double phasor_i = psin * x;
    double phasor_q = pcos * x;
    double new_i = ora_biquad_step(&p->bq_i, phasor_i);
    double new_q = ora_biquad_step(&p->bq_q, phasor_q);
    double phase_err = ora_fast_atan2f((float)new_q, (float)new_i); /* :103, H6 */
    p->pilot_level = sqrt((new_i * new_i) + (new_q * new_q));       /* :106 */
    double new_phase_err = ora_iir1_step(&p->lf, phase_err);
    p->freq_err = new_phase_err;
    p->freq += p->freq_err;
    p->freq = fmax(p->minfreq, fmin(p->maxfreq, p->freq));  /* :119 */
    p->phase += p->freq;
    if (p->phase > 2.0 * M_PI) {                            /* :134 */
      p->phase -= 2.0 * M_PI;
      p->pilot_periods++;
      if (p->pilot_periods == PLL_PILOT_FREQUENCY) {
        p->pilot_periods = 0;
        if (was_locked) {
          if (p->n_events == p->cap_events) {
            p->cap_events *= 2;
            p->events = (ora_pps_event *)realloc(p->events, sizeof(ora_pps_event) * p->cap_events);
          }
          ora_pps_event *ev = &p->events[p->n_events++];
          ev->pps_index = p->pps_cnt;
          ev->sample_index = p->sample_cnt + i;
          ev->block_position = (double)i / (double)n;
          p->pps_cnt++;
        }
      }
    }
  }
  if (2 * p->pilot_level > PLL_MINSIGNAL) {                 /* :154-160 */
    if (p->lock_cnt < p->lock_delay) p->lock_cnt += (int)n;
  } else {
    p->lock_cnt = 0;
  }
  if (p->lock_cnt < p->lock_delay) {                        /* :163-167 */
    p->pilot_periods = 0;
    p->pps_cnt = 0;
    p->n_events = 0;
  }
  p->sample_cnt += n;
}
/* test hooks: the 7 continuous state variables (phase, freq, loop-filter delay,
 * biquad I delays x1,x2, biquad Q delays x1,x2) */
void ora_pll_get_state(const ora_pll *p, double *s) {
  s[0] = p->phase; s[1] = p->freq; s[2] = p->lf.x1;
  s[3] = p->bq_i.x1; s[4] = p->bq_i.x2; s[5] = p->bq_q.x1; s[6] = p->bq_q.x2;
}
void ora_pll_set_state(ora_pll *p, const double *s) {
  p->phase = s[0]; p->freq = s[1]; p->lf.x1 = s[2];
  p->bq_i.x1 = s[3]; p->bq_i.x2 = s[4]; p->bq_q.x1 = s[5]; p->bq_q.x2 = s[6];
}
int ora_pll_locked(const ora_pll *p) { return p->lock_cnt >= p->lock_delay; }
double ora_pll_pilot_level(const ora_pll *p) { return 2 * p->pilot_level; }
double ora_pll_freq_err(const ora_pll *p) { return p->freq_err; }
double ora_pll_phase(const ora_pll *p) { return p->phase; }
double ora_pll_freq(const ora_pll *p) { return p->freq; }
int ora_pll_pps_events(const ora_pll *p, ora_pps_event *ev, int cap) {
  int n = p->n_events < cap ? p->n_events : cap;
  if (ev) memcpy(ev, p->events, sizeof(ora_pps_event) * n);
  return p->n_events;
}

/* ======================================================================== */
/* MultipathFilter -- sfmbase/MultipathFilter.cpp:39-197                      */
/* ======================================================================== */
struct ora_mpf {
  unsigned stages, ref, order;
  float mu;
  float *coeff, *state; /* interleaved complex, order entries */
  double error;
};
#define MPF_ALPHA 0.1 /* MultipathFilter.h:44 */

void ora_mpf_initialize_coefficients(ora_mpf *m) {          /* :77-89 */
  memset(m->coeff, 0, sizeof(float) * 2 * m->order);
  m->coeff[2 * m->ref] = 1.0f;
}
ora_mpf *ora_mpf_create(unsigned stages) {                  /* :39-75 */
  ora_mpf *m = (ora_mpf *)calloc(1, sizeof(*m));
  m->stages = stages;
  m->ref = stages * 3 + 1;
  m->order = stages * 4 + 1;
  m->mu = (float)(MPF_ALPHA / m->order);
  m->coeff = (float *)calloc(2 * (size_t)m->order, sizeof(float));
  m->state = (float *)calloc(2 * (size_t)m->order, sizeof(float));
  m->error = 0;
  ora_mpf_initialize_coefficients(m);
  return m;
}
void ora_mpf_destroy(ora_mpf *m) { if (m) { free(m->coeff); free(m->state); free(m); } }

int ora_mpf_process(ora_mpf *m, const float *in, int n_, float *out) { /* :164-197 */
  const unsigned n = (unsigned)n_, N = m->order;
  if (n == 0) return 1;
  for (unsigned i = 0; i < n; i++) {
    /* single_process :92-105 : drop oldest, append newest at the end */
    memmove(m->state, m->state + 2, sizeof(float) * 2 * (N - 1));
    m->state[2 * (N - 1)] = in[2 * i];
    m->state[2 * (N - 1) + 1] = in[2 * i + 1];
    float yr = 0, yi = 0;
    for (unsigned k = 0; k < N; k++) { /* V9 as the loop stated at :98-101 */
      float sr = m->state[2 * k], si = m->state[2 * k + 1];
      float cr = m->coeff[2 * k], ci = m->coeff[2 * k + 1];
      yr += sr * cr - si * ci;
      yi += sr * ci + si * cr;
    }
    if (!isfinite(yr) || !isfinite(yi)) return 0;           /* :182-184 */
    out[2 * i] = yr; out[2 * i + 1] = yi;
    if ((i & 0x03) == 0) {                                  /* :176,186; hazard H2 */
      /* update_coeff :108-161 */
      const double env = (double)(yr * yr + yi * yi);       /* std::norm (float) */
      const double error = 1.0 - env;
      float sum = 0;
      for (unsigned k = 0; k < N; k++) {                    /* V1 + V2 */
        float sr = m->state[2 * k], si = m->state[2 * k + 1];
        float ms = sr * sr + si * si;
        sum += ms;
      }
      m->mu = (float)(MPF_ALPHA / ((double)sum + 1e-10));   /* :130 */
      const float factor = (float)(error * (double)m->mu);  /* :133, H6 */
      const float fr = factor * yr, fi = factor * yi;
      for (unsigned k = 0; k < N; k++) {                    /* V10: c += conj(s)*f */
        float sr = m->state[2 * k], si = m->state[2 * k + 1];
        m->coeff[2 * k] += sr * fr + si * fi;
        m->coeff[2 * k + 1] += sr * fi - si * fr;
      }
      m->coeff[2 * m->ref] = 1.0f; m->coeff[2 * m->ref + 1] = 0.0f; /* :158 */
      m->error = error;
      if (!isfinite(m->error)) return 0;                    /* :190-192 */
    }
  }
  return 1;
}
double ora_mpf_error(const ora_mpf *m) { return m->error; }
int ora_mpf_order(const ora_mpf *m) { return (int)m->order; }
const float *ora_mpf_coeff(const ora_mpf *m) { return m->coeff; }

/* ======================================================================== */
/* FourthConverterIQ -- include/FourthConverterIQ.h:30-82                     */
/* ======================================================================== */
void ora_fourth_init(ora_fourth *f, int up) {
  f->index = 0;
  f->t0 = up ? 3 : 1; f->t1 = up ? 0 : 2; f->t2 = up ? 1 : 3; f->t3 = up ? 2 : 0;
}
void ora_fourth_process(ora_fourth *f, const float *in, int n, float *out) {
  unsigned idx = f->index;
  for (int i = 0; i < n; i++) {
    float re = in[2 * i], im = in[2 * i + 1];
    switch (idx) {
    case 0: out[2 * i] = re; out[2 * i + 1] = im; idx = f->t0; break;
    case 1: out[2 * i] = im; out[2 * i + 1] = -re; idx = f->t1; break;
    case 2: out[2 * i] = -re; out[2 * i + 1] = -im; idx = f->t2; break;
    default: out[2 * i] = -im; out[2 * i + 1] = re; idx = f->t3; break;
    }
  }
  f->index = idx;
}

/* ======================================================================== */
/* FmDecoder -- sfmbase/FmDecode.cpp:25-283                                   */
/* ======================================================================== */
struct ora_fm {
  int fmfilter_enable, pilot_shift, enable_multipath, stereo_enabled, stereo_detected;
  unsigned wait_multipath_blocks;
  float baseband_mean, baseband_level, if_rms;
  ora_firiq *fmfilter;
  ora_resampler *rs_mono, *rs_stereo;
  ora_firaudio *pilotcut_mono, *pilotcut_stereo;
  ora_disc disc;
  ora_pll *pll;
  ora_biquad dc_mono, dc_stereo;
  ora_iir1 de_mono, de_stereo;
  ora_ifagc ifagc;
  ora_mpf *mpf;
  /* work buffers */
  int cap;
  float *b_filt, *b_agc, *b_mpf, *b_dec;
  double *b_base, *b_raw, *b_mono1, *b_st1, *b_mono, *b_st;
  int last_n, last_n_audio;
};
#define FM_SAMPLE_RATE_IF 384000.0
#define FM_SAMPLE_RATE_PCM 48000.0
#define FM_FREQ_DEV 75000.0
#define FM_PILOT_FREQ 19000.0

ora_fm *ora_fm_create(int fmfilter_enable, const float *coeff, int ncoeff, int stereo,
                      double deemphasis, int pilot_shift, unsigned multipath_stages,
                      const double *pilotcut, int n_pilotcut) {
  ora_fm *fm = (ora_fm *)calloc(1, sizeof(*fm));
  fm->fmfilter_enable = fmfilter_enable;
  fm->pilot_shift = pilot_shift;
  fm->enable_multipath = multipath_stages > 0;
  fm->wait_multipath_blocks = 100;                          /* :33 */
  fm->stereo_enabled = stereo;
  fm->fmfilter = ora_firiq_create(coeff, ncoeff, 1);        /* :39 */
  fm->rs_mono = ora_rs_create(FM_SAMPLE_RATE_IF, FM_SAMPLE_RATE_PCM, ORA_AUDIO_ATTEN_DB);
  fm->rs_stereo = ora_rs_create(FM_SAMPLE_RATE_IF, FM_SAMPLE_RATE_PCM, ORA_AUDIO_ATTEN_DB);
  fm->pilotcut_mono = ora_firaudio_create(pilotcut, n_pilotcut);   /* :48-49 */
  fm->pilotcut_stereo = ora_firaudio_create(pilotcut, n_pilotcut);
  ora_disc_init(&fm->disc, FM_FREQ_DEV / FM_SAMPLE_RATE_IF);       /* :53 */
  fm->pll = ora_pll_create(FM_PILOT_FREQ / FM_SAMPLE_RATE_IF);     /* :57 */
  ora_highpass_init(&fm->dc_mono, 0.0001);                         /* :62 */
  ora_highpass_init(&fm->dc_stereo, 0.0001);
  double tc = (deemphasis == 0) ? 1.0 : (deemphasis * FM_SAMPLE_RATE_IF * 1.0e-6); /* :67-70 */
  ora_lowpass_rc_init(&fm->de_mono, tc);
  ora_lowpass_rc_init(&fm->de_stereo, tc);
  ora_ifagc_init(&fm->ifagc, 1.0f, 100000.0f, 0.0001f);            /* :74 */
  fm->mpf = ora_mpf_create(fm->enable_multipath ? multipath_stages : 1); /* :79 */
  return fm;
}
void ora_fm_destroy(ora_fm *fm) {
  if (!fm) return;
  ora_firiq_destroy(fm->fmfilter);
  ora_rs_destroy(fm->rs_mono); ora_rs_destroy(fm->rs_stereo);
  ora_firaudio_destroy(fm->pilotcut_mono); ora_firaudio_destroy(fm->pilotcut_stereo);
  ora_pll_destroy(fm->pll); ora_mpf_destroy(fm->mpf);
  free(fm->b_filt); free(fm->b_agc); free(fm->b_mpf); free(fm->b_dec);
  free(fm->b_base); free(fm->b_raw); free(fm->b_mono1); free(fm->b_st1);
  free(fm->b_mono); free(fm->b_st);
  free(fm);
}
static void fm_reserve(ora_fm *fm, int n) {
  if (n <= fm->cap) return;
  fm->cap = n;
  fm->b_filt = (float *)realloc(fm->b_filt, sizeof(float) * 2 * n);
  fm->b_agc = (float *)realloc(fm->b_agc, sizeof(float) * 2 * n);
  fm->b_mpf = (float *)realloc(fm->b_mpf, sizeof(float) * 2 * n);
  fm->b_dec = (float *)realloc(fm->b_dec, sizeof(float) * n);
  fm->b_base = (double *)realloc(fm->b_base, sizeof(double) * n);
  fm->b_raw = (double *)realloc(fm->b_raw, sizeof(double) * n);
  fm->b_mono1 = (double *)realloc(fm->b_mono1, sizeof(double) * (n + 16));
  fm->b_st1 = (double *)realloc(fm->b_st1, sizeof(double) * (n + 16));
  fm->b_mono = (double *)realloc(fm->b_mono, sizeof(double) * (n + 16));
  fm->b_st = (double *)realloc(fm->b_st, sizeof(double) * (n + 16));
}

int ora_fm_process(ora_fm *fm, const float *iq, int n, double *audio, int cap) { /* :85-221 */
  fm->last_n = 0; fm->last_n_audio = 0;
  if (n == 0) return 0;                                     /* :89-92 */
  fm_reserve(fm, n);
  fm->if_rms = ora_rms_level(iq, n);                        /* :95 */
  const float *filt = iq;
  if (fm->fmfilter_enable) {                                /* :98-102 */
    ora_firiq_process(fm->fmfilter, iq, n, fm->b_filt);
    filt = fm->b_filt;
  }
  ora_ifagc_process(&fm->ifagc, filt, n, fm->b_agc);        /* :105 */
  const float *mp = fm->b_agc;
  if (fm->wait_multipath_blocks > 0) {                      /* :107-110, hazard H3 */
    fm->wait_multipath_blocks--;
  } else if (fm->enable_multipath) {
    int ok = ora_mpf_process(fm->mpf, fm->b_agc, n, fm->b_mpf); /* :114 */
    if (!ok) ora_mpf_initialize_coefficients(fm->mpf);      /* :117-123 */
    else mp = fm->b_mpf;
  }
  ora_disc_process(&fm->disc, mp, n, fm->b_dec);            /* :131 */
  fm->last_n = n;
  for (int i = 0; i < n; i++) fm->b_base[i] = (double)fm->b_dec[i]; /* :143, V6 */
  float bmean, brms;
  ora_mean_rms(fm->b_dec, n, &bmean, &brms);                /* :147 */
  fm->baseband_mean = (float)(0.95 * fm->baseband_mean + 0.05 * bmean);  /* :149-150 */
  fm->baseband_level = (float)(0.95 * fm->baseband_level + 0.05 * brms);
  int n_st = 0;
  if (fm->stereo_enabled) {
    ora_pll_process(fm->pll, fm->b_base, n, fm->b_raw, fm->pilot_shift); /* :157 */
    fm->stereo_detected = ora_pll_locked(fm->pll);          /* :162, hazard H4 */
    for (int i = 0; i < n; i++) {                           /* demod_stereo :224-239 */
      double v = fm->b_raw[i] * fm->b_base[i];              /* V7 */
      fm->b_raw[i] = v * 2.0;                               /* adjust_gain */
    }
    if (!fm->pilot_shift)                                   /* :168-170 */
      for (int i = 0; i < n; i++) fm->b_raw[i] = ora_iir1_step(&fm->de_stereo, fm->b_raw[i]);
    n_st = ora_rs_process(fm->rs_stereo, fm->b_raw, n, fm->b_st1, n + 16); /* :176 */
  }
  for (int i = 0; i < n; i++) fm->b_base[i] = ora_iir1_step(&fm->de_mono, fm->b_base[i]); /* :180 */
  int n_mono = ora_rs_process(fm->rs_mono, fm->b_base, n, fm->b_mono1, n + 16); /* :183 */
  if (n_mono == 0) return 0;                                /* :185-188 */
  ora_firaudio_process(fm->pilotcut_mono, fm->b_mono1, n_mono, fm->b_mono); /* :190 */
  for (int i = 0; i < n_mono; i++) fm->b_mono[i] = ora_biquad_step(&fm->dc_mono, fm->b_mono[i]); /* :192 */
  fm->last_n_audio = n_mono;
  if (fm->stereo_enabled) {
    (void)n_st;
    ora_firaudio_process(fm->pilotcut_stereo, fm->b_st1, n_mono, fm->b_st); /* :196 */
    for (int i = 0; i < n_mono; i++) fm->b_st[i] = ora_biquad_step(&fm->dc_stereo, fm->b_st[i]);
    if (2 * n_mono > cap) return -1;
    if (fm->stereo_detected) {
      if (fm->pilot_shift) {                                /* mono_to_left_right(stereo) */
        for (int i = 0; i < n_mono; i++) { audio[2 * i] = fm->b_st[i]; audio[2 * i + 1] = fm->b_st[i]; }
      } else {                                              /* :255-270 */
        for (int i = 0; i < n_mono; i++) {
          double m = fm->b_mono[i];
          double s = 1.017 * fm->b_st[i];
          audio[2 * i] = m + s; audio[2 * i + 1] = m - s;
        }
      }
    } else {
      if (fm->pilot_shift) {                                /* zero_to_left_right */
        for (int i = 0; i < n_mono; i++) { audio[2 * i] = 0.0; audio[2 * i + 1] = 0.0; }
      } else {
        for (int i = 0; i < n_mono; i++) { audio[2 * i] = fm->b_mono[i]; audio[2 * i + 1] = fm->b_mono[i]; }
      }
    }
    return 2 * n_mono;
  }
  if (n_mono > cap) return -1;
  memcpy(audio, fm->b_mono, sizeof(double) * n_mono);       /* :219 */
  return n_mono;
}
int ora_fm_stereo_detected(const ora_fm *fm) { return fm->stereo_detected; }
float ora_fm_tuning_offset(const ora_fm *fm) { return (float)(fm->baseband_mean * FM_FREQ_DEV); }
float ora_fm_baseband_level(const ora_fm *fm) { return fm->baseband_level; }
double ora_fm_pilot_level(const ora_fm *fm) { return ora_pll_pilot_level(fm->pll); }
float ora_fm_if_rms(const ora_fm *fm) { return fm->if_rms; }
double ora_fm_multipath_error(const ora_fm *fm) { return ora_mpf_error(fm->mpf); }
float ora_fm_if_agc_gain(const ora_fm *fm) { return fm->ifagc.current_gain; }
int ora_fm_pps_events(const ora_fm *fm, ora_pps_event *ev, int cap) { return ora_pll_pps_events(fm->pll, ev, cap); }
const float *ora_fm_multipath_coeff(const ora_fm *fm, int *order) {
  if (order) *order = ora_mpf_order(fm->mpf);
  return ora_mpf_coeff(fm->mpf);
}
int ora_fm_debug_vector(const ora_fm *fm, int which, double *out, int cap) {
  int n = fm->last_n;
  if (n > cap) n = cap;
  switch (which) {
  case 0: for (int i = 0; i < n; i++) out[i] = fm->b_dec[i]; return n;
  case 1: for (int i = 0; i < n; i++) out[i] = fm->b_raw[i]; return n;
  case 2: for (int i = 0; i < n; i++) out[i] = fm->b_base[i]; return n;
  }
  return -1;
}

/* ======================================================================== */
/* AmDecoder (AM, DSB) -- sfmbase/AmDecode.cpp:25-234                         */
/* ======================================================================== */
/* FineTuner (sfmbase/FineTuner.cpp:25-73): table-driven complex mixer, phase-continuous across calls */
void ora_finetuner_init(ora_finetuner *ft, unsigned table_size, int freq_shift) {   /* :25-52 */
  ft->index = 0;
  ft->size = table_size;
  ft->tab = (float *)malloc(sizeof(float) * 2 * table_size);
  const double phase_offset = fmod(0.0, 2.0 * M_PI);      /* m_phase_table[0] == 0 at construction */
  const double phase_step = 2.0 * M_PI / (double)table_size;
  for (unsigned i = 0; i < table_size; i++) {
    const double phi = (double)(((int64_t)freq_shift * (int64_t)i) % (int64_t)table_size) * phase_step + phase_offset;
    ft->tab[2 * i] = (float)cos(phi);                      /* IQSample(pcos, psin): double -> float */
    ft->tab[2 * i + 1] = (float)sin(phi);
  }
}
void ora_finetuner_free(ora_finetuner *ft) { free(ft->tab); ft->tab = NULL; }
void ora_finetuner_process(ora_finetuner *ft, const float *iq, int n, float *out) { /* :55-73 */
  unsigned idx = ft->index;
  for (int i = 0; i < n; i++) {
    const float a = iq[2 * i], b = iq[2 * i + 1], c = ft->tab[2 * idx], d = ft->tab[2 * idx + 1];
    out[2 * i] = a * c - b * d;                            /* std::complex<float> operator* */
    out[2 * i + 1] = a * d + b * c;
    if (++idx == ft->size) idx = 0;
  }
  ft->index = idx;
}

struct ora_am {
  int mode;
  float baseband_mean, baseband_level, if_rms;
  ora_firiq *cwfilter, *ssbfilter;
  ora_finetuner cw_ft, up_ft, down_ft;
  float *b1a, *b1b;
  ora_firiq *amfilter;
  ora_biquad dcblock;
  ora_iir1 deemph;
  ora_afagc afagc;
  ora_ifagc ifagc;
  int cap;
  float *b2, *b3, *dec;
  double *demod;
};
ora_am *ora_am_create2(const float *coeff, int n_coeff, int mode, const float *cwcoeff, int n_cw,
                       const float *ssbcoeff, int n_ssb) {
  ora_am *am = (ora_am *)calloc(1, sizeof(*am));
  am->mode = mode;
  const int ssb_like = (mode == ORA_MODE_USB || mode == ORA_MODE_LSB || mode == ORA_MODE_CW || mode == ORA_MODE_WSPR);
  const int cw_like = (mode == ORA_MODE_CW || mode == ORA_MODE_WSPR);
  am->amfilter = ora_firiq_create(coeff, n_coeff, 1);       /* :32 */
  if (cwcoeff) am->cwfilter = ora_firiq_create(cwcoeff, n_cw, 1);     /* :36 jj1bdx_cw_48khz_500hz */
  if (ssbcoeff) am->ssbfilter = ora_firiq_create(ssbcoeff, n_ssb, 1); /* :40 jj1bdx_ssb_48khz_1500hz */
  ora_highpass_init(&am->dcblock, 60 / 48000.0);            /* :45 */
  ora_lowpass_rc_init(&am->deemph, 100 * 48000.0 * 1.0e-6); /* :49 */
  ora_afagc_init(&am->afagc, 1.0, 1.5, ssb_like ? 0.24 : 0.6, cw_like ? 0.00125 : 0.001);   /* :54-66 */
  ora_ifagc_init(&am->ifagc, 1.0f, 1000000.0f, cw_like ? 0.0006f : 0.0003f);                /* :71-77 */
  ora_finetuner_init(&am->cw_ft, 480, 5);                   /* :83  48000/100, 500/100 */
  ora_finetuner_init(&am->up_ft, 480, 15);                  /* :89  +1500 Hz */
  ora_finetuner_init(&am->down_ft, 480, -15);               /* :90  -1500 Hz */
  return am;
}
ora_am *ora_am_create(const float *coeff, int n_coeff, int mode) {
  return ora_am_create2(coeff, n_coeff, mode, NULL, 0, NULL, 0);
}
void ora_am_destroy(ora_am *am) {
  if (!am) return;
  ora_firiq_destroy(am->amfilter);
  if (am->cwfilter) ora_firiq_destroy(am->cwfilter);
  if (am->ssbfilter) ora_firiq_destroy(am->ssbfilter);
  ora_finetuner_free(&am->cw_ft); ora_finetuner_free(&am->up_ft); ora_finetuner_free(&am->down_ft);
  free(am->b1a); free(am->b1b);
  free(am->b2); free(am->b3); free(am->dec); free(am->demod); free(am);
}
int ora_am_process(ora_am *am, const float *iq, int n, double *audio, int cap) { /* :96-218 */
  if (n > am->cap) {
    am->cap = n;
    am->b2 = (float *)realloc(am->b2, sizeof(float) * 2 * n);
    am->b3 = (float *)realloc(am->b3, sizeof(float) * 2 * n);
    am->dec = (float *)realloc(am->dec, sizeof(float) * n);
    am->demod = (double *)realloc(am->demod, sizeof(double) * n);
    am->b1a = (float *)realloc(am->b1a, sizeof(float) * 2 * n);
    am->b1b = (float *)realloc(am->b1b, sizeof(float) * 2 * n);
  }
  int n2;
  switch (am->mode) {                                       /* :96-151 */
  case ORA_MODE_USB:                                        /* shift down 1500 Hz, SSB filter, shift up (:107-114) */
    ora_finetuner_process(&am->down_ft, iq, n, am->b1a);
    n2 = ora_firiq_process(am->ssbfilter, am->b1a, n, am->b1b);
    ora_finetuner_process(&am->up_ft, am->b1b, n2, am->b2);
    break;
  case ORA_MODE_LSB:                                        /* :115-122 */
    ora_finetuner_process(&am->up_ft, iq, n, am->b1a);
    n2 = ora_firiq_process(am->ssbfilter, am->b1a, n, am->b1b);
    ora_finetuner_process(&am->down_ft, am->b1b, n2, am->b2);
    break;
  case ORA_MODE_CW:                                         /* CW LPF, then up to a 500 Hz pitch (:123-128) */
    n2 = ora_firiq_process(am->cwfilter, iq, n, am->b1a);
    ora_finetuner_process(&am->cw_ft, am->b1a, n2, am->b2);
    break;
  case ORA_MODE_WSPR:                                       /* :129-136 */
    ora_finetuner_process(&am->down_ft, iq, n, am->b1a);
    n2 = ora_firiq_process(am->cwfilter, am->b1a, n, am->b1b);
    ora_finetuner_process(&am->up_ft, am->b1b, n2, am->b2);
    break;
  default:
    n2 = ora_firiq_process(am->amfilter, iq, n, am->b2);    /* :101 */
    break;
  }
  am->if_rms = ora_rms_level(am->b2, n2);                   /* :154 */
  ora_ifagc_process(&am->ifagc, am->b2, n2, am->b3);        /* :157 */
  if (am->mode == ORA_MODE_AM) {
    for (int i = 0; i < n2; i++)                            /* :221-226, V11 */
      am->dec[i] = sqrtf(am->b3[2 * i] * am->b3[2 * i] + am->b3[2 * i + 1] * am->b3[2 * i + 1]);
  } else {
    for (int i = 0; i < n2; i++) am->dec[i] = am->b3[2 * i]; /* :229-234, V12 */
  }
  if (n2 == 0) return 0;                                    /* :181-185 */
  if (n2 > cap) return -1;
  for (int i = 0; i < n2; i++) am->demod[i] = (double)am->dec[i]; /* :190 */
  for (int i = 0; i < n2; i++) am->demod[i] = ora_biquad_step(&am->dcblock, am->demod[i]); /* :194 */
  ora_afagc_process(&am->afagc, am->demod, n2, audio);      /* :203 */
  float bmean, brms;
  ora_mean_rms(am->dec, n2, &bmean, &brms);                 /* :206-209 */
  am->baseband_mean = (float)(0.95 * am->baseband_mean + 0.05 * bmean);
  am->baseband_level = (float)(0.95 * am->baseband_level + 0.05 * brms);
  if (am->mode == ORA_MODE_AM)                              /* :212-214 */
    for (int i = 0; i < n2; i++) audio[i] = ora_iir1_step(&am->deemph, audio[i]);
  return n2;
}
double ora_am_baseband_level(const ora_am *am) { return am->baseband_level; }
float ora_am_af_agc_gain(const ora_am *am) { return (float)am->afagc.current_gain; }
float ora_am_if_agc_gain(const ora_am *am) { return am->ifagc.current_gain; }
float ora_am_if_rms(const ora_am *am) { return am->if_rms; }

/* ---------------------------------------------------------------------------
 * NbfmDecoder (sfmbase/NbfmDecode.cpp:24-96; include/NbfmDecode.h:30-95).
 * audiocoeff = the jj1bdx_48khz_nbfmaudio table (FilterParameters.cpp), passed in as data.
 * ------------------------------------------------------------------------- */
struct ora_nbfm {
  double freq_dev;
  float baseband_mean, baseband_level, if_rms;
  ora_firiq *nbfmfilter;
  ora_disc disc;
  ora_firaudio *audiofilter;
  ora_ifagc ifagc;
  int cap;
  float *b2, *b3, *dec;
  double *base;
};
ora_nbfm *ora_nbfm_create(const float *coeff, int n_coeff, double freq_dev, const double *audiocoeff, int n_audio) {
  ora_nbfm *nb = (ora_nbfm *)calloc(1, sizeof(*nb));
  nb->freq_dev = freq_dev;
  nb->nbfmfilter = ora_firiq_create(coeff, n_coeff, 1);      /* NbfmDecode.cpp:31 */
  ora_disc_init(&nb->disc, freq_dev / 48000.0);              /* :35 */
  nb->audiofilter = ora_firaudio_create(audiocoeff, n_audio); /* :39 */
  ora_ifagc_init(&nb->ifagc, 1.0f, 100000.0f, 0.0001f);      /* :43 */
  return nb;
}
void ora_nbfm_destroy(ora_nbfm *nb) {
  if (!nb) return;
  ora_firiq_destroy(nb->nbfmfilter);
  ora_firaudio_destroy(nb->audiofilter);
  free(nb->b2); free(nb->b3); free(nb->dec); free(nb->base); free(nb);
}
int ora_nbfm_process(ora_nbfm *nb, const float *iq, int n, double *audio, int cap) { /* :47-96 */
  if (n > nb->cap) {
    nb->cap = n;
    nb->b2 = (float *)realloc(nb->b2, sizeof(float) * 2 * n);
    nb->b3 = (float *)realloc(nb->b3, sizeof(float) * 2 * n);
    nb->dec = (float *)realloc(nb->dec, sizeof(float) * n);
    nb->base = (double *)realloc(nb->base, sizeof(double) * n);
  }
  const int n2 = ora_firiq_process(nb->nbfmfilter, iq, n, nb->b2);   /* :51 */
  nb->if_rms = ora_rms_level(nb->b2, n2);                            /* :54 */
  ora_ifagc_process(&nb->ifagc, nb->b2, n2, nb->b3);                 /* :57 */
  ora_disc_process(&nb->disc, nb->b3, n2, nb->dec);                  /* :60 */
  if (n2 == 0) return 0;                                             /* :64-67 */
  if (n2 > cap) return -1;
  for (int i = 0; i < n2; i++) nb->base[i] = (double)nb->dec[i];     /* :70-72 */
  float bmean, brms;
  ora_mean_rms(nb->dec, n2, &bmean, &brms);                          /* :82-85 */
  nb->baseband_mean = (float)(0.95 * nb->baseband_mean + 0.05 * bmean);
  nb->baseband_level = (float)(0.95 * nb->baseband_level + 0.05 * brms);
  const int n3 = ora_firaudio_process(nb->audiofilter, nb->base, n2, audio);   /* :88 */
  const double audio_gain = pow(10.0, (-3.0 / 20.0));                /* :91 */
  for (int i = 0; i < n3; i++) audio[i] = audio[i] * audio_gain;     /* Utility.h:307-312 */
  return n3;
}
float ora_nbfm_tuning_offset(const ora_nbfm *nb) { return (float)(nb->baseband_mean * nb->freq_dev); }  /* NbfmDecode.h:62 */
float ora_nbfm_baseband_level(const ora_nbfm *nb) { return nb->baseband_level; }
float ora_nbfm_if_rms(const ora_nbfm *nb) { return nb->if_rms; }
float ora_nbfm_if_agc_gain(const ora_nbfm *nb) { return nb->ifagc.current_gain; }

/* ---------------------------------------------------------------------------
 * Source sample formats -> IQSample (complex float).
 *   fmt 1 S16_LE, 3 S8, 0 FLOAT: what sf_read_float() delivers for the FileSource formats
 *     (sfmbase/FileSource.cpp:120-128 format names, :491-531 get_sf_read_float).  libsndfile is a third-party
 *     dependency absent from the reference tree; its documented normalisation for integer PCM read as float is
 *     value / 2^(bits-1) (pcm.c: s2f_array normfact 1/0x8000, sc2f_array 1/0x80, uc2f_array (x-128)/0x80).
 *   fmt 2 U8: RTL-SDR offset binary, (b - 128) / 128 (sfmbase/RtlSdrSource.cpp:359-365) -- the same value as
 *     libsndfile's unsigned 8-bit read.
 * ------------------------------------------------------------------------- */
int ora_iq_convert(int fmt, const void *raw, int n, float *out_iq) {
  switch (fmt) {
  case 0: memcpy(out_iq, raw, sizeof(float) * 2 * (size_t)n); return 0;
  case 1: { const int16_t *p = (const int16_t *)raw; for (int i = 0; i < 2 * n; i++) out_iq[i] = (float)p[i] / 32768.0f; return 0; }
  case 2: { const uint8_t *p = (const uint8_t *)raw;
            for (int i = 0; i < 2 * n; i++) { int32_t v = (int32_t)p[i] - 128; out_iq[i] = v / 128.0f; } return 0; }
  case 3: { const int8_t *p = (const int8_t *)raw; for (int i = 0; i < 2 * n; i++) out_iq[i] = (float)p[i] / 128.0f; return 0; }
  default: return -1;
  }
}

