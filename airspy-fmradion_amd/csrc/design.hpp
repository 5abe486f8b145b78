// design.hpp -- host-side design arithmetic of the MI355X FM/AM chain.
//
// * ResamplerDesign: the product's implementation of the resampler
//   specification of DESIGN.md ("Resampler specification"): integer
//   pre-decimation by D (Kaiser FIR, NA taps) followed by an LB/MB polyphase
//   stage (TB taps per phase).  It replaces r8b::CDSPResampler24 /
//   r8b::CDSPResampler as used at sfmbase/IfResampler.cpp:26-29 and
//   sfmbase/AudioResampler.cpp:28-29 (r8brain is absent from the reference
//   tree: the spec is ours and stated in DESIGN.md).
// * ResamplerCounter: the deterministic output-count law (how many outputs a
//   call emits), integer arithmetic only; the host uses it to lay out the
//   per-block offset tables before any kernel runs.
// * IIR coefficient formulas of sfmbase/Filter.cpp:186-188 (LowPassFilterRC)
//   and :254-290 (HighPassFilterIir).
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace fmr {

inline double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  const double q = x * x * 0.25;
  for (int k = 1; k < 1000; k++) {
    term *= q / (double(k) * double(k));
    sum += term;
    if (term < 1e-21 * sum) break;
  }
  return sum;
}

inline double sinc_pi(double x) {
  if (std::fabs(x) < 1e-12) return 1.0;
  return std::sin(M_PI * x) / (M_PI * x);
}

inline long long gcd_ll(long long a, long long b) {
  while (b) { long long t = a % b; a = b; b = t; }
  return a < 0 ? -a : a;
}

// Parks-McClellan exchange for a type-I (odd length, symmetric) multiband FIR: the equiripple stage A of the IF-class
// resampler ("Resampler specification").  Grid: band b contributes round((hi - lo) / delf) points from lo in
// steps of delf = 0.5 / (16 (M + 1)), the last one moved onto hi; barycentric Lagrange interpolation in
// x = cos(2 pi f); the exchange stops when the extremal errors agree to 1e-4 relative.  Band edges in cycles per sample.
class ParksMcClellan {
 public:
  // returns false if the exchange did not settle within 100 steps
  bool design(int N, const std::vector<double> &edges, const std::vector<double> &desired, const std::vector<double> &weight,
              std::vector<double> &h) {
    const int M = (N - 1) / 2;
    r_ = M + 1;
    const double delf = 0.5 / (16.0 * r_);
    grid_.clear(); D_.clear(); W_.clear();
    for (size_t b = 0; b < desired.size(); b++) {
      const int k = (int)((edges[2 * b + 1] - edges[2 * b]) / delf + 0.5);
      double f = edges[2 * b];
      for (int i = 0; i < k; i++) { grid_.push_back(f); D_.push_back(desired[b]); W_.push_back(weight[b]); f += delf; }
      if (k > 0) grid_.back() = edges[2 * b + 1];
    }
    const int gs = (int)grid_.size();
    if (gs < r_ + 2) return false;
    E_.assign(gs, 0.0);
    ext_.resize(r_ + 1);
    for (int i = 0; i <= r_; i++) ext_[i] = i * (gs - 1) / r_;
    x_.resize(r_ + 1); y_.resize(r_ + 1); ad_.resize(r_ + 1);
    bool settled = false;
    for (int it = 0; it < 100 && !settled; it++) {
      interpolant();
      for (int i = 0; i < gs; i++) E_[i] = W_[i] * (D_[i] - response(grid_[i]));
      exchange();
      double mn = std::fabs(E_[ext_[0]]), mx = mn;
      for (int i = 1; i <= r_; i++) { const double c = std::fabs(E_[ext_[i]]); mn = std::fmin(mn, c); mx = std::fmax(mx, c); }
      settled = (mx - mn) / mx < 0.0001;
    }
    interpolant();
    // frequency sampling: the response at i / N, i = 0 .. M, then the inverse cosine sum
    std::vector<double> A(M + 1);
    for (int i = 0; i <= M; i++) A[i] = response(double(i) / N);
    h.assign(N, 0.0);
    for (int n = 0; n <= M; n++) {
      double val = A[0];
      const double xx = 2.0 * M_PI * (n - M) / N;
      for (int k = 1; k <= M; k++) val += 2.0 * A[k] * std::cos(xx * k);
      h[n] = h[N - 1 - n] = val / N;
    }
    return settled;
  }

 private:
  int r_ = 0;
  std::vector<double> grid_, D_, W_, E_, x_, y_, ad_;
  std::vector<int> ext_;

  void interpolant() {
    for (int i = 0; i <= r_; i++) x_[i] = std::cos(2.0 * M_PI * grid_[ext_[i]]);
    const int ld = (r_ - 1) / 15 + 1;          // strided products keep the barycentric weights in range
    for (int i = 0; i <= r_; i++) {
      double denom = 1.0;
      for (int j = 0; j < ld; j++)
        for (int k = j; k <= r_; k += ld)
          if (k != i) denom *= 2.0 * (x_[i] - x_[k]);
      if (std::fabs(denom) < 1e-5) denom = 1e-5;
      ad_[i] = 1.0 / denom;
    }
    double numer = 0.0, denom = 0.0, sign = 1.0;
    for (int i = 0; i <= r_; i++) { numer += ad_[i] * D_[ext_[i]]; denom += sign * ad_[i] / W_[ext_[i]]; sign = -sign; }
    const double delta = numer / denom;
    sign = 1.0;
    for (int i = 0; i <= r_; i++) { y_[i] = D_[ext_[i]] - sign * delta / W_[ext_[i]]; sign = -sign; }
  }
  double response(double f) const {
    const double xc = std::cos(2.0 * M_PI * f);
    double numer = 0.0, denom = 0.0;
    for (int i = 0; i <= r_; i++) {
      double c = xc - x_[i];
      if (std::fabs(c) < 1.0e-7) return y_[i];
      c = ad_[i] / c;
      denom += c;
      numer += c * y_[i];
    }
    return numer / denom;
  }
  void exchange() {
    const int gs = (int)grid_.size();
    std::vector<int> f;
    const std::vector<double> &E = E_;
    if ((E[0] > 0.0 && E[0] > E[1]) || (E[0] < 0.0 && E[0] < E[1])) f.push_back(0);
    for (int i = 1; i < gs - 1; i++)
      if ((E[i] >= E[i - 1] && E[i] > E[i + 1] && E[i] > 0.0) || (E[i] <= E[i - 1] && E[i] < E[i + 1] && E[i] < 0.0)) f.push_back(i);
    if ((E[gs - 1] > 0.0 && E[gs - 1] > E[gs - 2]) || (E[gs - 1] < 0.0 && E[gs - 1] < E[gs - 2])) f.push_back(gs - 1);
    while ((int)f.size() > r_ + 1) {
      const int k = (int)f.size();
      bool up = E[f[0]] > 0.0, alternating = true;
      int l = 0;
      for (int j = 1; j < k; j++) {
        if (std::fabs(E[f[j]]) < std::fabs(E[f[l]])) l = j;
        if (up && E[f[j]] < 0.0) up = false;
        else if (!up && E[f[j]] > 0.0) up = true;
        else { alternating = false; break; }      // two neighbours of one sign: drop the smaller seen so far
      }
      if (alternating && k - (r_ + 1) == 1) l = (std::fabs(E[f[k - 1]]) < std::fabs(E[f[0]])) ? k - 1 : 0;
      f.erase(f.begin() + l);
    }
    for (int i = 0; i <= r_ && i < (int)f.size(); i++) ext_[i] = f[i];
  }
};

struct ResamplerDesign {
  double in_rate = 0, out_rate = 0, atten = 0;
  long long L = 1, M = 1;
  int D = 1, NA = 0;
  long long LB = 1, MB = 1;
  int TB = 2;
  // LT = 0: one table row per phase.  LT > 0, the fractional-phase form for ratios whose LB rows would not fit (ppm
  // corrected source rates, main.cpp:708-711): rows are the prototype at mu = p / LT, p = 0 .. LT, and the taps of
  // the output at mu = (k MB mod LB) / LB are interpolated linearly between rows floor(mu LT) and floor(mu LT) + 1.
  int LT = 0;
  std::vector<double> hA;  // [NA]
  std::vector<double> hB;  // [LB][TB], or [LT + 1][TB]

  // pass_frac: pass band edge as a fraction of out / 2; stop_nyquist: the stop band starts at out / 2 (nothing aliases at
  // all) instead of at out - f_pass.  (0.885, false) is the FAST class of DESIGN.md; (0.98, true) at 180 dB the
  // R8B class -- the defaults of r8b::CDSPResampler24 (sfmbase/IfResampler.cpp:25-29).
  bool design(double in_r, double out_r, double atten_db, double pass_frac = 0.885, bool stop_nyquist = false) {
    in_rate = in_r; out_rate = out_r; atten = atten_db;
    const double A = atten_db;
    const double beta = 0.1102 * (A - 8.7);
    const double i0b = bessel_i0(beta);
    const double fpass = pass_frac * out_rate * 0.5;
    const double fstop = stop_nyquist ? out_rate * 0.5 : out_rate - fpass;
    long long in_i = llround(in_rate), out_i = llround(out_rate);
    if (!(in_rate > 0.5) || !(out_rate > 0.5) || in_rate > 4e9 || out_rate > 4e9) return false;
    if (std::fabs(in_rate - in_i) > 1e-6 || std::fabs(out_rate - out_i) > 1e-6) {
      // rates that are not whole hertz are taken to the millihertz: the ratio stays an exact rational
      in_i = llround(in_rate * 1000.0); out_i = llround(out_rate * 1000.0);
    }
    const long long g = gcd_ll(in_i, out_i);
    L = out_i / g; M = in_i / g;
    D = (int)std::floor(in_rate / (2.6 * out_rate));
    if (D < 1) D = 1;
    const double mid = in_rate / D;
    hA.clear();
    NA = 0;
    if (D > 1) {
      const double f1 = fpass, f2 = mid - fstop;
      const double dw = 2.0 * M_PI * (f2 - f1) / in_rate;
      int N = (int)std::ceil((A - 7.95) / (2.285 * dw)) + 1;
      if ((N & 1) == 0) N++;
      NA = N;
      hA.resize(N);
      const double fc = 0.5 * (f1 + f2) / in_rate;
      const double c = 0.5 * (N - 1);
      double sum = 0;
      for (int k = 0; k < N; k++) {
        const double t = k - c, r = t / c;
        const double w = bessel_i0(beta * std::sqrt(std::fmax(0.0, 1.0 - r * r))) / i0b;
        hA[k] = 2.0 * fc * sinc_pi(2.0 * fc * t) * w;
        sum += hA[k];
      }
      for (int k = 0; k < N; k++) hA[k] /= sum;
      if (A <= 150.0 && !stop_nyquist && D <= 78) {
        // The IF class (float32 data, 140 dB): an equiripple stage A of 0.68 x the Kaiser length N.  Pass band weight 1,
        // stop bands k mid -+ fstop (k = 1 .. D / 2, cut at in / 2) weight 800, the bands between them free: what falls
        // there is removed by stage B.  D = 2 .. 20: ripple <= 0.0010 dB peak to peak, aliases of the pass band
        // <= -141 dB (D = 10: -142).  The Kaiser design stays if the exchange does not settle or its result misses the class.
        std::vector<double> edges{0.0, fpass / in_rate}, des{1.0}, wt{1.0}, he;
        for (int k = 1; k <= D / 2 && des.size() < 40; k++) {
          const double lo = (k * mid - fstop) / in_rate;
          double hi = (k * mid + fstop) / in_rate;
          if (lo >= 0.5) break;
          if (hi > 0.5) hi = 0.5;
          edges.push_back(lo); edges.push_back(hi); des.push_back(0.0); wt.push_back(800.0);
        }
        // The exchange's own stopping rule says the extremal errors agree, not how large they are: the response is
        // checked against the class (every stop band <= -140 dB, pass band ripple <= 0.0012 dB peak to peak) on 64 points
        // per band plus the edges.  A design that misses -- 0.68 N does at D = 28 (by 0.2 dB) and at D = 61 -- is tried
        // again 4 % of N longer, up to 0.96 N; the Kaiser window stays if none passes.  The oracle applies the same rule
        // on the same grid and the same lengths (rs_design_equiripple).
        for (int pc = 68; pc <= 96; pc += 4) {
          int NE = (int)((N * (long long)pc + 99) / 100);
          if ((NE & 1) == 0) NE++;
          ParksMcClellan pm;
          if (!pm.design(NE, edges, des, wt, he)) continue;
          double se = 0;
          for (double v : he) se += v;
          for (double &v : he) v /= se;
          const double ce = 0.5 * (NE - 1);
          double stop_max = 0.0, pass_lo = 1.0, pass_hi = 1.0;
          for (size_t b = 0; b < des.size(); b++)
            for (int g = 0; g <= 64; g++) {
              const double f = edges[2 * b] + (edges[2 * b + 1] - edges[2 * b]) * (double)g / 64.0;
              double re = 0.0;
              for (int k = 0; k < NE; k++) re += he[k] * std::cos(2.0 * M_PI * f * ((double)k - ce));
              if (b == 0) { pass_lo = std::fmin(pass_lo, re); pass_hi = std::fmax(pass_hi, re); }
              else stop_max = std::fmax(stop_max, std::fabs(re));
            }
          if (stop_max <= 1.0e-7 && pass_hi - pass_lo <= 1.4e-4) { hA = he; NA = NE; break; }      // -140 dB; 0.0012 dB peak to peak
        }
      }
    }
    const long long num = L * D, den = M;
    const long long g2 = gcd_ll(num, den);
    LB = num / g2; MB = den / g2;
    const double dw = 2.0 * M_PI * (fstop - fpass) / (double(LB) * mid);
    const double nproto = (A - 7.95) / (2.285 * dw) + 1.0;
    int T = (int)std::ceil(nproto / double(LB));
    if (T & 1) T++;
    if (T < 2) T = 2;
    TB = T;
    // (up to 6144 taps per phase: the R8B class at a source rate four times the IF rate -- 1.536 MS/s, no integer
    // pre-decimation -- needs 4794; the chain checks the LDS window of the kernel that would run it, fmr_chain::init)
    if (T > 6144 || LB >= (1ll << 36) || MB >= (1ll << 36)) return false;   // outside the kernels' index arithmetic
    LT = (LB * (long long)T > (1ll << 22)) ? 1024 : 0;
    const long long rows = LT ? LT + 1 : LB, prow = LT ? LT : LB;
    hB.resize(size_t(rows) * T);
    const double W = 0.5 * T;
    const double fc = 0.5 * out_rate / mid;
    double sum = 0;
    for (long long p = 0; p < rows; p++)
      for (int j = 0; j < T; j++) {
        const double t = double(p) / double(prow) + W - 1.0 - j, r = t / W;
        const double w = bessel_i0(beta * std::sqrt(std::fmax(0.0, 1.0 - r * r))) / i0b;
        const double v = 2.0 * fc * sinc_pi(2.0 * fc * t) * w;
        hB[size_t(p) * T + j] = v;
        if (p < prow) sum += v;
      }
    const double scale = double(prow) / sum;
    for (auto &v : hB) v *= scale;
    return true;
  }
  int ca() const { return NA ? (NA - 1) / 2 : 0; }
  int W() const { return TB / 2; }
};

// Output-count law.  Stage A output m exists once input D*m + ca has arrived;
// stage B output k once mid sample floor(k*MB/LB) + W has been produced.
struct ResamplerCounter {
  long long n_in = 0, mA = 0, kB = 0;
  void reset() { n_in = mA = kB = 0; }
  static long long mA_avail(const ResamplerDesign &d, long long n) {
    if (d.D == 1) return n;
    const int ca = d.ca();
    return (n >= ca + 1) ? (n - 1 - ca) / d.D + 1 : 0;
  }
  static long long kB_avail(const ResamplerDesign &d, long long mA_) {
    const int W = d.W();
    if (mA_ < W + 1) return 0;
    return (long long)(((__int128)(mA_ - W) * d.LB + d.MB - 1) / d.MB);
  }
  // advance by n inputs; returns the number of new outputs
  long long advance(const ResamplerDesign &d, long long n) {
    n_in += n;
    mA = mA_avail(d, n_in);
    const long long k = kB_avail(d, mA);
    const long long out = k - kB;
    kB = k;
    return out;
  }
};

struct Iir1Coef { double b0, b1, a1; };
struct BiquadCoef { double b0, b1, b2, a1, a2; };

// LowPassFilterRC: sfmbase/Filter.cpp:186-188
inline Iir1Coef lowpass_rc(double timeconst) {
  const double a1 = -std::exp(-1 / timeconst);
  return {1 + a1, 0.0, a1};
}

// HighPassFilterIir: sfmbase/Filter.cpp:254-290 (matched-Z 2-pole Butterworth)
inline BiquadCoef highpass_iir(double cutoff) {
  using C = std::complex<double>;
  const double w = 2 * M_PI * cutoff;
  const C p1s = w / std::exp((2 * 1 + 2 - 1) / double(2 * 2) * C(0, M_PI));
  const C p1z = std::exp(p1s);
  double b0 = 1, b1 = -2, b2 = 1;
  const double a1 = -2 * std::real(p1z);
  const double a2 = std::abs(p1z * p1z);
  const double g = (b0 - b1 + b2) / (1 - a1 + a2);
  return {b0 / g, b1 / g, b2 / g, a1, a2};
}

// fast_atan_table (include/Utility.h:165-217), regenerated: entry i is
// float("%.6e" % atan(i/255)); entry 256 repeats entry 255.
inline void make_fast_atan_table(float *tab /*257*/) {
  char buf[64];
  for (int i = 0; i < 256; i++) {
    std::snprintf(buf, sizeof buf, "%.6e", std::atan(double(i) / 255.0));
    tab[i] = (float)std::strtod(buf, nullptr);
  }
  tab[256] = tab[255];
}

}  // namespace fmr
