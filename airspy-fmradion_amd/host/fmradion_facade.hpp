// fmradion_facade.hpp -- C++ facade above the C-ABI (include/fmradion_amd.h) with
// the reference's class names, constructor arguments and member signatures, so
// that the reference's stream loop (main.cpp:879-1002) compiles against it
// unchanged:
//
//   FourthConverterIQ   include/FourthConverterIQ.h:30-82
//   IfResampler         include/IfResampler.h:28-44
//   FmDecoder           include/FmDecode.h:35-164
//   AmDecoder           include/AmDecode.h:33-103
//   NbfmDecoder         include/NbfmDecode.h:30-95
//   FilterParameters    include/FilterParameters.h:29-53
//
// Header-only; link with libfmradion_amd.so.  Every process() call is one
// fmr_process()/fmr_resample() on a one-stream chain (host buffers in, host
// buffers out); the batched device-resident entry points of the C-ABI are what
// bench.py and multi-stream users call directly.
//
// Differences a maintainer has to know (all stated in DESIGN.md):
//  * FourthConverterIQ + IfResampler + decoder can be fused into ONE chain
//    (FmDecoder::attach_front_end) so that the IF samples never leave HBM; used
//    separately they behave like the reference classes (IF samples round-trip
//    through host vectors, as in main.cpp).
//  * IfResampler / AudioResampler arithmetic is this project's resampler
//    specification, not r8brain's (absent from the reference tree).
//  * Errors: the reference classes cannot fail and throw nothing.  A failing HIP call here is fatal in the way
//    main.cpp treats its own fatal errors (message on stderr, exit(1), main.cpp:764-767); define
//    FMR_FACADE_THROW before including this header to get std::runtime_error instead (the tests do).
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fmradion_amd.h"

using IQSample = std::complex<float>;
using IQSampleVector = std::vector<IQSample>;
using IQSampleDecodedVector = std::vector<float>;
using Sample = double;
using SampleVector = std::vector<Sample>;
using IQSampleCoeff = std::vector<IQSample::value_type>;
using SampleCoeff = std::vector<SampleVector::value_type>;
enum class ModType { FM, NBFM, AM, DSB, USB, LSB, CW, WSPR };   // include/SoftFM.h:49

namespace fmr_detail {
inline void check(int rc, const char *what) {
  if (rc == FMR_OK) return;
#ifdef FMR_FACADE_THROW
  throw std::runtime_error(std::string(what) + ": " + fmr_last_error());
#else
  std::fprintf(stderr, "ERROR: %s: %s\n", what, fmr_last_error());
  std::exit(1);
#endif
}
inline void fail(const char *what) {
#ifdef FMR_FACADE_THROW
  throw std::runtime_error(what);
#else
  std::fprintf(stderr, "ERROR: %s\n", what);
  std::exit(1);
#endif
}
inline fmr_chain *make(const fmr_config &cfg0) {
  fmr_config cfg = cfg0;
  cfg.struct_size = sizeof(fmr_config);      // the header this translation unit was built against
  cfg.in_order = 1;                          // every facade call goes through host buffers and synchronises: nothing for the
                                             // pipelined chain to overlap (include/fmradion_amd.h)
  fmr_chain *c = nullptr;
  check(fmr_create(&cfg, &c), "fmr_create");
  return c;
}
}  // namespace fmr_detail

// FilterParameters (include/FilterParameters.h:31-49): tables served by the library.
struct FilterParameters {
  static IQSampleCoeff iq(const char *name) {
    const void *p = nullptr;
    int dbl = 0;
    const int n = fmr_filter_table(name, &p, &dbl);
    if (n < 0 || dbl) { fmr_detail::fail("unknown IQ filter table"); return {}; }
    const float *f = static_cast<const float *>(p);
    return IQSampleCoeff(f, f + n);
  }
  static SampleCoeff audio(const char *name) {
    const void *p = nullptr;
    int dbl = 0;
    const int n = fmr_filter_table(name, &p, &dbl);
    if (n < 0 || !dbl) { fmr_detail::fail("unknown audio filter table"); return {}; }
    const double *f = static_cast<const double *>(p);
    return SampleCoeff(f, f + n);
  }
  static inline const IQSampleCoeff delay_3taps_only_iq = {0.0f, 1.0f, 0.0f};
  static inline const IQSampleCoeff jj1bdx_fm_384kHz_narrow = iq("jj1bdx_fm_384kHz_narrow");
  static inline const IQSampleCoeff jj1bdx_fm_384kHz_medium = iq("jj1bdx_fm_384kHz_medium");
  static inline const IQSampleCoeff jj1bdx_am_48khz_narrow = iq("jj1bdx_am_48khz_narrow");
  static inline const IQSampleCoeff jj1bdx_am_48khz_medium = iq("jj1bdx_am_48khz_medium");
  static inline const IQSampleCoeff jj1bdx_am_48khz_default = iq("jj1bdx_am_48khz_default");
  static inline const IQSampleCoeff jj1bdx_am_48khz_wide = iq("jj1bdx_am_48khz_wide");
  static inline const IQSampleCoeff jj1bdx_nbfm_48khz_default = iq("jj1bdx_nbfm_48khz_default");
  static inline const IQSampleCoeff jj1bdx_nbfm_48khz_narrow = iq("jj1bdx_nbfm_48khz_narrow");
  static inline const IQSampleCoeff jj1bdx_nbfm_48khz_medium = iq("jj1bdx_nbfm_48khz_medium");
  static inline const IQSampleCoeff jj1bdx_nbfm_48khz_wide = iq("jj1bdx_nbfm_48khz_wide");
  static inline const SampleCoeff jj1bdx_48khz_fmaudio = audio("jj1bdx_48khz_fmaudio");
  static inline const SampleCoeff jj1bdx_48khz_nbfmaudio = audio("jj1bdx_48khz_nbfmaudio");
};

// FourthConverterIQ (FourthConverterIQ.h:30-82): Fs/4 shift, exact.  Stand-alone use costs a PCIe round trip per
// block; a decoder with attach_front_end(rate, true) applies the same shift inside its front-end kernel.
class FourthConverterIQ {
public:
  explicit FourthConverterIQ(bool up, int device = 0) : m_up(up) {
    fmr_config cfg{};
    cfg.device = device; cfg.n_streams = 1; cfg.mode = -1; cfg.input_rate = 384000.0;
    cfg.max_block_len = 65536; cfg.max_blocks = 1;
    m_chain = fmr_detail::make(cfg);
  }
  ~FourthConverterIQ() { fmr_destroy(m_chain); }
  FourthConverterIQ(const FourthConverterIQ &) = delete;
  FourthConverterIQ &operator=(const FourthConverterIQ &) = delete;
  void process(const IQSampleVector &samples_in, IQSampleVector &samples_out) {
    samples_out.resize(samples_in.size());
    fmr_detail::check(fmr_fourth_convert(m_chain, reinterpret_cast<const float *>(samples_in.data()), samples_in.size(),
                                         reinterpret_cast<float *>(samples_out.data()), m_up ? 1 : 0, &m_index),
                      "fmr_fourth_convert");
  }

private:
  fmr_chain *m_chain = nullptr;
  unsigned m_index = 0;          // FourthConverterIQ.h:31
  const bool m_up;
};

// IfResampler::process(const IQSampleVector&, IQSampleVector&)  (IfResampler.h:35-38), any pair of integer rates the
// resampler design covers (384 kHz for FM, 48 kHz for the AM / NBFM decoders, main.cpp:775-777).  Equal rates: a copy
// (main.cpp does not call process() then, :778,925-929).
class IfResampler {
public:
  static constexpr int max_input_length = 65536;   // IfResampler.h:31
  // resampler_class: FMR_RESAMPLER_R8B (default) -- the defaults of the r8b::CDSPResampler24 this class wraps in the
  // reference (IfResampler.cpp:25-29: 2 % transition band ending at Nyquist, 180 dB), so that the decoder behind it is
  // fed what the reference's filter delivers; FMR_RESAMPLER_FAST -- the throughput specification of the benchmark (a
  // narrower pass band at 140 dB: 5.5e-6 from R8B in the audio on a clean band, not for a crowded one)
  IfResampler(const double input_rate, const double output_rate, int device = 0, int resampler_class = FMR_RESAMPLER_R8B) {
    if (input_rate == output_rate) return;
    fmr_config cfg{};
    cfg.device = device; cfg.n_streams = 1; cfg.mode = -1; cfg.input_rate = input_rate; cfg.output_rate = output_rate;
    cfg.enable_resampler = 1; cfg.max_block_len = max_input_length; cfg.max_blocks = 1; cfg.resampler_class = resampler_class;
    m_chain = fmr_detail::make(cfg);
  }
  ~IfResampler() { if (m_chain) fmr_destroy(m_chain); }
  IfResampler(const IfResampler &) = delete;
  IfResampler &operator=(const IfResampler &) = delete;
  void process(const IQSampleVector &samples_in, IQSampleVector &samples_out) {
    if (!m_chain) { samples_out = samples_in; return; }
    samples_out.resize(samples_in.size() + 64);
    size_t n = 0;
    fmr_detail::check(fmr_resample(m_chain, reinterpret_cast<const float *>(samples_in.data()), samples_in.size(),
                                   reinterpret_cast<float *>(samples_out.data()), samples_out.size(), &n),
                      "fmr_resample");
    samples_out.resize(n);
  }

private:
  fmr_chain *m_chain = nullptr;
};

// PilotPhaseLock::PpsEvent (PilotPhaseLock.h:40-44)
struct PilotPhaseLock {
  struct PpsEvent {
    std::uint64_t pps_index;
    std::uint64_t sample_index;
    double block_position;
  };
};

// FmDecoder (FmDecode.h:63-105)
class FmDecoder {
public:
  static constexpr double sample_rate_if = 384000;
  static constexpr double sample_rate_pcm = 48000;
  static constexpr double freq_dev = 75000;
  static constexpr double deemphasis_time_eu = 50;
  static constexpr double deemphasis_time_na = 75;

  FmDecoder(bool fmfilter_enable, IQSampleCoeff &fmfilter_coeff, bool stereo, double deemphasis, bool pilot_shift,
            unsigned int multipath_stages, int device = 0)
      : m_stereo(stereo) {
    m_cfg = fmr_config{};
    m_cfg.device = device; m_cfg.n_streams = 1; m_cfg.mode = FMR_MODE_FM; m_cfg.input_rate = sample_rate_if;
    m_cfg.fmfilter_enable = fmfilter_enable; m_cfg.filter_coeff = fmfilter_coeff.data();
    m_cfg.n_filter_coeff = (int)fmfilter_coeff.size(); m_cfg.stereo = stereo; m_cfg.deemphasis_us = deemphasis;
    m_cfg.pilot_shift = pilot_shift; m_cfg.multipath_stages = multipath_stages;
    m_cfg.max_block_len = 65536; m_cfg.max_blocks = 1;
    m_chain = fmr_detail::make(m_cfg);
  }
  ~FmDecoder() { fmr_destroy(m_chain); }
  FmDecoder(const FmDecoder &) = delete;
  FmDecoder &operator=(const FmDecoder &) = delete;

  // Fuse FourthConverterIQ + IfResampler into this decoder's chain: process() then takes
  // the source-rate IQ block (what main.cpp:889 pulls) and the IF never leaves the GPU.
  void attach_front_end(double input_rate, bool fourth_down, int resampler_class = FMR_RESAMPLER_R8B) {
    fmr_destroy(m_chain);
    m_cfg.input_rate = input_rate; m_cfg.enable_resampler = 1; m_cfg.enable_fourth_down = fourth_down;
    m_cfg.resampler_class = resampler_class;
    m_chain = fmr_detail::make(m_cfg);
  }

  // Latency for throughput: hold back `blocks` - 1 calls and decode `blocks` blocks in ONE batched call.  process()
  // then returns an empty vector ("nothing yet": the contract of FmDecode.cpp:89-92,185-188, which main.cpp:981-984
  // already handles) until the batch is full, and the audio of all its blocks at once.  One 65536-sample block per
  // call costs ~0.3 ms, almost all of it launch overhead (~85 kernel launches); a batch costs about the same.
  // Default 1 = the reference's call-by-call behaviour.
  //  * END OF STREAM: call flush() -- it decodes the blocks still held back (fewer than a batch).  Without it up to
  //    blocks - 1 source blocks of audio would be lost; the reference decodes every block.
  //  * The status getters (get_if_rms(), get_pps_events(), ...) describe the LAST block of the batch just decoded, the
  //    PPS events all of its blocks.
  //  * Raising the batch above the chain's capacity re-creates the chain and is therefore only possible before the first
  //    process(); lowering it (or raising it again up to the capacity) is possible at any time and keeps held-back blocks.
  void set_batch_blocks(unsigned blocks) {
    if (blocks < 1) blocks = 1;
    if (blocks == m_batch) return;
    if (blocks > m_capacity) {
      if (m_started) fmr_detail::fail("FmDecoder::set_batch_blocks: a larger batch than the chain was created for, after the first process()");
      fmr_destroy(m_chain);
      m_cfg.max_blocks = (int)blocks;
      m_chain = fmr_detail::make(m_cfg);
      m_capacity = blocks;
    }
    m_batch = blocks;
  }

  // samples_in by value, audio resized by the callee, empty = "nothing yet" (FmDecode.cpp:85-92)
  void process(IQSampleVector samples_in, SampleVector &audio) {
    m_pps_fetched = false;       // PilotPhaseLock::process clears m_pps_events on every call (PilotPhaseLock.cpp:62)
    m_pps.clear();
    m_started = true;
    if (m_batch > 1 || !m_pending_len.empty()) {
      m_pending.insert(m_pending.end(), samples_in.begin(), samples_in.end());
      m_pending_len.push_back((uint32_t)samples_in.size());
      audio.clear();
      if (m_pending_len.size() < m_batch) { m_pps_fetched = true; return; }
      decode_pending(audio);
      return;
    }
    audio.resize(2 * (samples_in.size() + 64));
    size_t n = 0;
    fmr_detail::check(fmr_process(m_chain, reinterpret_cast<const float *>(samples_in.data()), samples_in.size(),
                                  audio.data(), audio.size(), &n),
                      "fmr_process");
    audio.resize(n);
  }
  // Decode whatever set_batch_blocks() is still holding back (end of stream, or before a batch-size change that must
  // not add latency).  audio is empty if nothing was pending.
  void flush(SampleVector &audio) {
    m_pps_fetched = false;
    m_pps.clear();
    audio.clear();
    if (m_pending_len.empty()) { m_pps_fetched = true; return; }
    decode_pending(audio);
  }
  size_t pending_blocks() const { return m_pending_len.size(); }
  bool stereo_detected() { return status().stereo_detected != 0; }
  float get_tuning_offset() { return status().baseband_mean * freq_dev; }
  float get_baseband_level() { return status().baseband_level; }
  double get_pilot_level() { return status().pilot_level; }
  float get_if_rms() { return status().if_rms; }
  double get_multipath_error() { return status().multipath_error; }
  // Events of the most recent process() call; erase_first_pps_event() consumes them one by one (main.cpp:1087-1094).
  std::vector<PilotPhaseLock::PpsEvent> get_pps_events() {
    fetch_pps();
    return m_pps;
  }
  void erase_first_pps_event() {
    fetch_pps();
    if (!m_pps.empty()) m_pps.erase(m_pps.begin());
  }
  const std::vector<std::complex<float>> &get_multipath_coefficients() {
    m_coeff.resize(1300);
    const int n = fmr_get_multipath_coefficients(m_chain, 0, reinterpret_cast<float *>(m_coeff.data()), 2600);
    m_coeff.resize(n > 0 ? n : 0);
    return m_coeff;
  }

private:
  fmr_status status() {
    fmr_status st{};
    fmr_detail::check(fmr_get_status(m_chain, 0, &st), "fmr_get_status");
    return st;
  }
  // the held-back blocks in calls of at most the chain's capacity (fmr_process_blocks takes any count up to max_blocks)
  void decode_pending(SampleVector &audio) {
    audio.resize(2 * (m_pending.size() + 64 * m_pending_len.size()));
    size_t done_blocks = 0, done_samples = 0, n_audio = 0;
    while (done_blocks < m_pending_len.size()) {
      const size_t nb = std::min<size_t>(m_capacity, m_pending_len.size() - done_blocks);
      size_t ns = 0;
      for (size_t b = 0; b < nb; b++) ns += m_pending_len[done_blocks + b];
      std::vector<uint32_t> alen(nb);
      fmr_detail::check(fmr_process_blocks(m_chain, reinterpret_cast<const float *>(m_pending.data() + done_samples), ns,
                                           m_pending_len.data() + done_blocks, (int)nb, audio.data() + n_audio,
                                           audio.size() - n_audio, alen.data()),
                        "fmr_process_blocks");
      for (uint32_t v : alen) n_audio += v;
      done_blocks += nb; done_samples += ns;
    }
    audio.resize(n_audio);
    m_pending.clear();
    m_pending_len.clear();
  }
  void fetch_pps() {
    if (m_pps_fetched) return;
    fmr_pps_event ev[64];
    const int n = fmr_get_pps_events(m_chain, 0, ev, 64);
    m_pps.clear();
    for (int i = 0; i < n && i < 64; i++) m_pps.push_back({ev[i].pps_index, ev[i].sample_index, ev[i].block_position});
    m_pps_fetched = true;
  }
  fmr_config m_cfg{};
  fmr_chain *m_chain = nullptr;
  bool m_stereo;
  bool m_pps_fetched = true;
  std::vector<PilotPhaseLock::PpsEvent> m_pps;
  unsigned m_batch = 1, m_capacity = 1;
  bool m_started = false;
  IQSampleVector m_pending;
  std::vector<uint32_t> m_pending_len;
  std::vector<std::complex<float>> m_coeff;
};

// AmDecoder (AmDecode.h:48-65), all of its modes: AM, DSB, USB, LSB, CW, WSPR
class AmDecoder {
public:
  static constexpr double sample_rate_pcm = 48000;
  static constexpr double internal_rate_pcm = 48000;
  AmDecoder(IQSampleCoeff &amfilter_coeff, const ModType mode, int device = 0) {
    // main.cpp:813 constructs the AmDecoder whatever the mode; with an FM mode it is never used (AmDecode.cpp:96-147
    // has no case for it): build it as an AM decoder
    m_cfg = fmr_config{};
    m_cfg.device = device; m_cfg.n_streams = 1;
    m_cfg.mode = (mode == ModType::FM || mode == ModType::NBFM) ? FMR_MODE_AM : static_cast<int>(mode);   // FMR_MODE_* follow ModType
    m_cfg.input_rate = internal_rate_pcm; m_cfg.filter_coeff = amfilter_coeff.data();
    m_cfg.n_filter_coeff = (int)amfilter_coeff.size(); m_cfg.max_block_len = 65536; m_cfg.max_blocks = 1;
    m_chain = fmr_detail::make(m_cfg);
  }
  ~AmDecoder() { fmr_destroy(m_chain); }
  AmDecoder(const AmDecoder &) = delete;
  AmDecoder &operator=(const AmDecoder &) = delete;
  void attach_front_end(double input_rate, bool fourth_down, int resampler_class = FMR_RESAMPLER_R8B) {
    fmr_destroy(m_chain);
    m_cfg.input_rate = input_rate; m_cfg.enable_resampler = 1; m_cfg.enable_fourth_down = fourth_down;
    m_cfg.resampler_class = resampler_class;
    m_chain = fmr_detail::make(m_cfg);
  }
  void process(IQSampleVector samples_in, SampleVector &audio) {
    audio.resize(samples_in.size() + 64);
    size_t n = 0;
    fmr_detail::check(fmr_process(m_chain, reinterpret_cast<const float *>(samples_in.data()), samples_in.size(),
                                  audio.data(), audio.size(), &n),
                      "fmr_process");
    audio.resize(n);
  }
  double get_baseband_level() { return status().baseband_level; }
  float get_af_agc_current_gain() { return (float)status().af_agc_gain; }
  float get_if_agc_current_gain() { return status().if_agc_gain; }
  float get_if_rms() { return status().if_rms; }

private:
  fmr_status status() {
    fmr_status st{};
    fmr_detail::check(fmr_get_status(m_chain, 0, &st), "fmr_get_status");
    return st;
  }
  fmr_config m_cfg{};
  fmr_chain *m_chain = nullptr;
};

// NbfmDecoder (NbfmDecode.h:49-66)
class NbfmDecoder {
public:
  static constexpr double sample_rate_pcm = 48000;
  static constexpr double internal_rate_pcm = 48000;
  static constexpr double freq_dev_normal = 8000;
  static constexpr double freq_dev_wide = 17000;
  NbfmDecoder(IQSampleCoeff &nbfmfilter_coeff, const double freq_dev, int device = 0) : m_freq_dev(freq_dev) {
    m_cfg = fmr_config{};
    m_cfg.device = device; m_cfg.n_streams = 1; m_cfg.mode = FMR_MODE_NBFM; m_cfg.input_rate = internal_rate_pcm;
    m_cfg.filter_coeff = nbfmfilter_coeff.data(); m_cfg.n_filter_coeff = (int)nbfmfilter_coeff.size();
    m_cfg.nbfm_freq_dev = freq_dev; m_cfg.max_block_len = 65536; m_cfg.max_blocks = 1;
    m_chain = fmr_detail::make(m_cfg);
  }
  ~NbfmDecoder() { fmr_destroy(m_chain); }
  NbfmDecoder(const NbfmDecoder &) = delete;
  NbfmDecoder &operator=(const NbfmDecoder &) = delete;
  void attach_front_end(double input_rate, bool fourth_down, int resampler_class = FMR_RESAMPLER_R8B) {
    fmr_destroy(m_chain);
    m_cfg.input_rate = input_rate; m_cfg.enable_resampler = 1; m_cfg.enable_fourth_down = fourth_down;
    m_cfg.resampler_class = resampler_class;
    m_chain = fmr_detail::make(m_cfg);
  }
  void process(const IQSampleVector &samples_in, SampleVector &audio) {
    audio.resize(samples_in.size() + 64);
    size_t n = 0;
    fmr_detail::check(fmr_process(m_chain, reinterpret_cast<const float *>(samples_in.data()), samples_in.size(),
                                  audio.data(), audio.size(), &n),
                      "fmr_process");
    audio.resize(n);
  }
  float get_tuning_offset() { return (float)(status().baseband_mean * m_freq_dev); }
  float get_baseband_level() { return status().baseband_level; }
  float get_if_rms() { return status().if_rms; }

private:
  fmr_status status() {
    fmr_status st{};
    fmr_detail::check(fmr_get_status(m_chain, 0, &st), "fmr_get_status");
    return st;
  }
  const double m_freq_dev;
  fmr_config m_cfg{};
  fmr_chain *m_chain = nullptr;
};

