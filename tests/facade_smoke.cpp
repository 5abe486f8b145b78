// Compiles the C++ facade (reference class names/signatures) against the C-ABI and,
// when a GPU is present, pushes one block through FmDecoder::process.  CPU: the
// constructors must fail loudly (no fallback).
#include <cstdio>
#define FMR_FACADE_THROW
#include "../airspy-fmradion_amd/host/fmradion_facade.hpp"

int main() {
  IQSampleCoeff coeff = FilterParameters::delay_3taps_only_iq;
  if (FilterParameters::jj1bdx_fm_384kHz_medium.size() != 127) return 2;
  try {
    FmDecoder fm(false, coeff, true, FmDecoder::deemphasis_time_eu, false, 0);
    IQSampleVector blk(2048, IQSample(0.3f, 0.0f));
    SampleVector audio;
    fm.process(blk, audio);
    std::printf("gpu path: %zu audio samples, if_rms %.4f\n", audio.size(), fm.get_if_rms());
    IQSampleCoeff nbc = FilterParameters::jj1bdx_nbfm_48khz_default;
    NbfmDecoder nbfm(nbc, NbfmDecoder::freq_dev_normal);
    SampleVector a2;
    nbfm.process(blk, a2);
    std::printf("nbfm: %zu audio samples, if_rms %.4f\n", a2.size(), nbfm.get_if_rms());
    if (a2.size() != blk.size()) return 3;
    return 0;
  } catch (const std::exception &e) {
    std::printf("no gpu: %s\n", e.what());
    return 10;
  }
}
