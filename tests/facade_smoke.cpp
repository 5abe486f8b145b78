// Compiles the C++ facade (reference class names/signatures) against the C-ABI and,
// when a GPU is present, pushes one block through FmDecoder::process.  CPU: the
// constructors must fail loudly (no fallback).
#include <cstdio>
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
    return 0;
  } catch (const std::exception &e) {
    std::printf("no gpu: %s\n", e.what());
    return 10;
  }
}
