// A stand-in for the reference's stream loop (main.cpp:879-1002, variant A of INTEGRATION.md): the same objects,
// constructed the same way (main.cpp:771-829), called in the same order on every block -- against the facade header.
// Source and sink are files instead of SDR hardware / sound card (those sit outside the hot path):
//   stream_loop <mode fm|nbfm|am|dsb|usb|lsb|cw|wspr> <ifrate> <fourth 0|1> <blocklen> <in.cf32|in.wav> <audio.f64> <pps.txt> [audio.wav]
// The IQ file is read through host/fmradion_fileio.hpp (RAW float, or any WAV the reader knows when the name ends in
// .wav -- the rate then comes from the header, as FileSource does).  Audio is written WITH the -6 dB of
// main.cpp:1000-1002 as raw doubles (for the tests) and, optionally, as a 16-bit WAV; PPS lines carry pps_index,
// sample_index, block_position, block.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#define FMR_FACADE_THROW
#include "../airspy-fmradion_amd/host/fmradion_facade.hpp"
#include "../airspy-fmradion_amd/host/fmradion_fileio.hpp"

int main(int argc, char **argv) {
  if (argc < 8) return 2;
  const std::string m = argv[1];
  double ifrate = atof(argv[2]);
  const bool enable_fs_fourth_downconverter = atoi(argv[3]) != 0;
  const size_t blocklen = (size_t)atol(argv[4]);
  ModType modtype = ModType::FM;
  if (m == "nbfm") modtype = ModType::NBFM; else if (m == "am") modtype = ModType::AM; else if (m == "dsb") modtype = ModType::DSB;
  else if (m == "usb") modtype = ModType::USB; else if (m == "lsb") modtype = ModType::LSB; else if (m == "cw") modtype = ModType::CW;
  else if (m == "wspr") modtype = ModType::WSPR;
  const double demodulator_rate = (modtype == ModType::FM) ? FmDecoder::sample_rate_if : AmDecoder::internal_rate_pcm;   // main.cpp:713-723
  fmr_io::IqFileReader reader;
  const std::string inpath = argv[5];
  const bool is_wav = inpath.size() > 4 && inpath.substr(inpath.size() - 4) == ".wav";
  if (!reader.open(inpath, !is_wav, fmr_io::IqFormat::FLOAT, (uint32_t)ifrate)) { std::printf("%s\n", reader.error().c_str()); return 3; }
  if (is_wav) ifrate = reader.sample_rate();                     // FileSource.cpp:188-193: the header overrides
  FILE *fau = fopen(argv[6], "wb"), *fpps = fopen(argv[7], "w");
  if (!fau || !fpps) return 3;
  fmr_io::AudioFileWriter wav;
  if (argc >= 9 && !wav.open(argv[8], 48000, modtype == ModType::FM, fmr_io::AudioFormat::WAV_INT16)) return 3;
  try {
    FourthConverterIQ fourth_downconverter(false);
    IfResampler if_resampler(ifrate, demodulator_rate);
    const bool enable_downsampling = (ifrate != demodulator_rate);
    IQSampleCoeff amfilter_coeff = FilterParameters::jj1bdx_am_48khz_narrow;       // -f narrow for the AM family
    IQSampleCoeff fmfilter_coeff = FilterParameters::delay_3taps_only_iq;         // -f default for FM
    IQSampleCoeff nbfmfilter_coeff = FilterParameters::jj1bdx_nbfm_48khz_default;
    AmDecoder am(amfilter_coeff, modtype);
    FmDecoder fm(false, fmfilter_coeff, true, FmDecoder::deemphasis_time_eu, false, 0);
    NbfmDecoder nbfm(nbfmfilter_coeff, NbfmDecoder::freq_dev_normal);
    if (const char *e = getenv("FMR_LOOP_BATCH")) fm.set_batch_blocks((unsigned)atoi(e));   // latency-for-throughput mode of the facade
    const float squelch_level = 0.0f;
    for (unsigned long long block = 0;; block++) {
      IQSampleVector iqsamples;
      if (!reader.read_block(iqsamples, blocklen)) break;
      IQSampleVector if_shifted_samples, if_samples;
      SampleVector audiosamples(0);
      if (enable_fs_fourth_downconverter) fourth_downconverter.process(iqsamples, if_shifted_samples);
      else if_shifted_samples = std::move(iqsamples);
      if (enable_downsampling) if_resampler.process(if_shifted_samples, if_samples);
      else if_samples = std::move(if_shifted_samples);
      if (if_samples.empty()) continue;
      double if_rms = 0.0;
      switch (modtype) {
      case ModType::FM: fm.process(if_samples, audiosamples); if_rms = fm.get_if_rms(); break;
      case ModType::NBFM: nbfm.process(if_samples, audiosamples); if_rms = nbfm.get_if_rms(); break;
      default: am.process(if_samples, audiosamples); if_rms = am.get_if_rms(); break;
      }
      if (audiosamples.empty()) continue;
      fmr_io::adjust_gain(audiosamples, if_rms >= squelch_level ? 0.5 : 0.0);
      fwrite(audiosamples.data(), sizeof(double), audiosamples.size(), fau);
      if (argc >= 9) wav.write(audiosamples);
      if (modtype == ModType::FM) {
        for (const PilotPhaseLock::PpsEvent &ev : fm.get_pps_events()) {
          fprintf(fpps, "%llu %llu %.17g %llu\n", (unsigned long long)ev.pps_index, (unsigned long long)ev.sample_index, ev.block_position, block);
          fm.erase_first_pps_event();
        }
      }
    }
    if (modtype == ModType::FM && fm.pending_blocks()) {
      // end of stream in batch mode: the blocks the facade still holds back (variant C)
      SampleVector audiosamples(0);
      fm.flush(audiosamples);
      fmr_io::adjust_gain(audiosamples, fm.get_if_rms() >= squelch_level ? 0.5 : 0.0);
      fwrite(audiosamples.data(), sizeof(double), audiosamples.size(), fau);
      if (argc >= 9) wav.write(audiosamples);
    }
  } catch (const std::exception &e) {
    std::printf("no gpu: %s\n", e.what());
    return 10;
  }
  fclose(fau); fclose(fpps);
  return 0;
}
