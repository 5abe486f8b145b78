// Driver for tests/test_fileio.py: exercises host/fmradion_fileio.hpp without a GPU.
//   fileio_check read  <path> <raw 0|1> <U8_LE|S8_LE|S16_LE|S24_LE|FLOAT> <blocklen> <out.cf32>   -> prints rate and blocks
//   fileio_check write <RAW_INT16|RAW_FLOAT32|WAV_INT16|WAV_FLOAT32> <in.f64> <out> <rate> <stereo 0|1> <gain>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "../airspy-fmradion_amd/host/fmradion_fileio.hpp"
using namespace fmr_io;

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "read" && argc >= 7) {
    const std::string f = argv[4];
    IqFormat fmt = f == "U8_LE" ? IqFormat::U8_LE : f == "S8_LE" ? IqFormat::S8_LE : f == "S16_LE" ? IqFormat::S16_LE
                 : f == "S24_LE" ? IqFormat::S24_LE : IqFormat::FLOAT;
    IqFileReader r;
    if (!r.open(argv[2], atoi(argv[3]) != 0, fmt, 384000)) { std::printf("error: %s\n", r.error().c_str()); return 3; }
    FILE *fo = fopen(argv[6], "wb");
    IQSampleVector blk;
    size_t nblk = 0, total = 0;
    while (r.read_block(blk, (size_t)atol(argv[5]))) { fwrite(blk.data(), sizeof(IQSample), blk.size(), fo); nblk++; total += blk.size(); }
    fclose(fo);
    std::printf("rate %u blocks %zu samples %zu\n", r.sample_rate(), nblk, total);
    return 0;
  }
  if (cmd == "write" && argc >= 8) {
    const std::string f = argv[2];
    AudioFormat fmt = f == "RAW_INT16" ? AudioFormat::RAW_INT16 : f == "RAW_FLOAT32" ? AudioFormat::RAW_FLOAT32
                    : f == "WAV_INT16" ? AudioFormat::WAV_INT16 : AudioFormat::WAV_FLOAT32;
    FILE *fi = fopen(argv[3], "rb");
    if (!fi) return 3;
    AudioFileWriter w;
    if (!w.open(argv[4], (unsigned)atoi(argv[5]), atoi(argv[6]) != 0, fmt)) { std::printf("error: %s\n", w.error().c_str()); return 3; }
    SampleVector buf(1000);
    for (;;) {
      const size_t n = fread(buf.data(), sizeof(double), 1000, fi);
      if (!n) break;
      SampleVector part(buf.begin(), buf.begin() + n);
      adjust_gain(part, atof(argv[7]));
      if (!w.write(part)) return 4;
    }
    if (argc >= 9) {
      // snapshot of the file BEFORE close(): a receiver that is killed here must leave a playable file
      FILE *a = fopen(argv[4], "rb"), *b = fopen(argv[8], "wb");
      if (!a || !b) return 5;
      std::vector<unsigned char> all;
      unsigned char tmp[4096];
      size_t n;
      while ((n = fread(tmp, 1, sizeof tmp, a)) > 0) all.insert(all.end(), tmp, tmp + n);
      fwrite(all.data(), 1, all.size(), b);
      fclose(a); fclose(b);
    }
    w.close();
    std::printf("%s\n%s\n", pps_line(3, 123456789, 1700000000.25, -12.3456).c_str(), pps_block_line(42, 1700000000.5, 3.2).c_str());
    return 0;
  }
  return 2;
}
