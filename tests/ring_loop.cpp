// A live source in miniature: a "driver" thread delivers callback buffers (RTL-SDR offset-binary bytes,
// RtlSdrSource.cpp:359-365, or Airspy float pairs, AirspySource.cpp:488-500) into the page-locked ring, the decoder
// thread pulls whatever has queued up and decodes the run in ONE call, straight out of the ring.
//   ring_loop <u8|cf32> <rate> <block_len> <max_run> <in.raw> <audio.f64>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../airspy-fmradion_amd/host/fmradion_ring.hpp"

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  const bool u8 = !std::strcmp(argv[1], "u8");
  const double rate = std::atof(argv[2]);
  const size_t blk = (size_t)std::atol(argv[3]);
  const int max_run = std::atoi(argv[4]);
  const size_t bps = u8 ? 2 : 8, bb = blk * bps;
  static const float delay3[3] = {0.f, 1.f, 0.f};
  fmr_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.device = 0; cfg.n_streams = 1; cfg.mode = FMR_MODE_FM; cfg.input_rate = rate; cfg.enable_resampler = 1;
  cfg.filter_coeff = delay3; cfg.n_filter_coeff = 3; cfg.stereo = 1; cfg.deemphasis_us = 50.0;
  cfg.max_block_len = blk; cfg.max_blocks = max_run; cfg.input_format = u8 ? FMR_IQ_U8 : FMR_IQ_CF32;
  fmr_chain *fm = nullptr;
  if (fmr_create(&cfg, &fm) != FMR_OK) { std::printf("fmr_create: %s\n", fmr_last_error()); return 10; }
  fmr_io::PinnedIqRing ring(bb, 64);
  FILE *fi = std::fopen(argv[5], "rb"), *fo = std::fopen(argv[6], "wb");
  if (!fi || !fo) return 3;
  std::thread driver([&] {                             // the callback thread: raw bytes in, nothing converted
    std::vector<unsigned char> buf(bb);
    size_t n, k = 0;
    while ((n = std::fread(buf.data(), 1, bb, fi)) == bb) {
      while (ring.queued() >= ring.depth()) std::this_thread::yield();   // (a file can wait; a real driver would drop)
      ring.push(buf.data(), n);
      if ((++k % 7) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    ring.push_end();
  });
  std::vector<double> audio(2 * (blk + 64) * (size_t)max_run);
  std::vector<uint32_t> lens((size_t)max_run, (uint32_t)blk), alen((size_t)max_run);
  size_t calls = 0, blocks = 0, longest = 0;
  for (;;) {
    size_t n = 0;
    const void *p = ring.pull((size_t)max_run, n);
    if (!p) break;
    const int rc = fmr_process_blocks(fm, static_cast<const float *>(p), blk * n, lens.data(), (int)n, audio.data(), audio.size(), alen.data());
    if (rc != FMR_OK) { std::printf("fmr_process_blocks: %s\n", fmr_last_error()); return 11; }
    ring.release(n);
    size_t na = 0;
    for (size_t b = 0; b < n; b++) na += alen[b];
    std::fwrite(audio.data(), sizeof(double), na, fo);
    calls++; blocks += n; if (n > longest) longest = n;
  }
  driver.join();
  std::fclose(fi); std::fclose(fo);
  fmr_status st; fmr_get_status(fm, 0, &st);
  std::printf("blocks %zu calls %zu longest_run %zu overruns %zu stereo %d\n", blocks, calls, longest, ring.overruns(), st.stereo_detected);
  fmr_destroy(fm);
  return 0;
}
