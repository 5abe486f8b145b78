// Ring logic alone (no GPU): built with malloc / free as the allocator.  One producer, one consumer, pauses on both
// sides; blocks must arrive complete, in order, in runs that never wrap; overruns are counted, not blocking.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "../airspy-fmradion_amd/host/fmradion_ring.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  using fmr_io::PinnedIqRing;
  // ---- 1. ordered delivery under jitter
  {
    const size_t BB = 256, DEPTH = 8, NBLK = 20000;
    PinnedIqRing ring(BB, DEPTH, std::malloc, std::free);
    size_t dropped = 0;
    std::vector<unsigned> sent;                      // sequence numbers that went in
    std::thread prod([&] {
      std::mt19937 g(1);
      std::vector<unsigned char> buf(BB);
      for (unsigned i = 0; i < NBLK; i++) {
        for (size_t k = 0; k < BB; k += 4) std::memcpy(&buf[k], &i, 4);
        if (ring.push(buf.data(), BB)) sent.push_back(i); else dropped++;
        if ((g() & 63) == 0) std::this_thread::sleep_for(std::chrono::microseconds(g() & 255));
      }
      ring.push_end();
    });
    std::mt19937 g(2);
    size_t got = 0, runs = 0, max_run = 0;
    std::vector<unsigned> seen;
    for (;;) {
      size_t n = 0;
      const unsigned char *p = static_cast<const unsigned char *>(ring.pull(5, n));
      if (!p) break;
      CHECK(n >= 1 && n <= 5);
      for (size_t b = 0; b < n; b++) {
        unsigned v; std::memcpy(&v, p + b * BB, 4);
        for (size_t k = 0; k < BB; k += 4) { unsigned u; std::memcpy(&u, p + b * BB + k, 4); CHECK(u == v); }   // a block is never torn
        seen.push_back(v);
      }
      got += n; runs++; if (n > max_run) max_run = n;
      if ((g() & 31) == 0) std::this_thread::sleep_for(std::chrono::microseconds(g() & 511));
      ring.release(n);
    }
    prod.join();
    CHECK(ring.pull_end_reached());
    CHECK(seen == sent);                              // in order, nothing lost beyond the counted drops
    CHECK(got + dropped == NBLK && dropped == ring.overruns());
    CHECK(max_run > 1);                               // the backlog was batched
    std::printf("ordered delivery: %zu blocks in %zu runs (max %zu), %zu overruns counted\n", got, runs, max_run, dropped);
  }
  // ---- 2. a full ring rejects and counts; a short last block is zero filled
  {
    PinnedIqRing ring(64, 4, std::malloc, std::free);
    unsigned char buf[64]; std::memset(buf, 0xAB, sizeof buf);
    for (int i = 0; i < 4; i++) CHECK(ring.push(buf, 64));
    CHECK(!ring.push(buf, 64) && ring.overruns() == 1 && ring.queued() == 4);
    size_t n = 0;
    const void *p = ring.pull(16, n);
    CHECK(p && n == 4);
    ring.release(3);
    CHECK(ring.push(buf, 10) && ring.last_block_bytes() == 10);
    ring.release(1);
    p = ring.pull(16, n);
    CHECK(p && n == 1);
    const unsigned char *q = static_cast<const unsigned char *>(p);
    for (int k = 0; k < 64; k++) CHECK(q[k] == (k < 10 ? 0xAB : 0));
    CHECK(ring.run_block_bytes(0) == 10);             // valid bytes per slot, not only of the block pushed last
    ring.release(1);
    // a buffer larger than a block is refused and counted -- never truncated
    unsigned char big[65] = {0};
    CHECK(!ring.push(big, 65) && ring.oversize_rejected() == 1 && ring.queued() == 0 && ring.overruns() == 1);
    CHECK(ring.push(buf, 64) && ring.push(buf, 7));
    p = ring.pull(16, n);
    CHECK(p && n == 2 && ring.run_block_bytes(0) == 64 && ring.run_block_bytes(1) == 7 && ring.last_block_bytes() == 7);
    ring.release(2);
    ring.push_end();
    p = ring.pull(16, n);
    CHECK(!p && n == 0 && ring.pull_end_reached());
    std::printf("full ring / short block / end marker: ok\n");
  }
  // ---- 3. runs stop at the end of the ring (contiguous memory for one GPU call)
  {
    PinnedIqRing ring(16, 8, std::malloc, std::free);
    unsigned char buf[16] = {0};
    for (int i = 0; i < 6; i++) CHECK(ring.push(buf, 16));
    size_t n = 0;
    const unsigned char *p0 = static_cast<const unsigned char *>(ring.pull(6, n));
    CHECK(n == 6); ring.release(6);
    for (int i = 0; i < 5; i++) CHECK(ring.push(buf, 16));          // blocks 6, 7 | 0, 1, 2
    const unsigned char *p1 = static_cast<const unsigned char *>(ring.pull(8, n));
    CHECK(n == 2 && p1 == p0 + 6 * 16); ring.release(2);
    const unsigned char *p2 = static_cast<const unsigned char *>(ring.pull(8, n));
    CHECK(n == 3 && p2 == p0); ring.release(3);
    std::printf("runs are contiguous: ok\n");
  }
  std::printf("RING OK\n");
  return 0;
}
