// fmradion_ring.hpp -- the hand-off between a live source's callback thread and the decoder thread.
//
// The reference copies every driver buffer into a freshly allocated std::vector<IQSample> -- converting on the way:
// float pairs in AirspySource::callback (sfmbase/AirspySource.cpp:488-500), (b - 128) / 128 in
// RtlSdrSource::get_samples (sfmbase/RtlSdrSource.cpp:359-365) -- and queues the vectors in a mutex / condvar
// DataBuffer (include/DataBuffer.h:35-90) that the main loop pulls one block at a time (main.cpp:889).
//
// Here the callback copies the driver's RAW bytes into the next block of a ring in page-locked host memory
// (fmr_host_alloc) and the decoder thread hands runs of blocks to fmr_process_blocks in place: no allocation per
// buffer, no conversion on the CPU (fmr_config.input_format does it in the front-end kernel: HBM and PCIe carry 2 B per
// RTL-SDR sample instead of 8), DMA straight out of the ring, and as many blocks per GPU call as have arrived -- the
// batch size follows the backlog, which is what keeps a GPU decoder ahead of a source (one call costs about
// the same for 1 and for 64 blocks).
//
// One producer thread, one consumer thread (the reference's threading model: source thread + main thread).  The ring is
// bounded: when the consumer falls behind by the whole ring the producer's push fails and the overrun is counted -- the
// driver's own behaviour when its callback cannot deliver (the reference's DataBuffer grows without bound instead).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/fmradion_amd.h"

namespace fmr_io {

class PinnedIqRing {
 public:
  typedef void *(*alloc_fn)(size_t);
  typedef void (*free_fn)(void *);

  // block_bytes: one source block (block_len samples x bytes per IQ sample of the RAW format); n_blocks: ring depth.
  // The default allocator is the library's page-locked one; tests of the ring logic alone may pass malloc / free.
  PinnedIqRing(size_t block_bytes, size_t n_blocks, alloc_fn alloc = fmr_host_alloc, free_fn release = fmr_host_free)
      : m_block(block_bytes), m_n(n_blocks), m_free(release) {
    m_valid = new size_t[m_n]();
    m_mem = static_cast<unsigned char *>(alloc(m_block * m_n));
    if (!m_mem) {
      std::fprintf(stderr, "PinnedIqRing: cannot allocate %zu bytes of page-locked memory: %s\n", m_block * m_n, fmr_last_error());
      std::exit(1);                       // the reference's sources fail hard at start-up too; there is no pageable fallback
    }
  }
  ~PinnedIqRing() { if (m_mem) m_free(m_mem); delete[] m_valid; }
  PinnedIqRing(const PinnedIqRing &) = delete;
  PinnedIqRing &operator=(const PinnedIqRing &) = delete;

  // ---- producer (driver callback thread) ----------------------------------------------------------------------
  // Copy one block (exactly block_bytes, or fewer for the last block of a stream: the rest is zero filled and the
  // valid byte count is kept per slot).  false = ring full, the block is dropped and counted (DataBuffer::push, :35-45);
  // a buffer LARGER than a block is refused and counted too (nothing is truncated silently).
  bool push(const void *data, size_t bytes) {
    if (bytes == 0) return true;
    if (bytes > m_block) { m_oversize.fetch_add(1, std::memory_order_relaxed); return false; }
    const uint64_t h = m_head.load(std::memory_order_relaxed);
    if (h - m_tail.load(std::memory_order_acquire) >= m_n) { m_overruns.fetch_add(1, std::memory_order_relaxed); return false; }
    unsigned char *dst = m_mem + (h % m_n) * m_block;
    std::memcpy(dst, data, bytes);
    if (bytes < m_block) std::memset(dst + bytes, 0, m_block - bytes);
    m_valid[h % m_n] = bytes;                      // published by the release store of m_head below
    m_last_bytes.store(bytes, std::memory_order_relaxed);
    m_head.store(h + 1, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(m_mu); }
    m_cv.notify_one();
    return true;
  }
  // DataBuffer::push_end (:48-56)
  void push_end() {
    m_end.store(true, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(m_mu); }
    m_cv.notify_one();
  }

  // ---- consumer (decoder thread) ----------------------------------------------------------------------------------
  // Wait until at least one block is queued (or the end is marked), then return a run of up to max_blocks CONTIGUOUS
  // blocks (a run never wraps around the end of the ring).  n = 0 with a null pointer: end of stream
  // (DataBuffer::pull returning an empty vector, :69-81).
  const void *pull(size_t max_blocks, size_t &n) {
    std::unique_lock<std::mutex> lk(m_mu);
    m_cv.wait(lk, [&] { return queued() > 0 || m_end.load(std::memory_order_acquire); });
    lk.unlock();
    const size_t q = queued();
    if (q == 0) { n = 0; return nullptr; }
    const uint64_t t = m_tail.load(std::memory_order_relaxed);
    const size_t at = (size_t)(t % m_n);
    size_t run = q < max_blocks ? q : max_blocks;
    if (run > m_n - at) run = m_n - at;
    n = run;
    return m_mem + at * m_block;
  }
  // The run has been consumed (its GPU call has returned): the producer may reuse the blocks.
  void release(size_t n) { m_tail.fetch_add(n, std::memory_order_release); }
  // DataBuffer::pull_end_reached (:84-90)
  bool pull_end_reached() { return m_end.load(std::memory_order_acquire) && queued() == 0; }

  size_t queued() const { return (size_t)(m_head.load(std::memory_order_acquire) - m_tail.load(std::memory_order_acquire)); }   // DataBuffer::queue_size
  size_t overruns() const { return m_overruns.load(std::memory_order_relaxed); }
  size_t oversize_rejected() const { return m_oversize.load(std::memory_order_relaxed); }
  // valid bytes of block i of the run pull() has just returned (block_bytes except for a short final block)
  size_t run_block_bytes(size_t i) const { return m_valid[(size_t)((m_tail.load(std::memory_order_relaxed) + i) % m_n)]; }
  size_t block_bytes() const { return m_block; }
  size_t depth() const { return m_n; }
  // valid bytes of the block pushed LAST (ambiguous once more than one block is queued: prefer run_block_bytes)
  size_t last_block_bytes() const { return m_last_bytes.load(std::memory_order_relaxed); }

 private:
  size_t m_block, m_n;
  free_fn m_free;
  unsigned char *m_mem = nullptr;
  std::atomic<uint64_t> m_head{0}, m_tail{0};
  std::atomic<size_t> m_overruns{0}, m_last_bytes{0}, m_oversize{0};
  size_t *m_valid = nullptr;                       // [n_blocks]: valid bytes per slot
  std::atomic<bool> m_end{false};
  std::mutex m_mu;
  std::condition_variable m_cv;
};

}  // namespace fmr_io
