// fmradion_fileio.hpp -- file containers on both sides of the hot path, header only, no
// third-party library.  The reference goes through libsndfile (absent here; the conversions below are the ones its
// sf_read_float / sf_write_double apply with the default normalisation, libsndfile 1.x pcm.c / float32.c):
//
//   IqFileReader    the IQ side of FileSource (sfmbase/FileSource.cpp:120-128 formats, :163-251 open, :491-531 block
//                   read): WAV / RF64 with PCM u8, PCM 16, PCM 24 or IEEE float32 frames of 2 channels (I = ch 0,
//                   Q = ch 1), or headerless RAW in U8_LE, S8_LE, S16_LE, S24_LE, FLOAT.  Integer formats are scaled by
//                   2^-(bits-1) (u8: (b - 128) / 128), float is taken as is -- what sf_read_float delivers.
//   AudioFileWriter the file side of AudioOutput (sfmbase/AudioOutput.cpp:34-167 SndfileOutput): RAW or WAV in int16
//                   (lrint(x * 32767), libsndfile's normalised double -> short, no clipping) or float32; the WAV
//                   header is valid from the start and refreshed as the data grow (SFC_SET_UPDATE_HEADER_AUTO, :91-93),
//                   as RF64 ('ds64' chunk) when the data passes 4 GiB -- the outcome of
//                   SFC_RF64_AUTO_DOWNGRADE (:78-89).
//   adjust_gain     the -6 dB of main.cpp:1000-1002;  pps_line: the PPS text record of main.cpp:1084-1111.
//
// Host-side plumbing only: nothing here touches the GPU; the decoders take the blocks these classes deliver.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace fmr_io {

using IQSample = std::complex<float>;
using IQSampleVector = std::vector<IQSample>;
using SampleVector = std::vector<double>;

enum class IqFormat { U8_LE, S8_LE, S16_LE, S24_LE, FLOAT };   // FileSource.h FormatType

class IqFileReader {
public:
  ~IqFileReader() { close(); }
  // raw = true: headerless file in `format`; raw = false: WAV / RF64, format and rate come from the header
  bool open(const std::string &path, bool raw, IqFormat format = IqFormat::FLOAT, uint32_t sample_rate = 0) {
    close();
    m_fp = std::fopen(path.c_str(), "rb");
    if (!m_fp) { m_error = "Failed to open " + path; return false; }
    m_rate = sample_rate;
    m_format = format;
    m_left = UINT64_MAX;
    if (raw) return true;
    return parse_wav(path);
  }
  void close() { if (m_fp) std::fclose(m_fp); m_fp = nullptr; }
  uint32_t sample_rate() const { return m_rate; }
  IqFormat format() const { return m_format; }
  const std::string &error() const { return m_error; }
  // one block of up to block_length IQ samples (FileSource.cpp:491-531); false at the end of the data
  bool read_block(IQSampleVector &samples, size_t block_length) {
    if (!m_fp || block_length == 0) return false;
    const size_t bps = bytes_per_component();
    uint64_t want = (uint64_t)block_length * 2 * bps;
    if (want > m_left) want = m_left - (m_left % (2 * bps));
    m_buf.resize((size_t)want);
    const size_t got = want ? std::fread(m_buf.data(), 1, (size_t)want, m_fp) : 0;
    const size_t n = got / (2 * bps);
    if (n == 0) return false;
    if (m_left != UINT64_MAX) m_left -= got;
    samples.resize(n);
    const unsigned char *p = m_buf.data();
    for (size_t i = 0; i < n; i++) {
      float v[2];
      for (int c = 0; c < 2; c++, p += bps) {
        switch (m_format) {
        case IqFormat::U8_LE: v[c] = ((int)p[0] - 128) * (1.0f / 128.0f); break;
        case IqFormat::S8_LE: v[c] = (float)(signed char)p[0] * (1.0f / 128.0f); break;
        case IqFormat::S16_LE: v[c] = (float)(int16_t)(p[0] | (p[1] << 8)) * (1.0f / 32768.0f); break;
        case IqFormat::S24_LE: {
          int32_t x = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
          v[c] = (float)x * (1.0f / 8388608.0f);
        } break;
        default: std::memcpy(&v[c], p, 4); break;
        }
      }
      samples[i] = IQSample(v[0], v[1]);
    }
    return true;
  }

private:
  size_t bytes_per_component() const {
    switch (m_format) {
    case IqFormat::U8_LE: case IqFormat::S8_LE: return 1;
    case IqFormat::S16_LE: return 2;
    case IqFormat::S24_LE: return 3;
    default: return 4;
    }
  }
  static constexpr uint64_t kMaxHeaderChunk = 4096;
  static uint32_t rd32(const unsigned char *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
  bool parse_wav(const std::string &path) {
    unsigned char h[12];
    if (std::fread(h, 1, 12, m_fp) != 12 || (std::memcmp(h, "RIFF", 4) && std::memcmp(h, "RF64", 4)) || std::memcmp(h + 8, "WAVE", 4)) {
      m_error = "Unsupported major format " + path;
      return false;
    }
    const bool rf64 = !std::memcmp(h, "RF64", 4);
    uint64_t data64 = 0;
    bool have_fmt = false;
    for (;;) {
      unsigned char ck[8];
      if (std::fread(ck, 1, 8, m_fp) != 8) { m_error = "no data chunk in " + path; return false; }
      uint64_t size = rd32(ck + 4);
      if (!std::memcmp(ck, "ds64", 4)) {
        if (size < 16 || size > kMaxHeaderChunk) { m_error = "bad ds64 chunk"; return false; }    // sizes come from an untrusted header
        std::vector<unsigned char> b((size_t)size);
        if (std::fread(b.data(), 1, b.size(), m_fp) != b.size()) { m_error = "bad ds64 chunk"; return false; }
        data64 = (uint64_t)rd32(b.data() + 8) | (uint64_t)rd32(b.data() + 12) << 32;
      } else if (!std::memcmp(ck, "fmt ", 4)) {
        if (size < 16 || size > kMaxHeaderChunk) { m_error = "bad fmt chunk"; return false; }
        std::vector<unsigned char> b((size_t)size);
        if (std::fread(b.data(), 1, b.size(), m_fp) != b.size()) { m_error = "bad fmt chunk"; return false; }
        unsigned tag = b[0] | b[1] << 8;
        const unsigned channels = b[2] | b[3] << 8, bits = b[14] | b[15] << 8;
        m_rate = rd32(b.data() + 4);
        if (tag == 0xFFFE && size >= 26) tag = b[24] | b[25] << 8;          // WAVE_FORMAT_EXTENSIBLE: sub-format GUID
        if (channels != 2) { m_error = "IQ files have two channels"; return false; }
        if (tag == 1 && bits == 8) m_format = IqFormat::U8_LE;               // 8-bit WAV PCM is unsigned
        else if (tag == 1 && bits == 16) m_format = IqFormat::S16_LE;
        else if (tag == 1 && bits == 24) m_format = IqFormat::S24_LE;
        else if (tag == 3 && bits == 32) m_format = IqFormat::FLOAT;
        else { m_error = "Unsupported sub type in " + path; return false; }
        have_fmt = true;
      } else if (!std::memcmp(ck, "data", 4)) {
        if (!have_fmt) { m_error = "data chunk before fmt chunk"; return false; }
        m_left = (rf64 && size == 0xFFFFFFFFu) ? data64 : size;
        // what the file really holds behind this header (a regular file; a pipe reports nothing and is read to its end)
        uint64_t rest = UINT64_MAX;
        {
          const long here = std::ftell(m_fp);
          if (here >= 0 && std::fseek(m_fp, 0, SEEK_END) == 0) {
            const long end = std::ftell(m_fp);
            if (end >= here) rest = (uint64_t)(end - here);
            std::fseek(m_fp, here, SEEK_SET);
          }
        }
        // Streaming recorders leave 0 or 0xFFFFFFFF in the data size (the header is never finalised -- AudioFileWriter's
        // own provisional header, if the writer is killed within its first second): the data then run to the end of the
        // file.  Only if the data chunk is the LAST chunk, though: a size of 0 followed by something that reads as a
        // chunk header (four printable characters and a length that fits the rest of the file: LIST, bext, id3 ...) is an
        // empty recording with metadata behind it, not samples.
        const bool unknown = m_left == 0 || (!rf64 && size == 0xFFFFFFFFu) || (rf64 && size == 0xFFFFFFFFu && data64 == 0);
        if (unknown) {
          bool chunk_follows = false;
          if (size == 0 && rest != UINT64_MAX && rest >= 8) {
            unsigned char nx[8];
            const long here = std::ftell(m_fp);
            if (std::fread(nx, 1, 8, m_fp) == 8) {
              bool printable = true;
              for (int i = 0; i < 4; i++) printable = printable && nx[i] >= 0x20 && nx[i] < 0x7f;
              chunk_follows = printable && (uint64_t)rd32(nx + 4) + 8 <= rest;
            }
            std::fseek(m_fp, here, SEEK_SET);
          }
          m_left = chunk_follows ? 0 : UINT64_MAX;
        }
        if (m_left != UINT64_MAX && rest != UINT64_MAX && m_left > rest) m_left = rest;     // truncated file: what is there
        return true;
      } else {
        if (std::fseek(m_fp, (long)(size + (size & 1)), SEEK_CUR)) { m_error = "truncated file"; return false; }
        continue;
      }
      if (size & 1) std::fgetc(m_fp);
    }
  }
  std::FILE *m_fp = nullptr;
  uint32_t m_rate = 0;
  IqFormat m_format = IqFormat::FLOAT;
  uint64_t m_left = UINT64_MAX;
  std::vector<unsigned char> m_buf;
  std::string m_error;
};

enum class AudioFormat { RAW_INT16, RAW_FLOAT32, WAV_INT16, WAV_FLOAT32 };   // main.cpp -R / -F / -W / -G

class AudioFileWriter {
public:
  ~AudioFileWriter() { close(); }
  bool open(const std::string &path, unsigned samplerate, bool stereo, AudioFormat fmt) {
    close();
    m_fp = std::fopen(path.c_str(), "wb");
    if (!m_fp) { m_error = "can not open '" + path + "'"; return false; }
    m_rate = samplerate; m_channels = stereo ? 2 : 1; m_fmt = fmt; m_bytes = 0; m_header_at = 0;
    // a valid header from the first byte on, refreshed as the data grow (the reference sets
    // SFC_SET_UPDATE_HEADER_AUTO, AudioOutput.cpp:91-93): a receiver that is killed leaves a playable file
    if (is_wav() && !write_header()) { m_error = "can not write the header of '" + path + "' (not seekable?)"; close_raw(); return false; }
    return true;
  }
  bool write(const SampleVector &samples) {
    if (!m_fp) return false;
    if (m_fmt == AudioFormat::RAW_INT16 || m_fmt == AudioFormat::WAV_INT16) {
      m_i16.resize(samples.size());
      for (size_t i = 0; i < samples.size(); i++) m_i16[i] = (int16_t)std::lrint(samples[i] * 32767.0);
      if (std::fwrite(m_i16.data(), 2, m_i16.size(), m_fp) != m_i16.size()) { m_error = "write failed"; return false; }
      m_bytes += 2 * m_i16.size();
    } else {
      m_f32.resize(samples.size());
      for (size_t i = 0; i < samples.size(); i++) m_f32[i] = (float)samples[i];
      if (std::fwrite(m_f32.data(), 4, m_f32.size(), m_fp) != m_f32.size()) { m_error = "write failed"; return false; }
      m_bytes += 4 * m_f32.size();
    }
    // refresh the header every ~second of audio (and on close): cheap, and the file stays valid
    if (is_wav() && m_bytes - m_header_at >= (uint64_t)m_rate * m_channels * 2) {
      if (!write_header()) { m_error = "header update failed"; return false; }
    }
    return true;
  }
  void close() {
    if (!m_fp) return;
    if (is_wav()) {
      if (m_bytes & 1) std::fputc(0, m_fp);
      if (!write_header()) m_error = "header update failed";
    }
    close_raw();
  }
  const std::string &error() const { return m_error; }

private:
  bool is_wav() const { return m_fmt == AudioFormat::WAV_INT16 || m_fmt == AudioFormat::WAV_FLOAT32; }
  // one layout for both outcomes: 'RIFF' + 'JUNK' placeholder (plain WAV), or 'RF64' + 'ds64' (data >= 4 GiB)
  static constexpr size_t header_size() { return 12 + 8 + 28 + 8 + 16 + 8; }
  static void put32(unsigned char *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
  void close_raw() { if (m_fp) std::fclose(m_fp); m_fp = nullptr; }
  // (re)write the header for the bytes written so far and return to the end of the data
  bool write_header() {
    unsigned char h[80] = {0};
    const bool f32 = m_fmt == AudioFormat::WAV_FLOAT32;
    const unsigned bits = f32 ? 32 : 16, align = m_channels * bits / 8;
    const uint64_t riff = header_size() - 8 + m_bytes + (m_bytes & 1);
    const bool big = riff > 0xFFFFFFFFull;
    std::memcpy(h, big ? "RF64" : "RIFF", 4);
    put32(h + 4, big ? 0xFFFFFFFFu : (uint32_t)riff);
    std::memcpy(h + 8, "WAVE", 4);
    std::memcpy(h + 12, big ? "ds64" : "JUNK", 4);
    put32(h + 16, 28);
    if (big) {
      put32(h + 20, (uint32_t)riff); put32(h + 24, (uint32_t)(riff >> 32));
      put32(h + 28, (uint32_t)m_bytes); put32(h + 32, (uint32_t)(m_bytes >> 32));
      const uint64_t frames = m_bytes / align;
      put32(h + 36, (uint32_t)frames); put32(h + 40, (uint32_t)(frames >> 32));
    }
    std::memcpy(h + 48, "fmt ", 4);
    put32(h + 52, 16);
    h[56] = f32 ? 3 : 1; h[58] = (unsigned char)m_channels;
    put32(h + 60, m_rate); put32(h + 64, m_rate * align);
    h[68] = (unsigned char)align; h[70] = (unsigned char)bits;
    std::memcpy(h + 72, "data", 4);
    put32(h + 76, big ? 0xFFFFFFFFu : (uint32_t)m_bytes);
    if (std::fseek(m_fp, 0, SEEK_SET)) return false;
    if (std::fwrite(h, 1, header_size(), m_fp) != header_size()) return false;
    if (std::fseek(m_fp, 0, SEEK_END)) return false;
    m_header_at = m_bytes;
    return true;
  }
  std::FILE *m_fp = nullptr;
  unsigned m_rate = 0, m_channels = 2;
  AudioFormat m_fmt = AudioFormat::RAW_INT16;
  uint64_t m_bytes = 0, m_header_at = 0;
  std::vector<int16_t> m_i16;
  std::vector<float> m_f32;
  std::string m_error;
};

// Utility::adjust_gain (include/Utility.h), used at main.cpp:1000-1002 with 0.5 (squelch open) or 0.0
inline void adjust_gain(SampleVector &samples, double gain) { for (auto &v : samples) v *= gain; }

// the PPS record main.cpp:1087-1092 prints for an FM PpsEvent: "{:>8} {:>14} {:18.6f} {:+9.3f}"
inline std::string pps_line(uint64_t pps_index, uint64_t sample_index, double timestamp, double if_level_db) {
  char b[96];
  std::snprintf(b, sizeof b, "%8llu %14llu %18.6f %+9.3f", (unsigned long long)pps_index, (unsigned long long)sample_index, timestamp, if_level_db);
  return b;
}
// the record of the other modes (main.cpp:1104-1106): "{:11} {:18.6f} {:+9.3f}"
inline std::string pps_block_line(uint64_t block, double timestamp, double if_level_db) {
  char b[96];
  std::snprintf(b, sizeof b, "%11llu %18.6f %+9.3f", (unsigned long long)block, timestamp, if_level_db);
  return b;
}

}  // namespace fmr_io
