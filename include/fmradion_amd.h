/*
 * fmradion_amd.h -- C-ABI of the MI355X-native FM/AM demodulation hot path.
 *
 * Drop-in boundary: these entry points are what a
 * binding of the reference's stream loop (main.cpp:879-1002) needs in order
 * to replace
 *     FourthConverterIQ::process   include/FourthConverterIQ.h:38   (main.cpp:916)
 *     IfResampler::process         include/IfResampler.h:35-38      (main.cpp:923)
 *     FmDecoder::process + getters include/FmDecode.h:63-105        (main.cpp:956-957)
 *     AmDecoder::process + getters include/AmDecode.h:48-65         (main.cpp:971-972)
 * Plain pointers and sizes only; no C++ or torch types.  The C++ facade with
 * the reference's class names and signatures is
 * airspy-fmradion_amd/host/fmradion_facade.hpp; the reference-side binding is
 * shown in INTEGRATION.md.
 *
 * One object = one decoder chain for `n_streams` independent IQ streams that
 * all see the same block lengths (batch dimension, config 5 of BASELINE.json).
 * All compute runs in hand-written HIP kernels for gfx950; there is no CPU
 * fallback: every call fails with FMR_ERR_NO_DEVICE when no GPU is present.
 *
 * Error convention: 0 = FMR_OK, negative = error; `*n_out == 0` is the
 * reference's "nothing yet" (empty output vector, FmDecode.cpp:89-92,185-188).
 */
#ifndef FMRADION_AMD_H
#define FMRADION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  FMR_OK = 0,
  FMR_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime error at create */
  FMR_ERR_BAD_ARG = -2,
  FMR_ERR_UNSUPPORTED = -3,   /* e.g. resampling ratio outside the design range */
  FMR_ERR_CAPACITY = -4,      /* output buffer or configured maximum too small */
  FMR_ERR_HIP = -5            /* HIP runtime failure; see fmr_last_error() */
};

/* ModType values follow include/SoftFM.h:49 */
/* ModType order of the reference (include/SoftFM.h:49) */
enum { FMR_MODE_FM = 0, FMR_MODE_NBFM = 1, FMR_MODE_AM = 2, FMR_MODE_DSB = 3, FMR_MODE_USB = 4, FMR_MODE_LSB = 5,
       FMR_MODE_CW = 6, FMR_MODE_WSPR = 7 };
enum { FMR_IQ_CF32 = 0, FMR_IQ_S16 = 1, FMR_IQ_U8 = 2, FMR_IQ_S8 = 3 };
/* Specification of the IfResampler stand-in ("Resampler specification"; r8brain itself is absent from the
 * reference tree).  FAST: pass band 0.885 x Nyquist, aliases of the pass band rejected by 140 dB, what falls between
 * 0.885 x Nyquist and Nyquist rolls off -- the throughput configuration.  R8B: the defaults of the
 * r8b::CDSPResampler24 the reference constructs (sfmbase/IfResampler.cpp:25-29): pass band 0.98 x Nyquist, stop band
 * from Nyquist on, 180 dB -- the reference-equivalent configuration (a neighbouring station 200 kHz away is filtered
 * exactly as the reference filters it), 15 x the stage-B arithmetic. */
enum { FMR_RESAMPLER_FAST = 0, FMR_RESAMPLER_R8B = 1 };

/* PilotPhaseLock::PpsEvent (include/PilotPhaseLock.h:40-44) + the index of the
 * block (within the call) that produced it. */
typedef struct {
  uint64_t pps_index;
  uint64_t sample_index;
  double block_position;
  uint32_t block;
  uint32_t stream;
} fmr_pps_event;

/* Configuration of one chain.  ZERO-INITIALISE (memset / = {0} / fmr_config cfg{}), then set fields and struct_size:
 * every field added since the first version means "as before" when it is 0, and a field left uninitialised is refused
 * if its value is not one this library knows. */
typedef struct {
  int device;                 /* HIP device ordinal */
  int n_streams;              /* >= 1 independent IQ streams (batch) */
  int mode;                   /* FMR_MODE_FM | _NBFM | _AM | _DSB | _USB | _LSB | _CW | _WSPR (the last four ignore
                               * filter_coeff: AmDecoder uses its built-in 2049-tap SSB / CW tables there) */
  double input_rate;          /* sample rate of the IQ handed to process */
  /* Front end.  0 = the decoder is fed at its own rate (384 kHz FM / 48 kHz
   * AM) and no IfResampler runs (main.cpp:778 enable_downsampling=false). */
  int enable_resampler;
  int enable_fourth_down;     /* FourthConverterIQ(false) before the resampler */
  /* FmDecoder ctor arguments (include/FmDecode.h:63-64) */
  int fmfilter_enable;
  const float *filter_coeff;  /* FM IF filter / AM filter taps (caller-supplied, main.cpp:780-810) */
  int n_filter_coeff;
  int stereo;
  double deemphasis_us;       /* 50 / 75 / 0 */
  int pilot_shift;
  unsigned multipath_stages;
  /* capacity */
  size_t max_block_len;       /* largest input block (samples) per call */
  int max_blocks;             /* largest number of blocks per call */
  /* NbfmDecoder ctor argument (include/NbfmDecode.h:49): full-scale deviation in Hz, 0 = freq_dev_normal (8000) */
  double nbfm_freq_dev;
  /* Source sample format of every `iq` argument (fused ingest: converted while the front-end kernel stages its
   * tile; needs enable_resampler).  Conversions are the reference's: FMR_IQ_U8 = RTL-SDR offset binary
   * (RtlSdrSource.cpp:359-365), the others = what sf_read_float delivers for the FileSource formats
   * S16_LE / S8_LE / U8_LE / FLOAT (FileSource.cpp:120-128,491-531).  `iq` pointers are then pointers to
   * interleaved I,Q samples of that type; counts and strides stay in IQ samples; raw-format device buffers and
   * strides must be 16-byte aligned. */
  int input_format;           /* FMR_IQ_CF32 (default) | FMR_IQ_S16 | FMR_IQ_U8 | FMR_IQ_S8 */
  /* Front-end-only chains (mode = -1): IfResampler(input_rate, output_rate) of include/IfResampler.h:35; 0 = the FM
   * IF rate (384 kHz).  Decoder chains ignore it (their rate is fixed: FmDecode.h:38, AmDecode.h:36). */
  double output_rate;
  int resampler_class;        /* FMR_RESAMPLER_FAST (default) | FMR_RESAMPLER_R8B: specification of the IF resampler */
  /* sizeof(fmr_config) of the header the caller was built against; 0 = not stated (taken as this header's).  The struct
   * grows at its end from version to version: fmr_create refuses a size it does not know instead of reading past a
   * shorter struct or misreading a longer one.  Zero-initialise the whole struct first (an unset field must read 0). */
  unsigned struct_size;
  /* 1: one in-order launch chain per call instead of the pipelined one (front end of call N+1 behind the PLL stage of
   * call N, audio tail a call behind).  For callers that synchronise after every call -- fmr_process / fmr_process_blocks
   * through host buffers, the facade's FmDecoder::process -- the pipelined chain has nothing to overlap and pays for its
   * stage hand-offs: 341 against 303 us per 65536-sample block.  Same audio, bit for bit.  0 (default): pipelined. */
  int in_order;
} fmr_config;

/* Per-stream status after the most recent call (getters of FmDecode.h:77-105 /
 * AmDecode.h:56-65). */
typedef struct {
  float if_rms;
  float baseband_mean;        /* FM: get_tuning_offset() = baseband_mean * 75000 */
  float baseband_level;
  double pilot_level;         /* = 2 * m_pilot_level */
  int stereo_detected;
  float if_agc_gain;
  double af_agc_gain;         /* AM only */
  double multipath_error;
  double pll_freq_err;
  uint32_t multipath_resets;  /* blocks whose equaliser output was discarded */
  /* time-parallel recurrences of the most recent call: Newton
   * rounds used, and whether the serial fallback kernel had to run */
  int agc_iterations, pll_iterations, agc_fallback, pll_fallback;
  double pll_residual;
  float agc_residual_history[16];   /* residual after each Newton round */
  double pll_residual_history[16];
  double pll_residual_components[8];
  double pll_mismatch_history[16];  /* scaled chunk-boundary mismatch seen by each round's integration pass */
  int pll_mismatch_accepted;        /* 1: the last round was accepted on the mismatch alone (node pass skipped) */
  int af_agc_fallback;              /* AM: 1 = the audio tail (DC block / AfSimpleAgc / de-emphasis) ran in its serial form */
  uint32_t agc_sync_timeouts;       /* FM with the equaliser: times the equaliser kernel gave up waiting for the AGC kernel
                                     * that runs beside it (0 in a healthy chain; every synchronising call that sees a new one
                                     * fails with FMR_ERR_HIP: the audio of that call is void) */
} fmr_status;

typedef struct fmr_chain fmr_chain;

int fmr_create(const fmr_config *cfg, fmr_chain **out);
/* The same for a caller that may have been built against an OLDER header: cfg_size = sizeof(fmr_config) as the caller
 * knows it.  The library reads exactly that many bytes (fields the caller does not have mean "as before") and refuses a
 * size larger than its own.  fmr_create itself can only check the struct_size FIELD, which an older, shorter struct does
 * not contain. */
int fmr_create_sized(const fmr_config *cfg, size_t cfg_size, fmr_chain **out);
void fmr_destroy(fmr_chain *c);
const char *fmr_last_error(void);
const char *fmr_version(void);

/* Design introspection of the resampler stand-in (DESIGN.md "Resampler
 * specification").  which = 0:D 1:NA 2:LB 3:MB 4:TB 5:LT (rows of the interpolated
 * phase table of the fractional-phase form, 0 = one row per phase); -1 when no resampler. */
long long fmr_resampler_info(const fmr_chain *c, int which);

/* The product's resampler design on the host (no GPU needed): taps of stage A (stage = 0, NA doubles) or of the
 * polyphase stage B (stage = 1, LB x TB doubles, row = phase; (LT + 1) x TB in the fractional-phase form that ratios
 * with very large LB take, e.g. ppm-corrected source rates, main.cpp:708-711) for in_rate -> out_rate at atten_db;
 * info[0..5] receive D, NA, LB, MB, TB, LT.  Rates that are not whole hertz are taken to the millihertz.  Returns the number of taps of the stage, or a negative error (cap too small / unsupported
 * ratio).  Lets tests check the design against an independent construction without a device. */
long long fmr_design_taps(double in_rate, double out_rate, double atten_db, int stage, double *taps, long long cap,
                          long long *info);
/* The same for a resampler class (FMR_RESAMPLER_FAST / _R8B) of the IF resampler: IfResampler(in_rate, out_rate) as
 * fmr_create builds it (sfmbase/IfResampler.cpp:25-29). */
long long fmr_design_taps_class(double in_rate, double out_rate, int resampler_class, int stage, double *taps,
                                long long cap, long long *info);

/* --- live sources: page-locked host memory for the ring between a driver's callback thread and the decoder thread.
 * Replaces the heap vectors that AirspySource::callback (sfmbase/AirspySource.cpp:488-500) and RtlSdrSource::get_samples
 * (sfmbase/RtlSdrSource.cpp:359-365) fill and DataBuffer (include/DataBuffer.h:35-90) queues: the callback copies the
 * driver's RAW buffer (float pairs, or offset-binary bytes -- fmr_config.input_format converts on the GPU) into a block
 * of the ring, and fmr_process_blocks reads the block in place -- from page-locked memory the host-to-device copy is a
 * DMA at the link rate, without the runtime's pageable staging.  host/fmradion_ring.hpp holds the ring itself.
 * Returns NULL (fmr_last_error() says why) without a HIP device: there is no fallback to pageable memory. */
void *fmr_host_alloc(size_t bytes);
void fmr_host_free(void *p);

/* --- single block, host buffers: the shape of FmDecoder::process(IQSampleVector,
 * SampleVector&) (FmDecode.h:74) for stream 0 of a 1-stream chain.
 * iq: n interleaved complex float samples.  audio: doubles (interleaved L/R when
 * stereo).  */
int fmr_process(fmr_chain *c, const float *iq, size_t n, double *audio,
                size_t audio_cap, size_t *n_audio);

/* --- batched blocks, host buffers.  iq holds n_streams rows of `stream_stride`
 * complex samples; block_len[0..n_blocks) are consecutive block lengths inside
 * each row.  audio holds n_streams rows of audio_stride doubles;
 * audio_len[b] receives the number of doubles block b produced (same for all
 * streams).  Semantics = n_blocks sequential process() calls per stream. */
int fmr_process_blocks(fmr_chain *c, const float *iq, size_t stream_stride,
                       const uint32_t *block_len, int n_blocks, double *audio,
                       size_t audio_stride, uint32_t *audio_len);

/* --- same with device-resident buffers (HBM in, HBM out).
 * sync != 0: returns with the audio of this call in d_audio.
 * sync == 0: returns as soon as the call is enqueued.  THE AUDIO OF AN ASYNCHRONOUS CALL IS COMPLETE ONLY AFTER
 * fmr_synchronize() (or a later call with sync != 0, or any getter: they synchronise).  In the pipelined chain (FM with
 * the resampler, fmr_config.in_order == 0) the audio tail of call N is enqueued together with call N + 1 -- or by
 * fmr_synchronize() -- on a stream of the chain's own: waiting on the device or on a stream of yours
 * (hipDeviceSynchronize, a HIP event, torch.cuda.synchronize) does NOT make the last call's audio complete, because its
 * tail may not have been launched yet.  d_iq, d_audio and audio_len must stay valid until that synchronisation; a chain
 * destroyed before it drops the pending tail (nothing is written into d_audio after fmr_destroy returns).
 * Set fmr_config.in_order = 1 for a chain whose every call is complete on the chain's stream order. */
int fmr_process_blocks_device(fmr_chain *c, const float *d_iq,
                              size_t stream_stride, const uint32_t *block_len,
                              int n_blocks, double *d_audio, size_t audio_stride,
                              uint32_t *audio_len, int sync);
int fmr_synchronize(fmr_chain *c);

/* --- front end only: IfResampler::process (IfResampler.h:35-38).  Valid on a
 * chain created with enable_resampler; bypasses the decoder.  Host buffers. */
int fmr_resample(fmr_chain *c, const float *iq, size_t n, float *out_iq,
                 size_t out_cap, size_t *n_out);

/* --- FourthConverterIQ::process (include/FourthConverterIQ.h:38-82) on host buffers: multiply by the Fs/4 table,
 * `up` selects FourthConverterIQ(true); `index` (in/out, 0..3) is the object's m_index.  Exact (+-1, +-j swaps).
 * Decoder chains fuse the shift into their front-end kernel instead (enable_fourth_down). */
int fmr_fourth_convert(fmr_chain *c, const float *iq, size_t n, float *out_iq, int up, unsigned *index);

int fmr_get_status(fmr_chain *c, int stream, fmr_status *st);
/* ... for a caller built against an older header: st_size = sizeof(fmr_status) as the caller knows it; the library never
 * writes past it (fmr_status grows at its end). */
int fmr_get_status_sized(fmr_chain *c, int stream, void *st, size_t st_size);
/* PPS events of the most recent call (FmDecode.h:92); returns the count. */
int fmr_get_pps_events(fmr_chain *c, int stream, fmr_pps_event *ev, int cap);
/* get_multipath_coefficients (FmDecode.h:103): interleaved re,im; returns order */
int fmr_get_multipath_coefficients(fmr_chain *c, int stream, float *coeff, int cap);

/* Debug taps for stage-level parity tests: copies an intermediate vector of the
 * most recent call to the host.  which: 0 = IF samples entering the decoder
 * (complex float, 2 floats each), 1 = discriminator output (float),
 * 2 = stereo difference after demod+de-emphasis (double), 3 = mono after
 * de-emphasis (double), 4 = AGC gain sequence (float).  Returns element count, or FMR_ERR_BAD_ARG for a
 * tap the most recent call did not leave in memory: behind the fused front end's discriminator epilogue (FM at
 * 10 MS/s without IF filter / equaliser) taps 0, 1 and 4 exist only in a chain created with FMR_DEBUG_TAPS=1 in the
 * environment -- the product keeps those signals on chip. */
long long fmr_debug_read(fmr_chain *c, int stream, int which, void *out, size_t cap_bytes);

/* Kernel timing with HIP events on the chain's own streams.  enable = 1: every kernel of
 * the most recent call (diagnostics; the extra events cost host time).  enable = 2: only the
 * kernels of the FIR + discriminator stage ("ifr_fused", or "ifr_decim" / "ifr_poly" / "disc", and the IF FIR
 * "fm_block" of an FM chain), one entry per launch accumulated until queried
 * (what bench.py uses inside its timed region); enable = 4: the same on every fourth call only (the two event
 * markers of a stage kernel cost 7-10 us on the decoder stream: bench.py samples); enable = 5: as 4, and the fused front
 * end ("ifr_fused") on EVERY call.  The fused front end is timed with the start / stop events of its own dispatch
 * (hipExtLaunchKernelGGL: the command processor's time stamps of the kernel's begin and end, what rocprofv3's kernel
 * trace reads; no marker packets on the stream), the other kernels between two event markers on their stream.
 * Fills names/ms for up to cap entries, returns the count. */
int fmr_get_kernel_times(fmr_chain *c, const char **names, float *ms, int cap);
void fmr_enable_kernel_timing(fmr_chain *c, int enable);

/* enable = 3: trace.  Every instrumented kernel of every call since the mode was switched on keeps its event pair; this
 * call synchronises, fills name / stream (0 decoder, 1 side, 2 AGC [in-order chain], 4 audio tail) / start / end (ms since the
 * first traced launch) for up to cap entries, clears the trace and returns the count.  The schedule of the chain's streams
 * as the GPU ran it, without a profiler in the host's launch path (tools/step_timeline.py). */
int fmr_get_kernel_trace(fmr_chain *c, const char **names, int *streams, float *start_ms, float *end_ms, int cap);

/* Measurement aid (no counterpart in the reference): the rate at which a plain streaming-read kernel (16-byte loads,
 * every CU busy, nothing else) reads `bytes` of device memory at d_buf on this device, best of `reps` passes, in GB/s.
 * bench.py reports it next to the roofline fraction: the 8 TB/s of the roofline is the data-sheet peak, this is what
 * the box the run landed on delivers to a kernel that does nothing but read. */
int fmr_probe_read_bandwidth(int device, const void *d_buf, size_t bytes, int reps, double *gbytes_per_s);

/* Measurement aid (no counterpart in the reference): the shader clock of `device` right now, in MHz -- one wave counts
 * its compute unit's cycle counter over 20 us of the constant 100 MHz clock.  The clock ramps over tens of milliseconds
 * of load after an idle gap and the recurrence kernels of the decoder follow it: bench.py reports it on either side of
 * its timed region (what a 20-step region measures against a stream that runs continuously). */
int fmr_probe_shader_clock(int device, double *mhz);

/* Filter tables of FilterParameters (include/FilterParameters.h:31-49), by name
 * e.g. "jj1bdx_fm_384kHz_medium"; returns the length, *is_double tells the type. */
int fmr_filter_table(const char *name, const void **data, int *is_double);

#ifdef __cplusplus
}
#endif
#endif
