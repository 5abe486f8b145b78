// fmradion_amd.hip -- host engine + C-ABI (include/fmradion_amd.h) of the
// MI355X-native FM/AM demodulation chain.  One translation unit; the kernels
// are in kernels.hpp.  No CPU fallback: without a HIP device fmr_create fails.
//
// A call processes n_blocks consecutive input blocks of n_streams streams with
// the exact per-block semantics of n_blocks sequential calls of the reference
// classes (block-head FIR path, per-block lock decision, per-block equaliser
// cadence, ... SURVEY.md 8a hazards H1-H5): the time-invariant stages run over
// the whole call at once, the block structure travels as small offset tables
// the host derives from the resampler's integer output-count law.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <functional>
#include <thread>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/fmradion_amd.h"
#include "design.hpp"
#include "kernels.hpp"
#include "kernels_par.hpp"
#include "kernels_fused.hpp"
#include "kernels_decim16.hpp"

namespace {
#include "filter_tables.inc"

thread_local std::string g_err;

void set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return FMR_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

constexpr double kFmRate = 384000.0;   // FmDecoder::sample_rate_if   (FmDecode.h:38)
constexpr double kPcmRate = 48000.0;   // FmDecoder::sample_rate_pcm  (FmDecode.h:40)
constexpr double kAmRate = 48000.0;    // AmDecoder::internal_rate_pcm (AmDecode.h:36)
constexpr double kIfAtten = 140.0;     // resampler spec, DESIGN.md (FAST class)
constexpr double kR8bAtten = 180.0, kR8bPassFrac = 0.98;   // R8B class: the defaults of r8b::CDSPResampler24 (IfResampler.cpp:25-29)
constexpr double kAudioAtten = 180.0;
constexpr int FMR_MODE_NONE = -1;
// chunk lengths of the time-parallel recurrences (kernels_par.hpp)
constexpr int C_AGC = 256, C_DC = 64, C_DE = 256, C_AM = 256, C_AM_DE = 128, K_AF_ITERS = 6;
#ifndef FMR_C_PLL_MIN
#define FMR_C_PLL_MIN 32
#endif
constexpr int C_PLL_MIN = FMR_C_PLL_MIN;   // smallest PLL chunk (capacity); the actual length is c_pll
constexpr long long kSmallCall = 8192;   // IF samples: calls up to this size enqueue fewer spare Newton rounds
// ... and cut their blocks into PLL chunks of 16 samples instead of c_pll = 64: a lane integrates its chunk serially whatever
// the call's size, and ONE 65536-sample block (2517 IF samples) is 40 chunks of 64 -- one wave, 52 + 44 us for the two
// passes of a call whose whole budget is ~250 us (main.cpp:916-956 calls block by block) -- or 158 chunks of 16: three
// waves side by side, a quarter of the samples each (round 6)
constexpr int kCPllSmall = 16;
constexpr unsigned kAgcWaitTicks = 50000000u;   // 0.5 s of the 100 MHz clock: how long k_mpf4 waits for a chunk's gains (the AGC
                                                // kernel beside it is three times faster than the equaliser: it never waits in practice)
constexpr int K_AGC_ITERS = 6, K_PLL_ITERS = 4;   // PLL: 2 rounds in lock, 2 spare (an unused round is three launches that return at once: ~6 us measured)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    n = count ? count : 1;
    HIPCHK(hipMalloc((void **)&p, n * sizeof(T)));
    HIPCHK(hipMemset(p, 0, n * sizeof(T)));
    return FMR_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; }
};

struct KernelTime { const char *name; hipEvent_t a, b; };

}  // namespace

using namespace fmr;

// Diagnostic switches from the environment, read ONCE when a chain is created (INTEGRATION.md section 4 lists them);
// nothing on the call path touches the environment.
struct EnvKnobs {
  // Run-time switches (environment).  Everything else that rounds 1 and 2 compared side by side has been decided and
  // removed; what is left selects a MODE of the product or the slower form a test compares the product with.
  bool serial = false;          // FMR_SERIAL=1       serial recurrence kernels (reference loop order on one lane)
  int pipeline = -1;            // FMR_PIPELINE=0/1   the three stages of a call (front end | PLL | audio tail) of consecutive calls
                                //                    beside each other (1, the default for FM chains with the resampler) or one
                                //                    in-order chain per call (0: the form the tests compare the product with)
#ifdef FMR_AB_PARTNERS
  int test_agc_late = 0;        // FMR_TEST_AGC_LATE=ms test hook (equaliser chain): the AGC kernel beside the equaliser starts this late; -1: never
#endif
  int fe_cus = 0;               // FMR_FE_CUS=n       pipelined chain: workgroups (= CUs) the persistent front-end kernel takes
                                //                    (0: all but one per XCD)
  bool debug_taps = false;      // FMR_DEBUG_TAPS=1   keep intermediate signals readable through fmr_debug_read
  bool host_prof = false;       // FMR_HOST_PROF=1    host enqueue time per call on stderr
  bool no_fused = false;        // FMR_NO_FUSED=1     three-kernel front end, 384 k -> 48 k stage B on the vector ALUs (tests: the product against the form it replaced)
#ifdef FMR_AB_PARTNERS          // (libfmradion_amd_ab.so only: the slower forms two GPU tests compare the product with, and a test hook)
  bool pll_v1 = false;          // FMR_PLL_V1         seven launches per Newton round of the PLL instead of three: no hand-off
                                //                    between workgroups inside a launch (tests: bit-equality stress test)
#else
  static constexpr bool pll_v1 = false;
  static constexpr int test_agc_late = 0;
#endif
  double pll_rtol = -1.0;       // FMR_PLL_RTOL       PLL acceptance threshold (< 0 = default)
  bool fe_stamps = false;       // FMR_FE_STAMPS=1    diagnostics: every front-end workgroup leaves its start / end time and hardware id (fmr_debug_read 5)
  bool evt_markers = false;     // FMR_EVT_MARKERS=1  diagnostics: time the fused front end between two event MARKERS on its stream (rounds 1-5) instead of
                                //                    with the start / stop events of its own dispatch
#ifdef FMR_DIAG_KNOBS           // (diagnostic builds under tools/ only: what an A/B run varies without a rebuild)
  int x_spare_aside = -1;       // FMR_X_SPARE_ASIDE=n  the PLL's spare rounds go to the side stream for calls of <= n blocks
  int x_cpll = 0;               // FMR_X_CPLL=n         PLL chunk length
  int x_ballast = 0;            // FMR_X_BALLAST=mask   8 KB of LDS ballast (keeps a kernel off the front end's compute units): 1 lock walk, 2 commit
  int x_amtol = 0;              // FMR_X_AMTOL=n        AM: the IF AGC's acceptance from round 3 on, in 1e-6 (0: the product's 5e-6; 1000 + n: from round 2 on)
  int x_sumw = -1;              // FMR_X_SUMW=n         weight of a macro tile with partial sums, in 1/1000 above 1 (default kFusedSumWeight)
#else
  static constexpr int x_spare_aside = -1, x_cpll = 0, x_ballast = 0, x_sumw = -1, x_amtol = 0;
#endif
  static bool on(const char *n) { const char *e = getenv(n); return e && e[0] == '1'; }
  static bool set(const char *n) { return getenv(n) != nullptr; }
  void load() {
    serial = on("FMR_SERIAL"); debug_taps = on("FMR_DEBUG_TAPS");
    auto num = [](const char *n, int dflt) { const char *e = getenv(n); return (e && e[0]) ? atoi(e) : dflt; };
    pipeline = num("FMR_PIPELINE", -1); fe_cus = num("FMR_FE_CUS", 0);
    host_prof = on("FMR_HOST_PROF"); no_fused = on("FMR_NO_FUSED");
    fe_stamps = on("FMR_FE_STAMPS"); evt_markers = on("FMR_EVT_MARKERS");
#ifdef FMR_AB_PARTNERS
    test_agc_late = num("FMR_TEST_AGC_LATE", 0); pll_v1 = set("FMR_PLL_V1");
#endif
    if (const char *e = getenv("FMR_PLL_RTOL")) if (e[0]) pll_rtol = atof(e);
#ifdef FMR_DIAG_KNOBS
    x_spare_aside = num("FMR_X_SPARE_ASIDE", -1); x_cpll = num("FMR_X_CPLL", 0); x_ballast = num("FMR_X_BALLAST", 0); x_sumw = num("FMR_X_SUMW", -1); x_amtol = num("FMR_X_AMTOL", 0);
#endif
  }
};

struct fmr_chain {
  EnvKnobs env;
  fmr_config cfg{};
  int S = 1, mode = FMR_MODE_FM;
  bool has_rs = false, has_dec = true, fir_enable = false, stereo = false, pilot_shift = false;
  bool enable_mpf = false;
  hipStream_t stream = nullptr;
  // side stream: per-block bookkeeping (statistics EMAs, PLL lock logic / PPS) runs beside the
  // audio chain instead of in front of it
  bool dec_valid = true;
  bool if_valid = true;                 // the IF samples of the last call are in last_if (false: the fused front end's discriminator epilogue kept them on chip and stored |x|^2 there)
  bool gain_valid = true;               // the per-sample AGC gains of the last call are in d_gain (fmr_debug_read 4)
  bool debug_taps = false;               // FMR_DEBUG_TAPS=1: keep the de-emphasised 384 kHz signal readable (fmr_debug_read 2,3)
  DeScan de_scan{};
  DevBuf<double> d_de_pow;
  // Mismatch-based acceptance threshold of the PLL rounds, in units of the convergence scales (phase 1e-7 rad,
  // freq 1e-9, phase error 1e-5, biquad delays 1e-7 relative); env FMR_PLL_RTOL, 0 = off.  The second integration pass of
  // a call in lock sees a boundary mismatch of 8.7 .. 9.0 behind the FAST resampler class and 10.4 behind the R8B class
  // (tools/pll_mismatch.py: 7e4 after the nominal-ramp guess, 0.008 after a third pass): at 16 both are accepted there.
  // Measured against the 0.01 setting (third pass) the audio moves by 1.7e-9 RMS -- a tenth of the float32 front end's
  // own 1.8e-8 deviation from the fp64-accumulating oracle, 6000x inside the 1e-5 target -- and get_pilot_level by
  // up to 2e-6 relative.  (Round 3's 10 put the R8B class one pass -- 0.15 ms per 2^27-sample call -- behind.)
  double pll_rtol = 16.0;
  DevBuf<double> d_pll_wgr, d_pll_pre;
  DevBuf<PllSync> d_pll_sync;
  DevBuf<unsigned int> d_pll_tick2;
  int pll_jac_rounds = 1;                // rounds that re-integrate the sensitivities
  double hp_fe = 0, hp_tab = 0, hp_dec = 0; long long hp_calls = 0; bool host_prof = false;   // FMR_HOST_PROF=1
  // ---- cross-call pipelining (FM chains with the resampler; DESIGN.md section 5 "Three stages in flight").  A call is
  // three stages on three streams -- front end (fe: IfResampler + discriminator), PLL stage (stream / side / side2:
  // statistics, AGC, pilot PLL, lock logic) and audio tail (tail: de-emphasis, audio resampler, pilot cut, DC block,
  // mux) -- and stage k of call N runs beside stage k-1 of call N+1.  What one stage hands the next lives in a ring
  // of kPipe slots (IF samples, MPX, L-R, partial sums, per-block lock flags); a slot is refilled once the tail of the
  // call that used it kPipe calls ago has finished (h_marks[1], polled by the host).  Everything a stage carries from
  // call to call (halos, StreamState fields) is written by that stage only.
  hipStream_t tail = nullptr;
  bool pipelined = false;
  static constexpr int kPipe = 4;        // ring slots
  unsigned long long pipe_seq = 0;       // decoded calls issued (slot = pipe_seq % kPipe)
  int ring_prev = -1;                    // slot of the previous decoded call (-1: none yet) and its IF sample count:
  long long ring_prev_n = 0;             //   the halos of the next slot are carried over from there
  DevBuf<float2> d_if_pp[kPipe];         // slots 1 .. kPipe-1 (slot 0 = d_if / d_base / d_raw / d_fused_part / d_stereo_blk)
  DevBuf<double> d_base_pp[kPipe], d_raw_pp[kPipe];
  DevBuf<FusedPart> d_part_pp[kPipe];
  DevBuf<int> d_stereo_pp[kPipe];
  float2 *if_slot(int q) { return q ? d_if_pp[q].p : d_if.p; }
  fm_mpx_t *base_slot(int q) { return reinterpret_cast<fm_mpx_t *>(q ? d_base_pp[q].p : d_base.p); }   // FM: the MPX as floats (the allocation is shared with the 48 kHz modes' double signal)
  double *raw_slot(int q) { return q ? d_raw_pp[q].p : d_raw.p; }
  FusedPart *part_slot(int q) { return q ? d_part_pp[q].p : d_fused_part.p; }
  int *stereo_slot(int q) { return q ? d_stereo_pp[q].p : d_stereo_blk.p; }
  float2 *last_if = nullptr;
  hipEvent_t ev_fe[kPipe] = {};
  bool disc_commit_on_side = false;      // the last decoded call's discriminator phase is committed by its k_stats (side stream)
  int sync_all() {                       // every stream of the chain is idle afterwards
    if (int rc = flush_tail(nullptr)) return rc;
    for (hipStream_t st : {stream, side, side2, tail})
      if (st) HIPCHK(hipStreamSynchronize(st));
    return FMR_OK;
  }
  hipStream_t side = nullptr, side2 = nullptr;   // side2: the IF AGC when it is off the critical path
  hipEvent_t ev_pll1 = nullptr;      // pipelined chain: the PLL's second pass is done (the passes after it may run on the side stream)
  hipEvent_t ev_disc = nullptr, ev_pll = nullptr, ev_stats = nullptr, ev_fin = nullptr, ev_if = nullptr, ev_agc = nullptr,
             ev_tab = nullptr, ev_mono = nullptr;
  int pll_tick2_per_stream = 0;
  bool ev_agc_live = false;            // ev_agc has been recorded by an earlier call
  // designs + counters
  ResamplerDesign rs, ars;
  ResamplerCounter rsc, arsc;
  unsigned long long abs_in = 0;          // absolute input sample index (FourthConverter phase)
  unsigned wait_multipath_blocks = 100;   // FmDecode.cpp:33
  // capacities
  size_t max_in = 0, max_mid = 0, max_if = 0, max_amid = 0, max_au = 0;
  int max_blocks = 0;
  int H_in = 0, H_mid = 0, H_if = 0, H_a = 0, H_am = 0, H_pc = 0;
  int ntaps = 0, n_pilotcut = 0, mpf_N = 0, mpf_ref = 0;
  float h_coeff0 = 0.f;                // filter_coeff[0]
  // device buffers
  DevBuf<float2> d_in, d_in_halo, d_mid, d_if, d_fir, d_mpf, d_mpf_coeff, d_mpf_state;
  // FM with the equaliser: the serial IF AGC runs beside the equaliser kernel, which follows its progress counter
  // (IF samples of this call whose gain is in HBM, per stream; zeroed at the head of every call)
  DevBuf<unsigned long long> d_agc_progress;
  std::vector<unsigned> agc_timeouts_seen;      // per stream: StreamState::agc_sync_timeouts already reported
  bool agc_beside_mpf = false;
  DevBuf<int> d_bphi, d_boff;          // stage-B per-position tap phase / sample offset (k_ifr_poly2)
  int poly2_tile = 0;                  // staged mid samples per tile, 0 = v2 kernel not applicable
  double nbfm_freq_dev = 8000.0;
  // SSB / CW / WSPR (AmDecode.cpp:83-90,107-136): mixers before and after the 2049-tap filter
  bool ssb_like = false;
  DevBuf<float2> d_ft_pre, d_ft_post;   // FineTuner tables (480 entries), empty = no mixer at that place
  unsigned ft_index = 0;                 // FineTuner::m_index, the same for every tuner of the chain
  double af_ref = 0.6, af_rate = 0.001;  // AfSimpleAgc reference / rate (AmDecode.cpp:54-66)
  int in_fmt = 0, in_bps = 8;          // source sample format (fmr_config.input_format) and its bytes per IQ sample
  bool poly3 = false;                  // stage-B v3 (Q positions per wave share the LDS reads)
  bool poly4 = false;                  // stage-B v4 (f32 MFMA, 48/125 shape)
  bool fir_mfma_fm = false;            // FM with the 127-tap IF filter (-f): k_ifr_poly4<48, 48, 127, 2, Poly4FirDiscEpi> (filter + discriminator + block sums)
  DevBuf<float> d_afrag_fir_fm;
  bool fir_mfma = false;               // the 255-tap IF FIR of the 48 kHz modes on the matrix cores (k_ifr_poly4<48, 48, 255> + k_fir_finish)
  DevBuf<float> d_afrag_fir;
  bool poly4_am = false;               // ... the 3/8 shape (384 k -> 48 k, AM / NBFM) as sixteen periods per row block: 48/128
  int poly4_am_tile = 0;
  bool poly5h = false;                 // ... on the fp16 matrix cores, three-product split (k_ifr_poly5h): the form that runs
  int poly5h_nkb = 0;
  float poly5h_inv_scale = 1.f;
  size_t poly5h_lds = 0;
  DevBuf<_Float16> d_afrag5h;
  DevBuf<float> d_afrag;               // v4: constant A fragments
  // fused front end (kernels_fused.hpp): stage A + stage B + discriminator in one persistent kernel
  bool fused_ok = false;               // the chain's shape fits (10 MS/s class, FM, cf32, no Fs/4)
  bool decim16_ok = false;             // R8B class at 10 MS/s: stage A in the fused front end's matrix-core form (k_ifr_decim16<10, 195>)
  DevBuf<unsigned short> d_dec16_afrag;
  bool r8b_disc_ok = false;            // R8B class, FM, nothing between resampler and discriminator: the discriminator is k_ifr_poly5h's epilogue
  DevBuf<float> d_run_ph;              // ... [S][workgroup][2]: phases on either side of the workgroups' run boundaries (k_poly5h_heads)
  bool fused_disc_ok = false;          // ... and nothing sits between the resampler and the discriminator (no IF FIR, no equaliser)
  DevBuf<float> d_hB_last;             // stage-B tap row of position 47
  DevBuf<unsigned short> d_fused_afragA;   // stage-A tap fragments (fp16 high / low terms, both parities)
  DevBuf<unsigned short> d_fused_afragB;   // stage-B tap fragments (fp16 high / low terms) and the inverse of their scale
  float fused_hB_inv_scale = 1.f;
  DevBuf<FusedPart> d_fused_part;
  DevBuf<float> d_fused_mid32;              // per front-end workgroup: fp32 copies of mid samples beyond fp16's range (FusedRing::at32)
  DevBuf<float> d_zero16;                   // sixteen zero bytes (fused front end: the loader's source beyond the ends of a call)
  DevBuf<unsigned long long> d_fe_stamps;   // FMR_FE_STAMPS=1: {start, end, hardware id} of every workgroup of the last fused launch
  int fe_stamps_n = 0;                      //   ... and how many workgroups that launch had
  // The dominant kernel is timed with the start / stop events of ITS OWN dispatch (hipExtLaunchKernelGGL): the time stamps
  // the command processor takes when the kernel's first wave starts and its last wave has ended -- what rocprofv3's kernel
  // trace reads -- and no marker packets on the decoder stream (two markers around a kernel cost 7-10 us of the step and
  // put their own processing time, ~4 us, into the interval).  Set by timed_on for the launch it wraps.
  hipEvent_t ext_a = nullptr, ext_b = nullptr;
  int n_cu = 256;
  DevBuf<float> d_hBp;                 // zero-padded tap rows for v3
  DevBuf<float> d_hpA;                 // stage-A taps in polyphase order [D][Q] (k_ifr_decim2)
  int qa = 0;                          // taps per phase (even), 0 = v2 kernel not applicable
  int hB_pitch = 0;                    // fractional-phase stage B: row pitch of d_hB (floats)
  DevBuf<float> d_gain, d_dec, d_hA, d_hB, d_coeff, d_atan, d_if_rms_blk, d_bb_mean_blk, d_bb_rms_blk, d_blk_ph;
  DevBuf<double> d_base, d_raw, d_am0, d_am1, d_a10, d_a11, d_pc0, d_pc1, d_audio, d_ahA, d_ahB, d_pilotcut;
  DevBuf<int> d_tab, d_mpf_ok, d_stereo_blk;
  DevBuf<StreamState> d_state;
  // time-parallel recurrences
  bool serial_mode = false;            // FMR_SERIAL=1: plain serial kernels (A/B, debugging)
  int c_pll = 64;                      // PLL chunk length (>= C_PLL_MIN; 32: 0.40 ms, 48: 0.345, 64: 0.33, 96 / 128: 0.47 per 2^27-sample call)
  int H_b = 0;                         // halo of the pre-de-emphasis buffers (>= warm-up)
  size_t max_ck = 0, max_agc_nc = 0, max_dc_nc = 0;
  DevBuf<double> d_pll_wfirst;          // start node of every integration wave's first chunk (the fused down-sweep reads it)
  DevBuf<double> d_base_de, d_raw_de, d_pll_nodes, d_pll_G, d_pll_M, d_pll_PQ, d_pll_dstart, d_pll_PQ2, d_pll_dstart2, d_pll_gres,
      d_blk_level, d_agc_M,
      d_dc_G, d_dc_start;
  DevBuf<float> d_agc_nodes, d_agc_G;
  DevBuf<unsigned int> d_agc_tick;      // k_agc_round's last-arrival ticket, one per stream (left at zero by its users)
  DevBuf<unsigned int> d_af_tick;       // ... and k_af_round's
  DevBuf<double> d_af_nodes, d_af_G, d_af_M, d_af_out;   // AmDecoder audio tail, time-parallel form
  DcCoef am_dk{};
  DevBuf<int> d_ck_wraps, d_blk_wraps, d_walk_go;
  size_t ck_copy = 0;             // elements of one copy of d_ck_wraps
  DevBuf<unsigned long long> d_ck_mask;
  int mask_words = 2;
  DevBuf<IterFlags> d_flags;
  std::vector<IterFlags> h_flags;
  // block tables: ring of pinned host slots + device slots so that queued
  // asynchronous calls never overwrite a table that is still being copied
  static constexpr int kTabSlots = 8;
  static constexpr int kMaxFusedWg = 2048;
  static constexpr int kFirBlk0Off = 1024;                // ... [1024, 2048) first block of every run of the IF filter's matrix-core kernel
  static constexpr int kFusedTile0Off = 512;             // the table's tail: [0, 512) first block of every run of the fused front end, [512, 1024) first macro tile of every run (+ the end)
  static constexpr double kFusedSumWeight = 0.10;        // what a macro tile with per-block partial sums costs more than one without (run_tables)
#ifndef FMR_FE_SPARE_CUS
#define FMR_FE_SPARE_CUS 8
#endif
  // Pipelined chain with many short streams: the front end takes whole workgroups per stream and leaves more than its 8
  // spare compute units (32 streams: 7 x 32 = 224 of 256).  The small kernels beside it then ask for 8 KB of LDS they never
  // touch: a front-end workgroup leaves 7.7 KB of its unit's LDS free, so they can only go to the units it leaves alone
  // instead of sharing one with eight front-end waves (config5: 0.523 -> 0.502 ms per step, the front end 0.273 -> 0.252).
  // With only the 8 spare units (one long stream) the same ballast costs 4% -- the side stream's kernels queue on them.
  static constexpr int kBallastBytes = 8192, kBallastMinSpare = 24;
  static constexpr int kSpareAsideMaxBlocks = 512;   // (run_fm_pll)
  int fe_spare_cus = 0;           // compute units the last front-end launch left alone
  size_t side_ballast() const { return (pipelined && fe_spare_cus >= kBallastMinSpare) ? (size_t)kBallastBytes : 0; }
  static constexpr int kFeSpareCus = FMR_FE_SPARE_CUS;      // CUs the pipelined chain's front end leaves to the kernels beside it
  int *h_tab_all = nullptr;  // pinned, kTabSlots * tab_ints
  size_t tab_ints = 0;       // 5*max_blocks block table + 3*max_ck chunk table + (max_blocks+1) first-chunk table
  unsigned long long call_seq = 0;      // calls issued
  unsigned long long *h_marks = nullptr; // pinned: [0] calls whose table copy has run, [1] pipelined calls whose decoder has finished
  std::vector<StreamState> h_state;
  // constants
  PllConst pllc{};
  Iir1Coef deemph{}, am_deemph{};
  BiquadCoef dcblock{}, am_dcblock{};
  double dc_ac[4] = {1, 0, 0, 1};       // A^C_DC of the DC-block biquad (state transition over one chunk)
  double dc_agp[6][4] = {};             // (A^(C_DC*K))^(2^k): lane-group transitions of the node scan
  float agc_init = 1.f, agc_max = 1e5f, agc_rate = 1e-4f;
  float disc_nf = 1.f, disc_bound = 1.f;
  // last call
  long long last_n_if = 0, last_n_au = 0;
  int last_nb = 0;
  int timing = 0;                       // 0 off, 1 every kernel (diagnostics), 2 the dominant kernel only
  std::vector<KernelTime> dom_times;    // mode 2: ifr_decim events accumulated over calls until queried
  std::vector<KernelTime> ktimes;
  std::vector<KernelTime> trace;        // mode 3
  std::vector<int> trace_stream;        //   0 decoder, 1 side, 2 side2, 3 front end, 4 tail
  hipEvent_t trace_base = nullptr;
  static constexpr size_t kMaxTrace = 1u << 16;      // event pairs kept until fmr_get_kernel_trace fetches them (further kernels run untraced)

  ~fmr_chain() {
#ifdef FMR_PLL_TRACE
    if (d_pll_wgr.p && getenv("FMR_PLL_TRACE_OUT")) {
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(d_pll_wgr.n);
      (void)hipMemcpy(h.data(), d_pll_wgr.p, h.size() * 8, hipMemcpyDeviceToHost);
      if (FILE *f = fopen(getenv("FMR_PLL_TRACE_OUT"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
#endif
    // a tail stage that was never enqueued (an asynchronous last call nobody synchronised) is dropped, not launched: its
    // output mux would write into the caller's audio buffer, which the caller may have freed by now
    tail_pending = false; walk_pending = false; walk_job = nullptr;
    for (hipStream_t st : {stream, side, side2, tail}) if (st) (void)hipStreamSynchronize(st);
    for (auto &k : ktimes) { (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b); }
    for (auto &k : trace) { (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b); }
    if (trace_base) (void)hipEventDestroy(trace_base);
    for (auto &k : dom_times) { (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b); }
    d_in.release(); d_in_halo.release(); d_mid.release(); d_if.release(); d_fir.release();
    d_mpf.release(); d_mpf_coeff.release(); d_mpf_state.release(); d_gain.release(); d_dec.release(); d_agc_progress.release();
    d_hA.release(); d_hB.release(); d_coeff.release(); d_atan.release(); d_if_rms_blk.release();
    d_bb_mean_blk.release(); d_bb_rms_blk.release(); d_blk_ph.release(); d_base.release(); d_raw.release();
    d_am0.release(); d_am1.release(); d_a10.release(); d_a11.release(); d_pc0.release();
    d_pc1.release(); d_audio.release(); d_ahA.release(); d_ahB.release(); d_pilotcut.release();
    d_ft_pre.release(); d_ft_post.release(); d_hB_last.release(); d_zero16.release(); d_fe_stamps.release(); d_fused_mid32.release(); d_fused_afragA.release(); d_fused_afragB.release(); d_fused_part.release(); d_afrag.release(); d_afrag_fir_fm.release(); d_dec16_afrag.release(); d_run_ph.release(); d_afrag_fir.release(); d_afrag5h.release(); d_hBp.release(); d_hpA.release(); d_bphi.release(); d_boff.release(); d_tab.release(); d_mpf_ok.release(); d_stereo_blk.release(); d_state.release();
    d_base_de.release(); d_raw_de.release(); d_pll_nodes.release(); d_pll_G.release(); d_pll_M.release();
    d_pll_wgr.release(); d_pll_pre.release(); d_pll_wfirst.release(); d_pll_sync.release(); d_pll_tick2.release(); d_ck_mask.release(); d_walk_go.release(); d_pll_gres.release(); d_pll_PQ2.release(); d_pll_dstart2.release(); d_pll_PQ.release(); d_pll_dstart.release(); d_blk_level.release(); d_blk_wraps.release(); d_agc_M.release(); d_agc_tick.release(); d_af_tick.release(); d_dc_G.release(); d_dc_start.release(); d_agc_nodes.release();
    d_agc_G.release(); d_ck_wraps.release(); d_flags.release();
    d_af_nodes.release(); d_af_G.release(); d_af_M.release(); d_af_out.release();
    if (h_tab_all) (void)hipHostFree(h_tab_all);
    if (h_marks) (void)hipHostFree(h_marks);
    for (hipEvent_t e : {ev_disc, ev_pll, ev_stats, ev_fin, ev_if}) if (e) (void)hipEventDestroy(e);
    if (side2 == side) side2 = nullptr;      // (pipelined chain: the AGC runs on the side stream)
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (host_prof && hp_calls)
      fprintf(stderr, "[fmr host prof] calls %lld  front-end %.1f us  tables %.1f us  decoder %.1f us per call\n", hp_calls,
              hp_fe / hp_calls, hp_tab / hp_calls, hp_dec / hp_calls);
    if (side2 && side2 != side) { (void)hipStreamSynchronize(side2); (void)hipStreamDestroy(side2); }
    if (tail) (void)hipStreamDestroy(tail);
    for (auto &e : ev_fe) if (e) (void)hipEventDestroy(e);
    for (auto &b : d_if_pp) b.release();
    for (auto &b : d_base_pp) b.release();
    for (auto &b : d_raw_pp) b.release();
    for (auto &b : d_part_pp) b.release();
    for (auto &b : d_stereo_pp) b.release();
    if (ev_agc) (void)hipEventDestroy(ev_agc);
    if (ev_pll1) (void)hipEventDestroy(ev_pll1);
    if (ev_tab) (void)hipEventDestroy(ev_tab);
    if (ev_mono) (void)hipEventDestroy(ev_mono);
    if (stream) (void)hipStreamDestroy(stream);
  }

  // ---- kernel launch with optional HIP-event timing on the chain's stream ----
  // FMR_FE_STAMPS=1: where on the constant clock a stream reached the end of one of its kernels, in a ring of eight calls
  // (slot 16 (call_seq % kStampCalls) + id behind the workgroup stamps; id 0 / 1: in front of / behind the fused launch; a second
  // ring of the same shape behind the first holds the shader clock [kHz] the stamp in front of the launch measured)
  static constexpr int kStampCalls = 32;
  unsigned long long *stamp_slot(int id) { return d_fe_stamps.p + 3 * (size_t)kMaxFusedWg * S + 16 * (size_t)(call_seq % kStampCalls) + id; }
  void stamp_after(hipStream_t st, const char *name) {
    static const char *const ids[] = {"", "", "deemph_decim", "aud_poly", "pilotcut", "dc_pass1", "fm_out", "pll", "stats", "if_agc", "pll_commit", "pll_finish", "pll_shoot_jac"};
    for (int i = 2; i < 13; i++)
      if (std::strcmp(name, ids[i]) == 0) { hipLaunchKernelGGL(k_fused_stamp, dim3(1), dim3(1), 0, st, stamp_slot(i), 0); return; }
  }
  template <class F>
  void timed_on(hipStream_t st, const char *name, F &&launch_) {
    auto launch = [&] { launch_(); if (d_fe_stamps.p) stamp_after(st, name); };
    // mode 2: only the kernels of the FIR+discriminator stage carry events (two per kernel per call)
    const bool stage_kernel = std::strcmp(name, "ifr_decim") == 0 || std::strcmp(name, "ifr_poly") == 0 ||
                              std::strcmp(name, "disc") == 0 || std::strcmp(name, "ifr_fused") == 0 ||
                              std::strcmp(name, "blk_reduce") == 0 ||
                              (mode == FMR_MODE_FM && std::strcmp(name, "fm_block") == 0);   // FM with the IF FIR on
    // mode 4: the same on every fourth call only (two event markers on the decoder stream cost 7-10 us of a 0.56 ms step)
    // mode 5: as mode 4, but the fused front end on EVERY call: it is timed with the events of its own dispatch, which put no
    // markers on the stream
    const bool own_events = !env.evt_markers && std::strcmp(name, "ifr_fused") == 0;
    const bool sampled = timing == 2 || ((timing == 4 || timing == 5) && (call_seq & 3ull) == 0);
    if (stage_kernel && (sampled || (timing == 5 && own_events))) {
      KernelTime kt{name, nullptr, nullptr};
      (void)hipEventCreate(&kt.a);
      (void)hipEventCreate(&kt.b);
      if (own_events) {
        ext_a = kt.a; ext_b = kt.b;
        launch();
        ext_a = ext_b = nullptr;
      } else {
        (void)hipEventRecord(kt.a, st);
        launch();
        (void)hipEventRecord(kt.b, st);
      }
      dom_times.push_back(kt);
      return;
    }
    if (timing == 3 && trace.size() < kMaxTrace) {      // trace: every instrumented kernel of every call since the mode was switched on, with its stream
      if (!trace_base) { (void)hipEventCreate(&trace_base); (void)hipEventRecord(trace_base, st); }
      KernelTime kt{name, nullptr, nullptr};
      (void)hipEventCreate(&kt.a);
      (void)hipEventCreate(&kt.b);
      (void)hipEventRecord(kt.a, st);
      launch();
      (void)hipEventRecord(kt.b, st);
      trace.push_back(kt);
      trace_stream.push_back(st == stream ? 0 : st == side ? 1 : st == side2 ? 2 : 4);
      return;
    }
    if (timing != 1) { launch(); return; }
    KernelTime kt{name, nullptr, nullptr};
    (void)hipEventCreate(&kt.a);
    (void)hipEventCreate(&kt.b);
    (void)hipEventRecord(kt.a, st);
    launch();
    (void)hipEventRecord(kt.b, st);
    ktimes.push_back(kt);
  }
  template <class F>
  void timed(const char *name, F &&launch) { timed_on(stream, name, launch); }

  // poll a counter in pinned host memory that a one-thread kernel advances (k_signal_host)
  int wait_mark(unsigned long long *mark, unsigned long long need) {
    if (need == 0) return FMR_OK;
    const auto t_lim = std::chrono::steady_clock::now() + std::chrono::seconds(60);
    while (__atomic_load_n(mark, __ATOMIC_ACQUIRE) < need) {
      std::this_thread::yield();
      if (std::chrono::steady_clock::now() > t_lim) { set_err("the GPU did not reach mark %llu within 60 s", need); return FMR_ERR_HIP; }
    }
    return FMR_OK;
  }
  // Equaliser chain: k_mpf4 counts the waits for the AGC kernel it gave up (StreamState::agc_sync_timeouts).  Called by
  // the entry points that synchronise: a new time-out is an error of that call (its audio is void).
  int check_agc_sync() {
    if (!enable_mpf || !d_agc_progress.p) return FMR_OK;
    if (agc_timeouts_seen.size() != (size_t)S) agc_timeouts_seen.assign(S, 0u);
    int bad = -1;
    for (int s = 0; s < S; s++) {
      unsigned v = 0;
      HIPCHK(hipMemcpy(&v, reinterpret_cast<const char *>(d_state.p + s) + offsetof(StreamState, agc_sync_timeouts), sizeof v, hipMemcpyDeviceToHost));
      if (v != agc_timeouts_seen[s]) { agc_timeouts_seen[s] = v; bad = s; }
    }
    if (bad >= 0) {
      set_err("stream %d: the equaliser gave up waiting for the AGC kernel beside it (%u time-outs so far); the audio of this call is void", bad, agc_timeouts_seen[bad]);
      return FMR_ERR_HIP;
    }
    return FMR_OK;
  }
  int init(const fmr_config *c);
  bool cold = true;                     // no call yet: AGC at its initial gain, PLL unlocked
  int pps_block_base = 0;               // blocks of the call that ran before the part whose PPS events the state holds
  int run_cold_aware(const float2 *d_iq, size_t stride, const uint32_t *block_len, int nb, double *d_aud,
                     size_t astride, uint32_t *audio_len);
  // values one call's stages hand each other (run() fills the head, every stage adds its part)
  struct FusedGeom { long long mA_prev, kB_prev, n_prev; int count_mid; };
  struct CallCtx {
    const float2 *d_iq = nullptr;
    size_t stride{};
    const uint32_t *block_len = nullptr;
    int nb{};
    double *d_aud = nullptr;
    size_t astride{};
    uint32_t *audio_len = nullptr;
    long long N_in{};
    int slot{};
    int *h_tab = nullptr;
    int *d_tab_slot = nullptr;
    int *t_if_off = nullptr;
    int *t_if_len = nullptr;
    int *t_au_off = nullptr;
    int *t_au_len = nullptr;
    int *t_mpf = nullptr;
    long long N_if{};
    bool use_fused{};
    bool fused_disc{};     // the fused kernel's epilogue is the discriminator (else: IF samples only)
    bool fir_disc{};       // the IF filter kernel's epilogue is the discriminator (k_fm_block3<.., true>)
    bool tail_deferred{};  // the previous call's tail stage is still to be enqueued (behind this call's IF filter kernel)
    bool fir_tail{};       // FM -f: the IF filter on the matrix cores with the discriminator epilogue (fir_disc is set too; block sums as FusedPart)
    int fir_grid{}, fir_tpw{}, fir_rem{}, fir_tiles{}, fir_part_from{};
    bool r8b_tail{};       // R8B class: stage B is launched from run_tables with the discriminator as its epilogue (fused_disc is set too)
    long long r8b_mA_prev{}, r8b_kB_prev{};
    int r8b_count_mid{};
    FusedGeom fused_geom{};
    int par{};
    float2 *ifbuf = nullptr;
    HaloTable ht{};
    long long N_au{};
    bool any_mpf{};
    long long amA_prev{};
    long long akB_prev{};
    long long an_prev{};
    int nck{};
    int fused_n_tiles{};
    int fused_kb_ref{};
    ChunkTab ct{};
    bool iter_on_side{};
    BlockTab bt{};
    long long if_stride{};
    bool rms_in_disc{};
    const float2 *xin = nullptr;
    long long x_stride{};
    int x_off{};
    const float *disc_gain = nullptr;
    bool agc_on_side{};
    bool agc_deferred{};
    std::function<int(hipEvent_t)> enqueue_agc{};
    bool done = false;                 // the front end found nothing to decode
    hipEvent_t ev_mpx = nullptr;       // recorded where this call's MPX (discriminator output) is complete
    std::function<void()> fe_post{};   // pipelined chain: the front-end stage's end-of-call kernel, when it is still to be launched
    long long count_mid_call{};        // stage-A outputs of this call
    // this call's slot of the rings the stages hand each other (plain chain: the one buffer of each kind)
    fm_mpx_t *base = nullptr;
    double *raw = nullptr;
    float *nrm = nullptr;              // fused front end with the discriminator epilogue: |x|^2 of the IF samples (in the IF slot's memory) instead of the IF samples
    long long nrm_stride{};
    FusedPart *part = nullptr;
    int *stereo_blk = nullptr;
    void add_halo(void *buf, long long stride_e, int H, long long N, int words = 2) {   // history to move to the buffer heads at the end of the call (words: 32-bit words per element)
      if (H > 0 && N > 0) ht.d[ht.n++] = HaloDesc{(unsigned *)buf, stride_e * words, H * words, (int)N * words};
    }
  };
  // what the audio tail of one call needs, by value: in the pipelined chain the tail stage is enqueued a call later
  struct TailCtx {
    fm_mpx_t *base = nullptr;
    double *raw = nullptr;
    int *stereo_blk = nullptr;
    long long N_if{}, N_au{}, a_top0{}, amA_prev{}, akB_prev{}, astride{};
    int nb{}, count_am{}, de_tout{}, dc_nc{}, nch{};
    int au_max{};                // longest audio block of the call
    bool de_fused{}, fin_on_side{}, fin_covers_all{}, agc_on_side{}, mono_enqueued{};
    bool can_split{};            // the mono channel can be enqueued by itself (the shapes the per-channel tail kernels take)
    hipEvent_t ev_mpx = nullptr; // "the MPX of this call is there" (pipelined chain: what a drained tail's mono channel waits for)
    BlockTab bt{};
    double *d_aud = nullptr;
    DcCoef dk{};
    HaloTable ht{};
    unsigned long long seq{};
  };
  static constexpr int kDeBlock = 256;
  TailCtx tail_job{};
  bool tail_pending = false;
  // Pipelined chain: the lock logic's walk over the blocks of call N (k_pll_finish) is enqueued a call late -- on the side
  // stream behind the tables of call N+1, with the tail of call N whose output mux waits for it (flush_tail), beside the
  // front end of call N+1 -- or by whatever drains the chain.  What the next PLL needs of it, k_pll_commit has committed
  // in call N (kernels_par.hpp): the next call's tables no longer wait for a one-wave walk over 2048 blocks.
  bool walk_pending = false;
  std::function<int()> walk_job{};
  int flush_walk();
  int walk_par = 0;               // which copy of the PLL's wrap counts / wrap masks / verdict the next call writes
  void tail_channels(const TailCtx &t, hipStream_t st, int ch_base, int nch_l);
  int tail_stage(const TailCtx &t, hipStream_t ts);
  int flush_tail(hipEvent_t gate);
  int enqueue_tail(hipEvent_t gate);
  int run_front_end(CallCtx &k);
  int finish_front_end_stage(CallCtx &k);
  int run_tables(CallCtx &k);
  int run_if_stage(CallCtx &k);
  int run_fm(CallCtx &k);
  int run_fm_pll(CallCtx &k, long long base_stride, bool split_mono,
                 const std::function<void(hipStream_t, int, int)> &enqueue_tail_channels, bool &mono_enqueued,
                 bool &fin_on_side, bool &fin_covers_all);
  int run_nbfm(CallCtx &k);
  int run_am(CallCtx &k);
  int run(const float2 *d_iq, size_t stride, const uint32_t *block_len, int nb, double *d_aud,
          size_t astride, uint32_t *audio_len);
};

template <class T>
static int upload(DevBuf<T> &b, const T *src, size_t n) {
  int rc = b.alloc(n);
  if (rc) return rc;
  if (n) HIPCHK(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return FMR_OK;
}

int fmr_chain::init(const fmr_config *c) {
  env.load();
  if (env.x_cpll >= C_PLL_MIN) c_pll = env.x_cpll;
  cfg = *c;
  S = c->n_streams;
  mode = c->mode;
  if (S < 1 || c->max_block_len == 0 || c->max_blocks < 1) { set_err("bad capacity / n_streams"); return FMR_ERR_BAD_ARG; }
  if (mode != FMR_MODE_NONE && (mode < FMR_MODE_FM || mode > FMR_MODE_WSPR)) {
    set_err("mode %d is not on the hot path (FM, AM, DSB only)", mode);
    return FMR_ERR_UNSUPPORTED;
  }
  has_dec = (mode != FMR_MODE_NONE);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err("no HIP device"); return FMR_ERR_NO_DEVICE; }
  if (c->device < 0 || c->device >= ndev) { set_err("device %d out of range (%d devices)", c->device, ndev); return FMR_ERR_BAD_ARG; }
  HIPCHK(hipSetDevice(c->device));
  // The stages of consecutive calls run beside each other for FM chains with the resampler (the default; FMR_PIPELINE=0:
  // one in-order chain per call).  THREE streams ("Three stages in flight"):
  //   stream  the critical chain: front end of call N, PLL of call N, front end of call N+1 ... -- both hold the whole chip
  //           (152 KB of LDS per CU; 1258 one-wave workgroups), so they alternate anyway, and on ONE queue the hand-off
  //           between them is a packet boundary, not an event travelling between two queues (~50 us each way, measured)
  //   side    tables, statistics, the IF AGC, lock logic
  //   tail    the audio tail, a call behind
  // Three, because a process gets four hardware queues from HIP, the host application's own stream included, and more
  // busy queues than that take turns on a pipe in slices of 3.55 ms (measured with four and five busy queues, with and
  // without stream priorities: one process in two then runs at 4-36 ms per step).
  pipelined = mode == FMR_MODE_FM && c->enable_resampler != 0 && !env.serial && env.pipeline != 0 && c->in_order == 0;
  HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  if (pipelined) {
    side2 = side;
    HIPCHK(hipStreamCreateWithFlags(&tail, hipStreamNonBlocking));
    for (int q = 0; q < kPipe; q++) HIPCHK(hipEventCreateWithFlags(&ev_fe[q], hipEventDisableTiming));
  } else {
    HIPCHK(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking));
  }
  HIPCHK(hipEventCreateWithFlags(&ev_agc, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ev_pll1, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ev_tab, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&ev_mono, hipEventDisableTiming));
  for (hipEvent_t *e : {&ev_disc, &ev_pll, &ev_stats, &ev_fin, &ev_if}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  double dec_rate = (mode == FMR_MODE_FM || mode == FMR_MODE_NONE) ? kFmRate : kAmRate;
  if (mode == FMR_MODE_NONE && c->output_rate > 0) dec_rate = c->output_rate;     // IfResampler(in, out), IfResampler.h:35
  has_rs = c->enable_resampler != 0;
  max_blocks = c->max_blocks;
  max_in = c->max_block_len * (size_t)c->max_blocks;
  max_in = (max_in + 15) & ~(size_t)15;          // row pitch keeps every stream 16-byte aligned in every format
  in_fmt = c->input_format;
  if (in_fmt < 0 || in_fmt > 3) { set_err("input_format %d unknown", in_fmt); return FMR_ERR_BAD_ARG; }
  in_bps = in_fmt == 0 ? 8 : in_fmt == 1 ? 4 : 2;
  if (in_fmt != 0 && !c->enable_resampler) {
    set_err("input_format != cf32 is converted inside the front-end kernel: enable_resampler must be set");
    return FMR_ERR_UNSUPPORTED;
  }
  if (has_rs) {
    if (c->resampler_class != FMR_RESAMPLER_FAST && c->resampler_class != FMR_RESAMPLER_R8B) {
      set_err("unknown resampler_class %d", c->resampler_class);
      return FMR_ERR_BAD_ARG;
    }
    const bool r8b = c->resampler_class == FMR_RESAMPLER_R8B;
    if (!(r8b ? rs.design(c->input_rate, dec_rate, kR8bAtten, kR8bPassFrac, true) : rs.design(c->input_rate, dec_rate, kIfAtten))) {
      set_err("resampling ratio %.9g -> %.9g is outside the supported design range", c->input_rate, dec_rate);
      return FMR_ERR_UNSUPPORTED;
    }
    if (rs.D == 1) {  // stage A degenerates to a copy (one unit tap)
      rs.NA = 1; rs.hA.assign(1, 1.0);
    }
    if (sizeof(float2) * ((size_t)64 * rs.D + rs.NA - 1) > 60000) {
      set_err("decimation %d of the front end is outside the supported range", rs.D);
      return FMR_ERR_UNSUPPORTED;
    }
    {
      // the generic stage-B kernels stage a window of (255 MB / LB + TB + 6) mid samples in LDS (64 KB without an attribute)
      const unsigned long long span = (255ull * (unsigned long long)rs.MB) / (unsigned long long)rs.LB + (unsigned long long)rs.TB + 6;
      if (span * sizeof(float2) > 65536) {
        set_err("resampling ratio %.9g -> %.9g: stage B needs %d taps per phase at this ratio and class, more than the kernels stage",
                c->input_rate, dec_rate, rs.TB);
        return FMR_ERR_UNSUPPORTED;
      }
    }
    H_in = rs.NA - 1 + rs.D;
    H_mid = rs.TB;
    max_mid = max_in / rs.D + 2;
    max_if = (size_t)((double)max_in * rs.L / rs.M) + 4;
    std::vector<float> fa(rs.hA.begin(), rs.hA.end()), fb(rs.hB.begin(), rs.hB.end());
    if (rs.LT) {            // fractional-phase form: rows padded to a multiple of four taps (float4 row loads)
      hB_pitch = (rs.TB + 3) & ~3;
      fb.assign((size_t)(rs.LT + 1) * hB_pitch, 0.f);
      for (int p = 0; p <= rs.LT; p++)
        for (int j = 0; j < rs.TB; j++) fb[(size_t)p * hB_pitch + j] = (float)rs.hB[(size_t)p * rs.TB + j];
    }
    int rc;
    if ((rc = upload(d_hA, fa.data(), fa.size()))) return rc;
    if ((rc = upload(d_hB, fb.data(), fb.size()))) return rc;
    if (rs.D >= 2) {
      int q = (rs.NA + rs.D - 1) / rs.D;
      if (q & 1) q++;
      if (q <= 24) {
        qa = (q <= 16) ? 16 : 24;          // 24: stage A of the R8B class (195 taps at D = 10), cf32 input only
        std::vector<float> hp((size_t)rs.D * qa, 0.f);
        for (int k = 0; k < rs.NA; k++) hp[(size_t)(k % rs.D) * qa + k / rs.D] = fa[k];
        if ((rc = upload(d_hpA, hp.data(), hp.size()))) return rc;
      }
    }
    if (rs.D == 10 && rs.NA == 195 && in_fmt == 0 && !c->enable_fourth_down && !env.no_fused && !env.serial) {
      bool sym = true;
      for (int k = 0; sym && k < rs.NA / 2; k++) sym = (fa[k] == fa[rs.NA - 1 - k]);
      if (sym) {
        using SH16 = Decim16Shape<10, 195>;
        std::vector<unsigned short> fr((size_t)2 * SH16::NKT * 2 * 64 * 8);
        decim16_make_afragA<10, 195>(fa.data(), fr.data());
        if ((rc = upload(d_dec16_afrag, fr.data(), fr.size()))) return rc;
        if ((rc = d_zero16.alloc(4))) return rc;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_decim16<10, 195, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, SH16::LDS_BYTES));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_decim16<10, 195, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, SH16::LDS_BYTES));
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        decim16_ok = true;
      }
    }
    if (in_fmt != 0) {
      // the fused sample conversion lives in the v2 front-end kernel only: refuse the chain now, not on every call
      constexpr int BL2 = 128, T2 = 2 * BL2;
      int s_pad = T2 + 16;
      while ((s_pad & 15) != 2) s_pad++;
      const size_t lds2 = sizeof(float2) * ((size_t)rs.D * s_pad + 2);
      if (!(qa == 16 && lds2 <= 64000 && (size_t)rs.D * (T2 + 16) <= (size_t)2 * 16 * BL2)) {
        set_err("input_format != cf32 needs the FAST resampler class and a source rate with integer pre-decimation >= 2 (%.0f -> %.0f Hz gives D = %d): "
                "convert on the host or use cf32 input", c->input_rate, dec_rate, rs.D);
        return FMR_ERR_UNSUPPORTED;
      }
    }
    if (!rs.LT) {
      // stage-B v2: 64 periods per tile must fit in LDS, odd MB keeps the lane stride conflict-free
      std::vector<int> phi((size_t)rs.LB), off((size_t)rs.LB);
      for (long long q = 0; q < rs.LB; q++) { phi[q] = (int)((q * rs.MB) % rs.LB); off[q] = (int)((q * rs.MB) / rs.LB); }
      const long long tl = 64 * rs.MB + off[rs.LB - 1] + rs.TB;
      if (tl * 8 <= 98304 && rs.LB <= 4096) {
        poly2_tile = (int)tl;
        if ((rc = upload(d_bphi, phi.data(), phi.size()))) return rc;
        if ((rc = upload(d_boff, off.data(), off.size()))) return rc;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly2<512>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        // v3: Q = 4 consecutive positions per wave; their window shift must fit the zero padding
        constexpr int Q3 = 4;
        int dmax = 0;
        for (long long g = 0; g * Q3 < rs.LB; g++)
          dmax = std::max(dmax, off[std::min<long long>(g * Q3 + Q3 - 1, rs.LB - 1)] - off[g * Q3]);
        if (dmax <= FMR_POLY_PADZ - 8 && (tl + 64) * 8 <= 98304) {
          const int TBP = rs.TB + 2 * FMR_POLY_PADZ;
          std::vector<float> hp((size_t)rs.LB * TBP, 0.f);
          for (long long r = 0; r < rs.LB; r++)
            for (int j = 0; j < rs.TB; j++) hp[(size_t)r * TBP + FMR_POLY_PADZ + j] = fb[(size_t)r * rs.TB + j];
          if ((rc = upload(d_hBp, hp.data(), hp.size()))) return rc;
          poly2_tile = (int)tl + 64;       // slack: the last 8-sample step may run past the union window
          poly3 = true;
          HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly3<384, Q3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        }
        if (rs.LB == 48 && rs.MB == 125 && rs.TB != 210 && (rs.TB & 1) == 0) {
          // any other stage-B length at 48 / 125 (R8B class: TB = 3122): dense product on the fp16 matrix cores, A fragments
          // streamed through LDS (k_ifr_poly5h)
          {
            poly2_tile = (int)tl + 64;
            // the fp16 three-product form (k_ifr_poly5h): A fragments [k-block of 32 taps][row tile][h | l][lane][8], the
            // taps scaled by the power of two that puts the largest into [512, 1024)
            {
              constexpr int KCH = FMR_POLY5H_KCH;
              const int nkb = (((off[47] + rs.TB + 31) / 32 + KCH - 1) / KCH) * KCH;
              double tmax = 0.0;
              for (float v : fb) tmax = std::max(tmax, (double)std::fabs(v));
              int ea = 0;
              if (tmax > 0.0) { int e2; (void)std::frexp(tmax, &e2); ea = 10 - e2; }      // tmax 2^ea in [512, 1024)
              const float sa = std::ldexp(1.0f, ea);
              std::vector<_Float16> ah((size_t)nkb * 3 * 2 * 64 * 8, (_Float16)0.f);
              for (int kb = 0; kb < nkb; kb++)
                for (int mt = 0; mt < 3; mt++)
                  for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 8; e++) {
                      const int pp = 16 * mt + (l & 15), m = 32 * kb + 8 * (l >> 4) + e, j = m - off[pp];
                      if (j < 0 || j >= rs.TB) continue;
                      const float t = fb[(size_t)phi[pp] * rs.TB + j] * sa;
                      const _Float16 hi = (_Float16)t;
                      const size_t base = ((((size_t)kb * 3 + mt) * 2) * 64 + l) * 8 + e;
                      ah[base] = hi;
                      ah[base + 64 * 8] = (_Float16)(t - (float)hi);
                    }
              const size_t x_len = (size_t)((((int)tl + 64 + 127) / 128) * 128 + 96);
              const size_t lds5h = 8 * x_len + 2 * (size_t)KCH * 3 * 2 * 64 * 16;      // (planes + two A chunks; the results are staged over the A buffers)
              if (lds5h <= 160 * 1024 - 64 && (size_t)63 * 125 + 32 * (size_t)nkb <= x_len && x_len <= 48 * 256) {
                if ((rc = upload(d_afrag5h, ah.data(), ah.size()))) return rc;
                poly5h = true; poly5h_nkb = nkb; poly5h_inv_scale = std::ldexp(1.0f, -ea); poly5h_lds = lds5h;
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly5h<48, 125>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5h));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly5h<48, 125, Poly5hDiscEpi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5h));
                r8b_disc_ok = mode == FMR_MODE_FM && in_fmt == 0 && !c->fmfilter_enable && c->multipath_stages == 0 && !env.no_fused;
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
              }
            }
          }
        }
        if (poly3 && rs.LB == 3 && rs.MB == 8 && rs.TB == 214 && !env.no_fused) {
          // 384 k -> 48 k (config 3): sixteen periods of 3 outputs / 8 mid samples are one period of 48 / 128 -- output
          // 3 a + b of it starts (8 (3 a + b)) / 3 = 8 a + off[b] samples in and takes phase row phi[b] -- so the banded
          // matrix-core kernel of the 48/125 shape serves it (round 6: k_ifr_poly3 took 0.147 ms per 2.1 M IF samples)
          using SH = Poly4Shape<48, 128, 214>;
          std::vector<float> af((size_t)SH::MT * SH::NK * 64, 0.f);
          for (int mt = 0; mt < SH::MT; mt++)
            for (int i = 0; i < SH::nks(mt); i++)
              for (int l = 0; l < 64; l++) {
                const int pp = 16 * mt + (l & 15), m = 4 * (SH::ks_lo(mt) + i) + (l >> 4), j = m - SH::off(pp);
                if (j >= 0 && j < rs.TB) af[((size_t)mt * SH::NK + i) * 64 + l] = fb[(size_t)phi[pp % 3] * rs.TB + j];
              }
          if ((rc = upload(d_afrag, af.data(), af.size()))) return rc;
          poly4_am = true;
          poly4_am_tile = 64 * 128 + SH::off(47) + rs.TB + 64;
          HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 128, 214>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        }
        if (poly3 && rs.LB == 48 && rs.MB == 125 && rs.TB == 210) {
          using SH = Poly4Shape<48, 125, 210>;
          std::vector<float> af((size_t)SH::MT * SH::NK * 64, 0.f);
          for (int mt = 0; mt < SH::MT; mt++)
            for (int i = 0; i < SH::nks(mt); i++)
              for (int l = 0; l < 64; l++) {
                const int pp = 16 * mt + (l & 15), m = 4 * (SH::ks_lo(mt) + i) + (l >> 4), j = m - off[pp];
                if (j >= 0 && j < rs.TB) af[((size_t)mt * SH::NK + i) * 64 + l] = fb[(size_t)phi[pp] * rs.TB + j];
              }
          if ((rc = upload(d_afrag, af.data(), af.size()))) return rc;
          poly4 = true;
          HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 125, 210>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
          // fused front end: the 10 MS/s shape (D = 10, NA = 103) with a symmetric stage-A filter, FM, cf32 input, no
          // Fs/4 shift.  Without IF FIR and equaliser the discriminator reads the IF directly and is the kernel's
          // epilogue; with either of them the epilogue stores the IF samples and the chain goes on as usual.
          bool sym = rs.D == kFusedD && rs.NA == kFusedNA;
          for (int k = 0; sym && k < rs.NA / 2; k++) sym = (fa[k] == fa[rs.NA - 1 - k]);
          if (sym && mode == FMR_MODE_FM && in_fmt == 0 && !c->enable_fourth_down && !env.no_fused) {
            if ((rc = upload(d_hB_last, fb.data() + (size_t)phi[47] * rs.TB, (size_t)rs.TB))) return rc;
            { std::vector<unsigned short> fr((size_t)2 * 8 * 2 * 64 * 8);
              fused_make_afragA<kFusedD, kFusedNA>(fa.data(), fr.data());
              if ((rc = upload(d_fused_afragA, fr.data(), fr.size()))) return rc;
              std::vector<unsigned short> frb((size_t)3 * 9 * 2 * 64 * 8);
              fused_hB_inv_scale = fused_make_afragB(fb.data(), frb.data());
              if ((rc = upload(d_fused_afragB, frb.data(), frb.size()))) return rc; }
            constexpr int kL = FusedShape<kFusedD, kFusedNA>::LDS_BYTES;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_fused<kFusedD, kFusedNA, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kL));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_fused<kFusedD, kFusedNA, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kL));
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
            fused_ok = true;
            fused_disc_ok = !c->fmfilter_enable && c->multipath_stages == 0;
          }
        }
      }
    }
    if ((rc = d_in_halo.alloc((size_t)S * H_in))) return rc;
    if ((rc = d_mid.alloc((size_t)S * (H_mid + max_mid)))) return rc;
  } else {
    max_if = max_in;
  }
  int rc;
  if ((rc = d_in.alloc((size_t)S * max_in))) return rc;
  ntaps = c->n_filter_coeff;
  ssb_like = (mode == FMR_MODE_USB || mode == FMR_MODE_LSB || mode == FMR_MODE_CW || mode == FMR_MODE_WSPR);
  const float *filter_src = c->filter_coeff;
  if (ssb_like) {     // AmDecoder's own filters: m_ssbfilter for USB/LSB, m_cwfilter for CW/WSPR (AmDecode.cpp:36,40)
    filter_src = (mode == FMR_MODE_USB || mode == FMR_MODE_LSB) ? k_jj1bdx_ssb_48khz_1500hz : k_jj1bdx_cw_48khz_500hz;
    ntaps = 2049;
  }
  fir_enable = (mode == FMR_MODE_FM) ? (c->fmfilter_enable != 0) : (mode != FMR_MODE_NONE);
  if (has_dec && (ntaps < 1 || !filter_src)) { set_err("filter_coeff missing"); return FMR_ERR_BAD_ARG; }
  H_if = has_dec ? (ntaps > 1 ? ntaps - 1 : 1) : 1;
  if ((rc = d_if.alloc((size_t)S * (H_if + max_if)))) return rc;
  last_if = d_if.p;
  host_prof = env.host_prof; debug_taps = env.debug_taps;
  if (env.pll_rtol >= 0.0) pll_rtol = env.pll_rtol;
  if (pipelined)
    for (int q = 1; q < kPipe; q++)
      if ((rc = d_if_pp[q].alloc((size_t)S * (H_if + max_if)))) return rc;
  h_state.assign(S, StreamState{});
  for (auto &st : h_state) {
    st.agc_gain = 1.0f;
    st.pll_freq = (19000.0 / kFmRate) * 2.0 * M_PI;   // PilotPhaseLock.cpp:40
    st.af_gain = 1.0;
  }
  if ((rc = upload(d_state, h_state.data(), h_state.size()))) return rc;
  max_ck = std::max(max_if / (size_t)c_pll, std::min<size_t>(max_if, (size_t)kSmallCall) / (size_t)kCPllSmall) + (size_t)max_blocks + 2;
  tab_ints = 5 * (size_t)max_blocks + 3 * max_ck + (size_t)max_blocks + 1 + kMaxFusedWg;   // tail: first block of each fused workgroup
  HIPCHK(hipHostMalloc((void **)&h_tab_all, sizeof(int) * kTabSlots * tab_ints));
  // The marks are written by one-thread kernels and polled by the host while more work is queued behind them: COHERENT
  // (fine-grained) host memory, so that the store leaves the GPU when it is made.  With the default flags the line may sit
  // in the GPU's L2 until some later system-scope release writes it back, and a host that waits for a mark sees it
  // milliseconds late (measured: one process in two ran at 4-36 ms per step, in multiples of 3.55 ms).
  HIPCHK(hipHostMalloc((void **)&h_marks, 2 * sizeof(unsigned long long), hipHostMallocCoherent));
  h_marks[0] = h_marks[1] = 0;
  if ((rc = d_tab.alloc((size_t)kTabSlots * tab_ints))) return rc;
  h_flags.assign(S, IterFlags{});
  if ((rc = d_flags.alloc((size_t)S))) return rc;
  {
    serial_mode = env.serial;
  }
  max_agc_nc = max_if / C_AGC + 2;
  if (has_dec) {
    if ((rc = d_agc_nodes.alloc((size_t)S * (max_agc_nc + 1)))) return rc;
    if ((rc = d_agc_tick.alloc((size_t)S))) return rc;
    if ((rc = d_af_tick.alloc((size_t)S))) return rc;
    if ((rc = d_agc_G.alloc((size_t)S * max_agc_nc))) return rc;
    if ((rc = d_agc_M.alloc((size_t)S * max_agc_nc))) return rc;
  }
  if (!has_dec) return FMR_OK;

  if ((rc = upload(d_coeff, filter_src, (size_t)ntaps))) return rc;
  h_coeff0 = filter_src[0];
  if (fir_enable && (rc = d_fir.alloc((size_t)S * max_if))) return rc;
  if (fir_enable && mode == FMR_MODE_FM && ntaps == 127 && c->multipath_stages == 0 && !env.no_fused && !env.serial) {
    // FM -f: lags 1 .. 126 as the 48 / 48 shape of the banded matrix-core kernel (see the 48 kHz modes below), lag 0 and the
    // discriminator in its epilogue (Poly4FirDiscEpi)
    using SH = Poly4Shape<48, 48, 127>;
    std::vector<float> af((size_t)SH::MT * SH::NK * 64, 0.f);
    for (int mt = 0; mt < SH::MT; mt++)
      for (int i = 0; i < SH::nks(mt); i++)
        for (int l = 0; l < 64; l++) {
          const int pp = 16 * mt + (l & 15), m = 4 * (SH::ks_lo(mt) + i) + (l >> 4), j = m - SH::off(pp);
          if (j >= 0 && j < 126) af[((size_t)mt * SH::NK + i) * 64 + l] = filter_src[126 - j];
        }
    if ((rc = upload(d_afrag_fir_fm, af.data(), af.size()))) return rc;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 48, 127, 2, Poly4FirDiscEpi>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    fir_mfma_fm = true;
  }
  if (fir_enable && mode != FMR_MODE_FM && !ssb_like && (ntaps == 255 || ntaps == 127) && !env.no_fused && !env.serial) {
    // AM / DSB / NBFM (255 or 127 taps): out[i] = sum_{j = 1 .. order} c[j] x[i - j] as the 1 : 1 "polyphase" shape 48 / 48 of the
    // banded matrix-core kernel -- one phase, window start = output index; tap row h[j'] = c[order - j'] over x[i - order + j'],
    // with the lag-0 tap (j' = order) left out: k_fir_finish adds it where the reference has it.  (Round 6: k_fm_block2 took
    // 0.139 ms per 2.1 M IF samples, a fifth of the AM step.)
    auto make = [&](auto sh_tag) {
      using SH = decltype(sh_tag);
      const int order = ntaps - 1;
      std::vector<float> af((size_t)SH::MT * SH::NK * 64, 0.f);
      for (int mt = 0; mt < SH::MT; mt++)
        for (int i = 0; i < SH::nks(mt); i++)
          for (int l = 0; l < 64; l++) {
            const int pp = 16 * mt + (l & 15), m = 4 * (SH::ks_lo(mt) + i) + (l >> 4), j = m - SH::off(pp);
            if (j >= 0 && j < order) af[((size_t)mt * SH::NK + i) * 64 + l] = filter_src[order - j];
          }
      return upload(d_afrag_fir, af.data(), af.size());
    };
    if (ntaps == 255) {
      if ((rc = make(Poly4Shape<48, 48, 255>{}))) return rc;
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 48, 255, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    } else {
      if ((rc = make(Poly4Shape<48, 48, 127>{}))) return rc;
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ifr_poly4<48, 48, 127, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    }
    fir_mfma = true;
  }
  if ((rc = d_gain.alloc((size_t)S * max_if))) return rc;
  if ((rc = d_dec.alloc((size_t)S * max_if))) return rc;
  if ((fused_ok || r8b_disc_ok || fir_mfma_fm) && (rc = d_fused_part.alloc((size_t)S * 3 * (max_if / 384 + 20)))) return rc;
  if ((r8b_disc_ok || fir_mfma_fm) && (rc = d_run_ph.alloc((size_t)S * 2 * kMaxFusedWg))) return rc;
  if (fused_ok && (rc = d_fused_mid32.alloc((size_t)std::max(std::max(n_cu, S), 256) * 2 * FusedShape<kFusedD, kFusedNA>::MIDR))) return rc;
  if (fused_ok && (rc = d_zero16.alloc(4))) return rc;
  if (fused_ok && env.fe_stamps && (rc = d_fe_stamps.alloc(3 * (size_t)kMaxFusedWg * S + 2 * kStampCalls * 16))) return rc;
  if ((rc = d_if_rms_blk.alloc((size_t)S * max_blocks))) return rc;
  if ((rc = d_bb_mean_blk.alloc((size_t)S * max_blocks))) return rc;
  if ((rc = d_bb_rms_blk.alloc((size_t)S * max_blocks))) return rc;
  if ((rc = d_blk_ph.alloc((size_t)S * max_blocks * 2))) return rc;
  if ((rc = d_mpf_ok.alloc((size_t)S * max_blocks))) return rc;
  if ((rc = d_stereo_blk.alloc((size_t)S * max_blocks))) return rc;

  if (mode == FMR_MODE_FM) {
    stereo = c->stereo != 0;
    pilot_shift = c->pilot_shift != 0;
    enable_mpf = c->multipath_stages > 0;
    if (c->multipath_stages > 300) { set_err("multipath_stages > 300 unsupported"); return FMR_ERR_UNSUPPORTED; }
    agc_init = 1.0f; agc_max = 100000.0f; agc_rate = 0.0001f;          // FmDecode.cpp:74
    disc_nf = (float)((75000.0 / kFmRate) * 2.0 * M_PI);                 // PhaseDiscriminator.cpp:28
    disc_bound = (float)(1.0 / ((75000.0 / kFmRate) * 2.0));             // :30
    const double freq = 19000.0 / kFmRate, bw = 30 / kFmRate;           // PilotPhaseLock.h:31-35
    pllc.minfreq = (freq - bw) * 2.0 * M_PI;
    pllc.maxfreq = (freq + bw) * 2.0 * M_PI;
    pllc.bq_b0 = 1.46974784e-06; pllc.bq_a1 = -1.99682419; pllc.bq_a2 = 0.996825659;  // PilotPhaseLock.cpp:48
    pllc.lf_b0 = 0.000304341788; pllc.lf_b1 = -0.000304324564;                          // :51
    pllc.lock_delay = int(15.0 / bw);
    pllc.pilot_frequency = 19000;
    pllc.minsignal = 0.001;
    const double de = c->deemphasis_us;
    deemph = lowpass_rc((de == 0) ? 1.0 : (de * kFmRate * 1.0e-6));      // FmDecode.cpp:67-70
    dcblock = highpass_iir(0.0001);                                       // FmDecode.cpp:62
    {
      double a16 = 1.0;
      for (int i = 0; i < FMR_DE_LPL; i++) a16 *= -deemph.a1;             // A^LPL, A = -a1 = exp(-1/tau)
      std::vector<double> apow(65);
      apow[0] = 1.0;
      for (int k = 1; k <= 64; k++) apow[k] = apow[k - 1] * a16;
      double pwr = a16;
      for (int j = 0; j < 7; j++) { de_scan.pw[j] = pwr; pwr *= pwr; }
      if ((rc = upload(d_de_pow, apow.data(), apow.size()))) return rc;
      de_scan.apow = d_de_pow.p;
    }
    {
      // A = [[-a1, -a2], [1, 0]] acts on (w[n-1], w[n-2]); A^C by repeated multiplication
      double a[4] = {-dcblock.a1, -dcblock.a2, 1.0, 0.0}, r[4] = {1, 0, 0, 1};
      for (int i = 0; i < C_DC; i++) {
        const double t[4] = {a[0] * r[0] + a[1] * r[2], a[0] * r[1] + a[1] * r[3], a[2] * r[0] + a[3] * r[2],
                             a[2] * r[1] + a[3] * r[3]};
        for (int j = 0; j < 4; j++) r[j] = t[j];
      }
      for (int j = 0; j < 4; j++) dc_ac[j] = r[j];
      auto mul = [](const double *x, const double *y, double *z) {
        const double t[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2],
                             x[2] * y[1] + x[3] * y[3]};
        for (int j = 0; j < 4; j++) z[j] = t[j];
      };
      double gk[4] = {1, 0, 0, 1};
      for (int i = 0; i < FMR_DC_K; i++) mul(dc_ac, gk, gk);          // A^(C*K)
      for (int lv = 0; lv < 6; lv++) {
        for (int j = 0; j < 4; j++) dc_agp[lv][j] = gk[j];
        mul(gk, gk, gk);
      }
    }
    if (!ars.design(kFmRate, kPcmRate, kAudioAtten)) { set_err("audio resampler design failed"); return FMR_ERR_UNSUPPORTED; }
    if (ars.D == 1) { ars.NA = 1; ars.hA.assign(1, 1.0); }
    H_a = ars.NA - 1 + ars.D;
    H_am = ars.TB;
    n_pilotcut = 127;
    H_pc = n_pilotcut - 1;
    max_amid = max_if / ars.D + 2;
    max_au = (size_t)((double)max_if * ars.L / ars.M) + 4;
    if ((rc = upload(d_ahA, ars.hA.data(), ars.hA.size()))) return rc;
    if ((rc = upload(d_ahB, ars.hB.data(), ars.hB.size()))) return rc;
    if ((rc = upload(d_pilotcut, k_jj1bdx_48khz_fmaudio, (size_t)127))) return rc;   // FmDecode.cpp:48-49
    float tab[257];
    make_fast_atan_table(tab);
    if ((rc = upload(d_atan, tab, (size_t)257))) return rc;
    H_b = FMR_DE_WARMUP + H_a;           // warm-up of the fused de-emphasis reaches below the oldest stage-A tap
    if ((rc = d_base.alloc((size_t)S * (H_b + max_if)))) return rc;
    if ((rc = d_raw.alloc((size_t)S * (H_b + max_if)))) return rc;
    if (pipelined)
      for (int q = 1; q < kPipe; q++) {
        if ((rc = d_base_pp[q].alloc((size_t)S * (H_b + max_if)))) return rc;
        if ((rc = d_raw_pp[q].alloc((size_t)S * (H_b + max_if)))) return rc;
        if ((fused_ok || r8b_disc_ok || fir_mfma_fm) && (rc = d_part_pp[q].alloc(d_fused_part.n))) return rc;
        if ((rc = d_stereo_pp[q].alloc((size_t)S * max_blocks))) return rc;
      }
    if ((rc = d_base_de.alloc((size_t)S * (H_a + max_if)))) return rc;
    if ((rc = d_raw_de.alloc((size_t)S * (H_a + max_if)))) return rc;
    if ((rc = d_pll_nodes.alloc((size_t)S * (max_ck + 1) * 7))) return rc;
    if ((rc = d_pll_G.alloc((size_t)S * max_ck * 9))) return rc;
    if ((rc = d_pll_M.alloc((size_t)S * max_ck * 49))) return rc;
    if ((rc = d_ck_wraps.alloc((size_t)S * max_ck * (pipelined ? 2 : 1)))) return rc;      // (pipelined chain: two copies, see walk_par)
    ck_copy = (size_t)S * max_ck;
    if ((rc = d_walk_go.alloc((size_t)S * 2))) return rc;
    {
      const size_t max_grp = max_ck / FMR_NODE_GRP + 2;
      if ((rc = d_pll_PQ.alloc((size_t)S * max_grp * 56))) return rc;
      if ((rc = d_pll_dstart.alloc((size_t)S * max_grp * 7))) return rc;
      if ((rc = d_pll_gres.alloc((size_t)S * max_grp * 8))) return rc;
      if ((rc = d_pll_wgr.alloc((size_t)S * (max_ck / 64 + 2) * 8))) return rc;      // (x 8: room for the FMR_PLL_TRACE records)
      const size_t max_grp2 = max_grp / FMR_NODE_GRP2 + 2;
      if ((rc = d_pll_PQ2.alloc((size_t)S * max_grp2 * 56))) return rc;
      if ((rc = d_pll_dstart2.alloc((size_t)S * max_grp2 * 7))) return rc;
      if ((rc = d_pll_pre.alloc((size_t)S * max_grp * 56))) return rc;
      if ((rc = d_pll_wfirst.alloc((size_t)S * (max_ck / 64 + 2) * 7))) return rc;
      if ((rc = d_pll_sync.alloc((size_t)S))) return rc;               // zeroed here; the kernels leave it zeroed
      if ((rc = d_pll_tick2.alloc((size_t)S * max_grp2))) return rc;
      pll_tick2_per_stream = (int)max_grp2;
    }
    mask_words = (std::max(c_pll, 128) + 63) / 64;   // wrap bit masks: one word per 64 samples of a chunk
    if ((rc = d_ck_mask.alloc((size_t)S * max_ck * mask_words * (pipelined ? 2 : 1)))) return rc;
    if ((rc = d_blk_wraps.alloc((size_t)S * max_blocks))) return rc;
    if ((rc = d_blk_level.alloc((size_t)S * max_blocks))) return rc;
    max_dc_nc = max_au / C_DC + 2;
    if ((rc = d_dc_G.alloc((size_t)S * 2 * max_dc_nc * 2))) return rc;
    if ((rc = d_dc_start.alloc((size_t)S * 2 * max_dc_nc * 2))) return rc;
    if ((rc = d_am0.alloc((size_t)S * (H_am + max_amid)))) return rc;
    if ((rc = d_am1.alloc((size_t)S * (H_am + max_amid)))) return rc;
    if ((rc = d_a10.alloc((size_t)S * (H_pc + max_au)))) return rc;
    if ((rc = d_a11.alloc((size_t)S * (H_pc + max_au)))) return rc;
    if ((rc = d_pc0.alloc((size_t)S * max_au))) return rc;
    if ((rc = d_pc1.alloc((size_t)S * max_au))) return rc;
    if ((rc = d_audio.alloc((size_t)S * 2 * max_au))) return rc;
    // MultipathFilter(m_enable ? stages : 1)  (FmDecode.cpp:79)
    const unsigned stages = enable_mpf ? c->multipath_stages : 1;
    mpf_N = (int)(stages * 4 + 1);
    mpf_ref = (int)(stages * 3 + 1);
    std::vector<float2> cf((size_t)S * mpf_N, make_float2(0.f, 0.f));
    for (int s = 0; s < S; s++) cf[(size_t)s * mpf_N + mpf_ref] = make_float2(1.f, 0.f);
    if ((rc = upload(d_mpf_coeff, cf.data(), cf.size()))) return rc;
    if ((rc = d_mpf_state.alloc((size_t)S * mpf_N))) return rc;
    if (enable_mpf && (rc = d_mpf.alloc((size_t)S * max_if))) return rc;
    if (enable_mpf && (rc = d_agc_progress.alloc((size_t)S))) return rc;
  } else if (mode == FMR_MODE_NBFM) {
    nbfm_freq_dev = (c->nbfm_freq_dev > 0) ? c->nbfm_freq_dev : 8000.0;  // NbfmDecode.h:39 freq_dev_normal
    agc_init = 1.0f; agc_max = 100000.0f; agc_rate = 0.0001f;           // NbfmDecode.cpp:43
    disc_nf = (float)((nbfm_freq_dev / kAmRate) * 2.0 * M_PI);           // NbfmDecode.cpp:35, PhaseDiscriminator.cpp:28
    disc_bound = (float)(1.0 / ((nbfm_freq_dev / kAmRate) * 2.0));
    n_pilotcut = 63;                                                      // jj1bdx_48khz_nbfmaudio, NbfmDecode.cpp:39
    H_b = n_pilotcut - 1;
    max_au = max_if;
    if ((rc = upload(d_pilotcut, k_jj1bdx_48khz_nbfmaudio, (size_t)n_pilotcut))) return rc;
    if ((rc = d_base.alloc((size_t)S * (H_b + max_if)))) return rc;
    if ((rc = d_audio.alloc((size_t)S * max_au))) return rc;
  } else {
    const bool cw_like = (mode == FMR_MODE_CW || mode == FMR_MODE_WSPR);
    agc_init = 1.0f; agc_max = 1000000.0f; agc_rate = cw_like ? 0.0006f : 0.0003f;   // AmDecode.cpp:71-77
    af_ref = ssb_like ? 0.24 : 0.6;                                       // AmDecode.cpp:54-66
    af_rate = cw_like ? 0.00125 : 0.001;
    if (ssb_like) {
      // FineTuner(table_size 480 = 48000/100, freq_shift) tables: FineTuner.cpp:25-52 with m_index = 0, phase offset 0
      auto table = [](int freq_shift) {
        const int ts = 480;
        std::vector<float2> t((size_t)ts);
        const double phase_step = 2.0 * M_PI / double(ts);
        for (int i = 0; i < ts; i++) {
          const double phi = (double)(((int64_t)freq_shift * i) % ts) * phase_step;
          t[(size_t)i] = make_float2((float)std::cos(phi), (float)std::sin(phi));
        }
        return t;
      };
      const auto up = table(15), down = table(-15), cw = table(5);       // AmDecode.cpp:83-90
      switch (mode) {
      case FMR_MODE_USB: rc = upload(d_ft_pre, down.data(), down.size()); if (!rc) rc = upload(d_ft_post, up.data(), up.size()); break;
      case FMR_MODE_LSB: rc = upload(d_ft_pre, up.data(), up.size()); if (!rc) rc = upload(d_ft_post, down.data(), down.size()); break;
      case FMR_MODE_CW: rc = upload(d_ft_post, cw.data(), cw.size()); break;
      default: rc = upload(d_ft_pre, down.data(), down.size()); if (!rc) rc = upload(d_ft_post, up.data(), up.size()); break;   // WSPR
      }
      if (rc) return rc;
    }
    am_dcblock = highpass_iir(60 / kAmRate);                              // AmDecode.cpp:45
    am_deemph = lowpass_rc(100 * kPcmRate * 1.0e-6);                      // AmDecode.cpp:49
    max_au = max_if;
    if ((rc = d_base.alloc((size_t)S * max_if))) return rc;
    if ((rc = d_audio.alloc((size_t)S * max_au))) return rc;
    {
      // time-parallel audio tail: DC-block transition matrices over one chunk and over the node scan's lane groups
      const size_t nc = max_if / C_AM + 2;
      if ((rc = d_dc_G.alloc((size_t)S * 2 * nc * 2))) return rc;
      if ((rc = d_dc_start.alloc((size_t)S * 2 * nc * 2))) return rc;
      if ((rc = d_af_nodes.alloc((size_t)S * (nc + 1)))) return rc;
      if ((rc = d_af_G.alloc((size_t)S * nc))) return rc;
      if ((rc = d_af_M.alloc((size_t)S * nc))) return rc;
      if ((rc = d_af_out.alloc((size_t)S * max_if))) return rc;
      am_dk.b0 = am_dcblock.b0; am_dk.b1 = am_dcblock.b1; am_dk.b2 = am_dcblock.b2; am_dk.a1 = am_dcblock.a1; am_dk.a2 = am_dcblock.a2;
      auto mul = [](const double *x, const double *y, double *z) {
        const double t[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2],
                             x[2] * y[1] + x[3] * y[3]};
        for (int j = 0; j < 4; j++) z[j] = t[j];
      };
      const double a[4] = {-am_dcblock.a1, -am_dcblock.a2, 1.0, 0.0};
      double r[4] = {1, 0, 0, 1};
      for (int i = 0; i < C_AM; i++) mul(a, r, r);
      for (int j = 0; j < 4; j++) am_dk.ac[j] = r[j];
      double gk[4] = {1, 0, 0, 1};
      for (int i = 0; i < FMR_DC_K; i++) mul(am_dk.ac, gk, gk);
      for (int lv = 0; lv < 6; lv++) { for (int j = 0; j < 4; j++) am_dk.agp[lv][j] = gk[j]; mul(gk, gk, gk); }
    }
  }
  return FMR_OK;
}

// A chain's first call starts from the initial AGC gain and an unlocked PLL; over that transient the Newton
// iterations of the time-parallel recurrences diverge and the serial kernels take over -- 2 s for a 2048-block
// call.  A call is by construction equal to its blocks processed one after the other, so a long first call is
// cut after ~0.8 s of signal: the head pays the serial price (~0.1 s), the rest starts locked and runs parallel.
int fmr_chain::run_cold_aware(const float2 *d_iq, size_t stride, const uint32_t *block_len, int nb, double *d_aud,
                              size_t astride, uint32_t *audio_len) {
  const bool was_cold = cold;
  pps_block_base = 0;
  auto plain = [&]() { const int rc0 = run(d_iq, stride, block_len, nb, d_aud, astride, audio_len); if (rc0 == FMR_OK) cold = false; return rc0; };
  if (!was_cold || mode != FMR_MODE_FM || !has_rs || nb < 2) return plain();
  const double target = 0.8 * cfg.input_rate;
  double total = 0;
  for (int b = 0; b < nb; b++) total += block_len[b];
  if (total < 2.0 * target) return plain();
  size_t in_off = 0;
  int k = 0;
  while (k < nb - 1 && (double)in_off < target) in_off += block_len[k++];
  if (in_fmt != 0 && (in_off * (size_t)in_bps) % 16 != 0) return plain();
  std::vector<uint32_t> al((size_t)nb, 0);
  int rc = run(d_iq, stride, block_len, k, d_aud, astride, al.data());
  if (rc) return rc;
  cold = false;
  // The head (< 0.8 s of signal from a cold PLL) cannot hold a PPS event: the first one needs the lock (0.5 s) plus
  // 19000 pilot periods (1 s).  The state keeps the tail's events; their block index is reported relative to the call.
  pps_block_base = k;
  size_t au_off = 0;
  for (int b = 0; b < k; b++) au_off += al[b];
  const float2 *iq2 = reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(d_iq) + in_off * (size_t)in_bps);
  rc = run(iq2, stride, block_len + k, nb - k, d_aud ? d_aud + au_off : nullptr, astride, al.data() + k);
  if (rc) return rc;
  if (audio_len) for (int b = 0; b < nb; b++) audio_len[b] = al[b];
  return FMR_OK;
}

int fmr_chain::run(const float2 *d_iq, size_t stride, const uint32_t *block_len, int nb, double *d_aud,
                   size_t astride, uint32_t *audio_len) {
  if (nb < 1 || nb > max_blocks) { set_err("n_blocks %d outside 1..%d", nb, max_blocks); return FMR_ERR_CAPACITY; }
  long long N_in = 0;
  for (int b = 0; b < nb; b++) {
    if (block_len[b] > cfg.max_block_len) { set_err("block %d longer than max_block_len", b); return FMR_ERR_CAPACITY; }
    N_in += block_len[b];
  }
  if (in_fmt != 0 && ((stride * (size_t)in_bps) % 16 != 0 || ((uintptr_t)d_iq % 16) != 0)) {
    set_err("raw-format input: the buffer and the stream stride must be 16-byte aligned");
    return FMR_ERR_BAD_ARG;
  }
  HIPCHK(hipSetDevice(cfg.device));
  const auto hp0 = std::chrono::steady_clock::now();
  auto hp1 = hp0, hp2 = hp0;
  struct HostProf {
    fmr_chain *c; const std::chrono::steady_clock::time_point &t0, &t1, &t2;
    ~HostProf() {
      if (!c->host_prof) return;
      const auto t3 = std::chrono::steady_clock::now();
      c->hp_fe += std::chrono::duration<double, std::micro>(t1 - t0).count();
      c->hp_tab += std::chrono::duration<double, std::micro>(t2 - t1).count();
      c->hp_dec += std::chrono::duration<double, std::micro>(t3 - t2).count();
      c->hp_calls++;
    }
  } host_prof_guard{this, hp0, hp1, hp2};
  for (auto &k : ktimes) { (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b); }
  ktimes.clear();
  // Table slot ring: the slot is free once the copy kernel of the call that used it kTabSlots calls ago has run.
  // That kernel's successor writes a counter into pinned host memory which is polled here -- hipEventSynchronize
  // would block until the newest signal of the side stream at call time (most of the PREVIOUS call) and stop the
  // host from enqueueing ahead (measured).
  call_seq++;
  const int slot = (int)(call_seq % kTabSlots);
  if (int rcw = wait_mark(&h_marks[0], call_seq > (unsigned long long)kTabSlots ? call_seq - kTabSlots : 0)) return rcw;
  int *h_tab = h_tab_all + (size_t)slot * tab_ints;
  int *d_tab_slot = d_tab.p + (size_t)slot * tab_ints;
  int *t_if_off = h_tab, *t_if_len = h_tab + max_blocks, *t_au_off = h_tab + 2 * max_blocks,
      *t_au_len = h_tab + 3 * max_blocks, *t_mpf = h_tab + 4 * max_blocks;
  // ---- the stages of one call share their per-call values through CallCtx (run_front_end ... run_am below)
  CallCtx k{};
  k.d_iq = d_iq; k.stride = stride; k.block_len = block_len; k.nb = nb; k.d_aud = d_aud; k.astride = astride;
  k.audio_len = audio_len; k.N_in = N_in; k.slot = slot; k.h_tab = h_tab; k.d_tab_slot = d_tab_slot;
  k.t_if_off = t_if_off; k.t_if_len = t_if_len; k.t_au_off = t_au_off; k.t_au_len = t_au_len; k.t_mpf = t_mpf;
  if (int rc = run_front_end(k)) return rc;
  if (k.done) return flush_tail(nullptr);     // (nothing to decode: a pending tail refers to a table slot this call's successors will reuse)
  k.base = base_slot(k.par); k.raw = raw_slot(k.par); k.part = part_slot(k.par); k.stereo_blk = stereo_slot(k.par);
  hp1 = std::chrono::steady_clock::now();
  if (int rc = run_tables(k)) return rc;
  hp2 = std::chrono::steady_clock::now();
  if (int rc = run_if_stage(k)) return rc;
  if (int rc = (mode == FMR_MODE_FM) ? run_fm(k) : (mode == FMR_MODE_NBFM) ? run_nbfm(k) : run_am(k)) return rc;
  HaloTable &ht = k.ht;
  if (ht.n) {      // (pipelined chain: the tail stage has taken the table with it, flush_tail)
    timed("shift_halo", [&] { hipLaunchKernelGGL(k_shift_halo<256>, dim3(ht.n, S), dim3(256), 0, stream, ht); });
  }
  if (pipelined) { ring_prev = k.par; ring_prev_n = k.N_if; }
  HIPCHK(hipGetLastError());
  return FMR_OK;
}

// IfResampler (or the pass-through copy): input -> IF buffer, input history, first halo entries
int fmr_chain::run_front_end(CallCtx &k) {
  auto &d_iq = k.d_iq; auto &stride = k.stride; auto &block_len = k.block_len; auto &nb = k.nb;
  auto &audio_len = k.audio_len; auto &N_in = k.N_in; auto &t_if_off = k.t_if_off;
  auto &t_if_len = k.t_if_len; auto &N_if = k.N_if; auto &use_fused = k.use_fused; auto &fused_geom = k.fused_geom;
  auto &par = k.par; auto &ifbuf = k.ifbuf; auto &ht = k.ht;
  auto add_halo = [&](void *buf, long long stride_e, int H, long long N, int words = 2) { k.add_halo(buf, stride_e, H, N, words); };
  // ------------------------------------------------------------------ front end
  long long count_mid_call = 0;
  N_if = 0;
  use_fused = false;
  fe_spare_cus = 0;
  k.fused_disc = false;
  fused_geom = FusedGeom{};
  hipStream_t fes = stream;      // (pipelined chain too: the front end alternates with the PLL stage on the decoder stream)
  par = 0;
  if (has_rs) {
    const long long mA_prev = rsc.mA, kB_prev = rsc.kB, n_prev = rsc.n_in;
    for (int b = 0; b < nb; b++) {
      t_if_off[b] = (int)N_if;
      const long long k = rsc.advance(rs, block_len[b]);
      t_if_len[b] = (int)k;
      N_if += k;
    }
    // Cross-call pipelining: this call's front end (its own stream, its own slot of the IF / MPX / partial-sum ring) runs
    // beside the PLL stage of the call before it and the audio tail of the call before that.  The slot was last read by the
    // tail of the call kPipe calls ago: the host polls a counter that a one-thread kernel at the end of every tail writes
    // into pinned host memory -- HIP event waits (stream-side or host-side) resolve against the newest signal of the other
    // stream at call time and would tie the stages together again (measured).  A call that yields no IF sample
    // decodes nothing and takes no slot.
    if (pipelined && N_if > 0) {
      pipe_seq++;
      par = (int)(pipe_seq % kPipe);
      // (one slot less than the ring holds: the slot of call N is still read at the head of call N+1, by the kernel that
      // carries its halos over, and that kernel is only ordered before the tail of call N+1)
      if (int rcw = wait_mark(&h_marks[1], pipe_seq > (unsigned long long)(kPipe - 1) ? pipe_seq - (kPipe - 1) : 0)) return rcw;
    }
    ifbuf = if_slot(par);
    last_if = ifbuf;
    const int count_mid = (int)(rsc.mA - mA_prev);
    count_mid_call = count_mid;
    if ((size_t)count_mid > max_mid || (size_t)N_if > max_if) { set_err("internal capacity exceeded"); return FMR_ERR_CAPACITY; }
    // Fused front end (stage A + stage B + discriminator in one persistent kernel) when this call is long enough and
    // its blocks are not tiny; any other call takes the three-kernel path -- both keep the same carried state.
    if (fused_ok && has_dec && !serial_mode && N_if >= 4 * 384 && count_mid >= H_mid &&
        ((uintptr_t)d_iq % 16) == 0 && (stride % 2) == 0) {
      use_fused = true;
      for (int b = 0; b < nb; b++) if (t_if_len[b] != 0 && t_if_len[b] < 128) use_fused = false;
    }
    k.fused_disc = use_fused && fused_disc_ok;
    // R8B class: the dense stage B carries the discriminator epilogue (it needs the block table: launched from run_tables)
    k.r8b_tail = false;
    if (!use_fused && r8b_disc_ok && poly5h && has_dec && !serial_mode && N_if >= 3072 && count_mid > 0) {
      k.r8b_tail = true;
      for (int b = 0; b < nb; b++) if (t_if_len[b] != 0 && t_if_len[b] < 128) k.r8b_tail = false;
    }
    if (k.r8b_tail) { k.fused_disc = true; k.r8b_mA_prev = mA_prev; k.r8b_kB_prev = kB_prev; k.r8b_count_mid = count_mid; }
    dec_valid = !k.fused_disc || debug_taps;   // the float copy of the discriminator output is a debug tap of the fused kernel
    if_valid = dec_valid;                      // ... and so are the IF samples behind its discriminator epilogue (the slot holds |x|^2 then)
    if (use_fused) {
      fused_geom = {mA_prev, kB_prev, n_prev, count_mid};
    } else if (count_mid > 0) {
      const long long top0 = (long long)rs.D * mA_prev + rs.ca() - n_prev;
      auto launch_decim = [&](auto bl_tag) {
        constexpr int BL = decltype(bl_tag)::value;
        const dim3 grid((count_mid + BL - 1) / BL, S);
        const size_t lds = sizeof(float2) * ((size_t)BL * rs.D + rs.NA - 1);
        hipLaunchKernelGGL(k_ifr_decim<BL>, grid, dim3(BL), lds, fes, d_iq, (long long)stride, N_in,
                           d_in_halo.p, H_in, d_hA.p, rs.NA, rs.D, top0, count_mid, d_mid.p,
                           (long long)(H_mid + max_mid), H_mid, (unsigned)(abs_in & 3u), cfg.enable_fourth_down);
      };
      const size_t per_out = sizeof(float2) * (size_t)rs.D, tail = sizeof(float2) * (size_t)(rs.NA - 1);
      // v2 kernel, 128 lanes per workgroup (256 -- half the tile-halo over-fetch, twice the LDS per workgroup -- measured
      // no faster in round 2)
      bool v2_done = false;
      auto launch_decim2 = [&](auto bl_tag) {
        constexpr int BL2 = decltype(bl_tag)::value, T2 = 2 * BL2;
        int s_pad = T2 + qa;
        while ((s_pad & 15) != 2) s_pad++;
        const size_t lds2 = sizeof(float2) * ((size_t)rs.D * s_pad + 2);   // + the spare slot
        if (!((qa == 16 || (qa == 24 && in_fmt == 0)) && lds2 <= 64000 && (size_t)rs.D * (T2 + qa) <= (size_t)2 * 16 * BL2)) return;
        const unsigned magic = (unsigned)((1u << 24) / (unsigned)rs.D + 1);
        const dim3 grid2((count_mid + T2 - 1) / T2, S);
        timed_on(fes, "ifr_decim", [&] {
          auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid2, dim3(BL2), lds2, fes, d_iq, (long long)stride, N_in, d_in_halo.p, H_in,
                               d_hpA.p, rs.D, rs.ca(), top0 - rs.ca(), count_mid, d_mid.p,
                               (long long)(H_mid + max_mid), H_mid, (unsigned)(abs_in & 3u),
                               (int)cfg.enable_fourth_down, s_pad, magic);
          };
          const bool f4 = cfg.enable_fourth_down != 0;
          if (qa == 24) { f4 ? go(k_ifr_decim2<BL2, 24, 0, true>) : go(k_ifr_decim2<BL2, 24, 0, false>); return; }
          switch (in_fmt) {
          case 1: f4 ? go(k_ifr_decim2<BL2, 16, 0, true, 1, 1>) : go(k_ifr_decim2<BL2, 16, 0, false, 1, 1>); break;
          case 2: f4 ? go(k_ifr_decim2<BL2, 16, 0, true, 1, 2>) : go(k_ifr_decim2<BL2, 16, 0, false, 1, 2>); break;
          case 3: f4 ? go(k_ifr_decim2<BL2, 16, 0, true, 1, 3>) : go(k_ifr_decim2<BL2, 16, 0, false, 1, 3>); break;
          default: f4 ? go(k_ifr_decim2<BL2, 16, 0, true>) : go(k_ifr_decim2<BL2, 16, 0, false>); break;
          }
        });
        v2_done = true;
      };
      // R8B class, 10 MS/s: the matrix-core form behind the fused front end's input ring, a contiguous run of 500-output epochs
      // per workgroup (calls of a few epochs per compute unit and up)
      if (decim16_ok && count_mid >= 4 * 500 && ((uintptr_t)d_iq % 16) == 0 && (stride % 2) == 0) {
        using SH16 = Decim16Shape<10, 195>;
        FusedArgs a{};
        a.iq = d_iq; a.iq_stride = (long long)stride; a.n_valid = N_in;
        a.in_halo = d_in_halo.p; a.H_in = H_in; a.afragA = reinterpret_cast<const uint4 *>(d_dec16_afrag.p); a.hA = d_hA.p;
        a.zero16 = reinterpret_cast<const float2 *>(d_zero16.p);
        const long long lo0 = top0 - (rs.NA - 1);
        const int par16 = (int)(((lo0 % 2) + 2) % 2);
        a.nbase = lo0 - par16;
        a.count_mid = count_mid;
        a.mid = d_mid.p; a.mid_stride = (long long)(H_mid + max_mid); a.H_mid = H_mid;
        a.n_tiles = (count_mid + SH16::ME - 1) / SH16::ME;
        const int fe_cus = pipelined ? std::max(8, n_cu - kFeSpareCus) : n_cu;
        const int wgs = std::max(1, fe_cus / S);
        a.tiles_per_wg = (a.n_tiles + wgs - 1) / wgs;
        const int grid16 = (a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
        timed_on(fes, "ifr_decim", [&] {
          if (par16) hipLaunchKernelGGL((k_ifr_decim16<10, 195, 1>), dim3(grid16, S), dim3(DECIM16_THREADS), SH16::LDS_BYTES, fes, a);
          else hipLaunchKernelGGL((k_ifr_decim16<10, 195, 0>), dim3(grid16, S), dim3(DECIM16_THREADS), SH16::LDS_BYTES, fes, a);
        });
        v2_done = true;
      } else
      launch_decim2(std::integral_constant<int, 128>{});
      if (v2_done) {
      } else if (in_fmt != 0) {
        set_err("input_format != cf32 needs the v2 front-end kernel (decimation ratio out of its range)");
        return FMR_ERR_UNSUPPORTED;
      } else {
        timed_on(fes, "ifr_decim", [&] {
          if (256 * per_out + tail <= 60000) launch_decim(std::integral_constant<int, 256>{});
          else if (128 * per_out + tail <= 60000) launch_decim(std::integral_constant<int, 128>{});
          else launch_decim(std::integral_constant<int, 64>{});
        });
      }
    }
    if (use_fused || k.r8b_tail) {
    } else if (N_if > 0 && poly2_tile > 0) {
      const long long P_first = kB_prev / rs.LB, P_last = (kB_prev + N_if - 1) / rs.LB;
      const int tiles = (int)((P_last - P_first) / 64 + 1);
      timed_on(fes, "ifr_poly", [&] {
        if (poly5h)
          hipLaunchKernelGGL((k_ifr_poly5h<48, 125>), dim3(std::min(tiles, n_cu), S), dim3(64 * FMR_POLY5H_WAVES), poly5h_lds,
                             fes, d_mid.p, (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_afrag5h.p,
                             poly5h_nkb, poly5h_inv_scale, rs.TB, kB_prev, (int)N_if, ifbuf, (long long)(H_if + max_if), H_if,
                             poly2_tile, tiles);
        else if (poly4_am) {
          const long long Pf = kB_prev / 48, Pl = (kB_prev + N_if - 1) / 48;
          const int tiles_am = (int)((Pl - Pf) / 64 + 1);
          hipLaunchKernelGGL((k_ifr_poly4<48, 128, 214>), dim3(std::min(tiles_am, 512), S), dim3(256),
                             sizeof(float2) * (size_t)(((poly4_am_tile + 127) / 128) * 128 + 4 * 8 * 48), fes, d_mid.p,
                             (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_afrag.p, kB_prev,
                             (int)N_if, ifbuf, (long long)(H_if + max_if), H_if, poly4_am_tile, tiles_am);
        } else if (poly4)
          hipLaunchKernelGGL((k_ifr_poly4<48, 125, 210>), dim3(std::min(tiles, 512), S), dim3(256),
                             sizeof(float2) * (size_t)(((poly2_tile + 127) / 128) * 128 + 4 * 8 * 48), fes, d_mid.p,
                             (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_afrag.p, kB_prev,
                             (int)N_if, ifbuf, (long long)(H_if + max_if), H_if, poly2_tile, tiles);
        else if (poly3)
          hipLaunchKernelGGL((k_ifr_poly3<384, 4>), dim3(tiles, S), dim3(384), sizeof(float2) * (size_t)poly2_tile, fes,
                             d_mid.p, (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_hBp.p, rs.TB,
                             (int)rs.LB, (int)rs.MB, d_bphi.p, d_boff.p, kB_prev, (int)N_if, ifbuf,
                             (long long)(H_if + max_if), H_if, poly2_tile);
        else
        hipLaunchKernelGGL(k_ifr_poly2<512>, dim3(tiles, S), dim3(512), sizeof(float2) * (size_t)poly2_tile, fes,
                           d_mid.p, (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_hB.p, rs.TB,
                           (int)rs.LB, (int)rs.MB, d_bphi.p, d_boff.p, kB_prev, (int)N_if, ifbuf,
                           (long long)(H_if + max_if), H_if, poly2_tile);
      });
    } else if (N_if > 0) {
      constexpr int BL = 256;
      const dim3 grid((unsigned)((N_if + BL - 1) / BL), S);
      const int span = (int)(((unsigned long long)(BL - 1) * rs.MB) / rs.LB) + rs.TB + 2;
      timed_on(fes, "ifr_poly", [&] {
        if (rs.LT) {
          // fractional-phase form: exact integer positions, call-relative on the device
          const __int128 tt = (__int128)kB_prev * rs.MB;
          const long long nk0 = (long long)(tt / rs.LB);
          const unsigned long long rem0 = (unsigned long long)(tt % rs.LB);
          hipLaunchKernelGGL(k_ifr_poly_frac<BL>, grid, dim3(BL), sizeof(float2) * (span + 4), fes, d_mid.p,
                             (long long)(H_mid + max_mid), nk0 - rs.W() + 1 - (mA_prev - H_mid), H_mid + count_mid, d_hB.p,
                             rs.TB, hB_pitch, rs.LT, (unsigned long long)rs.LB, (unsigned long long)rs.MB, rem0, (int)N_if,
                             ifbuf, (long long)(H_if + max_if), H_if);
        } else
        hipLaunchKernelGGL(k_ifr_poly<BL>, grid, dim3(BL), sizeof(float2) * span, fes, d_mid.p,
                           (long long)(H_mid + max_mid), mA_prev - H_mid, H_mid + count_mid, d_hB.p, rs.TB,
                           (unsigned)rs.LB, (unsigned)rs.MB, (unsigned long long)kB_prev * rs.MB, (int)N_if,
                           ifbuf, (long long)(H_if + max_if), H_if);
      });
    }
    if (N_in > 0 && !use_fused && !(k.r8b_tail && pipelined)) {
      timed_on(fes, "in_halo", [&] {
        switch (in_fmt) {
        case 1: hipLaunchKernelGGL((k_update_in_halo<256, 1>), dim3(1, S), dim3(256), 0, fes, d_in_halo.p, H_in, d_iq, (long long)stride, N_in); break;
        case 2: hipLaunchKernelGGL((k_update_in_halo<256, 2>), dim3(1, S), dim3(256), 0, fes, d_in_halo.p, H_in, d_iq, (long long)stride, N_in); break;
        case 3: hipLaunchKernelGGL((k_update_in_halo<256, 3>), dim3(1, S), dim3(256), 0, fes, d_in_halo.p, H_in, d_iq, (long long)stride, N_in); break;
        default: hipLaunchKernelGGL((k_update_in_halo<256, 0>), dim3(1, S), dim3(256), 0, fes, d_in_halo.p, H_in, d_iq, (long long)stride, N_in); break;
        }
      });
    }
  } else {
    ifbuf = d_if.p;
    last_if = ifbuf;
    if_valid = true;
    for (int b = 0; b < nb; b++) { t_if_off[b] = (int)N_if; t_if_len[b] = (int)block_len[b]; N_if += block_len[b]; }
    if (N_if > 0)
      HIPCHK(hipMemcpy2DAsync(ifbuf + H_if, sizeof(float2) * (H_if + max_if), d_iq, sizeof(float2) * stride,
                              sizeof(float2) * N_if, S, hipMemcpyDeviceToDevice, stream));
  }
  abs_in += (unsigned long long)N_in;
  last_n_if = N_if; last_nb = nb; last_n_au = 0;
  ht = HaloTable{};
  ht.n = 0;
  k.count_mid_call = count_mid_call;
  if (has_rs && pipelined) {
    // the front-end stage keeps its own history: the stage-B halo is re-seated on its stream (after the fused kernel,
    // which is launched from run_tables, when that one runs)
    if (!use_fused && !k.r8b_tail) {
      if (int rcf = finish_front_end_stage(k)) return rcf;
    }
  } else if (has_rs) {
    add_halo(d_mid.p, H_mid + (long long)max_mid, H_mid, count_mid_call);
  }
  if (!has_dec || N_if == 0) {
    hipLaunchKernelGGL(k_signal_host, dim3(1), dim3(1), 0, side, &h_marks[0], call_seq);   // no table this call
    if (audio_len) for (int b = 0; b < nb; b++) audio_len[b] = 0;
    if (has_dec && fir_enable && !pipelined) add_halo(ifbuf, H_if + (long long)max_if, H_if, N_if);
    if (ht.n) hipLaunchKernelGGL(k_shift_halo<256>, dim3(ht.n, S), dim3(256), 0, stream, ht);
    HIPCHK(hipGetLastError());
    k.done = true;                      // nothing to decode this call
    return FMR_OK;
  }
  return FMR_OK;
}

// Pipelined chain, end of the front-end stage on its stream: stage-B history for the next call, then the event the PLL
// stage of this call waits for.
int fmr_chain::finish_front_end_stage(CallCtx &k) {
  if (k.count_mid_call > 0) {
    HaloTable hm{};
    hm.d[0] = HaloDesc{(unsigned *)d_mid.p, 2 * (H_mid + (long long)max_mid), 2 * H_mid, 2 * (int)k.count_mid_call};
    hm.n = 1;
    hipLaunchKernelGGL(k_shift_halo<256>, dim3(1, S), dim3(256), 0, stream, hm);
  }
  if (has_dec && k.N_if > 0) {
    HIPCHK(hipEventRecord(ev_fe[k.par], stream));
    // the previous call's tail stage: behind this front end, beside this call's PLL stage
    if (int rc = flush_tail(ev_fe[k.par])) return rc;
  }
  return FMR_OK;
}

// per-call block / chunk tables on the side stream, then the fused front end (it needs the block table)
int fmr_chain::run_tables(CallCtx &k) {
  auto &d_iq = k.d_iq; auto &stride = k.stride; auto &nb = k.nb; auto &N_in = k.N_in;
  auto &h_tab = k.h_tab; auto &d_tab_slot = k.d_tab_slot; auto &t_if_off = k.t_if_off; auto &t_if_len = k.t_if_len;
  auto &t_au_off = k.t_au_off; auto &t_au_len = k.t_au_len; auto &t_mpf = k.t_mpf; auto &N_if = k.N_if;
  auto &use_fused = k.use_fused; auto &fused_geom = k.fused_geom; auto &ifbuf = k.ifbuf;
  auto &N_au = k.N_au; auto &any_mpf = k.any_mpf; auto &amA_prev = k.amA_prev; auto &akB_prev = k.akB_prev;
  auto &an_prev = k.an_prev; auto &nck = k.nck; auto &fused_n_tiles = k.fused_n_tiles;
  auto &fused_kb_ref = k.fused_kb_ref; auto &ct = k.ct; auto &iter_on_side = k.iter_on_side; auto &bt = k.bt;
  auto &if_stride = k.if_stride;
  // --------------------------------------------------------------- block tables
  N_au = 0;
  any_mpf = false;
  amA_prev = arsc.mA; akB_prev = arsc.kB; an_prev = arsc.n_in;
  for (int b = 0; b < nb; b++) {
    t_mpf[b] = 0;
    t_au_off[b] = (int)N_au;
    t_au_len[b] = 0;
    if (t_if_len[b] == 0) continue;          // the decoder is not called for an empty IF block (main.cpp:931-934)
    if (mode == FMR_MODE_FM) {
      if (wait_multipath_blocks > 0) wait_multipath_blocks--;     // FmDecode.cpp:107-110
      else if (enable_mpf) { t_mpf[b] = 1; any_mpf = true; }
      const long long k = arsc.advance(ars, t_if_len[b]);
      t_au_len[b] = (int)k;
      N_au += k;
    } else {
      t_au_len[b] = t_if_len[b];
      N_au += t_if_len[b];
    }
  }
  last_n_au = N_au;
  // PLL chunk table: every decoder block is cut into chunks of <= C_PLL samples
  // The per-chunk arrays (off, len, blk) are filled on the device from the block table; the
  // host only sends 6*max_blocks+1 ints per call.
  int *t_first = h_tab + 5 * (size_t)max_blocks;
  nck = 0;
  const int c_call = (N_if <= kSmallCall && env.x_cpll == 0) ? kCPllSmall : c_pll;       // (the same rule in both chain forms: N_if only)
  for (int b = 0; b < nb; b++) {
    t_first[b] = nck;
    nck += (t_if_len[b] + c_call - 1) / c_call;
  }
  t_first[nb] = nck;
  const size_t head_ints = 6 * (size_t)max_blocks + 1;
  // a kernel pulls the table out of the pinned host slot: hipMemcpyAsync H2D made the caller wait for the
  // stream to drain up to the copy (0.5-1 ms of host time per call), a launch does not
  // Table kernels and the PLL's initial node guess run on the side stream, beside the front end.
  hipLaunchKernelGGL(k_copy_ints, dim3((unsigned)((head_ints + 255) / 256)), dim3(256), 0, side,
                     (const int *)h_tab, d_tab_slot, (int)head_ints);
  int fused_grid = 0, fused_tiles_per_wg = 0, fused_part_from = 0;
  fused_n_tiles = 0; fused_kb_ref = 0;
  long long fused_T_first = 0;
  if (use_fused) {
    // one workgroup per CU: contiguous runs of macro tiles, the streams share the CUs.  The table's tail carries the
    // block that holds the first IF sample of every workgroup's run (the epilogue walks the block table from there).
    const long long P_first = fused_geom.kB_prev / 48, P_last = (fused_geom.kB_prev + N_if - 1) / 48;
    fused_T_first = P_first / 8;
    fused_n_tiles = (int)(P_last / 8 - fused_T_first + 1);
    // Pipelined chain: the front end leaves one CU of every XCD free.  A workgroup is dispatched inside the XCD its index
    // maps to, and the kernels that run beside the front end -- lock logic (32 KB of LDS), the DC block's node pass (213
    // VGPRs), the spare PLL rounds -- do not fit beside a 152 KB workgroup: on an XCD the front end fills they wait for
    // it to end (measured: the lock logic 250 instead of 55 us, and the next PLL pass behind it).
    const int fe_dflt = pipelined ? std::max(8, n_cu - kFeSpareCus) : n_cu;
    const int fe_cus = (pipelined && env.fe_cus > 0) ? std::min(env.fe_cus, n_cu) : fe_dflt;
    const int wg_per_stream = std::max(1, std::min(kFusedTile0Off - 1, fe_cus / S));
    // (ceil(n / grid) macro tiles for all but the last workgroup.  Balanced runs -- n mod grid workgroups with one tile more --
    // were measured in round 6: 248 instead of 245 workgroups at 2^27 samples, and the launch 5 us LONGER in the chain: the
    // three compute units more that the uneven split leaves free serve the kernels beside it)
    fused_tiles_per_wg = (fused_n_tiles + wg_per_stream - 1) / wg_per_stream;
    fused_grid = (fused_n_tiles + fused_tiles_per_wg - 1) / fused_tiles_per_wg;
    fe_spare_cus = std::max(0, n_cu - fused_grid * S);
    int *t_wg = h_tab + (tab_ints - kMaxFusedWg);
    const long long kb_ref = 384 * fused_T_first - fused_geom.kB_prev;
    {   // the first block k_stats walks (kernels.hpp, same rule): earlier blocks need no partial sums
      int seen = 0, b_first = 0;
      for (int b0 = ((nb - 1) / 64) * 64; b0 > 0 && !b_first; b0 -= 64) {
        for (int b = b0; b < std::min(b0 + 64, nb); b++) seen += t_if_len[b] != 0;
        if (seen >= 400) b_first = b0;
      }
      fused_part_from = t_if_off[b_first];
    }
    // Where the runs start: a macro tile whose epilogue writes the per-block partial sums (the last ~400 blocks of a call:
    // block walk, six wave reductions and a store per 128 samples) costs its workgroup kFusedSumWeight more than one that does
    // not -- measured, round 6: the workgroups of the last fifth of a 2048-block call took 207-213 us against the others' 194-197
    // and the launch ended with them (with weight 0.05 they still did, with 0.11 the last to end are spread over the chip).  Runs of equal WEIGHT instead of equal length; the table's tail carries the first tile of
    // every run behind the first block of every run.
    int *t_tile0 = t_wg + kFusedTile0Off;
    {
      const double eps = k.fused_disc ? (env.x_sumw >= 0 ? env.x_sumw * 1e-3 : kFusedSumWeight) : 0.0;
      const long long tp = std::min<long long>(fused_n_tiles, std::max<long long>(0, (fused_part_from - kb_ref) / 384));   // first tile with sums
      const double W = (double)tp + (double)(fused_n_tiles - tp) * (1.0 + eps);
      t_tile0[0] = 0;
      for (int w = 1; w < fused_grid; w++) {
        const double cum = W * w / fused_grid;
        const double tt = cum <= (double)tp ? cum : (double)tp + (cum - (double)tp) / (1.0 + eps);
        int ti = (int)(tt + 0.5);
        ti = std::max(ti, t_tile0[w - 1] + 1);                              // every run holds a tile
        ti = std::min(ti, fused_n_tiles - (fused_grid - w));                // ... the later ones too
        t_tile0[w] = ti;
      }
      t_tile0[fused_grid] = fused_n_tiles;
    }
    int b = 0;
    for (int w = 0; w < fused_grid; w++) {
      const long long kf = std::max<long long>(0, kb_ref + 384ll * t_tile0[w]);
      while (b < nb && (long long)t_if_off[b] + t_if_len[b] <= kf) b++;
      t_wg[w] = b;
    }
    // the kernel's own tables: first block of every workgroup and, when the front end runs a call ahead of the decoder,
    // the block table too (the side stream's copy of it sits behind the previous call's lock logic; both copies write
    // the same values)
    if (pipelined)
      hipLaunchKernelGGL(k_copy_ints2, dim3((unsigned)((kFusedTile0Off + fused_grid + 1 + 2 * (size_t)max_blocks + 255) / 256)), dim3(256), 0, stream,
                         (const int *)t_wg, d_tab_slot + (tab_ints - kMaxFusedWg), kFusedTile0Off + fused_grid + 1, (const int *)h_tab, d_tab_slot,
                         2 * max_blocks);
    else
      hipLaunchKernelGGL(k_copy_ints, dim3((unsigned)((kFusedTile0Off + fused_grid + 1 + 255) / 256)), dim3(256), 0, side,
                         (const int *)t_wg, d_tab_slot + (tab_ints - kMaxFusedWg), kFusedTile0Off + fused_grid + 1);
  }
  k.fir_tail = false;
  if (fir_mfma_fm && mode == FMR_MODE_FM && fir_enable && !enable_mpf && !serial_mode && N_if >= 3072) {
    // FM -f on the matrix cores: contiguous runs of 3072-output tiles per workgroup (call-relative), first block of every run
    k.fir_tail = true;
    for (int b = 0; b < nb; b++) if (t_if_len[b] != 0 && t_if_len[b] < 128) k.fir_tail = false;
  }
  if (k.fir_tail) {
    k.fir_tiles = (int)((N_if - 1) / 3072 + 1);
    const int wgs = std::max(1, std::min(kMaxFusedWg - kFirBlk0Off, 2 * n_cu / S));
    // (two workgroups share a compute unit: balanced runs -- fir_rem of them one tile longer -- keep all of them busy, where
    // ceil(tiles / slots) for everybody left 46 of 256 units idle at 2^27 samples per call)
    k.fir_grid = std::min(wgs, k.fir_tiles);
    k.fir_tpw = k.fir_tiles / k.fir_grid;
    k.fir_rem = k.fir_tiles - k.fir_tpw * k.fir_grid;
    {
      int seen = 0, b_first = 0;
      for (int b0 = ((nb - 1) / 64) * 64; b0 > 0 && !b_first; b0 -= 64) {
        for (int b = b0; b < std::min(b0 + 64, nb); b++) seen += t_if_len[b] != 0;
        if (seen >= 400) b_first = b0;
      }
      k.fir_part_from = t_if_off[b_first];
    }
    int *t_fb = h_tab + (tab_ints - kMaxFusedWg) + kFirBlk0Off;
    int b = 0;
    for (int w = 0; w < k.fir_grid; w++) {
      const long long kf = 3072ll * ((long long)w * k.fir_tpw + std::min(w, k.fir_rem));
      while (b < nb && (long long)t_if_off[b] + t_if_len[b] <= kf) b++;
      t_fb[w] = b;
    }
    hipLaunchKernelGGL(k_copy_ints, dim3((unsigned)((k.fir_grid + 255) / 256)), dim3(256), 0, side,
                       (const int *)t_fb, d_tab_slot + (tab_ints - kMaxFusedWg) + kFirBlk0Off, k.fir_grid);
  }
  int r8b_grid = 0, r8b_tpw = 0, r8b_rem = 0, r8b_tiles = 0;
  if (k.r8b_tail) {
    // R8B class: contiguous runs of stage-B tiles (64 periods = 3072 IF samples) per workgroup; the table's tail carries the
    // block that holds the first IF sample of every run, as for the fused front end
    const long long P_first = k.r8b_kB_prev / 48, P_last = (k.r8b_kB_prev + N_if - 1) / 48;
    r8b_tiles = (int)((P_last - P_first) / 64 + 1);
    // (pipelined chain: a compute unit per XCD stays free for the kernels beside the front end, as under the fused kernel)
    const int wgs = std::max(1, std::min(kFusedTile0Off - 1, (pipelined ? std::max(8, n_cu - kFeSpareCus) : n_cu) / S));
    // balanced runs: r8b_rem of them one tile longer (ceil(tiles / workgroups) for everybody left 16 of 256 compute units idle
    // at 2^27 samples per call and put seven tiles WITH partial sums on the workgroups that end the launch)
    r8b_grid = std::min(wgs, r8b_tiles);
    r8b_tpw = r8b_tiles / r8b_grid;
    r8b_rem = r8b_tiles - r8b_tpw * r8b_grid;
    fused_n_tiles = 8 * r8b_tiles;                             // in the epilogue's macro tiles of 384 samples
    const long long kb_ref = 48 * P_first - k.r8b_kB_prev;
    fused_kb_ref = (int)kb_ref;
    {   // the first block k_stats walks (same rule as above)
      int seen = 0, b_first = 0;
      for (int b0 = ((nb - 1) / 64) * 64; b0 > 0 && !b_first; b0 -= 64) {
        for (int b = b0; b < std::min(b0 + 64, nb); b++) seen += t_if_len[b] != 0;
        if (seen >= 400) b_first = b0;
      }
      fused_part_from = t_if_off[b_first];
    }
    int *t_wg = h_tab + (tab_ints - kMaxFusedWg);
    int b = 0;
    for (int w = 0; w < r8b_grid; w++) {
      const long long kf = std::max<long long>(0, kb_ref + 3072ll * ((long long)w * r8b_tpw + std::min(w, r8b_rem)));
      while (b < nb && (long long)t_if_off[b] + t_if_len[b] <= kf) b++;
      t_wg[w] = b;
    }
    if (pipelined)
      hipLaunchKernelGGL(k_copy_ints2, dim3((unsigned)((r8b_grid + 2 * (size_t)max_blocks + 255) / 256)), dim3(256), 0, stream,
                         (const int *)t_wg, d_tab_slot + (tab_ints - kMaxFusedWg), r8b_grid, (const int *)h_tab, d_tab_slot,
                         2 * max_blocks);
    else
      hipLaunchKernelGGL(k_copy_ints, dim3((unsigned)((r8b_grid + 255) / 256)), dim3(256), 0, side,
                         (const int *)t_wg, d_tab_slot + (tab_ints - kMaxFusedWg), r8b_grid);
  }
  int *d_first = d_tab_slot + 5 * (size_t)max_blocks;
  int *d_ck = d_tab_slot + head_ints;
  ct = ChunkTab{d_ck, d_ck + max_ck, d_ck + 2 * max_ck, d_first, nck};
  hipLaunchKernelGGL(k_signal_host, dim3(1), dim3(1), 0, side, &h_marks[0], call_seq);
  if (mode == FMR_MODE_FM && nck > 0) {
    hipLaunchKernelGGL(k_chunk_tab, dim3(nb), dim3(64), 0, side, d_tab_slot, d_tab_slot + max_blocks, d_first, c_call,
                       d_ck, d_ck + max_ck, d_ck + 2 * max_ck);
    if (stereo && !serial_mode)
      timed_on(side, "pll_begin", [&] {
        hipLaunchKernelGGL(k_pll_begin, dim3((nck + 1 + 63) / 64, S), dim3(64), 0, side, d_pll_nodes.p, ct, d_state.p,
                           pllc);
      });
  }
  // FM without the equaliser: the round flags and the AGC's start nodes are reset here too, beside the front end (5-10 us
  // of the critical path when launched between the front end and the PLL's first pass).  Their last readers of the previous
  // call are the PLL kernels (ordered before this stream's k_pll_finish) and the side-stream AGC (ev_agc).
  iter_on_side = (mode == FMR_MODE_FM) && !serial_mode && !enable_mpf;
  if (iter_on_side) {
    if (ev_agc_live) HIPCHK(hipStreamWaitEvent(side, ev_agc, 0));
    const int nc = (int)((N_if + C_AGC - 1) / C_AGC);
    hipLaunchKernelGGL(k_iter_begin, dim3(S), dim3(256), 0, side, d_flags.p, d_agc_nodes.p, nc, d_state.p, S,
                       (unsigned long long *)d_pll_sync.p, (int)(sizeof(PllSync) / 8), d_pll_tick2.p, pll_tick2_per_stream, d_agc_tick.p);
  }
  if (pipelined && ring_prev >= 0 && ring_prev != k.par) {
    // halos of this call's ring slot = the tail of the previous call's slot (its writers -- front end, discriminator, PLL
    // -- are ordered before this stream's lock logic of that call; nothing of this call touches the head of a slot)
    CarryTable ctab{};
    const long long bstr = H_b + (long long)max_if;
    const int np = (int)ring_prev_n;
    ctab.d[ctab.n++] = CarryDesc{(const unsigned *)base_slot(ring_prev), (unsigned *)k.base, bstr, H_b, np};
    if (stereo) ctab.d[ctab.n++] = CarryDesc{(const unsigned *)raw_slot(ring_prev), (unsigned *)k.raw, 2 * bstr, 2 * H_b, 2 * np};
    if (fir_enable) ctab.d[ctab.n++] = CarryDesc{(const unsigned *)if_slot(ring_prev), (unsigned *)ifbuf, 2 * (H_if + (long long)max_if), 2 * H_if, 2 * np};
    hipLaunchKernelGGL(k_carry_halo<256>, dim3(ctab.n, S), dim3(256), 0, side, ctab);
  }
  // the decoder waits for the tables.  In the pipelined chain the front end shares its stream and must not: the wait is
  // enqueued behind the front end's launch (below)
  HIPCHK(hipEventRecord(ev_tab, side));
  if (!pipelined) HIPCHK(hipStreamWaitEvent(stream, ev_tab, 0));
  bt = BlockTab{d_tab_slot, d_tab_slot + max_blocks, d_tab_slot + 2 * max_blocks, d_tab_slot + 3 * max_blocks,
              d_tab_slot + 4 * max_blocks, nb};
  if_stride = H_if + (long long)max_if;
  if (use_fused) {
    // ---- fused front end: needs the block table (per-block statistics), hence launched here, after the table copy
    constexpr int D = kFusedD, NA = kFusedNA;
    FusedArgs a{};
    a.iq = d_iq; a.iq_stride = (long long)stride; a.n_valid = N_in;
    a.in_halo = d_in_halo.p; a.H_in = H_in; a.afragA = reinterpret_cast<const uint4 *>(d_fused_afragA.p); a.hA = d_hA.p; a.hB = d_hB.p;
    const long long n0 = (long long)rs.D * fused_geom.mA_prev - fused_geom.n_prev;
    const long long lo0 = n0 + rs.ca() - (NA - 1);
    const int par = (int)(((lo0 % 2) + 2) % 2);
    a.nbase = lo0 - par;
    constexpr int kME = FusedShape<kFusedD, kFusedNA>::ME, kEPT = FusedShape<kFusedD, kFusedNA>::EPT;
    const long long T_first = fused_T_first, E_ref = kEPT * T_first - 1;
    a.j_ref = (int)(kME * E_ref + 104 - fused_geom.mA_prev);
    a.pos_ref = (int)((((kME * E_ref + 208) % 3000) + 3000) % 3000);
    a.t3_ref = (int)(T_first % 3);
    a.kb_ref = (int)(384 * T_first - fused_geom.kB_prev);
    a.count_mid = fused_geom.count_mid;
    a.mid = d_mid.p; a.mid_stride = (long long)(H_mid + max_mid); a.H_mid = H_mid;
    a.afragB = reinterpret_cast<const uint4 *>(d_fused_afragB.p); a.hB_inv_scale = fused_hB_inv_scale; a.n_if = (int)N_if;
    a.out = ifbuf; a.out_stride = if_stride; a.out_off = H_if;
    k.nrm = nullptr;
    if (k.fused_disc && !debug_taps) {
      // nobody but the AGC's state solve reads the IF behind the discriminator epilogue: it gets |x|^2 (4 B per sample, in the
      // IF slot's memory) and the IF samples stay on chip
      a.out = nullptr;
      a.nrm = reinterpret_cast<float *>(ifbuf); a.nrm_stride = 2 * if_stride; a.nrm_off = 0;
      k.nrm = a.nrm; k.nrm_stride = a.nrm_stride;
    }
    a.n_tiles = fused_n_tiles;
    a.tiles_per_wg = fused_tiles_per_wg;
    const int grid = fused_grid;
    a.wg_blk0 = d_tab_slot + (tab_ints - kMaxFusedWg);
    a.wg_tile0 = a.wg_blk0 + kFusedTile0Off;
    a.zero16 = reinterpret_cast<const float2 *>(d_zero16.p);
    a.base = k.fused_disc ? k.base : nullptr;        // null: IF samples only (an IF FIR or the equaliser comes first)
    a.base_stride = H_b + (long long)max_if; a.base_off = H_b;
    a.dec = debug_taps ? d_dec.p : nullptr; a.dec_stride = (long long)max_if;
    a.nf = disc_nf; a.bound = disc_bound;
    a.st = d_state.p; a.hB_last = d_hB_last.p; a.part = k.part;
    a.if_off = bt.if_off; a.if_len = bt.if_len; a.nb = nb;
    a.part_from = fused_part_from;
    if ((size_t)a.n_tiles * 3 * S > d_fused_part.n) { set_err("internal capacity exceeded (fused tiles)"); return FMR_ERR_CAPACITY; }
    constexpr size_t kLds = FusedShape<D, NA>::LDS_BYTES;
    hipStream_t fes = stream;
    // the discriminator's carried phase: when the previous call ran the discriminator in its decoder stage (a call too
    // short for the fused kernel), its phase is committed by that call's statistics kernel on the side stream
    if (pipelined && k.fused_disc && disc_commit_on_side) HIPCHK(hipStreamWaitEvent(stream, ev_stats, 0));
    if (d_fe_stamps.p) { a.stamps = d_fe_stamps.p; fe_stamps_n = grid * S; }
    if ((size_t)grid * S * 2 * FusedShape<D, NA>::MIDR > d_fused_mid32.n) { set_err("internal capacity exceeded (fused workgroups)"); return FMR_ERR_CAPACITY; }
    a.mid32 = d_fused_mid32.p;
    if (d_fe_stamps.p) hipLaunchKernelGGL(k_fused_stamp, dim3(1), dim3(1), 0, fes, stamp_slot(0), 16 * kStampCalls);
    timed_on(fes, "ifr_fused", [&] {
      if (ext_a) {      // (timed with the events of its own dispatch)
        if (par) hipExtLaunchKernelGGL((k_ifr_fused<D, NA, 1, 0>), dim3(grid, S), dim3(FUSED_THREADS), kLds, fes, ext_a, ext_b, 0, a);
        else hipExtLaunchKernelGGL((k_ifr_fused<D, NA, 0, 0>), dim3(grid, S), dim3(FUSED_THREADS), kLds, fes, ext_a, ext_b, 0, a);
        return;
      }
      if (par) hipLaunchKernelGGL((k_ifr_fused<D, NA, 1, 0>), dim3(grid, S), dim3(FUSED_THREADS), kLds, fes, a);
      else hipLaunchKernelGGL((k_ifr_fused<D, NA, 0, 0>), dim3(grid, S), dim3(FUSED_THREADS), kLds, fes, a);
    });
    if (d_fe_stamps.p) hipLaunchKernelGGL(k_fused_stamp, dim3(1), dim3(1), 0, fes, stamp_slot(1), 0);
    fused_kb_ref = a.kb_ref;
    if (pipelined) {
      // The PLL stage starts from here.  What the front-end stage carries into its next call -- the input history, the
      // stage-B history, the discriminator's last phase -- is one small kernel on its stream; when that stream is the
      // decoder's it is launched behind the PLL's first pass (run_fm_pll), not between the front end and that pass.
      HIPCHK(hipEventRecord(ev_fe[k.par], stream));
      const int commit = (int)k.fused_disc, count_mid = fused_geom.count_mid;
      k.fe_post = [=] {
        hipLaunchKernelGGL(k_fe_post<256>, dim3(3, S), dim3(256), 0, stream, d_in_halo.p, H_in, d_iq, (long long)stride, N_in,
                           d_mid.p, (long long)(H_mid + max_mid), H_mid, count_mid, d_state.p, commit);
      };
      // the previous call's tail stage: behind this front end, beside this call's PLL stage
#ifndef FMR_FIR_TAIL_EARLY
      // (... and behind the IF filter's kernel where that one sits between front end and PLL: run_fm)
      if (k.fir_tail) { k.tail_deferred = true; if (int rc = flush_walk()) return rc; } else
#endif
      if (int rc = flush_tail(ev_fe[k.par])) return rc;
    }
  }
  if (k.r8b_tail) {
    // ---- R8B class: stage B with the discriminator epilogue (the fused front end's, a wave per 384 staged samples)
    FusedArgs a{};
    a.n_if = (int)N_if; a.kb_ref = fused_kb_ref;
    a.out = nullptr; a.out_stride = if_stride; a.out_off = H_if;
    a.nrm = reinterpret_cast<float *>(ifbuf); a.nrm_stride = 2 * if_stride; a.nrm_off = 0;
    if (debug_taps) { a.out = ifbuf; a.nrm = nullptr; }      // the IF samples themselves (fmr_debug_read 0); the AGC reads them then
    k.nrm = a.nrm; k.nrm_stride = a.nrm_stride;
    a.base = k.base; a.base_stride = H_b + (long long)max_if; a.base_off = H_b;
    a.dec = debug_taps ? d_dec.p : nullptr; a.dec_stride = (long long)max_if;
    a.nf = disc_nf; a.bound = disc_bound;
    a.st = d_state.p; a.part = k.part; a.n_tiles = fused_n_tiles; a.part_from = fused_part_from;
    a.if_off = bt.if_off; a.if_len = bt.if_len; a.nb = nb;
    a.wg_blk0 = d_tab_slot + (tab_ints - kMaxFusedWg);
    a.mid32 = d_run_ph.p;
    if ((size_t)a.n_tiles * 3 * S > d_fused_part.n || (size_t)r8b_grid * 2 * S > d_run_ph.n) { set_err("internal capacity exceeded (stage-B tiles)"); return FMR_ERR_CAPACITY; }
    if (pipelined && disc_commit_on_side) HIPCHK(hipStreamWaitEvent(stream, ev_stats, 0));
    timed_on(stream, "ifr_poly", [&] {
      hipLaunchKernelGGL((k_ifr_poly5h<48, 125, Poly5hDiscEpi>), dim3(r8b_grid, S), dim3(64 * FMR_POLY5H_WAVES), poly5h_lds, stream,
                         d_mid.p, (long long)(H_mid + max_mid), k.r8b_mA_prev - H_mid, H_mid + k.r8b_count_mid, d_afrag5h.p,
                         poly5h_nkb, poly5h_inv_scale, rs.TB, k.r8b_kB_prev, (int)N_if, (float2 *)nullptr, 0ll, 0, poly2_tile, r8b_tiles,
                         a, r8b_tpw, r8b_rem);
      if (r8b_grid > 1)
        hipLaunchKernelGGL(k_poly5h_heads, dim3((r8b_grid + 62) / 64, S), dim3(64), 0, stream, a, r8b_grid, r8b_tpw, r8b_rem);
    });
    if (pipelined) {
      // as behind the fused front end: the PLL stage starts from here; input history, stage-B history and the
      // discriminator's phase are one small kernel behind the PLL's first pass
      HIPCHK(hipEventRecord(ev_fe[k.par], stream));
      const int count_mid = k.r8b_count_mid;
      k.fe_post = [=] {
        hipLaunchKernelGGL(k_fe_post<256>, dim3(3, S), dim3(256), 0, stream, d_in_halo.p, H_in, d_iq, (long long)stride, N_in,
                           d_mid.p, (long long)(H_mid + max_mid), H_mid, count_mid, d_state.p, 1);
      };
      if (int rc = flush_tail(ev_fe[k.par])) return rc;
    }
  }
  if (pipelined) HIPCHK(hipStreamWaitEvent(stream, ev_tab, 0));
  return FMR_OK;
}

// IF-rate part common to all decoders: fine tuner, IF filter + level, IF AGC
int fmr_chain::run_if_stage(CallCtx &k) {
  auto &nb = k.nb; auto &N_if = k.N_if; auto &ifbuf = k.ifbuf; auto &iter_on_side = k.iter_on_side; auto &bt = k.bt;
  auto &if_stride = k.if_stride; auto &rms_in_disc = k.rms_in_disc; auto &xin = k.xin; auto &x_stride = k.x_stride;
  auto &x_off = k.x_off; auto &disc_gain = k.disc_gain; auto &agc_on_side = k.agc_on_side;
  auto &agc_deferred = k.agc_deferred; auto &enqueue_agc = k.enqueue_agc;
  // ------------------------------------------------------- decoder, IF-rate part
  // SSB / WSPR: mix the new IF samples in place before the filter (the filter history in the halo is already mixed)
  if (ssb_like && d_ft_pre.p)
    timed("finetune_pre", [&] {
      hipLaunchKernelGGL(k_finetune<256>, dim3(nb, S), dim3(256), 0, stream, ifbuf, if_stride, H_if, bt, d_ft_pre.p,
                         480, ft_index, (float *)nullptr);
    });
  // FM without the IF FIR: the block RMS is taken inside the discriminator kernel (same samples, same lane order)
  rms_in_disc = (mode == FMR_MODE_FM) && !fir_enable && !serial_mode;
  if (!rms_in_disc) {
    timed("fm_block", [&] {
      // four outputs per lane; FM without the equaliser: the discriminator is its epilogue (k_disc is not launched)
      // tile length: 1024 outputs, or the longest IF block of the call rounded up to four when that is shorter
      int tl = 4;
      for (int b = 0; b < nb; b++) tl = std::max(tl, (k.t_if_len[b] + 3) & ~3);
      const int TL = std::min(1024, tl);
      const size_t lds_fb = sizeof(float2) * (4 * (size_t)fm_block3_plane(ntaps - 1, TL) + ((size_t)ntaps + 4) / 2 + (size_t)(ntaps - 1) + TL);
      const bool blocked = fir_enable && ntaps >= 2 && lds_fb <= 60000 && !serial_mode;
      k.fir_disc = blocked && mode == FMR_MODE_FM && !enable_mpf;
      if (!k.fir_disc) k.fir_tail = false;
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nb, S), dim3(256), lds_fb, stream, ifbuf, if_stride, H_if, bt,
                           d_coeff.p, ntaps, (int)(mode != FMR_MODE_FM), d_fir.p, (long long)max_if, d_if_rms_blk.p,
                           disc_nf, disc_bound, d_dec.p, (long long)max_if, k.base, H_b + (long long)max_if, H_b,
                           d_bb_mean_blk.p, d_bb_rms_blk.p, d_blk_ph.p, TL);
      };
      if (k.fir_disc && k.fir_tail) {
        FusedArgs a{};
        a.n_if = (int)N_if; a.kb_ref = 0;
        a.out = d_fir.p; a.out_stride = (long long)max_if; a.out_off = 0;
        a.base = k.base; a.base_stride = H_b + (long long)max_if; a.base_off = H_b;
        a.dec = debug_taps ? d_dec.p : nullptr; a.dec_stride = (long long)max_if;
        a.nf = disc_nf; a.bound = disc_bound;
        a.st = d_state.p; a.part = k.part; a.n_tiles = 8 * k.fir_tiles; a.part_from = k.fir_part_from;
        a.if_off = bt.if_off; a.if_len = bt.if_len; a.nb = nb;
        a.wg_blk0 = k.d_tab_slot + (tab_ints - kMaxFusedWg) + kFirBlk0Off;
        a.mid32 = d_run_ph.p;
        a.fir_c0 = h_coeff0; a.fir_order = ntaps - 1; a.hA = d_coeff.p;      // (hA: the filter's taps, for the repair of tiles that hold a non-finite sample)
        if ((size_t)a.n_tiles * 3 * S > d_fused_part.n || (size_t)k.fir_grid * 2 * S > d_run_ph.n) { set_err("internal capacity exceeded (IF filter tiles)"); return; }
        constexpr int kTile = 64 * 48 + 47 + 127 + 64;
        hipLaunchKernelGGL((k_ifr_poly4<48, 48, 127, 2, Poly4FirDiscEpi>), dim3(k.fir_grid, S), dim3(256),
                           sizeof(float2) * (size_t)(((kTile + 127) / 128) * 128 + 4 * 8 * 48), stream, ifbuf, if_stride,
                           (long long)(64 - H_if), H_if + (int)N_if, d_afrag_fir_fm.p, 0ll, (int)N_if, (float2 *)nullptr, 0ll, 0,
                           kTile, k.fir_tiles, a, k.fir_tpw, k.fir_rem);
        if (k.fir_grid > 1)
          hipLaunchKernelGGL(k_poly5h_heads, dim3((k.fir_grid + 62) / 64, S), dim3(64), 0, stream, a, k.fir_grid, k.fir_tpw, k.fir_rem);
      } else if (k.fir_disc) {
        go(k_fm_block3<256, true>);
        hipLaunchKernelGGL(k_disc_heads, dim3((nb + 255) / 256, S), dim3(256), 0, stream, bt, d_blk_ph.p, disc_bound, d_dec.p,
                           (long long)max_if, k.base, H_b + (long long)max_if, H_b, d_bb_mean_blk.p, d_bb_rms_blk.p, d_state.p);
      }
      else if (blocked && mode == FMR_MODE_FM) go(k_fm_block3<256, false>);       // (FM with the equaliser behind the filter)
      else if (blocked && fir_mfma && N_if >= 48) {
        // AM / DSB / NBFM, 255 or 127 taps: the matrix-core form (call-relative periods of 48 outputs; buffer index m of the
        // kernel's window arithmetic is x[m - (order - (W - 1))], W = taps / 2: the "absolute" index of the buffer's first element
        // is order - W + 1 - H_if)
        const int order = ntaps - 1, Wf = ntaps >> 1;
        const int kTile = 64 * 48 + 47 + ntaps + 64;
        const int tiles_f = (int)((N_if - 1) / 48 / 64 + 1);
        auto gof = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(std::min(tiles_f, 1024), S), dim3(256),
                             sizeof(float2) * (size_t)(((kTile + 127) / 128) * 128 + 4 * 8 * 48), stream, ifbuf, if_stride,
                             (long long)(order - Wf + 1 - H_if), H_if + (int)N_if, d_afrag_fir.p, 0ll, (int)N_if, d_fir.p, (long long)max_if, 0,
                             kTile, tiles_f, Poly5hStoreIf::Args{}, 0, 0);
        };
        if (ntaps == 255) gof(k_ifr_poly4<48, 48, 255, 1>); else gof(k_ifr_poly4<48, 48, 127, 2>);
        hipLaunchKernelGGL(k_fir_finish<256>, dim3(nb, S), dim3(256), 0, stream, ifbuf, if_stride, H_if, bt, d_coeff.p, ntaps,
                           d_fir.p, (long long)max_if, d_if_rms_blk.p);
      }
      else if (blocked) {
        // the 48 kHz modes: blocks of a few hundred samples behind 255 or 2049 taps, nearly every output a head output
        constexpr int TL2 = 1024;
        const size_t lds2 = sizeof(float2) * ((size_t)(ntaps - 1) + TL2) + sizeof(float) * (size_t)ntaps;
        hipLaunchKernelGGL((k_fm_block2<256, TL2>), dim3(nb, S), dim3(256), lds2, stream, ifbuf, if_stride, H_if, bt,
                           d_coeff.p, ntaps, (int)(mode != FMR_MODE_FM), d_fir.p, (long long)max_if, d_if_rms_blk.p);
      }
      else
      hipLaunchKernelGGL(k_fm_block<256>, dim3(nb, S), dim3(256), 0, stream, ifbuf, if_stride, H_if, bt, d_coeff.p,
                         ntaps, (int)fir_enable, (int)(mode != FMR_MODE_FM), d_fir.p, (long long)max_if,
                         d_if_rms_blk.p);
    });
  }
  if (ssb_like) {    // mix the filter output in place; the IF level is measured after it (AmDecode.cpp:114,122,128,136,154)
    timed("finetune_post", [&] {
      hipLaunchKernelGGL(k_finetune<256>, dim3(nb, S), dim3(256), 0, stream, d_fir.p, (long long)max_if, 0, bt,
                         d_ft_post.p, 480, ft_index, d_if_rms_blk.p);
    });
    ft_index = (unsigned)((ft_index + (unsigned long long)N_if) % 480u);
  }
  xin = fir_enable ? d_fir.p : ifbuf;
  x_stride = fir_enable ? (long long)max_if : if_stride;
  x_off = fir_enable ? 0 : H_if;
  // ---- IF AGC: Newton multiple shooting over chunks of C_AGC samples (kernels_par.hpp)
  disc_gain = d_gain.p;      // gain sequence the discriminator multiplies in (nullptr = none)
  agc_on_side = false; agc_deferred = false; agc_beside_mpf = false;
  enqueue_agc = nullptr;      // argument: event that gates the side stream (null: a new marker on the main stream)
  const int agc_nc = (int)((N_if + C_AGC - 1) / C_AGC);
  if (!iter_on_side)
    hipLaunchKernelGGL(k_iter_begin, dim3(S), dim3(256), 0, stream, d_flags.p,
                       (serial_mode || enable_mpf) ? (float *)nullptr : d_agc_nodes.p,
                       agc_nc, d_state.p, S, (unsigned long long *)d_pll_sync.p, (int)(sizeof(PllSync) / 8), d_pll_tick2.p,
                       pll_tick2_per_stream, d_agc_tick.p);
  // With the equaliser on, the AGC'd amplitude feeds the constant-modulus error, and
  // the equaliser kernel is the serial bottleneck anyway: use the exact serial AGC.
  if (enable_mpf && !serial_mode && mode == FMR_MODE_FM) {
    // beside the equaliser, which consumes the gains as they are published (k_if_agc_wave / k_mpf4, kernels.hpp)
    // (the progress words count the samples of THIS call: zeroed here, in front of both kernels -- a call that failed
    // half way cannot leave a count behind that a later call would take for its own)
    HIPCHK(hipMemsetAsync(d_agc_progress.p, 0, sizeof(unsigned long long) * (size_t)S, stream));
    HIPCHK(hipEventRecord(ev_if, stream));
    HIPCHK(hipStreamWaitEvent(side2, ev_if, 0));
#ifdef FMR_AB_PARTNERS
    if (env.test_agc_late > 0)      // test hook: the AGC kernel starts this many milliseconds late
      hipLaunchKernelGGL(k_hold_stream, dim3(1), dim3(1), 0, side2, (unsigned long long)env.test_agc_late * 100000ull);
#endif
    timed_on(side2, "if_agc", [&] {
      if (env.test_agc_late >= 0)   // (test hook, < 0: the AGC kernel is not launched at all)
      hipLaunchKernelGGL(k_if_agc_wave, dim3(S), dim3(64), 0, side2, xin, x_stride, x_off, (int)N_if, d_gain.p,
                         (long long)max_if, d_state.p, agc_init, agc_max, agc_rate, d_agc_progress.p);
    });
    HIPCHK(hipEventRecord(ev_agc, side2));
    ev_agc_live = true;
    agc_beside_mpf = true;
  } else if (serial_mode || enable_mpf) {
    timed("if_agc", [&] {
      if (enable_mpf && !serial_mode)
        hipLaunchKernelGGL(k_if_agc_wave, dim3(S), dim3(64), 0, stream, xin, x_stride, x_off, (int)N_if, d_gain.p,
                           (long long)max_if, d_state.p, agc_init, agc_max, agc_rate, (unsigned long long *)nullptr);
      else
      hipLaunchKernelGGL(k_if_agc, dim3((S + 63) / 64), dim3(64), 0, stream, xin, x_stride, x_off, (int)N_if, d_gain.p,
                         (long long)max_if, d_state.p, S, agc_init, agc_max, agc_rate, (unsigned long long *)nullptr);
    });
  } else {
    // FM without the equaliser: the AGC output only feeds atan2, which is invariant to the
    // (positive) gain, so the discriminator reads the un-gained samples and the gain recurrence
    // -- still solved exactly as before, its state carries -- runs on the side stream, off the
    // critical path (the two paths differ by 3.8e-8 RMS of float rounding).
    const bool agc_aside = (mode == FMR_MODE_FM);
    // a short call (one or two source blocks) is launch-bound on the host: one spare round instead of two to four; if
    // that is not enough the serial kernel runs over these few thousand samples (milliseconds)
    const int agc_iters = (agc_aside && N_if <= kSmallCall) ? 3 : K_AGC_ITERS;
    hipStream_t as = agc_aside ? side2 : stream;
    // With the PLL on, the side-stream AGC starts only after the PLL's first (Jacobian) integration pass: that
    // pass runs one wave per SIMD and every co-resident AGC wave stretches it (measured 118 -> 160 us).
    agc_deferred = agc_aside && stereo;
    const float *const nrm_in = (agc_aside && !fir_enable) ? k.nrm : nullptr;     // (non-null: the front end stored |x|^2, not the IF samples)
    const long long nrm_in_stride = k.nrm_stride;
    enqueue_agc = [=](hipEvent_t gate) -> int {
    if (agc_aside) {
      if (gate) {
        HIPCHK(hipStreamWaitEvent(side2, gate, 0));
      } else {
        HIPCHK(hipEventRecord(ev_if, stream));
        HIPCHK(hipStreamWaitEvent(side2, ev_if, 0));
      }
    }
    // FM without the equaliser: the discriminator does not see the gains (atan2 is invariant to them) -- the recurrence is
    // solved for its state only, and its 4 bytes per IF sample stay out of HBM unless the debug tap asks for them
    float *const gain_out = (agc_aside && !debug_taps) ? (float *)nullptr : d_gain.p;
    timed_on(as, "if_agc", [&] {
      const int ginv = (mode == FMR_MODE_FM || mode == FMR_MODE_NBFM) ? (gain_out ? 1 : 2) : -env.x_amtol;
      // (four waves, one per SIMD: a workgroup of sixteen finds no compute unit with room for all of them while the PLL's
      // first pass -- one 260-register wave per SIMD, 1258 workgroups queueing -- holds the chip)
      constexpr int kAgcWg = 256;
      const size_t agc_ballast = agc_aside ? side_ballast() : 0;
      const dim3 rgrid((agc_nc + kAgcWg - 1) / kAgcWg, S);
      for (int it = 0; it < agc_iters; it++) {
        if (nrm_in)
          hipLaunchKernelGGL((k_agc_round<C_AGC, float>), rgrid, dim3(kAgcWg), agc_ballast, as, nrm_in, nrm_in_stride, 0, (int)N_if,
                             gain_out, (long long)max_if, d_agc_nodes.p, d_agc_G.p, d_agc_M.p, agc_nc, agc_init, agc_max,
                             agc_rate, d_state.p, d_flags.p, ginv, d_agc_tick.p);
        else
          hipLaunchKernelGGL((k_agc_round<C_AGC, float2>), rgrid, dim3(kAgcWg), agc_ballast, as, xin, x_stride, x_off, (int)N_if,
                             gain_out, (long long)max_if, d_agc_nodes.p, d_agc_G.p, d_agc_M.p, agc_nc, agc_init, agc_max,
                             agc_rate, d_state.p, d_flags.p, ginv, d_agc_tick.p);
      }
      if (nrm_in)
        hipLaunchKernelGGL(k_if_agc_fallback<float>, dim3((S + 63) / 64), dim3(64), 0, as, nrm_in, nrm_in_stride, 0, (int)N_if,
                           gain_out, (long long)max_if, d_state.p, S, agc_init, agc_max, agc_rate, d_flags.p);
      else
      hipLaunchKernelGGL(k_if_agc_fallback<float2>, dim3((S + 63) / 64), dim3(64), 0, as, xin, x_stride, x_off, (int)N_if,
                         gain_out, (long long)max_if, d_state.p, S, agc_init, agc_max, agc_rate, d_flags.p);
    });
    if (agc_aside) { HIPCHK(hipEventRecord(ev_agc, side2)); ev_agc_live = true; }
    return FMR_OK;
    };
    if (agc_aside) { disc_gain = nullptr; agc_on_side = true; }
    gain_valid = !(agc_aside && !debug_taps);
    if (!agc_deferred) { if (int rca = enqueue_agc(nullptr)) return rca; }
  }
  return FMR_OK;
}

// PilotPhaseLock: Newton multiple shooting over chunks (kernels_par.hpp); beside it, on side2, the IF AGC and the mono
// audio tail; after it, on side, the lock logic.  tail(stream, first channel, channels) enqueues the per-channel audio tail.
int fmr_chain::run_fm_pll(CallCtx &k, long long base_stride, bool split_mono,
                          const std::function<void(hipStream_t, int, int)> &enqueue_tail_channels, bool &mono_enqueued,
                          bool &fin_on_side, bool &fin_covers_all) {
  auto &nb = k.nb; auto &N_if = k.N_if; auto &nck = k.nck; auto &ct = k.ct; auto &bt = k.bt; auto &agc_on_side = k.agc_on_side; auto &agc_deferred = k.agc_deferred; auto &enqueue_agc = k.enqueue_agc;
  if (serial_mode) {
    timed("pll", [&] {
      hipLaunchKernelGGL(k_pll, dim3((S + 63) / 64), dim3(64), 0, stream, k.base, base_stride, H_b, bt, k.raw,
                         base_stride, H_b, d_atan.p, pllc, (int)pilot_shift, k.stereo_blk, d_state.p, S);
    });
  } else {
    // ---- pilot PLL: Newton multiple shooting over chunks of C_PLL samples
    int rc_agc = FMR_OK;          // a failure inside the lambda must leave run_fm_pll, not only the lambda
    bool spare_moved = false;     // the passes after the second went to the side stream
    // the wrap counts and masks the lock logic's walk reads: two copies in the pipelined chain, where the walk of this call
    // runs beside the next call's front end and is not ordered before that call's PLL passes, which write them again
    const bool walk_late = pipelined && !env.pll_v1 && !serial_mode;
    const int wpar = walk_late ? walk_par : 0;
    if (walk_late) walk_par ^= 1;
    int *const ck_wraps_now = d_ck_wraps.p + (size_t)wpar * ck_copy;
    unsigned long long *const ck_mask_now = d_ck_mask.p + (size_t)wpar * ck_copy * mask_words;
    // (trace mode: every kernel of the group carries its own event pair)
    auto sub = [&](hipStream_t st, const char *name, auto &&launch) { if (timing == 3) timed_on(st, name, launch); else launch(); };
    timed("pll", [&] {
      const int ngrp = (nck + FMR_NODE_GRP - 1) / FMR_NODE_GRP;
      const int ngrp2 = (ngrp + FMR_NODE_GRP2 - 1) / FMR_NODE_GRP2;
      const int pll_iters = (N_if <= kSmallCall) ? 3 : K_PLL_ITERS;       // (short calls: see the AGC above)
      hipStream_t ps = stream;
      // Pipelined chain, streams of few blocks: the passes after the second -- which a call in lock does not need, and which
      // then cost five launches of workgroups that read a flag and leave, ~25 us between this call's accepted pass and the
      // next call's front end on this stream -- go to the side stream, in front of the lock logic that waits for them anyway
      // (32 streams x 64 blocks: 0.503 -> 0.478 ms per step).  When they are needed (rounds 3+, the serial fallback) they
      // run beside the next front end; nothing of that call reads what they write before its tables, which are behind the
      // lock logic on the same stream.  Not with one long stream: there the side stream -- lock logic over 2048 blocks in
      // one wave, the AGC's rounds -- is as long as the step already, and five more launches on it hold the next call's
      // tables back (0.517 -> 0.539).
      const bool spare_aside = pipelined && !env.pll_v1 && nb <= (env.x_spare_aside >= 0 ? env.x_spare_aside : kSpareAsideMaxBlocks);
      for (int it = 0; it < pll_iters; it++) {
        if (spare_aside && it == 2 && !spare_moved) { (void)hipEventRecord(ev_pll1, stream); (void)hipStreamWaitEvent(side, ev_pll1, 0); spare_moved = true; ps = side; }
        // round 0 integrates the sensitivities too; later rounds reuse them (chord Newton: measured
        // contraction 5e-4 per round in lock, so the round count is the same as with fresh Jacobians)
        PllSync *const sy = env.pll_v1 ? nullptr : d_pll_sync.p;      // null: seven-kernel round (k_pll_check etc.)
        // The pass that integrates the Jacobians composes them in its own tail (the node pass's up-sweep: k_pll_up's work
        // without its launch and without reading 31 MB of Jacobians back), see k_pll_shoot
        const bool up_in_shoot = sy != nullptr && it < pll_jac_rounds && it + 1 < pll_iters;
        const PllUpArgs upa = up_in_shoot ? PllUpArgs{d_pll_PQ.p, d_pll_pre.p, d_pll_PQ2.p, ngrp, ngrp2, d_pll_dstart2.p, d_pll_sync.p, d_pll_tick2.p}
                                          : PllUpArgs{};
        // ... and every later pass runs the down-sweep of the node pass before it in its own head (k_pll_down's work without
        // its launch): a round is two launches, integration + up-sweep kernel (the first round: one)
        const bool down_in_shoot = sy != nullptr && it > 0;
        const PllDownArgs dna = down_in_shoot ? PllDownArgs{d_pll_pre.p, d_pll_dstart2.p, ngrp, ngrp2, pllc.minfreq, pllc.maxfreq, d_pll_wfirst.p}
                                              : PllDownArgs{};
        auto shoot = [&](auto kern) {
          sub(ps, it == 0 ? "pll_shoot_jac" : "pll_shoot", [&] {
          hipLaunchKernelGGL(kern, dim3((nck + 63) / 64, S), dim3(64), 0, ps, k.base, base_stride, H_b, ct,
                             k.raw, base_stride, H_b, d_atan.p, pllc, (int)pilot_shift, d_pll_nodes.p, d_pll_G.p,
                             d_pll_M.p, ck_wraps_now, ck_mask_now, mask_words, d_flags.p, d_pll_wgr.p, sy, 1.0,
                             pll_rtol, (int)(it > 0), upa, dna, sy ? d_pll_wfirst.p : (double *)nullptr);
          });
        };
        // the first round writes no L-R samples unless it can be the accepted one (a call of one or two chunks)
        const bool wout = it > 0 || env.pll_v1 || nck <= 2;
        if (it < pll_jac_rounds) { if (wout) shoot(k_pll_shoot<true, true>); else shoot(k_pll_shoot<true, false>); }
        else shoot(k_pll_shoot<false, true>);
        if (it == 0 && k.fe_post) { k.fe_post(); k.fe_post = nullptr; }
        if (it == 0 && agc_deferred) {
          // side2: the AGC and the mono audio tail beside the PLL.  Gated on an event that exists already -- the
          // front end's (ev_disc) -- because a marker of its own on this stream costs ~10 us between the first pass
          // and the node pass.
          agc_deferred = false;
          hipEvent_t gate = k.ev_mpx;
          if (!gate) { (void)hipEventRecord(ev_if, stream); gate = ev_if; }
          auto mono_aside = [&] {
            (void)hipStreamWaitEvent(side2, gate, 0);
            enqueue_tail_channels(side2, 0, 1);
            (void)hipEventRecord(ev_mono, side2);
            mono_enqueued = true;
          };
          if ((rc_agc = enqueue_agc(gate))) return;
          if (split_mono) mono_aside();
        }
#ifdef FMR_AB_PARTNERS
        if (env.pll_v1)
          hipLaunchKernelGGL(k_pll_check, dim3(S), dim3(1024), 0, stream, d_flags.p, S, 1.0, d_pll_gres.p, ngrp,
                             (int)(it > 0), d_pll_wgr.p, (nck + 63) / 64, pll_rtol);
#endif
        if (it == pll_iters - 1) break;        // nothing integrates the nodes a last update would give
        if (!env.pll_v1) {
          if (!up_in_shoot) {
          if (spare_aside && it == 1) { (void)hipEventRecord(ev_pll1, stream); (void)hipStreamWaitEvent(side, ev_pll1, 0); spare_moved = true; ps = side; }
          sub(ps, "pll_up", [&] {
          hipLaunchKernelGGL(k_pll_up, dim3(ngrp, S), dim3(64), 0, ps, d_pll_nodes.p, d_pll_G.p, d_pll_M.p, nck,
                             d_pll_PQ.p, d_pll_pre.p, d_pll_PQ2.p, ngrp2, d_pll_dstart2.p, d_flags.p, d_pll_sync.p,
                             d_pll_tick2.p);
          });
          }
          continue;
        }
#ifdef FMR_AB_PARTNERS
        hipLaunchKernelGGL(k_pll_nodes_a, dim3(ngrp, S), dim3(64), 0, stream, d_pll_nodes.p, d_pll_G.p, d_pll_M.p,
                           nck, d_pll_PQ.p, d_flags.p);
        hipLaunchKernelGGL(k_pll_nodes_a2, dim3(ngrp2, S), dim3(64), 0, stream, d_pll_PQ.p, ngrp, d_pll_PQ2.p,
                           d_flags.p);
        hipLaunchKernelGGL(k_pll_nodes_b, dim3(S), dim3(64), 0, stream, d_pll_PQ2.p, ngrp2, d_pll_dstart2.p,
                           d_flags.p);
        hipLaunchKernelGGL(k_pll_nodes_c2, dim3(ngrp2, S), dim3(64), 0, stream, d_pll_PQ.p, ngrp, d_pll_dstart2.p,
                           d_pll_dstart.p, d_flags.p);
        hipLaunchKernelGGL(k_pll_nodes_c, dim3(ngrp, S), dim3(64), 0, stream, d_pll_nodes.p, d_pll_G.p, d_pll_M.p,
                           nck, d_pll_dstart.p, d_flags.p, pllc.minfreq, pllc.maxfreq, d_pll_gres.p);
#endif
      }
      // the serial fallback (a kernel that reads a flag and leaves unless the rounds gave up) goes with the lock logic in the
      // pipelined chain: it carries the lock counters on, as the late walk of the call before does on that stream
      if (walk_late && !spare_moved) { (void)hipEventRecord(ev_pll1, stream); (void)hipStreamWaitEvent(side, ev_pll1, 0); spare_moved = true; ps = side; }
      if (pilot_shift)
        hipLaunchKernelGGL(k_pll_fallback<true>, dim3(S), dim3(64), 0, ps, k.base, base_stride, H_b, bt,
                           k.raw, base_stride, H_b, d_atan.p, pllc, k.stereo_blk, d_state.p, S, d_flags.p);
      else
        hipLaunchKernelGGL(k_pll_fallback<false>, dim3(S), dim3(64), 0, ps, k.base, base_stride, H_b, bt,
                           k.raw, base_stride, H_b, d_atan.p, pllc, k.stereo_blk, d_state.p, S, d_flags.p);
    });
    if (rc_agc) return rc_agc;
    HIPCHK(hipGetLastError());    // a launch of the rounds above that could not be enqueued
    // lock logic / PPS / state commit beside the audio chain (needed again only by fm_out)
    if (spare_moved) HIPCHK(hipEventRecord(ev_pll, side));
    else {
      HIPCHK(hipEventRecord(ev_pll, stream));
      HIPCHK(hipStreamWaitEvent(side, ev_pll, 0));
    }
    const size_t fin_ballast = side_ballast();
    int *const walk_go = d_walk_go.p + (size_t)wpar * S;
    const fm_mpx_t *const base_l = k.base; int *const stereo_blk_l = k.stereo_blk;
    const BlockTab bt_l = bt; const ChunkTab ct_l = ct;
    auto walk = [=]() -> int {
      timed_on(side, "pll_finish", [&] {
        hipLaunchKernelGGL(k_pll_finish, dim3(S), dim3(64), (env.x_ballast & 1) ? (size_t)kBallastBytes : fin_ballast, side, base_l, base_stride, H_b, bt_l, ct_l, d_atan.p,
                           pllc, (int)pilot_shift, d_pll_nodes.p, d_pll_G.p, ck_wraps_now, ck_mask_now, mask_words,
                           d_blk_wraps.p, d_blk_level.p, stereo_blk_l, d_state.p, d_flags.p,
                           walk_late ? walk_go : (const int *)nullptr);
      });
      if (walk_late) HIPCHK(hipEventRecord(ev_fin, side));
      return FMR_OK;
    };
    timed_on(side, walk_late ? "pll_commit" : "pll_blocks", [&] {
      hipLaunchKernelGGL(k_pll_blocks, dim3((nb + 3) / 4, S), dim3(256), fin_ballast, side, bt, ct, d_pll_G.p,
                         ck_wraps_now, d_blk_wraps.p, d_blk_level.p, d_flags.p);
      if (walk_late)
        hipLaunchKernelGGL(k_pll_commit, dim3(S), dim3(FMR_COMMIT_THREADS), (env.x_ballast & 2) ? (size_t)kBallastBytes : fin_ballast, side, bt, ct, pllc, d_pll_G.p, d_blk_wraps.p,
                           d_blk_level.p, d_state.p, d_flags.p, walk_go);
    });
    if (walk_late) {
      // (the tail waits for this stream's statistics and AGC itself: ev_fin, recorded behind the late walk, covers neither)
      walk_job = walk; walk_pending = true;
      fin_on_side = true;
      return FMR_OK;
    }
    if (int rcw = walk()) return rcw;
    // one event for everything beside the main stream: this stream's own work (statistics, lock logic) and the
    // AGC stream's -- the main stream then waits once, before the output mux, instead of four times
    if (agc_on_side && !agc_deferred) { HIPCHK(hipStreamWaitEvent(side, ev_agc, 0)); fin_covers_all = true; }
    HIPCHK(hipEventRecord(ev_fin, side));
    fin_on_side = true;
  }
  return FMR_OK;
}

// FmDecoder: equaliser, discriminator, statistics, pilot PLL, audio resampler + tail, DC block + mux
int fmr_chain::run_fm(CallCtx &k) {
  auto &d_iq = k.d_iq; auto &stride = k.stride; auto &nb = k.nb; auto &d_aud = k.d_aud; auto &astride = k.astride;
  auto &audio_len = k.audio_len; auto &N_in = k.N_in; auto &t_au_len = k.t_au_len; auto &N_if = k.N_if;
  auto &use_fused = k.use_fused; auto &ifbuf = k.ifbuf; auto &N_au = k.N_au; auto &any_mpf = k.any_mpf;
  auto &amA_prev = k.amA_prev; auto &akB_prev = k.akB_prev; auto &an_prev = k.an_prev;
  auto &fused_n_tiles = k.fused_n_tiles; auto &fused_kb_ref = k.fused_kb_ref; auto &bt = k.bt;
  auto &if_stride = k.if_stride; auto &rms_in_disc = k.rms_in_disc; auto &xin = k.xin; auto &x_stride = k.x_stride;
  auto &x_off = k.x_off; auto &disc_gain = k.disc_gain; auto &agc_on_side = k.agc_on_side;
  auto &agc_deferred = k.agc_deferred; auto &enqueue_agc = k.enqueue_agc;
  auto add_halo = [&](void *buf, long long stride_e, int H, long long N, int words = 2) { k.add_halo(buf, stride_e, H, N, words); };
  if (any_mpf) {
    timed("mpf", [&] {
      // the chain-and-helpers form (k_mpf4: no barrier inside a chunk)
      constexpr int NG4 = FMR_MPF_CH / 4 + 2;
      const int tpl4 = mpf_N <= 320 ? 5 : mpf_N <= 640 ? 10 : 20;
      const size_t lds4 = sizeof(float2) * ((size_t)mpf_N + FMR_MPF_CH + 8) + sizeof(float) * NG4 + sizeof(float2) * FMR_MPF_CH +
                          sizeof(double) * NG4 + sizeof(float2) * (tpl4 <= 10 ? 16 : 8) * 64 * (size_t)tpl4 + sizeof(int) * 16 +
                          sizeof(float2);
      auto go4 = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        hipLaunchKernelGGL(kern, dim3(S), dim3(256), lds4, stream, xin, x_stride, x_off, d_gain.p, (long long)max_if,
                           bt, d_mpf.p, (long long)max_if, d_mpf_coeff.p, d_mpf_state.p, mpf_N, mpf_ref,
                           d_mpf_ok.p, d_state.p, agc_beside_mpf ? d_agc_progress.p : (const unsigned long long *)nullptr,
                           kAgcWaitTicks);
      };
      if (mpf_N <= 64 * 5) go4(k_mpf4<5>);
      else if (mpf_N <= 64 * 10) go4(k_mpf4<10>);
      else if (mpf_N <= 64 * 20) go4(k_mpf4<20>);                                     // N <= 1280
      else set_err("equaliser length out of range");
    });
  }
  if (agc_beside_mpf) {
    // the discriminator multiplies the gains into the blocks the equaliser passed over (warm-up, resets), and the AGC's
    // state must be committed before the next call: the AGC kernel finished long ago (it is three times faster)
    HIPCHK(hipStreamWaitEvent(stream, ev_agc, 0));
  }
  const long long base_stride = H_b + (long long)max_if;   // pre-de-emphasis buffers
  const long long de_stride = H_a + (long long)max_if;     // de-emphasised copies feeding the audio resampler
  if (!k.fused_disc && !k.fir_disc)     // (the fused front end's / the IF filter's discriminator epilogue has already written the MPX and the block statistics)
  timed("disc", [&] {
    hipLaunchKernelGGL((k_disc<256, fm_mpx_t>), dim3(nb, S), dim3(256), 0, stream, xin, x_stride, x_off, disc_gain,
                       (long long)max_if, any_mpf ? d_mpf.p : (float2 *)nullptr, (long long)max_if, d_mpf_ok.p, bt,
                       disc_nf, disc_bound, d_dec.p, (long long)max_if, k.base, base_stride, H_b,
                       d_bb_mean_blk.p, d_bb_rms_blk.p, d_state.p, rms_in_disc ? d_if_rms_blk.p : (float *)nullptr);
  });
  // "the MPX is there": what the side streams start from.  Behind the fused front end on the decoder's own stream that is
  // the event recorded behind it already (a second marker on the critical stream costs what a small kernel costs).
  k.ev_mpx = (pipelined && k.fused_disc) ? ev_fe[k.par] : ev_disc;
  if (k.ev_mpx == ev_disc) HIPCHK(hipEventRecord(ev_disc, stream));
  if (k.tail_deferred) { k.tail_deferred = false; if (int rc = flush_tail(ev_disc)) return rc; }
  HIPCHK(hipStreamWaitEvent(side, k.ev_mpx, 0));
  if (use_fused && !pipelined)      // input history for the next call's front end: off the critical path (the next
    timed_on(side, "in_halo", [&] {   // front end waits for this stream's table kernels anyway)
      hipLaunchKernelGGL((k_update_in_halo<256, 0>), dim3(1, S), dim3(256), 0, side, d_in_halo.p, H_in, d_iq, (long long)stride, N_in);
    });
  const size_t stats_ballast = side_ballast();
  timed_on(side, "stats", [&] {     // (fused front end: the block values are summed from its partial sums on the fly)
    hipLaunchKernelGGL(k_stats, dim3(S), dim3(FMR_STATS_THREADS), stats_ballast, side, bt, d_if_rms_blk.p, d_bb_mean_blk.p,
                       d_bb_rms_blk.p, d_state.p, S, (int)!(pipelined && k.fused_disc),   // (the front-end stage commits its own phase)
                       (k.fused_disc || k.fir_tail) ? k.part : (const FusedPart *)nullptr, k.fir_tail ? 8 * k.fir_tiles : fused_n_tiles,
                       k.fir_tail ? 0 : fused_kb_ref);
  });
  HIPCHK(hipEventRecord(ev_stats, side));
  disc_commit_on_side = !(pipelined && k.fused_disc);
  bool fin_on_side = false, fin_covers_all = false;
  // ---------------------------------------------------- audio resampler + tail
  TailCtx t{};
  t.base = k.base; t.raw = k.raw; t.stereo_blk = k.stereo_blk; t.N_if = N_if; t.N_au = N_au; t.nb = nb; t.bt = bt;
  t.d_aud = d_aud; t.astride = (long long)astride; t.amA_prev = amA_prev; t.akB_prev = akB_prev;
  t.nch = stereo ? 2 : 1;
  for (int b = 0; b < nb; b++) t.au_max = std::max(t.au_max, t_au_len[b]);
  t.count_am = (int)(arsc.mA - amA_prev);
  if ((size_t)t.count_am > max_amid || (size_t)N_au > max_au) { set_err("internal audio capacity exceeded"); return FMR_ERR_CAPACITY; }
  t.a_top0 = (long long)ars.D * amA_prev + ars.ca() - an_prev;
  // fused de-emphasis + stage A: the tile (warm-up + (TOUT-1) D + NA samples) must fit BLOCK * LPL LDS slots;
  // at most 4 * DE_BLOCK outputs per tile: every lane then owns exactly one run of 4 outputs in the FIR phase
  t.de_tout = std::min(4 * kDeBlock, ((kDeBlock * FMR_DE_LPL - FMR_DE_WARMUP - ars.NA - ars.D) / ars.D) & ~3);
  t.de_fused = !serial_mode && t.de_tout >= 64;
  t.dc_nc = (int)((N_au + C_DC - 1) / C_DC);
  t.dk.b0 = dcblock.b0; t.dk.b1 = dcblock.b1; t.dk.b2 = dcblock.b2; t.dk.a1 = dcblock.a1; t.dk.a2 = dcblock.a2;
  for (int j = 0; j < 4; j++) t.dk.ac[j] = dc_ac[j];
  for (int lv = 0; lv < 6; lv++) for (int j = 0; j < 4; j++) t.dk.agp[lv][j] = dc_agp[lv][j];
  const long long base_stride_ = base_stride;
  const long long am_stride = H_am + (long long)max_amid;
  const long long a1_stride = H_pc + (long long)max_au;
  // Channel 0 (mono = L+R) does not depend on the PLL: in the in-order chain with stereo on it runs on the AGC stream while
  // the PLL iterates, and only L-R stays behind the PLL on the decoder stream.  In the pipelined chain both channels belong
  // to the tail STAGE (its own stream, a call behind the PLL stage): the mono channel's buffers are still being read by the
  // previous call's DC block and mux while this call's PLL iterates.
  t.can_split = stereo && !serial_mode && t.de_fused && (ars.LB == 3 && ars.MB == 8) && n_pilotcut <= FMR_PCUT_MAXTAPS;
  t.ev_mpx = k.ev_mpx;
  const bool split_mono = t.can_split && !pipelined;
  bool mono_enqueued = false;
  if (stereo) {
    auto tail_fn = [&](hipStream_t st, int ch_base, int nch_l) { tail_channels(t, st, ch_base, nch_l); };
    if (int rcp = run_fm_pll(k, base_stride_, split_mono, tail_fn, mono_enqueued, fin_on_side, fin_covers_all)) return rcp;
  }
  if (agc_deferred) { agc_deferred = false; if (int rca = enqueue_agc(nullptr)) return rca; }   // PLL path not taken
  if (k.fe_post) { k.fe_post(); k.fe_post = nullptr; }
  t.fin_on_side = fin_on_side; t.fin_covers_all = fin_covers_all; t.agc_on_side = agc_on_side; t.mono_enqueued = mono_enqueued;

  if (!pipelined) {
    if (fir_enable) add_halo(ifbuf, if_stride, H_if, N_if);
    add_halo(k.base, base_stride, H_b, N_if, 1);
    if (stereo) add_halo(k.raw, base_stride, H_b, N_if);
  }       // (pipelined: the halos of the ring slots are carried over at the head of the next call, run_tables)
  if (!t.de_fused || debug_taps) {     // (the fused de-emphasis keeps the 384 kHz signal in LDS: these are taps then)
    add_halo(d_base_de.p, de_stride, H_a, N_if);
    if (stereo) add_halo(d_raw_de.p, de_stride, H_a, N_if);
  }
  add_halo(d_am0.p, am_stride, H_am, t.count_am);
  if (stereo) add_halo(d_am1.p, am_stride, H_am, t.count_am);
  add_halo(d_a10.p, a1_stride, H_pc, N_au);
  if (stereo) add_halo(d_a11.p, a1_stride, H_pc, N_au);
  if (audio_len) for (int b = 0; b < nb; b++) audio_len[b] = (uint32_t)(stereo ? 2 * t_au_len[b] : t_au_len[b]);
  if (pipelined) {
    // end of the PLL stage on the decoder stream: the tail of this call starts from it.  The tail stage itself is enqueued behind the NEXT call's front end (or by whatever
    // synchronises the chain first): it then runs beside that call's PLL stage and leaves the front end the whole chip.
    if (!stereo) HIPCHK(hipEventRecord(ev_pll, stream));
    t.ht = k.ht; k.ht.n = 0;
    t.seq = pipe_seq;
    tail_job = t;
    tail_pending = true;
    return FMR_OK;
  }
  return tail_stage(t, stream);
}

// Per-channel part of the audio tail (de-emphasis + audio resampler + pilot cut + DC-block pass 1) for channels
// ch_base .. ch_base + nch_l - 1 on stream st.
void fmr_chain::tail_channels(const TailCtx &t, hipStream_t st, int ch_base, int nch_l) {
  const long long base_stride = H_b + (long long)max_if, de_stride = H_a + (long long)max_if;
  const long long am_stride = H_am + (long long)max_amid, a1_stride = H_pc + (long long)max_au;
  const int count_am = t.count_am, de_tout = t.de_tout, dc_nc = t.dc_nc;
  const long long N_if = t.N_if, N_au = t.N_au, a_top0 = t.a_top0, amA_prev = t.amA_prev, akB_prev = t.akB_prev;
  const BlockTab &bt = t.bt;
  const int nb = t.nb;
  constexpr int DE_BLOCK = kDeBlock, DE_SLOTS = DE_BLOCK * FMR_DE_LPL;
  if (t.de_fused) {
    if (count_am > 0) {
      timed_on(st, "deemph_decim", [&] {
        const int tiles = (count_am + de_tout - 1) / de_tout;
        const size_t lds = sizeof(double) * (size_t)(DE_BLOCK * FMR_DE_RUN + 1);      // (an even run length carries a pad word: de_idx)
        auto go = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(tiles, S, nch_l), dim3(DE_BLOCK), lds, st, t.base, t.raw, base_stride, H_b,
                             (int)N_if, deemph.b0, deemph.a1, de_scan, 1, (int)(stereo && !pilot_shift), d_ahA.p,
                             ars.NA, ars.D, a_top0, count_am, de_tout, d_am0.p, d_am1.p, am_stride, H_am,
                             debug_taps ? d_base_de.p : (double *)nullptr,
                             debug_taps ? d_raw_de.p : (double *)nullptr, de_stride, H_a, ch_base);
        };
        if (ars.NA == 59 && ars.D == 3) go(k_deemph_decim<DE_BLOCK, 59, 3>);     // 384 kHz -> 48 kHz
        else go(k_deemph_decim<DE_BLOCK, 0, 0>);
      });
    }
  } else {
    // ---- de-emphasis by warm-up, out of place: base/raw -> base_de/raw_de
    timed_on(st, "deemph", [&] {
      const int nt = (int)((N_if + C_DE - 1) / C_DE);
      hipLaunchKernelGGL(k_deemph_par<C_DE>, dim3((nt + 63) / 64, S, nch_l), dim3(64), 0, st, t.base, t.raw,
                         base_stride, H_b, d_base_de.p, d_raw_de.p, de_stride, H_a, (int)N_if, deemph.b0, deemph.a1, 1,
                         (int)(stereo && !pilot_shift));
    });
    if (count_am > 0) {
      timed_on(st, "aud_decim", [&] {
        hipLaunchKernelGGL(k_aud_decim<128>, dim3((count_am + 127) / 128, S, nch_l), dim3(128), 0, st, d_base_de.p,
                           d_raw_de.p, de_stride, H_a, d_ahA.p, ars.NA, ars.D, a_top0, count_am, d_am0.p, d_am1.p,
                           am_stride, H_am);
      });
    }
  }
  if (N_au > 0) {
    timed_on(st, "aud_poly", [&] {
      if (ars.LB == 3 && ars.MB == 8) {
        // period form: one lane per period (3 outputs), taps through the scalar cache
        constexpr int BLP = 256;
        const long long P_first = akB_prev / 3, P_last = (akB_prev + N_au - 1) / 3;
        const int tiles = (int)((P_last - P_first) / BLP + 1);
        const int lx = (BLP - 1) * (int)ars.MB + (int)((2 * ars.MB) / 3) + ars.TB;
        int ni_pad = (lx + (int)ars.MB - 1) / (int)ars.MB + 1;
        if ((ni_pad & 1) == 0) ni_pad++;
        hipLaunchKernelGGL((k_aud_poly2<BLP, 3, 8>), dim3(tiles, S, nch_l), dim3(BLP), sizeof(double) * (size_t)ars.MB * ni_pad,
                           st, d_am0.p, d_am1.p, am_stride, amA_prev - H_am, d_ahB.p, ars.TB,
                           akB_prev, (int)N_au, d_a10.p, d_a11.p, a1_stride, H_pc, ni_pad, H_am + count_am, ch_base);
      } else {
        hipLaunchKernelGGL(k_aud_poly<128>, dim3((unsigned)((N_au + 127) / 128), S, nch_l), dim3(128), 0, st,
                           d_am0.p, d_am1.p, am_stride, amA_prev - H_am, d_ahB.p, ars.TB, (unsigned)ars.LB,
                           (unsigned)ars.MB, (unsigned long long)akB_prev * ars.MB, (int)N_au, d_a10.p, d_a11.p,
                           a1_stride, H_pc);
      }
    });
    timed_on(st, "pilotcut", [&] {
      // (a 65536-sample block is 314 or 315 audio samples: the one-tile form takes 4.6 KB of LDS instead of 12.3, and thirteen
      // instead of five of its workgroups fit beside a PLL pass on a compute unit)
      if (n_pilotcut <= FMR_PCUT_MAXTAPS && t.au_max <= 320)
        hipLaunchKernelGGL((k_pilotcut2<320, 320>), dim3(nb, S, nch_l), dim3(320), 0, st, d_a10.p, d_a11.p,
                           a1_stride, H_pc, bt, d_pilotcut.p, n_pilotcut, d_pc0.p, d_pc1.p, (long long)max_au, 1.0, ch_base);
      else if (n_pilotcut <= FMR_PCUT_MAXTAPS)
        hipLaunchKernelGGL((k_pilotcut2<320, 1280>), dim3(nb, S, nch_l), dim3(320), 0, st, d_a10.p, d_a11.p,
                           a1_stride, H_pc, bt, d_pilotcut.p, n_pilotcut, d_pc0.p, d_pc1.p, (long long)max_au, 1.0, ch_base);
      else
        hipLaunchKernelGGL(k_pilotcut<128>, dim3(nb, S, nch_l), dim3(128), 0, st, d_a10.p, d_a11.p, a1_stride, H_pc,
                           bt, d_pilotcut.p, n_pilotcut, d_pc0.p, d_pc1.p, (long long)max_au);
    });
  }
  if (N_au > 0 && !serial_mode)
    timed_on(st, "dc_pass1", [&] {
      hipLaunchKernelGGL(k_dc_pass1<C_DC>, dim3((dc_nc + 63) / 64, S, nch_l), dim3(64), 0, st, d_pc0.p, d_pc1.p,
                         (long long)max_au, (int)N_au, t.dk, d_dc_G.p, dc_nc, ch_base);
    });
}

// The audio tail of one call on stream ts: the channels the PLL stage has not already run beside itself, the DC block's
// node pass, the output mux, and the joins with what ran beside the decoder stream (statistics, AGC, lock logic).
int fmr_chain::tail_stage(const TailCtx &t, hipStream_t ts) {
  const int nch = t.nch, dc_nc = t.dc_nc;
  const long long N_au = t.N_au;
  if (t.mono_enqueued) tail_channels(t, ts, 1, 1);
  else tail_channels(t, ts, 0, nch);
  if (t.mono_enqueued) HIPCHK(hipStreamWaitEvent(ts, ev_mono, 0));   // DC-block node pass needs both channels
  if (N_au > 0) {
    if (t.fin_on_side && serial_mode) HIPCHK(hipStreamWaitEvent(ts, ev_fin, 0));
    if (serial_mode) {
      timed_on(ts, "fm_out", [&] {
        hipLaunchKernelGGL(k_fm_out, dim3(S), dim3(64), 0, ts, d_pc0.p, d_pc1.p, (long long)max_au, t.bt, (int)N_au,
                           dcblock.b0, dcblock.b1, dcblock.b2, dcblock.a1, dcblock.a2, (int)stereo, (int)pilot_shift,
                           t.stereo_blk, t.d_aud, t.astride, d_state.p);
      });
    } else {
      // ---- DC block by linear multiple shooting + output mux
      timed_on(ts, "fm_out", [&] {
        const int dc_nw = std::max(1, std::min(FMR_DC_MAXW, (dc_nc + 64 * FMR_DC_K - 1) / (64 * FMR_DC_K)));
        hipLaunchKernelGGL(k_dc_nodes, dim3(S * nch), dim3(64 * dc_nw), 0, ts, d_dc_G.p, d_dc_start.p, dc_nc, t.dk,
                           d_state.p, S, nch);
        if (t.fin_on_side) (void)hipStreamWaitEvent(ts, ev_fin, 0);   // only the mux needs the lock flags
        hipLaunchKernelGGL(k_dc_pass2_mux<C_DC>, dim3((dc_nc + 63) / 64, S), dim3(64), 0, ts, d_pc0.p, d_pc1.p,
                           (long long)max_au, t.bt, (int)N_au, t.dk, d_dc_start.p, dc_nc, (int)stereo, (int)pilot_shift,
                           t.stereo_blk, t.d_aud, t.astride, d_state.p);
      });
    }
  }
  if (!(t.fin_covers_all && N_au > 0)) {        // (otherwise the wait before the output mux covered all three)
    HIPCHK(hipStreamWaitEvent(ts, ev_stats, 0));
    if (t.agc_on_side) HIPCHK(hipStreamWaitEvent(ts, ev_agc, 0));
    if (t.fin_on_side) HIPCHK(hipStreamWaitEvent(ts, ev_fin, 0));
  }
  return FMR_OK;
}

// Pipelined chain: enqueue the tail stage of the last decoded call on the tail stream, behind `gate` (the front end of
// the call that follows it; null: nothing to wait for but the call's own PLL stage).
int fmr_chain::flush_walk() {
  if (!walk_pending) return FMR_OK;
  walk_pending = false;
  const int rc = walk_job();
  walk_job = nullptr;
  return rc;
}
int fmr_chain::flush_tail(hipEvent_t gate) {
  const int rc_walk = flush_walk();       // (the tail's output mux waits for its call's walk: enqueued first, or the wait finds an older record)
  if (rc_walk != FMR_OK && !tail_pending) return rc_walk;
  if (!tail_pending) return FMR_OK;
  tail_pending = false;
  const int rc = enqueue_tail(gate);
  if (rc != FMR_OK)      // the slot of the ring must still be released, or the calls that reuse it wait for a mark that never comes
    hipLaunchKernelGGL(k_signal_host, dim3(1), dim3(1), 0, tail, &h_marks[1], tail_job.seq);
  return rc != FMR_OK ? rc : rc_walk;     // (a walk that could not be enqueued leaves that call's lock flags and PPS events stale: an error of this call)
}
int fmr_chain::enqueue_tail(hipEvent_t gate) {
  TailCtx t = tail_job;
  if (!gate && t.can_split && !t.mono_enqueued && t.ev_mpx) {
    // Drain (a synchronising call, no front end follows): the mono channel does not depend on the PLL -- it starts from the
    // MPX, behind the tail of the call before (same stream, enqueued when this call's front end was), beside this call's
    // PLL passes.  A caller that synchronises after every call gets its audio 0.1 ms earlier; a timed region of K calls
    // ends that much earlier.  (In steady state the tail waits for the NEXT front end instead: see run_fm.)
    HIPCHK(hipStreamWaitEvent(tail, t.ev_mpx, 0));
    tail_channels(t, tail, 0, 1);
    HIPCHK(hipEventRecord(ev_mono, tail));
    t.mono_enqueued = true;
  }
  HIPCHK(hipStreamWaitEvent(tail, ev_pll, 0));
  if (gate) HIPCHK(hipStreamWaitEvent(tail, gate, 0));
#ifndef FMR_DIAG_NO_TAIL      // (diagnostic builds under tools/: what the step costs without the tail's kernels beside the next front end)
  if (int rc = tail_stage(t, tail)) return rc;
#endif
  if (t.ht.n) {
    timed_on(tail, "shift_halo", [&] { hipLaunchKernelGGL(k_shift_halo<256>, dim3(t.ht.n, S), dim3(256), 0, tail, t.ht); });
  }
  hipLaunchKernelGGL(k_signal_host, dim3(1), dim3(1), 0, tail, &h_marks[1], t.seq);
  HIPCHK(hipGetLastError());
  return FMR_OK;
}

// NbfmDecoder
int fmr_chain::run_nbfm(CallCtx &k) {
  auto &nb = k.nb; auto &d_aud = k.d_aud; auto &astride = k.astride; auto &audio_len = k.audio_len;
  auto &t_au_len = k.t_au_len; auto &N_if = k.N_if; auto &ifbuf = k.ifbuf; auto &bt = k.bt;
  auto &if_stride = k.if_stride; auto &xin = k.xin; auto &x_stride = k.x_stride; auto &x_off = k.x_off;
  auto add_halo = [&](void *buf, long long stride_e, int H, long long N, int words = 2) { k.add_halo(buf, stride_e, H, N, words); };
  // NbfmDecoder (NbfmDecode.cpp:47-96): discriminator on the AGC'd IF, statistics, 63-tap audio FIR (same
  // block-head path as the FM pilot cut: LowPassFilterFirAudio), -3 dB.  No resampling: audio block = IF block.
  const long long base_stride = H_b + (long long)max_if;
  timed("disc", [&] {
    hipLaunchKernelGGL((k_disc<256, double>), dim3(nb, S), dim3(256), 0, stream, xin, x_stride, x_off, d_gain.p,
                       (long long)max_if, (float2 *)nullptr, (long long)max_if, d_mpf_ok.p, bt, disc_nf, disc_bound,
                       d_dec.p, (long long)max_if, d_base.p, base_stride, H_b, d_bb_mean_blk.p, d_bb_rms_blk.p,
                       d_state.p, (float *)nullptr);
  });
  timed("stats", [&] {
    hipLaunchKernelGGL(k_stats, dim3(S), dim3(FMR_STATS_THREADS), 0, stream, bt, d_if_rms_blk.p, d_bb_mean_blk.p,
                       d_bb_rms_blk.p, d_state.p, S, 1);
  });
  timed("nbfm_audio", [&] {
    hipLaunchKernelGGL((k_pilotcut2<320, 1280>), dim3(nb, S, 1), dim3(320), 0, stream, d_base.p, (double *)nullptr,
                       base_stride, H_b, bt, d_pilotcut.p, n_pilotcut, d_aud, (double *)nullptr, (long long)astride,
                       0.70794578438413791, 0);       // std::pow(10.0, -3.0 / 20.0), NbfmDecode.cpp:91
  });
  add_halo(ifbuf, if_stride, H_if, N_if);
  add_halo(d_base.p, base_stride, H_b, N_if);
  if (audio_len) for (int b = 0; b < nb; b++) audio_len[b] = (uint32_t)t_au_len[b];
  return FMR_OK;
}

// AmDecoder (AM / DSB / USB / LSB / CW / WSPR)
int fmr_chain::run_am(CallCtx &k) {
  auto &nb = k.nb; auto &d_aud = k.d_aud; auto &astride = k.astride; auto &audio_len = k.audio_len;
  auto &t_au_len = k.t_au_len; auto &N_if = k.N_if; auto &ifbuf = k.ifbuf; auto &bt = k.bt;
  auto &if_stride = k.if_stride; auto &xin = k.xin; auto &x_stride = k.x_stride; auto &x_off = k.x_off;
  auto add_halo = [&](void *buf, long long stride_e, int H, long long N, int words = 2) { k.add_halo(buf, stride_e, H, N, words); };
  timed("am_demod", [&] {
    hipLaunchKernelGGL(k_am_demod<256>, dim3(nb, S), dim3(256), 0, stream, xin, x_stride, x_off, d_gain.p,
                       (long long)max_if, bt, (int)(mode != FMR_MODE_AM), d_dec.p, (long long)max_if, d_base.p,
                       (long long)max_if, d_bb_mean_blk.p, d_bb_rms_blk.p);
  });
  timed("stats", [&] {
    hipLaunchKernelGGL(k_stats, dim3(S), dim3(FMR_STATS_THREADS), 0, stream, bt, d_if_rms_blk.p, d_bb_mean_blk.p,
                       d_bb_rms_blk.p, d_state.p, S, 0);
  });
  timed("am_tail", [&] {
    const bool par_tail = !serial_mode;
    if (par_tail) {
      // DC block -> AfSimpleAgc -> de-emphasis in time-parallel form (kernels_par.hpp); the serial kernel below only
      // runs for a stream whose Newton rounds did not converge
      const int nc = (int)((N_if + C_AM - 1) / C_AM);
      const AfAgcCoef af{1.0, 1.5, af_ref, af_rate};      // AfSimpleAgc(1.0, 1.5, reference, rate): AmDecode.cpp:54-66
      hipLaunchKernelGGL(k_dc_pass1<C_AM>, dim3((nc + 63) / 64, S, 1), dim3(64), 0, stream, d_base.p, (const double *)nullptr,
                         (long long)max_if, (int)N_if, am_dk, d_dc_G.p, nc, 0);
      const int dc_nw = std::max(1, std::min(FMR_DC_MAXW, (nc + 64 * FMR_DC_K - 1) / (64 * FMR_DC_K)));
      hipLaunchKernelGGL(k_dc_nodes, dim3(S), dim3(64 * dc_nw), 0, stream, d_dc_G.p, d_dc_start.p, nc, am_dk, d_state.p, S, 0);
      hipLaunchKernelGGL(k_af_begin, dim3(S), dim3(256), 0, stream, d_flags.p, d_af_nodes.p, nc, d_state.p, d_af_tick.p);
      for (int it = 0; it < K_AF_ITERS; it++)
        hipLaunchKernelGGL(k_af_round<C_AM>, dim3((nc + 63) / 64, S), dim3(64), 0, stream, d_base.p, (long long)max_if, (int)N_if,
                           am_dk, d_dc_start.p, af, d_af_nodes.p, d_af_G.p, d_af_M.p, d_af_out.p, (long long)max_if, nc,
                           d_state.p, d_flags.p, d_af_tick.p);
      const int ncd = (int)((N_if + C_AM_DE - 1) / C_AM_DE);
      hipLaunchKernelGGL(k_am_deemph_out<C_AM_DE>, dim3((ncd + 63) / 64, S), dim3(64), 0, stream, d_af_out.p, (long long)max_if,
                         (int)N_if, am_deemph.b0, am_deemph.a1, (int)(mode == FMR_MODE_AM), d_aud, (long long)astride,
                         d_state.p, d_flags.p);
      hipLaunchKernelGGL(k_am_commit, dim3((S + 63) / 64), dim3(64), 0, stream, d_state.p, d_flags.p, S);
    }
    hipLaunchKernelGGL(k_am_tail, dim3((S + 63) / 64), dim3(64), 0, stream, d_base.p, (long long)max_if, (int)N_if,
                       am_dcblock.b0, am_dcblock.b1, am_dcblock.b2, am_dcblock.a1, am_dcblock.a2, 1.0, 1.5, af_ref,
                       af_rate, am_deemph.b0, am_deemph.a1, (int)(mode == FMR_MODE_AM), d_aud, (long long)astride,
                       d_state.p, S, par_tail ? &d_flags.p->af_converged : (const int *)nullptr, (int)sizeof(IterFlags),
                       par_tail ? &d_flags.p->af_fallback : (int *)nullptr);
  });
  add_halo(ifbuf, if_stride, H_if, N_if);
  if (audio_len) for (int b = 0; b < nb; b++) audio_len[b] = (uint32_t)t_au_len[b];
  return FMR_OK;
}

// ============================================================================
// C-ABI
// ============================================================================
extern "C" {

const char *fmr_last_error(void) { return g_err.c_str(); }
const char *fmr_version(void) { return "fmradion_amd 0.4 (gfx950)"; }

int fmr_create_sized(const fmr_config *cfg, size_t cfg_size, fmr_chain **out) {
  if (!cfg || !out) return FMR_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg_size > sizeof(fmr_config)) {
    set_err("fmr_create_sized: the caller's fmr_config has %zu bytes, this library's %zu: the caller is newer than the library", cfg_size, sizeof(fmr_config));
    return FMR_ERR_BAD_ARG;
  }
  fmr_config full;
  memset(&full, 0, sizeof full);               // fields the caller's header does not have: 0 = "as before"
  memcpy(&full, cfg, cfg_size);
  if (cfg_size >= offsetof(fmr_config, struct_size) + sizeof(unsigned)) full.struct_size = 0;     // (stated through the argument)
  return fmr_create(&full, out);
}

int fmr_create(const fmr_config *cfg, fmr_chain **out) {
  if (!cfg || !out) return FMR_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->struct_size != 0 && cfg->struct_size != sizeof(fmr_config)) {
    set_err("fmr_config.struct_size %u is not this library's %zu: caller and library were built against different headers "
            "(or the struct was not zero-initialised)", cfg->struct_size, sizeof(fmr_config));
    return FMR_ERR_BAD_ARG;
  }
  fmr_chain *c = new fmr_chain();
  const int rc = c->init(cfg);
  if (rc != FMR_OK) { delete c; return rc; }
  if (c->sync_all() != FMR_OK) { delete c; return FMR_ERR_HIP; }
  *out = c;
  return FMR_OK;
}

void fmr_destroy(fmr_chain *c) { delete c; }

long long fmr_resampler_info(const fmr_chain *c, int which) {
  if (!c || !c->has_rs) return -1;
  switch (which) {
  case 0: return c->rs.D;
  case 1: return c->rs.NA;
  case 2: return c->rs.LB;
  case 3: return c->rs.MB;
  case 4: return c->rs.TB;
  case 5: return c->rs.LT;
  }
  return -1;
}

static long long design_taps_out(const ResamplerDesign &d, int stage, double *taps, long long cap, long long *info);
long long fmr_design_taps_class(double in_rate, double out_rate, int resampler_class, int stage, double *taps,
                                long long cap, long long *info) {
  ResamplerDesign d;
  const bool r8b = resampler_class == FMR_RESAMPLER_R8B;
  if (resampler_class != FMR_RESAMPLER_FAST && !r8b) { set_err("unknown resampler_class %d", resampler_class); return FMR_ERR_BAD_ARG; }
  if (!(r8b ? d.design(in_rate, out_rate, kR8bAtten, kR8bPassFrac, true) : d.design(in_rate, out_rate, kIfAtten))) {
    set_err("resampling ratio outside the design range");
    return FMR_ERR_UNSUPPORTED;
  }
  return design_taps_out(d, stage, taps, cap, info);
}
long long fmr_design_taps(double in_rate, double out_rate, double atten_db, int stage, double *taps, long long cap,
                          long long *info) {
  ResamplerDesign d;
  if (!d.design(in_rate, out_rate, atten_db)) { set_err("resampling ratio outside the design range"); return FMR_ERR_UNSUPPORTED; }
  return design_taps_out(d, stage, taps, cap, info);
}
static long long design_taps_out(const ResamplerDesign &d, int stage, double *taps, long long cap, long long *info) {
  if (info) { info[0] = d.D; info[1] = d.NA; info[2] = d.LB; info[3] = d.MB; info[4] = d.TB; info[5] = d.LT; }
  const std::vector<double> &h = stage ? d.hB : d.hA;
  if (!taps) return (long long)h.size();
  if ((long long)h.size() > cap) return FMR_ERR_CAPACITY;
  for (size_t i = 0; i < h.size(); i++) taps[i] = h[i];
  return (long long)h.size();
}

void *fmr_host_alloc(size_t bytes) {
  void *p = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err("no HIP device"); return nullptr; }
  const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
  if (e != hipSuccess) { set_err("hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
  return p;
}

void fmr_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int fmr_synchronize(fmr_chain *c) {
  if (!c) return FMR_ERR_BAD_ARG;
  if (int rc = c->sync_all()) return rc;
  return c->check_agc_sync();
}

// how many doubles per stream the blocks would produce, from copies of the count-law counters (nothing is advanced)
static size_t predict_audio(const fmr_chain *c, const uint32_t *block_len, int nb) {
  ResamplerCounter r = c->rsc, ar = c->arsc;
  size_t total = 0;
  for (int b = 0; b < nb; b++) {
    const long long n_if = c->has_rs ? r.advance(c->rs, block_len[b]) : (long long)block_len[b];
    if (!c->has_dec || n_if == 0) continue;
    if (c->mode == FMR_MODE_FM) total += (size_t)ar.advance(c->ars, n_if) * (c->stereo ? 2 : 1);
    else total += (size_t)n_if;
  }
  return total;
}

int fmr_process_blocks_device(fmr_chain *c, const float *d_iq, size_t stream_stride, const uint32_t *block_len,
                              int n_blocks, double *d_audio, size_t audio_stride, uint32_t *audio_len, int sync) {
  if (!c || !d_iq || !block_len || n_blocks < 1) return FMR_ERR_BAD_ARG;
  try {
    if (c->has_dec) {
      if (!d_audio) { set_err("d_audio is null"); return FMR_ERR_BAD_ARG; }
      if (n_blocks <= c->max_blocks && predict_audio(c, block_len, n_blocks) > audio_stride) {
        set_err("audio_stride too small for the audio these blocks produce (nothing was processed)");
        return FMR_ERR_CAPACITY;
      }
    }
    const int rc = c->run_cold_aware((const float2 *)d_iq, stream_stride, block_len, n_blocks, d_audio, audio_stride, audio_len);
    if (rc) return rc;
    if (sync) { if (int rcs = c->sync_all()) return rcs; return c->check_agc_sync(); }
    return FMR_OK;
  } catch (const std::exception &e) { set_err("exception: %s", e.what()); return FMR_ERR_HIP; }
}

int fmr_process_blocks(fmr_chain *c, const float *iq, size_t stream_stride, const uint32_t *block_len, int n_blocks,
                       double *audio, size_t audio_stride, uint32_t *audio_len) {
  if (!c || !iq || !block_len || n_blocks < 1) return FMR_ERR_BAD_ARG;
  try {
  size_t N_in = 0;
  for (int b = 0; b < n_blocks; b++) N_in += block_len[b];
  if (N_in > c->max_in) { set_err("input longer than max_block_len*max_blocks"); return FMR_ERR_CAPACITY; }
  if (n_blocks <= c->max_blocks && c->has_dec && audio) {
    // capacity is checked BEFORE any decoder state advances: a refused call can be retried with a larger buffer
    const size_t need = predict_audio(c, block_len, n_blocks);
    if (need > audio_stride) { set_err("audio capacity %zu too small: these blocks produce %zu doubles (nothing was processed)", audio_stride, need); return FMR_ERR_CAPACITY; }
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  if (N_in)
    HIPCHK(hipMemcpy2DAsync(c->d_in.p, (size_t)c->in_bps * c->max_in, iq, (size_t)c->in_bps * stream_stride,
                            (size_t)c->in_bps * N_in, c->S, hipMemcpyHostToDevice, c->stream));
  const size_t dstride = c->stereo ? 2 * c->max_au : c->max_au;
  std::vector<uint32_t> alen(n_blocks, 0);
  const int rc = c->run_cold_aware(c->d_in.p, c->max_in, block_len, n_blocks, c->d_audio.p, dstride, alen.data());
  if (rc) return rc;
  size_t total = 0;
  for (int b = 0; b < n_blocks; b++) { total += alen[b]; if (audio_len) audio_len[b] = alen[b]; }
  if (total > audio_stride && c->S > 1) { set_err("audio_stride too small"); return FMR_ERR_CAPACITY; }
  if (int rcf = c->flush_tail(nullptr)) return rcf;      // (pipelined chain: this call's tail stage, now)
  if (total && audio) {
    if (total > audio_stride) { set_err("audio capacity too small"); return FMR_ERR_CAPACITY; }
    HIPCHK(hipMemcpy2DAsync(audio, sizeof(double) * audio_stride, c->d_audio.p, sizeof(double) * dstride,
                            sizeof(double) * total, c->S, hipMemcpyDeviceToHost, c->pipelined ? c->tail : c->stream));
  }
  if (int rcs = c->sync_all()) return rcs;
  return c->check_agc_sync();
  } catch (const std::exception &e) { set_err("exception: %s", e.what()); return FMR_ERR_HIP; }
}

int fmr_fourth_convert(fmr_chain *c, const float *iq, size_t n, float *out_iq, int up, unsigned *index) {
  if (!c || !index || (n && (!iq || !out_iq))) return FMR_ERR_BAD_ARG;
  if (n == 0) return FMR_OK;
  if (n > c->max_in || c->in_fmt != 0) { set_err("fmr_fourth_convert: block longer than the chain's capacity, or raw-format chain"); return FMR_ERR_CAPACITY; }
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipMemcpyAsync(c->d_in.p, iq, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_fourth_shift, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, c->d_in.p,
                     c->d_in.p, (long long)n, *index & 3u, up);
  HIPCHK(hipMemcpyAsync(out_iq, c->d_in.p, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *index = up ? (unsigned)((*index + 4u - (unsigned)(n & 3)) & 3u) : (unsigned)((*index + (unsigned)(n & 3)) & 3u);
  return FMR_OK;
}

int fmr_process(fmr_chain *c, const float *iq, size_t n, double *audio, size_t audio_cap, size_t *n_audio) {
  if (!c || !n_audio) return FMR_ERR_BAD_ARG;
  *n_audio = 0;
  if (n == 0) return FMR_OK;             // FmDecode.cpp:89-92
  uint32_t bl = (uint32_t)n, al = 0;
  const int rc = fmr_process_blocks(c, iq, n, &bl, 1, audio, audio_cap, &al);
  if (rc) return rc;
  *n_audio = al;
  return FMR_OK;
}

int fmr_resample(fmr_chain *c, const float *iq, size_t n, float *out_iq, size_t out_cap, size_t *n_out) {
  if (!c || !n_out || !c->has_rs) return FMR_ERR_BAD_ARG;
  *n_out = 0;
  if (c->has_dec) { set_err("fmr_resample needs a chain created with mode -1 (front end only)"); return FMR_ERR_BAD_ARG; }
  if (n == 0) return FMR_OK;
  if (n > c->max_in) return FMR_ERR_CAPACITY;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipMemcpyAsync(c->d_in.p, iq, (size_t)c->in_bps * n, hipMemcpyHostToDevice, c->stream));
  uint32_t bl = (uint32_t)n;
  const int rc = c->run(c->d_in.p, c->max_in, &bl, 1, nullptr, 0, nullptr);
  if (rc) return rc;
  if ((size_t)c->last_n_if > out_cap) { set_err("out_cap too small"); return FMR_ERR_CAPACITY; }
  if (c->last_n_if)
    HIPCHK(hipMemcpyAsync(out_iq, c->d_if.p + c->H_if, sizeof(float2) * c->last_n_if, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_out = (size_t)c->last_n_if;
  return FMR_OK;
}

static int fetch_state(fmr_chain *c) {
  HIPCHK(hipSetDevice(c->cfg.device));
  if (int rc = c->sync_all()) return rc;
  HIPCHK(hipMemcpy(c->h_state.data(), c->d_state.p, sizeof(StreamState) * c->S, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(c->h_flags.data(), c->d_flags.p, sizeof(IterFlags) * c->S, hipMemcpyDeviceToHost));
  return FMR_OK;
}

int fmr_get_status_sized(fmr_chain *c, int stream, void *st, size_t st_size) {
  fmr_status full;
  const int rc = fmr_get_status(c, stream, &full);
  if (rc == FMR_OK && st) memcpy(st, &full, st_size < sizeof full ? st_size : sizeof full);      // never past the caller's struct
  return st ? rc : FMR_ERR_BAD_ARG;
}

int fmr_get_status(fmr_chain *c, int stream, fmr_status *st) {
  if (!c || !st || stream < 0 || stream >= c->S) return FMR_ERR_BAD_ARG;
  const int rc = fetch_state(c);
  if (rc) return rc;
  const StreamState &s = c->h_state[stream];
  st->if_rms = s.if_rms;
  st->baseband_mean = s.baseband_mean;
  st->baseband_level = s.baseband_level;
  st->pilot_level = 2 * s.pll_level;
  st->stereo_detected = s.stereo_detected;
  st->if_agc_gain = s.agc_gain;
  st->af_agc_gain = s.af_gain;
  st->multipath_error = s.mpf_error;
  st->pll_freq_err = s.pll_freq_err;
  st->multipath_resets = s.mpf_resets;
  st->agc_sync_timeouts = s.agc_sync_timeouts;
  const IterFlags &f = c->h_flags[stream];
  st->agc_iterations = f.agc_iters;
  st->pll_iterations = f.pll_iters;
  st->agc_fallback = f.agc_fallback;
  st->pll_fallback = f.pll_fallback;
  st->pll_residual = f.pll_resid;
  for (int i = 0; i < 16; i++) { st->agc_residual_history[i] = f.agc_hist[i]; st->pll_residual_history[i] = f.pll_hist[i]; }
  for (int i = 0; i < 8; i++) st->pll_residual_components[i] = f.pll_comp[i];
  for (int i = 0; i < 16; i++) st->pll_mismatch_history[i] = f.pll_rhist[i];
  st->pll_mismatch_accepted = f.pll_r_accepted;
  st->af_agc_fallback = f.af_fallback;
  return FMR_OK;
}

int fmr_get_pps_events(fmr_chain *c, int stream, fmr_pps_event *ev, int cap) {
  if (!c || stream < 0 || stream >= c->S || cap < 0 || (cap > 0 && !ev)) return FMR_ERR_BAD_ARG;
  const int rc = fetch_state(c);
  if (rc) return rc;
  const StreamState &s = c->h_state[stream];
  for (int i = 0; i < s.n_pps && i < cap; i++) {
    ev[i].pps_index = s.pps[i].pps_index;
    ev[i].sample_index = s.pps[i].sample_index;
    ev[i].block_position = s.pps[i].block_position;
    ev[i].block = s.pps[i].block + (uint32_t)c->pps_block_base;
    ev[i].stream = (uint32_t)stream;
  }
  return s.n_pps;
}

int fmr_get_multipath_coefficients(fmr_chain *c, int stream, float *coeff, int cap) {
  if (!c || stream < 0 || stream >= c->S || c->mpf_N == 0) return FMR_ERR_BAD_ARG;
  if (cap < 2 * c->mpf_N) return FMR_ERR_CAPACITY;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (int rc = c->sync_all()) return rc;
  HIPCHK(hipMemcpy(coeff, c->d_mpf_coeff.p + (size_t)stream * c->mpf_N, sizeof(float2) * c->mpf_N, hipMemcpyDeviceToHost));
  return c->mpf_N;
}

long long fmr_debug_read(fmr_chain *c, int stream, int which, void *out, size_t cap_bytes) {
  if (!c || stream < 0 || stream >= c->S) return FMR_ERR_BAD_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (int rc = c->sync_all()) return rc;
  long long n = c->last_n_if;
  const void *src = nullptr;
  size_t esz = 0;
  if (which == 5) {       // FMR_FE_STAMPS=1: {start, end [10 ns units of the constant clock], hardware id} per workgroup of the last fused launch
    if (!c->d_fe_stamps.p || stream != 0) return FMR_ERR_BAD_ARG;
    constexpr long long R = 2 * 16 * fmr_chain::kStampCalls;
    n = 3ll * c->fe_stamps_n;        // ... followed by the two rings of stream stamps (32 calls x 16 ids: constant clock, shader cycles) and the sequence number of the last call
    if ((size_t)(n + R + 1) * 8 > cap_bytes) return FMR_ERR_CAPACITY;
    if (n) HIPCHK(hipMemcpy(out, c->d_fe_stamps.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy((char *)out + (size_t)n * 8, c->d_fe_stamps.p + 3 * (size_t)c->kMaxFusedWg * c->S, R * 8, hipMemcpyDeviceToHost));
    reinterpret_cast<unsigned long long *>(out)[n + R] = c->call_seq;
    return n + R + 1;
  }
  switch (which) {
  case 0: src = (c->last_if && c->if_valid) ? c->last_if + (size_t)stream * (c->H_if + c->max_if) + c->H_if : nullptr; esz = sizeof(float2); break;
  case 1: src = (c->d_dec.p && c->dec_valid) ? c->d_dec.p + (size_t)stream * c->max_if : nullptr; esz = sizeof(float); break;
  case 2: src = c->d_raw_de.p ? c->d_raw_de.p + (size_t)stream * (c->H_a + c->max_if) + c->H_a : nullptr; esz = sizeof(double); break;
  case 3: src = c->d_base_de.p ? c->d_base_de.p + (size_t)stream * (c->H_a + c->max_if) + c->H_a : nullptr; esz = sizeof(double); break;
  case 4: src = (c->d_gain.p && c->gain_valid) ? c->d_gain.p + (size_t)stream * c->max_if : nullptr; esz = sizeof(float); break;
  default: return FMR_ERR_BAD_ARG;
  }
  if (!src) return FMR_ERR_BAD_ARG;
  if ((size_t)n * esz > cap_bytes) return FMR_ERR_CAPACITY;
  if (n) HIPCHK(hipMemcpy(out, src, (size_t)n * esz, hipMemcpyDeviceToHost));
  return n;
}

// plain streaming read: what the box delivers to a kernel that only reads (bench.py: context for the roofline fraction)
__global__ __launch_bounds__(256) void k_probe_read(const float *__restrict__ p0, size_t n16, float *sink) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f *p = reinterpret_cast<const v4f *>(p0);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const v4f a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride),
              c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc += (a + b) + (c + d);
  }
  for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
  if (acc.x + acc.y + acc.z + acc.w == 1.2345678e-30f) *sink = acc.x;       // keeps the loads alive; never true in practice
}

int fmr_probe_read_bandwidth(int device, const void *d_buf, size_t bytes, int reps, double *gbytes_per_s) {
  if (!d_buf || !gbytes_per_s || bytes < (1u << 20) || reps < 1) return FMR_ERR_BAD_ARG;
  // everything this function takes is given back on every path out of it, the caller's current device included
  struct Scope {
    int prev_dev = -1;
    float *sink = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ~Scope() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      if (st) (void)hipStreamDestroy(st);
      if (sink) (void)hipFree(sink);
      if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    }
  } sc;
  HIPCHK(hipGetDevice(&sc.prev_dev));
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCHK(hipMalloc((void **)&sc.sink, sizeof(float)));
  HIPCHK(hipStreamCreateWithFlags(&sc.st, hipStreamNonBlocking));      // (the null stream would wait for every blocking stream)
  HIPCHK(hipEventCreate(&sc.a));
  HIPCHK(hipEventCreate(&sc.b));
  double best = 0.0;
  for (int r = 0; r < reps + 1; r++) {          // (the first pass warms up and is not counted)
    HIPCHK(hipEventRecord(sc.a, sc.st));
    hipLaunchKernelGGL(k_probe_read, dim3(n_cu * 8), dim3(256), 0, sc.st, (const float *)d_buf, bytes / 16, sc.sink);
    HIPCHK(hipEventRecord(sc.b, sc.st));
    HIPCHK(hipEventSynchronize(sc.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, sc.a, sc.b));
    if (r > 0 && ms > 0.f) best = std::max(best, (double)(bytes / 16 * 16) / (ms * 1e-3) / 1e9);
  }
  *gbytes_per_s = best;
  return FMR_OK;
}

int fmr_probe_shader_clock(int device, double *mhz) {
  if (!mhz) return FMR_ERR_BAD_ARG;
  struct Scope {
    int prev_dev = -1;
    unsigned long long *out = nullptr;
    hipStream_t st = nullptr;
    ~Scope() {
      if (st) (void)hipStreamDestroy(st);
      if (out) (void)hipHostFree(out);
      if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    }
  } sc;
  HIPCHK(hipGetDevice(&sc.prev_dev));
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipHostMalloc((void **)&sc.out, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  HIPCHK(hipStreamCreateWithFlags(&sc.st, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_probe_clock, dim3(1), dim3(64), 0, sc.st, sc.out, 2000);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(sc.st));
  *mhz = sc.out[1] ? (double)sc.out[0] / ((double)sc.out[1] * 0.01) : 0.0;     // cycles per microsecond
  return FMR_OK;
}

void fmr_enable_kernel_timing(fmr_chain *c, int enable) { if (c) c->timing = enable; }

int fmr_get_kernel_trace(fmr_chain *c, const char **names, int *streams, float *start_ms, float *end_ms, int cap) {
  if (!c) return FMR_ERR_BAD_ARG;
  if (int rc = c->sync_all()) return rc;
  int n = 0;
  for (size_t i = 0; i < c->trace.size(); i++) {
    float t0 = 0.f, t1 = 0.f;
    (void)hipEventElapsedTime(&t0, c->trace_base, c->trace[i].a);
    (void)hipEventElapsedTime(&t1, c->trace_base, c->trace[i].b);
    if (n < cap) { names[n] = c->trace[i].name; streams[n] = c->trace_stream[i]; start_ms[n] = t0; end_ms[n] = t1; }
    n++;
    (void)hipEventDestroy(c->trace[i].a); (void)hipEventDestroy(c->trace[i].b);
  }
  c->trace.clear(); c->trace_stream.clear();
  if (c->trace_base) { (void)hipEventDestroy(c->trace_base); c->trace_base = nullptr; }
  return n;
}

int fmr_get_kernel_times(fmr_chain *c, const char **names, float *ms, int cap) {
  if (!c) return FMR_ERR_BAD_ARG;
  if (int rc = c->sync_all()) return rc;
  int n = 0;
  if (c->timing == 2 || c->timing == 4 || c->timing == 5) {   // dominant kernel only: one entry per call since the last query
    for (auto &k : c->dom_times) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, k.a, k.b);
      if (n < cap) { names[n] = k.name; ms[n] = t; }
      n++;
      (void)hipEventDestroy(k.a); (void)hipEventDestroy(k.b);
    }
    c->dom_times.clear();
    return n;
  }
  for (auto &k : c->ktimes) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, k.a, k.b);
    if (n < cap) { names[n] = k.name; ms[n] = t; }
    n++;
  }
  return n;
}

int fmr_filter_table(const char *name, const void **data, int *is_double) {
  struct Ent { const char *name; const void *p; int n; int dbl; };
  static const Ent tabs[] = {
      {"jj1bdx_48khz_fmaudio", k_jj1bdx_48khz_fmaudio, 127, 1},
      {"jj1bdx_48khz_nbfmaudio", k_jj1bdx_48khz_nbfmaudio, 63, 1},
      {"jj1bdx_am_48khz_narrow", k_jj1bdx_am_48khz_narrow, 255, 0},
      {"jj1bdx_am_48khz_medium", k_jj1bdx_am_48khz_medium, 255, 0},
      {"jj1bdx_am_48khz_default", k_jj1bdx_am_48khz_default, 255, 0},
      {"jj1bdx_am_48khz_wide", k_jj1bdx_am_48khz_wide, 127, 0},
      {"jj1bdx_nbfm_48khz_default", k_jj1bdx_nbfm_48khz_default, 127, 0},
      {"jj1bdx_nbfm_48khz_narrow", k_jj1bdx_nbfm_48khz_narrow, 127, 0},
      {"jj1bdx_nbfm_48khz_medium", k_jj1bdx_nbfm_48khz_medium, 127, 0},
      {"jj1bdx_nbfm_48khz_wide", k_jj1bdx_nbfm_48khz_wide, 127, 0},
      {"jj1bdx_fm_384kHz_narrow", k_jj1bdx_fm_384kHz_narrow, 127, 0},
      {"jj1bdx_fm_384kHz_medium", k_jj1bdx_fm_384kHz_medium, 127, 0},
      {"jj1bdx_cw_48khz_500hz", k_jj1bdx_cw_48khz_500hz, 2049, 0},
      {"jj1bdx_ssb_48khz_1500hz", k_jj1bdx_ssb_48khz_1500hz, 2049, 0},
  };
  if (!name) return FMR_ERR_BAD_ARG;
  for (const auto &e : tabs)
    if (std::strcmp(e.name, name) == 0) {
      if (data) *data = e.p;
      if (is_double) *is_double = e.dbl;
      return e.n;
    }
  return FMR_ERR_BAD_ARG;
}

}  // extern "C"
