/*
 * fmradion_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the airspy-fmradion FmDecoder / AmDecoder /
 * IfResampler hot path.  Only tests/, the smoke()
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may load
 * this library, and only as the checker.  The product (libfmradion_amd.so)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS
 *   - The reference cannot be compiled in this image without writing
 *     stand-ins for VOLK and r8brain-free-src headers (every reference
 *     translation unit includes <volk/volk_alloc.hh> through SoftFM.h), so no
 *     oracle/_ref build exists.
 *   - The reference holds no tests, golden vectors or IQ fixtures.
 *   - What pins this oracle: the known answers recorded from the compiled
 *     reference during the survey and the numeric data
 *     files of the reference (filter tables, fast_atan table, PLL constants),
 *     see tests/test_oracle_known_answers.py and tests/test_reference_pins.py.
 *   - PARITY UNPINNED for the two resamplers (r8brain-free-src 7.1 is absent):
 *     ora_rs_* is our own specification (DESIGN.md "Resampler specification"),
 *     accepted on specification tests, not sample-exact against r8brain.
 *   - VOLK primitives follow the loops the reference states as their
 *     equivalents in its own comments ("generic" semantics, sequential
 *     float accumulation).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference tree).
 */
#ifndef FMRADION_ORACLE_H
#define FMRADION_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- resampler specification (ours; stands in for r8brain) -------- */
typedef struct ora_resampler ora_resampler;
ora_resampler *ora_rs_create(double in_rate, double out_rate, double atten_db);
ora_resampler *ora_rs_create2(double in_rate, double out_rate, double atten_db, double pass_frac, int stop_nyquist);
void ora_rs_destroy(ora_resampler *rs);
/* returns number of outputs written (<= cap), or -1 if cap is too small */
int ora_rs_process(ora_resampler *rs, const double *in, int n, double *out,
                   int cap);
/* design introspection: which = 0:D 1:NA 2:LB 3:MB 4:TB 5:L 6:M */
long long ora_rs_info(const ora_resampler *rs, int which);
const double *ora_rs_taps_a(const ora_resampler *rs);
const double *ora_rs_taps_b(const ora_resampler *rs);

/* IfResampler (sfmbase/IfResampler.cpp:25-78): two real resamplers in
 * lock-step on Re and Im, output narrowed to float. */
typedef struct ora_ifr ora_ifr;
ora_ifr *ora_ifr_create(double in_rate, double out_rate);
ora_ifr *ora_ifr_create2(double in_rate, double out_rate, double atten_db, double pass_frac, int stop_nyquist);
void ora_ifr_destroy(ora_ifr *h);
int ora_ifr_process(ora_ifr *h, const float *iq, int n, float *out_iq, int cap);

/* ---------- DSP blocks --------------------------------------------------- */
/* LowPassFilterFirIQ (sfmbase/Filter.cpp:27-96) */
typedef struct ora_firiq ora_firiq;
ora_firiq *ora_firiq_create(const float *coeff, int ntaps, int downsample);
void ora_firiq_destroy(ora_firiq *f);
int ora_firiq_process(ora_firiq *f, const float *iq, int n, float *out_iq);

/* LowPassFilterFirAudio (sfmbase/Filter.cpp:101-163) */
typedef struct ora_firaudio ora_firaudio;
ora_firaudio *ora_firaudio_create(const double *coeff, int ntaps);
void ora_firaudio_destroy(ora_firaudio *f);
int ora_firaudio_process(ora_firaudio *f, const double *in, int n, double *out);

/* FirstOrderIirFilter / BiquadIirFilter (sfmbase/Filter.cpp:167-250) */
typedef struct { double b0, b1, a1, x1; } ora_iir1;
typedef struct { double b0, b1, b2, a1, a2, x1, x2; } ora_biquad;
void ora_iir1_init(ora_iir1 *f, double b0, double b1, double a1);
double ora_iir1_step(ora_iir1 *f, double x);
void ora_biquad_init(ora_biquad *f, double b0, double b1, double b2, double a1,
                     double a2);
double ora_biquad_step(ora_biquad *f, double x);
/* LowPassFilterRC ctor (sfmbase/Filter.cpp:186-188) */
void ora_lowpass_rc_init(ora_iir1 *f, double timeconst);
/* HighPassFilterIir ctor (sfmbase/Filter.cpp:254-290) */
void ora_highpass_init(ora_biquad *f, double cutoff);

/* Utility.h:118-152 */
float ora_rms_level(const float *iq, int n);
void ora_mean_rms(const float *x, int n, float *mean, float *rms);
/* Utility.h:236-304 */
float ora_fast_atan2f(float y, float x);
const float *ora_fast_atan_table(void); /* 257 entries */

/* IfSimpleAgc (sfmbase/IfSimpleAgc.cpp:26-57) */
typedef struct { float initial_gain, current_gain, max_gain, rate; } ora_ifagc;
void ora_ifagc_init(ora_ifagc *a, float initial, float max_gain, float rate);
void ora_ifagc_process(ora_ifagc *a, const float *iq, int n, float *out_iq);

/* AfSimpleAgc (sfmbase/AfSimpleAgc.cpp:26-56) */
typedef struct { double initial_gain, current_gain, max_gain, reference, rate; } ora_afagc;
void ora_afagc_init(ora_afagc *a, double initial, double max_gain,
                    double reference, double rate);
void ora_afagc_process(ora_afagc *a, const double *in, int n, double *out);

/* PhaseDiscriminator (sfmbase/PhaseDiscriminator.cpp:27-46) */
typedef struct { float normalize_factor, boundary, save_value; } ora_disc;
void ora_disc_init(ora_disc *d, double max_freq_dev);
void ora_disc_process(ora_disc *d, const float *iq, int n, float *out);

/* PilotPhaseLock (sfmbase/PilotPhaseLock.cpp:35-171) */
typedef struct {
  uint64_t pps_index;
  uint64_t sample_index;
  double block_position;
} ora_pps_event;
typedef struct ora_pll ora_pll;
ora_pll *ora_pll_create(double freq);
void ora_pll_destroy(ora_pll *p);
void ora_pll_process(ora_pll *p, const double *in, int n, double *out,
                     int pilot_shift);
void ora_pll_get_state(const ora_pll *p, double *s7);
void ora_pll_set_state(ora_pll *p, const double *s7);
int ora_pll_locked(const ora_pll *p);
double ora_pll_pilot_level(const ora_pll *p); /* = 2*m_pilot_level */
double ora_pll_freq_err(const ora_pll *p);
double ora_pll_phase(const ora_pll *p);
double ora_pll_freq(const ora_pll *p);
int ora_pll_pps_events(const ora_pll *p, ora_pps_event *ev, int cap);

/* MultipathFilter (sfmbase/MultipathFilter.cpp:39-197) */
typedef struct ora_mpf ora_mpf;
ora_mpf *ora_mpf_create(unsigned stages);
void ora_mpf_destroy(ora_mpf *m);
void ora_mpf_initialize_coefficients(ora_mpf *m);
/* returns 1 on success, 0 when a non-finite value was met */
int ora_mpf_process(ora_mpf *m, const float *iq, int n, float *out_iq);
double ora_mpf_error(const ora_mpf *m);
int ora_mpf_order(const ora_mpf *m);
const float *ora_mpf_coeff(const ora_mpf *m); /* interleaved re,im */

/* FineTuner (sfmbase/FineTuner.cpp:25-73) */
typedef struct { unsigned index, size; float *tab; } ora_finetuner;
void ora_finetuner_init(ora_finetuner *ft, unsigned table_size, int freq_shift);
void ora_finetuner_free(ora_finetuner *ft);
void ora_finetuner_process(ora_finetuner *ft, const float *iq, int n, float *out);

/* FourthConverterIQ (include/FourthConverterIQ.h:30-82) */
typedef struct { unsigned index; unsigned t0, t1, t2, t3; } ora_fourth;
void ora_fourth_init(ora_fourth *f, int up);
void ora_fourth_process(ora_fourth *f, const float *iq, int n, float *out_iq);

/* ---------- decoders ------------------------------------------------------ */
/* FmDecoder (sfmbase/FmDecode.cpp:25-283).  pilotcut = the
 * jj1bdx_48khz_fmaudio table (FilterParameters.cpp:26), passed in as data. */
typedef struct ora_fm ora_fm;
ora_fm *ora_fm_create(int fmfilter_enable, const float *fmfilter_coeff,
                      int n_fmfilter_coeff, int stereo, double deemphasis_us,
                      int pilot_shift, unsigned multipath_stages,
                      const double *pilotcut, int n_pilotcut);
void ora_fm_destroy(ora_fm *fm);
/* returns number of doubles written to audio (0 legal), -1 if cap too small */
int ora_fm_process(ora_fm *fm, const float *iq, int n, double *audio, int cap);
int ora_fm_stereo_detected(const ora_fm *fm);
float ora_fm_tuning_offset(const ora_fm *fm);
float ora_fm_baseband_level(const ora_fm *fm);
double ora_fm_pilot_level(const ora_fm *fm);
float ora_fm_if_rms(const ora_fm *fm);
double ora_fm_multipath_error(const ora_fm *fm);
float ora_fm_if_agc_gain(const ora_fm *fm);
int ora_fm_pps_events(const ora_fm *fm, ora_pps_event *ev, int cap);
const float *ora_fm_multipath_coeff(const ora_fm *fm, int *order);
/* debug taps for stage-level parity: last block's intermediate vectors.
 * which: 0 = discriminator output (float, n), 1 = rawstereo after demod and
 * de-emphasis (double), 2 = mono after de-emphasis (double) */
int ora_fm_debug_vector(const ora_fm *fm, int which, double *out, int cap);

/* AmDecoder, modes AM and DSB (sfmbase/AmDecode.cpp:25-234) */
enum { ORA_MODE_AM = 2, ORA_MODE_DSB = 3, ORA_MODE_USB = 4, ORA_MODE_LSB = 5, ORA_MODE_CW = 6, ORA_MODE_WSPR = 7 }; /* ModType order, SoftFM.h:49 */
typedef struct ora_am ora_am;
ora_am *ora_am_create(const float *amfilter_coeff, int n_coeff, int mode);
/* all modes: cwcoeff / ssbcoeff = the jj1bdx_cw_48khz_500hz / jj1bdx_ssb_48khz_1500hz tables (AmDecode.cpp:36,40) */
ora_am *ora_am_create2(const float *amfilter_coeff, int n_coeff, int mode, const float *cwcoeff, int n_cw,
                       const float *ssbcoeff, int n_ssb);
void ora_am_destroy(ora_am *am);
int ora_am_process(ora_am *am, const float *iq, int n, double *audio, int cap);
double ora_am_baseband_level(const ora_am *am);
float ora_am_af_agc_gain(const ora_am *am);
float ora_am_if_agc_gain(const ora_am *am);
float ora_am_if_rms(const ora_am *am);

/* NbfmDecoder (sfmbase/NbfmDecode.cpp:24-96).  audiocoeff = the jj1bdx_48khz_nbfmaudio table. */
typedef struct ora_nbfm ora_nbfm;
ora_nbfm *ora_nbfm_create(const float *nbfmfilter_coeff, int n_coeff, double freq_dev, const double *audiocoeff,
                          int n_audio);
void ora_nbfm_destroy(ora_nbfm *nb);
int ora_nbfm_process(ora_nbfm *nb, const float *iq, int n, double *audio, int cap);
float ora_nbfm_tuning_offset(const ora_nbfm *nb);
float ora_nbfm_baseband_level(const ora_nbfm *nb);
float ora_nbfm_if_rms(const ora_nbfm *nb);
float ora_nbfm_if_agc_gain(const ora_nbfm *nb);

/* Source sample formats (FileSource.cpp:120-128,491-531 via sf_read_float; RtlSdrSource.cpp:359-365):
 * fmt 0 cf32, 1 s16, 2 u8 offset binary, 3 s8; n IQ samples, out_iq = 2n floats.  Returns 0, -1 for unknown fmt. */
int ora_iq_convert(int fmt, const void *raw, int n, float *out_iq);

#ifdef __cplusplus
}
#endif
#endif
