"""ctypes binding of the CPU oracle (oracle/libfmoracle.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_LIB = None

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)


class PpsEvent(C.Structure):
    _fields_ = [("pps_index", C.c_uint64), ("sample_index", C.c_uint64), ("block_position", C.c_double)]


def build():
    subprocess.run(["make", "-C", _ODIR, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_ODIR, "libfmoracle.so")
    src = os.path.join(_ODIR, "fmradion_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        build()
    L = C.CDLL(so)
    vp = C.c_void_p
    sig = {
        "ora_rs_create": (vp, [C.c_double, C.c_double, C.c_double]),
        "ora_rs_create2": (vp, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
        "ora_ifr_create2": (vp, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
        "ora_rs_destroy": (None, [vp]),
        "ora_rs_process": (C.c_int, [vp, c_double_p, C.c_int, c_double_p, C.c_int]),
        "ora_rs_info": (C.c_longlong, [vp, C.c_int]),
        "ora_rs_taps_a": (c_double_p, [vp]),
        "ora_rs_taps_b": (c_double_p, [vp]),
        "ora_ifr_create": (vp, [C.c_double, C.c_double]),
        "ora_ifr_destroy": (None, [vp]),
        "ora_ifr_process": (C.c_int, [vp, c_float_p, C.c_int, c_float_p, C.c_int]),
        "ora_firiq_create": (vp, [c_float_p, C.c_int, C.c_int]),
        "ora_firiq_destroy": (None, [vp]),
        "ora_firiq_process": (C.c_int, [vp, c_float_p, C.c_int, c_float_p]),
        "ora_firaudio_create": (vp, [c_double_p, C.c_int]),
        "ora_firaudio_destroy": (None, [vp]),
        "ora_firaudio_process": (C.c_int, [vp, c_double_p, C.c_int, c_double_p]),
        "ora_rms_level": (C.c_float, [c_float_p, C.c_int]),
        "ora_mean_rms": (None, [c_float_p, C.c_int, c_float_p, c_float_p]),
        "ora_fast_atan2f": (C.c_float, [C.c_float, C.c_float]),
        "ora_fast_atan_table": (c_float_p, []),
        "ora_pll_create": (vp, [C.c_double]),
        "ora_pll_destroy": (None, [vp]),
        "ora_pll_process": (None, [vp, c_double_p, C.c_int, c_double_p, C.c_int]),
        "ora_pll_locked": (C.c_int, [vp]),
        "ora_pll_pilot_level": (C.c_double, [vp]),
        "ora_pll_freq_err": (C.c_double, [vp]),
        "ora_pll_phase": (C.c_double, [vp]),
        "ora_pll_freq": (C.c_double, [vp]),
        "ora_pll_pps_events": (C.c_int, [vp, C.POINTER(PpsEvent), C.c_int]),
        "ora_mpf_create": (vp, [C.c_uint]),
        "ora_mpf_destroy": (None, [vp]),
        "ora_mpf_initialize_coefficients": (None, [vp]),
        "ora_mpf_process": (C.c_int, [vp, c_float_p, C.c_int, c_float_p]),
        "ora_mpf_error": (C.c_double, [vp]),
        "ora_mpf_order": (C.c_int, [vp]),
        "ora_mpf_coeff": (c_float_p, [vp]),
        "ora_fm_create": (vp, [C.c_int, c_float_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint, c_double_p, C.c_int]),
        "ora_fm_destroy": (None, [vp]),
        "ora_fm_process": (C.c_int, [vp, c_float_p, C.c_int, c_double_p, C.c_int]),
        "ora_fm_stereo_detected": (C.c_int, [vp]),
        "ora_fm_tuning_offset": (C.c_float, [vp]),
        "ora_fm_baseband_level": (C.c_float, [vp]),
        "ora_fm_pilot_level": (C.c_double, [vp]),
        "ora_fm_if_rms": (C.c_float, [vp]),
        "ora_fm_multipath_error": (C.c_double, [vp]),
        "ora_fm_if_agc_gain": (C.c_float, [vp]),
        "ora_fm_pps_events": (C.c_int, [vp, C.POINTER(PpsEvent), C.c_int]),
        "ora_fm_multipath_coeff": (c_float_p, [vp, C.POINTER(C.c_int)]),
        "ora_fm_debug_vector": (C.c_int, [vp, C.c_int, c_double_p, C.c_int]),
        "ora_am_create": (vp, [c_float_p, C.c_int, C.c_int]),
        "ora_am_create2": (vp, [c_float_p, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p, C.c_int]),
        "ora_am_destroy": (None, [vp]),
        "ora_am_process": (C.c_int, [vp, c_float_p, C.c_int, c_double_p, C.c_int]),
        "ora_am_baseband_level": (C.c_double, [vp]),
        "ora_am_af_agc_gain": (C.c_float, [vp]),
        "ora_am_if_agc_gain": (C.c_float, [vp]),
        "ora_am_if_rms": (C.c_float, [vp]),
        "ora_iq_convert": (C.c_int, [C.c_int, vp, C.c_int, c_float_p]),
        "ora_nbfm_create": (vp, [c_float_p, C.c_int, C.c_double, c_double_p, C.c_int]),
        "ora_nbfm_destroy": (None, [vp]),
        "ora_nbfm_process": (C.c_int, [vp, c_float_p, C.c_int, c_double_p, C.c_int]),
        "ora_nbfm_tuning_offset": (C.c_float, [vp]),
        "ora_nbfm_baseband_level": (C.c_float, [vp]),
        "ora_nbfm_if_rms": (C.c_float, [vp]),
        "ora_nbfm_if_agc_gain": (C.c_float, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def as_iq32(x):
    """complex array -> contiguous complex64 (viewable as interleaved float32)."""
    return np.ascontiguousarray(x, dtype=np.complex64)


# ---- stateless helpers -------------------------------------------------------
class IfAgc(C.Structure):
    _fields_ = [("initial_gain", C.c_float), ("current_gain", C.c_float), ("max_gain", C.c_float), ("rate", C.c_float)]


class AfAgc(C.Structure):
    _fields_ = [("initial_gain", C.c_double), ("current_gain", C.c_double), ("max_gain", C.c_double),
                ("reference", C.c_double), ("rate", C.c_double)]


class Disc(C.Structure):
    _fields_ = [("normalize_factor", C.c_float), ("boundary", C.c_float), ("save_value", C.c_float)]


class Iir1(C.Structure):
    _fields_ = [("b0", C.c_double), ("b1", C.c_double), ("a1", C.c_double), ("x1", C.c_double)]


class Biquad(C.Structure):
    _fields_ = [("b0", C.c_double), ("b1", C.c_double), ("b2", C.c_double), ("a1", C.c_double),
                ("a2", C.c_double), ("x1", C.c_double), ("x2", C.c_double)]


class Fourth(C.Structure):
    _fields_ = [("index", C.c_uint), ("t0", C.c_uint), ("t1", C.c_uint), ("t2", C.c_uint), ("t3", C.c_uint)]


def _raw(name, res, args):
    fn = getattr(lib(), name)
    fn.restype = res
    fn.argtypes = args
    return fn


class IfSimpleAgc:
    def __init__(self, initial, max_gain, rate):
        self.s = IfAgc()
        _raw("ora_ifagc_init", None, [C.POINTER(IfAgc), C.c_float, C.c_float, C.c_float])(C.byref(self.s), initial, max_gain, rate)

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty_like(iq)
        _raw("ora_ifagc_process", None, [C.POINTER(IfAgc), c_float_p, C.c_int, c_float_p])(C.byref(self.s), _fp(iq), len(iq), _fp(out))
        return out

    @property
    def gain(self):
        return self.s.current_gain


class AfSimpleAgc:
    def __init__(self, initial, max_gain, reference, rate):
        self.s = AfAgc()
        _raw("ora_afagc_init", None, [C.POINTER(AfAgc)] + [C.c_double] * 4)(C.byref(self.s), initial, max_gain, reference, rate)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        _raw("ora_afagc_process", None, [C.POINTER(AfAgc), c_double_p, C.c_int, c_double_p])(C.byref(self.s), _dp(x), len(x), _dp(out))
        return out

    @property
    def gain(self):
        return self.s.current_gain


class PhaseDiscriminator:
    def __init__(self, max_freq_dev):
        self.s = Disc()
        _raw("ora_disc_init", None, [C.POINTER(Disc), C.c_double])(C.byref(self.s), max_freq_dev)

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(len(iq), dtype=np.float32)
        _raw("ora_disc_process", None, [C.POINTER(Disc), c_float_p, C.c_int, c_float_p])(C.byref(self.s), _fp(iq), len(iq), _fp(out))
        return out


class LowPassFilterRC:
    def __init__(self, timeconst):
        self.s = Iir1()
        _raw("ora_lowpass_rc_init", None, [C.POINTER(Iir1), C.c_double])(C.byref(self.s), timeconst)
        self._step = _raw("ora_iir1_step", C.c_double, [C.POINTER(Iir1), C.c_double])

    def process(self, x):
        return np.array([self._step(C.byref(self.s), float(v)) for v in x])


class HighPassFilterIir:
    def __init__(self, cutoff):
        self.s = Biquad()
        _raw("ora_highpass_init", None, [C.POINTER(Biquad), C.c_double])(C.byref(self.s), cutoff)
        self._step = _raw("ora_biquad_step", C.c_double, [C.POINTER(Biquad), C.c_double])

    def process(self, x):
        return np.array([self._step(C.byref(self.s), float(v)) for v in x])


class FourthConverterIQ:
    def __init__(self, up=False):
        self.s = Fourth()
        _raw("ora_fourth_init", None, [C.POINTER(Fourth), C.c_int])(C.byref(self.s), int(up))

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty_like(iq)
        _raw("ora_fourth_process", None, [C.POINTER(Fourth), c_float_p, C.c_int, c_float_p])(C.byref(self.s), _fp(iq), len(iq), _fp(out))
        return out


def rms_level(iq):
    iq = as_iq32(iq)
    return lib().ora_rms_level(_fp(iq), len(iq))


def mean_rms(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    m, r = C.c_float(), C.c_float()
    lib().ora_mean_rms(_fp(x), len(x), C.byref(m), C.byref(r))
    return m.value, r.value


def fast_atan2f(y, x):
    return lib().ora_fast_atan2f(y, x)


def fast_atan_table():
    p = lib().ora_fast_atan_table()
    return np.ctypeslib.as_array(p, shape=(257,)).copy()


# ---- stateful objects ----------------------------------------------------------
class Resampler:
    def __init__(self, in_rate, out_rate, atten_db, pass_frac=None, stop_nyquist=False):
        """pass_frac=None: the product's specification (0.885 x Nyquist, stop band from out - f_pass)."""
        if pass_frac is None:
            self.h = lib().ora_rs_create(in_rate, out_rate, atten_db)
        else:
            self.h = lib().ora_rs_create2(in_rate, out_rate, atten_db, pass_frac, int(stop_nyquist))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_rs_destroy(self.h)
            self.h = None

    def info(self):
        names = ["D", "NA", "LB", "MB", "TB", "L", "M", "LT"]
        return {n: lib().ora_rs_info(self.h, i) for i, n in enumerate(names)}

    def taps_a(self):
        i = self.info()
        if i["NA"] == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(lib().ora_rs_taps_a(self.h), shape=(i["NA"],)).copy()

    def taps_b(self):
        i = self.info()
        rows = i["LT"] + 1 if i["LT"] else i["LB"]     # fractional-phase form: LT + 1 rows, interpolated
        return np.ctypeslib.as_array(lib().ora_rs_taps_b(self.h), shape=(rows, i["TB"])).copy()

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(len(x) + 16, dtype=np.float64)
        n = lib().ora_rs_process(self.h, _dp(x), len(x), _dp(out), len(out))
        assert n >= 0
        return out[:n].copy()


class IfResampler:
    def __init__(self, in_rate, out_rate, atten_db=None, pass_frac=0.98, stop_nyquist=True):
        """atten_db=None: the product's specification; otherwise another one (default: r8brain-class)."""
        if atten_db is None:
            self.h = lib().ora_ifr_create(in_rate, out_rate)
        else:
            self.h = lib().ora_ifr_create2(in_rate, out_rate, atten_db, pass_frac, int(stop_nyquist))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_ifr_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(len(iq) + 16, dtype=np.complex64)
        n = lib().ora_ifr_process(self.h, _fp(iq), len(iq), _fp(out), len(out))
        assert n >= 0
        return out[:n].copy()


class LowPassFilterFirIQ:
    def __init__(self, coeff, downsample=1):
        c = np.ascontiguousarray(coeff, dtype=np.float32)
        self.ds = downsample
        self.h = lib().ora_firiq_create(_fp(c), len(c), downsample)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_firiq_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(len(iq) + 1, dtype=np.complex64)
        n = lib().ora_firiq_process(self.h, _fp(iq), len(iq), _fp(out))
        return out[:n].copy()


class LowPassFilterFirAudio:
    def __init__(self, coeff):
        c = np.ascontiguousarray(coeff, dtype=np.float64)
        self.h = lib().ora_firaudio_create(_dp(c), len(c))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_firaudio_destroy(self.h)
            self.h = None

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(len(x) + 1, dtype=np.float64)
        n = lib().ora_firaudio_process(self.h, _dp(x), len(x), _dp(out))
        return out[:n].copy()


class PilotPhaseLock:
    def __init__(self, freq):
        self.h = lib().ora_pll_create(freq)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_pll_destroy(self.h)
            self.h = None

    def process(self, x, pilot_shift=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        lib().ora_pll_process(self.h, _dp(x), len(x), _dp(out), int(pilot_shift))
        return out

    def locked(self):
        return bool(lib().ora_pll_locked(self.h))

    def pilot_level(self):
        return lib().ora_pll_pilot_level(self.h)

    def freq_err(self):
        return lib().ora_pll_freq_err(self.h)

    def phase(self):
        return lib().ora_pll_phase(self.h)

    def freq(self):
        return lib().ora_pll_freq(self.h)

    def pps_events(self):
        ev = (PpsEvent * 16)()
        n = lib().ora_pll_pps_events(self.h, ev, 16)
        return [(e.pps_index, e.sample_index, e.block_position) for e in ev[:min(n, 16)]]


class MultipathFilter:
    def __init__(self, stages):
        self.h = lib().ora_mpf_create(stages)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_mpf_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty_like(iq)
        ok = lib().ora_mpf_process(self.h, _fp(iq), len(iq), _fp(out))
        return bool(ok), out

    def initialize_coefficients(self):
        lib().ora_mpf_initialize_coefficients(self.h)

    def error(self):
        return lib().ora_mpf_error(self.h)

    def coeff(self):
        n = lib().ora_mpf_order(self.h)
        a = np.ctypeslib.as_array(lib().ora_mpf_coeff(self.h), shape=(2 * n,)).copy()
        return a.view(np.complex64)


class FmDecoder:
    """Mirror of FmDecoder (include/FmDecode.h:63-103)."""

    def __init__(self, fmfilter_enable, fmfilter_coeff, stereo, deemphasis, pilot_shift, multipath_stages, pilotcut):
        c = np.ascontiguousarray(fmfilter_coeff, dtype=np.float32)
        p = np.ascontiguousarray(pilotcut, dtype=np.float64)
        self.stereo = stereo
        self.h = lib().ora_fm_create(int(fmfilter_enable), _fp(c), len(c), int(stereo), float(deemphasis),
                                     int(pilot_shift), int(multipath_stages), _dp(p), len(p))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_fm_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(2 * (len(iq) // 4 + 32), dtype=np.float64)
        n = lib().ora_fm_process(self.h, _fp(iq), len(iq), _dp(out), len(out))
        assert n >= 0
        return out[:n].copy()

    def debug_vector(self, which, n):
        out = np.empty(n, dtype=np.float64)
        m = lib().ora_fm_debug_vector(self.h, which, _dp(out), n)
        return out[:m]

    def stereo_detected(self):
        return bool(lib().ora_fm_stereo_detected(self.h))

    def get_tuning_offset(self):
        return lib().ora_fm_tuning_offset(self.h)

    def get_baseband_level(self):
        return lib().ora_fm_baseband_level(self.h)

    def get_pilot_level(self):
        return lib().ora_fm_pilot_level(self.h)

    def get_if_rms(self):
        return lib().ora_fm_if_rms(self.h)

    def get_multipath_error(self):
        return lib().ora_fm_multipath_error(self.h)

    def get_if_agc_gain(self):
        return lib().ora_fm_if_agc_gain(self.h)

    def get_pps_events(self):
        ev = (PpsEvent * 16)()
        n = lib().ora_fm_pps_events(self.h, ev, 16)
        return [(e.pps_index, e.sample_index, e.block_position) for e in ev[:min(n, 16)]]

    def get_multipath_coefficients(self):
        n = C.c_int()
        p = lib().ora_fm_multipath_coeff(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(2 * n.value,)).copy().view(np.complex64)


MODE_AM, MODE_DSB, MODE_USB, MODE_LSB, MODE_CW, MODE_WSPR = 2, 3, 4, 5, 6, 7


class AmDecoder:
    """Mirror of AmDecoder (include/AmDecode.h:48-65), modes AM and DSB."""

    def __init__(self, amfilter_coeff, mode=MODE_AM, cw_coeff=None, ssb_coeff=None):
        c = np.ascontiguousarray(amfilter_coeff, dtype=np.float32)
        if cw_coeff is None and ssb_coeff is None:
            self.h = lib().ora_am_create(_fp(c), len(c), mode)
        else:   # USB / LSB / CW / WSPR need the 2049-tap tables of AmDecode.cpp:36,40
            cw = np.ascontiguousarray(cw_coeff, dtype=np.float32)
            ssb = np.ascontiguousarray(ssb_coeff, dtype=np.float32)
            self.h = lib().ora_am_create2(_fp(c), len(c), mode, _fp(cw), len(cw), _fp(ssb), len(ssb))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_am_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(len(iq) + 16, dtype=np.float64)
        n = lib().ora_am_process(self.h, _fp(iq), len(iq), _dp(out), len(out))
        assert n >= 0
        return out[:n].copy()

    def get_baseband_level(self):
        return lib().ora_am_baseband_level(self.h)

    def get_af_agc_current_gain(self):
        return lib().ora_am_af_agc_gain(self.h)

    def get_if_agc_current_gain(self):
        return lib().ora_am_if_agc_gain(self.h)

    def get_if_rms(self):
        return lib().ora_am_if_rms(self.h)


class NbfmDecoder:
    """Mirror of NbfmDecoder (include/NbfmDecode.h:49-66)."""

    freq_dev_normal = 8000.0
    freq_dev_wide = 17000.0

    def __init__(self, nbfmfilter_coeff, freq_dev, audio_coeff):
        c = np.ascontiguousarray(nbfmfilter_coeff, dtype=np.float32)
        a = np.ascontiguousarray(audio_coeff, dtype=np.float64)
        self.h = lib().ora_nbfm_create(_fp(c), len(c), float(freq_dev), _dp(a), len(a))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_nbfm_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = as_iq32(iq)
        out = np.empty(len(iq) + 16, dtype=np.float64)
        n = lib().ora_nbfm_process(self.h, _fp(iq), len(iq), _dp(out), len(out))
        assert n >= 0
        return out[:n].copy()

    def get_tuning_offset(self):
        return lib().ora_nbfm_tuning_offset(self.h)

    def get_baseband_level(self):
        return lib().ora_nbfm_baseband_level(self.h)

    def get_if_rms(self):
        return lib().ora_nbfm_if_rms(self.h)

    def get_if_agc_current_gain(self):
        return lib().ora_nbfm_if_agc_gain(self.h)


def iq_convert(fmt, raw):
    """raw: (N, 2) integer (or complex64 for fmt 0) -> complex64, the reference's source-side conversion."""
    raw = np.ascontiguousarray(raw)
    n = raw.shape[0]
    out = np.empty(n, dtype=np.complex64)
    rc = lib().ora_iq_convert(fmt, raw.ctypes.data_as(C.c_void_p), n, _fp(out.view(np.float32)))
    assert rc == 0
    return out

