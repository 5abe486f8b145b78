"""airspy-fmradion_amd -- MI355X-native FM/AM demodulation hot path.

Python-side binding of the C-ABI (include/fmradion_amd.h) used by the tests
and by bench.py.  The product is libfmradion_amd.so (hand-written HIP kernels,
csrc/); this module only loads it with ctypes.  There is no CPU fallback: if
the library is missing or no GPU is present, creation fails loudly.

The directory name carries a hyphen (the contract's package name), so import it
with importlib:  fmr = importlib.import_module("airspy-fmradion_amd").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libfmradion_amd.so")
# The same sources with -DFMR_AB_PARTNERS: the product plus the slower forms two GPU tests compare it with (the PLL's
# seven-launch Newton round, FMR_PLL_V1) and a test hook (FMR_TEST_AGC_LATE).  Loaded only by chains that are created while
# Chain(..., ab=True) -- the two tests that need them say so; the switches themselves are read from the environment by that
# library only.  The product library does not carry them, and a leftover variable in the environment selects nothing.
LIB_PATH_AB = os.path.join(_DIR, "libfmradion_amd_ab.so")
SRC = os.path.join(_DIR, "csrc", "fmradion_amd.hip")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

MODE_NONE, MODE_FM, MODE_NBFM, MODE_AM, MODE_DSB, MODE_USB, MODE_LSB, MODE_CW, MODE_WSPR = -1, 0, 1, 2, 3, 4, 5, 6, 7
IQ_CF32, IQ_S16, IQ_U8, IQ_S8 = 0, 1, 2, 3
RESAMPLER_FAST, RESAMPLER_R8B = 0, 1
_IQ_DTYPE = {0: np.complex64, 1: np.int16, 2: np.uint8, 3: np.int8}
OK, ERR_NO_DEVICE, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_HIP = 0, -1, -2, -3, -4, -5

EXPORTS = [
    "fmr_create", "fmr_destroy", "fmr_last_error", "fmr_version", "fmr_resampler_info", "fmr_process",
    "fmr_process_blocks", "fmr_process_blocks_device", "fmr_synchronize", "fmr_resample", "fmr_get_status",
    "fmr_get_pps_events", "fmr_get_multipath_coefficients", "fmr_debug_read", "fmr_get_kernel_times",
    "fmr_probe_read_bandwidth", "fmr_probe_shader_clock", "fmr_get_kernel_trace",
    "fmr_enable_kernel_timing", "fmr_filter_table", "fmr_fourth_convert", "fmr_design_taps", "fmr_design_taps_class",
    "fmr_host_alloc", "fmr_host_free", "fmr_create_sized", "fmr_get_status_sized",
]


class FmrError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("n_streams", C.c_int), ("mode", C.c_int), ("input_rate", C.c_double),
        ("enable_resampler", C.c_int), ("enable_fourth_down", C.c_int), ("fmfilter_enable", C.c_int),
        ("filter_coeff", C.POINTER(C.c_float)), ("n_filter_coeff", C.c_int), ("stereo", C.c_int),
        ("deemphasis_us", C.c_double), ("pilot_shift", C.c_int), ("multipath_stages", C.c_uint),
        ("max_block_len", C.c_size_t), ("max_blocks", C.c_int), ("nbfm_freq_dev", C.c_double),
        ("input_format", C.c_int), ("output_rate", C.c_double), ("resampler_class", C.c_int), ("struct_size", C.c_uint),
        ("in_order", C.c_int),
    ]


class Status(C.Structure):
    _fields_ = [
        ("if_rms", C.c_float), ("baseband_mean", C.c_float), ("baseband_level", C.c_float),
        ("pilot_level", C.c_double), ("stereo_detected", C.c_int), ("if_agc_gain", C.c_float),
        ("af_agc_gain", C.c_double), ("multipath_error", C.c_double), ("pll_freq_err", C.c_double),
        ("multipath_resets", C.c_uint32),
        ("agc_iterations", C.c_int), ("pll_iterations", C.c_int), ("agc_fallback", C.c_int),
        ("pll_fallback", C.c_int), ("pll_residual", C.c_double),
        ("agc_residual_history", C.c_float * 16), ("pll_residual_history", C.c_double * 16),
        ("pll_residual_components", C.c_double * 8),
        ("pll_mismatch_history", C.c_double * 16), ("pll_mismatch_accepted", C.c_int), ("af_agc_fallback", C.c_int),
        ("agc_sync_timeouts", C.c_uint32),
    ]


class PpsEvent(C.Structure):
    _fields_ = [("pps_index", C.c_uint64), ("sample_index", C.c_uint64), ("block_position", C.c_double),
                ("block", C.c_uint32), ("stream", C.c_uint32)]


def build_library(force=False, verbose=False):
    """Compile the HIP library (and its A/B partner build) in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import glob
    deps = sorted(glob.glob(os.path.join(_DIR, "csrc", "*")))      # every header of the translation unit
    deps.append(os.path.join(os.path.dirname(_DIR), "include", "fmradion_amd.h"))
    procs = []
    for path, extra in ((LIB_PATH, []), (LIB_PATH_AB, ["-DFMR_AB_PARTNERS"])):
        if not force and os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = ["hipcc"] + HIPCC_FLAGS + extra + ["-o", path, SRC]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return LIB_PATH


_libs = {}


def lib(ab=False):
    """The product library; ab=True: its A/B partner build (tests only)."""
    if ab in _libs:
        return _libs[ab]
    path = LIB_PATH_AB if ab else LIB_PATH
    if not os.path.exists(path):
        raise FmrError(f"{path} is missing: run __graft_entry__.build() (there is no CPU fallback)" if not ab else
                       f"{path} (the A/B partner build two GPU tests load) is missing: run __graft_entry__.build()")
    L = C.CDLL(path)
    vp, u32p, fp, dp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.fmr_create.restype = C.c_int
    L.fmr_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.fmr_destroy.restype = None
    L.fmr_destroy.argtypes = [vp]
    L.fmr_last_error.restype = C.c_char_p
    L.fmr_version.restype = C.c_char_p
    L.fmr_resampler_info.restype = C.c_longlong
    L.fmr_resampler_info.argtypes = [vp, C.c_int]
    L.fmr_process.restype = C.c_int
    L.fmr_process.argtypes = [vp, fp, C.c_size_t, dp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fmr_process_blocks.restype = C.c_int
    L.fmr_process_blocks.argtypes = [vp, fp, C.c_size_t, u32p, C.c_int, dp, C.c_size_t, u32p]
    L.fmr_process_blocks_device.restype = C.c_int
    L.fmr_process_blocks_device.argtypes = [vp, vp, C.c_size_t, u32p, C.c_int, vp, C.c_size_t, u32p, C.c_int]
    L.fmr_synchronize.restype = C.c_int
    L.fmr_synchronize.argtypes = [vp]
    L.fmr_resample.restype = C.c_int
    L.fmr_resample.argtypes = [vp, fp, C.c_size_t, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fmr_get_status.restype = C.c_int
    L.fmr_get_status.argtypes = [vp, C.c_int, C.POINTER(Status)]
    L.fmr_get_pps_events.restype = C.c_int
    L.fmr_get_pps_events.argtypes = [vp, C.c_int, C.POINTER(PpsEvent), C.c_int]
    L.fmr_get_multipath_coefficients.restype = C.c_int
    L.fmr_get_multipath_coefficients.argtypes = [vp, C.c_int, fp, C.c_int]
    L.fmr_debug_read.restype = C.c_longlong
    L.fmr_debug_read.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.fmr_get_kernel_times.restype = C.c_int
    L.fmr_get_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), fp, C.c_int]
    L.fmr_enable_kernel_timing.restype = None
    L.fmr_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.fmr_fourth_convert.restype = C.c_int
    L.fmr_fourth_convert.argtypes = [vp, fp, C.c_size_t, fp, C.c_int, C.POINTER(C.c_uint)]
    L.fmr_design_taps.restype = C.c_longlong
    L.fmr_design_taps.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, dp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.fmr_design_taps_class.restype = C.c_longlong
    L.fmr_design_taps_class.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, dp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.fmr_filter_table.restype = C.c_int
    L.fmr_filter_table.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int)]
    _libs[ab] = L
    return L


def filter_table(name):
    """FilterParameters::<name> (include/FilterParameters.h:31-49) as a numpy array."""
    p, dbl = C.c_void_p(), C.c_int()
    n = lib().fmr_filter_table(name.encode(), C.byref(p), C.byref(dbl))
    if n < 0:
        raise FmrError(f"unknown filter table {name}")
    ct = C.c_double if dbl.value else C.c_float
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).copy()


def design_taps(in_rate, out_rate, atten_db, stage):
    """The product's resampler design (host arithmetic of csrc/design.hpp; no GPU needed): (taps, info dict)."""
    info = (C.c_longlong * 6)()
    n = lib().fmr_design_taps(in_rate, out_rate, atten_db, stage, None, 0, info)
    if n < 0:
        raise FmrError(f"fmr_design_taps failed ({n}): {lib().fmr_last_error().decode()}")
    buf = np.empty(n, dtype=np.float64)
    lib().fmr_design_taps(in_rate, out_rate, atten_db, stage, buf.ctypes.data_as(C.POINTER(C.c_double)), n, info)
    d = dict(zip(["D", "NA", "LB", "MB", "TB", "LT"], [int(v) for v in info]))
    rows = d["LT"] + 1 if d["LT"] else d["LB"]          # fractional-phase form: LT + 1 rows, interpolated
    return (buf.reshape(rows, d["TB"]) if stage else buf), d


def design_taps_class(in_rate, out_rate, resampler_class, stage):
    """IfResampler(in_rate, out_rate) of a resampler class (RESAMPLER_FAST / RESAMPLER_R8B): (taps, info dict)."""
    info = (C.c_longlong * 6)()
    n = lib().fmr_design_taps_class(in_rate, out_rate, resampler_class, stage, None, 0, info)
    if n < 0:
        raise FmrError(f"fmr_design_taps_class failed ({n}): {lib().fmr_last_error().decode()}")
    buf = np.empty(n, dtype=np.float64)
    lib().fmr_design_taps_class(in_rate, out_rate, resampler_class, stage, buf.ctypes.data_as(C.POINTER(C.c_double)), n, info)
    d = dict(zip(["D", "NA", "LB", "MB", "TB", "LT"], [int(v) for v in info]))
    rows = d["LT"] + 1 if d["LT"] else d["LB"]
    return (buf.reshape(rows, d["TB"]) if stage else buf), d


def probe_shader_clock(device=0):
    """The shader clock of the device right now [MHz] (a 20 us count on one wave)."""
    out = C.c_double(0.0)
    L = lib()
    L.fmr_probe_shader_clock.restype = C.c_int
    L.fmr_probe_shader_clock.argtypes = [C.c_int, C.POINTER(C.c_double)]
    rc = L.fmr_probe_shader_clock(int(device), C.byref(out))
    if rc != 0:
        raise FmrError(f"fmr_probe_shader_clock failed ({rc}): {L.fmr_last_error().decode()}")
    return out.value


def probe_read_bandwidth(device, dev_ptr, nbytes, reps=5):
    """GB/s of a plain streaming-read kernel over device memory [dev_ptr, dev_ptr + nbytes): what this box delivers."""
    out = C.c_double(0.0)
    L = lib()
    L.fmr_probe_read_bandwidth.restype = C.c_int
    L.fmr_probe_read_bandwidth.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    rc = L.fmr_probe_read_bandwidth(int(device), C.c_void_p(int(dev_ptr)), int(nbytes), int(reps), C.byref(out))
    if rc != 0:
        raise FmrError(f"fmr_probe_read_bandwidth failed ({rc}): {L.fmr_last_error().decode()}")
    return out.value


DELAY_3TAPS = np.array([0.0, 1.0, 0.0], dtype=np.float32)  # FilterParameters::delay_3taps_only_iq


class Chain:
    """One decoder chain (fmr_chain) for n_streams independent IQ streams."""

    def __init__(self, mode=MODE_FM, input_rate=384000.0, enable_resampler=False, fourth_down=False,
                 fmfilter_enable=False, filter_coeff=None, stereo=True, deemphasis_us=50.0, pilot_shift=False,
                 multipath_stages=0, max_block_len=65536, max_blocks=1, n_streams=1, device=0, nbfm_freq_dev=0.0, input_format=0,
                 output_rate=0.0, resampler_class=RESAMPLER_FAST, in_order=False, ab=False):
        self._L = lib(ab=bool(ab))
        coeff = np.ascontiguousarray(DELAY_3TAPS if filter_coeff is None else filter_coeff, dtype=np.float32)
        self._coeff = coeff
        cfg = Config()
        cfg.device, cfg.n_streams, cfg.mode, cfg.input_rate = device, n_streams, mode, float(input_rate)
        cfg.enable_resampler, cfg.enable_fourth_down = int(enable_resampler), int(fourth_down)
        cfg.fmfilter_enable = int(fmfilter_enable)
        cfg.filter_coeff = coeff.ctypes.data_as(C.POINTER(C.c_float))
        cfg.n_filter_coeff = len(coeff)
        cfg.stereo, cfg.deemphasis_us, cfg.pilot_shift = int(stereo), float(deemphasis_us), int(pilot_shift)
        cfg.multipath_stages = int(multipath_stages)
        cfg.max_block_len, cfg.max_blocks = int(max_block_len), int(max_blocks)
        cfg.nbfm_freq_dev = float(nbfm_freq_dev)
        cfg.input_format = int(input_format)
        cfg.output_rate = float(output_rate)
        cfg.resampler_class = int(resampler_class)
        cfg.struct_size = C.sizeof(Config)
        cfg.in_order = int(bool(in_order))
        self.input_format = int(input_format)
        self.n_streams, self.mode, self.stereo = n_streams, mode, bool(stereo) and mode == MODE_FM
        self.h = C.c_void_p()
        rc = self._L.fmr_create(C.byref(cfg), C.byref(self.h))
        if rc != OK:
            self.h = None
            raise FmrError(f"fmr_create failed ({rc}): {self._L.fmr_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None) and getattr(self, "_L", None) is not None:
            self._L.fmr_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise FmrError(f"fmradion_amd error {rc}: {self._L.fmr_last_error().decode()}")
        return rc

    def resampler_info(self):
        return {k: self._L.fmr_resampler_info(self.h, i) for i, k in enumerate(["D", "NA", "LB", "MB", "TB", "LT"])}

    # --- host-buffer API ---------------------------------------------------------
    def process(self, iq):
        """FmDecoder::process / AmDecoder::process shape: one block in, audio doubles out."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(2 * (len(iq) + 64), dtype=np.float64)
        n = C.c_size_t()
        self._chk(self._L.fmr_process(self.h, iq.ctypes.data_as(C.POINTER(C.c_float)), len(iq),
                                    out.ctypes.data_as(C.POINTER(C.c_double)), len(out), C.byref(n)))
        return out[:n.value].copy()

    def process_blocks(self, iq, block_len):
        """iq: (n_streams, N) complex64; block_len: consecutive block lengths. Returns (audio[S][total], audio_len)."""
        if self.input_format == IQ_CF32:
            iq = np.ascontiguousarray(np.atleast_2d(iq), dtype=np.complex64)
        else:   # raw formats: (n_streams, N, 2) interleaved I,Q of the format's integer type
            iq = np.ascontiguousarray(iq, dtype=_IQ_DTYPE[self.input_format])
            assert iq.ndim == 3 and iq.shape[2] == 2
        assert iq.shape[0] == self.n_streams
        bl = np.ascontiguousarray(block_len, dtype=np.uint32)
        assert int(bl.sum()) <= iq.shape[1]
        acap = 2 * (int(bl.sum()) + 64 * len(bl))
        audio = np.zeros((self.n_streams, acap), dtype=np.float64)
        alen = np.zeros(len(bl), dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self._chk(self._L.fmr_process_blocks(self.h, iq.ctypes.data_as(C.POINTER(C.c_float)), iq.shape[1],
                                           bl.ctypes.data_as(u32p), len(bl),
                                           audio.ctypes.data_as(C.POINTER(C.c_double)), acap,
                                           alen.ctypes.data_as(u32p)))
        return audio[:, :int(alen.sum())].copy(), alen

    def fourth_convert(self, iq, index=0, up=False):
        """FourthConverterIQ::process on one block; returns (shifted block, new table index)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty_like(iq)
        idx = C.c_uint(index)
        self._chk(self._L.fmr_fourth_convert(self.h, iq.ctypes.data_as(C.POINTER(C.c_float)), len(iq),
                                           out.ctypes.data_as(C.POINTER(C.c_float)), int(up), C.byref(idx)))
        return out, idx.value

    def resample(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(len(iq) + 64, dtype=np.complex64)
        n = C.c_size_t()
        self._chk(self._L.fmr_resample(self.h, iq.ctypes.data_as(C.POINTER(C.c_float)), len(iq),
                                     out.ctypes.data_as(C.POINTER(C.c_float)), len(out), C.byref(n)))
        return out[:n.value].copy()

    # --- device-buffer API (pointers are raw device addresses, e.g. torch .data_ptr()) -------------
    def process_blocks_device(self, d_iq_ptr, stream_stride, block_len, d_audio_ptr, audio_stride, sync=False):
        bl = np.ascontiguousarray(block_len, dtype=np.uint32)
        alen = np.zeros(len(bl), dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self._chk(self._L.fmr_process_blocks_device(self.h, C.c_void_p(d_iq_ptr), stream_stride,
                                                  bl.ctypes.data_as(u32p), len(bl), C.c_void_p(d_audio_ptr),
                                                  audio_stride, alen.ctypes.data_as(u32p), int(sync)))
        return alen

    def synchronize(self):
        self._chk(self._L.fmr_synchronize(self.h))

    # --- getters -----------------------------------------------------------------------
    def status(self, stream=0):
        st = Status()
        self._chk(self._L.fmr_get_status(self.h, stream, C.byref(st)))
        return st

    def pps_events(self, stream=0):
        ev = (PpsEvent * 64)()
        n = self._chk(self._L.fmr_get_pps_events(self.h, stream, ev, 64))
        return [(e.pps_index, e.sample_index, e.block_position, e.block) for e in ev[:min(n, 64)]]

    def multipath_coefficients(self, stream=0):
        buf = np.empty(2 * 1300, dtype=np.float32)
        n = self._chk(self._L.fmr_get_multipath_coefficients(self.h, stream, buf.ctypes.data_as(C.POINTER(C.c_float)), len(buf)))
        return buf[:2 * n].view(np.complex64).copy()

    def debug_read(self, which, stream=0, cap=1 << 24):
        dt = {0: np.complex64, 1: np.float32, 2: np.float64, 3: np.float64, 4: np.float32, 5: np.uint64}[which]
        buf = np.empty(cap, dtype=dt)
        n = self._chk(self._L.fmr_debug_read(self.h, stream, which, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
        return buf[:n].copy()

    def enable_kernel_timing(self, mode=1):
        """0 off, 1 every kernel of the last call, 2 the dominant kernel only (accumulated over calls)."""
        self._L.fmr_enable_kernel_timing(self.h, int(mode))

    def kernel_trace(self, cap=1 << 16):
        """After enable_kernel_timing(3): [(name, stream, start_ms, end_ms)] of every instrumented kernel since then."""
        names = (C.c_char_p * cap)()
        st = (C.c_int * cap)()
        t0 = (C.c_float * cap)()
        t1 = (C.c_float * cap)()
        L = self._L
        L.fmr_get_kernel_trace.restype = C.c_int
        L.fmr_get_kernel_trace.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        n = self._chk(L.fmr_get_kernel_trace(self.h, names, st, t0, t1, cap))
        return [(names[i].decode(), st[i], t0[i], t1[i]) for i in range(min(n, cap))]

    def kernel_times(self, cap=4096):
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = self._chk(self._L.fmr_get_kernel_times(self.h, names, ms, cap))
        return [(names[i].decode(), ms[i]) for i in range(min(n, cap))]
