"""Loader of libwfahip.so (the HIP hot path).  There is no fallback: if the library is missing the
import of any product entry point fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwfahip.so")

# whole-call codes / per-pair status (include/wfa_hip.h)
OK, ERR_NO_DEVICE, ERR_BAD_ARG, ERR_OOM, ERR_HIP, ERR_UNSUPPORTED, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6
PAIR_OK, PAIR_EMPTY, PAIR_TOO_LONG, PAIR_NO_MEMORY = 0, 1, 2, 4
MAX_SEQ_LEN = (1 << 29) - 1
REC_WORDS = 16
(REC_STATUS, REC_SCORE, REC_TBEGIN, REC_TEND, REC_QBEGIN, REC_QEND, REC_ALIGN_LEN, REC_MATCHES, REC_GAPS,
 REC_GAP_REGIONS, REC_OPS_LEN, REC_OPS_OFF_LO, REC_OPS_OFF_HI, REC_CELLS_LO, REC_CELLS_HI, REC_N_SCORES) = range(16)

# every symbol include/wfa_hip.h declares
EXPORTS = [
    "wfahip_version", "wfahip_strerror", "wfahip_device_count", "wfahip_create", "wfahip_destroy",
    "wfahip_align_batch", "wfahip_results_free", "wfahip_align_batch_device", "wfahip_last_timing",
    "wfahip_set_option", "wfahip_debug_wavefronts", "wfahip_free", "wfahip_gen_stride",
    "wfahip_generate_pairs", "wfahip_packed_words", "wfahip_pack_pairs", "wfahip_align_batch_packed", "wfahip_submit",
    "wfahip_pending", "wfahip_collect", "wfahip_create_multi", "wfahip_destroy_multi", "wfahip_multi_size",
    "wfahip_multi_ctx", "wfahip_align_batch_multi", "wfahip_debug_compact_arena",
    "wfahip_generate_pairs_device", "wfahip_align_pair", "wfahip_last_error", "wfahip_debug_clock",
    "wfahip_debug_team_compact",
]


class Params(C.Structure):
    _fields_ = [("mismatch", C.c_uint32), ("gap_open", C.c_uint32), ("gap_ext", C.c_uint32),
                ("global_alignment", C.c_uint8), ("adaptive", C.c_uint8), ("reserved", C.c_uint8 * 2),
                ("min_wf_len", C.c_uint32), ("max_dist_diff", C.c_uint32), ("cutoff_step", C.c_uint32)]


class Results(C.Structure):
    _fields_ = [("n", C.c_uint64), ("status", C.POINTER(C.c_int32)), ("score", C.POINTER(C.c_uint32)),
                ("tbegin", C.POINTER(C.c_int32)), ("tend", C.POINTER(C.c_int32)),
                ("qbegin", C.POINTER(C.c_int32)), ("qend", C.POINTER(C.c_int32)),
                ("align_len", C.POINTER(C.c_uint32)), ("matches", C.POINTER(C.c_uint32)),
                ("gaps", C.POINTER(C.c_uint32)), ("gap_regions", C.POINTER(C.c_uint32)),
                ("ops_off", C.POINTER(C.c_uint64)), ("ops_len", C.POINTER(C.c_uint32)),
                ("ops", C.POINTER(C.c_uint64)), ("n_ops", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("n_launches", C.c_uint32),
                ("n_retried_pairs", C.c_uint32), ("cells_stored", C.c_uint64), ("ops_written", C.c_uint64),
                ("arena_bytes", C.c_uint64), ("main_kernel_ms", C.c_double), ("n_main_launches", C.c_uint32),
                ("n_packed_pairs", C.c_uint32), ("main_kernel_kind", C.c_uint32), ("ladder_start_level", C.c_uint32)]


class Row(C.Structure):
    _fields_ = [("score", C.c_uint32), ("lo", C.c_int32), ("width", C.c_uint32), ("word_off", C.c_uint64)]


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (make, or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build the HIP extension first "
                               "(`make` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                               "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.wfahip_version.restype = C.c_int
        L.wfahip_strerror.restype = C.c_char_p
        L.wfahip_strerror.argtypes = [C.c_int]
        L.wfahip_device_count.restype = C.c_int
        L.wfahip_create.restype = C.c_int
        L.wfahip_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.wfahip_destroy.argtypes = [vp]
        L.wfahip_align_batch.restype = C.c_int
        L.wfahip_align_batch.argtypes = [vp, C.POINTER(Params), vp, u64, vp, vp, vp, vp, u64, C.POINTER(Results)]
        L.wfahip_results_free.argtypes = [C.POINTER(Results)]
        L.wfahip_align_batch_device.restype = C.c_int
        L.wfahip_align_batch_device.argtypes = [vp, C.POINTER(Params), vp, u64, vp, vp, vp, vp, u64, u32, vp, vp,
                                                u64, C.POINTER(u64), vp]
        L.wfahip_last_timing.restype = C.c_int
        L.wfahip_last_timing.argtypes = [vp, C.POINTER(Timing)]
        L.wfahip_set_option.restype = C.c_int
        L.wfahip_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.wfahip_debug_wavefronts.restype = C.c_int
        L.wfahip_debug_wavefronts.argtypes = [vp, C.POINTER(Params), C.c_char_p, u32, C.c_char_p, u32,
                                              C.POINTER(C.POINTER(Row)), C.POINTER(u64),
                                              C.POINTER(C.POINTER(u32)), C.POINTER(u64), C.POINTER(Results)]
        L.wfahip_debug_team_compact.restype = C.c_int
        L.wfahip_debug_team_compact.argtypes = [vp, C.POINTER(Params), C.c_char_p, u32, C.c_char_p, u32,
                                              C.POINTER(C.POINTER(Row)), C.POINTER(u64),
                                              C.POINTER(C.POINTER(u32)), C.POINTER(u64), C.POINTER(Results)]
        L.wfahip_free.argtypes = [vp]
        L.wfahip_gen_stride.restype = u64
        L.wfahip_gen_stride.argtypes = [u32, C.c_double]
        L.wfahip_generate_pairs.restype = C.c_int
        L.wfahip_generate_pairs.argtypes = [u64, u64, u64, u32, C.c_double, C.c_int, vp, vp, vp, vp, vp]
        L.wfahip_packed_words.restype = u64
        L.wfahip_packed_words.argtypes = [u32]
        L.wfahip_pack_pairs.restype = C.c_int
        L.wfahip_pack_pairs.argtypes = [vp, vp, vp, vp, vp, u64, C.c_int, vp, vp, vp, C.POINTER(u64)]
        L.wfahip_align_batch_packed.restype = C.c_int
        L.wfahip_align_batch_packed.argtypes = [vp, C.POINTER(Params), vp, u64, vp, vp, vp, vp, u64, C.POINTER(Results)]
        L.wfahip_submit.restype = C.c_int
        L.wfahip_submit.argtypes = [vp, C.c_char_p, u32, C.c_char_p, u32, C.POINTER(u64)]
        L.wfahip_pending.restype = u64
        L.wfahip_debug_clock.restype = C.c_int
        L.wfahip_debug_clock.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.wfahip_align_pair.restype = C.c_int
        L.wfahip_align_pair.argtypes = [vp, C.POINTER(Params), C.c_char_p, u32, C.c_char_p, u32, vp, vp, u64, C.POINTER(u64)]
        L.wfahip_last_error.restype = C.c_char_p
        L.wfahip_last_error.argtypes = [vp]
        L.wfahip_pending.argtypes = [vp]
        L.wfahip_collect.restype = C.c_int
        L.wfahip_collect.argtypes = [vp, C.POINTER(Params), C.POINTER(Results)]
        L.wfahip_create_multi.restype = C.c_int
        L.wfahip_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.wfahip_destroy_multi.argtypes = [vp]
        L.wfahip_multi_size.restype = C.c_int
        L.wfahip_multi_size.argtypes = [vp]
        L.wfahip_multi_ctx.restype = vp
        L.wfahip_multi_ctx.argtypes = [vp, C.c_int]
        L.wfahip_align_batch_multi.restype = C.c_int
        L.wfahip_align_batch_multi.argtypes = [vp, C.POINTER(Params), vp, u64, vp, vp, vp, vp, u64, C.POINTER(Results)]
        L.wfahip_debug_compact_arena.restype = C.c_int
        L.wfahip_debug_compact_arena.argtypes = [vp, u64, C.POINTER(C.POINTER(u32)), C.POINTER(u64), C.POINTER(u32),
                                                 C.POINTER(u32 * 4)]
        L.wfahip_generate_pairs_device.restype = C.c_int
        L.wfahip_generate_pairs_device.argtypes = [vp, u64, u64, u64, u32, C.c_double, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


class WfaHipError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = code
        msg = lib().wfahip_strerror(code).decode()
        super().__init__(f"{what}: {msg} ({code})" if what else f"{msg} ({code})")


def check(code: int, what: str = ""):
    if code != OK:
        raise WfaHipError(code, what)
