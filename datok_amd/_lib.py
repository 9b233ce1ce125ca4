"""ctypes binding of libdatok_gpu.so (include/datok_gpu.h).

There is no fallback: if the HIP library is missing or no GPU is usable the
calls raise.  Nothing in this package imports the CPU oracle.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# DATOK_GPU_LIB selects another build of the same library (A/B timing on one box)
LIB_PATH = os.environ.get("DATOK_GPU_LIB") or os.path.join(_HERE, "libdatok_gpu.so")

# error codes / flags (datok_gpu.h)
OK, E_IO, E_FORMAT, E_NO_DEVICE, E_HIP, E_ARG, E_MODEL, E_CAPACITY, E_STATE, E_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9
ST_WINDOW_OVERFLOW, ST_EMPTY_TEXT, ST_BAD_MODEL, ST_IRREGULAR, ST_STEP_LIMIT, ST_INTERNAL = 1, 2, 4, 8, 16, 32
ST_BAD_OFFSET = 64

EXPORTS = [
    "dtk_device_count", "dtk_set_device", "dtk_strerror", "dtk_last_hip_error",
    "dtk_model_load", "dtk_model_load_mem", "dtk_model_free", "dtk_model_type", "dtk_model_get_info",
    "dtk_batch_create", "dtk_batch_free", "dtk_batch_set_input", "dtk_batch_set_input_device",
    "dtk_batch_run", "dtk_batch_sync", "dtk_batch_stream", "dtk_batch_totals",
    "dtk_batch_set_profiling", "dtk_batch_stage_ms", "dtk_batch_set_chunking", "dtk_batch_set_warm_extend",
    "dtk_batch_result_device", "dtk_batch_result_host", "dtk_transduce", "dtk_free",
    "dtk_foma_to_matok", "dtk_foma_to_datok", "dtk_transduce_replay", "dtk_batch_render_device", "dtk_batch_render_host",
    "dtk_batch_status_host", "dtk_transduce_release", "dtk_transduce_result",
    "dtk_pipeline_create", "dtk_pipeline_free", "dtk_pipeline_set_chunking", "dtk_pipeline_run",
    "dtk_pinned_alloc", "dtk_pinned_free",
    "dtk_batch_set_result_fields", "dtk_batch_download_begin", "dtk_pipeline_set_result_fields",
    "dtk_batch_set_download_stream", "dtk_batch_download_stream", "dtk_batch_done", "dtk_batch_set_streams",
    "dtk_debug_configure",
    "dtk_multi_create", "dtk_multi_free", "dtk_multi_type", "dtk_multi_set_result_fields", "dtk_multi_set_chunking", "dtk_multi_run",
]


class ModelInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("epsilon", C.c_int32), ("unknown", C.c_int32),
                ("identity", C.c_int32), ("final_state", C.c_int32), ("sigma_count", C.c_int32),
                ("state_count", C.c_uint32), ("array_len", C.c_uint64),
                ("n_eps_states", C.c_uint32), ("max_eps_chain", C.c_uint32),
                ("entry_bytes", C.c_uint32), ("device_bytes", C.c_uint64),
                ("unknown_used", C.c_uint32), ("dense_states", C.c_uint32),
                ("stream_codes", C.c_uint32)]


class Totals(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("n_bytes", C.c_uint64), ("n_tokens", C.c_uint64),
                ("n_sent", C.c_uint64), ("n_texts", C.c_uint64), ("n_flagged", C.c_uint64),
                ("walk_steps", C.c_uint64), ("n_lanes", C.c_uint32), ("chunk_bytes", C.c_uint32),
                ("repair_rounds", C.c_uint32)]


class ResultView(C.Structure):
    _fields_ = [("tok_off", C.c_void_p), ("sent_off", C.c_void_p), ("text_off", C.c_void_p),
                ("tok_rstart", C.c_void_p), ("tok_rend", C.c_void_p),
                ("tok_bstart", C.c_void_p), ("tok_bend", C.c_void_p),
                ("sent", C.c_void_p), ("text_tok_end", C.c_void_p), ("text_sent_end", C.c_void_p),
                ("status", C.c_void_p), ("ev_bits", C.c_void_p), ("ev_words", C.c_uint64), ("doc_tail", C.c_void_p),
                ("n_exact", C.c_uint32), ("exact_doc", C.c_void_p), ("exact_off", C.c_void_p),
                ("calls", C.c_void_p), ("tok_r16", C.c_void_p)]


SLICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)


class RenderView(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("doc_off", C.c_void_p), ("total", C.c_uint64)]


class DatokGpuError(RuntimeError):
    def __init__(self, code, what=""):
        L = lib()
        msg = L.dtk_strerror(code).decode()
        if code == E_HIP:
            msg += ": " + L.dtk_last_hip_error().decode()
        super().__init__("%s%s (code %d)" % (what + ": " if what else "", msg, code))
        self.code = code


def build(force=False):
    """Compile libdatok_gpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)
            if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("datok_gpu.h", "datok.hpp")]
    cli = os.path.join(_HERE, "datok")     # the command line (csrc/datok_cli.cpp), built by the same Makefile
    if (not force and os.path.exists(LIB_PATH) and os.path.exists(cli)
            and all(min(os.path.getmtime(LIB_PATH), os.path.getmtime(cli)) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", src_dir, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("datok_amd: %s is missing; run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    L.dtk_device_count.restype = C.c_int
    L.dtk_set_device.argtypes = [C.c_int]
    L.dtk_strerror.restype = C.c_char_p
    L.dtk_strerror.argtypes = [C.c_int]
    L.dtk_last_hip_error.restype = C.c_char_p
    L.dtk_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.dtk_model_load_mem.argtypes = [C.c_char_p, sz, C.POINTER(vp)]
    L.dtk_model_free.argtypes = [vp]
    L.dtk_model_free.restype = None
    L.dtk_model_type.restype = C.c_char_p
    L.dtk_model_type.argtypes = [vp]
    L.dtk_model_get_info.argtypes = [vp, C.POINTER(ModelInfo)]
    L.dtk_batch_create.argtypes = [u64, u32, C.POINTER(vp)]
    L.dtk_batch_free.argtypes = [vp]
    L.dtk_batch_free.restype = None
    L.dtk_batch_set_input.argtypes = [vp, vp, vp, u32]
    L.dtk_batch_set_input_device.argtypes = [vp, vp, vp, u32, u64]
    L.dtk_batch_run.argtypes = [vp, vp, u32]
    L.dtk_batch_sync.argtypes = [vp]
    L.dtk_batch_stream.restype = vp
    L.dtk_batch_stream.argtypes = [vp]
    L.dtk_batch_totals.argtypes = [vp, C.POINTER(Totals)]
    L.dtk_batch_set_chunking.argtypes = [vp, u32, u32]
    L.dtk_batch_set_warm_extend.argtypes = [vp, u32]
    L.dtk_batch_set_profiling.argtypes = [vp, C.c_int]
    L.dtk_batch_stage_ms.argtypes = [vp, C.POINTER(C.c_float * 9)]
    L.dtk_batch_result_device.argtypes = [vp, C.POINTER(ResultView)]
    L.dtk_batch_result_host.argtypes = [vp, C.POINTER(ResultView)]
    L.dtk_batch_set_result_fields.argtypes = [vp, u32]
    L.dtk_batch_download_begin.argtypes = [vp]
    L.dtk_batch_done.argtypes = [vp]
    L.dtk_batch_set_streams.argtypes = [vp, vp, vp]
    L.dtk_batch_set_download_stream.argtypes = [vp, vp]
    L.dtk_batch_download_stream.argtypes = [vp]
    L.dtk_batch_download_stream.restype = vp
    L.dtk_pipeline_set_result_fields.argtypes = [vp, u32]
    L.dtk_transduce.argtypes = [vp, C.c_char_p, sz, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(u32)]
    L.dtk_transduce_replay.argtypes = L.dtk_transduce.argtypes
    L.dtk_transduce_release.restype = None
    L.dtk_transduce_result.argtypes = [vp, C.c_char_p, sz, u32, C.POINTER(ResultView)]
    L.dtk_batch_render_device.argtypes = [vp, u32, C.POINTER(RenderView)]
    L.dtk_batch_render_host.argtypes = [vp, u32, C.POINTER(RenderView)]
    L.dtk_batch_status_host.argtypes = [vp, vp, u32]
    L.dtk_foma_to_matok.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.dtk_foma_to_matok.restype = C.c_int
    L.dtk_foma_to_datok.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.dtk_foma_to_datok.restype = C.c_int
    L.dtk_free.argtypes = [vp]
    L.dtk_free.restype = None
    L.dtk_pipeline_create.argtypes = [u64, u32, u32, C.POINTER(vp)]
    L.dtk_pipeline_free.argtypes = [vp]
    L.dtk_pipeline_free.restype = None
    L.dtk_pipeline_set_chunking.argtypes = [vp, u32, u32]
    L.dtk_pipeline_run.argtypes = [vp, vp, vp, vp, u32, u32, SLICE_FN, vp]
    L.dtk_multi_create.argtypes = [C.c_char_p, C.POINTER(C.c_int), u32, u64, u32, u32, C.POINTER(vp)]
    L.dtk_multi_free.argtypes = [vp]
    L.dtk_multi_free.restype = None
    L.dtk_multi_type.argtypes = [vp]
    L.dtk_multi_type.restype = C.c_char_p
    L.dtk_multi_set_result_fields.argtypes = [vp, u32]
    L.dtk_multi_set_chunking.argtypes = [vp, u32, u32]
    L.dtk_multi_run.argtypes = [vp, vp, vp, u32, u32, SLICE_FN, vp]
    L.dtk_pinned_alloc.restype = vp
    L.dtk_pinned_alloc.argtypes = [sz]
    L.dtk_pinned_free.argtypes = [vp]
    L.dtk_pinned_free.restype = None
    # Test hooks: the library itself never reads the environment; this harness forwards the DATOK_* switches the tests
    # and scripts set (dtk_debug_configure; unknown names are not the library's and are left alone).
    L.dtk_debug_configure.argtypes = [C.c_char_p, C.c_char_p]
    for k, v in os.environ.items():
        if k.startswith("DATOK_") and k not in ("DATOK_GPU_LIB", "DATOK_GATHER_TIMEOUT"):
            L.dtk_debug_configure(k.encode(), v.encode())
    _lib = L
    return L


def check(code, what=""):
    if code != OK:
        raise DatokGpuError(code, what)
