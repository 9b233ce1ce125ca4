"""ctypes binding of the CPU oracle (oracle/datok_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path (datok_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

TOKENS, SENTENCES, TOKEN_POS, SENTENCE_POS, NEWLINE_AFTER_EOT = 1, 2, 4, 8, 16
SIMPLE = TOKENS | SENTENCES
ST_WINDOW_OVERFLOW, ST_EMPTY_TEXT, ST_BAD_MODEL = 1, 2, 4
ST_BAD_OFFSET = 64


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("datok_oracle.c", "datok_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB


class _DocResult(C.Structure):
    _fields_ = [
        ("n_tok", C.c_uint32),
        ("tok_rstart", C.POINTER(C.c_int32)), ("tok_rend", C.POINTER(C.c_int32)),
        ("tok_bstart", C.POINTER(C.c_uint32)), ("tok_bend", C.POINTER(C.c_uint32)),
        ("n_sent", C.c_uint32), ("sent", C.POINTER(C.c_int32)),
        ("n_text", C.c_uint32),
        ("text_tok_end", C.POINTER(C.c_uint32)), ("text_sent_end", C.POINTER(C.c_uint32)),
        ("n_sent_events", C.c_uint32),
        ("status", C.c_uint),
        ("steps", C.c_uint64),
    ]


class _Info(C.Structure):
    _fields_ = [("kind", C.c_int), ("epsilon", C.c_int), ("unknown", C.c_int),
                ("identity", C.c_int), ("final_", C.c_int), ("sigma_count", C.c_int),
                ("state_count", C.c_uint32), ("array_len", C.c_uint64),
                ("n_sigma_runes", C.c_int)]


class _Event(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_int32), ("b", C.c_uint32),
                ("c", C.c_uint32), ("d", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_load_file.restype = C.c_void_p
        L.orc_load_file.argtypes = [C.c_char_p]
        L.orc_load_foma_file.restype = C.c_void_p
        L.orc_load_foma_file.argtypes = [C.c_char_p]
        L.orc_parse.restype = C.c_void_p
        L.orc_parse.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_free_model.argtypes = [C.c_void_p]
        L.orc_type.restype = C.c_char_p
        L.orc_type.argtypes = [C.c_void_p]
        L.orc_info.argtypes = [C.c_void_p, C.POINTER(_Info)]
        L.orc_array.restype = C.POINTER(C.c_uint32)
        L.orc_array.argtypes = [C.c_void_p]
        L.orc_sigma_ascii.restype = C.POINTER(C.c_int)
        L.orc_sigma_ascii.argtypes = [C.c_void_p]
        L.orc_sigma_lookup.restype = C.c_int
        L.orc_sigma_lookup.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
        L.orc_decode_rune.restype = C.c_int
        L.orc_decode_rune.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.orc_transduce_string.restype = C.c_void_p
        L.orc_transduce_string.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_uint)]
        L.orc_transduce_doc.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint,
                                        C.POINTER(_DocResult)]
        L.orc_free_doc_result.argtypes = [C.POINTER(_DocResult)]
        L.orc_transduce_events.restype = C.POINTER(_Event)
        L.orc_transduce_events.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_uint)]
        L.orc_count_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                      C.c_int, C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class DocResult:
    __slots__ = ("tok_rstart", "tok_rend", "tok_bstart", "tok_bend", "sent",
                 "text_tok_end", "text_sent_end", "n_sent_events", "status", "steps")

    def __repr__(self):
        return ("DocResult(tok=%d sent=%d text=%d status=%d)" %
                (len(self.tok_rstart), len(self.sent), len(self.text_tok_end), self.status))


class Model:
    """One loaded tokenizer (fomafile.go:452-484 LoadTokenizerFile)."""

    def __init__(self, path=None, raw=None):
        L = lib()
        if path is not None and str(path).endswith(".fst"):
            self._h = L.orc_load_foma_file(os.fsencode(path))   # LoadFomaFile(...).ToMatrix()
        elif path is not None:
            self._h = L.orc_load_file(os.fsencode(path))
        else:
            self._h = L.orc_parse(raw, len(raw))
        if not self._h:
            raise ValueError("oracle: cannot load tokenizer %r" % (path,))
        info = _Info()
        L.orc_info(self._h, C.byref(info))
        self.info = {k: getattr(info, k) for k, _ in _Info._fields_}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.orc_free_model(h)

    def type(self):
        return lib().orc_type(self._h).decode()

    def sigma_ascii(self):
        return np.ctypeslib.as_array(lib().orc_sigma_ascii(self._h), shape=(256,))

    def array(self):
        n = self.info["array_len"] * (2 if self.info["kind"] == 1 else 1)
        return np.ctypeslib.as_array(lib().orc_array(self._h), shape=(n,))

    def transduce(self, text: bytes, flags=SIMPLE):
        """Bytes the reference writes for Transduce / TransduceTokenWriter."""
        n, st = C.c_size_t(0), C.c_uint(0)
        p = lib().orc_transduce_string(self._h, text, len(text), flags, C.byref(n), C.byref(st))
        try:
            return C.string_at(p, n.value), st.value
        finally:
            lib().orc_free(p)

    def transduce_doc(self, text: bytes, flags=0) -> DocResult:
        r = _DocResult()
        L = lib()
        L.orc_transduce_doc(self._h, text, len(text), flags, C.byref(r))
        out = DocResult()

        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)
        out.tok_rstart = arr(r.tok_rstart, r.n_tok, np.int32)
        out.tok_rend = arr(r.tok_rend, r.n_tok, np.int32)
        out.tok_bstart = arr(r.tok_bstart, r.n_tok, np.uint32)
        out.tok_bend = arr(r.tok_bend, r.n_tok, np.uint32)
        out.sent = arr(r.sent, r.n_sent, np.int32)
        out.text_tok_end = arr(r.text_tok_end, r.n_text, np.uint32)
        out.text_sent_end = arr(r.text_sent_end, r.n_text, np.uint32)
        out.n_sent_events = r.n_sent_events
        out.status = r.status
        out.steps = r.steps
        L.orc_free_doc_result(C.byref(r))
        return out

    def events(self, text: bytes):
        n, st = C.c_size_t(0), C.c_uint(0)
        p = lib().orc_transduce_events(self._h, text, len(text), C.byref(n), C.byref(st))
        try:
            return [(p[i].kind, p[i].a, p[i].b, p[i].c, p[i].d) for i in range(n.value)], st.value
        finally:
            lib().orc_free(p)

    def count_batch(self, text: np.ndarray, doc_off: np.ndarray, nthreads=1):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        n_docs = len(doc_off) - 1
        counts = np.zeros((n_docs, 3), dtype=np.uint32)
        lib().orc_count_batch(self._h, text.ctypes.data, doc_off.ctypes.data, n_docs,
                              int(nthreads), counts.ctypes.data)
        return counts


def decode_rune(b: bytes):
    r = C.c_uint32(0)
    w = lib().orc_decode_rune(b, len(b), C.byref(r))
    return r.value, w
