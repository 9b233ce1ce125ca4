"""Host-side mirror of Datok's Go API over the C-ABI (no compute here)."""
import ctypes as C
import io
import sys

import numpy as np

from . import _lib
from ._lib import ModelInfo, RenderView, ResultView, Totals, check, lib

# token_writer.go:17-25
TOKENS, SENTENCES, TOKEN_POS, SENTENCE_POS, NEWLINE_AFTER_EOT = 1, 2, 4, 8, 16
SIMPLE = TOKENS | SENTENCES
# dtk_batch_run only (datok_gpu.h): no renderer bookkeeping / only one kind of token offsets
OFFSETS_ONLY, NO_BYTE_OFFSETS, NO_RUNE_OFFSETS = 256, 512, 1024
OFFSETS_ONLY = 256   # Batch.run only: skip the device renderer's bookkeeping

# calls at one cursor position, in the order they fire (flag byte of event_bytes(); datok_gpu.h DTK_EVB_* / DTK_TAIL_*)
EV_S_EOT, EV_E_EOT, EV_TOK_END, EV_S_EPS, EV_S_EOF, EV_E_EOF = 1, 2, 4, 8, 32, 64
EVB_END, EVB_START, EVB_SEPS, EVB_TEOT, EVB_SEOT = 0, 1, 2, 3, 4


# ------------------------------------------------------------------ UTF-8 (Go)
def _decode_runes(b: bytes):
    """Go's rune iteration: invalid bytes become U+FFFD of width 1."""
    out, i, n = [], 0, len(b)
    while i < n:
        b0 = b[i]
        r, w = 0xFFFD, 1
        if b0 < 0x80:
            r = b0
        elif 0xC2 <= b0 < 0xE0:
            if i + 1 < n and (b[i + 1] & 0xC0) == 0x80:
                r, w = ((b0 & 0x1F) << 6) | (b[i + 1] & 0x3F), 2
        elif 0xE0 <= b0 < 0xF0:
            lo, hi = (0xA0 if b0 == 0xE0 else 0x80), (0x9F if b0 == 0xED else 0xBF)
            if i + 2 < n and lo <= b[i + 1] <= hi and (b[i + 2] & 0xC0) == 0x80:
                r, w = ((b0 & 0x0F) << 12) | ((b[i + 1] & 0x3F) << 6) | (b[i + 2] & 0x3F), 3
        elif 0xF0 <= b0 <= 0xF4:
            lo, hi = (0x90 if b0 == 0xF0 else 0x80), (0x8F if b0 == 0xF4 else 0xBF)
            if (i + 3 < n and lo <= b[i + 1] <= hi and (b[i + 2] & 0xC0) == 0x80
                    and (b[i + 3] & 0xC0) == 0x80):
                r, w = (((b0 & 0x07) << 18) | ((b[i + 1] & 0x3F) << 12)
                        | ((b[i + 2] & 0x3F) << 6) | (b[i + 3] & 0x3F)), 4
        out.append(r)
        i += w
    return out


def _runes_to_bytes(runes):
    return "".join(chr(r) if not (0xD800 <= r <= 0xDFFF or r > 0x10FFFF) else "�"
                   for r in runes).encode("utf-8")


# ----------------------------------------------------------------- TokenWriter
class TokenWriter:
    """token_writer.go:27-33: four closures."""
    __slots__ = ("SentenceEnd", "TextEnd", "Flush", "Token")


def new_token_writer(w, flags) -> TokenWriter:
    """token_writer.go:36-175 NewTokenWriter. `w` is a binary file-like object."""
    st = {"posC": 0, "pos": [], "sentB": True, "sent": [], "init": True}
    out = bytearray()
    tw = TokenWriter()

    def flush():
        if out:
            w.write(bytes(out))
            out.clear()
        if hasattr(w, "flush"):
            w.flush()

    def surface(offset, buf):
        out.extend(_runes_to_bytes(buf[offset:]))
        out.append(10)

    if flags & (TOKEN_POS | SENTENCE_POS):
        def token(offset, buf):
            if st["posC"] == 0 and flags & NEWLINE_AFTER_EOT and buf and buf[0] == 10 and not st["init"]:
                st["posC"] -= 1
            st["init"] = False
            st["posC"] += offset
            st["pos"].append(st["posC"])
            if st["sentB"]:
                st["sentB"] = False
                st["sent"].append(st["posC"])
            st["posC"] += len(buf) - offset
            st["pos"].append(st["posC"])
            if flags & TOKENS:
                surface(offset, buf)
    elif flags & TOKENS:
        token = surface
    else:
        def token(offset, buf):
            pass
    tw.Token = token

    if flags & SENTENCE_POS:
        def sentence_end(_):
            st["sent"].append(st["pos"][-1])     # Go panics on an empty pos
            st["sentB"] = True
            if flags & SENTENCES:
                out.append(10)
    elif flags & SENTENCES:
        def sentence_end(_):
            out.append(10)
            flush()
    else:
        def sentence_end(_):
            pass
    tw.SentenceEnd = sentence_end

    if flags & (TOKEN_POS | SENTENCE_POS):
        def text_end(_):
            if flags & TOKEN_POS:
                out.extend(" ".join(map(str, st["pos"])).encode() if st["pos"] else b"")
                st["pos"][0]                      # Go: pos[0] panics on an empty text
                out.append(10)
            if flags & SENTENCE_POS:
                st["sent"][0]
                out.extend(" ".join(map(str, st["sent"])).encode())
                out.append(10)
                st["sent"] = []
                st["sentB"] = True
            flush()
            st["posC"] = 0
            st["pos"] = []
    else:
        def text_end(_):
            out.append(10)
            flush()
    tw.TextEnd = text_end
    tw.Flush = flush
    return tw


def event_bytes(ev_bits, doc_off_d, d, n, tail):
    """One flag byte per cursor position 0..n of document d (EV_* bits, bit order = call order) from the event
    bitmaps of a host dtk_result_view (ev_bits: uint32[kinds, words]) and the document's tail word."""
    g0 = int(doc_off_d) + int(d)
    w0, w1 = g0 >> 5, (g0 + n + 32) >> 5
    out = np.zeros(n + 1, dtype=np.uint8)
    for kind, flag in ((EVB_SEOT, EV_S_EOT), (EVB_TEOT, EV_E_EOT), (EVB_END, EV_TOK_END), (EVB_SEPS, EV_S_EPS)):
        words = np.ascontiguousarray(ev_bits[kind, w0:w1 + 1])
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[g0 - 32 * w0: g0 - 32 * w0 + n + 1]
        out |= bits * np.uint8(flag)
    tail = int(tail)
    if tail & 3:
        out[min(tail >> 2, n)] |= (EV_S_EOF if tail & 1 else 0) | (EV_E_EOF if tail & 2 else 0)
    return out


def replay(is_matrix, text: bytes, events, tok_bstart, tw: TokenWriter):
    """Feeds one document's events (event_bytes()) to the closures in reference call order.

    Int arguments as upstream: the matrix passes buffc (matrix.go:575,597,600,684,691);
    the double array 0, except SentenceEnd(buffc) at EOT (datok.go:1015,1023,1026)."""
    n = len(text)
    B = k = 0
    nz = np.flatnonzero(np.asarray(events[:n + 1]))
    for p in nz.tolist():
        e = int(events[p])
        def buffc():
            return len(_decode_runes(text[B:p]))
        if e & EV_S_EOT:
            tw.SentenceEnd(buffc())
        if e & EV_E_EOT:
            tw.TextEnd(buffc() if is_matrix else 0)
            if is_matrix:
                B = p
        if e & EV_TOK_END:
            start = int(tok_bstart[k])
            k += 1
            buf = _decode_runes(text[B:p])
            tw.Token(len(_decode_runes(text[B:start])), buf)
            B = p
        if e & EV_S_EPS:
            tw.SentenceEnd(buffc() if is_matrix else 0)
        if e & EV_S_EOF:
            tw.SentenceEnd(buffc() if is_matrix else 0)
        if e & EV_E_EOF:
            tw.TextEnd(buffc() if is_matrix else 0)


def replay_calls(text: bytes, calls, tw: TokenWriter):
    """Feeds the call list of a document walked by the exact pass (dtk_result_view.calls: rows of
    (kind, a, b, c)) to the closures: kind 0 Token(offset, buf) with buf = runes of text[a:c] and offset =
    runes of text[a:b]; kind 1 SentenceEnd(a); kind 2 TextEnd(a)."""
    for kind, a, b, c in calls:
        if kind == 0:
            tw.Token(len(_decode_runes(text[a:b])), _decode_runes(text[a:c]))
        elif kind == 1:
            tw.SentenceEnd(a)
        else:
            tw.TextEnd(a)


def _exact_calls(v, arr):
    """dict doc id -> list of (kind, a, b, c) from a host dtk_result_view."""
    n = int(v.n_exact)
    if n == 0:
        return {}
    ids = arr(v.exact_doc, n, np.uint32)
    off = arr(v.exact_off, n + 1, np.uint64)
    calls = arr(v.calls, int(off[-1]) * 4, np.int32).reshape(-1, 4)
    return {int(ids[i]): [(int(k) & 0xFFFFFFFF, int(a), int(b) & 0xFFFFFFFF, int(c) & 0xFFFFFFFF)
                          for k, a, b, c in calls[int(off[i]):int(off[i + 1])]] for i in range(n)}


# statuses on which the reference cannot finish a document whatever the writer does (matrix.go:365,406 index
# panic; a walk that left the table; the lookup cap; an internal check): TransduceTokenWriter returns false
_FATAL = _lib.ST_WINDOW_OVERFLOW | _lib.ST_BAD_MODEL | _lib.ST_STEP_LIMIT | _lib.ST_INTERNAL | _lib.ST_IRREGULAR


# ------------------------------------------------------------------- Tokenizer
class Tokenizer:
    """fomafile.go:29-33 Tokenizer, backed by a device-resident model."""

    def __init__(self, handle):
        self._h = handle
        info = ModelInfo()
        check(lib().dtk_model_get_info(self._h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in ModelInfo._fields_}
        self.last_status = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().dtk_model_free(h)

    def type(self) -> str:
        """matrix.go:102 / datok.go:252"""
        return lib().dtk_model_type(self._h).decode()

    def transduce(self, r, w) -> bool:
        """matrix.go:340-342: TransduceTokenWriter(r, NewTokenWriter(w, SIMPLE))."""
        return self.transduce_token_writer(r, new_token_writer(w, SIMPLE))

    def transduce_token_writer(self, r, tw: TokenWriter) -> bool:
        """matrix.go:348-698 / datok.go:781-1135. One reader = one document."""
        text = r.read() if hasattr(r, "read") else bytes(r)
        if isinstance(text, str):
            text = text.encode("utf-8")
        v = ResultView()   # dtk_transduce_result: the library's per-thread batch, host pointers
        rc = lib().dtk_transduce_result(self._h, text, len(text), 0, C.byref(v))
        if rc != _lib.OK:                      # the reference returns false
            print("datok_amd:", _lib.DatokGpuError(rc, "dtk_transduce_result"), file=sys.stderr)
            return False
        n_ev = len(text) + 1
        ntok = int(np.frombuffer((C.c_char * 16).from_address(v.tok_off), dtype=np.uint64)[1])

        def arr(ptr, n, dt):
            return np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt) if n else np.zeros(0, dt)
        self.last_status = int(arr(v.status, 1, np.uint32)[0])
        exact = _exact_calls(v, arr)
        if 0 in exact:                          # walked by the exact pass: its calls are listed in order
            replay_calls(text, exact[0], tw)
        else:
            bits = arr(v.ev_bits, 5 * int(v.ev_words), np.uint32).reshape(5, -1)
            events = event_bytes(bits, 0, 0, len(text), arr(v.doc_tail, 1, np.uint32)[0])
            replay(self.type() == "MATOK", text, events, arr(v.tok_bstart, ntok, np.uint32), tw)
        tw.Flush()                              # `defer w.Flush()`, matrix.go:374
        return not (self.last_status & _FATAL)

    def transduce_bytes(self, text: bytes, flags=SIMPLE, replay=False):
        """dtk_transduce(): walked and rendered on the device; replay=True: dtk_transduce_replay(), the
        event bytes replayed into the C++ host mirror of NewTokenWriter. Returns (output, status)."""
        out, n, st = C.c_void_p(), C.c_size_t(), C.c_uint32()
        fn = lib().dtk_transduce_replay if replay else lib().dtk_transduce
        check(fn(self._h, text, len(text), flags, C.byref(out), C.byref(n), C.byref(st)))
        try:
            return C.string_at(out, n.value), st.value
        finally:
            lib().dtk_free(out)


def load_tokenizer_file(path):
    """fomafile.go:452-484: returns None (and logs) on any failure."""
    h = C.c_void_p()
    rc = lib().dtk_model_load(str(path).encode(), C.byref(h))
    if rc != _lib.OK:
        if rc in (_lib.E_NO_DEVICE, _lib.E_HIP):
            raise _lib.DatokGpuError(rc, "load_tokenizer_file")   # environment, not the file
        print("datok_amd: %s: %s" % (path, lib().dtk_strerror(rc).decode()), file=sys.stderr)
        return None
    return Tokenizer(h)


# ----------------------------------------------------------------------- Batch
def event_bit(doc_off_d, d):
    """DTK_EVENT_BIT: bit of position 0 of document d in the event bitmaps."""
    return int(doc_off_d) + int(d)


class BatchResult:
    """Host copy of dtk_result_view (CSR over documents)."""
    __slots__ = ("tok_off", "sent_off", "text_off", "tok_rstart", "tok_rend", "tok_bstart",
                 "tok_bend", "sent", "text_tok_end", "text_sent_end", "status", "ev_bits", "doc_tail", "doc_off", "exact",
                 "tok_r16")

    def events(self, d):
        """Flag byte per cursor position of document d (event_bytes()), for replays."""
        n = int(self.doc_off[d + 1]) - int(self.doc_off[d])
        return event_bytes(self.ev_bits, self.doc_off[d], d, n, self.doc_tail[d])

    def doc(self, d):
        """The rows of document d (arrays the run did not write or the caller did not select are empty)."""
        a, b = int(self.tok_off[d]), int(self.tok_off[d + 1])
        s0, s1 = int(self.sent_off[d]), int(self.sent_off[d + 1])
        t0, t1 = int(self.text_off[d]), int(self.text_off[d + 1])
        rs, re = self.tok_rstart[a:b], self.tok_rend[a:b]
        if len(self.tok_r16) and not len(self.tok_rstart):  # (R_TOK_RUNE16: int16 pairs {start, end})
            rs, re = self.tok_r16[a:b, 0].astype(np.int32), self.tok_r16[a:b, 1].astype(np.int32)
        return dict(tok_rstart=rs, tok_rend=re,
                    tok_bstart=self.tok_bstart[a:b], tok_bend=self.tok_bend[a:b],
                    sent=self.sent[s0:s1], text_tok_end=self.text_tok_end[t0:t1],
                    text_sent_end=self.text_sent_end[t0:t1], status=int(self.status[d]))


class Batch:
    """dtk_batch: device buffers + one HIP stream for many documents per launch."""

    def __init__(self, max_bytes, max_docs):
        self._h = C.c_void_p()
        check(lib().dtk_batch_create(int(max_bytes), int(max_docs), C.byref(self._h)), "dtk_batch_create")
        self._keep = None
        self.n_docs = 0
        self.total = 0
        self._doc_off = None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().dtk_batch_free(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def stream(self):
        return lib().dtk_batch_stream(self._h)

    def set_input(self, text: np.ndarray, doc_off: np.ndarray):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        self._keep = (text, doc_off)
        self.n_docs = len(doc_off) - 1
        self.total = int(doc_off[-1])
        self._doc_off = doc_off
        check(lib().dtk_batch_set_input(self._h, text.ctypes.data, doc_off.ctypes.data, self.n_docs),
              "dtk_batch_set_input")

    def set_input_device(self, d_text_ptr, d_doc_off_ptr, n_docs, total_bytes, keep=None, doc_off_host=None):
        """Device-resident input (e.g. torch tensors' data_ptr()); `keep` pins the owners."""
        self._keep = keep
        self.n_docs = int(n_docs)
        self.total = int(total_bytes)
        self._doc_off = doc_off_host
        check(lib().dtk_batch_set_input_device(self._h, d_text_ptr, d_doc_off_ptr, self.n_docs, self.total),
              "dtk_batch_set_input_device")

    def run(self, tok: Tokenizer, flags=0):
        check(lib().dtk_batch_run(tok._h, self._h, flags), "dtk_batch_run")

    def sync(self):
        check(lib().dtk_batch_sync(self._h), "dtk_batch_sync")

    AUTO_CHUNK = 0xFFFFFFFF

    def set_chunking(self, chunk_bytes=AUTO_CHUNK, warm_bytes=8, extend=None):
        """0: one lane per document; otherwise speculative chunk lanes (exact either way).
        extend: how far the warm-up start may move back to the previous blank (None: library default,
        0: fixed distance only -- what tests use to force mispredictions)."""
        check(lib().dtk_batch_set_chunking(self._h, int(chunk_bytes), int(warm_bytes)), "dtk_batch_set_chunking")
        if extend is not None:
            check(lib().dtk_batch_set_warm_extend(self._h, int(extend)), "dtk_batch_set_warm_extend")

    def set_profiling(self, enable=True):
        check(lib().dtk_batch_set_profiling(self._h, int(bool(enable))), "dtk_batch_set_profiling")

    STAGES = ("clear", "symbolize", "spec_start", "spec_link", "walk", "spec_verify", "spec_fix", "scan", "compact")

    def stage_ms(self):
        """Milliseconds per stage of the last run (needs set_profiling(True))."""
        ms = (C.c_float * 9)()
        check(lib().dtk_batch_stage_ms(self._h, C.byref(ms)), "dtk_batch_stage_ms")
        return dict(zip(self.STAGES, [float(x) for x in ms]))

    def totals(self):
        t = Totals()
        check(lib().dtk_batch_totals(self._h, C.byref(t)), "dtk_batch_totals")
        return {k: getattr(t, k) for k, _ in Totals._fields_}

    def result_device(self) -> ResultView:
        v = ResultView()
        check(lib().dtk_batch_result_device(self._h, C.byref(v)), "dtk_batch_result_device")
        return v

    def render_device(self, flags=SIMPLE) -> RenderView:
        """NewTokenWriter(w, flags) for every document, left on the device (pointers + total)."""
        v = RenderView()
        check(lib().dtk_batch_render_device(self._h, flags, C.byref(v)), "dtk_batch_render_device")
        return v

    def render(self, flags=SIMPLE):
        """Returns (bytes, doc_off): bytes[doc_off[d]:doc_off[d+1]] is what the reference writes for document d."""
        v = RenderView()
        check(lib().dtk_batch_render_host(self._h, flags, C.byref(v)), "dtk_batch_render_host")
        off = np.frombuffer((C.c_char * ((self.n_docs + 1) * 8)).from_address(v.doc_off), dtype=np.uint64).copy()
        data = C.string_at(v.bytes, v.total) if v.total else b""
        return data, off

    # dtk_batch_set_result_fields (datok_gpu.h DTK_R_*)
    R_CSR, R_TOK_RUNE, R_TOK_BYTE, R_SENT, R_TEXTS, R_STATUS, R_EVENTS, R_ALL = 1, 2, 4, 8, 16, 32, 64, 127
    R_TOK_RUNE16 = 128   # the rune offsets as int16 pairs (BatchResult.tok_r16); a batch with a document longer than
                         # 32 767 bytes gets tok_rstart / tok_rend in their place

    def set_result_fields(self, fields=R_ALL):
        """Which arrays result() brings to the host (the others come back empty)."""
        check(lib().dtk_batch_set_result_fields(self._h, int(fields)), "dtk_batch_set_result_fields")

    def download_begin(self):
        """Completes the run and starts the asynchronous copies of the selected arrays; result() waits for them."""
        check(lib().dtk_batch_download_begin(self._h), "dtk_batch_download_begin")

    def result(self, copy=True) -> BatchResult:
        """The result arrays on the host (dtk_batch_result_host).  copy=False: views of the batch's page-locked
        buffers, valid until its next run."""
        v = ResultView()
        check(lib().dtk_batch_result_host(self._h, C.byref(v)), "dtk_batch_result_host")
        t = self.totals()
        nd = self.n_docs

        def arr(ptr, n, dt):
            if n == 0 or not ptr:   # (an array that was not selected comes back NULL)
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dtype=dt)
            return a.copy() if copy else a
        r = BatchResult()
        r.tok_off = arr(v.tok_off, nd + 1, np.uint64)
        r.sent_off = arr(v.sent_off, nd + 1, np.uint64)
        r.text_off = arr(v.text_off, nd + 1, np.uint64)
        r.tok_rstart = arr(v.tok_rstart, t["n_tokens"], np.int32)
        r.tok_rend = arr(v.tok_rend, t["n_tokens"], np.int32)
        r.tok_bstart = arr(v.tok_bstart, t["n_tokens"], np.uint32)
        r.tok_bend = arr(v.tok_bend, t["n_tokens"], np.uint32)
        r.sent = arr(v.sent, t["n_sent"], np.int32)
        r.text_tok_end = arr(v.text_tok_end, t["n_texts"], np.uint32)
        r.text_sent_end = arr(v.text_sent_end, t["n_texts"], np.uint32)
        r.status = arr(v.status, nd, np.uint32)
        r.ev_bits = arr(v.ev_bits, 5 * int(v.ev_words), np.uint32).reshape(5, -1) if v.ev_bits else np.zeros((5, 0), np.uint32)
        r.doc_tail = arr(v.doc_tail, nd, np.uint32)
        r.tok_r16 = arr(v.tok_r16, 2 * t["n_tokens"], np.int16).reshape(-1, 2)
        r.doc_off = self._doc_off
        r.exact = _exact_calls(v, arr)   # documents walked by the exact pass: id -> calls in order
        return r


class PinnedBuffer:
    """Page-locked host memory (dtk_pinned_alloc) as a numpy uint8 array: uploads from it are asynchronous."""

    def __init__(self, n):
        self._p = lib().dtk_pinned_alloc(int(n))
        if not self._p:
            raise MemoryError("dtk_pinned_alloc(%d)" % n)
        self.array = np.frombuffer((C.c_char * int(n)).from_address(self._p), dtype=np.uint8)

    def close(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            self.array = None
            lib().dtk_pinned_free(p)

    __del__ = close


class Pipeline:
    """dtk_pipeline: a corpus larger than one batch, cut into slices at document boundaries; the upload of one
    slice overlaps the walk of the others.  run() calls `on_slice(first_doc, n_docs, batch)` for every finished
    slice in order (batch: a Batch view of the slice, valid inside the callback)."""

    def __init__(self, slice_bytes, slice_docs, depth=3):
        self._h = C.c_void_p()
        check(lib().dtk_pipeline_create(int(slice_bytes), int(slice_docs), int(depth), C.byref(self._h)), "dtk_pipeline_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().dtk_pipeline_free(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_chunking(self, chunk_bytes=0xFFFFFFFF, warm_bytes=8):
        check(lib().dtk_pipeline_set_chunking(self._h, int(chunk_bytes), int(warm_bytes)), "dtk_pipeline_set_chunking")

    def set_result_fields(self, fields):
        """Bring these result arrays (Batch.R_*) of every slice to the host, overlapped with the next slices' work."""
        check(lib().dtk_pipeline_set_result_fields(self._h, int(fields)), "dtk_pipeline_set_result_fields")

    def run(self, tok, text: np.ndarray, doc_off: np.ndarray, flags=0, on_slice=None):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        err = []
        fn = _slice_callback(on_slice, doc_off, err)
        rc = lib().dtk_pipeline_run(self._h, tok._h, text.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1, flags, fn, None)
        if err:
            raise err[0]
        check(rc, "dtk_pipeline_run")


def _slice_callback(on_slice, doc_off, err):
    """The C callback of dtk_pipeline_run / dtk_multi_run around a Python on_slice(first, n, batch_view)."""
    def cb(_user, first, n, handle):
        if on_slice is None:
            return 0
        view = Batch.__new__(Batch)
        try:
            view._h = C.c_void_p(handle); view._keep = None; view.n_docs = int(n)
            view._doc_off = (doc_off[first:first + n + 1] - doc_off[first]).astype(np.uint64)
            view.total = int(view._doc_off[-1])
            on_slice(int(first), int(n), view)
            return 0
        except Exception as e:  # noqa: BLE001  (do not unwind through the C frame)
            err.append(e)
            return _lib.E_STATE
        finally:
            # the pipeline owns the batch: the view must never free it, also not when on_slice raised and the
            # exception's traceback keeps the view alive (a double dtk_batch_free otherwise)
            view._h = None
    return _lib.SLICE_FN(cb)


class MultiPipeline:
    """dtk_multi: a corpus sharded over several GPUs of one node behind the C-ABI -- one worker thread, model replica
    and pipeline per listed device, slices dealt round-robin, finished slices handed to on_slice in corpus order."""

    def __init__(self, model_path, devices, slice_bytes, slice_docs, depth=3):
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib().dtk_multi_create(str(model_path).encode(), devs, len(devices), int(slice_bytes), int(slice_docs),
                                     int(depth), C.byref(self._h)), "dtk_multi_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().dtk_multi_free(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def type(self):
        return lib().dtk_multi_type(self._h).decode()

    def set_result_fields(self, fields):
        check(lib().dtk_multi_set_result_fields(self._h, int(fields)), "dtk_multi_set_result_fields")

    def set_chunking(self, chunk_bytes=0xFFFFFFFF, warm_bytes=8):
        check(lib().dtk_multi_set_chunking(self._h, int(chunk_bytes), int(warm_bytes)), "dtk_multi_set_chunking")

    def run(self, text: np.ndarray, doc_off: np.ndarray, flags=0, on_slice=None):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        err = []
        fn = _slice_callback(on_slice, doc_off, err)
        rc = lib().dtk_multi_run(self._h, text.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1, flags, fn, None)
        if err:
            raise err[0]
        check(rc, "dtk_multi_run")


def foma_to_matok(foma_gz: bytes) -> bytes:
    """`datok convert` (cmd/datok.go:50-70, matrix variant): LoadFomaFile + ToMatrix + Save.

    Takes the bytes of a gzip'd Foma text net and returns the bytes of the .matok file. Host only.
    """
    out, n = C.c_void_p(), C.c_size_t()
    check(lib().dtk_foma_to_matok(foma_gz, len(foma_gz), C.byref(out), C.byref(n)), "foma_to_matok")
    try:
        return C.string_at(out, n.value)
    finally:
        lib().dtk_free(out)


def foma_to_datok(foma_gz: bytes) -> bytes:
    """`datok convert --double-array` (cmd/datok.go:50-70): LoadFomaFile + ToDoubleArray (datok.go:82-238) + Save.

    The reference's layout depends on Go's map iteration order; this one takes a state's symbols in ascending order.
    Host only.
    """
    out, n = C.c_void_p(), C.c_size_t()
    check(lib().dtk_foma_to_datok(foma_gz, len(foma_gz), C.byref(out), C.byref(n)), "foma_to_datok")
    try:
        return C.string_at(out, n.value)
    finally:
        lib().dtk_free(out)


def load_foma_file(path):
    """LoadFomaFile(path).ToMatrix() (fomafile.go:56-75, matrix.go:30-99) on the device."""
    return load_tokenizer_file(path)
