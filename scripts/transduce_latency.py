import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datok_amd
tok = datok_amd.load_tokenizer_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'models', 'tokenizer_de.matok'))
s = "Der alte Mann ging nach Hause. Es war spät!".encode()
tok.transduce_bytes(s)
t0 = time.perf_counter()
for _ in range(200): tok.transduce_bytes(s)
print("dtk_transduce on %d bytes: %.1f us per call" % (len(s), (time.perf_counter() - t0) / 200 * 1e6))
big = s * 20000
tok.transduce_bytes(big)
t0 = time.perf_counter()
for _ in range(20): tok.transduce_bytes(big)
dt = (time.perf_counter() - t0) / 20
print("dtk_transduce on %.1f MB: %.2f ms per call = %.2f GB/s incl. upload, render, download" % (len(big) / 1e6, dt * 1e3, len(big) / dt / 1e9))
# the closure-replay path's walk + results on the host (dtk_transduce_result: what TransduceTokenWriter with a custom
# writer calls): events, byte ranges, row offsets, status in page-locked memory
import ctypes as C
from datok_amd import _lib
L = datok_amd.lib()
v = _lib.ResultView()
for text in (s, s * 100, big):
    L.dtk_transduce_result(tok._h, text, len(text), 0, C.byref(v))
    n = 200 if len(text) < 100000 else 20
    t0 = time.perf_counter()
    for _ in range(n): L.dtk_transduce_result(tok._h, text, len(text), 0, C.byref(v))
    dt = (time.perf_counter() - t0) / n
    print("dtk_transduce_result on %d bytes: %.1f us per call (%.2f GB/s)" % (len(text), dt * 1e6, len(text) / dt / 1e9))
