#!/usr/bin/env python3
"""Cycle probe of the lean walk loop (a library built with -DDTK_PROBE, see walk_fused): how long a wave waits
for the cell and the stream entry per iteration.  usage: DATOK_GPU_LIB=ab/libP.so probe.py [batches in flight]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import datok_amd  # noqa: E402
from datok_amd import corpus, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tok = datok_amd.load_tokenizer_file(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok"))
lib = ctypes.CDLL(os.environ["DATOK_GPU_LIB"])
bs = []
for k in range(n):
    t, o = corpus.german_docs(4096, 4096, seed=2 + k)
    b = datok_amd.Batch(len(t), 4096)
    b.set_input(t, o)
    b.run(tok, 256); b.totals()
    bs.append(b)
out = (ctypes.c_ulonglong * 8)()
lib.dtk_probe_read(out, 1)
lib.dtk_phase_read((ctypes.c_ulonglong * 8)(), 1)
lib.dtk_cphase_read((ctypes.c_ulonglong * 8)(), 1)
t0 = time.perf_counter()
R = 10
for i in range(R):
    for b in bs:
        b.run(tok, 256)
    for b in bs:
        b.totals()
dt = time.perf_counter() - t0
lib.dtk_probe_read(out, 1)
v = list(out)
if v[3]:
    print("window refills: %.1f per wave (of %.1f iterations), %.0f cycles each, %.0f cycles per wave" % (
        v[5] / v[3], v[2] / v[3], v[4] / max(v[5], 1), v[4] / v[3]))
for name, (w, tot, it, waves) in (("chunk", v[0:4]),):
    if waves:
        print("%-8s waves %d  iterations/wave %.1f  loop cycles/wave %.0f  cycles/iteration %.0f  of which waiting %.0f (%.0f %%)" % (
            name, waves // (R * n), it / waves, tot / waves, tot / max(it, 1), w / max(it, 1), 100.0 * w / max(tot, 1)))
ph = (ctypes.c_ulonglong * 8)()
lib.dtk_phase_read(ph, 0)
w = max(ph[4], 1)
print("k_spec_both cycles per wave: prologue + blank/tag search %.0f, warm-up walk %.0f, chunk walk %.0f, epilogue %.0f" % (
    ph[0] / w, ph[1] / w, ph[2] / w, ph[3] / w))
cp = (ctypes.c_ulonglong * 8)()
lib.dtk_cphase_read(cp, 0)
w = max(cp[6], 1)
print("k_compact_plain cycles per wave (%.1f tiles): prologue %.0f, tile loads + rune scan %.0f, counts + latch %.0f, token loop %.0f, "
      "sentence loop + carries %.0f, tail %.0f" % ((cp[7] / w,) + tuple(cp[i] / w for i in range(6))))
print("step %.1f us with %d in flight" % (dt / R / n * 1e6, n))
