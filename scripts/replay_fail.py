#!/usr/bin/env python3
"""Re-runs the batch soak.py kept in gpurun_out/soak_fail.npz (same model, flags, chunking) and says which
documents differ from the oracle (offsets and rendered SIMPLE stream)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from oracle import oracle as O
from parity import oracle_doc
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_fail.npz"))
M = os.path.join(ROOT, "tests", "golden", "models")
name = str(z["model"]); text = z["text"]; off = z["off"]; flags = int(z["flags"])
chunk = int(z["chunk"]); warm = int(z["warm"]); extend = int(z["extend"])
tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
with datok_amd.Batch(len(text), len(off) - 1) as b:
    if chunk >= 0:
        b.set_chunking(chunk, warm, extend=None if extend < 0 else extend)
    b.set_input(text, off); b.run(tok, flags)
    res, tot = b.result(), b.totals()
    data, o = b.render(3 | flags)
    raw = text.tobytes()
    bad_off, bad_txt = [], []
    for d in range(len(off) - 1):
        doc = raw[int(off[d]):int(off[d + 1])]
        exp = oracle_doc(om, doc, flags); got = res.doc(d)
        if got["status"] != exp["status"] or (exp["status"] == 0 and any(not np.array_equal(got[f], exp[f]) for f in ("tok_bstart", "tok_bend", "tok_rstart", "tok_rend", "sent"))):
            bad_off.append(d)
        e, est = om.transduce(doc, 3 | flags)
        if est == 0 and not (int(res.status[d]) & ~datok_amd.ST_EMPTY_TEXT) and data[int(o[d]):int(o[d + 1])] != e:
            bad_txt.append((d, len(data[int(o[d]):int(o[d + 1])]) - len(e)))
    print(name, "flags", flags, "chunk", chunk, "warm", warm, "extend", extend, "repair rounds", tot["repair_rounds"],
          "| offsets differ:", bad_off, "| rendered text differs (doc, extra bytes):", bad_txt)
