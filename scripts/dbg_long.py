import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import oracle_doc
M = os.path.join(ROOT, "tests", "golden", "models")
rng = np.random.default_rng(21)
text, off = corpus.german_docs(96, 4096, seed=21)
raw = bytearray(text.tobytes())
alpha = list(b"abcdefghijklmnopqrstuvwxyz0123456789/_-%")
for d in range(96):
    p = d * 4096 + int(rng.integers(200, 1500))
    for _ in range(3):
        q = raw.find(b" ", p)
        L = int(rng.choice([60, 90, 130, 200, 330, 700]))
        if q < 0 or q + L + 2 >= (d + 1) * 4096 - 8:
            break
        body = (bytes(rng.choice(alpha, size=L)) if os.environ.get('NUL') else bytes(rng.choice(alpha, size=L).astype(np.uint8))) if rng.integers(0, 2) else b"x" * L
        tok_bytes = (b"https://www.example.org/" + body)[:L]
        raw[q + 1:q + 1 + L] = tok_bytes
        raw[q + 1 + L] = 0x20
        p = q + L + int(rng.integers(100, 600))
text = np.frombuffer(bytes(raw), dtype=np.uint8).copy()
name = "tokenizer_de.matok"
tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for extend in (240, 0):
  for chunk in ((0, chunk) if os.environ.get('BOTH') else (chunk,)):
    with datok_amd.Batch(len(text), len(off) - 1) as b:
        b.set_chunking(chunk, 48, extend=extend)
        b.set_input(text, off); b.run(tok, 0)
        res, tot = b.result(), b.totals()
        badl = []
        for d in range(96):
            a, e = int(off[d]), int(off[d + 1])
            exp = oracle_doc(om, bytes(raw[a:e]), 0); got = res.doc(d)
            if got["status"] != exp["status"] or not np.array_equal(got["tok_bstart"], exp["tok_bstart"]) or not np.array_equal(got["tok_bend"], exp["tok_bend"]):
                badl.append(d)
        print("chunk", chunk, "extend", extend, "repair rounds", tot["repair_rounds"], "bad docs", badl)
        for d in badl[:2]:
            a, e = int(off[d]), int(off[d + 1])
            exp = oracle_doc(om, bytes(raw[a:e]), 0); got = res.doc(d)
            gs, ge, es, ee = got["tok_bstart"], got["tok_bend"], exp["tok_bstart"], exp["tok_bend"]
            n = min(len(gs), len(es))
            k = next((i for i in range(n) if gs[i] != es[i] or ge[i] != ee[i]), n)
            print("  doc", d, "status", got["status"], "tokens", len(gs), "expected", len(es), "first diff at token", k)
            print("   got", list(zip(gs[max(0,k-2):k+4].tolist(), ge[max(0,k-2):k+4].tolist())))
            print("   exp", list(zip(es[max(0,k-2):k+4].tolist(), ee[max(0,k-2):k+4].tolist())))
            lo = int(es[k]) if k < len(es) else 0
            tp_, e_ = int(es[k]), int(ee[k])
            evb = a + 4 * d
            ev = res.events
            print("   events at token start", tp_, [int(x) for x in ev[evb + tp_ - 1: evb + tp_ + 2]], "at its end", e_, [int(x) for x in ev[evb + e_ - 1: evb + e_ + 2]],
                  "nonzero inside:", [(int(i) - evb, int(ev[i])) for i in np.flatnonzero(ev[evb + tp_ + 1: evb + e_]) + evb + tp_ + 1][:12])
            print("   text around:", bytes(raw[a + max(0, lo - 40):a + lo + 80]))
