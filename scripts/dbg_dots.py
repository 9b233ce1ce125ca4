import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd
from datok_amd import corpus
from oracle import oracle as O
from parity import oracle_doc
M = os.path.join(ROOT, "tests", "golden", "models")
name = "tokenizer_de.matok"
tok = datok_amd.load_tokenizer_file(os.path.join(M, name)); om = O.Model(os.path.join(M, name))
for doc in (b"." * 120, b"." * 401, b"a " + b"." * 300 + b" b", b"x" * 500 + b" y"):
    text, off = corpus.concat_docs([b"Hallo Welt. ", doc, b"Ende."])
    for chunk, warm, extend in ((0, 48, None), (128, 48, None), (128, 48, 0), (64, 48, None), (64, 0, 0), (None, 48, None)):
        with datok_amd.Batch(len(text), len(off) - 1) as b:
            if chunk is not None:
                b.set_chunking(chunk, warm, extend=extend)
            b.set_input(text, off); b.run(tok, 0)
            res, tot = b.result(), b.totals()
            exp = oracle_doc(om, doc, 0); got = res.doc(1)
            ok = got["status"] == exp["status"] and np.array_equal(got["tok_bstart"], exp["tok_bstart"]) and np.array_equal(got["tok_bend"], exp["tok_bend"])
            print(len(doc), doc[:6], "chunk", chunk, "warm", warm, "extend", extend, "rounds", tot["repair_rounds"], "OK" if ok else
                  "BAD got %s/%s exp %s/%s st %d/%d" % (got["tok_bstart"][:4], got["tok_bend"][:4], exp["tok_bstart"][:4], exp["tok_bend"][:4], got["status"], exp["status"]))
