#!/usr/bin/env python3
"""Soak parity run: many seeds x corpora x walk configurations, every document of every batch
against the oracle (bit exact).  Not part of pytest (minutes); run on an MI355X:
    python scripts/soak.py [seeds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datok_amd  # noqa: E402
from datok_amd import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity import assert_batch_equals_oracle  # noqa: E402

M = os.path.join(ROOT, "tests", "golden", "models")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000

ALPHA = list(" \n\t.,;:!?'\"()-@/&%abcdefgABCDE0123äöüß„“»«…€<>=") + ["\x04"]


def random_docs(rng, n):
    docs = []
    for _ in range(n):
        k = int(rng.integers(0, 600))
        kind = rng.integers(0, 6)
        if kind == 0:     # raw bytes, mostly invalid UTF-8
            docs.append(bytes(rng.integers(0, 256, size=k, dtype=np.uint8)))
        elif kind == 4:   # runs of one character class, tokens up to and beyond the 1024-rune window
            parts = []
            while sum(map(len, parts)) < 3 * k:
                c = str(rng.choice([".", "!", "a ", "1.", "\n", "x", "-", "ab", "\u00e4", "\u201e", ". ", "?!"]))
                parts.append(c * int(rng.choice([1, 2, 5, 30, 64, 129, 400, 1020, 1100])))
                parts.append(str(rng.choice(["", " ", "\n", " "])))
            docs.append("".join(parts).encode())
        elif kind == 1:   # long tokens around the 31-byte length field
            parts = []
            while sum(map(len, parts)) < k:
                parts.append("x" * int(rng.integers(1, 70)) + str(rng.choice([" ", ". ", "\n", ", ", "! ", " - "])))
            docs.append("".join(parts).encode())
        else:
            docs.append("".join(ALPHA[int(i)] for i in rng.integers(0, len(ALPHA), size=k)).encode())
    return corpus.concat_docs(docs)


def long_token_docs(rng, n):
    """German text with blank-free tokens of 40..900 bytes (URLs, letter runs, digits): longer than the warm-up,
    many longer than several chunks."""
    text, off = corpus.german_docs(n, 4096, seed=int(rng.integers(0, 1 << 30)))
    raw = bytearray(text.tobytes())
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789/_-%.,:;?&=+#", dtype=np.uint8)
    for d in range(n):
        p = d * 4096 + int(rng.integers(0, 1500))
        for _ in range(int(rng.integers(1, 6))):
            q = raw.find(b" ", p)
            L = int(rng.choice([40, 60, 90, 130, 200, 330, 500, 900]))
            if q < 0 or q + L + 2 >= (d + 1) * 4096 - 8:
                break
            kind = int(rng.integers(0, 3))
            body = bytes(rng.choice(alpha, size=L)) if kind == 0 else (b"x" * L if kind == 1 else b"0123456789" * (L // 10 + 1))
            raw[q + 1:q + 1 + L] = ((b"https://www.example.org/" if kind == 0 else b"") + body)[:L]
            raw[q + 1 + L] = 0x20
            p = q + L + int(rng.integers(20, 600))
    return np.frombuffer(bytes(raw), dtype=np.uint8).copy(), off


def long_docs(rng, n):
    out = []
    for _ in range(n):
        text, off = random_docs(rng, int(rng.integers(40, 200)))
        raw = bytearray(text.tobytes())
        if rng.integers(0, 3) == 0:   # blank-free tokens longer than chunks inside a long document
            g, _ = long_token_docs(rng, int(rng.integers(2, 8)))
            cut = int(rng.integers(0, len(raw) + 1))
            raw[cut:cut] = g.tobytes() + b" "
        if rng.integers(0, 2):
            g, _ = corpus.german_docs(int(rng.integers(4, 24)), 4096, seed=int(rng.integers(0, 1 << 30)))
            cut = int(rng.integers(0, len(raw) + 1))
            raw[cut:cut] = g.tobytes() + b" \x04\n"
        out.append(bytes(raw))
    return corpus.concat_docs(out)


models = {name: (datok_amd.load_tokenizer_file(os.path.join(M, name)), O.Model(os.path.join(M, name)))
          for name in ("tokenizer_de.matok", "tokenizer_en.matok", "tokenizer_de.datok", "clitic_test.matok",
                       "simpletok.matok", "simpletok.datok", "bauamt.fst", "wahlamt.fst", "ignorable_mcs.fst")}
t0 = time.time()
total_docs = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    cases = [("tokenizer_de.matok", corpus.german_docs(int(rng.integers(64, 600)), int(rng.choice([512, 4096, 9000])), seed=seed)),
             ("tokenizer_en.matok", corpus.english_zipf_docs(int(rng.integers(64, 400)), seed=seed, max_bytes=16384)),
             ("tokenizer_de.datok", corpus.german_docs(128, 4096, seed=seed + 7)),
             (str(rng.choice(list(models))), random_docs(rng, 300)),
             # a few long documents full of EOT texts, blanks and odd bytes: many compaction segments
             (str(rng.choice(["tokenizer_de.matok", "tokenizer_en.matok", "clitic_test.matok", "tokenizer_de.datok"])),
              long_docs(rng, int(rng.integers(1, 4)))),
             # blank-free tokens longer than warm-up and chunks (empty lanes, repairs behind them)
             (str(rng.choice(["tokenizer_de.matok", "tokenizer_de.datok", "tokenizer_en.matok"])), long_token_docs(rng, int(rng.integers(8, 48)))),
             # the double array on long documents without EOT (segments) 
             ("tokenizer_de.datok", corpus.german_docs(int(rng.integers(2, 6)), int(rng.choice([20000, 70000])), seed=seed + 3))]
    for name, (text, off) in cases:
        tok, om = models[name]
        chunk, warm = [(None, 48), (0, 48), (64, 48), (128, 16), (256, 48), (48, 0), (1024, 48), (64, 0), (128, 48)][int(rng.integers(0, 9))]
        extend = [None, 0, 240, 16][int(rng.integers(0, 4))]
        flags = int(rng.choice([0, 16]))
        with datok_amd.Batch(max(len(text), 1), len(off) - 1) as b:
            if chunk is not None:
                b.set_chunking(chunk, warm, extend=extend)
            b.set_input(text, off)
            b.run(tok, flags)
            res, tot = b.result(), b.totals()
            assert not any(int(x) & datok_amd.ST_IRREGULAR for x in res.status), (seed, name)  # the exact pass took them
            keep = list(range(len(off) - 1))
            try:
                assert_batch_equals_oracle(om, res, text, off, flags, docs=keep)
            except AssertionError:  # keep the failing batch for a closer look
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), text=text, off=off, model=name, flags=flags,
                         chunk=-1 if chunk is None else chunk, warm=warm, extend=-1 if extend is None else extend, seed=seed)
                raise
            # and the rendered SIMPLE stream of a sample
            data, o = b.render(3 | flags)
            raw = text.tobytes()
            for d in keep[::max(1, len(keep) // 40)]:
                exp, est = om.transduce(raw[int(off[d]):int(off[d + 1])], 3 | flags)
                if est == 0 and not (int(res.status[d]) & ~datok_amd.ST_EMPTY_TEXT):
                    if data[int(o[d]):int(o[d + 1])] != exp:
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                        np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), text=text, off=off, model=name, flags=flags,
                                 chunk=-1 if chunk is None else chunk, warm=warm, extend=-1 if extend is None else extend,
                                 seed=seed, doc=d, got=np.frombuffer(data[int(o[d]):int(o[d + 1])], dtype=np.uint8),
                                 exp=np.frombuffer(exp, dtype=np.uint8))
                        raise AssertionError(("rendered text differs", seed, name, d))
            total_docs += len(keep)
    if (seed - first) % 5 == 4:
        print("seed %d ok: %d documents checked, %.0f s" % (seed, total_docs, time.time() - t0), flush=True)
print("SOAK OK: %d seeds, %d documents, %.0f s" % (n_seeds, total_docs, time.time() - t0))
