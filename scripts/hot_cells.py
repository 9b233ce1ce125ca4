#!/usr/bin/env python3
"""CPU study for the LDS hot sub-table: which (state, symbol) cells does the fused walk ask for?

Walks text with the matrix of a .matok file exactly as matrix.go:384-635 does, counts the lookups as the device's
fused table sees them (a failed lookup whose epsilon backtrack goes to the very state it failed in is ONE fused
lookup), ranks states on a TRAINING text and reports which share of a TEST text's lookups falls into the dense
rectangle  [state rank < T] x [column < C]  of the device layout (epsilon states and the others ranked apart,
as the layout keeps them apart).

usage: hot_cells.py [model.matok]"""
import gzip
import os
import struct
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from datok_amd import corpus  # noqa: E402

FIRSTBIT = 0x80000000
ORDER = [' ', 'e', 'n', 'i', 's', 'r', 'a', 't', 'd', 'h', 'u', 'l', 'c', 'g', 'm', 'o', 'b', 'w', 'f',
         'k', 'z', 'p', 'v', '.', ',', '\n', '\xfc', '\xe4', '\xf6', '\xdf', 'j', 'y', 'x', 'q', '-', '\'', '"',
         '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'S', 'D', 'A', 'E', 'B', 'M', 'K', 'W', 'G',
         'H', 'T', 'I', 'P', 'L', 'R', 'F', 'N', 'V', 'Z', 'U', 'O', 'J', 'C', ':', ';', '?', '!', '(', ')',
         '/', '\t', '\r']


def load(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:5] == b"MATOK"
    ver, eps, unk, ident, n, s = struct.unpack_from("<HHHHIH", raw, 5)
    off = 19
    sigma = {}
    txt = raw[off:]
    i = 0
    for a in range(s):
        c = txt[i:i + 4].decode("utf-8", "ignore")[:1] if txt[i] else "\0"
        w = len(c.encode()) if txt[i] else 1
        if txt[i]:
            sigma[c] = a
        i += w
    off += i
    assert raw[off:off + 1] == b"M"
    arr = np.frombuffer(raw, dtype="<u4", count=(n + 1) * s, offset=off + 1)
    return dict(eps=eps, unk=unk, ident=ident, N=n, S=s, sigma=sigma, arr=arr)


def walk(m, text, counts):
    """matrix.go:384-635 on one document (no EOT, no unknown arcs); counts[(t, a)] += 1 per lookup of the device's
    fused table (dtk_host.cpp layout_matrix: where (t, a) has no arc, t has an epsilon arc to e and (e, a) has one,
    one cell stands for fail + epsilon step + rune)."""
    arr, N, eps, ident = m["arr"], m["N"], m["eps"], m["ident"]
    sigma = m["sigma"]
    ascii_ = [sigma.get(chr(c), ident) for c in range(256)]
    n = len(text)
    t, p, tp = 1, 0, 0
    eps_t, eps_p = 0, 0
    newchar = True
    a, t0 = 0, 1
    while True:
        if newchar:
            if p >= n:
                break  # (the EOF drain's few lookups do not matter for the statistic)
            o = ord(text[p])
            a = ascii_[o] if o < 256 else sigma.get(text[p], ident)
            t0 = t
            if arr[(eps - 1) * N + t0] != 0:
                eps_t, eps_p = t0, p
        x = int(arr[(a - 1) * N + t0]) if a else 0
        tgt = x & ~FIRSTBIT
        counts[(t0, a)] += 1
        if tgt == 0:
            if a != eps and eps_t != 0:
                if eps_t == t0 and eps_p == p:
                    e = int(arr[(eps - 1) * N + t0]) & ~FIRSTBIT
                    x2 = int(arr[(a - 1) * N + e])
                    if x2 & ~FIRSTBIT:  # a fused cell: this one lookup did it all
                        if p > tp:
                            tp = p
                        eps_t = 0
                        if arr[(eps - 1) * N + e] != 0:
                            eps_t, eps_p = e, p
                        if p == tp and (x2 & FIRSTBIT):
                            tp = p + 1
                        p += 1
                        t = x2 & ~FIRSTBIT
                        newchar = True
                        continue
                t0 = eps_t; eps_t = 0; p = eps_p; a = eps; newchar = False
                continue
            if p <= tp:  # hard fail, matrix.go:499-552
                p += 1
            tp = p
            t = 1; eps_t = 0; newchar = True
            continue
        if a == eps:
            if p > tp:
                tp = p
        else:
            if p == tp and (x & FIRSTBIT):
                tp = p + 1
            p += 1
        t = tgt
        newchar = True


def columns(m):
    col = {}
    nxt = 1
    for ch in ORDER:
        a = m["sigma"].get(ch)
        if a and a not in col:
            col[a] = nxt; nxt += 1
    for a in range(1, m["S"]):
        if a not in col:
            col[a] = nxt; nxt += 1
    return col


def profile(m, docs, limit):
    counts = Counter()
    done = 0
    for d in docs:
        walk(m, d, counts)
        done += len(d)
        if done >= limit:
            break
    return counts


def docs_of(gen, n_docs, **kw):
    text, off = gen(n_docs, 4096, **kw)
    b = text.tobytes()
    return [b[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(n_docs)]


SAMPLE = None


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok")
    m = load(path)
    col = columns(m)
    has_eps = lambda t: m["arr"][(m["eps"] - 1) * m["N"] + t] != 0
    sets = {
        "bench": docs_of(corpus.german_docs, 64, seed=2),
        "bench7": docs_of(corpus.german_docs, 64, seed=7),
        "rich": docs_of(corpus.german_rich_docs, 64, seed=2),
    }
    if SAMPLE:
        sets["sample"] = [SAMPLE]
    prof = {k: profile(m, v, 150000) for k, v in sets.items()}
    for k, c in prof.items():
        tot = sum(c.values())
        states = Counter()
        for (t, a), v in c.items():
            states[t] += v
        print("%s: %d lookups, %d distinct cells, %d distinct states; top-1024 cells %.2f%%" % (
            k, tot, len(c), len(states), 100.0 * sum(v for _, v in c.most_common(1024)) / tot))
    for train in prof:
        st = Counter()
        for (t, a), v in prof[train].items():
            st[t] += v
        rank_e, rank_n = {}, {}
        for t, _ in st.most_common():
            if has_eps(t):
                rank_e[t] = len(rank_e)
            else:
                rank_n[t] = len(rank_n)
        for test in prof:
            tot = sum(prof[test].values())
            line = []
            for (Te, Tn, C) in ((64, 64, 64), (96, 32, 64), (128, 64, 48), (128, 128, 32), (192, 64, 64), (256, 128, 64), (128, 128, 64)):
                hit = 0
                for (t, a), v in prof[test].items():
                    r = rank_e.get(t, 1 << 30) if has_eps(t) else rank_n.get(t, 1 << 30)
                    if r < (Te if has_eps(t) else Tn) and col[a] < C:
                        hit += v
                line.append("%dx%d+%d:%.1f%%(%dKB)" % (Te, C, Tn, 100.0 * hit / tot, (Te + Tn) * C * 4 // 1024))
            print("train %-7s test %-7s %s" % (train, test, " ".join(line)))


if __name__ == "__main__":
    main()
