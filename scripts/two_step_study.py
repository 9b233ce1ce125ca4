#!/usr/bin/env python3
"""CPU study: how many lookups of the fused walk could take TWO runes at once?

A lookup pairs with its successor if both are plain arcs that consume a rune (no fused cell, no epsilon step, no
fail) -- then nothing happens between them but the epsilon slot.  Greedy pairing from the left, per 128-byte chunk
lane; reported: the share of paired lookups, the iterations a wave of 64 lanes would need (its slowest lane) with
and without pairing, and how many distinct (state, rune, rune) triples a table must hold for 90 / 95 / 99 % of the
pairs.      usage: two_step_study.py [model.matok] [docs]"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from datok_amd import corpus  # noqa: E402
import hot_cells as H  # noqa: E402

FIRSTBIT = H.FIRSTBIT


def walk_log(m, text):
    """The walk of hot_cells.walk; returns a list of (position, state, symbol, plain) per fused lookup."""
    arr, N, eps, ident = m["arr"], m["N"], m["eps"], m["ident"]
    sigma = m["sigma"]
    ascii_ = [sigma.get(chr(c), ident) for c in range(256)]
    n = len(text)
    t, p, tp = 1, 0, 0
    eps_t, eps_p = 0, 0
    newchar = True
    a, t0 = 0, 1
    log = []
    while True:
        if newchar:
            if p >= n:
                break
            o = ord(text[p])
            a = ascii_[o] if o < 256 else sigma.get(text[p], ident)
            t0 = t
            if arr[(eps - 1) * N + t0] != 0:
                eps_t, eps_p = t0, p
        x = int(arr[(a - 1) * N + t0]) if a else 0
        tgt = x & ~FIRSTBIT
        if tgt == 0:
            log.append((p, t0, a, False))
            if a != eps and eps_t != 0:
                if eps_t == t0 and eps_p == p:
                    e = int(arr[(eps - 1) * N + t0]) & ~FIRSTBIT
                    x2 = int(arr[(a - 1) * N + e])
                    if x2 & ~FIRSTBIT:
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
            if p <= tp:
                p += 1
            tp = p
            t = 1; eps_t = 0; newchar = True
            continue
        if a == eps:
            log.append((p, t0, a, False))
            if p > tp:
                tp = p
        else:
            log.append((p, t0, a, True))
            if p == tp and (x & FIRSTBIT):
                tp = p + 1
            p += 1
        t = tgt
        newchar = True
    return log


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok")
    n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    m = H.load(model)
    for name, gen in (("bench corpus (german_docs)", corpus.german_docs), ("robustness corpus (german_rich_docs)", corpus.german_rich_docs)):
        docs = H.docs_of(gen, n_docs, seed=2)
        triples = Counter()
        lane_it, lane_it2 = [], []
        tot = pairs = 0
        for d in docs:
            log = walk_log(m, d)
            # byte positions: the text is a str; chunk lanes by character position (close enough for a study)
            per1, per2 = Counter(), Counter()
            i = 0
            while i < len(log):
                p, t0, a, plain = log[i]
                lane = p // 128
                per1[lane] += 1
                if plain and i + 1 < len(log) and log[i + 1][3] and log[i + 1][0] == p + 1:
                    triples[(t0, a, log[i + 1][2])] += 1
                    per1[lane] += 1
                    per2[lane] += 1
                    pairs += 1
                    tot += 2
                    i += 2
                else:
                    per2[lane] += 1
                    tot += 1
                    i += 1
            for lane in sorted(per1):
                lane_it.append(per1[lane]); lane_it2.append(per2[lane])
        lane_it, lane_it2 = np.array(lane_it), np.array(lane_it2)
        w = len(lane_it) // 64 * 64
        m1 = lane_it[:w].reshape(-1, 64).max(axis=1).mean()
        m2 = lane_it2[:w].reshape(-1, 64).max(axis=1).mean()
        print("%s: %d lookups, %.1f %% of them in pairs; per lane %.1f -> %.1f lookups (mean), a wave's slowest lane %.1f -> %.1f (%.0f %%)" % (
            name, tot, 200.0 * pairs / tot, lane_it.mean(), lane_it2.mean(), m1, m2, 100.0 * m2 / m1))
        c = np.array(sorted(triples.values(), reverse=True))
        cs = np.cumsum(c) / c.sum()
        print("   distinct (state, rune, rune) triples: %d; for 90 / 95 / 99 %% of the pairs: %d / %d / %d" % (
            len(c), int(np.searchsorted(cs, 0.90)) + 1, int(np.searchsorted(cs, 0.95)) + 1, int(np.searchsorted(cs, 0.99)) + 1))


if __name__ == "__main__":
    main()
