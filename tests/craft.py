"""Hand-made tokenizers for the constructs no shipped model exercises (test infrastructure).

The reference's two encodings are written from a small arc table: a `.matok` image as
MatrixTokenizer.WriteTo lays it out (matrix.go:126-210) and a `.datok` image as DaTokenizer.WriteTo does
(datok.go:502-596), with every arc slot "separate" and pointing to its target's representative
(datok.go:266-325) -- a valid double array that needs no Mizobuchi packing.  Both are parsed by the oracle
and by the product's loaders like any file of the reference.

REVISIT: a word may continue over an EOT ("a\\x04b" is one token).  On "a\\x04a" the double array consumes the
EOT (SentenceEnd + TextEnd, window kept, datok.go:1019-1030), fails on the second "a", backtracks to the epsilon
state remembered BEFORE the EOT (datok.go:916-926), flushes "a" and reads the same EOT again: the first
SentenceEnd/TextEnd pair precedes a Token that ends before it.  TRIPLE: three epsilon arcs in a row on an empty
token: three SentenceEnd calls at one cursor (matrix.go:573-576 has no limit).
"""
import gzip
import struct

FIRSTBIT = 1 << 31
EOT = "\x04"
SIGMA = ["\0", "\0", "\0", "\0", "a", "b", EOT, " ", ".", "\n"]   # 1 epsilon, 2 unknown, 3 identity
EPS, UNKNOWN, IDENTITY = 1, 2, 3
A, B, E, SP, DOT, NL = 4, 5, 6, 7, 8, 9


def _automaton(triple):
    """state -> {symbol: (target, nontoken)}; state 1 is the start state."""
    blank = {SP: True, NL: True, E: True}

    def idle(me):
        d = {A: (2, False), B: (2, False), DOT: (5, False)}
        d.update({s: (me, True) for s in blank})
        return d
    arcs = {1: idle(1), 3: idle(3),
            2: {A: (2, False), B: (2, False), EPS: (3, False), E: (4, False)},
            4: {B: (2, False)},          # the word goes on behind an EOT only with "b"
            5: {EPS: (6, False)}}        # "." is a token of its own ...
    if triple:                           # ... followed by one or three sentence ends
        arcs.update({6: {EPS: (7, False)}, 7: {EPS: (8, False)}, 8: {EPS: (3, False)}})
    else:
        arcs[6] = {EPS: (3, False)}
    return arcs


def _sigma_bytes(sigma=None):
    return "".join(sigma or SIGMA).encode("utf-8")


def matok(triple=False) -> bytes:
    return matok_from(_automaton(triple))


def matok_from(arcs, sigma=None) -> bytes:
    """The `.matok` image of an arc table  state -> {symbol: (target, nontoken)}  (states 1..n, 1 = start);
    sigma: SIGMA with more characters appended (symbol = index)."""
    n, s = max(max(arcs), max(to for row in arcs.values() for to, _ in row.values())), len(sigma or SIGMA)
    arr = [0] * ((n + 1) * s)
    for t, row in arcs.items():
        for a, (to, nontoken) in row.items():
            arr[(a - 1) * n + t] = to | (FIRSTBIT if nontoken else 0)      # matrix.go:85-90
    raw = b"MATOK" + struct.pack("<HHHHIH", 1, EPS, UNKNOWN, IDENTITY, n, s) + _sigma_bytes(sigma) + b"M"
    raw += struct.pack("<%dI" % len(arr), *arr)
    return gzip.compress(raw)


def datok(triple=False) -> bytes:
    return datok_from(_automaton(triple))


def datok_from(arcs, sigma=None, size_cut=0) -> bytes:
    """The `.datok` image of the same kind of arc table (every arc slot "separate").
    size_cut: array[1].check -- the size every transition index is tested against (datok.go:896) -- is set that much
    below the highest slot in use: a hand-made file in which slots behind the size still carry matching check words."""
    n, s = max(max(arcs), max(to for row in arcs.values() for to, _ in row.values())), len(sigma or SIGMA)
    size = n + 1 + (n + 1) * s
    base = [0] * (size + s + 2)
    check = [0] * (size + s + 2)
    used = set(range(0, n + 1))           # 0 unused, 1..n the representatives (index == state)
    top = 0
    for t in sorted(arcs):
        b = n + 1
        while any(b + a in used for a in arcs[t]):
            b += 1
        base[t] = b
        for a, (to, nontoken) in arcs[t].items():
            used.add(b + a)
            base[b + a] = to | FIRSTBIT                                     # separate: move on to the representative
            check[b + a] = t | (FIRSTBIT if nontoken else 0)
            top = max(top, b + a)
    check[1] = max(n + 1, top - size_cut)                                   # datok.go:328-335: the array's size
    pairs = [x for i in range(len(base)) for x in (base[i], check[i])]
    raw = b"DATOK" + struct.pack("<HHHHHHI", 1, EPS, UNKNOWN, IDENTITY, s, s, len(pairs)) + _sigma_bytes(sigma) + b"T"
    raw += struct.pack("<%dI" % len(pairs), *pairs)
    return gzip.compress(raw)


ALPHABET = "aab b \x04.\n"


def documents(rng, n=400, max_len=60):
    docs = [b"a\x04a", b"ab\x04ab a\x04b. a", b"a\x04\x04a", b"a\x04a\x04a\x04a", b"a.\x04a", b"a. b.", b".", b"..", b"a",
            b"", b"\x04", b"a\x04", b"a\x04b\x04a. b\x04\x04a\n", b"ab ab.\nab\x04ab\x04.a", b"a\x04.", b". . .\x04a\x04a"]
    for _ in range(n):
        k = int(rng.integers(0, max_len))
        docs.append("".join(ALPHABET[int(i)] for i in rng.integers(0, len(ALPHABET), size=k)).encode())
    return docs


def random_automaton(rng, max_states=8):
    """A random arc table for matok_from / datok_from (scripts/fuzz_automata.py, tests/test_exact_and_replay.py):
    3-8 states, arcs on the six sigma symbols, now and then on `unknown` / `identity`; epsilon arcs only upwards
    (no cycles: the loaders reject those)."""
    n = int(rng.integers(3, max_states + 1))
    arcs = {}
    for t in range(1, n + 1):
        row = {}
        for a in (A, B, E, SP, DOT, NL):
            if rng.random() < 0.55:
                row[a] = (int(rng.integers(1, n + 1)), bool(rng.random() < (0.6 if a in (SP, NL, E) else 0.1)))
        if t < n and rng.random() < 0.45:
            row[EPS] = (int(rng.integers(t + 1, n + 1)), False)
        if rng.random() < 0.08:
            row[UNKNOWN] = (int(rng.integers(1, n + 1)), False)
        if rng.random() < 0.08:
            row[IDENTITY] = (int(rng.integers(1, n + 1)), bool(rng.random() < 0.3))
        arcs[t] = row
    if not arcs[1]:
        arcs[1][A] = (1, False)
    return arcs


def random_documents(rng, n=160, max_len=90, raw=()):
    """`raw`: extra byte strings drawn like letters (invalid UTF-8, long runes)."""
    alpha = ALPHABET + ("x\u00e4" if rng.random() < 0.5 else "")   # x, a-umlaut: not in the sigma (identity / unknown)
    letters = [c.encode() for c in alpha] + list(raw)
    docs = [b"", b"a", b"\x04", b" ", b"a\x04a", b"a. b.\x04\n\na"]
    for _ in range(n):
        k = int(rng.integers(0, max_len))
        docs.append(b"".join(letters[int(i)] for i in rng.integers(0, len(letters), size=k)))
    long_ = [b"".join(docs[int(i)] for i in rng.integers(0, len(docs), size=30)) for _ in range(6)]
    return docs + long_
