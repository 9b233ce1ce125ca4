"""Compares a GPU BatchResult with the CPU oracle, document by document (bit exact)."""
import numpy as np

FIELDS = ("tok_rstart", "tok_rend", "tok_bstart", "tok_bend", "sent", "text_tok_end", "text_sent_end")


def oracle_doc(omodel, doc: bytes, flags=0):
    r = omodel.transduce_doc(doc, flags)
    return {f: getattr(r, f) for f in FIELDS} | {"status": r.status}


def assert_batch_equals_oracle(omodel, res, text: np.ndarray, doc_off: np.ndarray, flags=0,
                               docs=None, allow_status=0, skip_status=0, fields=FIELDS):
    """res: datok_amd.BatchResult. docs: iterable of doc ids to check (default all).

    Documents whose oracle status is non-zero are out of contract (the reference
    panics): for those only the status bits are compared."""
    n_docs = len(doc_off) - 1
    ids = range(n_docs) if docs is None else docs
    raw = text.tobytes()
    checked = 0
    for d in ids:
        a, b = int(doc_off[d]), int(doc_off[d + 1])
        exp = oracle_doc(omodel, raw[a:b], flags)
        got = res.doc(d)
        if got["status"] & skip_status:   # (the caller accepts that the library declares such a document out of contract)
            continue
        if exp["status"] & 1:
            # WINDOW_OVERFLOW: the reference dies at the first overflow (matrix.go:365,406 index panic); the GPU path
            # stops a walk whose window has overflowed for certain (more bytes than 1024 runes can have) and closes
            # the document there, the oracle runs on -- what either of them flags behind that point means nothing
            assert got["status"] & 1, (d, got["status"], exp["status"], raw[a:b][:80])
            continue
        assert (got["status"] & ~allow_status) == (exp["status"] & ~allow_status), (d, got["status"], exp["status"], raw[a:b][:80])
        if exp["status"]:
            continue
        for f in fields:
            g, e = np.asarray(got[f]).astype(np.int64), np.asarray(exp[f]).astype(np.int64)
            if g.shape != e.shape or not np.array_equal(g, e):
                k = 0
                while k < min(len(g), len(e)) and g[k] == e[k]:
                    k += 1
                raise AssertionError("doc %d field %s differs at %d: gpu %s oracle %s (len %d/%d) text=%r" % (
                    d, f, k, g[max(0, k - 2):k + 3], e[max(0, k - 2):k + 3], len(g), len(e), raw[a:b][:120]))
        checked += 1
    return checked
