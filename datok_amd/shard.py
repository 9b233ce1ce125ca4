"""Multi-GPU sharding of a corpus (BASELINE.json configs[4]).

Documents are independent units of the hot path (the walk state is per call,
matrix.go:349-381), so a corpus shards by contiguous document ranges balanced
by bytes with no data-path collective.  The only exchange is the gather of the
per-shard offset arrays to rank 0 (variable length: RCCL has no gatherv, so
counts travel first and the payload goes as point-to-point sends; on gloo the
same code runs on CPU tensors).
"""
import numpy as np


def shard_ranges(doc_off: np.ndarray, world: int):
    """Contiguous document ranges [lo, hi) per rank, cut at the n/world byte marks."""
    doc_off = np.asarray(doc_off, dtype=np.uint64)
    n_docs = len(doc_off) - 1
    total = int(doc_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        # first document whose start is at or behind the mark
        d = int(np.searchsorted(doc_off[:-1], np.uint64(target), side="left"))
        cuts.append(min(max(d, cuts[-1]), n_docs))
    cuts.append(n_docs)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_input(text: np.ndarray, doc_off: np.ndarray, lo: int, hi: int):
    """The text slice and rebased offsets of documents [lo, hi)."""
    a, b = int(doc_off[lo]), int(doc_off[hi])
    return text[a:b], (doc_off[lo:hi + 1] - doc_off[lo]).astype(np.uint64)


def gather_offsets(arrays, rank, world, dist, device=None):
    """Gathers one int32 array per name from every rank to rank 0, in rank order.

    arrays: dict name -> 1-D torch.int32 tensor (on `device` for RCCL, CPU for gloo).
    Returns on rank 0 a dict name -> list of tensors (one per rank); None elsewhere.
    """
    import torch
    names = sorted(arrays)
    counts = torch.tensor([int(arrays[n].numel()) for n in names], dtype=torch.int64, device=device)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    allc = torch.stack(allc).cpu().numpy()
    if rank == 0:
        out = {n: [arrays[n]] for n in names}
        reqs = []
        for r in range(1, world):
            for i, n in enumerate(names):
                buf = torch.empty(int(allc[r, i]), dtype=torch.int32, device=device)
                out[n].append(buf)
                if buf.numel():
                    reqs.append(dist.irecv(buf, src=r))
        for q in reqs:
            q.wait()
        return out
    for n in names:
        if arrays[n].numel():
            dist.send(arrays[n].contiguous(), dst=0)
    return None
