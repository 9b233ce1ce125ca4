"""World-size-2 gloo test of the multi-GPU path: documents sharded by
contiguous ranges, no data-path collective, offset arrays gathered to rank 0.

No GPU here: each rank produces its shard's offset arrays with the CPU oracle (standing in
for dtk_batch_run); what is under test is the product's sharding + gather code
(datok_amd/shard.py), which is what bench.py and a multi-GPU caller use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_docs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from datok_amd import corpus, shard
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        text, doc_off = corpus.english_zipf_docs(n_docs, seed=13, max_bytes=4096)   # ragged lengths
        ranges = shard.shard_ranges(doc_off, world)
        lo, hi = ranges[rank]
        stext, soff = shard.shard_input(text, doc_off, lo, hi)
        om = O.Model(os.path.join(ROOT, "tests", "golden", "models", "tokenizer_en.matok"))
        raw = stext.tobytes()
        rs, re_, se = [], [], []
        for d in range(hi - lo):
            r = om.transduce_doc(raw[int(soff[d]):int(soff[d + 1])], 0)
            rs.append(r.tok_rstart); re_.append(r.tok_rend); se.append(r.sent)
        cat = lambda xs: torch.from_numpy(np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32))
        got = shard.gather_offsets({"tok_rstart": cat(rs), "tok_rend": cat(re_), "sent": cat(se)},
                                   rank, world, dist)
        if rank == 0:
            # reference: the whole corpus in one go
            raw_all = text.tobytes()
            full = {"tok_rstart": [], "tok_rend": [], "sent": []}
            for d in range(n_docs):
                r = om.transduce_doc(raw_all[int(doc_off[d]):int(doc_off[d + 1])], 0)
                full["tok_rstart"].append(r.tok_rstart); full["tok_rend"].append(r.tok_rend)
                full["sent"].append(r.sent)
            ok = True
            for name in full:
                a = np.concatenate([t.numpy() for t in got[name]])
                b = np.concatenate(full[name]).astype(np.int32)
                ok = ok and a.shape == b.shape and np.array_equal(a, b)
            sizes = [int(doc_off[h]) - int(doc_off[l]) for l, h in ranges]
            q.put((ok, ranges, sizes))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_corpus_gathers_to_the_unsharded_result(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_docs = 96
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_docs, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, ranges, sizes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert ranges[0][0] == 0 and ranges[-1][1] == n_docs
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert max(sizes) - min(sizes) <= 2 * 4096   # balanced by bytes up to one document


def test_shard_ranges_edge_cases():
    from datok_amd import shard
    off = np.array([0, 10, 10, 10, 50, 100], dtype=np.uint64)
    r = shard.shard_ranges(off, 2)
    assert r[0][0] == 0 and r[-1][1] == 5 and r[0][1] == r[1][0]
    assert shard.shard_ranges(off, 1) == [(0, 5)]
    r8 = shard.shard_ranges(off, 8)                     # more ranks than documents
    assert len(r8) == 8 and r8[-1][1] == 5 and all(a <= b for a, b in r8)
