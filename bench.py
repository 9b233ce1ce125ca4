#!/usr/bin/env python3
"""bench.py -- input MB/s tokenized, tokenizer_de.matok (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (symbolise -> walk -> compact, i.e. one
data-parallel TransduceTokenWriter) over one batch of synthetic German
documents that is already resident in HBM.  N = 1 runs BASELINE.json
configs[1] (4096 equal-length 4 KiB documents); N > 1 is launched by
torch.distributed.run, one rank per GPU, each rank walking its own shard of
the same shape (weak scaling, no data-path collective: documents are
independent).  After the timed region the per-shard offset arrays are gathered
to rank 0 over RCCL once, outside the clock, and reported as gather_ms.

Rank 0 prints one JSON line.  `value` is whole-job MB/s (1e6 bytes/s).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = os.path.join(ROOT, "tests", "golden", "models", "tokenizer_de.matok")
# the measured path produces the offset arrays (the north star); the bookkeeping that only the device
# renderer of the writer's text output needs is not requested (datok_gpu.h: DTK_OFFSETS_ONLY)
RUN_FLAGS = 256
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s, 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=4096, help="documents per GPU")
    ap.add_argument("--doc-bytes", type=int, default=4096)
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--chunk", type=int, default=-1, help="-1 automatic, 0 one lane per document, else bytes")
    ap.add_argument("--warm", type=int, default=16)
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight: consecutive steps alternate between this many dtk_batch objects "
                         "(each with its own HIP stream and buffers) over the same resident input")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--parity-docs", type=int, default=256, help="documents checked against the oracle before timing")
    return ap.parse_args()


def walk_kernel_name(total_bytes):
    """The dominant kernel as rocprofv3 prints it: start record + chunk walk in one launch (k_spec_both) unless
    DATOK_SPLIT_START=1; its last template argument says whether event bytes go through LDS lists
    (batches of 48 MiB and more, dtk_host.cpp spec_args)."""
    if os.environ.get("DATOK_SPLIT_START", "0") not in ("", "0"):
        base = "k_spec_walk"
    else:
        base = "k_spec_both"
    e = os.environ.get("DATOK_EV_LISTS")
    lists = (e not in ("", "0")) if e is not None else total_bytes >= int(os.environ.get("DATOK_EV_LISTS_MIN", 48 << 20))
    return base + "<%s, true, " + ("true" if lists else "false") + ">"


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`"
                     % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import datok_amd
    from datok_amd import corpus

    datok_amd.lib().dtk_set_device(local_rank)
    tok = datok_amd.load_tokenizer_file(args.model)
    if tok is None:
        sys.exit("bench.py: cannot load " + args.model)

    # ---- this rank's shard: same generator, rank-dependent seed
    seed = 2 if world == 1 else 5 * 1000 + rank
    text, doc_off = corpus.german_docs(args.docs, args.doc_bytes, seed=seed)
    total = int(doc_off[-1])
    t_text = torch.from_numpy(text).to(dev)
    t_off = torch.from_numpy(doc_off.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    batches = []
    for _ in range(max(1, args.streams)):
        bb = datok_amd.Batch(total, args.docs)
        bb.set_chunking(datok_amd.Batch.AUTO_CHUNK if args.chunk < 0 else args.chunk, args.warm)
        bb.set_input_device(t_text.data_ptr(), t_off.data_ptr(), args.docs, total,
                            keep=(t_text, t_off), doc_off_host=doc_off)
        batches.append(bb)
    batch = batches[0]

    # ---- parity gate (oracle is the checker, never the thing measured)
    batch.run(tok, RUN_FLAGS)
    tot = batch.totals()
    if rank == 0 and args.parity_docs:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from parity import assert_batch_equals_oracle
        from oracle import oracle as O
        om = O.Model(args.model)
        res = batch.result()
        step = max(1, args.docs // args.parity_docs)
        n = assert_batch_equals_oracle(om, res, text, doc_off, docs=range(0, args.docs, step))
        assert tot["n_flagged"] == 0 and n > 0
        del res

    # ---- warmup
    for i in range(args.warmup):
        batches[i % len(batches)].run(tok, RUN_FLAGS)
    for bb in batches:
        bb.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps
    batch.set_profiling(False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        batches[i % len(batches)].run(tok, RUN_FLAGS)
    for bb in batches:
        bb.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- per-kernel time of the dominant kernel (the walk): HIP events on each batch's own
    #      stream, in the same regime as the timed region (same number of batches in flight,
    #      no host synchronisation between steps); the events of the last run of every batch
    #      are read after the region, and the region is repeated a few times
    for bb in batches:
        bb.set_profiling(True)
    stage_sum, n_samples = {}, 0
    for rep in range(4):
        for i in range(max(args.steps // 4, 2 * len(batches))):
            batches[i % len(batches)].run(tok, RUN_FLAGS)
        for bb in batches:
            for k, v in bb.stage_ms().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            n_samples += 1
    stage_avg = {k: v / n_samples for k, v in stage_sum.items()}
    for bb in batches:
        bb.set_profiling(False)

    # ---- offset gather to rank 0 over RCCL (config 5's exchange), outside the clock
    gather_ms, gather_hung = None, False
    if world > 1:
        from datok_amd import shard
        v = batch.result_device()
        ntok, nsent = tot["n_tokens"], tot["n_sent"]

        class _DevI32:  # zero-copy view of a library-owned device array
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, True),
                                                 "version": 2}

        def dev_i32(ptr, n):
            if n == 0:
                return torch.empty(0, dtype=torch.int32, device=dev)
            return torch.as_tensor(_DevI32(ptr, n), device=dev).clone()
        mine = {"tok_rstart": dev_i32(v.tok_rstart, ntok), "tok_rend": dev_i32(v.tok_rend, ntok),
                "sent": dev_i32(v.sent, nsent)}
        torch.cuda.synchronize()
        dist.barrier()
        # The gather runs in a helper thread with a deadline: it is outside the timed region, and a
        # point-to-point problem between two GPUs must not cost the measured line.
        import threading
        box = {}

        def _gather():
            try:
                torch.cuda.set_device(dev)
                g0 = time.perf_counter()
                got = shard.gather_offsets(mine, rank, world, dist, device=dev)
                torch.cuda.synchronize()
                box["ms"] = (time.perf_counter() - g0) * 1e3
                if rank == 0:
                    assert sum(int(t.numel()) for t in got["tok_rstart"]) >= ntok
            except Exception as e:  # report, do not lose the run
                box["err"] = repr(e)
        th = threading.Thread(target=_gather, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("DATOK_GATHER_TIMEOUT", "90")))
        gather_hung = th.is_alive()
        if gather_hung or "err" in box:
            print("bench.py: offset gather %s on rank %d" % ("timed out" if gather_hung else "failed: " + box["err"], rank),
                  file=sys.stderr)
        gather_ms = box.get("ms")
        # no barrier behind the gather: a rank whose peer is stuck must still get to its output

    # ---- CPU baseline: the C restatement of the Go algorithm, rank 0, N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        om = O.Model(args.model)
        cores = os.cpu_count() or 1
        om.count_batch(text[: 64 * args.doc_bytes], doc_off[:65], 1)
        t1 = time.perf_counter()
        om.count_batch(text, doc_off, 1)
        one = total / (time.perf_counter() - t1) / 1e6
        reps, spent = 0, 0.0
        t1 = time.perf_counter()
        while spent < args.cpu_seconds:
            om.count_batch(text, doc_off, cores)
            reps += 1
            spent = time.perf_counter() - t1
        cpu = {"value": round(reps * total / spent / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
               "sample": "the same %d x %d B batch, %d passes, %d threads (oracle/datok_oracle.c, C restatement "
                         "of matrix.go:348-698, counting sink)" % (args.docs, args.doc_bytes, reps, cores),
               "single_thread_MBps": round(one, 1),
               "reference_published_MBps_per_core": 24.8}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * total * args.steps / elapsed / 1e6
        # algorithmic bytes of one launch: input + doc offsets + i32 offsets out + counts
        b_alg = total + 4 * (args.docs + 1) + 4 * (2 * tot["n_tokens"] + tot["n_sent"]) + 8 * args.docs
        walk_s = stage_avg["walk"] * 1e-3
        achieved = b_alg / walk_s / 1e9
        split = os.environ.get("DATOK_SPLIT_START", "0") not in ("", "0")
        both_s = (stage_avg["walk"] + (stage_avg["spec_start"] if split else 0.0)) * 1e-3
        # HBM bytes of that kernel from the last committed rocprofv3 --pmc passes (profiles/)
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tr.get("docs") == args.docs and tr.get("doc_bytes") == args.doc_bytes and \
                    tr.get("chunk_bytes") == tot["chunk_bytes"]:
                traffic = tr["walk_kernel_hbm_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "input MB/s tokenized, tokenizer_de.matok",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "tokenizer_de.matok, %d equal-length %d B synthetic German docs per GPU "
                                   "(BASELINE.json configs[1])" % (args.docs, args.doc_bytes),
                       "docs_per_gpu": args.docs, "doc_bytes": args.doc_bytes,
                       "parallelism": "documents sharded over %d GPU(s), no data-path collective" % world,
                       "batches_in_flight": len(batches)},
            "roofline": {"bound": "hbm",
                         "kernel": (walk_kernel_name(total) if tot["chunk_bytes"] else "k_walk_doc<%s, true>") % (
                             ("MatrixLeanTrans" if not tok.info["unknown_used"] else "MatrixFusedTrans")
                             if tok.info["entry_bytes"] == 4 and tok.info["state_count"] < 32767
                             else "MatrixTrans<u%d>" % (8 * tok.info["entry_bytes"])),
                         "achieved": round(achieved, 2),
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6),
                         "traffic": traffic, "algorithmic_bytes": int(b_alg),
                         "kernel_ms": round(stage_avg["walk"], 4),
                         "lookups_per_launch": int(tot["walk_steps"]),
                         "Glookups_per_s_start_plus_walk": round(tot["walk_steps"] / both_s / 1e9, 3)},
            "stages_ms": {k: round(v, 4) for k, v in stage_avg.items()},
            "tokens_per_launch": int(tot["n_tokens"]),
            "walk": {"lanes": int(tot["n_lanes"]), "chunk_bytes": int(tot["chunk_bytes"]), "warm_bytes": args.warm,
                     "repair_rounds": int(tot["repair_rounds"])},
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        if gather_hung:
            sys.stdout.flush()
            os._exit(0)  # a stuck point-to-point operation would also block the teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
