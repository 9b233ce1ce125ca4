#!/usr/bin/env python3
"""bench.py -- input MB/s tokenized, tokenizer_de.matok (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (symbolise -> walk -> compact, i.e. one data-parallel
TransduceTokenWriter) over one batch of synthetic German documents that is already resident in HBM,
INCLUDING the host-side completion of that batch (dtk_batch_totals: the speculation check, the repair
rounds if any, the output-capacity check, the totals) before its buffers are used again.

N = 1 runs BASELINE.json configs[1] (4096 equal-length 4 KiB documents per batch; a few batches in
flight, each with its own input, HIP stream and buffers).  N > 1 is launched by torch.distributed.run,
one rank per GPU, and runs configs[4] as each GPU sees it: the 10 GiB corpus = 2 621 440 documents of
4 KiB sharded 8 ways = 327 680 documents (1.25 GiB) per rank, seed 5000 + rank, one batch (weak scaling,
no data-path collective: documents are independent).  After the timed region the per-shard offset
arrays (tok_rstart, tok_rend, sent) are gathered to rank 0 over RCCL once, outside the clock
(gather_ms).

Rank 0 prints one JSON line.  `value` is whole-job MB/s (1e6 bytes/s) with inputs resident in HBM;
`end_to_end` is the same work fed from pinned host memory through dtk_batch_set_input (PCIe inclusive).
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
# the measured path produces the offset arrays of the north star -- token and sentence boundaries as rune offsets, the
# reference's pos[] / sent[] (token_writer.go:72-81), which is what B_alg counts: 2 ints per token.  The byte offsets
# that slice surfaces and the bookkeeping of the device renderer are not requested (datok_gpu.h: DTK_OFFSETS_ONLY,
# DTK_NO_BYTE_OFFSETS)
RUN_FLAGS = 256 | 512
RUNE_FIELDS = ("tok_rstart", "tok_rend", "sent", "text_tok_end", "text_sent_end")
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s, 6.29 TB/s measured copy)
SHARD_DOCS = 327680  # configs[4]: 10 GiB / 4 KiB / 8 GPUs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--docs", type=int, default=0, help="documents per batch (default: 4096 at N = 1, 327680 at N > 1)")
    ap.add_argument("--doc-bytes", type=int, default=4096)
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--chunk", type=int, default=-1, help="-1 automatic, 0 one lane per document, else bytes")
    ap.add_argument("--warm", type=int, default=8)
    ap.add_argument("--streams", type=int, default=0,
                    help="batches in flight (default 3 at N = 1, 1 at N > 1): consecutive steps alternate between "
                         "this many dtk_batch objects, each with its own input (seeds 2, 3, ...), HIP stream and buffers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--parity-docs", type=int, default=256, help="documents checked against the oracle before timing")
    ap.add_argument("--e2e-only", action="store_true", help="(internal) the PCIe-inclusive block in a process of its own")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive block (profiling runs: no child process)")
    return ap.parse_args()


def walk_kernel_name(info, chunked):
    """The dominant kernel as rocprofv3 prints it: start record + chunk walk in one launch (k_spec_both) unless
    DATOK_SPLIT_START=1; the transition type and the EOT rules follow the model (dtk_model_info)."""
    base = "k_spec_walk" if os.environ.get("DATOK_SPLIT_START", "0") not in ("", "0") else "k_spec_both"
    if not chunked:
        base = "k_walk_doc"
    is_da = info["kind"] == 1
    if is_da and not info["dense_states"]:
        trans = "DaTrans"
    elif info["entry_bytes"] == 4 and (info["dense_states"] or info["state_count"] < 32767):
        trans = "MatrixLeanTrans" if not info["unknown_used"] and info["stream_codes"] else "MatrixFusedTrans"
    else:
        trans = "MatrixTrans<u%d>" % (8 * info["entry_bytes"])
    return "%s<%s, %s>" % (base, trans, "false" if is_da else "true")


def timed_steps(batches, tok, steps, barrier):
    """K steps, round-robin over the batches.  A batch is completed on the host (totals(): speculation check,
    repairs, capacity check) before it is run again and at the end -- all of it inside the clock."""
    barrier()
    t0 = time.perf_counter()
    ran = [False] * len(batches)
    for i in range(steps):
        k = i % len(batches)
        if ran[k]:
            batches[k].totals()
        batches[k].run(tok, RUN_FLAGS)
        ran[k] = True
    for k, bb in enumerate(batches):
        if ran[k]:
            bb.totals()
    elapsed = time.perf_counter() - t0
    barrier()
    return elapsed


def stage_times(batches, tok, steps):
    """Average milliseconds per stage (HIP events on each batch's own stream, recorded around the kernels of
    dtk_batch_run), in the same regime as the timed region: same batches in flight, same completion calls."""
    for bb in batches:
        bb.set_profiling(True)
    acc, n = {}, 0
    for rep in range(4):
        ran = [False] * len(batches)
        for i in range(max(steps // 4, 2 * len(batches))):
            k = i % len(batches)
            if ran[k]:
                batches[k].totals()
            batches[k].run(tok, RUN_FLAGS)
            ran[k] = True
        for bb in batches:
            for name, v in bb.stage_ms().items():
                acc[name] = acc.get(name, 0.0) + v
            n += 1
    for bb in batches:
        bb.set_profiling(False)
    return {k: v / n for k, v in acc.items()}


def e2e_only(args):
    """The PCIe-inclusive block (see main): no torch in this process."""
    from datok_amd import corpus
    import datok_amd
    n_docs = args.docs or 4096
    n_streams = args.streams or 3
    inputs = [corpus.german_docs(n_docs, args.doc_bytes, seed=2 + k) for k in range(n_streams)]
    total = int(inputs[0][1][-1])
    datok_amd.lib().dtk_set_device(0)
    tok = datok_amd.load_tokenizer_file(args.model)
    with datok_amd.Batch(total, n_docs) as b0:
        b0.set_input(*inputs[0])
        b0.run(tok, RUN_FLAGS)
        tot = b0.totals()
    reps = max(1, (24 if total <= (64 << 20) else 3) // len(inputs))
    n_slices = reps * len(inputs)
    pin = datok_amd.PinnedBuffer(total * n_slices)
    for i in range(n_slices):
        pin.array[i * total:(i + 1) * total] = inputs[i % len(inputs)][0]
    big_off = np.concatenate([inputs[i % len(inputs)][1][(1 if i else 0):] + np.uint64(i * total) for i in range(n_slices)])
    # plain upload rate of one batch's text, for reference
    hb = datok_amd.Batch(total, n_docs)
    t0 = time.perf_counter()
    for i in range(6):
        hb.set_input(pin.array[(i % n_slices) * total:(i % n_slices + 1) * total], inputs[0][1])
        hb.sync()
    h2d_ms = (time.perf_counter() - t0) / 6 * 1e3
    hb.close()
    B = datok_amd.Batch

    def run_pipe(fields, depth):
        """The corpus through dtk_pipeline_run, best of three; fields != 0: every slice's selected result arrays are
        brought to page-locked host memory (dtk_pipeline_set_result_fields) and read there by the callback."""
        pipe = datok_amd.Pipeline(total, n_docs, depth=depth)
        pipe.set_chunking(datok_amd.Batch.AUTO_CHUNK if args.chunk < 0 else args.chunk, args.warm)
        if fields:
            pipe.set_result_fields(fields)
        done = [0, 0]

        def on_slice(first, n, bb):
            done[0] += bb.totals()["n_tokens"]
            if fields:
                res = bb.result(copy=False)   # host pointers into the slice's page-locked buffers
                if fields & B.R_TOK_RUNE:
                    done[1] += int(res.tok_rend[-1]) + int(res.tok_off[-1])
                if fields & B.R_TOK_RUNE16:
                    done[1] += int(res.tok_r16[-1, 1]) + int(res.tok_off[-1])
                if fields & B.R_EVENTS:
                    done[1] += int(res.ev_bits[0, 0]) + int(res.doc_tail[-1])
        pipe.run(tok, pin.array, big_off, RUN_FLAGS, on_slice)      # allocations, lane plans, page-locked buffers
        best = 0.0
        for rep in range(3):
            done[0] = 0
            t0 = time.perf_counter()
            pipe.run(tok, pin.array, big_off, RUN_FLAGS, on_slice)
            best = max(best, total * n_slices / (time.perf_counter() - t0) / 1e6)
        assert done[0] >= tot["n_tokens"] * reps
        pipe.close()
        return round(best, 1)
    f_off = B.R_TOK_RUNE | B.R_SENT | B.R_CSR | B.R_STATUS
    f_ev = B.R_EVENTS | B.R_CSR | B.R_STATUS
    f_off16 = B.R_TOK_RUNE16 | B.R_SENT | B.R_CSR | B.R_STATUS
    out_bytes_off16 = 4 * (tot["n_tokens"] + tot["n_sent"]) + 28 * n_docs
    out_bytes_off = 4 * (2 * tot["n_tokens"] + tot["n_sent"]) + 28 * n_docs
    h2d = {"h2d_ms": round(h2d_ms, 4), "h2d_GBps": round(total / h2d_ms / 1e6, 2),
           "end_to_end_MBps": run_pipe(0, 3), "slices": n_slices,
           "with_results_MBps": run_pipe(f_off, 4),
           "with_results_what": "host text in -> host offsets out: tok_rstart, tok_rend, sent, row offsets and status of "
                                "every slice copied to page-locked host memory (%.2f B per input byte) under the next "
                                "slices' upload and walk, read there by the callback; depth 4" % (out_bytes_off / total),
           "with_results_int16_MBps": run_pipe(f_off16, 4),
           "with_results_int16_what": "the same with a token's two rune offsets as the int16 halves of one word "
                                      "(DTK_R_TOK_RUNE16: no document is longer than 32 767 bytes; %.2f B per input byte, "
                                      "less than the upload)" % (out_bytes_off16 / total),
           "with_event_bitmaps_MBps": run_pipe(f_ev, 4),
           "with_event_bitmaps_what": "the same with the five event bitmaps + tail words instead of the offset arrays "
                                      "(0.63 B per input byte: what a TokenWriter closure replay reads)",
           "what": "a corpus of %d slices of one batch each in page-locked host memory -> dtk_pipeline_run (upload, walk, "
                   "completion and download of different slices overlap); end_to_end_MBps: results stay in HBM; measured in a "
                   "child process of bench.py without torch" % n_slices}
    pin.close()
    print(json.dumps(h2d))


def main():
    args = parse()
    if args.e2e_only:
        return e2e_only(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`"
                     % (args.gpus, args.gpus))
        args.gpus = world
    n_docs = args.docs or (4096 if world == 1 else SHARD_DOCS)
    n_streams = args.streams or (3 if world == 1 else 1)

    # ---- inputs first (the sharded generator uses worker processes: before anything touches the GPU)
    from datok_amd import corpus
    inputs, seeds = [], []
    for k in range(n_streams):
        if world == 1 and n_docs <= 65536:
            seeds.append(2 + k)
            inputs.append(corpus.german_docs(n_docs, args.doc_bytes, seed=seeds[-1]))
        else:
            seeds.append(5000 + rank + 100 * k)
            inputs.append(corpus.german_docs_sharded(n_docs, args.doc_bytes, seed=seeds[-1]))
    total = int(inputs[0][1][-1])

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import datok_amd
    datok_amd.lib().dtk_set_device(local_rank)
    tok = datok_amd.load_tokenizer_file(args.model)
    if tok is None:
        sys.exit("bench.py: cannot load " + args.model)

    # ---- resident inputs: one per batch in flight
    resident, batches = [], []
    for text, doc_off in inputs:
        t_text = torch.from_numpy(text).to(dev)
        t_off = torch.from_numpy(doc_off.view(np.int64)).to(dev)
        resident.append((t_text, t_off))
    torch.cuda.synchronize()
    for (text, doc_off), (t_text, t_off) in zip(inputs, resident):
        bb = datok_amd.Batch(total, n_docs)
        bb.set_chunking(datok_amd.Batch.AUTO_CHUNK if args.chunk < 0 else args.chunk, args.warm)
        bb.set_input_device(t_text.data_ptr(), t_off.data_ptr(), n_docs, total, keep=(t_text, t_off), doc_off_host=doc_off)
        batches.append(bb)
    batch = batches[0]
    text, doc_off = inputs[0]

    # ---- parity gate (oracle is the checker, never the thing measured): every batch in flight, sampled documents
    tot = None
    for k, bb in enumerate(batches):
        bb.run(tok, RUN_FLAGS)
        tk = bb.totals()
        tot = tot or tk
        if args.parity_docs:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from parity import assert_batch_equals_oracle
            from oracle import oracle as O
            om = O.Model(args.model)
            res = bb.result()
            step = max(1, n_docs // max(1, args.parity_docs // len(batches)))
            n = assert_batch_equals_oracle(om, res, inputs[k][0], inputs[k][1], docs=range(0, n_docs, step), fields=RUNE_FIELDS)
            assert tk["n_flagged"] == 0 and n > 0
            del res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- per-kernel times (HIP events on the batches' streams): one batch alone on the GPU, then the regime of the
    #      timed region.  They run BEFORE the timed region, not behind it: the oracle's parity check above leaves the GPU
    #      idle for seconds, and W = 5 warm-up steps (half a millisecond) do not bring its clocks and caches back --
    #      K = 20 steps then read 148-151 GB/s where the same 20 steps behind 200 warm-up steps read 157-158.  After
    #      these measurements the timed region finds the GPU as a running service has it.
    one = None
    pre_steps = len(batches)  # (the parity gate's runs)
    if len(batches) > 1:
        timed_steps(batches[:1], tok, max(args.steps, 30), barrier)  # (untimed: the first launches after the idle phase)
        e1 = timed_steps(batches[:1], tok, args.steps, barrier)
        s1 = stage_times(batches[:1], tok, args.steps)
        pre_steps += max(args.steps, 30) + args.steps + 4 * max(args.steps // 4, 2)
        one = {"value": round(total * args.steps / e1 / 1e6, 1), "ms_per_step": round(e1 / args.steps * 1e3, 4),
               "kernel_ms": round(s1["walk"], 4), "stages_ms": {k: round(v, 4) for k, v in s1.items()}}
    stage_avg = stage_times(batches, tok, args.steps)
    pre_steps += 4 * max(args.steps // 4, 2 * len(batches))

    # ---- warmup
    ran = [False] * len(batches)
    for i in range(args.warmup):
        k = i % len(batches)
        if ran[k]:
            batches[k].totals()
        batches[k].run(tok, RUN_FLAGS)
        ran[k] = True
    for bb in batches:
        bb.sync()

    # ---- timed region: exactly K steps
    elapsed = timed_steps(batches, tok, args.steps, barrier)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    repair_rounds = max(int(bb.totals()["repair_rounds"]) for bb in batches)

    # ---- PCIe inclusive: the same inputs as one corpus in page-locked host memory, through dtk_pipeline.  In a process
    #      of its own, without torch: a Go caller has none, and with torch initialised in the process the same pipeline's
    #      downloads ran at two thirds of the rate (scripts/e2e_diag2.py WITH_TORCH=1: 24 against 36.7 GB/s).
    h2d = None
    if rank == 0 and world == 1 and not args.no_e2e:
        for bb in batches:   # (their HIP streams: the runtime has four hardware queues)
            bb.close()
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--e2e-only", "--docs", str(n_docs), "--doc-bytes", str(args.doc_bytes),
               "--model", args.model, "--chunk", str(args.chunk), "--warm", str(args.warm), "--streams", str(n_streams)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                h2d = json.loads(line)
        if h2d is None:
            print("bench.py: end-to-end block failed: %s" % r.stderr[-400:], file=sys.stderr)

    # ---- offset gather to rank 0 over RCCL (config 5's exchange), outside the clock
    gather_ms, gather_hung, gather_err = None, False, None
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
                    box["bytes"] = sum(int(t.numel()) * 4 for ts in got.values() for t in ts)
            except Exception as e:  # report, do not lose the run
                box["err"] = repr(e)
        th = threading.Thread(target=_gather, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("DATOK_GATHER_TIMEOUT", "120")))
        gather_hung = th.is_alive()
        gather_err = box.get("err")
        if gather_hung or gather_err:
            print("bench.py: offset gather %s on rank %d" % ("timed out" if gather_hung else "failed: " + gather_err, rank),
                  file=sys.stderr)
        gather_ms = box.get("ms")
        gather_bytes = box.get("bytes")
        # no barrier behind the gather: a rank whose peer is stuck must still get to its output

    # ---- CPU baseline: the C restatement of the Go algorithm, rank 0, N = 1 only, on the first batch's input
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        om = O.Model(args.model)
        cores = os.cpu_count() or 1
        om.count_batch(text[: 64 * args.doc_bytes], doc_off[:65], 1)
        t1 = time.perf_counter()
        om.count_batch(text, doc_off, 1)
        one_thread = total / (time.perf_counter() - t1) / 1e6
        reps, spent = 0, 0.0
        t1 = time.perf_counter()
        while spent < args.cpu_seconds:
            om.count_batch(text, doc_off, cores)
            reps += 1
            spent = time.perf_counter() - t1
        cpu = {"value": round(reps * total / spent / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
               "sample": "the first %d x %d B batch, %d passes, %d threads (oracle/datok_oracle.c, C restatement "
                         "of matrix.go:348-698, counting sink)" % (n_docs, args.doc_bytes, reps, cores),
               "single_thread_MBps": round(one_thread, 1),
               "reference_published_MBps_per_core": 24.8}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * total * args.steps / elapsed / 1e6
        # algorithmic bytes of one launch: input + doc offsets + i32 offsets out + counts
        b_alg = total + 4 * (n_docs + 1) + 4 * (2 * tot["n_tokens"] + tot["n_sent"]) + 8 * n_docs
        walk_s = stage_avg["walk"] * 1e-3
        achieved = b_alg / walk_s / 1e9
        split = os.environ.get("DATOK_SPLIT_START", "0") not in ("", "0")
        both_s = (stage_avg["walk"] + (stage_avg["spec_start"] if split else 0.0)) * 1e-3
        # HBM bytes of that kernel from the last committed rocprofv3 --pmc passes (profiles/)
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tr.get("docs") == n_docs and tr.get("doc_bytes") == args.doc_bytes and \
                    tr.get("chunk_bytes") == tot["chunk_bytes"]:
                traffic = tr["walk_kernel_hbm_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        mname = os.path.basename(args.model)
        cfg_no = "configs[3]" if mname.endswith(".datok") else "configs[1]"
        workload = (mname + ", %d equal-length %d B synthetic German docs per batch (BASELINE.json " + cfg_no + ")"
                    if world == 1 else
                    mname + ", one shard of the 10 GiB corpus per GPU: %d docs x %d B (BASELINE.json configs[4])")
        info = tok.info
        if info["kind"] == 1:
            layout = ("dense: the double array's transitions laid out as a fused matrix at load (%d states)" % info["dense_states"]
                      if info["dense_states"] else "pairs: the file's {base, check} pairs, two dependent loads per step")
        else:
            layout = "matrix: state-major %s cells" % ("fused u32" if info["entry_bytes"] == 4 else "u16")
        out = {
            "metric": "input MB/s tokenized, %s" % os.path.basename(args.model),
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "pre_timed_steps": pre_steps + args.warmup,
            "pre_timed_steps_note": "untimed launches before the K timed steps: parity gate, the per-kernel measurements "
                                    "(streams_1, stages_ms) and the W warm-up steps -- the GPU's clocks come up during them "
                                    "(DESIGN.md section 4)",
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 table cells, u8 symbol codes", "data": "synthetic",
            "config": {"workload": workload % (n_docs, args.doc_bytes),
                       "table_layout": layout,
                       "docs_per_gpu": n_docs, "doc_bytes": args.doc_bytes,
                       "parallelism": "documents sharded over %d GPU(s), no data-path collective" % world,
                       "batches_in_flight": len(batches),
                       "inputs": "one per batch in flight (generator seeds %s on rank 0)" % ", ".join(map(str, seeds)),
                       "step": "dtk_batch_run + dtk_batch_totals (speculation check, repairs, capacity check) per batch"},
            "roofline": {"bound": "hbm",
                         "kernel": walk_kernel_name(tok.info, bool(tot["chunk_bytes"])),
                         "achieved": round(achieved, 2),
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6),
                         "traffic": traffic,
                         "traffic_source": None if traffic is None else "profiles/traffic.json (rocprofv3 --pmc passes of "
                                           "this workload, %s)" % tr.get("measured", "committed with the round's profiles"),
                         "algorithmic_bytes": int(b_alg),
                         "kernel_ms": round(stage_avg["walk"], 4),
                         "kernel_ms_note": "average launch duration with %d batches in flight (kernels of different "
                                           "batches share the CUs); one batch alone: streams_1.kernel_ms" % len(batches),
                         "lookups_per_launch": int(tot["walk_steps"]),
                         "Glookups_per_s_start_plus_walk": round(tot["walk_steps"] / both_s / 1e9, 3)},
            "stages_ms": {k: round(v, 4) for k, v in stage_avg.items()},
            "streams_1": one,
            "end_to_end": h2d,
            "tokens_per_launch": int(tot["n_tokens"]),
            "walk": {"lanes": int(tot["n_lanes"]), "chunk_bytes": int(tot["chunk_bytes"]), "warm_bytes": args.warm,
                     "repair_rounds": repair_rounds},
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "gather_hung": bool(gather_hung), "gather_error": gather_err,
            "cpu_baseline": cpu,
        }
        if world > 1 and gather_ms is not None:
            out["gather_GBps"] = round(gather_bytes / gather_ms / 1e6, 2)
        print(json.dumps(out))
    if world > 1:
        if gather_hung or gather_err:
            sys.stdout.flush()
            os._exit(3)  # a stuck point-to-point operation would also block the teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
