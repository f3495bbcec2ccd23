#!/usr/bin/env python3
"""bench.py -- query sentences/sec + rules/sec of the hot path on N MI355X, one JSON line.

A step = one pass of the whole hot path (H2D of the query ids, batched SA interval search,
gappy-phrase search, rule extraction, lexicon + MaxLex features, grammar files written) over
one batch of synthetic query sentences, with the corpus index already resident in HBM.
Index construction (device suffix array, frequent-pair precomputation) and, for N > 1, the
one-time RCCL broadcast of the index are outside the timed region and reported separately.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Weak scaling: every rank processes `--queries` sentences per step against a full replica of
the index; value = (N * queries * steps) / max-over-ranks wall time.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(n_tokens, lm):
    """B(N,l) = 2*ceil(log2 N)*(4+4l) + 4l + 8 bytes per interval lookup (t,l), l <= 5."""
    lg = int(np.ceil(np.log2(max(n_tokens, 2))))
    total = 0; lookups = 0
    for l in range(1, 6):
        c = int((lm >= l).sum())
        total += c * (2 * lg * (4 + 4 * l) + 4 * l + 8); lookups += c
    return total, lookups


def cpu_baseline(corpus, args):
    """The CPU oracle (kind "port") timed on a bounded sample: a prefix of the same corpus and
    queries drawn from it by the same recipe.  Index build is excluded, as for the GPU."""
    from cgx_amd import synth
    import cgx_amd
    lib_path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib_path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(lib_path)
    sub = synth.prefix(corpus, min(args.cpu_pairs, corpus["pairs"]))
    qoff, qtok = synth.make_queries(sub, args.cpu_queries, args.seed + 99)
    # suffix array of the sample from the device builder (index construction is not what is timed)
    ex = cgx_amd.Extractor(0)
    P = np.zeros(len(sub["str"]), np.uint32)
    ex.upload_index(sub["str"], P, sub["tstr"], sub["ltar"], sub["rtar"], sub["lexk"][:1], sub["lexv"][:1])
    ex.build_sa(); sa = ex.fetch("sa"); ex.close()
    vp = C.c_void_p
    lib.orc_index_from_arrays.restype = vp
    lib.orc_index_from_arrays.argtypes = [vp, C.c_uint32, vp, C.c_int32, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp]
    lib.orc_batch_from_ids.restype = vp; lib.orc_batch_from_ids.argtypes = [vp, C.c_int32, vp, C.c_int32]
    lib.orc_run_all.argtypes = [vp, vp, C.c_char_p]; lib.orc_batch_free.argtypes = [vp]; lib.orc_index_free.argtypes = [vp]
    p = lambda a: a.ctypes.data_as(vp)
    arrs = [np.ascontiguousarray(sub[k]) for k in ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv")]
    ix = lib.orc_index_from_arrays(p(arrs[0]), len(arrs[0]), p(arrs[1]), len(arrs[1]) - 1, p(arrs[2]), len(arrs[2]), p(arrs[3]), p(arrs[4]), p(arrs[5]),
                                   p(arrs[6]), p(arrs[7]), p(arrs[8]), p(arrs[9]), len(arrs[8]), p(sa))
    out = tempfile.mkdtemp(prefix="cgx_cpu_")
    b = lib.orc_batch_from_ids(p(qoff), len(qoff), p(qtok), len(qtok))
    t0 = time.perf_counter()
    lib.orc_run_all(ix, b, out.encode())
    dt = time.perf_counter() - t0
    lines = sum(sum(1 for _ in open(os.path.join(out, f), "rb")) for f in os.listdir(out))
    lib.orc_batch_free(b); lib.orc_index_free(ix); shutil.rmtree(out, ignore_errors=True)
    return {"value": round(len(qoff) / dt, 3), "unit": "query sentences/s", "cores": 1, "kind": "port",
            "rules_per_s": round(lines / dt, 1), "seconds": round(dt, 3),
            "sample": "%d-pair prefix of the corpus (N=%d source tokens), %d queries by the same recipe, single-thread C oracle, index build excluded"
                      % (sub["pairs"], len(sub["str"]), len(qoff))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=10000000, help="sentence pairs in the synthetic corpus (BASELINE configs[2]: 10M)")
    ap.add_argument("--vocab", type=int, default=200000)
    ap.add_argument("--queries", type=int, default=10000, help="query sentences per rank per step (BASELINE configs[2]: 10k)")
    ap.add_argument("--outdir", default=None, help="where the grammar files go (default: a fresh directory under /dev/shm, else $TMPDIR)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-pairs", type=int, default=100000)
    ap.add_argument("--cpu-queries", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl (= RCCL over xGMI) for real runs, gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--sub-batch", type=int, default=0, help="queries per internal batch inside one step (0 = the whole step at once)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not bind the writer threads to the GPU's NUMA node")
    ap.add_argument("--no-rewrite-run", action="store_true", help="skip the informational in-place rewrite measurement")
    ap.add_argument("--outdir-mode", choices=("auto", "fresh", "inplace"), default="auto", help="fresh: a new directory per step (default when there is room for four steps' files); inplace: every step rewrites the same files")
    ap.add_argument("--sync-write", action="store_true", help="write each step's files before starting the next step")
    ap.add_argument("--no-write", action="store_true", help="format nothing, write no files (kernel-side study only; not the headline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cgx_amd import synth, shard
    import cgx_amd

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the extractor has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(args.backend)
    ex = cgx_amd.Extractor(local)
    if args.sub_batch:
        ex.set_option("sub_batch", args.sub_batch)
    if args.no_numa_pin:
        ex.set_option("numa_pin", 0)
    if not args.sync_write:
        ex.set_option("async_write", 1); ex.set_option("prealloc_text", 1)       # files of step k are written by host threads while the GPU runs step k+1; flushed inside the timed region

    # ---- synthetic corpus (same seed on every rank: host arrays are needed by the host stages) ----
    t0 = time.perf_counter()
    corpus = synth.make_corpus(args.pairs, args.vocab, args.seed)
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                                   corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    t_gen = time.perf_counter() - t0
    # ---- index: built on rank 0, broadcast once over RCCL/xGMI ----
    t0 = time.perf_counter(); t_bcast = 0.0
    if rank == 0:
        ex.upload_corpus(host)
    if world > 1:
        meta = [None]
        if rank == 0:
            meta = [ex.index_shape()]
        dist.broadcast_object_list(meta, src=0)
        if rank != 0:
            ex.index_alloc(meta[0])
        torch.cuda.synchronize(); dist.barrier(); tb = time.perf_counter()
        for i, (name, nbytes) in enumerate(ex.index_buffers()):
            if nbytes == 0:
                continue
            stage = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            if rank == 0:
                ex.index_d2d(i, stage.data_ptr(), 0)
            dist.broadcast(stage, src=0)
            torch.cuda.synchronize()
            if rank != 0:
                ex.index_d2d(i, stage.data_ptr(), 1)
            del stage
        if rank != 0:
            ex.index_finalize()
        torch.cuda.synchronize(); dist.barrier(); t_bcast = time.perf_counter() - tb
    t_index = time.perf_counter() - t0

    # ---- queries: every rank gets its own shard of the global batch (weak scaling) ----
    gq_off, gq_tok = synth.make_queries(corpus, args.queries * world, args.seed + 3087)
    first, qoff, qtok = shard.take_shard(gq_off, gq_tok, rank, world)
    # output directory: the fastest writable place with room for this rank's files (about 2.2 MB per query on this workload)
    need = int(len(qoff) * 3.0e6 * 1.2) + (1 << 30)
    cands = [args.outdir] if args.outdir else [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d) and os.access(d, os.W_OK)]
    base = None
    for d in cands:
        try:
            if shutil.disk_usage(d).free >= need * (world if not args.outdir else 1):
                base = d; break
        except OSError:
            pass
    if base is None and cands:
        base = max(cands, key=lambda d: shutil.disk_usage(d).free)
    outroot = None if args.no_write else tempfile.mkdtemp(prefix="cgx_bench_r%d_" % rank, dir=base)
    # Every step writes its files into a FRESH directory, as a real run does (a step that rewrites the previous step's
    # files in place measures the page cache's second-touch behaviour instead: the first rewrite of a file set takes
    # 4x the CPU time of either a fresh write or a later rewrite).  Directories of finished steps are unlinked by
    # background threads while later steps run; with too little room for three live directories the steps share one
    # directory and one extra untimed step makes sure the timed ones are not the first rewrite.
    fresh = bool(outroot) and args.outdir_mode != "inplace" and shutil.disk_usage(base).free >= 4 * need * (world if not args.outdir else 1)
    if args.outdir_mode == "fresh" and outroot and not fresh:
        raise SystemExit("bench.py: not enough room under %s for --outdir-mode fresh" % base)
    # Unlinking an older step's 33 GB while later steps run is bench hygiene, not workload, and it is not free (it
    # contends with the writers inside the page cache: +13 % step time).  So finished directories are kept as long as
    # they fit in 60 % of the scratch space (shared by all ranks) and only the excess is unlinked in the background.
    keep_live = 2
    if fresh:
        keep_live = max(2, int(0.6 * shutil.disk_usage(base).free / (need * (world if not args.outdir else 1))) - 1)
    import concurrent.futures, collections, threading
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
    deletions = []; live = collections.deque(); stepno = [0]

    def unlink_many(paths):
        try:                                                   # lowest priority: take only the CPU time the writer threads leave idle
            os.setpriority(os.PRIO_PROCESS, threading.get_native_id(), 19)
        except (OSError, AttributeError):
            pass
        for p in paths:
            try:
                os.unlink(p)
            except OSError:
                pass

    def remove_dir_async(d):
        names = [os.path.join(d, n) for n in os.listdir(d)]
        futs = [pool.submit(unlink_many, names[i::4]) for i in range(4)]
        deletions.append((d, futs))

    def step():
        if not outroot:
            return ex.extract_grammars_ids(host, qoff, qtok, None, first)
        if not fresh:
            return ex.extract_grammars_ids(host, qoff, qtok, outroot, first)
        d = os.path.join(outroot, "step%d" % stepno[0]); stepno[0] += 1
        os.mkdir(d)
        n = ex.extract_grammars_ids(host, qoff, qtok, d, first)   # returns once this step's text is laid out; the previous step's files are complete by then
        live.append(d)
        while len(live) > keep_live:                                  # the newest directory is still being written, the one before it was just completed
            remove_dir_async(live.popleft())
        return n

    priming = 0
    for _ in range(args.warmup):
        step()
    if outroot and not fresh and args.warmup < 2:             # in-place mode: the timed steps must not be the first rewrite of the files
        step(); priming = 1
    ex.flush()
    kernel_ms = []; stage = {k: 0.0 for k in ("sa_lookup", "blocks", "gappy", "extract", "lexicon", "format", "fmt_lists", "fmt_count", "fmt_alloc", "fmt_write", "look1_kernel", "look2_kernel")}; hoststage = {k: 0.0 for k in ("lists", "lexicon", "write", "write_wait_d2h", "write_file", "total", "t_upload_sa", "t_fetch_lm", "t_blocks", "t_qblocks", "t_gappy", "t_extract", "t_lexicon", "t_format", "t_offsets", "t_flush_wait")}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); rules = 0
    for _ in range(args.steps):
        rules += step()
        kernel_ms.append(ex.stage_ms("sa_lookup_kernel"))
        for k in stage: stage[k] += ex.stage_ms(k)
        for k in hoststage: hoststage[k] += ex.host_ms(k)
    ex.flush()                                # every grammar file of every timed step is on disk before the clock stops
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None)
    total_q = shard.sum_over_ranks(len(qoff) * args.steps, dist if world > 1 else None)
    total_rules = shard.sum_over_ranks(rules, dist if world > 1 else None)

    # informational second measurement (N=1 only): the same K steps rewriting ONE directory in place, after two
    # untimed steps so that no timed step is the first rewrite.  No page allocation, no unlinking: what is left is
    # the pipeline itself (PCIe D2H at ~55 GB/s is the limit).  Never used for `value`.
    rewrite = None
    if fresh and world == 1 and not args.no_rewrite_run:
        rd = os.path.join(outroot, "rewrite"); os.mkdir(rd)
        for _ in range(2):
            ex.extract_grammars_ids(host, qoff, qtok, rd, first)
        ex.flush(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ex.extract_grammars_ids(host, qoff, qtok, rd, first)
        ex.flush(); torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        rewrite = {"value": round(len(qoff) * args.steps / dt1, 3), "unit": "query sentences/s", "ms_per_step": round(dt1 / args.steps * 1e3, 3),
                   "what": "same steps, one output directory rewritten in place (files and their page-cache pages already exist)"}
    lastdir = (live[-1] if fresh else outroot) if outroot else None
    out_bytes = sum(e.stat().st_size for e in os.scandir(lastdir) if e.is_file()) if lastdir else 0
    for d, futs in deletions:                                  # outside the timed region: only bookkeeping is left here, the unlinks ran during the steps
        for f in futs:
            f.result()
    pool.shutdown()
    free_b, total_b = torch.cuda.mem_get_info()                # after the timed steps: index + cached batch buffers + both text slots

    if rank == 0:
        lm = ex.fetch("lm"); c = ex.counts()
        abytes, lookups = algorithmic_bytes(len(corpus["str"]), lm)
        kms = float(np.mean(kernel_ms))
        traffic = None                      # HBM bytes per launch from the committed PMC passes (same config only)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_k_sa_lookup.json")))
            pc = pmc["config"]
            if (pc["pairs"], pc["queries"], pc["seed"], pc["vocab"]) == (args.pairs, args.queries, args.seed, args.vocab):
                traffic = int(pmc["traffic_bytes_raw"])
        except Exception:
            traffic = None
        ach = abytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # the two kernels that take the most time per batch, priced the same way: bytes the algorithm must touch per
        # occurrence (SA entry 4 + text window 64 + alignment window 64 + sentence offset 4 [+ the 9-byte hit record
        # for look2]) + 8 per hit written, over the stage's event time (includes the launch sizing round trip)
        w1, w2 = ex.stage_ms("look1_items"), ex.stage_ms("look2_items")
        k1, k2 = stage["look1_kernel"] / args.steps, stage["look2_kernel"] / args.steps
        by_time = []
        for name, w, per, hits, ms in (("k_look1 (one-gap corpus lookups)", w1, 136, c["h1"], k1), ("k_look2 (two-gap corpus lookups)", w2, 145, c["h2"], k2)):
            if w > 0 and ms > 0:
                ab = w * per + hits * 8
                by_time.append({"kernel": name, "bound": "hbm", "occurrences_per_step": int(w), "algorithmic_bytes_per_step": int(ab), "ms_per_step": round(ms, 3),
                                "achieved": round(ab / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ab / (ms * 1e-3) / 1e9 / 8000.0, 4)})
        line = {
            "metric": "query sentences/sec", "value": round(total_q / dt, 3), "unit": "query sentences/s",
            "rules_per_s": round(total_rules / dt, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32", "data": "synthetic",
            "config": {"workload": "synthetic Zipf parallel corpus (BASELINE configs[2] recipe, %d sentence pairs), %d queries per GPU per step"
                                   % (args.pairs, args.queries),
                       "sentence_pairs": args.pairs, "source_tokens": int(len(corpus["str"])), "vocab": args.vocab,
                       "queries_per_gpu": int(len(qoff)), "query_tokens_per_gpu": int(len(qtok)), "parallelism": "query-shard x%d, index replicated" % world,
                       "grammar_files_written": not args.no_write, "grammar_bytes_per_step": out_bytes, "writer": "sync" if args.sync_write else "async (host threads overlap the next step; flushed before the clock stops)", "outdir": base if outroot else None,
                       "outdir_mode": (("fresh directory per step, up to %d finished ones kept, older ones unlinked in the background" % keep_live) if fresh else "one directory rewritten in place, %d untimed priming step(s)" % priming) if outroot else None},
            "roofline": {"bound": "hbm", "kernel": "k_sa_lookup (batched SA interval search)", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 4), "traffic": traffic, "algorithmic_bytes_per_launch": int(abytes), "lookups_per_launch": int(lookups),
                         "kernel_ms": round(kms, 4), "pmc_GBps": (round(traffic / (kms * 1e-3) / 1e9, 1) if traffic and kms > 0 else None),
                         "note": "achieved prices every lookup at the reference's full-depth binary search (SURVEY 8d); this kernel replaces l=1,2 by table probes, so achieved can exceed the peak while pmc_GBps is the traffic it really moves"},
            "rewrite_in_place": rewrite,
            "roofline_largest_per_batch_kernels": by_time,
            "stages_ms_per_step": {k: round(v / args.steps, 3) for k, v in {**stage, **{"host_" + k: v for k, v in hoststage.items()}}.items()},
            "index": {"build_sa_ms": round(ex.stage_ms("build_sa"), 1), "precompute_ms": round(ex.stage_ms("precompute"), 1),
                      "broadcast_s": round(t_bcast, 3), "total_s": round(t_index, 2), "corpus_gen_s": round(t_gen, 2), "frequent_pair_hits": c["nphits"]},
            "hbm_in_use_gb": round((total_b - free_b) / 1e9, 1),
            "counts": {k: c[k] for k in ("d1", "d2", "h1", "h2", "g", "n0", "n1", "n2", "guard_exits")},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(corpus, args)
        print(json.dumps(line))
    if outroot:
        shutil.rmtree(outroot, ignore_errors=True)
    ex.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
