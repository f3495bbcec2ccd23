#!/usr/bin/env python3
"""bench.py -- query sentences/sec + rules/sec of the hot path on N MI355X, one JSON line.

A step = one pass of the whole hot path (H2D of the query ids, batched SA interval search,
gappy-phrase search, rule extraction, lexicon + MaxLex features, grammar files written) over
one batch of synthetic query sentences, with the corpus index already resident in HBM.
Index construction (device suffix array, frequent-pair precomputation, l-gram tables) and, for
N > 1, the one-time RCCL broadcast of the index are outside the timed region and reported
separately.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads (BASELINE.json configs): --config toy (20 k sentence pairs, 7 queries per step: the toy line north_star asks for;
the reference's hansards files are not in its checkout, so the corpus is synthetic), cfg3 (default; 10 M sentence pairs, 10 k queries per
GPU per step, weak scaling), cfg4 (Europarl scale, N = 5.5e7 source tokens, one 50 k-query batch
split over the ranks: strong scaling), cfg5 (1e8 source tokens, 1 M queries split over the ranks).

Host memory is bounded whatever --steps/--warmup say: the grammar files of a step go into ONE
spool directory per rank that every step rewrites in place (sized against the memory this
process may really use: MemAvailable and the cgroup limit, not the tmpfs mount size, and counting
the writer's page-locked buffers as well); when one step's files do not fit, the step is cut into
chunks that reuse the same file slots.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (sentence pairs, vocabulary, queries, queries are per GPU?, default scaling, default steps, default warmup)
    "toy": dict(pairs=20_000, vocab=160, queries=7, scaling="weak", steps=40, warmup=5, lo=4, hi=18,
                what="BASELINE configs[1] stand-in (toy/hansards is absent from the reference checkout): synthetic 20k-pair toy corpus of the test suite's shape (vocabulary 160, sentences of 4..18 tokens), 7 queries per GPU per step"),
    "cfg3": dict(pairs=10_000_000, vocab=200_000, queries=10_000, scaling="weak", steps=5, warmup=2,
                 what="BASELINE configs[2]: synthetic 10M-sentence parallel corpus (Zipf token ids), 10k queries per GPU per step"),
    "cfg4": dict(pairs=2_115_000, vocab=150_000, queries=50_000, scaling="strong", steps=2, warmup=1,
                 what="BASELINE configs[3]: Europarl-scale synthetic corpus (about 2.1M sentence pairs, N = 5.5e7 source tokens), one 50k-query batch split over the ranks"),
    "cfg5": dict(pairs=3_846_000, vocab=200_000, queries=1_000_000, scaling="strong", steps=1, warmup=0,
                 what="BASELINE configs[4]: synthetic corpus of 1e8 source tokens, 1M queries split over the ranks (stress run)"),
}
EST_BYTES_PER_QUERY = 4.0e6          # grammar text per query before the first measurement (3.3e6 measured on cfg3)


def make_chunks(qoff, ntok, max_queries, tok_cap=300000):
    """Query ranges [a, b) of one step: at most `max_queries` sentences (what the spool holds) and at most `tok_cap` query
    tokens (one internal batch of the library) each; a sentence longer than the cap is a chunk of its own."""
    qoff = np.asarray(qoff, np.int64); nq = len(qoff); chunks = []; a = 0
    qend = np.append(qoff[1:], ntok).astype(np.int64) if nq else np.zeros(0, np.int64)
    while a < nq:
        b = int(np.searchsorted(qend, int(qoff[a]) + tok_cap, side="right"))     # queries a..b-1 hold <= tok_cap tokens
        b = max(a + 1, min(b, a + max(int(max_queries), 1), nq))
        chunks.append((a, b)); a = b
    return chunks or [(0, 0)]


def survey_bytes(n_tokens, lm):
    """B(N,l) = 2*ceil(log2 N)*(4+4l) + 4l + 8 bytes per interval lookup (t,l), l <= 5 --
    what the REFERENCE's full-depth binary search would touch for the same lookups."""
    lg = int(np.ceil(np.log2(max(n_tokens, 2))))
    total = 0; lookups = 0
    for l in range(1, 6):
        c = int((lm >= l).sum())
        total += c * (2 * lg * (4 + 4 * l) + 4 * l + 8); lookups += c
    return total, lookups


def memory_budget():
    """Bytes of host memory this process tree may still take: MemAvailable, capped by the cgroup limit."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    limit = None
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                limit = int(v) - int(open(cur).read().strip())
            break
        except (OSError, ValueError):
            continue
    cands = [x for x in (avail, limit) if x is not None and x > 0]
    return min(cands) if cands else 64 << 30, avail, limit


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def save_corpus_for_cpu(corpus, qs, args):
    """The benchmark's own corpus arrays and the first queries of its first batch as .npy files for the CPU leg on the full corpus
    (tools/cpu_baseline.py --full-dir); the suffix array and the frequent-pair tables are added after the timed region."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="cgx_bench_cpu_full_", dir=base)
    for k in ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv"):
        np.save(os.path.join(d, k + ".npy"), np.ascontiguousarray(corpus[k]))
    nqs = min(len(qs["qoff"]), args.cpu_full_queries * 20)       # the one-core leg's queries plus the workers' shares
    qo = np.asarray(qs["qoff"][:nqs], np.int64); end = int(qs["qoff"][nqs]) if nqs < len(qs["qoff"]) else len(qs["qtok"])
    np.save(os.path.join(d, "qoff.npy"), qo); np.save(os.path.join(d, "qtok.npy"), np.ascontiguousarray(qs["qtok"][:end], np.int32))
    return d


def run_cpu_full_corpus(ex, full_dir, args):
    """After the timed regions: the GPU-built suffix array and frequent-pair tables join the corpus files, and the CPU restatement runs
    the first queries of the first batch against the WHOLE benchmark corpus (one core, then all cores), in its own process."""
    try:
        for k in ("sa", "freq", "pidx", "miss", "phit_start", "phit_len"):
            np.save(os.path.join(full_dir, k + ".npy"), ex.fetch(k))
        cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--full-dir", full_dir, "--full-queries", str(args.cpu_full_queries), "--full-seconds", str(args.cpu_full_seconds)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=4 * args.cpu_full_seconds + 240)
        for line in reversed(p.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "full-corpus CPU leg printed no result (exit %s): %s" % (p.returncode, p.stderr[-300:])}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(full_dir, ignore_errors=True)


def start_cpu_baseline(args, cfg):
    """tools/cpu_baseline.py in its own process (never touches the GPU); the result is collected later."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--pairs", str(min(args.cpu_pairs, cfg["pairs"])), "--vocab", str(cfg["vocab"]),
           "--seed", str(args.seed), "--budget-s", str(args.cpu_seconds), "--corpus-pairs", str(cfg["pairs"]), "--full-sa-seconds", str(args.cpu_full_sa_seconds if (cfg.get("lo", 5), cfg.get("hi", 45)) == (5, 45) else 0)]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def collect_cpu_baseline(proc, timeout):
    try:
        out, _ = proc.communicate(timeout=timeout)
        for line in reversed(out.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "cpu baseline printed no result (exit %s)" % proc.returncode}
    except subprocess.TimeoutExpired:
        proc.kill()
        return {"error": "cpu baseline exceeded %d s" % timeout}
    except Exception as e:                                     # a failed baseline must not cost the benchmark its line
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--pairs", type=int, default=None, help="sentence pairs in the synthetic corpus (default: the config's)")
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--queries", type=int, default=None, help="query sentences per step: per GPU with --scaling weak, in all with --scaling strong")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="weak: every rank gets --queries per step; strong: --queries are split over the ranks")
    ap.add_argument("--outdir", default=None, help="parent of the spool directory (default: /dev/shm, else $TMPDIR)")
    ap.add_argument("--mem-fraction", type=float, default=0.80, help="share of the usable host memory (MemAvailable, cgroup headroom) this job may take: per rank a fixed part, the spool files and the writer's page-locked buffers (1.6x the spool)")
    ap.add_argument("--chunk-queries", type=int, default=0, help="queries per spool chunk (0: as many as fit the spool budget, at most one step)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-pairs", type=int, default=200000, help="sentence pairs of the CPU baseline's sample corpus")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="seconds of timed CPU work per baseline leg (one core, all cores)")
    ap.add_argument("--cpu-full-sa-seconds", type=float, default=150.0, help="the CPU baseline process also runs the reference's own suffixArrayConstruct on the WHOLE corpus (one core, beside its other legs), for at most this long (0: skip)")
    ap.add_argument("--cpu-full-queries", type=int, default=64, help="the CPU baseline ALSO runs this many queries of the first timed batch against the benchmark's own corpus (GPU-built suffix array and frequent-pair tables handed to the CPU restatement), on one core and then on all cores, after the timed region (0: skip; only for N = 1 and corpora of at least 1M pairs)")
    ap.add_argument("--cpu-full-seconds", type=float, default=60.0, help="time cap per leg of the full-corpus CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl (= RCCL over xGMI) for real runs, gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--bcast", choices=("torch", "c"), default="torch", help="index broadcast: torch = torch.distributed.broadcast (RCCL) on a staging tensor per buffer; c = the library's own cgx_broadcast_index on an RCCL communicator made here (what a C host would call)")
    ap.add_argument("--force-dist", action="store_true", help="with one rank: still create the torch.distributed process group (backend as given) and take the N > 1 code path -- index broadcast through RCCL on a one-rank communicator, barriers, gathers -- so that path runs on a one-GPU box")
    ap.add_argument("--sub-batch", type=int, default=0, help="at most this many queries per batch (submitted as chunks of a step; 0 = automatic: as many as fit the spool, at most one step)")
    ap.add_argument("--no-numa-pin", action="store_true", help="do not bind the writer threads to the GPU's NUMA node")
    ap.add_argument("--sync-write", action="store_true", help="write each chunk's files before starting the next chunk")
    ap.add_argument("--fresh-steps", type=int, default=None, help="extra timed steps AFTER the contract's K steps that write every chunk into a NEW directory (value_fresh_files); default 3 (1 when a step is several chunks), 0 = skip")
    ap.add_argument("--query-sets", type=int, default=3, help="distinct query batches (different seeds) the steps rotate through: step i runs set i %% N, so capacity guesses, tables and file sizes change from step to step as in a serving run")
    ap.add_argument("--gz-steps", type=int, default=None, help="extra timed steps AFTER the contract's K steps with grammar.<q>.s.gz output, the gzip members made by the GPU formatter (value_gz); default K (as many as the plain leg, so that the drain of the writer pipeline after the last step weighs the same in both figures), 0 = skip")
    ap.add_argument("--two-context-steps", type=int, default=None, help="extra timed .gz steps AFTER the others, dealt alternately to TWO contexts over the one index (cgx_share_index), each driven by a thread of its own (value_gz_two_contexts); default 0 = skip")
    ap.add_argument("--no-write", action="store_true", help="count the rules on the GPU, lay out no text, write no files (kernel-side study; not the headline)")
    ap.add_argument("--option", action="append", default=[], help="name=value passed to cgx_set_option (repeatable)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("pairs", "vocab", "queries", "scaling"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if args.single_device:
        local = 0
    multi = world > 1 or args.force_dist                  # the torch.distributed path (a one-rank group with --force-dist)
    if local_world > 1 and "CGX_THREADS" not in os.environ:
        # the ranks of one host share its CPUs (and its cgroup quota): each writer gets its share, two CPUs of it left to
        # the thread that feeds the GPU (what the library does by itself for a single process)
        share = usable_cpus() // local_world
        os.environ["CGX_THREADS"] = str(max(2, min(16, share - 2 if share >= 8 else share)))   # never fewer than two: one writer thread cannot keep up with any GPU
    # the CPU baseline runs first, in its own process, while this one generates the corpus: its cores are free
    # again long before the timed region starts (it is joined before the warm-up steps)
    cpu_proc = start_cpu_baseline(args, cfg) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    import torch
    import torch.distributed as dist
    from cgx_amd import synth, shard
    import cgx_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the extractor has no CPU fallback")
    torch.cuda.set_device(local)
    if multi:
        if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
            dist.init_process_group(args.backend)
        else:                                             # --force-dist outside a launcher: a one-rank group on the loopback interface
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    ex = cgx_amd.Extractor(local)
    if args.sub_batch:
        ex.set_option("sub_batch", args.sub_batch)
    if args.no_numa_pin:
        ex.set_option("numa_pin", 0)
    if not args.sync_write:
        ex.set_option("async_write", 1); ex.set_option("prealloc_text", 1)       # files of chunk k are written by host threads while the GPU runs chunk k+1; flushed inside the timed region
    for ov in args.option:
        k, v = ov.split("="); ex.set_option(k, int(v))

    # ---- synthetic corpus: generated once per node (its first rank), the other ranks map the files it leaves in shared memory ----
    t0 = time.perf_counter()
    keys = ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv")
    leader = world == 1 or int(os.environ.get("LOCAL_RANK", "0")) == 0
    share = None
    if world > 1:
        base_shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        share = os.path.join(base_shm, "cgx_bench_corpus_%s_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), cfg["pairs"], cfg["vocab"], args.seed))
    if leader:
        corpus = synth.make_corpus(cfg["pairs"], cfg["vocab"], args.seed, cfg.get("lo", 5), cfg.get("hi", 45))
        if share:
            os.makedirs(share, exist_ok=True)
            for k in keys:
                np.save(os.path.join(share, k + ".npy"), corpus[k])
    if multi:
        dist.barrier()
        if not leader:
            corpus = {k: np.load(os.path.join(share, k + ".npy"), mmap_mode="r") for k in keys}
            corpus.update(pairs=cfg["pairs"], vocab=cfg["vocab"])
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                                   corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    n_src_tokens = int(len(corpus["str"]))
    t_gen = time.perf_counter() - t0

    # ---- index: built on rank 0, broadcast once over RCCL/xGMI, derived tables rebuilt per rank ----
    t0 = time.perf_counter(); t_bcast = 0.0; bcast_bytes = 0
    if rank == 0:
        ex.upload_corpus(host)
    if multi:
        meta = [None]
        if rank == 0:
            meta = [ex.index_shape()]
        dist.broadcast_object_list(meta, src=0)
        if rank != 0:
            ex.index_alloc(meta[0])
        bcast_bytes = sum(nb for _, nb in ex.index_buffers())
        if args.bcast == "c":
            # the C path of INTEGRATION.md: an RCCL communicator of our own (unique id passed through the process group), then ONE call
            import ctypes as C
            try:
                rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
            except OSError:
                rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]
            uid = UniqueId()
            rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
                raise SystemExit("ncclGetUniqueId failed")
            box = [C.string_at(C.byref(uid), 128)] if rank == 0 else [None]      # all 128 bytes (a c_char array read as a field stops at the first NUL: found by the one-rank test)
            dist.broadcast_object_list(box, src=0)
            assert len(box[0]) == 128
            C.memmove(C.byref(uid), box[0], 128)
            comm = C.c_void_p()
            rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            if rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) != 0:
                raise SystemExit("ncclCommInitRank failed")
            torch.cuda.synchronize(); dist.barrier(); tb = time.perf_counter()
            ex._chk(ex.lib.cgx_broadcast_index(ex.h, comm, 0, rank), "cgx_broadcast_index")     # non-root ranks finalize inside
            torch.cuda.synchronize(); dist.barrier(); t_bcast = time.perf_counter() - tb
            rccl.ncclCommDestroy.argtypes = [C.c_void_p]; rccl.ncclCommDestroy(comm)
        else:
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
            torch.cuda.synchronize(); dist.barrier(); t_bcast = time.perf_counter() - tb
            if rank != 0:
                ex.index_finalize()
        torch.cuda.synchronize(); dist.barrier()
    t_index = time.perf_counter() - t0

    # ---- queries: weak = every rank gets cfg.queries; strong = cfg.queries in all, split by token count ----
    # Several distinct batches (seeds s, s + 1000, ..): step i runs batch i % nsets.
    global_q = cfg["queries"] * world if cfg["scaling"] == "weak" else cfg["queries"]
    nsets = max(1, args.query_sets); qsets = []
    for k in range(nsets):
        gq_off, gq_tok = synth.make_queries(corpus, global_q, args.seed + 3087 + 1000 * k)
        first_k, qoff_k, qtok_k = shard.take_shard(gq_off, gq_tok, rank, world)
        qsets.append({"first": first_k, "qoff": np.asarray(qoff_k, np.int64), "qtok": qtok_k})
    first, qoff, qtok = qsets[0]["first"], qsets[0]["qoff"], qsets[0]["qtok"]; nq = len(qoff)
    nq_max = max(len(q["qoff"]) for q in qsets)
    if args.cpu_full_queries > 0 and rank == 0 and world == 1 and not args.no_cpu_baseline and cfg["pairs"] >= 1_000_000:
        full_dir = save_corpus_for_cpu(corpus, qsets[0], args)  # the CPU leg on the benchmark's own corpus reads these after the timed region
    else:
        full_dir = None
    del corpus, gq_off, gq_tok
    if multi:
        dist.barrier()
        if share and leader:
            shutil.rmtree(share, ignore_errors=True)            # every rank holds its own copy now

    # ---- spool: one directory per rank, rewritten in place; sized against the memory really available ----
    # per rank: ~12 GB fixed (host corpus copy, torch, staging) + the spool (one chunk's files) + the writer's two page-locked
    # buffer sets (unique text of a chunk x 1.25 x 2 = about 1.6x the chunk's files)
    budget, mem_avail, mem_cgroup = memory_budget()
    per_rank = max(0.0, args.mem_fraction * budget / max(local_world, 1) - 12e9) / 2.6
    cands = [args.outdir] if args.outdir else [d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d) and os.access(d, os.W_OK)]
    base = None
    for d in cands:
        try:
            if shutil.disk_usage(d).free / max(local_world, 1) >= min(per_rank, nq_max * EST_BYTES_PER_QUERY):
                base = d; break
        except OSError:
            pass
    write = not args.no_write and base is not None
    chunk = nq_max
    if write:
        fit = int(per_rank / (EST_BYTES_PER_QUERY * 1.15))
        chunk = args.chunk_queries if args.chunk_queries > 0 else min(nq_max, fit)
        if chunk < min(nq_max, 64):
            write = False                                       # no room for a useful spool: count the rules on the GPU instead (reported)
            chunk = nq_max
    spool = tempfile.mkdtemp(prefix="cgx_bench_r%d_" % rank, dir=base) if write else None
    # a chunk is also at most one internal batch of the library (300 000 query tokens by default): the per-batch timers and
    # tallies read after each chunk then describe the whole chunk, and every configuration runs the same pipeline
    if args.sub_batch > 0:
        chunk = min(chunk, args.sub_batch)                      # smaller batches are submitted as chunks here, so that the per-batch timers below still describe what ran
    for q in qsets:
        q["chunks"] = make_chunks(q["qoff"], len(q["qtok"]), chunk)
    chunks = qsets[0]["chunks"]
    chunk = max(b - a for q in qsets for a, b in q["chunks"])
    whole = all(len(q["chunks"]) == 1 for q in qsets)
    spool_bytes = [0]

    stage_names = ("sa_lookup", "blocks", "gappy", "extract", "lexicon", "format", "fmt_count", "fmt_write", "look1_kernel", "look2_kernel", "select_hits", "select_long1", "select_long2", "select_huge1", "select_huge2", "sort_lists1", "sort_lists2")
    host_names = ("write", "write_wait_d2h", "write_file", "total", "t_upload_sa", "t_blocks", "t_gappy", "t_extract", "t_lexicon", "t_format", "t_flush_wait")
    def new_acc():
        return {"on": False, "kernel_ms": [], "stage": {k: 0.0 for k in stage_names}, "host": {k: 0.0 for k in host_names},
                "look1_items": 0.0, "look2_items": 0.0, "h1": 0, "h2": 0, "batches": 0, "ubytes": 0.0, "pbytes": 0.0, "fbytes": 0.0}      # summed over every batch (chunk) of the timed steps
    acc = new_acc()
    step_no = [0]

    def writer_totals():
        """The writer's running totals (ms) over every batch it has finished: read after ex.flush() on both sides of a region, the
        differences are the region's own DMA waits and file phases (the per-batch figures lag two batches behind the submissions)."""
        return {k: max(ex.host_ms(k + "_sum"), 0.0) for k in ("write", "write_wait_d2h", "write_file")}

    def run_chunk(a, b, outdir=None, qs=None):
        # With several chunks per step the chunks reuse the file slots grammar.0.s ..: first_query_index is 0 for each of them, so the
        # library's write_period / write_count sampling (an index into the whole query list) would be chunk-local here; the bench does not use it.
        qs = qs or qsets[0]; qo, qt = qs["qoff"], qs["qtok"]
        t0_, t1_ = int(qo[a]), (int(qo[b]) if b < len(qo) else len(qt))
        n = ex.extract_grammars_ids(host, (qo[a:b] - t0_).astype(np.int32), qt[t0_:t1_], outdir or spool, qs["first"] + a if whole else 0)
        if acc["on"]:                                         # the per-batch timers and tallies of the library hold the batch that just ran
            acc["kernel_ms"].append(ex.stage_ms("sa_lookup_kernel"))
            for k in stage_names: acc["stage"][k] += max(ex.stage_ms(k), 0.0)
            for k in host_names: acc["host"][k] += max(ex.host_ms(k), 0.0)
            cc = ex.counts(); acc["look1_items"] += max(ex.stage_ms("look1_items"), 0.0); acc["look2_items"] += max(ex.stage_ms("look2_items"), 0.0)
            acc["h1"] += cc["h1"]; acc["h2"] += cc["h2"]; acc["batches"] += 1
            acc["ubytes"] += max(ex.stage_ms("fmt_unique_bytes"), 0.0); acc["pbytes"] += max(ex.stage_ms("fmt_plain_unique_bytes"), 0.0); acc["fbytes"] += max(ex.stage_ms("fmt_file_bytes"), 0.0)
        return n

    def step(outdir_of=None):
        """One step = one query batch; consecutive steps take consecutive batches of the rotation.  -> (rules, queries)"""
        qs = qsets[step_no[0] % nsets]; step_no[0] += 1
        n = 0
        for a, b in qs["chunks"]:
            n += run_chunk(a, b, outdir_of() if outdir_of else None, qs)
        return n, len(qs["qoff"])

    # the CPU baseline has had the corpus generation and the index build to finish; wait for the rest of it now
    cpu_res = collect_cpu_baseline(cpu_proc, 240 + int(args.cpu_full_sa_seconds)) if cpu_proc else None

    # timed steps must not be the first rewrite of the file slots (the first rewrite of a file set costs 4x the page-cache
    # time of any later one): the slots are filled at least twice before the clock starts
    priming = 0
    while write and warmup * len(chunks) + priming < 2:
        run_chunk(*chunks[0]); priming += 1
    for _ in range(warmup):
        step()
    ex.flush()
    if write:
        spool_bytes[0] = sum(e.stat().st_size for e in os.scandir(spool) if e.is_file())
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    reruns0 = max(ex.stage_ms("append_reruns"), 0.0)
    wt0 = writer_totals()
    t0 = time.perf_counter(); rules = 0; queries_done = 0
    acc["on"] = True
    for _ in range(steps):
        r_, q_ = step(); rules += r_; queries_done += q_
    acc["on"] = False
    kernel_ms, stage, hoststage = acc["kernel_ms"], acc["stage"], acc["host"]
    ex.flush()                                # every grammar file of every timed step is on disk before the clock stops
    wt1 = writer_totals()
    for k in wt0: hoststage[k] = wt1[k] - wt0[k]          # the timed steps' own batches
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist if multi else None)
    append_reruns = max(ex.stage_ms("append_reruns"), 0.0) - reruns0
    total_q = shard.sum_over_ranks(queries_done, dist if multi else None)
    total_rules = shard.sum_over_ranks(rules, dist if multi else None)

    # ---- what every rank did in the timed steps: the GPU chain alone, and the two host stages beside it ----
    chain_s = sum(stage[k] for k in ("sa_lookup", "blocks", "gappy", "extract", "lexicon", "format")) / 1e3
    mine = {"rank": rank, "queries": int(queries_done), "gpu_chain_s": chain_s, "dma_wait_s": hoststage["write_wait_d2h"] / 1e3, "file_phase_s": hoststage["write_file"] / 1e3,
            "writer_threads": int(max(ex.host_ms("writer_threads"), 0)), "cpus_usable": usable_cpus(), "cgx_threads_env": os.environ.get("CGX_THREADS")}
    ranks = [mine]
    if multi:
        ranks = [None] * world; dist.all_gather_object(ranks, mine)

    # ---- the same steps into FRESH directories (what a user who writes every batch into a new directory sees): every chunk
    # goes into a directory of its own; the directory from three chunks back is deleted by a helper thread (the library keeps
    # at most two batches in flight, so its files are complete), which bounds the memory at four generations ----
    fresh_steps = args.fresh_steps if args.fresh_steps is not None else (3 if len(chunks) == 1 else 1)
    fresh = {"steps": 0, "value": None, "ms_per_step": None, "note": None}
    if write and fresh_steps > 0:
        import threading
        gen_bytes = max(spool_bytes[0], 1)
        shutil.rmtree(spool, ignore_errors=True); os.makedirs(spool, exist_ok=True)     # the in-place spool has done its work: its memory goes to the generations
        fits = gen_bytes * 5.6 <= per_rank * 2.6                                         # three live generations + one being deleted + the writer's page-locked buffers (1.6)
        fits = bool(shard.min_over_ranks(1.0 if fits else 0.0, dist if multi else None))      # all ranks or none (the region has barriers)
        if not fits:
            fresh["note"] = "skipped: four generations of %.1f GB of files do not fit this rank's memory budget" % (gen_bytes / 1e9)
        else:
            gens, deleters = [], []
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            tf = time.perf_counter()
            fresh_q = 0

            def fresh_dir():
                while len(deleters) > 1:
                    deleters.pop(0).join()
                if len(gens) > 2:                               # the library keeps at most two batches in flight: the files of the third-last chunk are complete
                    th = threading.Thread(target=shutil.rmtree, args=(gens.pop(0), True)); th.start(); deleters.append(th)
                d = tempfile.mkdtemp(prefix="gen_", dir=spool); gens.append(d)
                return d
            for _ in range(fresh_steps):
                fresh_q += step(fresh_dir)[1]
            ex.flush()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dtf = shard.max_over_ranks(time.perf_counter() - tf, dist if multi else None)
            for th in deleters:
                th.join()
            tq = shard.sum_over_ranks(fresh_q, dist if multi else None)
            fresh.update(steps=fresh_steps, value=round(tq / dtf, 3), ms_per_step=round(dtf / fresh_steps * 1e3, 3),
                         note="every chunk written into a new directory; directories older than two chunks deleted by a helper thread inside the timed region")

    # ---- the same steps with grammar.<q>.s.gz output: the gzip members are made by the GPU formatter, so PCIe and the file phase
    # move a third of the bytes (the plain files above stay the default and the headline) ----
    gz_steps = args.gz_steps if args.gz_steps is not None else steps
    gzres = {"steps": 0, "value": None}
    if write and gz_steps > 0:
        shutil.rmtree(spool, ignore_errors=True); os.makedirs(spool, exist_ok=True)
        ex.set_option("gz_level", 1)
        main_acc = acc; acc = new_acc()
        for _ in range(2):                                    # fill the .gz file slots twice before the clock starts, as for the plain files
            step()
        ex.flush()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        wg0 = writer_totals()
        tg = time.perf_counter(); gz_rules = 0; gz_q = 0
        acc["on"] = True
        for _ in range(gz_steps):
            r_, q_ = step(); gz_rules += r_; gz_q += q_
        acc["on"] = False
        ex.flush()
        wg1 = writer_totals()
        for k in wg0: acc["host"][k] = wg1[k] - wg0[k]
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dtg = shard.max_over_ranks(time.perf_counter() - tg, dist if multi else None)
        tqg = shard.sum_over_ranks(gz_q, dist if multi else None); trg = shard.sum_over_ranks(gz_rules, dist if multi else None)
        gacc = acc; acc = main_acc
        ex.set_option("gz_level", 0)
        gchain = sum(gacc["stage"][k] for k in ("sa_lookup", "blocks", "gappy", "extract", "lexicon", "format")) / 1e3
        gchain_max = shard.max_over_ranks(gchain, dist if multi else None)
        gzres = {"steps": gz_steps, "value": round(tqg / dtg, 3), "rules_per_s": round(trg / dtg, 1), "ms_per_step": round(dtg / gz_steps * 1e3, 3),
                 "value_gpu_chain": round(tqg / max(gchain_max, 1e-9), 3),
                 "gpu_chain_ms_per_step": round(gchain / gz_steps * 1e3, 2), "dma_wait_ms_per_step": round(gacc["host"]["write_wait_d2h"] / gz_steps, 2), "file_phase_ms_per_step": round(gacc["host"]["write_file"] / gz_steps, 2),
                 "format_ms_per_step": round(gacc["stage"]["format"] / gz_steps, 2), "fmt_count_ms_per_step": round(gacc["stage"]["fmt_count"] / gz_steps, 2), "fmt_write_ms_per_step": round(gacc["stage"]["fmt_write"] / gz_steps, 2),
                 "d2h_bytes_per_step": int(gacc["ubytes"] / gz_steps), "plain_unique_text_bytes_per_step": int(gacc["pbytes"] / gz_steps), "gz_file_bytes_per_step": int(gacc["fbytes"] / gz_steps),
                 "compressed_to": round(gacc["ubytes"] / max(gacc["pbytes"], 1.0), 4),
                 "huffman_codes": "dynamic (made for each batch from a tally of its symbols)" if ex.stage_ms("fmt_gz_dynamic") == 1.0 else "fixed (RFC 1951 3.2.6)", "block_header_bits_per_group": int(max(ex.stage_ms("fmt_gz_header_bits"), 0)),
                 "stages_ms_per_step": {k: round(v / gz_steps, 2) for k, v in list(gacc["stage"].items()) + [("host_" + k, v) for k, v in gacc["host"].items()]},
                 "bound": max((("gpu_chain", gchain), ("dma", gacc["host"]["write_wait_d2h"] / 1e3), ("file_phase", gacc["host"]["write_file"] / 1e3)), key=lambda t: t[1])[0],
                 "note": "this rank's figures; files grammar.<q>.s.gz rewritten in place in the spool directory, slots filled twice before the clock starts; every emission group one DEFLATE block made by the GPU formatter (back-references from the line structure, Huffman codes made for the batch), every piece of the unique text a byte-aligned stretch of a deflate stream, every file one gzip member: header + pieces + trailer (CRC-32 / ISIZE folded on the device)"}

    # ---- the .gz steps again with TWO contexts over the one index (cgx_share_index), each driven by a thread of its own: two batches in
    # flight on the card.  The kernels of a batch wait most of their wave cycles (lookups 65-72 %, MaxLex 80 %); with a second batch the
    # card has other waves to run meanwhile.  Whole batches, not halves: the work of a batch is shared among its queries. ----
    two = {"steps": 0, "value": None}
    two_steps = args.two_context_steps or 0                   # off by default: measured at 1.01x for the .gz steps (1.07x for the GPU chain alone, 1.20x as two PROCESSES: HISTORY.md)
    if write and gz_steps > 0 and two_steps >= 2:
        import threading
        shutil.rmtree(spool, ignore_errors=True); os.makedirs(spool, exist_ok=True)
        spool2 = tempfile.mkdtemp(prefix="cgx_bench_r%d_b_" % rank, dir=base)
        ex2 = cgx_amd.Extractor(local)
        try:
            if args.sub_batch: ex2.set_option("sub_batch", args.sub_batch)
            if args.no_numa_pin: ex2.set_option("numa_pin", 0)
            if not args.sync_write: ex2.set_option("async_write", 1); ex2.set_option("prealloc_text", 1)
            for ov in args.option:
                k, v = ov.split("="); ex2.set_option(k, int(v))
            ex2.share_index(ex)
            lanes = [(ex, spool), (ex2, spool2)]
            for e_, _ in lanes: e_.set_option("gz_level", 1)
            chain_ms = [0.0, 0.0]; done = [0, 0]; rules2 = [0, 0]; errs = []

            def lane_batch(i, k, timed):
                e_, d_ = lanes[i]; qs = qsets[k % nsets]; qo, qt = qs["qoff"], qs["qtok"]
                for a, b in qs["chunks"]:
                    t0_, t1_ = int(qo[a]), (int(qo[b]) if b < len(qo) else len(qt))
                    n_ = e_.extract_grammars_ids(host, (qo[a:b] - t0_).astype(np.int32), qt[t0_:t1_], d_, qs["first"] + a if whole else 0)
                    if timed:
                        rules2[i] += n_; chain_ms[i] += sum(max(e_.stage_ms(k2), 0.0) for k2 in ("sa_lookup", "blocks", "gappy", "extract", "lexicon", "format"))
                if timed: done[i] += len(qo)

            def lane(i, ks, timed):
                try:
                    for k in ks: lane_batch(i, k, timed)
                    lanes[i][0].flush()
                except Exception as e_:                              # noqa: BLE001
                    errs.append(repr(e_))

            def both(ks0, ks1, timed):
                ths = [threading.Thread(target=lane, args=(0, ks0, timed)), threading.Thread(target=lane, args=(1, ks1, timed))]
                for th in ths: th.start()
                for th in ths: th.join()
            both([0, 2], [1, 3], False)                              # each lane fills its file slots twice and sizes its buffers
            if multi: dist.barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            both(list(range(0, two_steps, 2)), list(range(1, two_steps, 2)), True)
            torch.cuda.synchronize()
            if multi: dist.barrier()
            dt2 = shard.max_over_ranks(time.perf_counter() - t2, dist if multi else None)
            if errs: raise RuntimeError("; ".join(errs))
            tq2 = shard.sum_over_ranks(sum(done), dist if multi else None); tr2 = shard.sum_over_ranks(sum(rules2), dist if multi else None)
            two = {"steps": two_steps, "value": round(tq2 / dt2, 3), "rules_per_s": round(tr2 / dt2, 1), "ms_per_step": round(dt2 / two_steps * 1e3, 3),
                   "gpu_chain_ms_per_batch_while_sharing_the_card": [round(chain_ms[i] / max(len(range(i, two_steps, 2)), 1), 2) for i in (0, 1)],
                   "hbm_in_use_gb": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),
                   "note": "this rank: two contexts over one index (cgx_share_index), two host threads, steps dealt alternately, grammar.<q>.s.gz into one spool directory per context, slots filled twice before the clock starts, both contexts flushed before it stops"}
        except Exception as e_:                                      # noqa: BLE001 -- the leg is an extra: report, do not lose the line
            two = {"steps": 0, "value": None, "note": "failed: %r" % (e_,)}
        finally:
            for e_ in (ex, ex2):
                try: e_.set_option("gz_level", 0)
                except Exception: pass
            ex2.close(); shutil.rmtree(spool2, ignore_errors=True)

    if full_dir is not None:                                  # after every timed region: the CPUs and the memory bandwidth are free
        full_res = run_cpu_full_corpus(ex, full_dir, args)
        if cpu_res is not None:
            cpu_res["full_corpus"] = full_res

    if rank == 0:
        line = {"metric": "query sentences/sec", "value": round(total_q / dt, 3), "unit": "query sentences/s", "rules_per_s": round(total_rules / dt, 1),
                "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / max(steps, 1) * 1e3, 3), "higher_is_better": True,
                "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "i32", "data": "synthetic",
                # value = files rewritten in place in one spool directory per rank (page-cache friendly); the same path into new directories:
                "value_fresh_files": fresh["value"], "fresh_files": fresh,
                # the same steps writing grammar.<q>.s.gz, the gzip members made by the GPU formatter
                "value_gz": gzres["value"], "gz": gzres,
                # the .gz steps with two batches in flight: a second context over the same index, driven by a second thread
                "value_gz_two_contexts": two["value"], "gz_two_contexts": two,
                "query_batches_rotated": nsets, "append_pass_reruns_per_step": round(append_reruns / max(steps, 1), 3),
                # the GPU stages alone (lookup .. text layout, hipEvent-timed per stage), without DMA and file phases: what scales with the GPU count by construction
                "value_gpu_chain": round(sum(r["queries"] for r in ranks) / max(max(r["gpu_chain_s"] for r in ranks), 1e-9), 3),
                "per_rank": {"gpu_chain_ms_per_step": [round(min(r["gpu_chain_s"] for r in ranks) / max(steps, 1) * 1e3, 2), round(max(r["gpu_chain_s"] for r in ranks) / max(steps, 1) * 1e3, 2)],
                             "dma_wait_ms_per_step": [round(min(r["dma_wait_s"] for r in ranks) / max(steps, 1) * 1e3, 2), round(max(r["dma_wait_s"] for r in ranks) / max(steps, 1) * 1e3, 2)],
                             "file_phase_ms_per_step": [round(min(r["file_phase_s"] for r in ranks) / max(steps, 1) * 1e3, 2), round(max(r["file_phase_s"] for r in ranks) / max(steps, 1) * 1e3, 2)],
                             "writer_threads": [min(r["writer_threads"] for r in ranks), max(r["writer_threads"] for r in ranks)],
                             "cpus_usable_per_process": [min(r["cpus_usable"] for r in ranks), max(r["cpus_usable"] for r in ranks)],
                             "cgx_threads_env": mine["cgx_threads_env"], "note": "[min, max] over the ranks; per step of the timed region"}}
        try:
            line.update(report(ex, args, cfg, locals()))
        except Exception as e:                                   # the measurement above stands whatever the extras do
            line["report_error"] = repr(e)
        print(json.dumps(line)); sys.stdout.flush()
    if spool:
        shutil.rmtree(spool, ignore_errors=True)
    ex.close()
    if multi:
        dist.barrier(); dist.destroy_process_group()


def report(ex, args, cfg, L):
    """Everything of the JSON line beyond the contract's core fields (rank 0, after the clock has stopped)."""
    import torch
    steps, world, nq, qtok, n_src = L["steps"], L["world"], L["nq"], L["qtok"], L["n_src_tokens"]
    stage, hoststage = L["stage"], L["hoststage"]
    free_b, total_b = torch.cuda.mem_get_info()                # after the timed steps: index + cached batch buffers + both text slots
    c = ex.counts()                                            # of the last batch, before the extra lookup below resets them
    nch = len(L["chunks"]); acc = L["acc"]; nb = max(acc["batches"], 1)
    w1, w2 = acc["look1_items"] / nb, acc["look2_items"] / nb  # occurrences walked, hits appended and kernel time: means over the batches of the timed steps
    # ---- the batched SA interval search, priced by the probes it executes (counted by the kernel itself, untimed launch) ----
    kms = float(np.mean(L["kernel_ms"])) if L["kernel_ms"] else 0.0
    a, b = L["chunks"][-1]
    t0_, t1_ = int(L["qoff"][a]), (int(L["qoff"][b]) if b < nq else len(qtok))
    ex.set_option("count_probes", 1)
    ex.upload_queries((L["qoff"][a:b] - t0_).astype(np.int32), qtok[t0_:t1_]); ex.sa_lookup()
    ex.set_option("count_probes", 0)
    lm = ex.fetch("lm"); T = len(lm)
    pb, ps, pq, plk = (ex.stage_ms("sa_probe_" + k) for k in ("bucket", "slots", "search", "lookups"))
    kms_last = ex.stage_ms("sa_lookup_kernel")                # the same launch (the last chunk of a step), timed by its own events
    # bytes the kernel must move for these lookups: its query tokens (4 B each + 12 B of offsets), 4 B per bucket-table entry,
    # 16 B per l-gram slot, 8 B per search probe (SA entry + corpus token), 44 B of results per token
    slot_b = 16.0                                            # one l-gram slot {key, lo, hi}; it arrives in a 64-byte sector, which is what the counters see
    exec_bytes = T * (4 + 12 + 44) + 4 * pb + slot_b * ps + 8 * pq
    sv_bytes, lookups = survey_bytes(n_src, lm)
    traffic = None; requests = None; rr16 = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_k_sa_lookup.json")))
        pc = pmc["config"]
        if (pc["pairs"], pc["queries"], pc["seed"], pc["vocab"]) == (cfg["pairs"], nq, args.seed, cfg["vocab"]) and len(L["chunks"]) == 1:
            traffic = int(pmc["traffic_bytes"]); requests = float(pmc["TCC_EA0_RDREQ_per_launch"])
            rr16 = json.load(open(os.path.join(ROOT, "profiles", "pmc_lookup_kernels.json")))["random_read_peak"]["reads_per_s_16B"]
    except Exception:
        traffic = None
    kuse = kms if kms > 0 else kms_last                          # the mean over the launches of the timed steps (a single relaunch after the run is 2-3x slower: cold tables)
    ach = exec_bytes / (kuse * 1e-3) / 1e9 if kuse > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_sa_lookup (batched SA interval search, K1+K2)", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(ach / 8000.0, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(exec_bytes), "kernel_ms": round(kuse, 5), "kernel_ms_mean_timed_steps": round(kms, 5), "kernel_ms_single_relaunch_after_run": round(kms_last, 5),
                "traffic_source": (None if traffic is None else "profiles/pmc_k_sa_lookup.json (rocprofv3 --pmc passes of the same kernel on the same workload, committed) -- NOT measured in this run"),
                "units_per_launch": {"query_tokens": int(T), "lookups": int(plk), "bucket_entries_read": int(pb), "ngram_slots_read": int(ps), "search_probes": int(pq)},
                "lookups_per_s": round(plk / (kuse * 1e-3), 1) if kuse > 0 else None,
                "traffic_GBps": (round(traffic / (kuse * 1e-3) / 1e9, 1) if traffic and kuse > 0 else None),
                "survey_8d_formula": {"bytes": int(sv_bytes), "GBps": round(sv_bytes / (kuse * 1e-3) / 1e9, 1) if kuse > 0 else None,
                                      "note": "what the reference's full-depth binary search would touch for the same lookups (SURVEY 8d); not a fraction of anything this kernel moves"},
                "random_read_requests_per_launch": requests, "random_read_peak_per_s_16B": rr16,
                "frac_of_random_read_peak": (round(requests / (kuse * 1e-3) / rr16, 3) if requests and rr16 and kuse > 0 else None),
                "target_60pct_met": bool(ach / 8000.0 >= 0.6),
                "note": "achieved = bytes of the probes the kernel executed (counted by the kernel) / its event-timed duration, against the 8 TB/s streaming peak (target 0.6: see target_60pct_met); the probes are scattered 16-byte reads, so the bound that applies is the card's random-read request rate (tools/micro/gather_bw, 64-byte reads): frac_of_random_read_peak"}
    # ---- the kernels that take the most time per step, priced per corpus occurrence they visit ----
    k1, k2 = stage["look1_kernel"] / nb, stage["look2_kernel"] / nb
    by_time = []
    try:                                                        # counter traffic of the same kernels on the same workload (committed PMC passes) and the card's random-read rate
        pk = json.load(open(os.path.join(ROOT, "profiles", "pmc_lookup_kernels.json")))
        pc = pk["config"]
        if (pc["pairs"], pc["queries"], pc["seed"], pc["vocab"]) != (cfg["pairs"], nq, args.seed, cfg["vocab"]) or nch != 1:
            pk = None
    except Exception:
        pk = None
    for name, w, per, hits, ms in (("k_look1 (one-gap corpus lookups)", w1, ex.stage_ms("look1_bytes_per_item"), acc["h1"] / nb, k1), ("k_look2 (two-gap corpus lookups)", w2, ex.stage_ms("look2_bytes_per_item"), acc["h2"] / nb, k2)):
        if w > 0 and ms > 0 and per > 0:
            ab = w * per + hits * 8
            e = {"kernel": name, "bound": "hbm", "occurrences_per_launch": int(w), "bytes_per_occurrence": per, "algorithmic_bytes_per_launch": int(ab), "ms_per_launch": round(ms, 3),
                 "achieved": round(ab / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ab / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None}
            kk = pk["kernels"].get(name.split(" ")[0]) if pk else None
            if kk:                                               # these kernels gather one unaligned 128-byte window per occurrence at random: the card's measured rate of random 128-byte runs is the bound that applies
                # Counters summed over the batch's launches of this kernel (tile chunks).  FETCH_SIZE = read requests x 64 B, but a
                # request is 64 or 128 bytes (calibrated on known byte counts, profiles/r2bd_fetch_size_calibration_gather.txt):
                # `traffic` applies the request mix of this access pattern (an unaligned 128-byte window = one 128-byte and one
                # 64-byte request), the raw counter figure and the two bounds are given beside it.
                cal = pk.get("calibration", {}); bpr = float(cal.get("estimated_bytes_per_request", {}).get(name.split(" ")[0], 64.0))
                raw = (kk["TCC_EA0_RDREQ_per_batch"] + kk["TCC_EA0_WRREQ_per_batch"]) * 64.0
                tr = kk["TCC_EA0_RDREQ_per_batch"] * bpr + kk["TCC_EA0_WRREQ_per_batch"] * 64.0
                e.update(traffic=int(tr), traffic_GBps=round(tr / (ms * 1e-3) / 1e9, 1), sectors_per_occurrence=round(kk["TCC_EA0_RDREQ_per_batch"] / w, 2),
                         traffic_counter_raw=int(raw), traffic_bounds=[int(raw), int(kk["TCC_EA0_RDREQ_per_batch"] * 128.0 + kk["TCC_EA0_WRREQ_per_batch"] * 64.0)], bytes_per_read_request_assumed=bpr,
                         traffic_frac_of_peak=round(tr / (ms * 1e-3) / 1e9 / 8000.0, 4),     # what the HBM actually moved for this kernel, as a fraction of 8 TB/s
                         windows_per_s=round(w / (ms * 1e-3), 1), microbench_random_128B_runs_per_s=pk["random_read_peak"]["runs_per_s_128B"],   # tools/micro/gather_bw on the same card, one run per lane: a comparison, not a bound (neighbouring occurrences share sectors in the L2)
                         microbench_random_128B_runs_per_s_eight_lanes_per_run=pk["random_read_peak"].get("runs_per_s_128B_eight_lanes_per_run_8B_offsets"),   # tools/micro/gather_coop: the way the kernels read their windows now (runs at random 8-byte offsets)
                         dram_read_requests_per_s=round(kk["TCC_EA0_RDREQ_per_batch"] / (ms * 1e-3), 1))   # against 3.0e10 (64-byte) .. 5.2e10 (16-byte) random reads per second of the card
                im = pk.get("issue_model")
                if im and kk.get("SQ_INSTS_VALU"):               # the other resource these kernels load: instruction issue (round 4: a quarter fewer DRAM requests changed nothing, HISTORY.md)
                    e["issue"] = {"vector_instructions_per_launch": kk["SQ_INSTS_VALU"], "scalar_instructions_per_launch": kk.get("SQ_INSTS_SALU"),
                                  "frac_of_vector_issue_peak": round(kk["SQ_INSTS_VALU"] * im["cycles_per_vector_instruction"] / (im["simds"] * im["shader_clock_hz"] * ms * 1e-3), 3),
                                  "frac_of_scalar_issue_peak": (round(kk["SQ_INSTS_SALU"] / (im["scalar_units"] * im["shader_clock_hz"] * ms * 1e-3), 3) if kk.get("SQ_INSTS_SALU") else None),
                                  "wave_cycles_waiting": kk.get("SQ_WAIT_ANY_frac"),
                                  "note": "counts from the committed counter passes (pmc_lookup_kernels.json), duration from this run: vector issue, scalar issue and the random-read rate are each about half used -- the kernel alternates between them inside every wave (window round trip, 13-move walk, candidate phase) at six waves per SIMD"}
            by_time.append(e)
    by_time.sort(key=lambda e: -e["ms_per_launch"])              # roofline_dominant = the kernel that takes the most time per batch
    gb = lambda k: round(max(ex.stage_ms(k), 0.0) / 1e9, 2)
    hbm = {"index_as_broadcast": gb("mem_index"), "derived_tables": gb("mem_derived"), "derived_ngram_tables": gb("mem_derived_ngram_tables"), "derived_interleaved_layouts_and_target_blocks": gb("mem_derived_layouts"),
           "derived_lexical_pair_hash": gb("mem_derived_lex_hash"), "text_slots_and_piece_lists": gb("mem_text"), "last_batch_results": gb("mem_batch"),
           "allocator_idle_blocks_kept_for_the_next_batch": gb("mem_cached"), "other_live_blocks": gb("mem_other")}
    hbm["hip_runtime_torch_and_unaccounted"] = round((total_b - free_b) / 1e9 - sum(hbm[k] for k in ("index_as_broadcast", "derived_tables", "text_slots_and_piece_lists", "last_batch_results", "allocator_idle_blocks_kept_for_the_next_batch", "other_live_blocks")), 2)
    hbm["note"] = "GB, after the timed steps; idle blocks = the high-water mark of a batch's temporaries (lookup outputs, sort buffers, hit lists) held by the caching allocator for reuse"
    div = steps * 1.0
    out = {
        "config": {"workload": cfg["what"], "name": args.config, "sentence_pairs": cfg["pairs"], "source_tokens": n_src, "vocab": cfg["vocab"],
                   "queries_per_step_this_rank": int(nq), "query_tokens_per_step_this_rank": int(len(qtok)), "global_queries_per_step": int(L["global_q"]),
                   "parallelism": "query-shard x%d, index replicated (one RCCL broadcast, %s)" % (world, "cgx_broadcast_index" if args.bcast == "c" else "torch.distributed"),
                   "grammar_files_written": bool(L["write"]), "grammar_bytes_per_spool_fill": int(L["spool_bytes"][0]),
                   "writer": None if not L["write"] else ("sync" if args.sync_write else "async (host threads overlap the next chunk; flushed before the clock stops)"),
                   "writer_threads": None if not L["write"] else int(max(ex.host_ms("writer_threads"), 0)),
                   "outdir": L["base"] if L["write"] else None,
                   "outdir_mode": None if not L["write"] else ("one spool directory per rank rewritten in place by every step, %d untimed priming fill(s) beyond the warm-up; %d chunk(s) of <= %d queries per step%s"
                                                               % (L["priming"], nch, L["chunk"], "" if L["whole"] else ", chunks reuse the file slots grammar.0.s .. (first_query_index = 0)")),
                   "host_memory": {"MemAvailable": L["mem_avail"], "cgroup_headroom": L["mem_cgroup"], "spool_budget_per_rank": int(L["per_rank"]), "cpus_usable": usable_cpus()}},
        "roofline": roofline,
        "roofline_dominant": by_time[0] if by_time else None,
        "roofline_other_kernels": by_time[1:],
        "stages_ms_per_step": {k: round(v / div, 3) for k, v in {**stage, **{"host_" + k: v for k, v in hoststage.items()}}.items()},
        "index": {"build_sa_ms": round(ex.stage_ms("build_sa"), 1), "precompute_ms": round(ex.stage_ms("precompute"), 1), "ngram_tables_ms": round(ex.stage_ms("ngrams"), 1),
                  "ngram_table_bytes": int(max(ex.stage_ms("ngram_table_bytes"), 0)), "broadcast_s": round(L["t_bcast"], 3), "broadcast_bytes": int(L["bcast_bytes"]),
                  "total_s": round(L["t_index"], 2), "corpus_gen_s": round(L["t_gen"], 2), "frequent_pair_hits": c["nphits"]},
        "hbm_in_use_gb": round((total_b - free_b) / 1e9, 1), "hbm_breakdown": hbm,
        "counts": {**{k: c[k] for k in ("d1", "d2", "h1", "h2", "g", "n0", "n1", "n2", "guard_exits")},
                   "lexicon_lines": int(max(ex.stage_ms("lex_lines"), 0)), "grammar_lines": int(max(ex.stage_ms("fmt_nlines"), 0)), "emission_items": int(max(ex.stage_ms("fmt_items"), 0)),
                   "unique_text_bytes": int(max(ex.stage_ms("fmt_unique_bytes"), 0)), "segments": int(max(ex.stage_ms("fmt_segments"), 0))},
    }
    if L["cpu_res"] is not None:
        out["cpu_baseline"] = L["cpu_res"]
    return out


if __name__ == "__main__":
    main()
