#!/usr/bin/env python3
"""How much grammar text can ONE host write when R ranks write at the same time?

Every rank of a multi-GPU run ends its batch with the file phase: host threads assemble the per-query grammar files from the
unique text and the piece lists the GPU produced (cgx_host.c: dev_write_files -> write_one_file, one pwritev per <= 1024
pieces).  On an 8-GPU node the eight ranks share the host's CPUs and memory bandwidth, so the file phase is what bounds
query-shard scaling.  A one-GPU box cannot run eight GPU ranks, but it can run eight WRITERS:

  1. (GPU, once)   one real batch of --queries sentences on the --pairs corpus goes through lookup .. text layout; its
                   unique text and piece lists are recorded in --dir;
  2. (CPU only)    for R in --ranks: R fresh processes (they never touch the GPU) each load a private copy of the record
                   and replay the file phase through the library's own cgx_assemble_files (the same write_one_file the
                   writer runs) with the thread count bench.py would give a rank, first into fresh files, then twice in
                   place; all R start together.

Printed per R: threads per rank, aggregate GB/s (fresh / in place), CPU-seconds per rank and per GB, and the aggregate as a
multiple of the one-rank figure.  One JSON line per R at the end (for profiles/).

    python3 tools/rehearse_writers.py --pairs 10000000 --queries 2500 --ranks 1,2,4,8
    python3 tools/rehearse_writers.py --gz ...      the same batch as deflate pieces (grammar.<q>.s.gz, option gz_level): a third of the bytes

`--chain-ms T` (the GPU chain of the recorded batch on one card, from bench.py) adds the question a scaling run asks: do R writers
finish their batches within T, i.e. does the file phase keep up with R cards?  `keeps_up` = T / (slowest rank's in-place pass).
"""
import argparse
import ctypes as C
import json
import os
import resource
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def record(args):
    """GPU phase: one batch, its unique text and piece lists written to args.dir."""
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd
    from cgx_amd import synth
    corpus = synth.make_corpus(args.pairs, args.vocab, args.seed)
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                                   corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    ex = cgx_amd.Extractor(0); ex.upload_corpus(host)
    qoff, qtok = synth.make_queries(corpus, args.queries, args.seed + 3087)
    # the whole path once (only the first query's file is written: option write_period); the text slot of the batch stays
    # readable afterwards, and the recorder takes the unique text and the piece lists from it exactly as the writer does
    os.makedirs(args.dir, exist_ok=True)
    tmp = os.path.join(args.dir, "rec_out"); os.makedirs(tmp, exist_ok=True)
    ex.set_option("write_period", 1 << 40); ex.set_option("write_count", 1)
    if args.gz:
        ex.set_option("gz_level", 1)
    nlines = ex.extract_grammars_ids(host, qoff, qtok, tmp, 0); ex.flush()
    slot = int(ex.stage_ms("fmt_slot"))
    text, qseg, so, sl, qtext = ex.text(slot, args.queries); nbytes = int(qtext[args.queries])
    enc = ex.text_encoding(slot)
    if enc:
        np.save(os.path.join(args.dir, "trailers.npy"), ex.text_trailers(slot, args.queries))
    plain_unique = int(max(ex.stage_ms("fmt_plain_unique_bytes"), 0))
    shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(args.dir, "utext.bin"), "wb") as f:
        f.write(text)
    np.save(os.path.join(args.dir, "qseg.npy"), qseg); np.save(os.path.join(args.dir, "segoff.npy"), so); np.save(os.path.join(args.dir, "seglen.npy"), sl)
    meta = {"queries": args.queries, "pairs": args.pairs, "unique_text_bytes": len(text), "file_bytes": int(nbytes), "pieces": int(len(so)), "lines": int(nlines), "encoding": int(enc), "plain_unique_text_bytes": plain_unique}
    json.dump(meta, open(os.path.join(args.dir, "meta.json"), "w"))
    ex.close(); host.close()
    print("recorded: %s" % json.dumps(meta), flush=True)


def writer(args):
    """CPU phase, one rank: private copy of the record, three passes of the file phase, started together with the other ranks."""
    lib = C.CDLL(os.path.join(ROOT, "cgx_amd", "libcgx_hip.so"))
    lib.cgx_assemble_files_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p]
    meta = json.load(open(os.path.join(args.dir, "meta.json")))
    trl = np.load(os.path.join(args.dir, "trailers.npy")) if meta.get("encoding") else None
    text = np.fromfile(os.path.join(args.dir, "utext.bin"), dtype=np.uint8)           # private anonymous copy (a rank's page-locked buffer)
    qseg = np.load(os.path.join(args.dir, "qseg.npy")); so = np.load(os.path.join(args.dir, "segoff.npy")); sl = np.load(os.path.join(args.dir, "seglen.npy"))
    out = os.path.join(args.dir, "out_%d" % args.rank); shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
    open(os.path.join(args.dir, "ready_%d_%d" % (args.world, args.rank)), "w").close()
    go = os.path.join(args.dir, "go_%d" % args.world)
    while not os.path.exists(go):
        time.sleep(0.005)
    res = []
    for p in range(3):
        ms = C.c_double(); r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
        rc = lib.cgx_assemble_files_enc(text.ctypes.data, qseg.ctypes.data, so.ctypes.data, sl.ctypes.data, meta["queries"], 0, out.encode(), args.threads, C.byref(ms),
                                        int(meta.get("encoding", 0)), trl.ctypes.data if trl is not None else None)
        dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
        if rc != 0:
            raise SystemExit("cgx_assemble_files_enc failed: %d" % rc)
        res.append({"pass": ("fresh", "first rewrite", "rewrite")[p], "wall_s": dt, "cpu_s": (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime), "t_start": t0, "t_end": t0 + dt})
    print(json.dumps({"rank": args.rank, "passes": res}), flush=True)
    shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10_000_000); ap.add_argument("--vocab", type=int, default=200_000); ap.add_argument("--queries", type=int, default=2500)
    ap.add_argument("--seed", type=int, default=1234); ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--dir", default="/dev/shm/cgx_rehearse"); ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gz", action="store_true", help="record the batch as deflate pieces (gz_level 1, made by the GPU formatter) and write grammar.<q>.s.gz")
    ap.add_argument("--chain-ms", type=float, default=0.0, help="GPU chain of the recorded batch on one card: adds keeps_up = chain / slowest rank's in-place file phase")
    ap.add_argument("--phase", choices=("all", "record", "writer"), default="all"); ap.add_argument("--rank", type=int, default=0); ap.add_argument("--world", type=int, default=1); ap.add_argument("--threads", type=int, default=1)
    args = ap.parse_args()
    if args.phase == "writer":
        return writer(args)
    if args.phase in ("all", "record"):
        # the GPU phase runs in a child: this process, which starts the writers, never initialises the GPU
        subprocess.run([sys.executable, os.path.abspath(__file__), "--phase", "record", "--pairs", str(args.pairs), "--vocab", str(args.vocab), "--queries", str(args.queries), "--seed", str(args.seed), "--dir", args.dir] + (["--gz"] if args.gz else []), check=True) if args.phase == "all" else record(args)
        if args.phase == "record":
            return
    meta = json.load(open(os.path.join(args.dir, "meta.json"))); cpus = usable_cpus(); gb = meta["file_bytes"] / 1e9
    print("host: %d usable CPUs (affinity / cgroup quota); one rank's batch: %.2f GB of files from %.2f GB of unique text in %d pieces" % (cpus, gb, meta["unique_text_bytes"] / 1e9, meta["pieces"]), flush=True)
    base = None; lines = []
    for R in [int(x) for x in args.ranks.split(",")]:
        share = cpus // R
        threads = max(2, min(16, share - 2 if share >= 8 else share)) if R > 1 else max(1, min(16, cpus - 2))      # bench.py's CGX_THREADS rule / the library's default
        for f in os.listdir(args.dir):
            if f.startswith("ready_") or f.startswith("go_"):
                os.unlink(os.path.join(args.dir, f))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--phase", "writer", "--dir", args.dir, "--rank", str(r), "--world", str(R), "--threads", str(threads)], stdout=subprocess.PIPE, text=True) for r in range(R)]
        while sum(os.path.exists(os.path.join(args.dir, "ready_%d_%d" % (R, r))) for r in range(R)) < R:
            if any(p.poll() not in (None, 0) for p in procs):
                raise SystemExit("a writer died")
            time.sleep(0.01)
        open(os.path.join(args.dir, "go_%d" % R), "w").close()
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
        row = {"ranks": R, "threads_per_rank": threads, "cpus_usable": cpus, "file_GB_per_rank": round(gb, 2)}
        for k, name in enumerate(("fresh", "first_rewrite", "rewrite")):
            t0 = min(o["passes"][k]["t_start"] for o in outs); t1 = max(o["passes"][k]["t_end"] for o in outs)      # perf_counter is system-wide (CLOCK_MONOTONIC)
            cpu = sum(o["passes"][k]["cpu_s"] for o in outs)
            row[name] = {"aggregate_GBps": round(R * gb / (t1 - t0), 1), "slowest_rank_s": round(max(o["passes"][k]["wall_s"] for o in outs), 3), "cpu_s_per_rank": round(cpu / R, 2), "GB_per_cpu_s": round(R * gb / cpu, 2)}
        if base is None:
            base = row
        row["aggregate_vs_one_rank"] = {k: round(row[k]["aggregate_GBps"] / base[k]["aggregate_GBps"], 2) for k in ("fresh", "rewrite")}
        row["encoding"] = "deflate pieces (.gz)" if meta.get("encoding") else "plain"
        if args.chain_ms > 0:                                   # R cards each finish a batch every chain_ms: do R writers on this host keep that pace?
            row["keeps_up"] = {k: round(args.chain_ms / 1e3 / row[k]["slowest_rank_s"], 2) for k in ("fresh", "rewrite")}
            row["ranks_served_at_full_gpu_rate"] = {k: round(R * min(1.0, row["keeps_up"][k]), 2) for k in ("fresh", "rewrite")}
        lines.append(row)
        print("R=%d  %2d threads/rank | fresh %6.1f GB/s (%.2fx one rank) | in place %6.1f GB/s (%.2fx) | CPU %.2f s per rank, %.2f GB per CPU-second in place"
              % (R, threads, row["fresh"]["aggregate_GBps"], row["aggregate_vs_one_rank"]["fresh"], row["rewrite"]["aggregate_GBps"], row["aggregate_vs_one_rank"]["rewrite"], row["rewrite"]["cpu_s_per_rank"], row["rewrite"]["GB_per_cpu_s"]), flush=True)
    for row in lines:
        print(json.dumps(row))
    if not args.keep:
        shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
