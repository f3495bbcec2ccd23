#!/usr/bin/env python3
"""CPU baseline leg of bench.py (the only non-test user of oracle/): the CPU restatement of the
hot path timed on the host cores of the box the benchmark runs on.

    python3 tools/cpu_baseline.py --pairs 200000 --vocab 200000 --seed 1234 --budget-s 12 --cores 16

Runs in its OWN process (bench.py starts it before it touches the GPU loop and reads one JSON
object from stdout), so that nothing it does -- memory, a crash, a time-out -- can cost the
benchmark its result line.  It never imports torch or the HIP library.

What is timed (index construction excluded from queries/s, as on the GPU, and reported apart):
  * one core:   `orc_run_all` (lookup -> gappy search -> extraction -> lexicon/features -> files)
                over as many queries of the sample as fit the time budget;
  * all cores:  the same over `cores` forked workers, each with its own share of queries of the
                same size (the index is shared copy-on-write); rate = queries / slowest worker;
  * index:      the oracle's own suffix-array + frequent-pair build on the sample and, when
                oracle/_ref/ref_harness exists (built from /root/reference where that is present),
                the REFERENCE's suffixArrayConstruct (SuffixArray.c:196-242) on the same tokens and
                its createLexicon*Fast + print_query_GPU_Gappy (ExtractPair.c:515-1276,
                PrintResults.c:407-577) on one batch, next to the restatement's own timers.
The sample is a smaller corpus from the benchmark's generator (same model, vocabulary, seed)
with queries drawn from it by the benchmark's recipe.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VP = C.c_void_p


def load_oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(path)
    lib.orc_index_from_arrays.restype = VP
    lib.orc_index_from_arrays.argtypes = [VP, C.c_uint32, VP, C.c_int32, VP, C.c_uint32, VP, VP, VP, VP, VP, VP, VP, C.c_uint32, VP]
    lib.orc_batch_from_ids.restype = VP; lib.orc_batch_from_ids.argtypes = [VP, C.c_int32, VP, C.c_int32]
    lib.orc_run_all.argtypes = [VP, VP, C.c_char_p]; lib.orc_batch_free.argtypes = [VP]; lib.orc_index_free.argtypes = [VP]
    lib.orc_dump.argtypes = [VP, VP, C.c_char_p]
    lib.orc_batch_times.argtypes = [VP, VP]; lib.orc_batch_times.restype = C.c_uint64
    return lib


def ptr(a):
    return a.ctypes.data_as(VP)


def run_queries(lib, ix, qoff, qtok, a, b, scratch):
    """orc_run_all over queries [a, b); returns seconds, lines written and the oracle's stage timers."""
    t0 = int(qoff[a]); t1 = int(qoff[b]) if b < len(qoff) else len(qtok)
    off = np.ascontiguousarray(qoff[a:b] - t0, np.int32); tok = np.ascontiguousarray(qtok[t0:t1], np.int32)
    out = tempfile.mkdtemp(prefix="cgx_cpu_", dir=scratch)
    bt = lib.orc_batch_from_ids(ptr(off), len(off), ptr(tok), len(tok))
    w0 = time.perf_counter()
    lib.orc_run_all(ix, bt, out.encode())
    dt = time.perf_counter() - w0
    tm = np.zeros(6, np.float64)
    lines = int(lib.orc_batch_times(bt, ptr(tm)))
    lib.orc_batch_free(bt); shutil.rmtree(out, ignore_errors=True)
    return dt, lines, dict(zip(("lookup", "gappy_search", "extraction", "lexicon", "maxlex", "write"), (round(float(x), 4) for x in tm)))


_G = {}


def _forked(job):
    a, b = job
    return run_queries(_G["lib"], _G["ix"], _G["qoff"], _G["qtok"], a, b, _G["scratch"])


def full_corpus_leg(args, lib, cores, usable, ncpu_online, scratch):
    """--full-dir: the benchmark's OWN corpus (the arrays bench.py generated, the suffix array and the frequent-pair tables the product
    built on the GPU, saved by bench.py after its timed region) and the first queries of its first batch: the per-query path on
    one core and on all cores against the whole corpus -- occurrence lists grow with the corpus, so the rate on a 2 % sample is
    not the rate on the benchmark's inputs."""
    d = args.full_dir
    ld = lambda k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r")
    keys = ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv", "sa", "freq", "pidx", "miss", "phit_start", "phit_len", "qoff", "qtok")
    t0 = time.perf_counter()
    A = {k: np.ascontiguousarray(ld(k)) for k in keys}
    lib.orc_index_from_arrays_pre.restype = VP
    lib.orc_index_from_arrays_pre.argtypes = [VP, C.c_uint32, VP, C.c_int32, VP, C.c_uint32, VP, VP, VP, VP, VP, VP, VP, C.c_uint32, VP, VP, VP, VP, VP, VP, C.c_uint32]
    ix = lib.orc_index_from_arrays_pre(ptr(A["str"]), len(A["str"]), ptr(A["sentind"]), len(A["sentind"]) - 1, ptr(A["tstr"]), len(A["tstr"]), ptr(A["tsentind"]), ptr(A["lsrc"]), ptr(A["rsrc"]),
                                       ptr(A["ltar"]), ptr(A["rtar"]), ptr(A["lexk"]), ptr(A["lexv"]), len(A["lexk"]), ptr(A["sa"]),
                                       ptr(A["freq"]), ptr(A["pidx"]), ptr(A["miss"]), ptr(A["phit_start"]), ptr(A["phit_len"]), len(A["phit_start"]))
    if not ix:
        return {"error": "orc_index_from_arrays_pre refused the tables"}
    n_src = int(len(A["str"])); qoff = np.asarray(A["qoff"], np.int64); qtok = A["qtok"]
    for k in keys[:-2]:
        A.pop(k)                                               # the oracle holds its own copies
    t_load = time.perf_counter() - t0
    nq_have = len(qoff); want = min(args.full_queries, nq_have)
    # one core: batches of 8 queries until `want` are done or the time cap is reached (at least 8)
    done = 0; dt1 = 0.0; lines1 = 0; st1 = {}
    while done < want and (done < 8 or dt1 < args.full_seconds):
        b = min(done + 8, want)
        dt, ln, st = run_queries(lib, ix, qoff, qtok, done, b, scratch)
        dt1 += dt; lines1 += ln; done = b
        for k, v in st.items():
            st1[k] = round(st1.get(k, 0.0) + v, 4)
    res = {"source_tokens": n_src, "load_s": round(t_load, 2),
           "one_core": {"value": round(done / dt1, 4), "rules_per_s": round(lines1 / dt1, 1), "cores": 1, "queries": done, "seconds": round(dt1, 3), "stages_s": st1, "batch_queries": 8}}
    # all cores: every forked worker runs its own 8-query batches of the following queries, for about the same time
    per_q = dt1 / max(done, 1)
    each = int(max(8, min((nq_have - want) // max(cores, 1), 8 * round(min(args.full_seconds, dt1) / per_q / 8 + 0.5)))) if nq_have - want >= 8 * cores else 0
    if each:
        _G.update(lib=lib, ix=ix, qoff=qoff, qtok=qtok, scratch=scratch)
        jobs = [(want + k * each, want + (k + 1) * each) for k in range(cores)]
        w0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            outs = pool.map(_forked8, jobs, chunksize=1)
        wall = time.perf_counter() - w0
        slow = max(o[0] for o in outs)
        res["all_cores"] = {"value": round(cores * each / slow, 3), "rules_per_s": round(sum(o[1] for o in outs) / slow, 1), "cores": cores, "queries": cores * each, "seconds": round(slow, 3),
                            "wall_seconds_with_fork": round(wall, 3), "batch_queries": 8}
        res["value"] = res["all_cores"]["value"]; res["cores"] = cores
    res["sample"] = ("the benchmark's own corpus (N=%d source tokens) with the suffix array and frequent-pair tables the product built on the GPU; queries %d.. of the first timed batch in "
                     "batches of 8: %d on one core%s; C restatement of the whole per-query path incl. file writing" % (n_src, 0, done, (", %d x %d on %d forked workers" % (cores, each, cores)) if each else ""))
    lib.orc_index_free(ix)
    return res


def _forked8(job):
    a, b = job; tot = 0.0; lines = 0
    for x in range(a, b, 8):
        dt, ln, _ = run_queries(_G["lib"], _G["ix"], _G["qoff"], _G["qtok"], x, min(x + 8, b), _G["scratch"])
        tot += dt; lines += ln
    return tot, lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-dir", default=None, help="directory with the benchmark's own corpus arrays, GPU-built suffix array and frequent-pair tables and queries (.npy, written by bench.py): run ONLY the full-corpus leg")
    ap.add_argument("--full-queries", type=int, default=64, help="queries of the one-core leg on the full corpus")
    ap.add_argument("--full-seconds", type=float, default=60.0, help="time cap of each full-corpus leg (at least 8 queries run whatever it says)")
    ap.add_argument("--pairs", type=int, default=200000, help="sentence pairs of the sample corpus")
    ap.add_argument("--vocab", type=int, default=200000)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--corpus-pairs", type=int, default=0, help="size of the benchmark corpus (only reported)")
    ap.add_argument("--budget-s", type=float, default=12.0, help="seconds of timed work per leg (one core, all cores)")
    ap.add_argument("--cores", type=int, default=0, help="workers of the all-core leg (0: the CPUs this process may use)")
    ap.add_argument("--max-queries", type=int, default=4000)
    ap.add_argument("--full-sa-seconds", type=float, default=0.0, help="> 0: also time the REFERENCE's suffixArrayConstruct (oracle/_ref) on the source tokens of the whole --corpus-pairs corpus, in a child process beside the other legs, for at most this many seconds")
    args = ap.parse_args()
    from cgx_amd import synth
    ncpu_online = os.sysconf("SC_NPROCESSORS_ONLN")
    usable = len(os.sched_getaffinity(0))
    try:                                                        # cgroup v2 CPU quota of the container
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            usable = min(usable, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    cores = args.cores if args.cores > 0 else usable
    lib = load_oracle()
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    if args.full_dir:
        print(json.dumps(full_corpus_leg(args, lib, cores, usable, ncpu_online, scratch))); sys.stdout.flush()
        return

    t0 = time.perf_counter()
    corpus = synth.make_corpus(args.pairs, args.vocab, args.seed)        # same model, vocabulary and seed as the benchmark corpus, fewer sentence pairs
    t_gen = time.perf_counter() - t0
    arrs = [np.ascontiguousarray(corpus[k]) for k in ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv")]
    t0 = time.perf_counter()
    ix = lib.orc_index_from_arrays(ptr(arrs[0]), len(arrs[0]), ptr(arrs[1]), len(arrs[1]) - 1, ptr(arrs[2]), len(arrs[2]), ptr(arrs[3]), ptr(arrs[4]), ptr(arrs[5]),
                                   ptr(arrs[6]), ptr(arrs[7]), ptr(arrs[8]), ptr(arrs[9]), len(arrs[8]), None)
    t_index = time.perf_counter() - t0
    qoff, qtok = synth.make_queries(corpus, args.max_queries * max(cores, 1) + 64, args.seed + 99)
    qoff = np.asarray(qoff, np.int64)

    # calibration: a few queries on one core
    dt, _, _ = run_queries(lib, ix, qoff, qtok, 0, 16, scratch)
    per_q = dt / 16
    n1 = int(max(16, min(args.max_queries, args.budget_s / per_q)))
    res = {"unit": "query sentences/s", "kind": "port", "host_cpus_online": int(ncpu_online), "cpus_usable": int(usable)}

    # ---- one core ----
    dt1, lines1, st1 = run_queries(lib, ix, qoff, qtok, 0, n1, scratch)
    res["one_core"] = {"value": round(n1 / dt1, 3), "rules_per_s": round(lines1 / dt1, 1), "cores": 1, "queries": n1, "seconds": round(dt1, 3), "stages_s": st1}

    # ---- reference objects on the same sample (index build, lexicon + printing of one batch) ----
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    refd = {}
    if os.path.exists(ref):
        d = tempfile.mkdtemp(prefix="cgx_ref_", dir=scratch)
        try:
            nb = min(n1, 64)
            off = np.ascontiguousarray(qoff[:nb] - qoff[0], np.int32); tok = np.ascontiguousarray(qtok[:int(qoff[nb])], np.int32)
            bt = lib.orc_batch_from_ids(ptr(off), nb, ptr(tok), len(tok))
            od = os.path.join(d, "o"); os.mkdir(od)
            lib.orc_run_all(ix, bt, od.encode())
            tm = np.zeros(6, np.float64); lib.orc_batch_times(bt, ptr(tm))
            dump = os.path.join(d, "dump.bin")
            if lib.orc_dump(ix, bt, dump.encode()) == 0:
                r = subprocess.run([ref, "time-sa", dump], capture_output=True, text=True, timeout=300)
                if r.returncode == 0:
                    refd["suffixArrayConstruct_s"] = float(r.stdout.strip().split()[-1])
                rd = os.path.join(d, "r"); os.mkdir(rd)
                w0 = time.perf_counter()
                r = subprocess.run([ref, "time-grammar", dump, rd], capture_output=True, text=True, timeout=300)
                if r.returncode == 0 and r.stdout.strip():
                    refd["createLexicon_print_s"] = float(r.stdout.strip().split()[-1]); refd["batch_queries"] = nb
                    refd["port_lexicon_write_s_same_batch"] = round(float(tm[3] + tm[5]), 4)
            lib.orc_batch_free(bt)
        except Exception as e:                                   # the reference leg is a bonus: never fatal
            refd["error"] = repr(e)
        shutil.rmtree(d, ignore_errors=True)
    res["reference_objects"] = refd or None

    # ---- all cores: forked workers share the index copy-on-write ----
    _G.update(lib=lib, ix=ix, qoff=qoff, qtok=qtok, scratch=scratch)
    jobs = [(16 + k * n1, 16 + (k + 1) * n1) for k in range(cores)]
    w0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        outs = pool.map(_forked, jobs, chunksize=1)
    wall = time.perf_counter() - w0
    slow = max(o[0] for o in outs)
    res["all_cores"] = {"value": round(cores * n1 / slow, 3), "rules_per_s": round(sum(o[1] for o in outs) / slow, 1), "cores": cores, "queries": cores * n1,
                        "seconds": round(slow, 3), "wall_seconds_with_fork": round(wall, 3)}
    # the reference's own suffix-array builder on the FULL benchmark corpus (one core; started only now, after the timed legs: run beside them it took one of the 16 CPUs and a good part of the memory bandwidth and cut the all-core figure by a third)
    full = None; ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if args.full_sa_seconds > 0 and args.corpus_pairs > 0 and os.path.exists(ref_bin):
        try:
            tf = time.perf_counter()
            fs = synth.make_source_tokens(args.corpus_pairs, args.vocab, args.seed)
            fd = tempfile.mkdtemp(prefix="cgx_fullsa_", dir=scratch); fdump = os.path.join(fd, "str.dump")
            with open(fdump, "wb") as f:                            # the two sections ref_harness time-sa reads: header[0] = n, str = n tokens + 3 zero pads
                hdr = np.zeros(15, np.uint32); hdr[0] = len(fs)
                f.write(b"header".ljust(8, b"\0") + np.uint64(hdr.nbytes).tobytes() + hdr.tobytes())
                f.write(b"str".ljust(8, b"\0") + np.uint64((len(fs) + 3) * 4).tobytes()); fs.tofile(f); f.write(np.zeros(3, np.int32).tobytes())
            full = {"proc": subprocess.Popen([ref_bin, "time-sa", fdump], stdout=subprocess.PIPE, text=True), "dir": fd, "t0": time.perf_counter(), "tokens": int(len(fs)), "prep_s": time.perf_counter() - tf}
            del fs
        except Exception as e:
            full = {"error": repr(e)}
    if full and "proc" in full:                                  # join the full-size reference run (it has had the time of every leg above)
        try:
            left = max(1.0, args.full_sa_seconds - (time.perf_counter() - full["t0"]))
            out, _ = full["proc"].communicate(timeout=left)
            if full["proc"].returncode == 0 and out.strip():
                refd["suffixArrayConstruct_full_s"] = float(out.strip().split()[-1]); refd["suffixArrayConstruct_full_tokens"] = full["tokens"]
                refd["suffixArrayConstruct_full_note"] = "SuffixArray.c:196-242 (DC3 + LCP tables) compiled in place, one core, on the source tokens of the whole benchmark corpus; run after the other CPU legs"
            else:
                refd["suffixArrayConstruct_full_s"] = None; refd["suffixArrayConstruct_full_note"] = "ref_harness exited with %s" % full["proc"].returncode
        except subprocess.TimeoutExpired:
            full["proc"].kill(); refd["suffixArrayConstruct_full_s"] = None
            refd["suffixArrayConstruct_full_note"] = "not finished within %.0f s on %d tokens" % (args.full_sa_seconds, full["tokens"])
        shutil.rmtree(full["dir"], ignore_errors=True)
        res["reference_objects"] = refd or None
    elif full and "error" in full:
        refd["suffixArrayConstruct_full_note"] = full["error"]; res["reference_objects"] = refd
    res.update(value=res["all_cores"]["value"], cores=cores,
               index={"port_sa_precompute_s": round(t_index, 3), "sample_gen_s": round(t_gen, 2)},
               sample="%d sentence pairs (N=%d source tokens; same generator, vocabulary and seed as the %s-pair benchmark corpus), queries by the benchmark's recipe: "
                      "%d on one core, %d x %d on %d forked workers; C restatement of the whole path incl. file writing, index build excluded"
                      % (args.pairs, len(arrs[0]), args.corpus_pairs or "?", n1, cores, n1, cores))
    print(json.dumps(res)); sys.stdout.flush()
    lib.orc_index_free(ix)


if __name__ == "__main__":
    main()
