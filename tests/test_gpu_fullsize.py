"""Full-size GPU checks at the corpus sizes of BASELINE configs[2] (10 M sentence pairs, 2.6e8 source tokens), configs[3]
(Europarl scale, 5.5e7 source tokens) and configs[4] (1e8 source tokens).  The oracle cannot run at these sizes, so the HIP
path is checked through properties that do not depend on size: the suffix array is a sorted permutation, every reported
interval is exactly the set of suffixes that start with the phrase, the grammar files do not depend on how the batch is
split or scheduled, and (cfg4) two contexts that each take half of the queries against their own replica of the index
write the files one context writes."""
import hashlib
import os
import shutil
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAIRS = int(os.environ.get("CGX_FULLSIZE_PAIRS", "10000000"))
SIZES = {"cfg3": (PAIRS, 200000), "cfg5": (int(os.environ.get("CGX_CFG5_PAIRS", "3846000")), 200000)}


@pytest.fixture(scope="module", params=sorted(SIZES))
def world(request):
    pairs, vocab = SIZES[request.param]
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd
    from cgx_amd import synth
    cgx_amd.load_library()
    corpus = synth.make_corpus(pairs, vocab, 1)
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                                   corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    ex = cgx_amd.Extractor(0)
    ex.upload_corpus(host)
    yield cgx_amd, synth, corpus, host, ex
    ex.close(); host.close()


def _window(s, pos, w):
    """tokens s[pos + k], k < w, zero beyond the end"""
    idx = pos[:, None].astype(np.int64) + np.arange(w)[None, :]
    ok = idx < len(s)
    return np.where(ok, s[np.minimum(idx, len(s) - 1)], 0)


def test_suffix_array_is_a_sorted_permutation(world):
    cgx, synth, corpus, host, ex = world
    s = corpus["str"]; n = len(s)
    sa = ex.fetch("sa")
    assert len(sa) == n
    assert np.array_equal(np.bincount(sa, minlength=n), np.ones(n, np.int64))
    rng = np.random.default_rng(7)
    i = rng.integers(0, n - 1, 300000)
    W = 24
    a = _window(s, sa[i], W); b = _window(s, sa[i + 1], W)
    diff = a != b
    first = np.where(diff.any(1), diff.argmax(1), W)
    decided = first < W
    assert decided.mean() > 0.99                              # 24 tokens separate almost every adjacent pair
    k = first[decided]
    assert np.all(a[decided, k] < b[decided, k])


def test_intervals_are_exactly_the_matching_suffixes(world):
    cgx, synth, corpus, host, ex = world
    s = corpus["str"]; n = len(s)
    sa = ex.fetch("sa")
    qoff, qtok = synth.make_queries(corpus, 2000, 11)
    ex.upload_queries(qoff, qtok); ex.sa_lookup()
    lm, up, down = ex.fetch("lm"), ex.fetch("up").reshape(-1, 5), ex.fetch("down").reshape(-1, 5)
    T = len(qtok)
    qend = np.concatenate((qoff[1:], [T]))
    tok2end = np.repeat(qend, qend - qoff)
    rng = np.random.default_rng(3)
    t = rng.integers(0, T, 60000); t = t[lm[t] > 0]
    l = 1 + rng.integers(0, 5, len(t)) % lm[t]
    u = up[t, l - 1].astype(np.int64); d = down[t, l - 1].astype(np.int64)
    assert np.all(u <= d)
    want = _window(qtok, t, 5)
    mask = np.arange(5)[None, :] < l[:, None]
    def matches(pos, sel=slice(None)):
        return np.all((_window(s, pos, 5) == want[sel]) | ~mask[sel], axis=1)
    mid = u + ((d - u) * rng.random(len(t))).astype(np.int64)
    assert matches(sa[u]).all() and matches(sa[d]).all() and matches(sa[mid]).all()
    lo = u > 0; hi = d < n - 1                                # the neighbours just outside the interval do not match
    assert not matches(sa[u[lo] - 1], lo).any()
    assert not matches(sa[d[hi] + 1], hi).any()
    # lm is maximal: the next longer phrase (if the query has one and l < 5) does not occur
    room = (t + lm[t] < tok2end[t]) & (lm[t] < 5)
    tt = t[room]; ll = lm[tt]
    uu = up[tt, ll - 1].astype(np.int64); dd = down[tt, ll - 1].astype(np.int64)
    small = (dd - uu) < 64                                    # check exhaustively where the interval is short
    tt, ll, uu, dd = tt[small], ll[small], uu[small], dd[small]
    nxt = qtok[tt + ll]
    for k in range(64):
        pos = np.minimum(uu + k, dd)
        assert not np.any(s[np.minimum(sa[pos] + ll, n - 1)] == nxt)


def _sha_dir(d, nq):
    h = hashlib.sha256()
    for q in range(nq):
        with open(os.path.join(d, "grammar.%d.s" % q), "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b:
                    break
                h.update(b)
        h.update(b"\0")
    return h.hexdigest()


def test_grammar_files_do_not_depend_on_batching(world):
    cgx, synth, corpus, host, ex = world
    nq = 600
    qoff, qtok = synth.make_queries(corpus, nq, 5)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    out = tempfile.mkdtemp(prefix="cgx_full_", dir=base)
    try:
        dirs = [os.path.join(out, x) for x in "abc"]
        for d in dirs:
            os.mkdir(d)
        n_all = ex.extract_grammars_ids(host, qoff, qtok, dirs[0], 0)
        lines = 0
        for q in range(nq):
            with open(os.path.join(dirs[0], "grammar.%d.s" % q), "rb") as f:
                lines += sum(b.count(b"\n") for b in iter(lambda: f.read(1 << 24), b""))
        assert lines == n_all and n_all > 0
        ref = _sha_dir(dirs[0], nq)
        # two halves, asynchronous writer
        ex.set_option("async_write", 1)
        cut = nq // 3; tcut = int(qoff[cut])
        n1 = ex.extract_grammars_ids(host, qoff[:cut], qtok[:tcut], dirs[1], 0)
        n2 = ex.extract_grammars_ids(host, qoff[cut:] - tcut, qtok[tcut:], dirs[1], cut)
        ex.flush(); ex.set_option("async_write", 0)
        assert n1 + n2 == n_all and _sha_dir(dirs[1], nq) == ref
        # internal sub-batches and small lookup launches
        ex.set_option("sub_batch", 250); ex.set_option("chunk_items", 1 << 20)
        n3 = ex.extract_grammars_ids(host, qoff, qtok, dirs[2], 0)
        ex.set_option("sub_batch", 0); ex.set_option("chunk_items", 1 << 26)
        assert n3 == n_all and _sha_dir(dirs[2], nq) == ref
        # the derived layouts of the lookups switched off (occurrences in suffix order, the sentence's target offset through
        # the delimiter instead of the source-addressed blocks): at this size the table exists, and both ways agree
        assert 1 <= ex.stage_ms("src_blocks_factor") <= 4
        shutil.rmtree(dirs[2]); os.mkdir(dirs[2])
        ex.set_option("src_blocks", 0); ex.set_option("occ_order", 0)
        n4 = ex.extract_grammars_ids(host, qoff, qtok, dirs[2], 0)
        ex.set_option("src_blocks", 1); ex.set_option("occ_order", 1)
        assert n4 == n_all and _sha_dir(dirs[2], nq) == ref
        # the hit lists sorted completely on the card instead of by pattern and position bucket with the order statistics selected
        # (lists of hundreds of thousands of occurrences here: every path of k_sort_lists / k_select_hits / k_select_rank against the full sort);
        # and MaxLex by the one-lane-per-line kernel instead of the wave-wide task list (same floats, bit for bit)
        shutil.rmtree(dirs[2]); os.mkdir(dirs[2])
        ex.set_option("hit_order", 1); ex.set_option("lex_flat", 0)
        n5 = ex.extract_grammars_ids(host, qoff, qtok, dirs[2], 0)
        ex.set_option("hit_order", 0); ex.set_option("lex_flat", 1)
        assert n5 == n_all and _sha_dir(dirs[2], nq) == ref
        # the same files as DEFLATE data made by the formatter (text offsets of gigabytes, groups of thousands of lines, back-references of
        # kilobytes): every grammar.<q>.s.gz is ONE gzip member whose CRC-32 and ISIZE hold (gzip checks both) and whose content is the plain file
        import gzip
        for dyn in (1, 0):
            shutil.rmtree(dirs[2]); os.mkdir(dirs[2])
            ex.set_option("gz_level", 1); ex.set_option("gz_dynamic", dyn)
            n6 = ex.extract_grammars_ids(host, qoff, qtok, dirs[2], 0)
            ex.set_option("gz_level", 0); ex.set_option("gz_dynamic", 1)
            assert n6 == n_all and ex.stage_ms("fmt_gz") == 1.0 and ex.stage_ms("fmt_gz_dynamic") == float(dyn)
            h = hashlib.sha256()                                  # as _sha_dir: one digest over the files' contents in order
            for q in range(nq):
                with gzip.open(os.path.join(dirs[2], "grammar.%d.s.gz" % q), "rb") as f:
                    for b in iter(lambda: f.read(1 << 24), b""):
                        h.update(b)
                h.update(b"\0")
            assert h.hexdigest() == ref, "gz_dynamic=%d" % dyn
            assert ex.stage_ms("fmt_unique_bytes") < (0.30 if dyn else 0.40) * ex.stage_ms("fmt_plain_unique_bytes")
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _sample_files(d, period, count, nq):
    h = hashlib.sha256(); n = 0
    for q in range(nq):
        if q % period < count:
            with open(os.path.join(d, "grammar.%d.s" % q), "rb") as f:
                h.update(f.read()); h.update(b"\0"); n += 1
    return h.hexdigest(), n


def test_cfg5_one_million_queries(world, request):
    """BASELINE configs[4] at its stated query count: 10^6 query sentences (2.5e7 query tokens) against the 1e8-token corpus,
    once as ONE call (84 internal batches chosen by the library) and once cut into sub-batches of 6 000 sentences with small
    lookup launches and the asynchronous writer.  The rule counts must agree, and so must the grammar files of 200 sampled
    queries (four windows of 50; option write_period / write_count: the other 999 800 files are counted, not written) -- which
    in turn equal the files the same 50 sentences produce as a small batch of their own."""
    cgx, synth, corpus, host, ex = world
    if request.node.callspec.params["world"] != "cfg5":
        pytest.skip("the stress configuration only")
    nq = int(os.environ.get("CGX_CFG5_QUERIES", "1000000")); period, count = nq // 4, 50
    qoff, qtok = synth.make_queries(corpus, nq, 21)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    out = tempfile.mkdtemp(prefix="cgx_cfg5_", dir=base)
    try:
        a, b, c = (os.path.join(out, x) for x in "abc")
        for d in (a, b, c):
            os.mkdir(d)
        ex.set_option("write_period", period); ex.set_option("write_count", count); ex.set_option("async_write", 1)
        n_a = ex.extract_grammars_ids(host, qoff, qtok, a, 0); ex.flush()
        ex.set_option("sub_batch", 6000); ex.set_option("chunk_items", 1 << 24)
        n_b = ex.extract_grammars_ids(host, qoff, qtok, b, 0); ex.flush()
        ex.set_option("sub_batch", 0); ex.set_option("chunk_items", 1 << 26); ex.set_option("write_period", 0); ex.set_option("async_write", 0)
        assert n_a == n_b and n_a > 50 * nq                    # thousands of rules per sentence
        sa_, na = _sample_files(a, period, count, nq); sb_, nb = _sample_files(b, period, count, nq)
        assert na == nb == 4 * count and sa_ == sb_
        assert len(os.listdir(a)) == 4 * count                  # nothing else was written
        # the first window as a batch of its own
        t1 = int(qoff[count])
        ex.extract_grammars_ids(host, qoff[:count], qtok[:t1], c, 0)
        for q in range(count):
            assert open(os.path.join(a, "grammar.%d.s" % q), "rb").read() == open(os.path.join(c, "grammar.%d.s" % q), "rb").read(), q
    finally:
        shutil.rmtree(out, ignore_errors=True)


def test_cfg4_two_contexts_split_equals_one(tmp_path):
    """BASELINE configs[3] on one card: Europarl-scale corpus (about 2.1 M sentence pairs, N = 5.5e7), one batch of queries
    (a) on the context that built the index, (b) split by token count (cgx_amd.shard, the policy of bench.py and of
    strmatchcuda --shard) over that context and a second one holding a replica received buffer by buffer: the union of the
    two shards' files equals (a), and the rule counts add up."""
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd
    from cgx_amd import synth, shard
    pairs = int(os.environ.get("CGX_CFG4_PAIRS", "2115000"))
    corpus = synth.make_corpus(pairs, 150000, 4)
    assert 5.0e7 < len(corpus["str"]) < 6.0e7 or pairs != 2115000
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                                   corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    root = cgx_amd.Extractor(0); root.upload_corpus(host)
    rep = cgx_amd.Extractor(0); rep.index_alloc(root.index_shape())
    for i, (name, nbytes) in enumerate(root.index_buffers()):
        if nbytes:
            stage = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
            root.index_d2d(i, stage.data_ptr(), 0); torch.cuda.synchronize()
            rep.index_d2d(i, stage.data_ptr(), 1); torch.cuda.synchronize()
            del stage
    rep.index_finalize()
    nq = 1500
    qoff, qtok = synth.make_queries(corpus, nq, 8)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    out = tempfile.mkdtemp(prefix="cgx_cfg4_", dir=base)
    try:
        a, b = os.path.join(out, "a"), os.path.join(out, "b"); os.mkdir(a); os.mkdir(b)
        n_all = root.extract_grammars_ids(host, qoff, qtok, a, 0)
        total = 0
        for r, ex in enumerate((root, rep)):
            first, so, st = shard.take_shard(qoff, qtok, r, 2)
            assert len(so) > 0
            ex.set_option("async_write", 1)
            total += ex.extract_grammars_ids(host, so, st, b, first)
        root.flush(); rep.flush()
        assert total == n_all and n_all > 0 and _sha_dir(a, nq) == _sha_dir(b, nq)
        # ---- the configuration's own batch: 50 000 query sentences (1.3e6 tokens, five internal batches), whole on one
        # context and split by token count over the two; rule counts must agree, and the files of 200 sampled queries
        # (four windows of 50) written by the whole run and by the two shards must be the same bytes
        NQ = int(os.environ.get("CGX_CFG4_QUERIES", "50000")); period, count = NQ // 4, 50
        qoff, qtok = synth.make_queries(corpus, NQ, 9)
        c, d = os.path.join(out, "c"), os.path.join(out, "d"); os.mkdir(c); os.mkdir(d)
        for ex in (root, rep):
            ex.set_option("write_period", period); ex.set_option("write_count", count); ex.set_option("async_write", 1)
        n_whole = root.extract_grammars_ids(host, qoff, qtok, c, 0); root.flush()
        n_split = 0
        for r, ex in enumerate((root, rep)):
            first, so, st = shard.take_shard(qoff, qtok, r, 2)
            n_split += ex.extract_grammars_ids(host, so, st, d, first)
        root.flush(); rep.flush()
        assert n_whole == n_split and n_whole > 50 * NQ
        assert _sample_files(c, period, count, NQ) == _sample_files(d, period, count, NQ) and len(os.listdir(c)) == 4 * count == len(os.listdir(d))
    finally:
        shutil.rmtree(out, ignore_errors=True)
        rep.close(); root.close(); host.close()


def test_bench_contract_small(tmp_path):
    """bench.py end to end on a small corpus with the cfg5 mechanics (strong scaling, several spool chunks per step that
    reuse the file slots): one JSON line with the contract's fields, an honest roofline block and a CPU baseline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "cfg5", "--pairs", "60000", "--queries", "2500", "--chunk-queries", "1000", "--steps", "2",
                        "--warmup", "1", "--cpu-pairs", "20000", "--cpu-seconds", "1.5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["config"]["grammar_files_written"] and "3 chunk(s)" in d["config"]["outdir_mode"]
    cb = d["cpu_baseline"]
    assert cb.get("kind") == "port" and cb["one_core"]["value"] > 0 and cb["all_cores"]["cores"] >= 1 and cb["value"] > 0
    # beside the in-place figure: the same steps into fresh directories, the GPU chain alone, and the per-rank host stages
    assert d["fresh_files"]["steps"] == 1 and d["value_fresh_files"] > 0 and d["value_gpu_chain"] >= d["value"] * 0.9
    pr = d["per_rank"]
    assert pr["gpu_chain_ms_per_step"][0] > 0 and pr["writer_threads"][0] >= 1 and pr["cpus_usable_per_process"][0] >= 1
    assert rf["kernel_ms"] == rf["kernel_ms_mean_timed_steps"]            # the fraction is priced on the launches of the timed steps


def test_bench_toy_line():
    """bench.py --config toy: the toy line north_star asks for (20 k sentence pairs, 7 queries per step)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "toy", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["name"] == "toy" and d["config"]["sentence_pairs"] == 20000 and d["config"]["queries_per_step_this_rank"] == 7 and d["value"] > 0 and d["rules_per_s"] > 0


@pytest.mark.parametrize("bcast", ["torch", "c"])
def test_two_gpu_bench_broadcast_paths(bcast):
    """N = 2 on real hardware (skipped on a one-GPU box): bench.py under torchrun, index broadcast through torch.distributed
    and through the library's own cgx_broadcast_index on an RCCL communicator; both must report the same rule count per
    step as a single rank processing the same global batch."""
    import json
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "cfg4", "--pairs", "60000", "--queries", "2000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(root, "bench.py"), "--gpus", "2", "--bcast", bcast] + common, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["index"]["broadcast_bytes"] > 0
    a, b = two["rules_per_s"] * two["ms_per_step"], one["rules_per_s"] * one["ms_per_step"]                     # rules per step x 1000, from rounded fields
    assert abs(a - b) <= 1e-4 * b and two["counts"]["grammar_lines"] > 0


def test_two_ranks_on_one_gpu_run_the_product(tmp_path):
    """The N > 1 flow of the PRODUCT on the one card a test box has: bench.py under torchrun with two gloo ranks that both use
    cuda:0 (`--single-device`): corpus files shared through /dev/shm, rank 1 receives the index buffer by buffer and rebuilds
    the derived tables, the queries are split by token count, each rank writes its own grammar files, the per-rank statistics
    are gathered.  The rule count of a step must equal a single rank's over the same global batch.  (RCCL itself needs two
    cards: test_two_gpu_bench_broadcast_paths.)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "cfg4", "--pairs", "60000", "--queries", "2000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--fresh-steps", "1"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device"] + common, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["index"]["broadcast_bytes"] > 0 and two["value_fresh_files"] > 0
    pr = two["per_rank"]
    assert pr["writer_threads"][0] >= 2 and pr["gpu_chain_ms_per_step"][0] > 0 and pr["gpu_chain_ms_per_step"][1] >= pr["gpu_chain_ms_per_step"][0]
    a, b = two["rules_per_s"] * two["ms_per_step"], one["rules_per_s"] * one["ms_per_step"]                     # rules per step x 1000, from rounded fields
    assert abs(a - b) <= 1e-4 * b and two["counts"]["grammar_lines"] > 0


def _rccl():
    import ctypes as C
    try:
        lib = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
    except OSError:
        lib = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p; lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib, UniqueId


def test_rccl_broadcast_on_a_one_rank_communicator(tmp_path):
    """cgx_broadcast_index on REAL RCCL with the one card a test box has: `ncclCommInitRank` with nranks = 1 gives a communicator
    on which the library's own call runs end to end -- dlopen of librccl, the symbol look-ups, the datatype constant, the grouped
    ncclBroadcast of every index buffer on the context's stream -- first as the root (the buffers come back as they were: the same
    golden files afterwards), then on a second context that holds a replica and is told it is NOT the root, so that the call ends
    in cgx_index_finalize as on ranks 1..N-1 of a real job.  (Two ranks need two cards: test_two_gpu_bench_broadcast_paths.)"""
    import ctypes as C
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd as cgx
    import oracle_py as op
    from test_oracle import META, make_fixture
    fx = make_fixture("toy", str(tmp_path / "fx")); files = op.fixture_args(fx)
    rccl, UniqueId = _rccl()
    uid = UniqueId(); rc = rccl.ncclGetUniqueId(C.byref(uid)); assert rc == 0, rccl.ncclGetErrorString(rc)
    comm = C.c_void_p(); rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0); assert rc == 0, rccl.ncclGetErrorString(rc)
    root = cgx.Extractor(0); corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4]); root.upload_corpus(corpus)
    root._chk(root.lib.cgx_broadcast_index(root.h, comm, 0, 0), "cgx_broadcast_index (root)")
    nbytes = sum(nb for _, nb in root.index_buffers())
    assert nbytes > 0 and root.stage_ms("broadcast_bytes") == nbytes
    os.makedirs(str(tmp_path / "a")); root.extract_grammars(corpus, files[1], str(tmp_path / "a"))
    assert op.sha_dir(str(tmp_path / "a"), 7) == META["toy"]["grammar"]
    # a replica, filled buffer by buffer, then the same call as a non-root rank: broadcast (from itself: one rank) + finalize
    rep = cgx.Extractor(0); rep.index_alloc(root.index_shape())
    for i, (name, nb) in enumerate(root.index_buffers()):
        if nb == 0:
            continue
        stage = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        root.index_d2d(i, stage.data_ptr(), 0); torch.cuda.synchronize()
        rep.index_d2d(i, stage.data_ptr(), 1); torch.cuda.synchronize()
    rep._chk(rep.lib.cgx_broadcast_index(rep.h, comm, 0, 1), "cgx_broadcast_index (replica)")
    os.makedirs(str(tmp_path / "b")); rep.extract_grammars(corpus, files[1], str(tmp_path / "b"))
    assert op.sha_dir(str(tmp_path / "b"), 7) == META["toy"]["grammar"]
    # a context without buffers is refused, not crashed
    empty = cgx.Extractor(0)
    assert empty.lib.cgx_broadcast_index(empty.h, comm, 0, 1) != 0 and b"not allocated" in empty.lib.cgx_last_error(empty.h)
    empty.close(); rep.close(); root.close(); corpus.close()
    assert rccl.ncclCommDestroy(comm) == 0


@pytest.mark.parametrize("bcast", ["torch", "c"])
def test_bench_under_torch_distributed_nccl_with_one_rank(bcast):
    """bench.py's N > 1 code path on one card, in a fresh process: `--force-dist` creates the torch.distributed process group with
    backend nccl (= RCCL) for a world of one and takes every branch a multi-GPU run takes -- index shape broadcast, the index
    buffers through torch.distributed.broadcast / through cgx_broadcast_index on a communicator of our own, barriers, gathers,
    max over ranks.  The rule count must equal the plain single-process run's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "cfg4", "--pairs", "60000", "--queries", "2000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--fresh-steps", "0", "--gz-steps", "0"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl", "--bcast", bcast] + common, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert two["n_gpus"] == 1 and two["index"]["broadcast_bytes"] > 0 and two["index"]["broadcast_s"] >= 0
    a, b = two["rules_per_s"] * two["ms_per_step"], one["rules_per_s"] * one["ms_per_step"]
    assert abs(a - b) <= 1e-4 * b and two["counts"]["grammar_lines"] > 0
