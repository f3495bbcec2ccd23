"""The N>1 path on CPU: two gloo ranks shard a query batch with cgx_amd.shard, each computes its
shard (with the CPU oracle standing in as the compute, since there is no GPU here), and the
union of the per-rank grammar files must equal the single-process result; the bench's
max-over-ranks / sum-over-ranks reductions are exercised on the same process group."""
import os
import socket
import subprocess
import sys

import numpy as np

import oracle_py as op

ROOT = op.ROOT

WORKER = r'''
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from cgx_amd import shard, synth
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); out = sys.argv[2]
dist.init_process_group("gloo")
corpus = synth.make_corpus(1500, 300, 21)
qoff, qtok = synth.make_queries(corpus, 9, 22)
first, so, st = shard.take_shard(qoff, qtok, rank, world)
lib = C.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle.so")); vp = C.c_void_p
lib.orc_index_from_arrays.restype = vp
lib.orc_index_from_arrays.argtypes = [vp, C.c_uint32, vp, C.c_int32, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp]
lib.orc_batch_from_ids.restype = vp; lib.orc_batch_from_ids.argtypes = [vp, C.c_int32, vp, C.c_int32]
for f in ("orc_sa_lookup", "orc_gappy_search", "orc_extract", "orc_features"): getattr(lib, f).argtypes = [vp, vp]
lib.orc_write_grammars.argtypes = [vp, C.c_char_p, C.c_int]
p = lambda a: a.ctypes.data_as(vp)
arrs = [np.ascontiguousarray(corpus[k]) for k in ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv")]
ix = lib.orc_index_from_arrays(p(arrs[0]), len(arrs[0]), p(arrs[1]), len(arrs[1]) - 1, p(arrs[2]), len(arrs[2]), p(arrs[3]), p(arrs[4]), p(arrs[5]),
                               p(arrs[6]), p(arrs[7]), p(arrs[8]), p(arrs[9]), len(arrs[8]), None)
b = lib.orc_batch_from_ids(p(so), len(so), p(st), len(st))
for f in ("orc_sa_lookup", "orc_gappy_search", "orc_extract", "orc_features"): getattr(lib, f)(ix, b)
assert lib.orc_write_grammars(b, out.encode(), first) == 0
t = shard.max_over_ranks(1.0 + rank, dist); s = shard.sum_over_ranks(len(so), dist)
assert t == float(world) and s == len(qoff), (t, s)
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_bounds_cover_everything():
    from cgx_amd import shard
    qoff = np.array([0, 3, 3, 10, 40, 41], np.int32); ntok = 50
    for w in (1, 2, 3, 4, 8):
        b = shard.shard_bounds(qoff, ntok, w)
        assert b[0] == 0 and b[-1] == len(qoff) and np.all(np.diff(b) >= 0) and len(b) == w + 1
        got = []
        for r in range(w):
            first, so, st = shard.take_shard(qoff, np.arange(ntok, dtype=np.int32), r, w)
            got += st.tolist()
            assert first == b[r]
        assert got == list(range(ntok))


def test_two_gloo_ranks_reproduce_the_single_process_files(oracle_bin, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = tmp_path / "worker.py"; worker.write_text(WORKER)
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("w%d" % world); out.mkdir()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + world), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, str(worker), ROOT, str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        for p in procs:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0, o
        outs[world] = op.sha_dir(str(out), 9)
    assert outs[1] == outs[2]


def test_c_and_python_shard_policies_agree():
    """strmatchcuda --shard i/n (cgx_shard_bounds, host C) and bench.py (cgx_amd.shard) must cut a query list at the same places."""
    import ctypes as C
    import cgx_amd
    from cgx_amd import shard
    lib = cgx_amd.load_library()
    lib.cgx_shard_bounds.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
    rng = np.random.default_rng(5)
    for trial in range(200):
        nq = int(rng.integers(0, 40)); lens = rng.integers(0, 30, nq) * (rng.random(nq) < 0.8)
        qoff = np.concatenate(([0], np.cumsum(lens)))[:-1].astype(np.int32) if nq else np.zeros(0, np.int32)
        ntok = int(lens.sum())
        for world in (1, 2, 3, 8, 17):
            got = np.zeros(world + 1, np.int32)
            assert lib.cgx_shard_bounds(qoff.ctypes.data_as(C.c_void_p) if nq else None, nq, ntok, world, got.ctypes.data_as(C.c_void_p)) == 0
            want = shard.shard_bounds(qoff, ntok, world) if nq else np.zeros(world + 1, np.int64)
            assert np.array_equal(got, want), (trial, world, qoff, got, want)
