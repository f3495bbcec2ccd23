"""Third opinion for the stages whose oracle cannot be pinned against reference object code (one-/two-gap hit sets,
frequent-pair lists, ab / aXb / aXbXc rules): tests/bruteforce.py restates the DEFINITIONS of those results (SURVEY.md
Appendix B) in numpy, independently of both the kernels and oracle/.  CPU: the oracle against the definitions; GPU: the
HIP path against the definitions."""
import os

import numpy as np
import pytest

import bruteforce as bf
import oracle_py as op
from test_oracle import make_fixture


def _phits(pidx, starts, lens):
    out = {}
    pidx = np.asarray(pidx, np.int64).reshape(-1, 2)
    for pre, (a, b) in enumerate(pidx):
        if b >= a:
            out[pre] = [(int(starts[i]), int(lens[i])) for i in range(a, b + 1)]
    return out


def test_oracle_agrees_with_the_definitions(oracle_bin, fixtures_dir, tmp_path):
    fx = make_fixture("tiny", fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    c = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"])
    p1d = d["p1"][d["s1"]["position"]]; c2d = d["p2"]["pat"][:, 0][d["s2"]["position"]]
    done = bf.check_batch(c, d["sa"], d["freq"], _phits(d["pidx"], d["phits"]["start"], d["phits"]["length"]), d["s1"], p1d, d["hits1"],
                          d["s2"], c2d, d["hits2"], d["blocks"], d["r0"], d["r1"], h["sep1"], d["r2"], h["sep2a"], h["sep2b"])
    assert done > 100
    # the whole frequent-pair table, not only the pairs the queries use
    freq = [int(x) for x in d["freq"]]; ph = _phits(d["pidx"], d["phits"]["start"], d["phits"]["length"])
    rng = np.random.default_rng(3)
    for pre in rng.integers(0, 10000, 300):
        assert ph.get(int(pre), []) == bf.frequent_pair_list(c, freq[pre // 100], freq[pre % 100]), pre


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "toy", "mid"])
def test_hip_path_agrees_with_the_definitions(name, oracle_bin, fixtures_dir, tmp_path):
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd as cgx
    fx = make_fixture(name, fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)               # only for the id-level input arrays of the fixture
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    ex = cgx.Extractor(0)
    ex.upload_index(d["str"][:n], d["rlp"], d["tstr"][:nt], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa(); ex.precompute(); ex.upload_queries(d["qoff"][:-1], d["qtok"]); ex.sa_lookup(); ex.make_blocks(); ex.gappy_search(); ex.extract()
    c = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"]); k = ex.counts()
    # the larger fixture: the first 300 one-gap patterns and 500 drawn over the whole id range, with the two-gap patterns
    # built on them (python loops: a full sweep of toy takes minutes)
    # mid (5 000 pairs, 40 queries, vocabulary 300): every block and 3 000 one-gap patterns drawn over the id range
    limit = None if name == "tiny" else 300
    extra = 500 if name == "toy" else 3000
    pick = None if name == "tiny" else sorted(set(range(300)) | set(int(x) for x in np.random.default_rng(11).integers(0, max(int(k["d1"]), 1), extra)))
    done = bf.check_batch(c, ex.fetch("sa"), ex.fetch("freq"), _phits(ex.fetch("pidx"), ex.fetch("phit_start"), ex.fetch("phit_len")), ex.fetch("s1"), ex.fetch("p1d"),
                          ex.fetch("hits1"), ex.fetch("s2"), ex.fetch("c2d"), ex.fetch("hits2"), ex.fetch("blocks"), ex.fetch("r0"), ex.fetch("r1"), k["sep1"], ex.fetch("r2"), k["sep2a"], k["sep2b"], max_patterns=limit, pick=pick)
    assert done > 100 and k["guard_exits"] == 0
    ex.close()
