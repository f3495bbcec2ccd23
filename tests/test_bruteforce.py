"""Third opinion for the stages whose oracle cannot be pinned against reference object code (longest matches, query-side
enumeration and pattern ids, one-/two-gap hit sets, frequent-pair lists, every rule family -- ab, Xab, abX, XabX, aXb, XaXb,
aXbX, aXbXc -- and the MaxLex features): tests/bruteforce.py restates the DEFINITIONS of those results (SURVEY.md
Appendix B, the reference text) in numpy / python, independently of both the kernels and oracle/.  CPU: the oracle against the
definitions; GPU: the HIP path against the definitions."""
import os

import numpy as np
import pytest

import bruteforce as bf
import oracle_py as op
from test_oracle import make_fixture as _golden_fixture
import gen_fixture

# two more shapes for the definition checks, generated here from their seeds (no golden files: the checks compare against the
# definitions, not against stored output): a tiny vocabulary with long sentences (dense matches, long occurrence lists, many
# rules per phrase) and short sentences over a large vocabulary (few matches, many sentence boundaries inside the windows)
EXTRA = {"dense": dict(pairs=260, vocab=104, queries=5, seed=77, lo=6, hi=30), "sparse": dict(pairs=4000, vocab=500, queries=60, seed=78, lo=1, hi=6),
         # "thin": the dense shape with 45 % of its alignment links dropped -- phrase edges without a link, gaps none of whose tokens is aligned,
         # target words nobody points at: the cases the reference handles with its unsigned-char "no position" marks (255)
         "thin": dict(pairs=260, vocab=104, queries=5, seed=79, lo=6, hi=30)}
THIN = {"thin": 0.45}


def make_fixture(name, base):
    if name not in EXTRA:
        return _golden_fixture(name, base)
    fx = os.path.join(base, "bf_%s_%d_%d_%d_%d" % (name, EXTRA[name]["pairs"], EXTRA[name]["vocab"], EXTRA[name]["queries"], EXTRA[name]["seed"]))
    if not os.path.exists(os.path.join(fx, "lex.txt")):
        gen_fixture.write_fixture(fx, **EXTRA[name])
        if name in THIN:                                        # the same corpus with fewer links (every line keeps one)
            rng = np.random.default_rng(EXTRA[name]["seed"]); out = []
            for line in open(os.path.join(fx, "corpus.a")):
                links = line.split(); keep = [w for w in links if rng.random() >= THIN[name]]
                out.append(" ".join(keep if keep else links[:1]) + "\n")
            open(os.path.join(fx, "corpus.a"), "w").write("".join(out))
    return fx


def _canon(h):
    """fetched hit records in the reference's order (the device keeps a list ordered by position bucket only; test_gpu_parity.canon_hits)"""
    keys = [h[f] for f in reversed([f for f in ("position", "str_position", "length", "length2") if f in h.dtype.names])]
    return h[np.lexsort(keys)]


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


def _lists(flat, nq):
    out = []; p = 0
    for _ in range(nq):
        k = int(flat[p]); out.append([int(x) for x in flat[p + 1:p + 1 + k]]); p += 1 + k
    return out


def _csr(off, ids):
    return [[int(x) for x in ids[int(off[q]):int(off[q + 1])]] for q in range(len(off) - 1)]


@pytest.mark.parametrize("name", ["tiny", "toy", "mid", "dense", "sparse", "thin"])
def test_oracle_agrees_with_the_definitions_of_the_remaining_stages(name, oracle_bin, fixtures_dir, tmp_path):
    """Query-side enumeration + ids + per-query lists, the extension rules (Xab, abX, XabX, XaXb, aXbX) and MaxLex: every
    block, every pattern, every lexical task of the fixture."""
    fx = make_fixture(name, fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]; nq = h["nq"]
    c = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"])
    p1d = d["p1"][d["s1"]["position"]]; c2d = d["p2"]["pat"][:, 0][d["s2"]["position"]]
    a, b = bf.check_query_side(c, d["qoff"][:-1], d["qtok"], d["lm"], len(d["g1"]), d["s1"], p1d, _lists(d["qone"], nq), len(d["g2"]), d["s2"], c2d, _lists(d["qtwo"], nq))
    assert (a, b) == (h["d1"], h["d2"])
    hits_of = bf.hit_lists(d["s1"], d["hits1"], d["pidx"], d["phits"]["start"], d["phits"]["length"])
    assert bf.check_extension_rules(c, d["sa"], d["blocks"], d["s1"], hits_of, d["r1"], h["sep1"], d["r2"], h["sep2a"], h["sep2b"]) > 1000
    tab = bf.LexTable(d["lexk"], d["lexv"]); nl1, nl2 = len(d["lex1_int"]) // 4, len(d["lex2_int"]) // 4
    for i, t in enumerate(d["tasks"]):
        g1, g2 = (int(t["gap1"]), int(t["gap1_1"])), (int(t["gap2"]), int(t["gap2_1"]))
        fe, ef = bf.maxlex(tab, d["tstr"], [int(x) for x in t["src"][:int(t["nsrc"])]], int(t["tstart"]), int(t["end"]), [g1] if i < nl1 else [g1, g2] if i < nl1 + nl2 else [])
        assert fe.tobytes() == d["task_fe"][i].tobytes() and ef.tobytes() == d["task_ef"][i].tobytes(), ("MaxLex task", i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "toy", "mid", "dense", "sparse", "thin"])
def test_hip_path_agrees_with_the_definitions_of_the_remaining_stages(name, oracle_bin, fixtures_dir, tmp_path):
    """The HIP path's r1 / r2 ids b, G+b (Xab, abX), b (XabX), oneId, D1+oneId (XaXb, aXbX), its pattern ids and per-query
    lists, and lex*.fe / lex*.ef, all against tests/bruteforce.py -- no oracle result is compared here."""
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd as cgx
    fx = make_fixture(name, fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)               # only for the id-level input arrays of the fixture
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    ex = cgx.Extractor(0)
    ex.upload_index(d["str"][:n], d["rlp"], d["tstr"][:nt], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa(); ex.precompute(); ex.upload_queries(d["qoff"][:-1], d["qtok"]); ex.sa_lookup(); ex.make_blocks(); ex.gappy_search(); ex.extract(); ex.lexicon()
    ex.count_rules()                                                        # builds the per-query pattern lists
    c = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"]); k = ex.counts()
    s1, s2, p1d, c2d, one2, blocks = ex.fetch("s1"), ex.fetch("s2"), ex.fetch("p1d"), ex.fetch("c2d"), ex.fetch("one2"), ex.fetch("blocks")
    a, b = bf.check_query_side(c, d["qoff"][:-1], d["qtok"], ex.fetch("lm"), k["e1"], s1, p1d, _csr(ex.fetch("qo_off"), ex.fetch("qo_ids")),
                               k["e2"], s2, c2d, _csr(ex.fetch("qt_off"), ex.fetch("qt_ids")))
    assert (a, b) == (k["d1"], k["d2"])
    hits_of = bf.hit_lists(s1, _canon(ex.fetch("hits1")), ex.fetch("pidx"), ex.fetch("phit_start"), ex.fetch("phit_len"))
    assert bf.check_extension_rules(c, ex.fetch("sa"), blocks, s1, hits_of, ex.fetch("r1"), k["sep1"], ex.fetch("r2"), k["sep2a"], k["sep2b"]) > 1000
    done = bf.check_maxlex_lines(c, d["tstr"], d["lexk"], d["lexv"], {0: ex.fetch("lex0"), 1: ex.fetch("lex1"), 2: ex.fetch("lex2")}, k["g"], k["d1"], k["d2"], blocks, p1d, c2d, one2)
    assert done > 1000 and k["guard_exits"] == 0
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "toy", "mid", "dense", "sparse", "thin"])
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
                          _canon(ex.fetch("hits1")), ex.fetch("s2"), ex.fetch("c2d"), _canon(ex.fetch("hits2")), ex.fetch("blocks"), ex.fetch("r0"), ex.fetch("r1"), k["sep1"], ex.fetch("r2"), k["sep2a"], k["sep2b"], max_patterns=limit, pick=pick)
    assert done > 100 and k["guard_exits"] == 0
    ex.close()


def _text_corpus(d, fx):
    return bf.Corpus.from_text(d["str"][:d["hdr"]["n"]], os.path.join(fx, "corpus.e"), os.path.join(fx, "corpus.a"))


def test_alignment_built_from_the_text_equals_the_packed_words(oracle_bin, fixtures_dir, tmp_path):
    """bf.Corpus.from_text reads the alignment file itself (plain integer positions); on corpora the reference accepts it must
    give what the packed alignment words and byte tables of the loaders give -- L, R, P, the sentences' target offsets, ltar, rtar."""
    for name in ("tiny", "toy"):
        fx = make_fixture(name, fixtures_dir); dump = str(tmp_path / ("d_%s.bin" % name))
        op.run_oracle(oracle_bin, fx, str(tmp_path / ("o_" + name)), dump)
        d = op.read_dump(dump); n, nt = d["hdr"]["n"], d["hdr"]["nt"]
        a = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"]); b = _text_corpus(d, fx)
        words = np.nonzero(a.s[:n - 2] >= 2)[0]                  # without the two closing sentinels of the token array (Start.cu:300-312)
        na = lambda x, c: np.where(np.asarray(x) == c.NA, -1, np.asarray(x))
        assert (na(a.L[words], a) == na(b.L[words], b)).all() and (na(a.R[words], a) == na(b.R[words], b)).all() and (a.P[words] == b.P[words]).all()
        assert all(a.sentence(int(k)) == b.sentence(int(k)) for k in words[::7])
        m = min(len(a.ltar), len(b.ltar), nt)
        la, ra = na(a.ltar[:m], a), na(a.rtar[:m], a); lb, rb = na(b.ltar[:m], b), na(b.rtar[:m], b)
        un = (la < 0) | (ra < 0)                                # the byte tables mark a word as not aligned in either table
        assert ((lb < 0) == un).all() and (la[~un] == lb[~un]).all() and (ra[~un] == rb[~un]).all()


def test_long_sentence_mode_of_the_oracle_agrees_with_the_definitions(oracle_bin, fixtures_dir, tmp_path):
    """SURVEY 8(f4), the opt-in mode for sentences of 255 tokens and more: the oracle run with --long-sentences on a corpus of
    240..330-token sentence pairs against the definitions, with the alignment taken from the TEXT (plain integers: no byte
    positions, no position codes) -- hit sets, frequent pairs, ab / aXb / aXbXc and the extension rules."""
    import subprocess
    from test_oracle import make_long_fixture
    fx = make_long_fixture(fixtures_dir); dump = str(tmp_path / "d.bin"); out = tmp_path / "o"; out.mkdir()
    subprocess.run([oracle_bin, "--long-sentences"] + op.fixture_args(fx) + [str(out), "--dump", dump], check=True, capture_output=True)
    d = op.read_dump(dump); h = d["hdr"]
    c = _text_corpus(d, fx)
    assert int(c.P.max()) >= 255                                # the corpus really is beyond byte positions
    p1d = d["p1"][d["s1"]["position"]]; c2d = d["p2"]["pat"][:, 0][d["s2"]["position"]]
    pick = sorted(set(range(150)) | set(int(x) for x in np.random.default_rng(5).integers(0, max(h["d1"], 1), 600)))
    done = bf.check_batch(c, d["sa"], d["freq"], _phits(d["pidx"], d["phits"]["start"], d["phits"]["length"]), d["s1"], p1d, d["hits1"],
                          d["s2"], c2d, d["hits2"], d["blocks"], d["r0"], d["r1"], h["sep1"], d["r2"], h["sep2a"], h["sep2b"], max_patterns=150, pick=pick)
    assert done > 100
    hits_of = bf.hit_lists(d["s1"], d["hits1"], d["pidx"], d["phits"]["start"], d["phits"]["length"])
    assert bf.check_extension_rules(c, d["sa"], d["blocks"], d["s1"], hits_of, d["r1"], h["sep1"], d["r2"], h["sep2a"], h["sep2b"]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [128, 1000, 0])
def test_query_token_limit_against_the_definitions(limit, oracle_bin, fixtures_dir, tmp_path):
    """the reference's first kernel launches 128 threads per query sentence (SuffixArray.cu:1374-1378), so a query's
    tokens from the 129th on never match.  Option "k1_limit" (strmatchcuda --query-limit): the default keeps that, 1000 moves the
    limit, 0 lifts it.  Queries of 150..260 tokens cut from the corpus (so that they match far beyond position 128), the HIP path
    against the definitions: longest matches under the limit, enumeration, pattern ids, per-query lists, hit sets and rules."""
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd as cgx
    fx = make_fixture("mid", fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)               # only for the id-level input arrays of the fixture
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    s = np.asarray(d["str"][:n], np.int64); rng = np.random.default_rng(40 + limit)
    words = s[s >= 2]
    qtok, qoff = [], []
    for ln in (150, 260, 129, 7, 200, 128):                              # long and short queries mixed; pieces of corpus text glued together
        qoff.append(len(qtok)); got = 0
        while got < ln:
            a = int(rng.integers(0, len(words) - 20)); m = min(int(rng.integers(3, 12)), ln - got)
            qtok += [int(x) for x in words[a:a + m]]; got += m
    qoff = np.asarray(qoff, np.int32); qtok = np.asarray(qtok, np.int32)
    ex = cgx.Extractor(0)
    if limit != 128:
        ex.set_option("k1_limit", limit)
    ex.upload_index(d["str"][:n], d["rlp"], d["tstr"][:nt], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa(); ex.precompute(); ex.upload_queries(qoff, qtok); ex.sa_lookup(); ex.make_blocks(); ex.gappy_search(); ex.extract(); ex.lexicon(); ex.count_rules()
    c = bf.Corpus(d["str"][:n], d["rlp"], d["ltar"], d["rtar"]); k = ex.counts()
    lm = ex.fetch("lm"); eff = limit if limit > 0 else 1 << 30
    in_pos = np.concatenate([np.arange(b - a) for a, b in zip(qoff, list(qoff[1:]) + [len(qtok)])])
    assert (lm[in_pos >= eff] == 0).all()
    assert (lm[in_pos >= 128] > 0).any() == (eff > 128)                    # the limit is what keeps the late tokens from matching
    s1, s2, p1d, c2d = ex.fetch("s1"), ex.fetch("s2"), ex.fetch("p1d"), ex.fetch("c2d")
    a, b = bf.check_query_side(c, qoff, qtok, lm, k["e1"], s1, p1d, _csr(ex.fetch("qo_off"), ex.fetch("qo_ids")),
                               k["e2"], s2, c2d, _csr(ex.fetch("qt_off"), ex.fetch("qt_ids")), k1_limit=eff)
    assert (a, b) == (k["d1"], k["d2"]) and a > 100
    pick = sorted(set(range(100)) | set(int(x) for x in np.random.default_rng(12).integers(0, max(int(k["d1"]), 1), 800)))
    done = bf.check_batch(c, ex.fetch("sa"), ex.fetch("freq"), _phits(ex.fetch("pidx"), ex.fetch("phit_start"), ex.fetch("phit_len")), s1, p1d,
                          _canon(ex.fetch("hits1")), s2, c2d, _canon(ex.fetch("hits2")), ex.fetch("blocks"), ex.fetch("r0"), ex.fetch("r1"), k["sep1"], ex.fetch("r2"), k["sep2a"], k["sep2b"], max_patterns=100, pick=pick)
    assert done > 100 and k["guard_exits"] == 0
    ex.close()
