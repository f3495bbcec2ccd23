"""CPU tests of the oracle (test infrastructure) against the committed goldens and, where the
reference checkout exists, against the reference's own object code (oracle/_ref)."""
import json
import os
import shutil
import subprocess
import tarfile

import numpy as np
import pytest

import gen_fixture
import oracle_py as op

ROOT = op.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "golden.json")))


def make_fixture(name, base):
    pairs, vocab, queries, seed, longq = META[name]["spec"]
    fx = os.path.join(base, name)
    if name == "tiny":
        return os.path.join(GOLD, "tiny")
    if not os.path.exists(os.path.join(fx, "lex.txt")):
        gen_fixture.write_fixture(fx, pairs, vocab, queries, seed, long_query=longq)
    got = {n: op.hashlib.sha256(open(os.path.join(fx, n), "rb").read()).hexdigest() for n in META[name]["inputs"]}
    if got != META[name]["inputs"]:
        pytest.skip("fixture generator drifted from the one that made golden.json (numpy version?)")
    return fx


@pytest.mark.parametrize("name", ["tiny", "toy", "mid"])
def test_oracle_matches_golden(name, oracle_bin, fixtures_dir, tmp_path):
    fx = make_fixture(name, fixtures_dir)
    op.run_oracle(oracle_bin, fx, str(tmp_path))
    assert op.sha_dir(str(tmp_path), META[name]["spec"][2]) == META[name]["grammar"]


def test_golden_tarball_matches_hashes(tmp_path):
    with tarfile.open(os.path.join(GOLD, "tiny_expected.tar.gz")) as tar:
        tar.extractall(str(tmp_path))
    assert op.sha_dir(str(tmp_path), 7) == META["tiny"]["grammar"]


@pytest.mark.skipif(not os.path.exists("/root/reference/ExtractPair.c"), reason="reference checkout not present on this machine")
@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_oracle_host_stages_match_reference_objects(name, oracle_bin, fixtures_dir, tmp_path):
    """SuffixArray.c / ExtractPair.c / PrintResults.c compiled in place (oracle/_ref) must agree with the restatement."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_harness"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    fx = make_fixture(name, fixtures_dir)
    out = tmp_path / "o"; rout = tmp_path / "r"; rout.mkdir()
    dump = str(tmp_path / "dump.bin")
    op.run_oracle(oracle_bin, fx, str(out), dump)
    r = subprocess.run([ref, "sa", dump], capture_output=True, text=True)
    assert r.returncode == 0 and "SA OK" in r.stdout
    r = subprocess.run([ref, "grammar", dump, str(rout)], capture_output=True, text=True)
    assert r.returncode == 0 and "GRAMMAR OK" in r.stdout, r.stdout
    nq = META[name]["spec"][2]
    assert op.sha_dir(str(out), nq) == op.sha_dir(str(rout), nq)


@pytest.mark.skipif(not os.path.exists("/root/reference/ExtractPair.c"), reason="reference checkout not present on this machine")
def test_reference_objects_on_a_thinly_aligned_corpus(oracle_bin, fixtures_dir, tmp_path):
    """The reference's own createLexicon*Fast / print_query_GPU_Gappy (oracle/_ref) fed with the oracle's intermediates of a corpus that has
    lost 45 % of its alignment links -- phrase edges without a link, gaps without an aligned token, the reference's unsigned-char
    "no position" marks (255) everywhere: the files must be the oracle's, byte for byte (short-sentence mode, the reference's format)."""
    import test_bruteforce as tb
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_harness"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    fx = tb.make_fixture("thin", fixtures_dir)
    out = tmp_path / "o"; rout = tmp_path / "r"; rout.mkdir(); dump = str(tmp_path / "dump.bin")
    op.run_oracle(oracle_bin, fx, str(out), dump)
    r = subprocess.run([ref, "sa", dump], capture_output=True, text=True)
    assert r.returncode == 0 and "SA OK" in r.stdout
    r = subprocess.run([ref, "grammar", dump, str(rout)], capture_output=True, text=True)
    assert r.returncode == 0 and "GRAMMAR OK" in r.stdout, r.stdout
    nq = tb.EXTRA["thin"]["queries"]
    assert op.sha_dir(str(out), nq) == op.sha_dir(str(rout), nq)
    assert sum(os.path.getsize(out / ("grammar.%d.s" % q)) for q in range(nq)) > 10000


def test_query_sharding_is_output_invariant(oracle_bin, tmp_path):
    """Per-query files depend only on the index and that query: split 7 queries 4+3."""
    fx = os.path.join(GOLD, "tiny")
    lines = open(os.path.join(fx, "query.f")).read().splitlines(True)
    for part, sl in (("a", lines[:4]), ("b", lines[4:])):
        d = tmp_path / part; d.mkdir()
        for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt"):
            shutil.copy(os.path.join(fx, n), d / n)
        (d / "query.f").write_text("".join(sl))
        op.run_oracle(oracle_bin, str(d), str(tmp_path / ("out_" + part)))
    got = op.sha_dir(str(tmp_path / "out_a"), 4) + op.sha_dir(str(tmp_path / "out_b"), 3)
    assert got == META["tiny"]["grammar"]


def test_edge_queries(oracle_bin, tmp_path):
    """Empty line, all-OOV line, one-token line, repeated token: must run and write one file per line."""
    fx = os.path.join(GOLD, "tiny"); d = tmp_path / "fx"; d.mkdir()
    for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt"):
        shutil.copy(os.path.join(fx, n), d / n)
    (d / "query.f").write_text("\nOOV1 OOV2\ns1\ns1 s1 s1 s1 s1 s1\n")
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "out"))
    sizes = [os.path.getsize(tmp_path / "out" / ("grammar.%d.s" % q)) for q in range(4)]
    assert sizes[0] == 0 and sizes[1] == 0 and sizes[2] > 0 and sizes[3] > 0


def test_dump_invariants(oracle_bin, tmp_path):
    fx = os.path.join(GOLD, "tiny"); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); n = d["hdr"]["n"]
    sa, s = d["sa"], d["str"]
    assert sorted(sa.tolist()) == list(range(n))
    first = s[sa]
    assert np.all(np.diff(first) >= 0)                       # suffixes ordered by first token
    h1 = d["hits1"]
    key = h1["position"].astype(np.int64) * (1 << 40) + h1["str_position"].astype(np.int64) * 16 + h1["length"]
    assert np.all(np.diff(key) > 0)                          # canonical order, no duplicates
    assert d["freq"].tolist() == sorted(d["freq"].tolist()) and len(set(d["freq"].tolist())) == 100
    lm = d["lm"]; up = d["up"].reshape(-1, 5); down = d["down"].reshape(-1, 5)
    for t in range(len(lm)):
        for l in range(min(lm[t], 5)):
            assert up[t, l] <= down[t, l]
            if l:
                assert up[t, l - 1] <= up[t, l] and down[t, l] <= down[t, l - 1]   # nested intervals


LONG_SPEC = dict(pairs=120, vocab=160, queries=6, seed=11, lo=240, hi=330)       # sentence pairs of 240..330 tokens: beyond the reference's byte positions


def make_long_fixture(base):
    fx = os.path.join(base, "long")
    if not os.path.exists(os.path.join(fx, "lex.txt")):
        gen_fixture.write_fixture(fx, **LONG_SPEC)
    return fx


def test_long_sentence_mode_of_the_oracle(oracle_bin, fixtures_dir, tmp_path):
    """--long-sentences: wider positions change nothing for sentences the reference accepts (the tiny fixture gives
    the golden files with the switch on), and a corpus of 240..330-token sentences, which the default mode refuses with the
    reference's message and exit code (ExtractPair.cu:2683), runs with it."""
    fx = os.path.join(GOLD, "tiny"); out = tmp_path / "t"; out.mkdir()
    subprocess.run([oracle_bin, "--long-sentences"] + op.fixture_args(fx) + [str(out)], check=True, capture_output=True)
    assert op.sha_dir(str(out), 7) == META["tiny"]["grammar"]
    lf = make_long_fixture(fixtures_dir); o2 = tmp_path / "l"; o2.mkdir()
    r = subprocess.run([oracle_bin] + op.fixture_args(lf) + [str(o2)], capture_output=True, text=True)
    assert r.returncode == 1 and "Not possible, too long sentence" in r.stdout
    r = subprocess.run([oracle_bin, "--long-sentences"] + op.fixture_args(lf) + [str(o2)], capture_output=True, text=True)
    assert r.returncode == 0 and sum(1 for _ in open(o2 / "grammar.0.s")) > 1000
