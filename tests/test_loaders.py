"""Host loaders (no GPU): the multi-threaded readers must build exactly the corpus a single thread builds --
same first-appearance word ids, same arrays, same alignment bytes -- and report the same errors."""
import os
import shutil

import numpy as np
import pytest

import oracle_py as op
from test_oracle import make_fixture


@pytest.fixture(scope="module")
def cgx():
    import cgx_amd
    cgx_amd.load_library()
    return cgx_amd


def _load(cgx, fx, threads, piece_min):
    old = {k: os.environ.get(k) for k in ("CGX_THREADS", "CGX_LOAD_PIECE_MIN")}
    os.environ["CGX_THREADS"] = str(threads); os.environ["CGX_LOAD_PIECE_MIN"] = str(piece_min)
    try:
        f = op.fixture_args(fx)
        c = cgx.Corpus.load(f[0], f[2], f[3], f[4])
        s = c.checksum(); c.close()
        return s
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", ["tiny", "toy", "mid"])
def test_piecewise_loading_builds_the_same_corpus(name, cgx, fixtures_dir):
    fx = make_fixture(name, fixtures_dir)
    one = _load(cgx, fx, 1, 1 << 30)
    assert one != 0
    for threads in (2, 4, 7, 16, 64):
        assert _load(cgx, fx, threads, 1) == one, threads


def test_error_of_the_earliest_line_wins(cgx, fixtures_dir, tmp_path):
    fx = make_fixture("toy", fixtures_dir); d = tmp_path / "fx"
    shutil.copytree(fx, d)
    lines = (d / "corpus.a").read_text().split("\n")
    lines[len(lines) // 4] += " 3"                      # unpaired number early in the file
    lines[3 * len(lines) // 4] = "300-1"                # out-of-range position later
    (d / "corpus.a").write_text("\n".join(lines))
    msgs = []
    for threads, pm in ((1, 1 << 30), (8, 1)):
        with pytest.raises(cgx.CgxError) as e:
            _load(cgx, str(d), threads, pm)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "Not possible!" in msgs[0]


def test_lexical_entries_that_straddle_lines(cgx, fixtures_dir, tmp_path):
    """`file >> a >> b >> v1 >> v2` does not care about line ends; a table whose entries are split over lines must load
    the same way with many threads (the loader notices the ragged pieces and reads it sequentially)."""
    fx = make_fixture("toy", fixtures_dir); d = tmp_path / "fx"
    shutil.copytree(fx, d)
    rows = (d / "lex.txt").read_text().split()
    (d / "lex.txt").write_text("\n".join(" ".join(rows[i:i + 3]) for i in range(0, len(rows), 3)) + "\n")   # three fields per line
    assert _load(cgx, str(d), 16, 1) == _load(cgx, str(d), 1, 1 << 30) == _load(cgx, fx, 1, 1 << 30)


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_corpus_cache_round_trip(name, cgx, fixtures_dir, tmp_path):
    """cgx_corpus_save / cgx_corpus_load_cache: the reloaded corpus is the corpus (checksum over every array and
    spelling), and damaged files are refused."""
    fx = make_fixture(name, fixtures_dir); f = op.fixture_args(fx)
    c = cgx.Corpus.load(f[0], f[2], f[3], f[4]); want = c.checksum()
    path = str(tmp_path / "corpus.cgx"); c.save(path); c.close()
    c2 = cgx.Corpus.load_cache(path); assert c2.checksum() == want; c2.close()
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:-7])
    with pytest.raises(cgx.CgxError, match="truncated or corrupt"):
        cgx.Corpus.load_cache(path)
    open(path, "wb").write(b"NOTACGX!" + blob[8:])
    with pytest.raises(cgx.CgxError, match="not a corpus cache"):
        cgx.Corpus.load_cache(path)
    with pytest.raises(cgx.CgxError, match="cannot open"):
        cgx.Corpus.load_cache(str(tmp_path / "missing.cgx"))


def test_corpus_cache_is_not_trusted(cgx, fixtures_dir, tmp_path):
    """A cache of the right size but with flipped bytes fails its checksum or its range checks; a cache whose text files
    changed afterwards is reported stale; the file appears under its final name only when complete (no .tmp left behind)."""
    import ctypes as C
    fx = make_fixture("tiny", fixtures_dir); d = tmp_path / "fx"; shutil.copytree(fx, d); f = op.fixture_args(str(d))
    c = cgx.Corpus.load(f[0], f[2], f[3], f[4])
    path = str(tmp_path / "corpus.cgx"); c.save(path); c.close()
    assert [n for n in os.listdir(tmp_path) if ".tmp." in n] == []
    lib = cgx.load_library()
    lib.cgx_corpus_matches_sources.argtypes = [C.c_void_p] + [C.c_char_p] * 4
    c2 = cgx.Corpus.load_cache(path)
    args = [f[0].encode(), f[2].encode(), f[3].encode(), f[4].encode()]
    assert lib.cgx_corpus_matches_sources(c2.h, *args) == 1
    with open(f[0], "a") as fh:
        fh.write("one more sentence\n")
    assert lib.cgx_corpus_matches_sources(c2.h, *args) == 0            # stale: the CLI reparses
    os.remove(f[2])
    assert lib.cgx_corpus_matches_sources(c2.h, *args) == -1           # the text files are gone: the cache is all there is
    c2.close()
    blob = bytearray(open(path, "rb").read())
    hdr = 8 + 8 + 8 * 4 + 16 + 64                                      # magic, checksum, eight counts, two byte totals, source fingerprint
    bad = bytearray(blob); bad[hdr + 4 * 5] ^= 0x01                    # a source token id changes: the checksum no longer matches
    open(path, "wb").write(bad)
    with pytest.raises(cgx.CgxError, match="corrupt"):
        cgx.Corpus.load_cache(path)
    bad = bytearray(blob); bad[hdr + 4 * 5 + 3] = 0x7F                 # ... to a value outside the vocabulary: refused by the range check
    open(path, "wb").write(bad)
    with pytest.raises(cgx.CgxError, match="source token id"):
        cgx.Corpus.load_cache(path)


def test_mutated_inputs_never_fault(cgx):
    """tools/fuzz_loaders.py: mutated corpus files and cache files come back as a corpus or as an error message (the same sweep
    runs under ASan / UBSan with tools/asan_host.sh; here the product build, in a child process so that a fault is a test failure)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_loaders.py"), "--cases", "80", "--seed", "5"], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    assert b"fuzz_loaders: 80 cases" in r.stdout


def test_corpus_cache_keeps_long_sentence_mode(cgx, fixtures_dir, tmp_path):
    """Cache format 4: a corpus parsed with CGX_CORPUS_LONG_SENTENCES -- sentence pairs of 240..330 tokens, which the
    reference refuses -- is written with its 16-bit target tables and comes back as the same corpus in the same mode; a corpus
    within the reference's limits comes back in the mode it was parsed in; a flipped 16-bit position fails the checks."""
    from test_oracle import make_long_fixture
    lf = make_long_fixture(fixtures_dir); files = op.fixture_args(lf)
    with pytest.raises(cgx.CgxError, match="too long sentence"):
        cgx.Corpus.load(files[0], files[2], files[3], files[4])
    c = cgx.Corpus.load(files[0], files[2], files[3], files[4], long_sentences=True)
    assert c.flags() == 1
    path = str(tmp_path / "long.cgx"); c.save(path); want = c.checksum(); c.close()
    c2 = cgx.Corpus.load_cache(path); assert c2.checksum() == want and c2.flags() == 1; c2.close()
    fx = make_fixture("tiny", fixtures_dir); f = op.fixture_args(fx)
    for mode in (False, True):
        c = cgx.Corpus.load(f[0], f[2], f[3], f[4], long_sentences=mode); p2 = str(tmp_path / ("tiny%d.cgx" % mode)); c.save(p2); w = c.checksum(); c.close()
        c2 = cgx.Corpus.load_cache(p2); assert c2.checksum() == w and c2.flags() == int(mode); c2.close()
    assert os.path.getsize(str(tmp_path / "tiny1.cgx")) > os.path.getsize(str(tmp_path / "tiny0.cgx"))
    # the 16-bit tables sit behind the byte tables: a position pushed out of range there is caught (range check or checksum)
    raw = bytearray(open(path, "rb").read()); hdr = np.frombuffer(bytes(raw[:48]), np.uint32)
    n, nt, nsent = int(hdr[4]), int(hdr[5]), int(hdr[6])
    hdr_bytes = 8 + 8 + 8 * 4 + 16 + 64
    off16 = hdr_bytes + n * 4 + nt * 4 + (nsent + 1) * 8 + n + n * 4 + nt + nt
    raw[off16 + 1] = 0x7F                                     # high byte of the first 16-bit position: 0x7Fxx is beyond 1024 and not the "not aligned" mark
    bad = str(tmp_path / "bad.cgx"); open(bad, "wb").write(bytes(raw))
    with pytest.raises(cgx.CgxError, match="corrupt"):
        cgx.Corpus.load_cache(bad)
