"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI must be
bit-exact against the CPU oracle, stage by stage and in the final grammar files."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_py as op
from test_oracle import META, make_fixture

pytestmark = pytest.mark.gpu
ROOT = op.ROOT


@pytest.fixture(scope="module")
def cgx():
    # torch first, as in bench.py: its bundled HIP runtime must be the one that initialises the device,
    # the replica test uses a torch tensor as the staging buffer
    import torch
    torch.zeros(1, device="cuda:0")
    import cgx_amd
    cgx_amd.load_library()
    return cgx_amd


def canon_hits(h):
    """The hit records of cgx_fetch in the reference's order (pattern, start, length[, length2]).  By default the device keeps a
    pattern's list ordered only by the top bits of the start position -- extraction needs order statistics, not the order
    (cgx_search.inc, hitview) -- so a fetched list is put in order here before it is compared; option hit_order = 1 sorts on the card."""
    keys = [h[f] for f in reversed([f for f in ("position", "str_position", "length", "length2") if f in h.dtype.names])]
    return h[np.lexsort(keys)]


def run_product(cgx, fx, outdir, **opts):
    files = op.fixture_args(fx)
    ex = cgx.Extractor(0)
    for k, v in opts.items():
        ex.set_option(k, v)
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4])
    ex.upload_corpus(corpus)
    os.makedirs(outdir, exist_ok=True)
    n = ex.extract_grammars(corpus, files[1], outdir)
    return ex, corpus, n


@pytest.mark.parametrize("device_format", [1, 0])
@pytest.mark.parametrize("name", ["tiny", "toy", "mid"])
def test_grammar_files_bit_exact(name, device_format, cgx, oracle_bin, fixtures_dir, tmp_path):
    """device_format=1: text laid out by the GPU formatter; 0: the threaded host formatter.  Both must equal the oracle."""
    fx = make_fixture(name, fixtures_dir)
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"))
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "p"), device_format=device_format)
    assert n == sum(sum(1 for _ in open(tmp_path / "o" / ("grammar.%d.s" % q), "rb")) for q in range(META[name]["spec"][2]))
    nq = META[name]["spec"][2]
    assert op.sha_dir(str(tmp_path / "p"), nq) == op.sha_dir(str(tmp_path / "o"), nq) == META[name]["grammar"]
    assert ex.counts()["guard_exits"] == 0
    ex.close(); corpus.close()


def test_corpus_cache_gives_the_same_files(cgx, fixtures_dir, tmp_path):
    """--index-cache: the first run parses the text files, builds the index and writes both caches (parsed corpus, built
    index); the second reads only the caches (the text corpus is gone by then; no suffix-array construction, no
    frequent-pair precomputation); a damaged index cache is detected and rebuilt.  All runs produce the golden files."""
    fx = make_fixture("toy", fixtures_dir); d = tmp_path / "fx"; shutil.copytree(fx, d)
    cache = str(tmp_path / "toy.cgx"); exe = os.path.join(ROOT, "bin", "strmatchcuda")
    for run in (0, 1, 2):
        out = tmp_path / ("o%d" % run); out.mkdir()
        r = subprocess.run([exe, "--index-cache", cache] + op.fixture_args(str(d)) + [str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert ("corpus read from cache" in r.stderr) == (run >= 1)
        assert ("index read from cache" in r.stderr) == (run == 1), r.stderr
        assert ("suffix array 0.0 ms, frequent pairs 0.0 ms" in r.stderr) == (run == 1)
        assert op.sha_dir(str(out), 7) == META["toy"]["grammar"]
        if run == 0:
            for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt"):
                os.remove(d / n)
        if run == 1:                                             # flip one byte in the middle of the index cache
            blob = bytearray(open(cache + ".idx", "rb").read()); blob[len(blob) // 2] ^= 0x40
            open(cache + ".idx", "wb").write(blob)
    # library level: a second context loads the file instead of building, a wrong corpus checksum is refused
    files = op.fixture_args(fx)
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4]); ck = corpus.checksum()
    ex = cgx.Extractor(0); ex.upload_corpus(corpus); path = str(tmp_path / "lib.idx"); ex.index_save(path, ck)
    ex2 = cgx.Extractor(0)
    with pytest.raises(cgx.CgxError, match="another corpus"):
        ex2.index_load(path, ck ^ 1)
    ex2.index_load(path, ck)
    assert np.array_equal(ex2.fetch("sa"), ex.fetch("sa")) and np.array_equal(ex2.fetch("phit_start"), ex.fetch("phit_start"))
    out = tmp_path / "lib"; out.mkdir()
    ex2.extract_grammars(corpus, files[1], str(out))
    assert op.sha_dir(str(out), 7) == META["toy"]["grammar"]
    ex.close(); ex2.close(); corpus.close()


def test_cli_is_a_drop_in(cgx, oracle_bin, fixtures_dir, tmp_path):
    """bin/strmatchcuda with the reference's six positionals writes the same files."""
    fx = make_fixture("tiny", fixtures_dir); out = tmp_path / "cli"; out.mkdir()
    r = subprocess.run([os.path.join(ROOT, "bin", "strmatchcuda")] + op.fixture_args(fx) + [str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Start Printing Gappy Phrases..." in r.stderr
    assert op.sha_dir(str(out), 7) == META["tiny"]["grammar"]
    r = subprocess.run([os.path.join(ROOT, "bin", "strmatchcuda")] + op.fixture_args(fx) + [str(tmp_path / "missing_dir")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "It is not valid" in r.stderr          # PrintResults.c:443-446


def test_cli_query_limit_switch(cgx, fixtures_dir, tmp_path):
    """strmatchcuda --query-limit N: the default and 128 keep the reference's 128 looked-up tokens per query sentence
    (SuffixArray.cu:1374-1378), 0 lifts the limit: a 300-token query then yields rules for its late tokens too."""
    fx = make_fixture("toy", fixtures_dir); files = op.fixture_args(fx)
    words = open(files[0]).read().split()
    q = tmp_path / "long.f"; q.write_text(" ".join(words[:300]) + "\n" + " ".join(words[300:320]) + "\n")
    args = [files[0], str(q), files[2], files[3], files[4]]
    sizes = {}
    for tag, extra in (("default", []), ("l128", ["--query-limit", "128"]), ("l0", ["--query-limit", "0"]), ("l200", ["--query-limit", "200"])):
        out = tmp_path / tag; out.mkdir()
        r = subprocess.run([os.path.join(ROOT, "bin", "strmatchcuda")] + extra + args + [str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        sizes[tag] = [open(str(out / ("grammar.%d.s" % i)), "rb").read() for i in range(2)]
    assert sizes["default"] == sizes["l128"]
    assert sizes["l0"][1] == sizes["default"][1]                              # the short query does not notice
    assert len(sizes["default"][0]) < len(sizes["l200"][0]) < len(sizes["l0"][0])
    r = subprocess.run([os.path.join(ROOT, "bin", "strmatchcuda"), "--query-limit", "-3"] + args + [str(tmp_path / "l0")], capture_output=True, text=True, timeout=60)
    assert "Please check your input arguments" in r.stdout


def _gunzip_three_ways(path):
    """The bytes of a .gz file as Python's gzip, the zcat program and zlib's gzread give them; all three must agree."""
    import ctypes as C
    import gzip
    a = gzip.open(path, "rb").read()
    b = subprocess.run(["zcat", str(path)], capture_output=True, check=True).stdout
    z = C.CDLL("libz.so.1"); z.gzopen.restype = C.c_void_p; z.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    z.gzread.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]; z.gzclose.argtypes = [C.c_void_p]
    f = z.gzopen(str(path).encode(), b"rb"); assert f
    buf = C.create_string_buffer(1 << 20); c = bytearray()
    while True:
        n = z.gzread(f, buf, len(buf))
        assert n >= 0
        if n == 0:
            break
        c += buf.raw[:n]
    assert z.gzclose(f) == 0
    assert a == b == bytes(c), path
    return a


@pytest.mark.parametrize("writer", ["gpu_deflate", "gpu_deflate_fixed_codes", "host_zlib_on_unique_text", "host_formatter_zlib"])
@pytest.mark.parametrize("name", ["toy", "mid"])
def test_gzip_output_holds_the_same_bytes(name, writer, cgx, fixtures_dir, tmp_path):
    """Option gz_level / strmatchcuda --gz N: grammar.<q>.s.gz, whose content is the golden file, for the writers: DEFLATE blocks
    emitted by the GPU formatter itself (the default with the device formatter; the file is one gzip member around them) with the
    batch's own Huffman codes or (gz_dynamic = 0) the fixed ones, the host's zlib over the pieces of the plain unique text
    (gz_device = 0), the host formatter's zlib.  Read back with Python's gzip, zcat and zlib's gzread."""
    import hashlib
    fx = make_fixture(name, fixtures_dir); nq = META[name]["spec"][2]
    opts = dict(gpu_deflate=dict(gz_level=6), gpu_deflate_fixed_codes=dict(gz_level=6, gz_dynamic=0), host_zlib_on_unique_text=dict(gz_level=6, gz_device=0), host_formatter_zlib=dict(gz_level=6, device_format=0))[writer]
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "z"), **opts)
    got = [hashlib.sha256(_gunzip_three_ways(tmp_path / "z" / ("grammar.%d.s.gz" % q))).hexdigest() for q in range(nq)]
    assert got == META[name]["grammar"] and not os.path.exists(tmp_path / "z" / "grammar.0.s")
    if writer.startswith("gpu_deflate"):
        assert ex.stage_ms("fmt_gz") == 1.0 and 0 < ex.stage_ms("fmt_unique_bytes") < (0.4 if writer == "gpu_deflate" else 0.5) * ex.stage_ms("fmt_plain_unique_bytes")
        assert ex.stage_ms("fmt_gz_dynamic") == (1.0 if writer == "gpu_deflate" else 0.0)
    ex.close(); corpus.close()
    if writer == "gpu_deflate" and name == "toy":
        out = tmp_path / "cli"; out.mkdir()
        r = subprocess.run([os.path.join(ROOT, "bin", "strmatchcuda"), "--gz", "1"] + op.fixture_args(fx) + [str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert [hashlib.sha256(_gunzip_three_ways(out / ("grammar.%d.s.gz" % q))).hexdigest() for q in range(7)] == META["toy"]["grammar"]


def test_gzip_pieces_edge_cases(cgx, oracle_bin, tmp_path):
    """Queries without rules become an empty gzip member (header + trailer: 20 bytes); the async writer and sub-batches keep working
    with deflate pieces."""
    fx = os.path.join(ROOT, "tests", "golden", "tiny"); d = tmp_path / "fx"; d.mkdir()
    for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt"):
        shutil.copy(os.path.join(fx, n), d / n)
    (d / "query.f").write_text("\nOOV1 OOV2\ns1\ns1 s1 s1 s1 s1 s1\ns0 s2 OOV s1 s0\n\n")
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "o"))
    for k, opts in enumerate((dict(), dict(async_write=1, sub_batch=2))):
        ex, corpus, n = run_product(cgx, str(d), str(tmp_path / ("p%d" % k)), gz_level=1, **opts)
        ex.flush()
        for q in range(6):
            assert _gunzip_three_ways(tmp_path / ("p%d" % k) / ("grammar.%d.s.gz" % q)) == open(tmp_path / "o" / ("grammar.%d.s" % q), "rb").read()
        assert os.path.getsize(tmp_path / ("p%d" % k) / "grammar.0.s.gz") == 20          # header, 03 00, CRC-32 0, ISIZE 0
        ex.close(); corpus.close()


def test_gzip_very_long_lines(cgx, oracle_bin, tmp_path):
    """A vocabulary of 75-byte words: lines of a kilobyte and more, back-references longer than one DEFLATE match (258 bytes: split), distances
    in the thousands.  The .gz files must inflate to the oracle's, and the plain files equal them too."""
    fx = os.path.join(ROOT, "tests", "golden", "tiny"); d = tmp_path / "fx"; d.mkdir()
    long_word = lambda w: w + "_" + "z" * 70
    for n in ("corpus.f", "corpus.e", "query.f"):
        (d / n).write_text("".join(" ".join(long_word(w) for w in line.split()) + "\n" for line in open(os.path.join(fx, n))))
    shutil.copy(os.path.join(fx, "corpus.a"), d / "corpus.a")
    (d / "lex.txt").write_text("".join(" ".join([w if w == "NULL" else long_word(w) for w in line.split()[:2]] + line.split()[2:]) + "\n" for line in open(os.path.join(fx, "lex.txt"))))
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "o"))
    ex, corpus, n = run_product(cgx, str(d), str(tmp_path / "z"), gz_level=1)
    sizes = []
    for q in range(7):
        want = open(tmp_path / "o" / ("grammar.%d.s" % q), "rb").read()
        assert _gunzip_three_ways(tmp_path / "z" / ("grammar.%d.s.gz" % q)) == want
        sizes.append(len(want))
    assert max(sizes) > 100000 and ex.stage_ms("fmt_gz") == 1.0
    ex.close(); corpus.close()
    ex, corpus, n = run_product(cgx, str(d), str(tmp_path / "p"))
    assert all(open(tmp_path / "p" / ("grammar.%d.s" % q), "rb").read() == open(tmp_path / "o" / ("grammar.%d.s" % q), "rb").read() for q in range(7))
    ex.close(); corpus.close()


def test_staged_text_api_reassembles_the_files(cgx, oracle_bin, fixtures_dir, tmp_path):
    """The writer's public interface, used the way INTEGRATION.md tells a binder to: stage calls up to cgx_lexicon, then
    cgx_upload_vocab / cgx_upload_score_tables / cgx_format, the unique text and the piece lists (cgx_text_info,
    cgx_text_segments, cgx_text_offsets, cgx_text_read) -- files reassembled here in Python are the golden files."""
    import ctypes as C
    import hashlib
    fx = make_fixture("toy", fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); h = d["hdr"]; nq = h["nq"]
    ex = cgx.Extractor(0)
    ex.upload_index(d["str"][:h["n"]], d["rlp"], d["tstr"][:h["nt"]], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa(); ex.precompute(); ex.upload_queries(d["qoff"][:-1], d["qtok"]); ex.sa_lookup(); ex.make_blocks(); ex.gappy_search(); ex.extract(); ex.lexicon()
    ex.upload_vocab(d["svocab"].split(b"\0")[:-1], d["tvocab"].split(b"\0")[:-1])
    libm = C.CDLL("libm.so.6"); libm.log10f.restype = C.c_float; libm.log10f.argtypes = [C.c_float]; libm.log10.restype = C.c_double; libm.log10.argtypes = [C.c_double]
    N = 302                                                             # ExtractPair.c:652-656 with the host libm, as the library's own tables
    aa = np.zeros((N, N), np.float32); bb = np.zeros(N, np.float32); fs = np.zeros(N, np.float32)
    for p in range(N):
        bb[p] = fs[p] = np.float32(libm.log10(float(1 + p)))
        for f in range(1, N):
            aa[p, f] = -libm.log10f(np.float32(np.float32(p) / np.float32(f)))
    ex.upload_score_tables(aa, bb, fs)
    nbytes, nlines, slot = ex.format()
    text, qseg, seg_off, seg_len, qtext = ex.text(slot, nq)
    assert int(qtext[nq]) == nbytes and int(qseg[nq]) == len(seg_off)
    got = []
    for q in range(nq):
        body = b"".join(text[int(seg_off[s]):int(seg_off[s]) + int(seg_len[s])] for s in range(int(qseg[q]), int(qseg[q + 1])))
        assert len(body) == int(qtext[q + 1] - qtext[q])
        got.append(hashlib.sha256(body).hexdigest())
    assert got == META["toy"]["grammar"]
    assert nlines == sum(sum(1 for _ in open(tmp_path / "o" / ("grammar.%d.s" % q), "rb")) for q in range(nq))
    assert len(text) < nbytes                                            # shared lines are stored once
    assert ex.text_encoding(slot) == 0
    # the same as deflate pieces: a file = gzip header + its pieces + 03 00 + CRC-32 + ISIZE (cgx_text_trailers), and every piece
    # by itself is a byte-aligned stretch of a deflate stream
    import gzip
    import struct
    import zlib
    ex.set_option("gz_level", 1)
    zbytes, zlines, zslot = ex.format()
    assert ex.text_encoding(zslot) == 1 and zlines == nlines and zbytes < nbytes // 2
    ztext, qseg, seg_off, seg_len, qtext = ex.text(zslot, nq)
    trl = ex.text_trailers(zslot, nq)
    hdr = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3])
    zgot = []
    for q in range(nq):
        pieces = [ztext[int(seg_off[s]):int(seg_off[s]) + int(seg_len[s])] for s in range(int(qseg[q]), int(qseg[q + 1]))]
        body = hdr + b"".join(pieces) + b"\x03\x00" + struct.pack("<II", int(trl[q, 0]), int(trl[q, 1]))
        assert len(body) == int(qtext[q + 1] - qtext[q])
        plain = gzip.decompress(body)                                     # checks CRC-32 and ISIZE
        assert int(trl[q, 1]) == len(plain) & 0xFFFFFFFF and int(trl[q, 0]) == zlib.crc32(plain)
        parts = b"".join(zlib.decompress(p + b"\x03\x00", wbits=-15) for p in pieces)    # a piece by itself: raw deflate, closed by an empty final block
        assert parts == plain
        zgot.append(hashlib.sha256(plain).hexdigest())
    assert zgot == META["toy"]["grammar"]
    ex.close()


def test_cli_query_shards_union_is_the_whole(cgx, fixtures_dir, tmp_path):
    """strmatchcuda --shard i/n (one process per GPU; contiguous shards balanced by token count): the union of the shards'
    files is the single-process output."""
    fx = make_fixture("mid", fixtures_dir); out = tmp_path / "sh"; out.mkdir(); exe = os.path.join(ROOT, "bin", "strmatchcuda")
    for i in range(3):
        r = subprocess.run([exe, "--shard", "%d/3" % i] + op.fixture_args(fx) + [str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    nq = META["mid"]["spec"][2]
    assert sorted(os.listdir(out)) == sorted("grammar.%d.s" % q for q in range(nq))
    assert op.sha_dir(str(out), nq) == META["mid"]["grammar"]


def test_cli_query_shards_run_in_several_internal_batches(cgx, fixtures_dir, tmp_path):
    """A shard other than the first whose queries take several internal batches (what a shard of more than 300 000 query
    tokens does by itself; --sub-batch forces it here): the offsets of a shard start in the middle of the query token array."""
    fx = make_fixture("mid", fixtures_dir); out = tmp_path / "sh"; out.mkdir(); exe = os.path.join(ROOT, "bin", "strmatchcuda")
    for i in range(3):
        r = subprocess.run([exe, "--shard", "%d/3" % i, "--sub-batch", "4"] + op.fixture_args(fx) + [str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    nq = META["mid"]["spec"][2]
    assert sorted(os.listdir(out)) == sorted("grammar.%d.s" % q for q in range(nq))
    assert op.sha_dir(str(out), nq) == META["mid"]["grammar"]


def test_long_sentences_option(cgx, oracle_bin, fixtures_dir, tmp_path):
    """strmatchcuda --long-sentences (f4; the reference rejects sentences of 255 tokens and more): a corpus of 240..330-token
    sentence pairs gives the files of the oracle run with the same switch; without the switch it is refused with the
    reference's message and exit code; and the switch changes nothing for a corpus the reference accepts."""
    from test_oracle import make_long_fixture, LONG_SPEC
    exe = os.path.join(ROOT, "bin", "strmatchcuda"); lf = make_long_fixture(fixtures_dir); nq = LONG_SPEC["queries"]
    want = tmp_path / "want"; want.mkdir(); got = tmp_path / "got"; got.mkdir()
    subprocess.run([oracle_bin, "--long-sentences"] + op.fixture_args(lf) + [str(want)], check=True, capture_output=True)
    r = subprocess.run([exe] + op.fixture_args(lf) + [str(got)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Not possible, too long sentence" in r.stdout and not os.listdir(got)
    r = subprocess.run([exe, "--long-sentences"] + op.fixture_args(lf) + [str(got)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert op.sha_dir(str(got), nq) == op.sha_dir(str(want), nq)
    # library interface, and a corpus within the reference's limits
    fx = make_fixture("tiny", fixtures_dir); files = op.fixture_args(fx); out = tmp_path / "tiny"; out.mkdir()
    ex = cgx.Extractor(0); corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4], long_sentences=True)
    ex.upload_corpus(corpus); ex.extract_grammars(corpus, files[1], str(out))
    assert op.sha_dir(str(out), 7) == META["tiny"]["grammar"]
    ex.close(); corpus.close()


def test_long_sentence_caches_and_id_level_corpus(cgx, oracle_bin, fixtures_dir, tmp_path):
    """f4, the rest: (1) strmatchcuda --long-sentences --index-cache writes a corpus cache (format 4: the 16-bit target tables
    travel with it) and an index cache, a second run reads both and writes the same files, and a run WITHOUT the switch does not
    take the long cache for its own; (2) cgx_corpus_from_ids16 -- the id-level constructor with 16-bit positions -- gives the
    same rules and lexicon lines as the corpus parsed from the text, stage by stage."""
    import bruteforce as bf
    from test_oracle import make_long_fixture, LONG_SPEC
    exe = os.path.join(ROOT, "bin", "strmatchcuda"); lf = make_long_fixture(fixtures_dir); nq = LONG_SPEC["queries"]; files = op.fixture_args(lf)
    want = tmp_path / "want"; want.mkdir(); dump = str(tmp_path / "d.bin")
    subprocess.run([oracle_bin, "--long-sentences"] + files + [str(want), "--dump", dump], check=True, capture_output=True)
    cache = str(tmp_path / "long.cache")
    for run, note in ((1, "cache"), (2, "read from cache")):
        got = tmp_path / ("got%d" % run); got.mkdir()
        r = subprocess.run([exe, "--long-sentences", "--index-cache", cache] + files + [str(got)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert ("corpus read from cache" in r.stderr) == (run == 2) and ("index read from cache" in r.stderr) == (run == 2), r.stderr
        assert op.sha_dir(str(got), nq) == op.sha_dir(str(want), nq)
    assert os.path.exists(cache) and os.path.exists(cache + ".idx")
    got = tmp_path / "got3"; got.mkdir()                                # the default mode must not adopt a cache written under the other position width
    r = subprocess.run([exe, "--index-cache", cache] + files + [str(got)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Not possible, too long sentence" in r.stdout and "written with --long-sentences" in r.stderr
    # (2) the same corpus at id level
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    c = bf.Corpus.from_text(d["str"][:n], files[2], files[3])
    s = np.asarray(d["str"][:n]); t = np.asarray(d["tstr"][:nt])
    sentind = np.concatenate(([0], np.nonzero(s[:n - 2] == 1)[0] + 1)).astype(np.int32)
    tsentind = np.concatenate(([0], np.nonzero(t[:nt - 2] == 1)[0] + 1)).astype(np.int32)
    assert len(sentind) == len(tsentind) == LONG_SPEC["pairs"] + 1
    w16 = lambda a, m: np.where(np.asarray(a[:m]) == c.NA, 0xFFFF, np.asarray(a[:m])).astype(np.uint16)
    lt = np.full(nt, c.NA, np.int64); m = min(nt, len(c.ltar)); lt[:m] = c.ltar[:m]; rt = np.full(nt, c.NA, np.int64); rt[:m] = c.rtar[:m]
    ids = cgx.Corpus.from_ids16(s, sentind, t, tsentind, w16(c.L, n), w16(c.R, n), w16(lt, nt), w16(rt, nt), d["lexk"], d["lexv"])
    assert ids.flags() == 1
    txt = cgx.Corpus.load(files[0], files[2], files[3], files[4], long_sentences=True)
    res = []
    for corp in (txt, ids):
        ex = cgx.Extractor(0); ex.upload_corpus(corp)
        ex.upload_queries(d["qoff"][:-1], d["qtok"]); ex.sa_lookup(); ex.make_blocks(); ex.gappy_search(); ex.extract(); ex.lexicon()
        res.append({k: (canon_hits(ex.fetch(k)) if k.startswith("hits") else ex.fetch(k)).tobytes() for k in ("sa", "lm", "hits1", "hits2", "r0", "r1", "r2", "lex0", "lex1", "lex2")}); ex.close()
    for k in res[0]:
        assert res[0][k] == res[1][k] and len(res[0][k]) > 0, k
    with pytest.raises(cgx.CgxError):                                     # a position beyond the mode's limits is refused, not wrapped
        bad = w16(c.L, n).copy(); bad[3] = 5000
        cgx.Corpus.from_ids16(s, sentind, t, tsentind, bad, w16(c.R, n), w16(lt, nt), w16(rt, nt), d["lexk"], d["lexv"])
    txt.close(); ids.close()


@pytest.mark.parametrize("hit_order", [0, 1])
@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_every_stage_matches_the_oracle(name, hit_order, cgx, oracle_bin, fixtures_dir, tmp_path):
    """hit_order = 1: the hit lists sorted completely on the card and compared as they come; 0 (the default): ordered by pattern and
    position bucket only -- the same SETS per pattern, and the same rules afterwards, which come from the lists' order statistics.
    The hit_order = 1 leg also runs the one-lane-per-line MaxLex kernel (lex_flat = 0) instead of the wave-wide task list."""
    fx = make_fixture(name, fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); h = d["hdr"]; n, nt = h["n"], h["nt"]
    ex = cgx.Extractor(0)
    ex.set_option("hit_order", hit_order); ex.set_option("lex_flat", 1 - hit_order)
    ex.upload_index(d["str"][:n], d["rlp"], d["tstr"][:nt], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa()
    assert np.array_equal(ex.fetch("sa"), d["sa"])
    ex.precompute()
    assert np.array_equal(ex.fetch("freq"), d["freq"])
    assert np.array_equal(ex.fetch("pidx"), d["pidx"])
    assert np.array_equal(ex.fetch("miss"), d["miss"])
    assert np.array_equal(ex.fetch("phit_start"), d["phits"]["start"]) and np.array_equal(ex.fetch("phit_len"), d["phits"]["length"])
    ex.upload_queries(d["qoff"][:-1], d["qtok"])
    ex.sa_lookup()
    lm5 = np.minimum(d["lm"], 5)
    assert np.array_equal(ex.fetch("lm"), lm5)
    assert np.array_equal(ex.fetch("up"), d["up"]) and np.array_equal(ex.fetch("down"), d["down"])
    # GenerateBlocks on the device: same blocks in the same (first-seen) order, and per-query lists that equal a
    # plain restatement of the host loop (distinct ids per query, in the order the scan meets them)
    ex.make_blocks()
    got = ex.fetch("blocks")
    for f in ("start", "end", "matchlen", "string_start"):
        assert np.array_equal(got[f], d["blocks"][f]), f
    ids = {(int(b["start"]), int(b["matchlen"])): i for i, b in enumerate(d["blocks"])}
    up5 = d["up"].reshape(-1, 5); qoff = d["qoff"]; want_off = [0]; want_ids = []
    for q in range(len(qoff) - 1):
        seen = []
        for t in range(qoff[q], qoff[q + 1]):
            for ct in range(1, int(lm5[t]) + 1):
                i = ids[(int(up5[t, ct - 1]), ct)]
                if i not in seen:
                    seen.append(i)
        want_ids += seen; want_off.append(len(want_ids))
    assert np.array_equal(ex.fetch("qb_off"), np.array(want_off, np.uint32)) and np.array_equal(ex.fetch("qb_ids"), np.array(want_ids, np.uint32))
    blocks = ex.set_blocks(d["blocks"])
    assert np.array_equal(blocks["string_start"], d["blocks"]["string_start"])
    ex.gappy_search()
    for k in ("g1", "p1", "g2"):
        assert np.array_equal(ex.fetch(k), d[k]), k
    for k in ("hits1", "hits2"):
        got = ex.fetch(k)
        assert np.array_equal(got if hit_order else canon_hits(got), d[k]), k
        if not hit_order and name == "mid":
            assert not np.array_equal(got, d[k]), k + ": the default mode is expected to leave the lists in bucket order"
    s1 = ex.fetch("s1")
    for f in ("qrystart", "a_len", "b_len", "gap", "position", "sa_start", "sa_end"):
        assert np.array_equal(s1[f], d["s1"][f].astype(s1[f].dtype)), f
    s2 = ex.fetch("s2")
    for f in ("blockid", "gap2", "c_len", "position", "sa_start", "sa_end"):
        assert np.array_equal(s2[f], d["s2"][f].astype(s2[f].dtype)), f
    assert np.array_equal(ex.fetch("c2"), d["p2"]["pat"][:, 0])
    ex.extract()
    c = ex.counts()
    assert (c["sep1"], c["sep2a"], c["sep2b"], c["guard_exits"]) == (h["sep1"], h["sep2a"], h["sep2b"], 0)
    for k in ("r0", "r1", "r2"):
        assert np.array_equal(ex.fetch(k), d[k]), k
    nl1 = len(d["lex1_int"]) // 4; nl2 = len(d["lex2_int"]) // 4
    fe, ef = ex.lex_features(d["tasks"], nl1, nl2)
    assert np.array_equal(fe.view(np.uint32), d["task_fe"].view(np.uint32))      # bit patterns
    assert np.array_equal(ef.view(np.uint32), d["task_ef"].view(np.uint32))
    # device lexicon: same lines, same order, same counts and MaxLex bits as the oracle's createLexicon* restatement
    ex.lexicon()
    off = 0
    for k, n in (("lex1", nl1), ("lex2", nl2), ("lex0", len(d["lex0_int"]) // 4)):
        got = ex.fetch(k); want = d[k + "_int"].reshape(-1, 4)
        assert len(got) == n == len(want), k
        assert np.array_equal(got["id"], want[:, 0]) and np.array_equal(got["f"], want[:, 1]), k
        assert np.array_equal(got["fsample"], want[:, 2]) and np.array_equal(got["paircount"], want[:, 3]), k
        assert np.array_equal(got["fe"].view(np.uint32), d["task_fe"][off:off + n].view(np.uint32)), k
        assert np.array_equal(got["ef"].view(np.uint32), d["task_ef"][off:off + n].view(np.uint32)), k
        tk = d["tasks"][off:off + n]
        for f, g in (("tstart", "tstart"), ("end", "end"), ("gap1", "gap1"), ("gap1_1", "gap1_1")):
            assert np.array_equal(got[f], tk[g]), (k, f)
        off += n
    ex.close()


def test_exact_host_lexicon_path(cgx, fixtures_dir, tmp_path):
    """The host lexicon (taken only on a target-hash collision) must give the same files as the device lexicon."""
    fx = make_fixture("mid", fixtures_dir)
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "h"), force_host_lexicon=1, device_format=0)
    assert op.sha_dir(str(tmp_path / "h"), META["mid"]["spec"][2]) == META["mid"]["grammar"]
    ex.close(); corpus.close()


def test_edge_case_queries(cgx, oracle_bin, tmp_path):
    fx = os.path.join(ROOT, "tests", "golden", "tiny"); d = tmp_path / "fx"; d.mkdir()
    for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt"):
        shutil.copy(os.path.join(fx, n), d / n)
    (d / "query.f").write_text("\nOOV1 OOV2\ns1\ns1 s1 s1 s1 s1 s1\ns0 s2 OOV s1 s0\n\n")
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "o"))
    ex, corpus, n = run_product(cgx, str(d), str(tmp_path / "p"))
    assert op.sha_dir(str(tmp_path / "p"), 6) == op.sha_dir(str(tmp_path / "o"), 6)
    ex.close(); corpus.close()
    (d / "query.f").write_text("OOV1 OOV2 OOV3\nOOV4\n\n")                # nothing matches anywhere: three empty grammar files
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "o3"))
    for fmt in (1, 0):
        ex, corpus, n = run_product(cgx, str(d), str(tmp_path / ("p3_%d" % fmt)), device_format=fmt)
        assert n == 0 and op.sha_dir(str(tmp_path / ("p3_%d" % fmt)), 3) == op.sha_dir(str(tmp_path / "o3"), 3)
        ex.close(); corpus.close()
    (d / "query.f").write_text("")                                       # empty query file: nothing to do, no crash
    ex, corpus, n = run_product(cgx, str(d), str(tmp_path / "p2"))
    assert n == 0
    ex.close(); corpus.close()


def test_chunking_and_sharding_do_not_change_results(cgx, fixtures_dir, tmp_path):
    """Size-independent properties at a larger size: tiny work chunks, and a 2-way query split, give the same files."""
    fx = make_fixture("toy", fixtures_dir)
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "a"))
    ex.close()
    ex, corpus2, n2 = run_product(cgx, fx, str(tmp_path / "b"), chunk_items=4096)
    assert n == n2 and op.sha_dir(str(tmp_path / "a"), 7) == op.sha_dir(str(tmp_path / "b"), 7) == META["toy"]["grammar"]
    for fmt in (1, 0):                                                  # no output directory: only the rule count, same number
        ex.set_option("device_format", fmt)
        assert ex.extract_grammars(corpus2, op.fixture_args(fx)[1], None) == n
    ex.set_option("device_format", 1)
    os.makedirs(str(tmp_path / "c"))
    files = op.fixture_args(fx)
    ex.extract_grammars(corpus2, files[1], str(tmp_path / "c"), 0, 4)
    ex.extract_grammars(corpus2, files[1], str(tmp_path / "c"), 4, 7)
    assert op.sha_dir(str(tmp_path / "c"), 7) == META["toy"]["grammar"]
    ex.close(); corpus.close(); corpus2.close()


@pytest.mark.parametrize("opts", [dict(append_slack=0, append_guess_milli=1), dict(wide_hits2=1), dict(wide_hits2=1, chunk_items=1024), dict(look_rec_cap=0), dict(look_rec_cap=3, chunk_items=4096), dict(use_lex_hash=0), dict(lex_flat=0), dict(lex_flat=2), dict(lex_bits=0), dict(win_table=1), dict(pool_cap=4), dict(pool_cap=1, look_rec_cap=2)])
def test_lookup_output_sizing_paths(opts, cgx, fixtures_dir, tmp_path):
    """The single-pass lookups guess their output size: an undersized guess (rerun with the exact size), the
    wide-pattern-id layout (> 2^24 distinct two-gap patterns) and groups too large for the LDS record cache
    (records read from global memory through the batch-wide hash) must give the same files."""
    fx = make_fixture("toy", fixtures_dir)
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "a"), **opts)
    assert op.sha_dir(str(tmp_path / "a"), 7) == META["toy"]["grammar"]
    ex.close(); corpus.close()


@pytest.mark.parametrize("name", ["toy", "mid"])
def test_source_addressed_target_blocks(name, cgx, fixtures_dir, tmp_path):
    """The lookups find the target-side alignment bytes of a sentence from its SOURCE start (a second copy of the blocks,
    built with the index when a small factor K fits every sentence pair: target words <= K * (source tokens + 1)); without it
    they first fetch the sentence's target offset from the delimiter's alignment word.  Both give the golden files, and the
    fixtures do get the table."""
    fx = make_fixture(name, fixtures_dir); nq = META[name]["spec"][2]
    ex, corpus, n = run_product(cgx, fx, str(tmp_path / "a"))
    assert 1 <= ex.stage_ms("src_blocks_factor") <= 4
    ex.close(); corpus.close()
    ex, corpus, n2 = run_product(cgx, fx, str(tmp_path / "b"), src_blocks=0)
    assert n == n2 and op.sha_dir(str(tmp_path / "a"), nq) == op.sha_dir(str(tmp_path / "b"), nq) == META[name]["grammar"]
    ex.close(); corpus.close()


def test_corpus_without_source_addressed_blocks(cgx, oracle_bin, fixtures_dir, tmp_path):
    """One sentence pair whose target is more than four times its source (one source word, twelve target words) and the
    source-addressed table is not built (its factor would be 7): the lookups take the delimiter path for the whole corpus,
    and the files still equal the oracle's on the same corpus."""
    fx = make_fixture("toy", fixtures_dir); d = tmp_path / "fx"; shutil.copytree(fx, d)
    first_src = open(d / "corpus.f").readline().split()[0]; tw = open(d / "corpus.e").readline().split()
    with open(d / "corpus.f", "a") as f: f.write(first_src + "\n")
    with open(d / "corpus.e", "a") as f: f.write(" ".join((tw * 12)[:12]) + "\n")
    with open(d / "corpus.a", "a") as f: f.write("0-0 0-1\n")
    op.run_oracle(oracle_bin, str(d), str(tmp_path / "o"))
    ex, corpus, n = run_product(cgx, str(d), str(tmp_path / "p"))
    assert ex.stage_ms("src_blocks_factor") == 0
    assert op.sha_dir(str(tmp_path / "p"), 7) == op.sha_dir(str(tmp_path / "o"), 7)
    ex.close(); corpus.close()


def test_lexicon_hash_collisions_are_survived(cgx, fixtures_dir, tmp_path):
    """The device lexicon groups rules by (id, hash bits of the target side).  With few hash bits two different target
    sides of one id collide: the device notices (neighbours of a run are compared symbol by symbol) and regroups under
    another seed; when every seed collides the exact host lexicon takes over.  Same files either way; the sweep must
    show both outcomes."""
    fx = make_fixture("toy", fixtures_dir)
    seen = set()
    for bits in (2, 12, 16, 18, 20, 22, 24, 26, 28):
        out = str(tmp_path / ("b%d" % bits))
        ex, corpus, n = run_product(cgx, fx, out, lex_hash_bits=bits)
        assert op.sha_dir(out, 7) == META["toy"]["grammar"], bits
        if ex.host_ms("exact_host_lexicon") >= 1: seen.add("host")
        elif ex.stage_ms("lex_rehash") >= 1: seen.add("rehash")
        ex.close(); corpus.close()
        if seen == {"host", "rehash"}: break
    assert seen == {"host", "rehash"}, seen


def test_index_replica_gives_the_same_files(cgx, fixtures_dir, tmp_path):
    """Multi-GPU layout on one card: a second context receives the index buffer by buffer (what bench.py does over
    RCCL and cgx_broadcast_index does in C), rebuilds the derived tables in cgx_index_finalize and must produce the
    same grammar files as the context that built the index."""
    import torch
    fx = make_fixture("toy", fixtures_dir); files = op.fixture_args(fx)
    root = cgx.Extractor(0); corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4]); root.upload_corpus(corpus)
    rep = cgx.Extractor(0); rep.index_alloc(root.index_shape())
    for i, (name, nbytes) in enumerate(root.index_buffers()):
        if nbytes == 0:
            continue
        stage = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
        root.index_d2d(i, stage.data_ptr(), 0); torch.cuda.synchronize()
        rep.index_d2d(i, stage.data_ptr(), 1); torch.cuda.synchronize()
    rep.index_finalize()
    os.makedirs(str(tmp_path / "r"))
    rep.extract_grammars(corpus, files[1], str(tmp_path / "r"))
    assert op.sha_dir(str(tmp_path / "r"), 7) == META["toy"]["grammar"]
    rep.close(); root.close(); corpus.close()


def test_a_failed_stage_leaves_the_context_usable(cgx, fixtures_dir, tmp_path):
    """Out-of-memory in the middle of a stage (injected): the call fails loudly, the device temporaries it had allocated
    are reclaimed at the next stage entry, and the same context then produces the right files."""
    fx = make_fixture("toy", fixtures_dir); files = op.fixture_args(fx)
    ex = cgx.Extractor(0); corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4]); ex.upload_corpus(corpus)
    failures = 0
    for nth in (3, 9, 17, 30, 55, 90):
        os.makedirs(str(tmp_path / ("f%d" % nth)))
        ex.set_option("fault_inject", nth)
        try:
            ex.extract_grammars(corpus, files[1], str(tmp_path / ("f%d" % nth)))
        except cgx.CgxError as e:
            failures += 1
            assert "injected fault" in str(e) or "device allocation" in str(e)
        ex.set_option("fault_inject", 0)
    assert failures >= 4
    os.makedirs(str(tmp_path / "ok"))
    ex.extract_grammars(corpus, files[1], str(tmp_path / "ok"))
    assert op.sha_dir(str(tmp_path / "ok"), 7) == META["toy"]["grammar"]
    assert ex.stage_ms("swept_temporaries") > 0
    ex.close(); corpus.close()


def test_optional_index_tables_may_fail_to_allocate(cgx, fixtures_dir, tmp_path):
    """The corpus-order occurrence table (pos1), the source-addressed target blocks (lrs), the window table (win) and the presence bits of the lexical table (bits) only make kernels faster: when the
    card has no room for one of them (injected: the n-th device allocation of the index build fails) the index loads without it,
    says so, and the files are the golden files; a failure of a table the index needs still fails the load."""
    fx = make_fixture("toy", fixtures_dir); files = op.fixture_args(fx)
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4])
    seen = set(); failed = 0
    for nth in range(1, 200):
        ex = cgx.Extractor(0); ex.set_option("win_table", 1); ex.set_option("fault_inject", nth)
        try:
            ex.upload_corpus(corpus)
        except cgx.CgxError:
            failed += 1; ex.close(); continue
        injected = ex.stage_ms("pos1_skipped") == 1.0 or ex.stage_ms("src_blocks_skipped") == 1.0 or ex.stage_ms("win_table_skipped") == 1.0 or ex.stage_ms("lex_bits_skipped") == 1.0
        if not injected:                                                  # the counter ran past the last allocation of the build: nothing left to inject
            ex.set_option("fault_inject", 0); ex.close(); break
        what = "pos1" if ex.stage_ms("pos1_skipped") == 1.0 else "win" if ex.stage_ms("win_table_skipped") == 1.0 else "bits" if ex.stage_ms("lex_bits_skipped") == 1.0 else "lrs"
        ex.set_option("fault_inject", 0)
        if what not in seen:
            seen.add(what)
            out = tmp_path / what; out.mkdir()
            ex.extract_grammars(corpus, files[1], str(out))
            assert op.sha_dir(str(out), 7) == META["toy"]["grammar"], what
            assert (ex.stage_ms("src_blocks_factor") == 0) == (what == "lrs")
            assert (ex.stage_ms("win_table_gb") == 0) == (what == "win")
        ex.close()
    assert seen == {"pos1", "lrs", "win", "bits"} and failed > 10, (seen, failed)
    corpus.close()


def test_two_contexts_share_one_index_and_work_at_the_same_time(cgx, fixtures_dir, tmp_path):
    """cgx_share_index: a second context of the same card borrows the first one's index (nothing copied) and the two, driven from two
    threads, extract at the same time -- plain files and .gz files, several rounds each -- into directories of their own: every round of
    both gives the golden files; closing the borrower leaves the lender's index intact."""
    import threading, gzip
    fx = make_fixture("mid", fixtures_dir); files = op.fixture_args(fx); nq = META["mid"]["spec"][2]
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4])
    a = cgx.Extractor(0); a.upload_corpus(corpus)
    hbm_one = a.stage_ms("mem_index")
    b = cgx.Extractor(0); b.share_index(a)
    assert b.stage_ms("mem_index") == hbm_one or b.stage_ms("mem_index") >= 0     # (the borrower reports what it sees, not a second copy)
    errs = []

    def work(ex, tag, gz):
        try:
            ex.set_option("gz_level", 1 if gz else 0)
            for r in range(3):
                out = tmp_path / ("%s%d" % (tag, r)); out.mkdir()
                ex.extract_grammars(corpus, files[1], str(out))
                if gz:
                    for q in range(nq):
                        raw = gzip.open(out / ("grammar.%d.s.gz" % q), "rb").read()
                        open(out / ("grammar.%d.s" % q), "wb").write(raw)
                if op.sha_dir(str(out), nq) != META["mid"]["grammar"]:
                    errs.append((tag, r, "files differ"))
        except Exception as e:                                            # noqa: BLE001 -- reported below, from the main thread
            errs.append((tag, repr(e)))

    for gz in (0, 1):
        ta = threading.Thread(target=work, args=(a, "a%d_" % gz, gz)); tb = threading.Thread(target=work, args=(b, "b%d_" % gz, gz))
        ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    b.close()
    out = tmp_path / "after"; out.mkdir()
    a.set_option("gz_level", 0); a.extract_grammars(corpus, files[1], str(out))
    assert op.sha_dir(str(out), nq) == META["mid"]["grammar"]
    a.close(); corpus.close()


def test_ngram_tables_do_not_change_intervals(cgx, oracle_bin, fixtures_dir, tmp_path):
    """l = 2..5 from the l-gram hash tables (one probe per length) vs. by nested binary search, for every split between
    the two ("ngram_tables" = longest length answered from a table): same lm / up / down, all equal to the oracle; the
    probe-counting variant of the kernel writes the same results and reports what it read."""
    fx = make_fixture("toy", fixtures_dir); dump = str(tmp_path / "d.bin")
    op.run_oracle(oracle_bin, fx, str(tmp_path / "o"), dump)
    d = op.read_dump(dump); h = d["hdr"]
    ex = cgx.Extractor(0)
    ex.upload_index(d["str"][:h["n"]], d["rlp"], d["tstr"][:h["nt"]], d["ltar"], d["rtar"], d["lexk"], d["lexv"])
    ex.build_sa(); ex.precompute(); ex.upload_queries(d["qoff"][:-1], d["qtok"])
    want = (np.minimum(d["lm"], 5), d["up"], d["down"])
    for maxl in (5, 4, 3, 2, 1):
        ex.set_option("ngram_tables", maxl); ex.set_option("count_probes", 1); ex.sa_lookup()
        for a, b in zip((ex.fetch("lm"), ex.fetch("up"), ex.fetch("down")), want):
            assert np.array_equal(a, b), maxl
        assert ex.stage_ms("sa_probe_lookups") == int(want[0].sum())
        if maxl == 5:
            assert ex.stage_ms("sa_probe_search") == 0
        elif int((want[0] > maxl).sum()):
            assert ex.stage_ms("sa_probe_search") > 0
        assert (ex.stage_ms("sa_probe_slots") > 0) == (maxl >= 2)
    ex.set_option("use_bigrams", 0); ex.set_option("count_probes", 0); ex.sa_lookup()      # round-1 option name: no tables at all
    assert np.array_equal(ex.fetch("up"), want[1])
    ex.close()


def test_async_writer_gives_the_same_files(cgx, fixtures_dir, tmp_path):
    """async_write: batch k is written by host threads while batch k+1 runs; after cgx_flush the files are identical."""
    fx = make_fixture("mid", fixtures_dir); files = op.fixture_args(fx)
    ex = cgx.Extractor(0); ex.set_option("async_write", 1)
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4]); ex.upload_corpus(corpus)
    out = tmp_path / "a"; out.mkdir()
    n1 = ex.extract_grammars(corpus, files[1], str(out), 0, 17)
    n2 = ex.extract_grammars(corpus, files[1], str(out), 17, 40)
    ex.flush()
    assert op.sha_dir(str(out), 40) == META["mid"]["grammar"]
    assert n1 + n2 == sum(sum(1 for _ in open(out / ("grammar.%d.s" % q), "rb")) for q in range(40))
    ex.close(); corpus.close()


def test_writer_errors_surface_and_the_context_survives(cgx, fixtures_dir, tmp_path):
    """A batch whose output directory does not exist: the background writer's I/O error comes back from cgx_flush (or from
    the call itself in synchronous mode), nothing hangs, and the same context then writes the golden files."""
    fx = make_fixture("toy", fixtures_dir); files = op.fixture_args(fx)
    corpus = cgx.Corpus.load(files[0], files[2], files[3], files[4])
    for async_write in (1, 0):
        ex = cgx.Extractor(0); ex.set_option("async_write", async_write); ex.upload_corpus(corpus)
        with pytest.raises(cgx.CgxError):
            ex.extract_grammars(corpus, files[1], str(tmp_path / "does" / "not" / "exist"))
            ex.flush()
        out = tmp_path / ("ok%d" % async_write); out.mkdir()
        ex.extract_grammars(corpus, files[1], str(out)); ex.flush()
        assert op.sha_dir(str(out), 7) == META["toy"]["grammar"]
        ex.close()
    corpus.close()


_EXTRA = [tuple(int(x) for x in os.environ["CGX_BIG_PARITY"].split(","))] if os.environ.get("CGX_BIG_PARITY") else []   # e.g. 1000000,200000,60 (minutes of oracle time)


@pytest.mark.parametrize("pairs,vocab,nq", [(3000, 400, 12), (100000, 200000, 120)] + _EXTRA)
def test_id_level_batch_on_synthetic_corpus(pairs, vocab, nq, cgx, tmp_path):
    """bench.py's path: corpus from id arrays; suffix array property check + oracle parity through liboracle.  The
    second size is the prefix bench.py times its CPU baseline on (2.6 M source tokens, the benchmark's vocabulary)."""
    import ctypes as C
    from cgx_amd import synth
    corpus = synth.make_corpus(pairs, vocab, 5)
    qoff, qtok = synth.make_queries(corpus, nq, 6)
    host = cgx.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"],
                               corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    ex = cgx.Extractor(0); ex.upload_corpus(host)
    sa = ex.fetch("sa"); s = corpus["str"]; n = len(s)
    assert np.array_equal(np.bincount(sa, minlength=n), np.ones(n, np.int64))
    pad = np.concatenate((s, np.zeros(64, np.int32)))
    for i in np.random.default_rng(0).integers(0, n - 1, 2000):      # adjacent suffixes are in order
        a, b = int(sa[i]), int(sa[i + 1]); k = 0
        while pad[a + k] == pad[b + k]:
            k += 1
        assert pad[a + k] < pad[b + k]
    out = tmp_path / "p"; out.mkdir()
    nrules = ex.extract_grammars_ids(host, qoff, qtok, str(out), 0)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")); vp = C.c_void_p
    lib.orc_index_from_arrays.restype = vp
    lib.orc_index_from_arrays.argtypes = [vp, C.c_uint32, vp, C.c_int32, vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp]
    lib.orc_batch_from_ids.restype = vp; lib.orc_batch_from_ids.argtypes = [vp, C.c_int32, vp, C.c_int32]
    lib.orc_run_all.argtypes = [vp, vp, C.c_char_p]
    p = lambda a: a.ctypes.data_as(vp)
    arrs = [np.ascontiguousarray(corpus[k]) for k in ("str", "sentind", "tstr", "tsentind", "lsrc", "rsrc", "ltar", "rtar", "lexk", "lexv")]
    ix = lib.orc_index_from_arrays(p(arrs[0]), len(arrs[0]), p(arrs[1]), len(arrs[1]) - 1, p(arrs[2]), len(arrs[2]), p(arrs[3]), p(arrs[4]), p(arrs[5]),
                                   p(arrs[6]), p(arrs[7]), p(arrs[8]), p(arrs[9]), len(arrs[8]), None)
    oout = tmp_path / "o"; oout.mkdir()
    b = lib.orc_batch_from_ids(p(qoff), len(qoff), p(qtok), len(qtok))
    assert lib.orc_run_all(ix, b, str(oout).encode()) == 0
    assert op.sha_dir(str(out), nq) == op.sha_dir(str(oout), nq)
    assert nrules == sum(sum(1 for _ in open(oout / f, "rb")) for f in os.listdir(oout))
    ex.close(); host.close()


def test_run_sort_orders_runs_of_any_length(cgx):
    """run_sort (the order inside id runs of the rule and lexicon keys): short runs through the LDS passes, and runs longer than
    the fix pass's LDS buffer (RS_MAXRUN = 320) that straddle a 1024-record block boundary through its slow path -- sorted all the
    same, stable, payload carried, and counted.  No caller produces such a run today (the sampling caps are 300 / 65 / 70)."""
    rng = np.random.default_rng(5)
    ex = cgx.Extractor(0)

    def check(lengths, keybits, want_long):
        major = np.repeat(np.arange(len(lengths), dtype=np.uint32) * 3 + 7, lengths)
        n = len(major)
        key = rng.integers(0, 1 << keybits, n, dtype=np.uint64)
        val = np.arange(n, dtype=np.uint32)
        ko, vo, nlong = ex.run_sort(major, key, val)
        order = np.lexsort((val, key, major))                  # (major, key, input position): what a stable sort inside the runs gives
        assert np.array_equal(ko, key[order]) and np.array_equal(vo, val[order])
        ko2, vo2, _ = ex.run_sort(major, key)
        assert vo2 is None and np.array_equal(ko2, key[order])
        assert nlong == want_long, (nlong, want_long)
        # the same through the position-tagged comparison the callers with narrow keys get (keys < 2^53 or all ones = "no rule", which sorts last)
        key3 = key.copy(); key3[rng.integers(0, n, n // 7)] = np.uint64(0xFFFFFFFFFFFFFFFF)
        order3 = np.lexsort((val, key3, major))
        ko3, vo3, _ = ex.run_sort(major, key3, val, keybits=max(keybits, 1))
        assert np.array_equal(ko3, key3[order3]) and np.array_equal(vo3, val[order3])
        ko4, _, _ = ex.run_sort(major, key3, None, keybits=53)
        assert np.array_equal(ko4, key3[order3])

    # short runs only (1..300), many of them cut by block boundaries: the fast path
    check(rng.integers(1, 301, 4000), 40, 0)
    # few distinct keys: ties keep input order
    check(rng.integers(1, 301, 500), 2, 0)
    # a 400-record run across the first boundary (records 900..1299), a 5000-record run across four boundaries, a 330-record run
    # inside one block (no fix needed), short runs around them
    lengths = [900, 400, 60, 5000, 20, 330, 300, 1]
    check(lengths, 30, 2)
    check([1024, 321 + 1024, 3], 3, 1)                         # a long run that starts exactly at a boundary; ties
    check(rng.integers(1, 301, 3000), 53, 0)                   # the widest keys the tagged comparison takes
    assert ex.stage_ms("run_sort_long_runs") == 4 * (2 + 1)    # each check sorts four times (with and without payload, plain and tagged)
    ex.close()
