"""CPU tests of the drop-in boundary: the library loads and exports exactly what include/cgx.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import cgx_amd
    if not os.path.exists(cgx_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return cgx_amd.load_library()


def declared():
    text = open(os.path.join(ROOT, "include", "cgx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cgx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    import cgx_amd
    names = declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(cgx_amd.ABI) == names


def test_signatures_are_plain_c():
    text = open(os.path.join(ROOT, "include", "cgx.h")).read()
    assert "torch" not in text and "std::" not in text and "hipStream" not in text


def test_no_gpu_means_failure_not_fallback(lib):
    """Without a HIP device cgx_create must fail; the product has no CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.cgx_create(0) is None
    import cgx_amd
    with pytest.raises(cgx_amd.CgxError):
        cgx_amd.Extractor(0)


def test_product_does_not_reference_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "cgx_amd")):
        for f in fs:
            if f.endswith((".c", ".h", ".hip", ".py")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(dp, f), errors="replace").read().lower().replace("oracle/_ref", ""), f
    out = subprocess.run(["ldd", os.path.join(ROOT, "cgx_amd", "libcgx_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_cli_argument_behaviour():
    """Main.c:47-54: wrong positional count -> help text on stdout, exit 0; bad -t -> exit 0."""
    exe = os.path.join(ROOT, "bin", "strmatchcuda")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    r = subprocess.run([exe, "a", "b"], capture_output=True, text=True)
    assert r.returncode == 0 and "Please check your input arguments" in r.stdout
    r = subprocess.run([exe, "-t", "11", "a", "b", "c", "d", "e", "f"], capture_output=True, text=True)
    assert r.returncode == 0 and "finger length must be between 1 and 10" in r.stderr


def _copy_tiny(tmp_path):
    import shutil
    fx = os.path.join(ROOT, "tests", "golden", "tiny"); d = tmp_path / "fx"; d.mkdir()
    for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt", "query.f"):
        shutil.copy(os.path.join(fx, n), d / n)
    return d


def test_loader_error_behaviour_matches_the_reference(lib, tmp_path):
    """Host loaders run without a GPU: alignment index >= 255 -> "too long sentence" exit 1 (ExtractPair.cu:2683),
    dangling "i-" -> "Not possible!" exit 0 (:2676), missing lexical table -> message, exit 0 (:2458-2461)."""
    import cgx_amd
    exe = os.path.join(ROOT, "bin", "strmatchcuda")
    d = _copy_tiny(tmp_path); out = tmp_path / "out"; out.mkdir()
    args = lambda: [str(d / n) for n in ("corpus.f", "query.f", "corpus.e", "corpus.a", "lex.txt")] + [str(out)]
    lines = (d / "corpus.a").read_text().splitlines()
    (d / "corpus.a").write_text("\n".join(["0-255"] + lines[1:]) + "\n")
    with pytest.raises(cgx_amd.CgxError, match="too long sentence"):
        cgx_amd.Corpus.load(*[str(d / n) for n in ("corpus.f", "corpus.e", "corpus.a", "lex.txt")])
    r = subprocess.run([exe] + args(), capture_output=True, text=True)
    assert r.returncode == 1 and "Not possible, too long sentence" in r.stdout
    (d / "corpus.a").write_text("\n".join(["0-0 1"] + lines[1:]) + "\n")
    r = subprocess.run([exe] + args(), capture_output=True, text=True)
    assert r.returncode == 0 and "Not possible!" in r.stdout
    (d / "corpus.a").write_text("\n".join(lines) + "\n")
    os.remove(d / "lex.txt")
    r = subprocess.run([exe] + args(), capture_output=True, text=True)
    assert r.returncode == 0 and "The Word Possibility File is not Found!" in r.stderr


def test_loader_tokenisation_quirks(lib, tmp_path):
    """Start.cu:273-305: blanks are the only separators, a word starting with other white space ends the line,
    an empty line is an empty sentence; source/target line counts must agree."""
    import cgx_amd
    d = tmp_path
    (d / "s.f").write_text("a b  c\n\nd\te f\n" + "\n".join("w%d" % i for i in range(120)) + "\n")
    (d / "s.e").write_text("x y\n\nz\n" + "\n".join("v%d" % i for i in range(120)) + "\n")
    (d / "s.a").write_text("0-0\n\n0-0\n" + "\n".join("0-0" for _ in range(120)) + "\n")
    (d / "lex").write_text("NULL NULL 0.1 0.1\na x 0.5 0.25\n")
    c = cgx_amd.Corpus.load(str(d / "s.f"), str(d / "s.e"), str(d / "s.a"), str(d / "lex"))
    c.close()
    (d / "s.e").write_text("x y\n")
    with pytest.raises(cgx_amd.CgxError, match="lines"):
        cgx_amd.Corpus.load(str(d / "s.f"), str(d / "s.e"), str(d / "s.a"), str(d / "lex"))


def test_file_phase_assembles_files_from_text_and_pieces(tmp_path):
    """cgx_assemble_files (no GPU, no context): the writer's file phase on a made-up unique text -- every file is the
    concatenation of its pieces, an older longer file is cut to length, an empty query gives an empty file."""
    import ctypes as C
    import numpy as np
    import cgx_amd
    lib = cgx_amd.load_library()
    lib.cgx_assemble_files.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int, C.POINTER(C.c_double)]
    rng = np.random.default_rng(1)
    text = rng.integers(32, 127, 200000, dtype=np.uint8)
    nq = 37; counts = rng.integers(0, 3000, nq); counts[5] = 0; counts[6] = 2500          # more pieces than one pwritev takes (IOV_MAX = 1024)
    qseg = np.concatenate(([0], np.cumsum(counts))).astype(np.uint64); ns = int(qseg[-1])
    seg_len = rng.integers(1, 90, ns).astype(np.uint32); seg_off = rng.integers(0, len(text) - 100, ns).astype(np.uint64)
    out = tmp_path / "g"; out.mkdir()
    (out / "grammar.103.s").write_bytes(b"x" * 10_000_000)                                # stale, longer than what replaces it
    for threads in (1, 5):
        ms = C.c_double()
        assert lib.cgx_assemble_files(text.ctypes.data, qseg.ctypes.data, seg_off.ctypes.data, seg_len.ctypes.data, nq, 100, str(out).encode(), threads, C.byref(ms)) == 0
        for q in range(nq):
            want = b"".join(text[int(seg_off[s]):int(seg_off[s]) + int(seg_len[s])].tobytes() for s in range(int(qseg[q]), int(qseg[q + 1])))
            assert (out / ("grammar.%d.s" % (100 + q))).read_bytes() == want, q
    assert lib.cgx_assemble_files(text.ctypes.data, qseg.ctypes.data, seg_off.ctypes.data, seg_len.ctypes.data, nq, 0, str(tmp_path / "missing").encode(), 2, None) != 0
