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
