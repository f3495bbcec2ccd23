"""Test helpers around the CPU oracle (oracle/): run its CLI, read its dump.  TESTS ONLY."""
import hashlib
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GAPPY = np.dtype([("qrystart", "<i4"), ("a_len", "u1"), ("b_len", "u1"), ("gap", "u1")])
GAPPAT = np.dtype([("pat", "<i4", (5,)), ("number", "u1")])
TWOGAPPY = np.dtype([("blockid", "<u4"), ("gap2", "<u4"), ("c_len", "u1")])
TWOGAPPAT = np.dtype([("pat", "<i4", (1,)), ("number", "u1"), ("blockid", "<u4")])
GAPSEARCH = np.dtype([("qrystart", "<i4"), ("a_len", "u1"), ("b_len", "u1"), ("gap", "u1"), ("position", "<u4"), ("sa_start", "<i4"), ("sa_end", "<i4")])
TWOGAPSEARCH = np.dtype([("blockid", "<u4"), ("gap2", "<u4"), ("c_len", "u1"), ("position", "<u4"), ("sa_start", "<i4"), ("sa_end", "<i4")])
HIT1 = np.dtype([("position", "<u4"), ("str_position", "<u4"), ("length", "u1")])
HIT2 = np.dtype([("position", "<u4"), ("str_position", "<u4"), ("length", "u1"), ("length2", "u1")])
RULE1 = np.dtype([("id", "<i4"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1")])
RULE2 = np.dtype([("id", "<i4"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1"), ("gap2", "u1"), ("gap2_1", "u1")])
RULE0 = np.dtype([("tar_start", "<i4"), ("block", "<i4"), ("tar_end", "u1")])
PREHIT = np.dtype([("start", "<u4"), ("length", "u1")])
LEXTASK = np.dtype([("lexid", "<u4"), ("src", "<i4", (5,)), ("nsrc", "u1"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1"), ("gap2", "u1"), ("gap2_1", "u1")])
BLOCK = np.dtype([("start", "<i4"), ("end", "<i4"), ("matchlen", "<i4"), ("string_start", "<i4")])
LEXKEY = np.dtype([("src", "<i4"), ("tgt", "<i4")])
LEXVAL = np.dtype([("v1", "<f4"), ("v2", "<f4")])

DTYPES = {
    "header": np.uint32, "str": np.int32, "sa": np.int32, "rlp": np.uint32, "tstr": np.int32, "ltar": np.uint8, "rtar": np.uint8,
    "sentind": np.int32, "tsentind": np.int32, "lexk": LEXKEY, "lexv": LEXVAL, "freq": np.int32, "pidx": np.uint32, "miss": np.int32,
    "phits": PREHIT, "qoff": np.int32, "qtok": np.int32, "lm": np.int32, "up": np.int32, "down": np.int32, "g1": GAPPY, "p1": GAPPAT,
    "s1": GAPSEARCH, "hits1": HIT1, "g2": TWOGAPPY, "p2": TWOGAPPAT, "s2": TWOGAPSEARCH, "hits2": HIT2, "qone": np.uint32, "qtwo": np.uint32,
    "qblocks": np.uint32, "blocks": BLOCK, "r0": RULE0, "r1": RULE1, "r2": RULE2, "rng0": np.int32, "rng1": np.int32, "rng2": np.int32,
    "tasks": LEXTASK, "task_fe": np.float32, "task_ef": np.float32,
    "lex0_int": np.int32, "lex1_int": np.int32, "lex2_int": np.int32, "lex0_flt": np.float32, "lex1_flt": np.float32, "lex2_flt": np.float32,
}
HEADER = ["n", "nt", "nsent", "nsvocab", "ntvocab", "nlex", "nphits", "nq", "ntok", "g", "d1", "d2", "sep1", "sep2a", "sep2b"]


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    p = 0
    while p + 16 <= len(data):
        tag = data[p:p + 8].rstrip(b"\0").decode()
        (nb,) = struct.unpack_from("<Q", data, p + 8)
        raw = data[p + 16:p + 16 + nb]
        p += 16 + nb
        out[tag] = np.frombuffer(raw, dtype=DTYPES[tag]).copy() if tag in DTYPES else raw
    out["hdr"] = dict(zip(HEADER, (int(x) for x in out["header"])))
    return out


def fixture_args(fx):
    return [os.path.join(fx, n) for n in ("corpus.f", "query.f", "corpus.e", "corpus.a", "lex.txt")]


def run_oracle(oracle_bin, fx, outdir, dump=None):
    os.makedirs(outdir, exist_ok=True)
    cmd = [oracle_bin] + fixture_args(fx) + [outdir]
    if dump:
        cmd += ["--dump", dump]
    return subprocess.run(cmd, check=True, capture_output=True, text=True)


def sha_dir(outdir, nq):
    h = []
    for q in range(nq):
        with open(os.path.join(outdir, "grammar.%d.s" % q), "rb") as f:
            h.append(hashlib.sha256(f.read()).hexdigest())
    return h
