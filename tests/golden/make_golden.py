#!/usr/bin/env python3
"""Regenerates tests/golden/: the tiny fixture's inputs and the expected grammar files.

The reference ships no golden vectors (no tests, toy data absent), and its CUDA kernels
cannot be built in this image, so the expected outputs come from the CPU oracle
(oracle/strmatch_oracle).  Where /root/reference exists the script ALSO pushes the oracle's
intermediates through the real reference objects (oracle/_ref/ref_harness: SuffixArray.c,
ExtractPair.c, PrintResults.c compiled in place) and refuses to write goldens unless the
reference's suffix array and grammar files are byte-identical to the oracle's.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fixture  # noqa: E402

SPECS = {  # name: (pairs, vocab, queries, seed, long_query, commit_inputs)
    "tiny": (400, 160, 7, 7, True, True),
    "toy": (20000, 160, 7, 11, True, False),
    "mid": (5000, 300, 40, 3, True, False),
}


def sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    meta = {}
    for name, (pairs, vocab, queries, seed, longq, commit) in SPECS.items():
        fx = os.path.join(HERE, name) if commit else tempfile.mkdtemp()
        gen_fixture.write_fixture(fx, pairs, vocab, queries, seed, long_query=longq)
        files = [os.path.join(fx, n) for n in ("corpus.f", "query.f", "corpus.e", "corpus.a", "lex.txt")]
        out = tempfile.mkdtemp(); dump = os.path.join(out, "dump.bin")
        subprocess.run([os.path.join(ROOT, "oracle", "strmatch_oracle")] + files + [out, "--dump", dump], check=True)
        pinned = False
        if os.path.exists(ref):
            subprocess.run([ref, "sa", dump], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            rout = tempfile.mkdtemp()
            subprocess.run([ref, "grammar", dump, rout], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for q in range(queries):
                assert sha(os.path.join(out, "grammar.%d.s" % q)) == sha(os.path.join(rout, "grammar.%d.s" % q)), (name, q)
            pinned = True
        meta[name] = {"spec": [pairs, vocab, queries, seed, longq], "inputs": {os.path.basename(p): sha(p) for p in files},
                      "grammar": [sha(os.path.join(out, "grammar.%d.s" % q)) for q in range(queries)],
                      "host_stages_checked_against_reference_objects": pinned}
        if commit:
            with tarfile.open(os.path.join(HERE, name + "_expected.tar.gz"), "w:gz") as tar:
                for q in range(queries):
                    tar.add(os.path.join(out, "grammar.%d.s" % q), arcname="grammar.%d.s" % q)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps({k: v["host_stages_checked_against_reference_objects"] for k, v in meta.items()}))


if __name__ == "__main__":
    main()
