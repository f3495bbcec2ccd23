"""Two contexts over one index (cgx_share_index) against one: the GPU chain alone (no text, no files), K batches of the cfg3 workload.
Usage on a GPU box: python tools/two_contexts.py [--pairs N] [--queries N] [--batches K] [--files DIR --gz]"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cgx_amd
from cgx_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=10_000_000); ap.add_argument("--vocab", type=int, default=200_000)
ap.add_argument("--queries", type=int, default=10_000); ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--own-index", action="store_true", help="the second context builds an index of its own instead of borrowing the first one's (is it the SHARING that limits the overlap?)")
a = ap.parse_args()
corpus = synth.make_corpus(a.pairs, a.vocab, 1234, 5, 45)
host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"], corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
qs = [synth.make_queries(corpus, a.queries, 4321 + 1000 * k) for k in range(3)]
ex = cgx_amd.Extractor(0); ex.upload_corpus(host)
ex2 = cgx_amd.Extractor(0)
if a.own_index: ex2.upload_corpus(host)
else: ex2.share_index(ex)

def run(e, ks):
    for k in ks:
        qo, qt = qs[k % 3]
        e.extract_grammars_ids(host, np.asarray(qo, np.int32), qt, None, 0)

for e in (ex, ex2): run(e, [0, 1])                          # warm both: buffers sized, capacity guesses settled
t0 = time.perf_counter(); run(ex, range(a.batches)); t1 = time.perf_counter() - t0
ths = [threading.Thread(target=run, args=(ex, range(0, a.batches, 2))), threading.Thread(target=run, args=(ex2, range(1, a.batches, 2)))]
t0 = time.perf_counter()
for th in ths: th.start()
for th in ths: th.join()
t2 = time.perf_counter() - t0
print(("own indexes; " if a.own_index else "one index; ") + "one context: %.1f ms per batch; two contexts, two threads: %.1f ms per batch; ratio %.3f" % (t1 / a.batches * 1e3, t2 / a.batches * 1e3, t1 / t2))
