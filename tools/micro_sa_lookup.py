#!/usr/bin/env python3
"""k_sa_lookup alone: the same batch of queries looked up N times back to back on an otherwise idle GPU (event-timed kernel
durations), for several corpus sizes.  Usage: python3 tools/micro_sa_lookup.py [pairs ...]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda:0")
import cgx_amd
from cgx_amd import synth

for pairs in [int(x) for x in (sys.argv[1:] or ["1000000", "10000000"])]:
    corpus = synth.make_corpus(pairs, 200000, 1234)
    host = cgx_amd.Corpus.from_ids(corpus["str"], corpus["sentind"], corpus["tstr"], corpus["tsentind"], corpus["lsrc"], corpus["rsrc"], corpus["ltar"], corpus["rtar"], corpus["lexk"], corpus["lexv"])
    ex = cgx_amd.Extractor(0); ex.upload_corpus(host)
    for nq in (10000, 40000):
        qoff, qtok = synth.make_queries(corpus, nq, 4321)
        ex.upload_queries(qoff, qtok)
        out = {}
        for maxl in (5, 3, 2):
            ex.set_option("ngram_tables", maxl)
            ms = []
            for _ in range(30):
                ex.sa_lookup(); ms.append(ex.stage_ms("sa_lookup_kernel"))
            ex.set_option("count_probes", 1); ex.sa_lookup(); ex.set_option("count_probes", 0)
            out["tables<=%d" % maxl] = {"min_ms": round(min(ms), 4), "median_ms": round(float(np.median(ms)), 4), "first_ms": round(ms[0], 4),
                                        "buckets_read": int(ex.stage_ms("sa_probe_slots")), "search_probes": int(ex.stage_ms("sa_probe_search")), "lookups": int(ex.stage_ms("sa_probe_lookups"))}
        ex.set_option("ngram_tables", 5)
        print(json.dumps({"pairs": pairs, "source_tokens": int(len(corpus["str"])), "queries": nq, "query_tokens": int(len(qtok)), "table_GB": round(ex.stage_ms("ngram_table_bytes") / 1e9, 1), **out}), flush=True)
    ex.close(); host.close()
