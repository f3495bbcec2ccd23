#!/usr/bin/env python3
"""Synthetic parallel-corpus fixtures for the cgx hot path.

The reference ships no test data (its README points at toy/hansards.*, absent from the
checkout), so every fixture is generated here from a seed.  Output is the reference's own
input format (README.md:57-66): one sentence per line, space separated tokens; alignment
lines of "i-j" pairs; a text lexical table "src tgt p1 p2" with NULL rows.

Constraints honoured: >= 100 distinct source tokens, sentences
shorter than 255 tokens, query sentences <= 128 tokens unless asked otherwise, no
duplicate (src,tgt) rows in the lexical table.

    python tools/gen_fixture.py OUTDIR --pairs 400 --vocab 160 --queries 7 --seed 7
"""
import argparse
import os

import numpy as np


def zipf_sampler(rng, vocab, s=1.0):
    w = 1.0 / np.arange(1, vocab + 1) ** s
    cdf = np.cumsum(w / w.sum())
    return lambda n: np.minimum(np.searchsorted(cdf, rng.random(n)), vocab - 1)


def make_corpus(rng, pairs, vocab, lo, hi):
    """Source ids are Zipf ranks; the target is a noisy, mostly monotone copy."""
    draw = zipf_sampler(rng, vocab)
    src, tgt, ali = [], [], []
    for _ in range(pairs):
        n = int(rng.integers(lo, hi + 1))
        s = draw(n)
        t, links = [], []
        for i, w in enumerate(s):
            r = rng.random()
            if r < 0.10:                      # unaligned source word
                continue
            if rng.random() < 0.10:           # inserted (unaligned) target word
                t.append(int(draw(1)[0]) + vocab)
            t.append(int(w))
            links.append((i, len(t) - 1))
            if rng.random() < 0.08:           # one-to-two link
                t.append(int(w) + 2 * vocab)
                links.append((i, len(t) - 1))
        if not t:                             # keep every line non-empty
            t.append(int(s[0])); links.append((0, 0))
        # local reordering: swap neighbouring target words now and then
        for j in range(len(t) - 1):
            if rng.random() < 0.07:
                t[j], t[j + 1] = t[j + 1], t[j]
                links = [(a, j + 1 if b == j else j if b == j + 1 else b) for a, b in links]
        src.append([int(x) for x in s]); tgt.append(t); ali.append(sorted(links))
    return src, tgt, ali


def sw(i):
    return "s%d" % i


def tw(i):
    return "t%d" % i


def write_fixture(out, pairs=400, vocab=160, queries=7, seed=7, lo=4, hi=18, oov=True, long_query=False):
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    src, tgt, ali = make_corpus(rng, pairs, vocab, lo, hi)
    with open(os.path.join(out, "corpus.f"), "w") as f:
        for s in src:
            f.write(" ".join(sw(x) for x in s) + "\n")
    with open(os.path.join(out, "corpus.e"), "w") as f:
        for t in tgt:
            f.write(" ".join(tw(x) for x in t) + "\n")
    with open(os.path.join(out, "corpus.a"), "w") as f:
        for a in ali:
            f.write(" ".join("%d-%d" % p for p in a) + "\n")
    # lexical table: every aligned pair + NULL rows, U(0,1) probabilities
    seen_pairs, swords, twords = set(), set(), set()
    for s, t, a in zip(src, tgt, ali):
        swords.update(s); twords.update(t)
        for i, j in a:
            seen_pairs.add((s[i], t[j]))
    rows = [("NULL", "NULL")]
    rows += [(sw(x), "NULL") for x in sorted(swords)]
    rows += [("NULL", tw(x)) for x in sorted(twords)]
    rows += [(sw(a), tw(b)) for a, b in sorted(seen_pairs)]
    order = rng.permutation(len(rows))
    with open(os.path.join(out, "lex.txt"), "w") as f:
        for k in order:
            a, b = rows[k]
            f.write("%s %s %.6f %.6f\n" % (a, b, rng.random(), rng.random()))
    # queries: half verbatim corpus sentences, half with 15 % substitutions; one OOV token
    draw = zipf_sampler(rng, vocab)
    with open(os.path.join(out, "query.f"), "w") as f:
        for q in range(queries):
            s = list(src[int(rng.integers(0, pairs))])
            if q % 2 == 1:
                for i in range(len(s)):
                    if rng.random() < 0.15:
                        s[i] = int(draw(1)[0])
            words = [sw(x) for x in s]
            if oov and q == queries // 2:
                words.insert(len(words) // 2, "OOVWORD")
            if long_query and q == queries - 1:   # > 128 tokens: exercises the K1 truncation
                extra = []
                while len(words) + len(extra) < 140:
                    extra += [sw(x) for x in src[int(rng.integers(0, pairs))]]
                words = (words + extra)[:140]
            f.write(" ".join(words) + "\n")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--pairs", type=int, default=400)
    ap.add_argument("--vocab", type=int, default=160)
    ap.add_argument("--queries", type=int, default=7)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--lo", type=int, default=4)
    ap.add_argument("--hi", type=int, default=18)
    ap.add_argument("--no-oov", action="store_true")
    ap.add_argument("--long-query", action="store_true")
    a = ap.parse_args()
    write_fixture(a.out, a.pairs, a.vocab, a.queries, a.seed, a.lo, a.hi, not a.no_oov, a.long_query)
