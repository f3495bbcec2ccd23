"""Synthetic parallel corpora at token-id level (BASELINE.json configs 3-5).

Vectorised numpy so that 10^7-sentence corpora are generated in seconds.  Model: source
tokens Zipf(s=1) over V words, sentence length U[lo,hi]; each source word is unaligned with
p=0.10, otherwise linked to one target word (same word id), with p=0.10 an unaligned target
word inserted before it and with p=0.08 a second target word linked to the same source word;
adjacent target words swap with p=0.07 (links follow).  Ids follow the reference convention
(words >= 2, 1 closes a sentence, trailing "1,last").  The lexical table holds every pair
the model can produce plus NULL rows, with U(0,1) probabilities.
"""
import numpy as np

LEXKEY = np.dtype([("src", "<i4"), ("tgt", "<i4")])
LEXVAL = np.dtype([("v1", "<f4"), ("v2", "<f4")])


def _zipf_ids(rng, n, vocab):
    """Zipf(s~1) ranks by inverse CDF: P(r) ~ ln((r+2)/(r+1)) ~ 1/(r+1)."""
    r = np.exp(rng.random(n, dtype=np.float32) * np.float32(np.log(vocab + 1.0))).astype(np.int32) - 1
    return np.clip(r, 0, vocab - 1)


def make_source_tokens(pairs, vocab, seed, lo=5, hi=45):
    """Only the source token array of make_corpus(pairs, vocab, seed, lo, hi) (the same first draws of the same generator):
    what a suffix-array builder needs, in a tenth of the time and memory."""
    rng = np.random.default_rng(seed)
    slen = rng.integers(lo, hi + 1, pairs).astype(np.int64)
    ns = int(slen.sum())
    words = _zipf_ids(rng, ns, vocab) + 2
    sent_of = np.repeat(np.arange(pairs, dtype=np.int32), slen)
    n = ns + pairs + 2
    str_ = np.ones(n, np.int32)
    str_[np.arange(ns, dtype=np.int64) + sent_of] = words
    str_[n - 1] = int(words.max()) + 1
    return str_


def make_corpus(pairs, vocab, seed, lo=5, hi=45):
    rng = np.random.default_rng(seed)
    slen = rng.integers(lo, hi + 1, pairs).astype(np.int64)
    ns = int(slen.sum())
    words = _zipf_ids(rng, ns, vocab) + 2                       # source word ids 2..V+1
    sstart = np.concatenate(([0], np.cumsum(slen)))
    sent_of = np.repeat(np.arange(pairs, dtype=np.int32), slen)
    pos = (np.arange(ns, dtype=np.int64) - sstart[:-1][sent_of]).astype(np.int32)   # in-sentence position
    aligned = rng.random(ns, dtype=np.float32) >= 0.10
    aligned[sstart[:-1]] = True                                 # keep at least one link per sentence
    ins = aligned & (rng.random(ns, dtype=np.float32) < 0.10)
    dbl = aligned & (rng.random(ns, dtype=np.float32) < 0.08)
    per = aligned.astype(np.int8) + ins + dbl                   # target words produced by each source word
    cper = np.concatenate(([0], np.cumsum(per, dtype=np.int64)))
    nt_words = int(cper[-1])
    tlen = cper[sstart[1:]] - cper[sstart[:-1]]
    tstart_w = np.concatenate(([0], np.cumsum(tlen)))
    owner = np.repeat(np.arange(ns, dtype=np.int64), per)       # source token producing each target word
    k = (np.arange(nt_words, dtype=np.int64) - cper[:-1][owner]).astype(np.int8)
    is_ins = (k == 0) & ins[owner]
    is_dbl = (k == per[owner] - 1) & dbl[owner] & ~is_ins
    tword = np.where(is_ins, _zipf_ids(rng, nt_words, vocab) + 2 + vocab, np.where(is_dbl, words[owner] + 2 * vocab, words[owner])).astype(np.int32)
    tsent_of = sent_of[owner]
    opos = (np.arange(nt_words, dtype=np.int64) - tstart_w[:-1][tsent_of]).astype(np.int32)   # position before reordering
    # local reordering: swap disjoint neighbouring target words inside a sentence (links follow the words)
    cand = (rng.random(nt_words, dtype=np.float32) < 0.07) & (opos + 1 < tlen[tsent_of])
    cand[1:] &= ~cand[:-1]
    fpos = opos.copy()
    i = np.nonzero(cand)[0]
    fpos[i] += 1; fpos[i + 1] -= 1
    # token arrays with delimiters and the trailing "1,last"
    n = ns + pairs + 2
    str_ = np.ones(n, np.int32)
    sentind = (sstart + np.arange(pairs + 1)).astype(np.int32)
    str_[np.arange(ns, dtype=np.int64) + sent_of] = words
    str_[n - 1] = int(words.max()) + 1
    nt = nt_words + pairs + 2
    tstr = np.ones(nt, np.int32)
    tsentind = (tstart_w + np.arange(pairs + 1)).astype(np.int32)
    tslot = tsentind[:-1].astype(np.int64)[tsent_of] + fpos     # final slot of each produced target word
    tstr[tslot] = tword
    tstr[nt - 1] = int(tword.max()) + 1
    if int(slen.max()) >= 255 or int(tlen.max()) >= 255:
        raise ValueError("sentence too long for the reference's byte positions")
    # alignment bytes: per source token min/max target position over its linked words (groups are contiguous in owner order)
    lsrc = np.full(n, 255, np.uint8); rsrc = np.full(n, 255, np.uint8)
    ltar = np.full(nt, 255, np.uint8); rtar = np.full(nt, 255, np.uint8)
    linked = ~is_ins
    lo_v = np.where(linked, fpos, 1 << 20); hi_v = np.where(linked, fpos, -1)
    has = np.nonzero(per > 0)[0]
    mn = np.minimum.reduceat(lo_v, cper[:-1][has]); mx = np.maximum.reduceat(hi_v, cper[:-1][has])
    sslot = has + sent_of[has]
    lsrc[sslot] = mn.astype(np.uint8); rsrc[sslot] = mx.astype(np.uint8)
    lt = tslot[linked]
    ltar[lt] = pos[owner[linked]].astype(np.uint8); rtar[lt] = ltar[lt]          # every target word has at most one link
    # lexical table
    sw = np.arange(2, vocab + 2, dtype=np.int32)
    rows_s = np.concatenate(([-1], sw, np.full(3 * vocab, -1, np.int32), sw, sw)).astype(np.int32)
    rows_t = np.concatenate(([-1], np.full(vocab, -1, np.int32), np.arange(2, 3 * vocab + 2, dtype=np.int32), sw, sw + 2 * vocab)).astype(np.int32)
    lexk = np.zeros(len(rows_s), LEXKEY); lexk["src"] = rows_s; lexk["tgt"] = rows_t
    lexv = np.zeros(len(rows_s), LEXVAL); lexv["v1"] = rng.random(len(rows_s), dtype=np.float32); lexv["v2"] = rng.random(len(rows_s), dtype=np.float32)
    return dict(str=str_, sentind=sentind, tstr=tstr, tsentind=tsentind, lsrc=lsrc, rsrc=rsrc, ltar=ltar, rtar=rtar, lexk=lexk, lexv=lexv,
                pairs=pairs, vocab=vocab)


def make_queries(corpus, nq, seed, sent_limit=None):
    """Half verbatim corpus sentences, half with 15 % of the words substituted."""
    rng = np.random.default_rng(seed)
    pairs = corpus["pairs"] if sent_limit is None else min(sent_limit, corpus["pairs"])
    pick = rng.integers(0, pairs, nq)
    si = corpus["sentind"]
    lens = (si[pick + 1] - si[pick] - 1).astype(np.int64)
    qoff = np.concatenate(([0], np.cumsum(lens)))[:-1].astype(np.int32)
    idx = np.repeat(si[pick].astype(np.int64), lens) + (np.arange(int(lens.sum())) - np.repeat(qoff.astype(np.int64), lens))
    qtok = corpus["str"][idx].copy()
    noisy = np.repeat(np.arange(nq) % 2 == 1, lens) & (rng.random(len(qtok)) < 0.15)
    qtok[noisy] = _zipf_ids(rng, int(noisy.sum()), corpus["vocab"]) + 2
    return qoff, qtok.astype(np.int32)


def prefix(corpus, pairs):
    """The first `pairs` sentence pairs as a corpus of their own (for the bounded CPU baseline)."""
    n = int(corpus["sentind"][pairs]); nt = int(corpus["tsentind"][pairs])
    def cut(a, m, last):
        out = np.empty(m + 2, a.dtype); out[:m] = a[:m]; out[m] = 1; out[m + 1] = last; return out
    str_ = cut(corpus["str"], n, int(corpus["str"][:n].max()) + 1); tstr = cut(corpus["tstr"], nt, int(corpus["tstr"][:nt].max()) + 1)
    pad = lambda a, m: np.concatenate((a[:m], np.full(2, 255, np.uint8)))
    return dict(str=str_, sentind=corpus["sentind"][:pairs + 1].copy(), tstr=tstr, tsentind=corpus["tsentind"][:pairs + 1].copy(),
                lsrc=pad(corpus["lsrc"], n), rsrc=pad(corpus["rsrc"], n), ltar=pad(corpus["ltar"], nt), rtar=pad(corpus["rtar"], nt),
                lexk=corpus["lexk"], lexv=corpus["lexv"], pairs=pairs, vocab=corpus["vocab"])
