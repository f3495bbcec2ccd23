"""Brute-force checker written from the DEFINITIONS of the hot path's results, not from the
kernels and not from oracle/: a third opinion for the stages whose oracle cannot be pinned against reference object code.
Plain numpy / Python loops, usable on small fixtures only.  TESTS ONLY.

Definitions used (reference file:line where the definition is read off):
  * token ids: words >= 2, 1 closes a sentence (Start.cu:288-327); RLP word = (L<<24)|(R<<16)|(P<<8), the delimiter slot
    before a sentence holds the target offset of that sentence (ExtractPair.cu:2717-2731, GappyLook.cu:62-78);
  * a source span [s,e] is CONSISTENT iff its first and last token are aligned, the target span [min L, max R] of its
    aligned tokens is narrower than 15 words, and the aligned target words of that span project back onto exactly [s,e]
    (consistent / checkBoundaryGap / checkBoundary, ExtractPair.cu:103-133, GappyLook.cu:43-126, ExtractPair.cu:252-342);
  * hit set of a X b: every (start, span-1) with a at start, b at start+|a|+g, g >= 1, the whole thing inside one sentence,
    span = |a|+g+|b| <= 15, and the gap [start+|a|, start+|a|+g-1] consistent (GappyLook.cu:128-474; Appendix B: the three
    scan strategies give one set);
  * hit set of a X b X c: a hit (start, l) of a X b extended by a second consistent gap of g2 >= 1 tokens and the single
    token c, span l+1+g2+1 <= 15 (GappyLook.cu:476-737);
  * frequent-pair list of (a, b), both among the 100 most frequent tokens: every (i, d) with str[i] = a, str[i+d] = b,
    d >= 2, d+1 <= 15, same sentence, gap consistent (GappyLook.cu:740-870);
  * sampling: all n occurrences when n <= S, else index ROUND(k * (float)n/(float)S) for k < S in IEEE float
    (ExtractPair.cu:1143-1160); S = 300 contiguous, 65 one gap, 70 two gaps;
  * rule `ab` of a sampled occurrence of a contiguous phrase: emitted iff the phrase span is consistent; its target span
    is that [min L, max R] (ExtractPair.cu:1203-1262); rule `aXb` / `aXbXc` of a sampled hit: emitted iff the whole span
    is consistent (and the gaps are, which the hit set already guarantees); gap target spans are the gaps' own
    [min L, max R] (ExtractPair.cu:351-624, 891-1053).
"""
import numpy as np

MAX_SPAN = 15


class Corpus:
    def __init__(self, str_, rlp, ltar, rtar):
        self.s = np.asarray(str_, np.int64); n = len(self.s)
        rlp = np.asarray(rlp, np.uint32)[:n]
        self.L = ((rlp >> 24) & 255).astype(np.int64); self.R = ((rlp >> 16) & 255).astype(np.int64); self.P = ((rlp >> 8) & 255).astype(np.int64)
        self.rlp = rlp
        self.ltar = np.asarray(ltar, np.int64); self.rtar = np.asarray(rtar, np.int64)
        delim = np.nonzero(self.s < 2)[0]
        # next_delim[i] = first index >= i holding a token < 2
        nd = np.full(n + 1, n, np.int64); nd[delim] = delim
        self.next_delim = np.minimum.accumulate(nd[::-1])[::-1]

    def aligned(self, k):
        return self.L[k] != 255 and self.R[k] != 255

    def sentence(self, k):
        """(index of the first token of k's sentence, target offset of that sentence)"""
        src0 = int(k - self.P[k])
        return src0, (0 if src0 == 0 else int(self.rlp[src0 - 1]))

    def target_span(self, s, e):
        """[lo, hi] in-sentence target positions of the aligned tokens of [s,e], or None"""
        ks = [k for k in range(s, e + 1) if self.aligned(k)]
        if not ks:
            return None
        return int(min(self.L[k] for k in ks)), int(max(self.R[k] for k in ks))

    def consistent(self, s, e):
        """target span (absolute positions) when source span [s,e] is a consistent phrase, else None"""
        if not (self.aligned(s) and self.aligned(e)):
            return None
        lo, hi = self.target_span(s, e)
        if hi - lo >= MAX_SPAN:
            return None
        src0, tb = self.sentence(s)
        back = [(self.ltar[j], self.rtar[j]) for j in range(tb + lo, tb + hi + 1) if self.ltar[j] != 255 and self.rtar[j] != 255]
        if not back:
            return None
        if src0 + min(b[0] for b in back) != s or src0 + max(b[1] for b in back) != e:
            return None
        return tb + lo, tb + hi

    def find(self, toks):
        """start positions of a phrase (all tokens >= 2, so it cannot cross a delimiter)"""
        toks = list(toks); n = len(self.s); m = len(toks)
        ok = np.ones(n - m + 1, bool)
        for j, t in enumerate(toks):
            ok &= self.s[j:n - m + 1 + j] == t
        return np.nonzero(ok)[0]


def onegap_hits(c, a, b):
    """sorted list of (start, span - 1) of a X b"""
    out = []
    for st in c.find(a):
        st = int(st); end_sent = int(c.next_delim[st])
        for g in range(1, MAX_SPAN):
            span = len(a) + g + len(b)
            if span > MAX_SPAN or st + span > end_sent:
                break
            bs = st + len(a) + g
            if all(c.s[bs + j] == b[j] for j in range(len(b))) and c.consistent(st + len(a), bs - 1) is not None:
                out.append((st, span - 1))
    return sorted(out)


def frequent_pair_list(c, a, b):
    """sorted (start, d) of the frequent-token pair (a, b)"""
    return [(st, l) for st, l in onegap_hits(c, [a], [b])]


def twogap_hits(c, base_hits, ctok):
    """extend hits (start, l) of a X b by X c: sorted (start, l, l2)"""
    out = []
    for st, l in base_hits:
        end_sent = int(c.next_delim[st])
        for g2 in range(1, MAX_SPAN):
            l2 = l + g2 + 1                                   # offset of c
            if l2 + 1 > MAX_SPAN or st + l2 >= end_sent:
                break
            if c.s[st + l2] == ctok and c.consistent(st + l + 1, st + l2 - 1) is not None:
                out.append((st, l, l2))
    return sorted(out)


def sample_indices(n, S):
    if n <= S:
        return list(range(n))
    step = np.float32(n) / np.float32(S)
    return [int(np.float64(np.float32(k) * step) + 0.5) for k in range(S)]


def contiguous_rules(c, sa, blocks):
    """sorted (block, tar_start, tar_end - tar_start) of the `ab` rules"""
    out = []
    for bn, b in enumerate(blocks):
        n = int(b["end"]) - int(b["start"]) + 1
        for x in sample_indices(n, 300):
            cur = int(sa[int(b["start"]) + x]); t = c.consistent(cur, cur + int(b["matchlen"]) - 1)
            if t is not None:
                out.append((bn, t[0], t[1] - t[0]))
    return sorted(out)


def onegap_rules(c, pid, hits, a_len, b_len):
    """`aXb` rules of the sampled hits [(start, l)] of pattern pid: sorted (pid, tstart, end, gap1, gap1_1)"""
    out = []
    for x in sample_indices(len(hits), 65):
        st, l = hits[x]
        whole = c.consistent(st, st + l); gap = c.consistent(st + a_len, st + l - b_len)
        if whole is not None and gap is not None:
            out.append((pid, whole[0], whole[1] - whole[0], gap[0] - whole[0], gap[1] - whole[0]))
    return sorted(out)


def twogap_rules(c, pid, hits, a_len, b_len):
    """`aXbXc` rules of the sampled hits [(start, l, l2)]: sorted (pid, tstart, end, gap1, gap1_1, gap2, gap2_1)"""
    out = []
    for x in sample_indices(len(hits), 70):
        st, l, l2 = hits[x]
        whole = c.consistent(st, st + l2); g1 = c.consistent(st + a_len, st + l - b_len); g2 = c.consistent(st + l + 1, st + l2 - 1)
        if whole is not None and g1 is not None and g2 is not None:
            out.append((pid, whole[0], whole[1] - whole[0], g1[0] - whole[0], g1[1] - whole[0], g2[0] - whole[0], g2[1] - whole[0]))
    return sorted(out)


def check_batch(c, sa, freq, phits, s1, p1d, hits1, s2, c2d, hits2, blocks, r0, r1, sep1, r2, sep2a, sep2b, max_patterns=None, pick=None):
    """Compares one batch's device (or oracle) results with the definitions above.  Arrays are the wire records of
    include/cgx.h (numpy structured arrays); `phits` = list of (start, len) of the whole frequent-pair table with `pidx`
    folded in by the caller as a dict pair -> list.  `max_patterns`: only the first so many one-gap patterns (and the
    two-gap patterns built on them); `pick`: only these one-gap pattern ids.  Returns the number of patterns / blocks checked."""
    freq = [int(x) for x in freq]; rank = {t: i for i, t in enumerate(freq)}
    checked = 0
    hits_of = {}
    d1 = len(s1) if max_patterns is None else min(len(s1), max_patterns)
    pids = sorted(set(int(x) for x in pick if 0 <= int(x) < len(s1))) if pick is not None else list(range(d1))
    chosen = set(pids)
    for pid in pids:
        s = s1[pid]; pat = [int(x) for x in p1d[pid]["pat"][:int(p1d[pid]["number"])]]
        al, bl = int(s["a_len"]), int(s["b_len"])
        a, b = pat[:al], pat[al + 1:al + 1 + bl]
        assert pat[al] == -1 and len(b) == bl
        want = onegap_hits(c, a, b)
        marker = al == 1 and bl == 1 and a[0] in rank and b[0] in rank
        if int(s["sa_start"]) == -1:
            assert want == [], (pid, "no hits reported, the definition finds", want[:3])
            hits_of[pid] = []
        elif marker:
            pre = rank[a[0]] * 100 + rank[b[0]]
            assert int(s["sa_start"]) == int(s["sa_end"]) and ("marker" not in s.dtype.names or int(s["marker"]) == 1)
            h = hits1[int(s["sa_start"])]
            assert (int(h["position"]), int(h["str_position"]), int(h["length"])) == (pid, pre, 0), pid
            assert phits.get(pre, []) == want, (pid, "frequent-pair list")
            hits_of[pid] = want
        else:
            got = [(int(h["str_position"]), int(h["length"])) for h in hits1[int(s["sa_start"]):int(s["sa_end"]) + 1]]
            assert all(int(h["position"]) == pid for h in hits1[int(s["sa_start"]):int(s["sa_end"]) + 1])
            assert got == want, (pid, a, b, got[:3], want[:3])
            hits_of[pid] = want
        checked += 1
    # aXb rules: r1[sep1:] holds them with id = pattern id
    got_r1 = sorted((int(r["id"]), int(r["tstart"]), int(r["end"]), int(r["gap1"]), int(r["gap1_1"])) for r in r1[sep1:] if int(r["id"]) in chosen)
    want_r1 = sorted(sum((onegap_rules(c, pid, hits_of[pid], int(s1[pid]["a_len"]), int(s1[pid]["b_len"])) for pid in pids), []))
    assert got_r1 == want_r1, ("aXb rules", len(got_r1), len(want_r1))
    # two-gap patterns
    two_hits = {}
    for tid in range(len(s2)):
        one = int(s2[tid]["blockid"])
        if one not in chosen:
            continue
        want = twogap_hits(c, hits_of[one], int(c2d[tid]))
        if int(s2[tid]["sa_start"]) == -1:
            assert want == [], (tid, want[:3])
        else:
            sl = hits2[int(s2[tid]["sa_start"]):int(s2[tid]["sa_end"]) + 1]
            got = [(int(h["str_position"]), int(h["length"]), int(h["length2"])) for h in sl]
            assert all(int(h["position"]) == tid for h in sl) and got == want, (tid, got[:3], want[:3])
        two_hits[tid] = want
        checked += 1
    got_r2 = sorted((int(r["id"]), int(r["tstart"]), int(r["end"]), int(r["gap1"]), int(r["gap1_1"]), int(r["gap2"]), int(r["gap2_1"])) for r in r2[sep2a:sep2b] if int(r["id"]) in two_hits)
    want_r2 = sorted(sum((twogap_rules(c, tid, two_hits[tid], int(s1[int(s2[tid]["blockid"])]["a_len"]), int(s1[int(s2[tid]["blockid"])]["b_len"])) for tid in two_hits), []))
    assert got_r2 == want_r2, ("aXbXc rules", len(got_r2), len(want_r2))
    # contiguous rules
    got_r0 = sorted((int(r["block"]), int(r["tar_start"]), int(r["tar_end"])) for r in r0)
    assert got_r0 == contiguous_rules(c, sa, blocks), "ab rules"
    return checked + len(blocks)
