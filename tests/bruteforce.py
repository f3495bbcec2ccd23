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


NOPOS = 1 << 30              # "no position yet" for a running minimum (the reference starts its unsigned chars at 255, above every position it allows)


class Corpus:
    NA = 255                 # "not aligned" in the position arrays (the reference's byte tables); from_text uses a value above every position

    def __init__(self, str_, rlp, ltar, rtar):
        self.s = np.asarray(str_, np.int64); n = len(self.s)
        rlp = np.asarray(rlp, np.uint32)[:n]
        self.L = ((rlp >> 24) & 255).astype(np.int64); self.R = ((rlp >> 16) & 255).astype(np.int64); self.P = ((rlp >> 8) & 255).astype(np.int64)
        self.rlp = rlp; self.tb = None
        self.ltar = np.asarray(ltar, np.int64); self.rtar = np.asarray(rtar, np.int64)
        self._delims()

    def _delims(self):
        n = len(self.s)
        delim = np.nonzero(self.s < 2)[0]
        # next_delim[i] = first index >= i holding a token < 2
        nd = np.full(n + 1, n, np.int64); nd[delim] = delim
        self.next_delim = np.minimum.accumulate(nd[::-1])[::-1]

    @classmethod
    def from_text(cls, str_, target_file, align_file):
        """The same object built from the alignment TEXT (one line of `i-j` links per sentence pair, ExtractPair.cu:2639-2739)
        instead of from packed alignment words: positions are plain integers of any size, so this also serves corpora whose
        sentences are longer than the reference's byte positions allow (the opt-in long-sentence mode).
        Token ids come from `str_` (1 closes a sentence); the target side only contributes its sentence lengths."""
        c = cls.__new__(cls); c.NA = NOPOS
        c.s = np.asarray(str_, np.int64); n = len(c.s)
        c.L = np.full(n, NOPOS, np.int64); c.R = np.full(n, NOPOS, np.int64); c.P = np.zeros(n, np.int64); c.tb = np.zeros(n, np.int64); c.rlp = None
        tlens = [len(line.split()) for line in open(target_file)]
        links = [[tuple(int(x) for x in w.split("-")) for w in line.split()] for line in open(align_file)]
        nt = sum(tlens) + len(tlens) + 4
        c.ltar = np.full(nt, NOPOS, np.int64); c.rtar = np.full(nt, NOPOS, np.int64)
        k = 0; tb = 0
        for tl, lk in zip(tlens, links):
            a = k
            while k < n and c.s[k] >= 2:
                c.P[k] = k - a; c.tb[k] = tb; k += 1
            c.tb[k] = tb
            sl = k - a; k += 1                                   # past the delimiter
            for i, j in lk:
                assert 0 <= i < sl and 0 <= j < tl, "alignment link outside its sentence pair"
                c.L[a + i] = j if c.L[a + i] == NOPOS else min(c.L[a + i], j); c.R[a + i] = j if c.R[a + i] == NOPOS else max(c.R[a + i], j)
                c.ltar[tb + j] = i if c.ltar[tb + j] == NOPOS else min(c.ltar[tb + j], i); c.rtar[tb + j] = i if c.rtar[tb + j] == NOPOS else max(c.rtar[tb + j], i)
            tb += tl + 1                                         # the target side has a delimiter per sentence too
        c._delims()
        return c

    def aligned(self, k):
        return self.L[k] != self.NA and self.R[k] != self.NA

    def sentence(self, k):
        """(index of the first token of k's sentence, target offset of that sentence)"""
        src0 = int(k - self.P[k])
        if self.tb is not None:
            return src0, int(self.tb[k])
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
        back = [(self.ltar[j], self.rtar[j]) for j in range(tb + lo, tb + hi + 1) if self.ltar[j] != self.NA and self.rtar[j] != self.NA]
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


# =====================================================================================================================
# Round 3: the remaining stages without a reference-held pin -- query-side enumeration, the extension rules
# (Xab / abX / XabX of a contiguous phrase, XaXb / aXbX of a one-gap pattern) and the MaxLex features.
# Written from the reference text itself (file:line below) and SURVEY Appendix B, NOT from cgx_rules.h and NOT from
# oracle/: a separate restatement in another language and another shape (per-occurrence generators over python ints).
# =====================================================================================================================
import ctypes as _C
import ctypes.util as _Cu

_libm = _C.CDLL(_Cu.find_library("m") or "libm.so.6")
_libm.log10f.restype = _C.c_float; _libm.log10f.argtypes = [_C.c_float]
F32 = np.float32
MAX_SYMBOLS = 5


def corpus_ngrams(c, maxn=5):
    """every phrase of <= maxn tokens that occurs inside a sentence, as a set of tuples"""
    s = [int(x) for x in c.s]; out = set()
    for i, t in enumerate(s):
        if t < 2:
            continue
        g = ()
        for l in range(maxn):
            if i + l >= len(s) or s[i + l] < 2:
                break
            g = g + (s[i + l],); out.add(g)
    return out


def longest_matches(grams, qoff, qtok, k1_limit=128, cap=5):
    """lm[t] = min(cap, longest prefix of q[t..end of sentence) that occurs in the corpus), OOV (-1) ends a match;
    0 for tokens at in-sentence position >= k1_limit (K1 launches 128 threads per sentence, SuffixArray.cu:1374-1378)"""
    T = len(qtok); lm = [0] * T; qoff = list(qoff) + [T]
    for q in range(len(qoff) - 1):
        for t in range(qoff[q], qoff[q + 1]):
            if t - qoff[q] >= k1_limit:
                continue
            g = (); l = 0
            while l < cap and t + l < qoff[q + 1] and qtok[t + l] >= 2 and (g + (int(qtok[t + l]),)) in grams:
                g = g + (int(qtok[t + l]),); l += 1
            lm[t] = l
    return lm


def enumerate_onegap(qoff, qtok, lm):
    """oneGapEnumeration (SuffixArray.cu:928-1039): instances (t, a_len, s, b_len) and their symbol tuples (gap = -1)."""
    T = len(qtok); ends = list(qoff[1:]) + [T]; inst = []
    for q in range(len(qoff)):
        end = ends[q]
        for t in range(qoff[q], end):
            if t >= T - 1 or t == end - 1 or t == end - 2:
                continue
            for al in range(1, lm[t] + 1):
                s = t + al + 1
                while s < end and s - t <= MAX_SPAN:
                    if qtok[s] != -1:
                        bl = 1
                        while al + 1 + bl <= MAX_SYMBOLS and bl <= lm[s] and s - t + bl - 1 <= MAX_SPAN:
                            pat = tuple(int(x) for x in qtok[t:t + al]) + (-1,) + tuple(int(x) for x in qtok[s:s + bl])
                            inst.append((q, t, al, s, bl, pat)); bl += 1
                    s += 1
    return inst


def onegap_ids(inst):
    """distinct patterns in the order of oneGapEnumerationCompare (SuffixArray.cu:51-67): number of symbols, then the
    symbols left to right as ints (the gap, -1, sorts before every token)"""
    pats = sorted(set(p for *_, p in inst), key=lambda p: (len(p), p))
    return pats, {p: i for i, p in enumerate(pats)}


def enumerate_twogap(qoff, qtok, lm, inst, idof, has_hits):
    """twoGapEnumeration (SuffixArray.cu:816-926): for every instance of a one-gap pattern that has corpus hits, the
    tokens c right of it (second gap >= 1): instances (query, one-gap id, position of c, (c,))."""
    T = len(qtok); ends = list(qoff[1:]) + [T]; out = []
    for q, t, al, s, bl, pat in inst:
        one = idof[pat]
        limit = MAX_SYMBOLS - 2 - al - bl
        if not has_hits[one] or limit < 1:
            continue
        last = s + bl - 1                                     # last token of b
        if last > T - 1:
            continue
        end = ends[q]
        for sc in range(last + 2, end):
            it = 1
            while it <= limit and it <= lm[sc] and sc - t + it - 1 <= MAX_SPAN:
                out.append((q, one, sc, tuple(int(x) for x in qtok[sc:sc + it]))); it += 1
    return out


def twogap_ids(inst2):
    """twoGapEnumerationCompare (SuffixArray.cu:31-49): one-gap id, number of symbols of c, symbols"""
    pats = sorted(set((one, c) for _, one, _, c in inst2), key=lambda x: (x[0], len(x[1]), x[1]))
    return pats, {p: i for i, p in enumerate(pats)}


def per_query_ids(nq, pairs):
    """oneGapQueryWithID / twoGapQueryWithID (SuffixArray.cu:1679-1718, 2058-2093): walking the instances in pattern
    order, a query gets each id once -> ascending distinct ids per query"""
    out = [set() for _ in range(nq)]
    for q, i in pairs:
        out[q].add(i)
    return [sorted(x) for x in out]


# ---- alignment helpers of the extension rules ------------------------------------------------------------------------
def _back(c, ts, te, s_chk, e_chk, src0):
    """consistent() (ExtractPair.cu:103-133): the aligned target words of [ts,te] project onto exactly [s_chk,e_chk]"""
    lo, hi = NOPOS, 0
    for j in range(ts, te + 1):
        l, r = int(c.ltar[j]), int(c.rtar[j])
        if l != c.NA and r != c.NA:
            lo = min(lo, l); hi = max(hi, r)
    return src0 + lo == s_chk and src0 + hi == e_chk


class _Side:
    """a gap growing one token at a time away from a phrase: running [min L, max R] of its aligned tokens"""
    def __init__(self):
        self.lo, self.hi = NOPOS, 0

    def add(self, c, k):
        if not c.aligned(k):
            return False
        self.lo = min(self.lo, int(c.L[k])); self.hi = max(self.hi, int(c.R[k]))
        return True


def block_extension_rules(c, cs, m):
    """Xab / abX / XabX of ONE occurrence (start cs, m tokens) of a contiguous phrase, extractConsistentPairs_Gappy
    (ExtractPair.cu:1162-1791).  Returns (xab, abx, xabx): each None or the rule tuple (tstart, end, gap1, gap1_1[, gap2, gap2_1])."""
    ender = cs + m - 1
    src0, tb = c.sentence(cs)
    ab = True; open_abx = True; open_xab = True                # "NoSuccess" flags: the side has not emitted yet
    mn, mx = NOPOS, 0
    for k in range(cs, ender + 1):                              # :1176-1212
        if not c.aligned(k):
            if k == cs or k == ender:
                ab = False
                if k == cs:
                    open_abx = False
                else:
                    open_xab = False
        else:
            mn = min(mn, int(c.L[k])); mx = max(mx, int(c.R[k]))
    xab = abx = xabx = True
    if mn > mx or mx - mn >= MAX_SPAN:                          # :1218-1224
        xab = abx = xabx = False
    if m + 1 > MAX_SYMBOLS:                                     # :1263-1269
        xab = abx = False
    if m + 2 > MAX_SYMBOLS:
        xabx = False
    left, right = _Side(), _Side(); nl = nr = 0                 # nl / nr = XabCount / abXCount: largest gap size validated so far
    r_xab = r_abx = r_xabx = None
    i = 1
    while m + i <= MAX_SPAN and (open_abx or open_xab or xabx):
        # ---- grow the left gap to i tokens (:1284-1399)
        if xab and cs - i >= 0 and c.s[cs - i] >= 2:
            ok = left.add(c, cs - i)
            if not ok and i == 1:
                xab = False; xabx = False
            if left.hi - left.lo >= MAX_SPAN:
                ok = False; xab = False
            if ok:
                g = (tb + left.lo, tb + left.hi)
                ok = _back(c, g[0], g[1], cs - i, cs - 1, src0)
                if ok:
                    nl = i
            if open_xab and ok:
                ts = tb + min(left.lo, mn); te = tb + max(left.hi, mx)
                if te - ts >= MAX_SPAN:
                    ok = False; xab = False
                if ok:
                    ok = _back(c, ts, te, cs - i, ender, src0)
                if ok:
                    r_xab = (ts, te - ts, g[0] - ts, g[1] - ts); open_xab = False
        else:
            xab = False
        # ---- grow the right gap to i tokens (:1404-1509)
        if abx and c.s[ender + i] >= 2:
            ok = right.add(c, ender + i)
            if not ok and i == 1:
                abx = False; xabx = False
            if right.hi - right.lo >= MAX_SPAN:
                ok = False; abx = False
            if ok:
                g = (tb + right.lo, tb + right.hi)
                ok = _back(c, g[0], g[1], ender + 1, ender + i, src0)
                if ok:
                    nr = i
            if open_abx and ok:
                ts = tb + min(right.lo, mn); te = tb + max(right.hi, mx)
                if te - ts >= MAX_SPAN:
                    ok = False; abx = False
                if ok:
                    ok = _back(c, ts, te, cs, ender + i, src0)
                if ok:
                    r_abx = (ts, te - ts, g[0] - ts, g[1] - ts); open_abx = False
        else:
            abx = False
        # ---- both gaps (:1514-1777): the side validated at THIS size is paired with every size of the other side up to
        # its best so far, smallest first
        if xabx and (abx or xab):
            if nl == i:
                oth = _Side(); ic = 1
                while xabx and ic <= nr:
                    if ic + nl + m > MAX_SPAN:
                        break
                    ok = oth.add(c, ender + ic)
                    if ok and oth.hi - oth.lo >= MAX_SPAN:
                        break
                    if ok:
                        g2 = (tb + oth.lo, tb + oth.hi)
                        ok = _back(c, g2[0], g2[1], ender + 1, ender + ic, src0)
                    if ok:
                        ts = tb + min(oth.lo, left.lo, mn); te = tb + max(oth.hi, left.hi, mx)
                        if te - ts >= MAX_SPAN:
                            break
                        if _back(c, ts, te, cs - nl, ender + ic, src0):
                            r_xabx = (ts, te - ts, tb + left.lo - ts, tb + left.hi - ts, g2[0] - ts, g2[1] - ts); xabx = False
                    ic += 1
            if xabx and nr == i:
                oth = _Side(); ic = 1
                while xabx and ic <= nl:
                    if ic + nr + m > MAX_SPAN:
                        break
                    ok = oth.add(c, cs - ic)
                    if ok and oth.hi - oth.lo >= MAX_SPAN:
                        break
                    if ok:
                        g1 = (tb + oth.lo, tb + oth.hi)
                        ok = _back(c, g1[0], g1[1], cs - ic, cs - 1, src0)
                    if ok:
                        ts = tb + min(oth.lo, right.lo, mn); te = tb + max(oth.hi, right.hi, mx)
                        if te - ts >= MAX_SPAN:
                            break
                        if _back(c, ts, te, cs - ic, ender + nr, src0):
                            r_xabx = (ts, te - ts, g1[0] - ts, g1[1] - ts, tb + right.lo - ts, tb + right.hi - ts); xabx = False
                    ic += 1
        else:
            xabx = False
        if not xabx:                                            # :1781-1788
            if not xab:
                open_xab = False
            if not abx:
                open_abx = False
        i += 1
    return r_xab, r_abx, r_xabx


def block_rules(c, sa, blocks):
    """all Xab / abX (one-gap ids b / G+b) and XabX (two-gap id b) rules of the sampled occurrences of every block"""
    G = len(blocks); r1 = []; r2 = []
    for bn, b in enumerate(blocks):
        n = int(b["end"]) - int(b["start"]) + 1; m = int(b["matchlen"])
        for x in sample_indices(n, 300):
            xab, abx, xabx = block_extension_rules(c, int(sa[int(b["start"]) + x]), m)
            if xab:
                r1.append((bn,) + xab)
            if abx:
                r1.append((G + bn,) + abx)
            if xabx:
                r2.append((bn,) + xabx)
    return sorted(r1), sorted(r2)


def onegap_outer_rules(c, st, l, al, bl):
    """XaXb / aXbX of ONE hit (start st, span l+1) of a X b, extractConsistentPairs_OneGap (ExtractPair.cu:540-884).
    Returns (xaxb, axbx): None or (tstart, end, gap1, gap1_1, gap2, gap2_1)."""
    ender = st + l; gs, ge = st + al, ender - bl
    if not (c.aligned(gs) and c.aligned(ge)):                  # checkBoundaryFast (:135-194): cannot happen for a hit
        return None, None
    src0, tb = c.sentence(gs)
    glo, ghi = c.target_span(gs, ge)
    if ghi - glo >= MAX_SPAN:
        return None, None
    gap = (tb + glo, tb + ghi)
    # checkBoundary (:252-342): whole span; codes 2 / 3 / 4 = first / last / both edge tokens unaligned
    left = c.aligned(ender); right = c.aligned(st)              # code 3 or 4 stops XaXb, code 2 or 4 stops aXbX
    mn, mx = c.target_span(st, ender)
    if al + bl + 2 > MAX_SYMBOLS:
        return None, None
    lft, rgt = _Side(), _Side(); xaxb = axbx = None
    i = 1
    while l + 1 + i <= MAX_SPAN and (left or right):
        if left and st - i >= 0 and c.s[st - i] >= 2:
            ok = lft.add(c, st - i)
            if not ok and i == 1:
                left = False
            if lft.hi - lft.lo >= MAX_SPAN:
                ok = False; left = False
            if ok:
                g = (tb + lft.lo, tb + lft.hi)
                ok = _back(c, g[0], g[1], st - i, st - 1, src0)
            if ok:
                ts = tb + min(lft.lo, mn); te = tb + max(lft.hi, mx)
                if te - ts >= MAX_SPAN:
                    ok = False; left = False
                if ok and _back(c, ts, te, st - i, ender, src0):
                    xaxb = (ts, te - ts, g[0] - ts, g[1] - ts, gap[0] - ts, gap[1] - ts); left = False
        else:
            left = False
        if right and c.s[ender + i] >= 2:
            ok = rgt.add(c, ender + i)
            if not ok and i == 1:
                right = False
            if rgt.hi - rgt.lo >= MAX_SPAN:
                ok = False; right = False
            if ok:
                g = (tb + rgt.lo, tb + rgt.hi)
                ok = _back(c, g[0], g[1], ender + 1, ender + i, src0)
            if ok:
                ts = tb + min(rgt.lo, mn); te = tb + max(rgt.hi, mx)
                if te - ts >= MAX_SPAN:
                    ok = False; right = False
                if ok and _back(c, ts, te, st, ender + i, src0):
                    axbx = (ts, te - ts, gap[0] - ts, gap[1] - ts, g[0] - ts, g[1] - ts); right = False
        else:
            right = False
        i += 1
    return xaxb, axbx


def outer_rules(c, hits_of, s1, D1):
    """XaXb (two-gap id = one-gap id) and aXbX (D1 + id) of the sampled hits of the given patterns"""
    out = []
    for pid, hits in hits_of.items():
        al, bl = int(s1[pid]["a_len"]), int(s1[pid]["b_len"])
        for x in sample_indices(len(hits), 65):
            st, l = hits[x]
            a, b = onegap_outer_rules(c, st, l, al, bl)
            if a:
                out.append((pid,) + a)
            if b:
                out.append((D1 + pid,) + b)
    return sorted(out)


# ---- MaxLex (lexicalTaskMaxEF, ExtractPair.cu:2144-2432) -------------------------------------------------------------
class LexTable:
    def __init__(self, lexk, lexv):
        self.t = {}
        for k, v in zip(lexk, lexv):
            self.t.setdefault((int(k["src"]), int(k["tgt"])), (F32(v["v1"]), F32(v["v2"])))

    def get(self, s, t, which):
        """value 1 or 2 of row (s, t); an absent row scores 0 (searchLexFile, :2108-2142)"""
        v = self.t.get((int(s), int(t)))
        return F32(0) if v is None else v[which - 1]


def maxlex(table, tstr, src, tstart, end, gaps):
    """(MaxLexFgivenE, MaxLexEgivenF) in float32: per source word the best value-2 score over the target words outside the
    gaps and NULL (-1), per target word outside the gaps the best value-1 score over NULL and the source words; each word
    adds -log10f(best) or 99 when nothing scores.  `gaps` = [(first, last)] relative to tstart."""
    tw = [j for j in range(tstart, tstart + end + 1) if not any(tstart + a <= j <= tstart + b for a, b in gaps)]
    fe = F32(0); ef = F32(0)
    for s in src:
        best = F32(0)
        if tw:
            best = max(best, table.get(s, -1, 2))
        for j in tw:
            best = max(best, table.get(s, int(tstr[j]), 2))
        fe = F32(fe + (F32(-_libm.log10f(_C.c_float(best))) if best > 0 else F32(99.0)))
    for j in tw:
        best = F32(0)
        if len(src):
            best = max(best, table.get(-1, int(tstr[j]), 1))
        for s in src:
            best = max(best, table.get(s, int(tstr[j]), 1))
        ef = F32(ef + (F32(-_libm.log10f(_C.c_float(best))) if best > 0 else F32(99.0)))
    return fe, ef


def lexline_source(kind, lid, G, D1, D2, blocks, c, p1d, c2d, one2):
    """source-side terminals of lexicon line id `lid` (converted ids: ExtractPair.c:724-728, 1000-1006, 609-611)"""
    def blk(b):
        return [int(x) for x in c.s[int(blocks[b]["string_start"]):int(blocks[b]["string_start"]) + int(blocks[b]["matchlen"])]]
    def pat(p):
        return [int(x) for x in p1d[p]["pat"][:int(p1d[p]["number"])] if int(x) >= 0]
    if kind == 0:
        return blk(lid)
    if kind == 1:
        return blk(lid) if lid < G else blk(lid - G) if lid < 2 * G else pat(lid - 2 * G)
    if lid < G:
        return blk(lid)
    if lid < G + D2:
        return pat(int(one2[lid - G])) + [int(c2d[lid - G])]
    return pat(lid - G - D2) if lid < G + D2 + D1 else pat(lid - G - D2 - D1)


def _t1(r):
    return (int(r["id"]), int(r["tstart"]), int(r["end"]), int(r["gap1"]), int(r["gap1_1"]))


def _t2(r):
    return _t1(r) + (int(r["gap2"]), int(r["gap2_1"]))


def check_query_side(c, qoff, qtok, lm, g1_count, s1, p1d, qone, g2_count, s2, c2d, qtwo, k1_limit=128):
    """Longest matches, one-/two-gap enumeration, distinct-pattern ids and per-query id lists of a batch against the
    definitions.  `qone` / `qtwo`: list (per query) of id lists.  Returns (#one-gap patterns, #two-gap patterns)."""
    qoff = [int(x) for x in qoff]; qtok = [int(x) for x in qtok]; nq = len(qoff)
    want_lm = longest_matches(corpus_ngrams(c), qoff, qtok, k1_limit)
    assert want_lm == [min(int(x), 5) for x in lm], "longest matches"
    inst = enumerate_onegap(qoff, qtok, want_lm)
    pats, idof = onegap_ids(inst)
    assert len(inst) == g1_count and len(pats) == len(s1), ("one-gap enumeration", len(inst), g1_count, len(pats), len(s1))
    for i, p in enumerate(pats):
        assert tuple(int(x) for x in p1d[i]["pat"][:int(p1d[i]["number"])]) == p, ("one-gap pattern id", i)
        s = s1[i]; t, al, bl, gp = int(s["qrystart"]), int(s["a_len"]), int(s["b_len"]), int(s["gap"])
        assert tuple(qtok[t:t + al]) + (-1,) + tuple(qtok[t + al + gp:t + al + gp + bl]) == p, ("representative instance", i)
    assert per_query_ids(nq, [(q, idof[p]) for q, *_, p in inst]) == [list(x) for x in qone], "per-query one-gap ids"
    has = [int(s["sa_start"]) != -1 for s in s1]
    inst2 = enumerate_twogap(qoff, qtok, want_lm, inst, idof, has)
    pats2, idof2 = twogap_ids(inst2)
    assert len(inst2) == g2_count and len(pats2) == len(s2), ("two-gap enumeration", len(inst2), g2_count, len(pats2), len(s2))
    for i, (one, cc) in enumerate(pats2):
        assert int(s2[i]["blockid"]) == one and (int(c2d[i]),) == cc and int(qtok[int(s2[i]["gap2"])]) == cc[0], ("two-gap pattern id", i)
    assert per_query_ids(nq, [(q, idof2[(one, cc)]) for q, one, sc, cc in inst2]) == [list(x) for x in qtwo], "per-query two-gap ids"
    return len(pats), len(pats2)


def hit_lists(s1, hits1, pidx, phit_start, phit_len):
    """pattern id -> [(start, span - 1)] as the extraction stage sees them: a marker record stands for its frequent-pair list"""
    pidx = np.asarray(pidx, np.int64).reshape(-1, 2); out = {}
    for pid in range(len(s1)):
        a = int(s1[pid]["sa_start"])
        if a == -1:
            continue
        sl = hits1[a:int(s1[pid]["sa_end"]) + 1]
        if len(sl) == 1 and int(sl[0]["length"]) == 0:
            lo, hi = pidx[int(sl[0]["str_position"])]
            out[pid] = [(int(phit_start[k]), int(phit_len[k])) for k in range(lo, hi + 1)]
        else:
            out[pid] = [(int(x["str_position"]), int(x["length"])) for x in sl]
    return out


def check_extension_rules(c, sa, blocks, s1, hits_of, r1, sep1, r2, sep2a, sep2b):
    """Xab / abX (r1[:sep1]), XabX (r2[:sep2a]) and XaXb / aXbX (r2[sep2b:]) against the definitions."""
    w1, w2 = block_rules(c, sa, blocks)
    assert sorted(_t1(r) for r in r1[:sep1]) == w1, "Xab / abX rules"
    assert sorted(_t2(r) for r in r2[:sep2a]) == w2, "XabX rules"
    assert sorted(_t2(r) for r in r2[sep2b:]) == outer_rules(c, hits_of, s1, len(s1)), "XaXb / aXbX rules"
    return len(w1) + len(w2) + len(r2) - sep2b


def check_maxlex_lines(c, tstr, lexk, lexv, lines, G, D1, D2, blocks, p1d, c2d, one2):
    """(fe, ef) of every device lexicon line (cgx_lexent arrays for kind 0, 1, 2) against the float32 definition, bit for bit."""
    tab = LexTable(lexk, lexv); n = 0
    for kind, arr in lines.items():
        for e in arr:
            gaps = [] if kind == 0 else [(int(e["gap1"]), int(e["gap1_1"]))] if kind == 1 else [(int(e["gap1"]), int(e["gap1_1"])), (int(e["gap2"]), int(e["gap2_1"]))]
            src = lexline_source(kind, int(e["id"]), G, D1, D2, blocks, c, p1d, c2d, one2)
            fe, ef = maxlex(tab, tstr, src, int(e["tstart"]), int(e["end"]), gaps)
            assert fe.tobytes() == np.float32(e["fe"]).tobytes() and ef.tobytes() == np.float32(e["ef"]).tobytes(), ("MaxLex", kind, int(e["id"]), fe, e["fe"], ef, e["ef"])
            n += 1
    return n
