"""cgx_amd -- ctypes binding of libcgx_hip.so (the MI355X-native grammar extractor).

The package is deliberately thin: the product is the C ABI in ``include/cgx.h`` (host C +
hand-written HIP for gfx950).  Python only loads the library, converts numpy arrays to
pointers and mirrors the reference's stage names (suffixArraySearch ->
:meth:`Extractor.sa_lookup` / :meth:`Extractor.gappy_search`, ExtractPairs_Large_Data_Gappy
-> :meth:`Extractor.extract` ...).  There is no CPU fallback: importing works anywhere, but
creating an :class:`Extractor` without the built library or without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGX_LIB") or os.path.join(_HERE, "libcgx_hip.so")   # CGX_LIB: another build of the same ABI (A/B measurements of kernel variants)

# numpy views of the wire structs in include/cgx.h
GAPPY = np.dtype([("qrystart", "<i4"), ("a_len", "u1"), ("b_len", "u1"), ("gap", "u1")])
GAPPAT = np.dtype([("pat", "<i4", (5,)), ("number", "u1")])
TWOGAPPY = np.dtype([("blockid", "<u4"), ("gap2", "<u4"), ("c_len", "u1")])
HIT1 = np.dtype([("position", "<u4"), ("str_position", "<u4"), ("length", "u1")])
HIT2 = np.dtype([("position", "<u4"), ("str_position", "<u4"), ("length", "u1"), ("length2", "u1")])
RULE1 = np.dtype([("id", "<i4"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1")])
RULE2 = np.dtype([("id", "<i4"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1"), ("gap2", "u1"), ("gap2_1", "u1")])
RULE0 = np.dtype([("tar_start", "<i4"), ("block", "<i4"), ("tar_end", "u1")])
LEXTASK = np.dtype([("lexid", "<u4"), ("src", "<i4", (5,)), ("nsrc", "u1"), ("tstart", "<u4"), ("end", "u1"),
                    ("gap1", "u1"), ("gap1_1", "u1"), ("gap2", "u1"), ("gap2_1", "u1")])
GAPSEARCH = np.dtype([("qrystart", "<i4"), ("a_len", "<i4"), ("b_len", "<i4"), ("gap", "<i4"), ("position", "<u4"),
                      ("sa_start", "<i4"), ("sa_end", "<i4"), ("marker", "<i4")])
TWOGAPSEARCH = np.dtype([("blockid", "<u4"), ("gap2", "<u4"), ("c_len", "<i4"), ("position", "<u4"), ("sa_start", "<i4"), ("sa_end", "<i4")])
LEXENT = np.dtype([("id", "<i4"), ("tstart", "<u4"), ("end", "u1"), ("gap1", "u1"), ("gap1_1", "u1"), ("gap2", "u1"), ("gap2_1", "u1"), ("kind", "u1"),
                   ("f", "<u2"), ("fsample", "<u2"), ("paircount", "<u2"), ("pad", "<u2"), ("pad2", "<u2"), ("fe", "<f4"), ("ef", "<f4")])
BLOCK = np.dtype([("start", "<i4"), ("end", "<i4"), ("matchlen", "<i4"), ("string_start", "<i4")])
LEXKEY = np.dtype([("src", "<i4"), ("tgt", "<i4")])
LEXVAL = np.dtype([("v1", "<f4"), ("v2", "<f4")])

FETCH_DTYPES = {
    "sa": np.int32, "tokstart": np.int32, "freq": np.int32, "pidx": np.uint32, "miss": np.int32, "phit_start": np.uint32,
    "phit_len": np.uint8, "lm": np.int32, "up": np.int32, "down": np.int32, "g1": GAPPY, "p1": GAPPAT, "pid1": np.uint32,
    "s1": GAPSEARCH, "hits1": HIT1, "g2": TWOGAPPY, "c2": np.int32, "pid2": np.uint32, "s2": TWOGAPSEARCH, "hits2": HIT2,
    "r0": RULE0, "r1": RULE1, "r2": RULE2, "counts": np.uint32, "p1d": GAPPAT, "c2d": np.int32, "one2": np.uint32,
    "lex0": LEXENT, "lex1": LEXENT, "lex2": LEXENT, "rng0": np.int32, "rng1": np.int32, "rng2": np.int32,
    "blocks": BLOCK, "qb_off": np.uint32, "qb_ids": np.uint32, "qo_off": np.uint32, "qo_ids": np.uint32, "qt_off": np.uint32, "qt_ids": np.uint32,
}
COUNT_NAMES = ["e1", "d1", "h1", "e2", "d2", "h2", "g", "n0", "n1", "n2", "sep1", "sep2a", "sep2b", "nphits", "guard_exits", "last"]

# every entry point declared in include/cgx.h
ABI = [
    "cgx_create", "cgx_destroy", "cgx_last_error", "cgx_set_option", "cgx_upload_index", "cgx_build_sa", "cgx_precompute",
    "cgx_index_shape", "cgx_index_alloc", "cgx_share_index", "cgx_index_nbuffers", "cgx_index_buffer", "cgx_index_d2d", "cgx_index_finalize", "cgx_index_save", "cgx_index_load", "cgx_broadcast_index",
    "cgx_upload_queries", "cgx_sa_lookup", "cgx_gappy_search", "cgx_make_blocks", "cgx_set_blocks", "cgx_extract", "cgx_lexicon", "cgx_lex_features", "cgx_fetch",
    "cgx_stage_ms", "cgx_corpus_load", "cgx_corpus_free", "cgx_corpus_checksum", "cgx_corpus_save", "cgx_corpus_load_cache", "cgx_corpus_matches_sources", "cgx_corpus_upload", "cgx_extract_grammars", "cgx_extract_grammars_shard", "cgx_shard_bounds", "cgx_extract_grammars_ids",
    "cgx_corpus_from_ids", "cgx_host_ms", "cgx_flush", "cgx_fetch_pinned", "cgx_pinned_next_batch", "cgx_upload_vocab", "cgx_upload_score_tables", "cgx_set_query_blocks",
    "cgx_format", "cgx_text_info", "cgx_text_encoding", "cgx_assemble_files_enc", "cgx_text_segments", "cgx_text_segments_begin", "cgx_text_offsets", "cgx_text_read", "cgx_text_read_begin", "cgx_text_read_wait", "cgx_pinned_alloc", "cgx_pinned_free", "cgx_assemble_files", "cgx_corpus_load_opt",
    "cgx_corpus_flags", "cgx_corpus_from_ids16", "cgx_text_trailers", "cgx_text_trailers_begin",
]


class CgxError(RuntimeError):
    pass


class IndexHost(C.Structure):
    _fields_ = [("str", C.c_void_p), ("n", C.c_uint32), ("rlp", C.c_void_p), ("tstr", C.c_void_p), ("nt", C.c_uint32),
                ("ltar", C.c_void_p), ("rtar", C.c_void_p), ("lexk", C.c_void_p), ("lexv", C.c_void_p), ("nlex", C.c_uint32),
                ("sa", C.c_void_p), ("ltar16", C.c_void_p), ("rtar16", C.c_void_p)]      # the last two: long-sentence mode only (None otherwise)


_lib = None


def load_library():
    """Load libcgx_hip.so; raises CgxError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CgxError("%s is missing: build it with `make -C cgx_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.cgx_create.restype = C.c_void_p; lib.cgx_create.argtypes = [C.c_int]
    lib.cgx_destroy.restype = None; lib.cgx_destroy.argtypes = [C.c_void_p]
    lib.cgx_last_error.restype = C.c_char_p; lib.cgx_last_error.argtypes = [C.c_void_p]
    lib.cgx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.cgx_upload_index.argtypes = [C.c_void_p, C.POINTER(IndexHost)]
    for f in ("cgx_flush", "cgx_build_sa", "cgx_precompute", "cgx_sa_lookup", "cgx_make_blocks", "cgx_gappy_search", "cgx_extract", "cgx_lexicon", "cgx_index_finalize", "cgx_index_nbuffers"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.cgx_index_alloc.argtypes = [C.c_void_p, C.c_void_p]
    lib.cgx_index_shape.argtypes = [C.c_void_p, C.c_void_p]
    lib.cgx_index_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.cgx_index_d2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.cgx_broadcast_index.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.cgx_upload_queries.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    lib.cgx_set_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.cgx_lex_features.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.cgx_fetch.restype = C.c_int64; lib.cgx_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.cgx_stage_ms.restype = C.c_double; lib.cgx_stage_ms.argtypes = [C.c_void_p, C.c_char_p]
    lib.cgx_host_ms.restype = C.c_double; lib.cgx_host_ms.argtypes = [C.c_void_p, C.c_char_p]
    lib.cgx_corpus_load.restype = C.c_void_p; lib.cgx_corpus_load.argtypes = [C.c_char_p] * 4 + [C.c_char_p, C.c_size_t]
    lib.cgx_corpus_free.restype = None; lib.cgx_corpus_free.argtypes = [C.c_void_p]
    lib.cgx_corpus_checksum.restype = C.c_uint64; lib.cgx_corpus_checksum.argtypes = [C.c_void_p]
    lib.cgx_corpus_save.argtypes = [C.c_void_p, C.c_char_p]
    lib.cgx_corpus_load_cache.restype = C.c_void_p; lib.cgx_corpus_load_cache.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.cgx_corpus_upload.argtypes = [C.c_void_p, C.c_void_p]
    lib.cgx_extract_grammars.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
    lib.cgx_extract_grammars_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_uint64)]
    lib.cgx_corpus_from_ids.restype = C.c_void_p
    lib.cgx_corpus_from_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.c_uint32]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Corpus:
    """Host-side corpus (tokens, alignment, lexical table): the reference's ref_set_t."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def load(cls, src, tgt, align, lex, long_sentences=False):
        lib = load_library()
        err = C.create_string_buffer(512)
        lib.cgx_corpus_load_opt.restype = C.c_void_p; lib.cgx_corpus_load_opt.argtypes = [C.c_char_p] * 4 + [C.c_int, C.c_char_p, C.c_size_t]
        h = lib.cgx_corpus_load_opt(src.encode(), tgt.encode(), align.encode(), lex.encode(), 1 if long_sentences else 0, err, 512)
        if not h:
            raise CgxError(err.value.decode(errors="replace"))
        return cls(h)

    @classmethod
    def from_ids(cls, str_, sentind, tstr, tsentind, lsrc, rsrc, ltar, rtar, lexk, lexv):
        lib = load_library()
        a = [_c(str_, np.int32), _c(sentind, np.int32), _c(tstr, np.int32), _c(tsentind, np.int32), _c(lsrc, np.uint8), _c(rsrc, np.uint8),
             _c(ltar, np.uint8), _c(rtar, np.uint8), _c(lexk, LEXKEY), _c(lexv, LEXVAL)]
        h = lib.cgx_corpus_from_ids(_ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]) - 1, _ptr(a[2]), len(a[2]), _ptr(a[3]),
                                    _ptr(a[4]), _ptr(a[5]), _ptr(a[6]), _ptr(a[7]), _ptr(a[8]), _ptr(a[9]), len(a[8]))
        if not h:
            raise CgxError("cgx_corpus_from_ids failed")
        return cls(h)

    @classmethod
    def from_ids16(cls, str_, sentind, tstr, tsentind, lsrc, rsrc, ltar, rtar, lexk, lexv):
        """from_ids with 16-bit alignment positions (0xFFFF = not aligned): long-sentence mode (cgx_corpus_from_ids16)."""
        lib = load_library()
        a = [_c(str_, np.int32), _c(sentind, np.int32), _c(tstr, np.int32), _c(tsentind, np.int32), _c(lsrc, np.uint16), _c(rsrc, np.uint16),
             _c(ltar, np.uint16), _c(rtar, np.uint16), _c(lexk, LEXKEY), _c(lexv, LEXVAL)]
        lib.cgx_corpus_from_ids16.restype = C.c_void_p
        lib.cgx_corpus_from_ids16.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.c_uint32]
        h = lib.cgx_corpus_from_ids16(_ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]) - 1, _ptr(a[2]), len(a[2]), _ptr(a[3]),
                                      _ptr(a[4]), _ptr(a[5]), _ptr(a[6]), _ptr(a[7]), _ptr(a[8]), _ptr(a[9]), len(a[8]))
        if not h:
            raise CgxError("cgx_corpus_from_ids16 failed (a position or a sentence beyond the long-sentence limits)")
        return cls(h)

    def flags(self):
        lib = load_library(); lib.cgx_corpus_flags.argtypes = [C.c_void_p]
        return int(lib.cgx_corpus_flags(self.h))

    def save(self, path):
        rc = load_library().cgx_corpus_save(self.h, path.encode())
        if rc != 0:
            raise CgxError("cgx_corpus_save failed (%d)" % rc)

    @classmethod
    def load_cache(cls, path):
        lib = load_library(); err = C.create_string_buffer(512)
        h = lib.cgx_corpus_load_cache(path.encode(), err, 512)
        if not h:
            raise CgxError(err.value.decode(errors="replace"))
        return cls(h)

    def checksum(self):
        return int(load_library().cgx_corpus_checksum(self.h))

    def close(self):
        if self.h:
            load_library().cgx_corpus_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Extractor:
    """One GPU context (one per process / rank).  Mirrors the reference's `void *ctx`."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = self.lib.cgx_create(device)
        if not self.h:
            raise CgxError("cgx_create(%d) failed: no HIP device (the extractor has no CPU fallback)" % device)

    def _chk(self, rc, what):
        if rc != 0:
            raise CgxError("%s failed (%d): %s" % (what, rc, self.lib.cgx_last_error(self.h).decode(errors="replace")))

    def close(self):
        if self.h:
            self.lib.cgx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        self._chk(self.lib.cgx_set_option(self.h, name.encode(), int(value)), "cgx_set_option")

    # ---- index ----
    def upload_index(self, str_, rlp, tstr, ltar, rtar, lexk, lexv, sa=None):
        a = dict(str=_c(str_, np.int32), rlp=_c(rlp, np.uint32), tstr=_c(tstr, np.int32), ltar=_c(ltar, np.uint8), rtar=_c(rtar, np.uint8),
                 lexk=_c(lexk, LEXKEY), lexv=_c(lexv, LEXVAL), sa=None if sa is None else _c(sa, np.int32))
        ix = IndexHost(_ptr(a["str"]), len(a["str"]), _ptr(a["rlp"]), _ptr(a["tstr"]), len(a["tstr"]), _ptr(a["ltar"]), _ptr(a["rtar"]),
                       _ptr(a["lexk"]), _ptr(a["lexv"]), len(a["lexk"]), _ptr(a["sa"]))
        self._chk(self.lib.cgx_upload_index(self.h, C.byref(ix)), "cgx_upload_index")

    def upload_corpus(self, corpus):
        self._chk(self.lib.cgx_corpus_upload(self.h, corpus.h), "cgx_corpus_upload")

    def build_sa(self):
        self._chk(self.lib.cgx_build_sa(self.h), "cgx_build_sa")

    def precompute(self):
        self._chk(self.lib.cgx_precompute(self.h), "cgx_precompute")

    def index_buffers(self):
        out = []
        for i in range(self.lib.cgx_index_nbuffers(self.h)):
            name = C.c_char_p(); nb = C.c_uint64()
            self._chk(self.lib.cgx_index_buffer(self.h, i, C.byref(name), C.byref(nb)), "cgx_index_buffer")
            out.append((name.value.decode(), int(nb.value)))
        return out

    def index_shape(self):
        d = np.zeros(8, np.int32)
        self._chk(self.lib.cgx_index_shape(self.h, _ptr(d)), "cgx_index_shape")
        return [int(x) for x in d]

    def index_alloc(self, dims):
        d = np.asarray(dims, np.int32)
        self._chk(self.lib.cgx_index_alloc(self.h, _ptr(d)), "cgx_index_alloc")

    def share_index(self, other):
        """This context borrows the index of `other` (same device; nothing is copied): two contexts, driven from two threads, keep two
        batches in flight on one card.  `other` must stay alive and must not rebuild its index while this one uses it."""
        self.lib.cgx_share_index.restype = C.c_int; self.lib.cgx_share_index.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.lib.cgx_share_index(self.h, other.h), "cgx_share_index")

    def index_d2d(self, i, dptr, direction):
        self._chk(self.lib.cgx_index_d2d(self.h, i, C.c_void_p(dptr), direction), "cgx_index_d2d")

    def index_save(self, path, corpus_checksum):
        self.lib.cgx_index_save.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        self._chk(self.lib.cgx_index_save(self.h, path.encode(), corpus_checksum), "cgx_index_save")

    def index_load(self, path, corpus_checksum):
        self.lib.cgx_index_load.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        self._chk(self.lib.cgx_index_load(self.h, path.encode(), corpus_checksum), "cgx_index_load")

    def index_finalize(self):
        self._chk(self.lib.cgx_index_finalize(self.h), "cgx_index_finalize")

    # ---- batch stages ----
    def upload_queries(self, qoff, qtok):
        qoff = _c(qoff, np.int32); qtok = _c(qtok, np.int32)
        self._chk(self.lib.cgx_upload_queries(self.h, _ptr(qoff), len(qoff), _ptr(qtok), len(qtok)), "cgx_upload_queries")

    def sa_lookup(self):
        self._chk(self.lib.cgx_sa_lookup(self.h), "cgx_sa_lookup")

    def gappy_search(self):
        self._chk(self.lib.cgx_gappy_search(self.h), "cgx_gappy_search")

    def make_blocks(self):
        self._chk(self.lib.cgx_make_blocks(self.h), "cgx_make_blocks")

    def set_blocks(self, blocks):
        blocks = _c(blocks, BLOCK)
        self._chk(self.lib.cgx_set_blocks(self.h, _ptr(blocks), len(blocks)), "cgx_set_blocks")
        return blocks

    def extract(self):
        self._chk(self.lib.cgx_extract(self.h), "cgx_extract")

    def lexicon(self):
        self._chk(self.lib.cgx_lexicon(self.h), "cgx_lexicon")

    def lex_features(self, tasks, n_onegap, n_twogap):
        tasks = _c(tasks, LEXTASK)
        fe = np.zeros(len(tasks), np.float32); ef = np.zeros(len(tasks), np.float32)
        self._chk(self.lib.cgx_lex_features(self.h, _ptr(tasks), len(tasks), n_onegap, n_twogap, _ptr(fe), _ptr(ef)), "cgx_lex_features")
        return fe, ef

    # ---- grammar text on the device (what cgx_extract_grammars* does internally) ----
    def upload_vocab(self, swords, twords):
        """swords / twords: spelling (bytes) of every source / target id, b"" for ids without one"""
        def pool(words):
            off = np.zeros(len(words) + 1, np.uint32); off[1:] = np.cumsum([len(w) for w in words])
            return b"".join(words), off
        sp, so = pool(swords); tp, to = pool(twords)
        self.lib.cgx_upload_vocab.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32]
        self._chk(self.lib.cgx_upload_vocab(self.h, sp, _ptr(so), len(swords), tp, _ptr(to), len(twords)), "cgx_upload_vocab")

    def upload_score_tables(self, aa, bb, fs):
        a = [_c(aa, np.float32), _c(bb, np.float32), _c(fs, np.float32)]
        self.lib.cgx_upload_score_tables.argtypes = [C.c_void_p] * 4
        self._chk(self.lib.cgx_upload_score_tables(self.h, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])), "cgx_upload_score_tables")

    def format(self):
        """-> (bytes of all grammar files together, rule lines, text slot)"""
        nb = C.c_uint64(); nl = C.c_uint64(); slot = C.c_int()
        self.lib.cgx_format.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        self._chk(self.lib.cgx_format(self.h, C.byref(nb), C.byref(nl), C.byref(slot)), "cgx_format")
        return int(nb.value), int(nl.value), int(slot.value)

    def count_rules(self):
        """cgx_format without a text slot: builds the per-query pattern lists ("qo_*", "qt_*") and counts the rule lines."""
        nl = C.c_uint64()
        self.lib.cgx_format.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        self._chk(self.lib.cgx_format(self.h, None, C.byref(nl), None), "cgx_format")
        return int(nl.value)

    def text(self, slot, nq):
        """The unique text of the slot and the piece lists: (text bytes, qseg u64[nq+1], seg_off u64[nseg], seg_len u32[nseg], qtext u64[nq+1])."""
        ub = C.c_uint64(); ns = C.c_uint64(); fb = C.c_uint64()
        self.lib.cgx_text_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_uint64)] * 3
        self._chk(self.lib.cgx_text_info(self.h, slot, C.byref(ub), C.byref(ns), C.byref(fb)), "cgx_text_info")
        qseg = np.zeros(nq + 1, np.uint64); so = np.zeros(max(int(ns.value), 1), np.uint64); sl = np.zeros(max(int(ns.value), 1), np.uint32); qtext = np.zeros(nq + 1, np.uint64)
        self.lib.cgx_text_segments.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.cgx_text_segments(self.h, slot, _ptr(qseg), _ptr(so), _ptr(sl)), "cgx_text_segments")
        self.lib.cgx_text_offsets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.lib.cgx_text_offsets(self.h, slot, _ptr(qtext)), "cgx_text_offsets")
        self.lib.cgx_pinned_alloc.restype = C.c_void_p; self.lib.cgx_pinned_alloc.argtypes = [C.c_size_t]; self.lib.cgx_pinned_free.argtypes = [C.c_void_p]
        n = int(ub.value); buf = self.lib.cgx_pinned_alloc(n + 64)
        if not buf:
            raise CgxError("cgx_pinned_alloc failed")
        self.lib.cgx_text_read.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        try:
            self._chk(self.lib.cgx_text_read(self.h, slot, 0, n, buf, 0), "cgx_text_read")
            text = bytes((C.c_char * n).from_address(buf)) if n else b""       # not C.string_at: its size argument is a C int (a unique text of 5 GB came back 1.6 GB long)
        finally:
            self.lib.cgx_pinned_free(buf)
        return text, qseg, so[:int(ns.value)], sl[:int(ns.value)], qtext

    def text_trailers(self, slot, nq):
        """CRC-32 and ISIZE of every file of a slot of deflate pieces: u32[nq, 2]."""
        trl = np.zeros((max(nq, 1), 2), np.uint32)
        self.lib.cgx_text_trailers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.lib.cgx_text_trailers(self.h, slot, _ptr(trl)), "cgx_text_trailers")
        return trl[:nq]

    def text_encoding(self, slot):
        """0: the slot holds plain text; 1: deflate blocks, one per emission group, every piece a byte-aligned stretch of a deflate
        stream (option gz_level with gz_device): a file is the gzip header, its pieces, 03 00, CRC-32, ISIZE (text_trailers)."""
        self.lib.cgx_text_encoding.argtypes = [C.c_void_p, C.c_int]
        rc = self.lib.cgx_text_encoding(self.h, slot)
        if rc < 0:
            self._chk(rc, "cgx_text_encoding")
        return rc

    def fetch(self, name):
        nb = self.lib.cgx_fetch(self.h, name.encode(), None, 0)
        if nb < 0:
            self._chk(int(nb), "cgx_fetch(%s)" % name)
        dt = np.dtype(FETCH_DTYPES[name])
        out = np.zeros(nb // dt.itemsize, dt)
        got = self.lib.cgx_fetch(self.h, name.encode(), _ptr(out), nb)
        if got < 0:
            self._chk(int(got), "cgx_fetch(%s)" % name)
        return out

    def counts(self):
        return dict(zip(COUNT_NAMES, (int(x) for x in self.fetch("counts"))))

    def stage_ms(self, name):
        return float(self.lib.cgx_stage_ms(self.h, name.encode()))

    def host_ms(self, name):
        return float(self.lib.cgx_host_ms(self.h, name.encode()))

    def flush(self):
        self._chk(self.lib.cgx_flush(self.h), "cgx_flush")

    def run_sort(self, major, key, val=None, keybits=64):
        """Test hook: run_sort (cgx_device.hip) on host arrays -> (keys, payload or None, long runs the fix pass met).
        keybits: the promise that every key is below 2^keybits or all ones (<= 53: the ranking loop compares position-tagged keys)."""
        major = _c(major, np.uint32); key = _c(key, np.uint64); val = None if val is None else _c(val, np.uint32)
        ko = np.zeros(len(key), np.uint64); vo = None if val is None else np.zeros(len(key), np.uint32)
        self.lib.cgx__test_run_sort_bits.restype = C.c_int64
        self.lib.cgx__test_run_sort_bits.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int]
        rc = self.lib.cgx__test_run_sort_bits(self.h, _ptr(major), _ptr(key), _ptr(val), _ptr(ko), _ptr(vo), len(key), keybits)
        if rc < 0:
            self._chk(int(rc), "cgx__test_run_sort")
        return ko, vo, int(rc)

    # ---- whole path ----
    def extract_grammars(self, corpus, qryfile, outdir, q_begin=0, q_end=-1):
        n = C.c_uint64()
        self._chk(self.lib.cgx_extract_grammars(self.h, corpus.h, qryfile.encode(), outdir.encode() if outdir else None, q_begin, q_end, C.byref(n)), "cgx_extract_grammars")
        return int(n.value)

    def extract_grammars_ids(self, corpus, qoff, qtok, outdir=None, first=0):
        qoff = _c(qoff, np.int32); qtok = _c(qtok, np.int32); n = C.c_uint64()
        self._chk(self.lib.cgx_extract_grammars_ids(self.h, corpus.h, _ptr(qoff), len(qoff), _ptr(qtok), len(qtok),
                                                    outdir.encode() if outdir else None, first, C.byref(n)), "cgx_extract_grammars_ids")
        return int(n.value)
