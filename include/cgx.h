/*
 * cgx.h -- C ABI of the MI355X-native hierarchical grammar extractor (libcgx_hip.so).
 *
 * Drop-in boundary for the reference's  suffix-array lookup -> gappy-phrase search ->
 * rule extraction  path.  Plain pointers and sizes only; every entry point returns 0 on
 * success and a negative code on failure, with the text available from cgx_last_error().
 * There is NO CPU fallback: without a HIP device cgx_create() fails.
 *
 * Reference interfaces replaced (all C++ linkage, results passed through qry_set_t fields):
 *   suffixArraySearchInit / suffixArraySearchFinalize[_One]   SuffixArray.h:9-22, SuffixArray.cu:769-813
 *   suffixArraySearch                                         SuffixArray.h:11-16, SuffixArray.cu:1342-2269
 *   ExtractPairs_Large_Data_Gappy / extractPairFinalize       ExtractPair.h:179-185,33, ExtractPair.cu:3215-4001
 *   initWordPossibilityIntKey (device part: table sort)       ExtractPair.h:20-21, ExtractPair.cu:2528-2537
 *   suffixArrayConstruct                                      SuffixArray.h:7, SuffixArray.c:196-242
 */
#ifndef CGX_H
#define CGX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGX_OK 0
#define CGX_ERR_ARG (-1)
#define CGX_ERR_HIP (-2)
#define CGX_ERR_STATE (-3)
#define CGX_ERR_NOMEM (-4)
#define CGX_ERR_IO (-5)
#define CGX_ERR_ALIGN_RANGE (-6)  /* alignment index >= 255: the reference prints "Not possible, too long sentence" and exits 1 (ExtractPair.cu:2683) */
#define CGX_ERR_ALIGN_PAIR (-7)   /* dangling "i-" without j: "Not possible!" and exit 0 (ExtractPair.cu:2676) */

typedef struct cgx_ctx cgx_ctx;

/* ---- wire records handed back to the host side (packed, little endian) ---- */
#pragma pack(push, 1)
typedef struct { int32_t qrystart; uint8_t a_len, b_len, gap; } cgx_gappy;            /* one query instance of aXb   (ComTypes.h:142) */
typedef struct { int32_t pat[5]; uint8_t number; } cgx_gappat;                        /* its symbols, gap = -1, pad = -2 (:194)       */
typedef struct { uint32_t blockid, gap2; uint8_t c_len; } cgx_twogappy;               /* one query instance of aXbXc (:151)           */
typedef struct { uint32_t position, str_position; uint8_t length; } cgx_hit1;         /* corpus occurrence of aXb    (:179)           */
typedef struct { uint32_t position, str_position; uint8_t length, length2; } cgx_hit2;/* corpus occurrence of aXbXc  (:186)           */
typedef struct { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1; } cgx_rule1; /* one-gap rule                (:224)           */
typedef struct { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1, gap2, gap2_1; } cgx_rule2; /* two-gap rule  (:233)           */
typedef struct { int32_t tar_start; int32_t block; uint8_t tar_end; } cgx_rule0;      /* contiguous rule             (:349)           */
typedef struct { uint32_t lexid; int32_t src[5]; uint8_t nsrc; uint32_t tstart;
                 uint8_t end, gap1, gap1_1, gap2, gap2_1; } cgx_lextask;              /* MaxLex work item            (:376)           */
#pragma pack(pop)
/* one lexicon line (red_dup_t, ComTypes.h:244) as produced on the device: converted id, the first rule's target span and gaps,
 * group size f, sample base, pair count and the two MaxLex features; kind 0 contiguous, 1 one gap, 2 two gaps */
typedef struct { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1, gap2, gap2_1, kind; uint16_t f, fsample, paircount; uint16_t pad; float fe, ef; } cgx_lexent;
typedef struct { int32_t qrystart; int32_t a_len, b_len, gap; uint32_t position; int32_t sa_start, sa_end; int32_t marker; } cgx_gapsearch; /* distinct aXb   (:168) */
typedef struct { uint32_t blockid, gap2; int32_t c_len; uint32_t position; int32_t sa_start, sa_end; } cgx_twogapsearch;                  /* distinct aXbXc (:158) */
typedef struct { int32_t start, end, matchlen, string_start; } cgx_block;            /* distinct contiguous phrase (saind_t, :342)    */
typedef struct { int32_t src, tgt; } cgx_lexkey;                                      /* (:355) -1 = NULL word */
typedef struct { float v1, v2; } cgx_lexval;                                          /* (:360) */

/* ---- corpus-side index, host view used by cgx_upload_index ---- */
typedef struct {
    const int32_t *str;      /* source token ids, n entries: words >= 2, 1 closes a sentence, trailing "1,last" (Start.cu:306-327) */
    uint32_t n;
    const uint32_t *rlp;     /* per source token (L<<24)|(R<<16)|(P<<8); delimiter slots hold the next sentence's target offset (ExtractPair.cu:2717-2731) */
    const int32_t *tstr;     /* target token ids */
    uint32_t nt;
    const uint8_t *ltar, *rtar; /* per target token min/max aligned source position, 255 = unaligned */
    const cgx_lexkey *lexk;  /* lexical table rows in file order (no duplicates) */
    const cgx_lexval *lexv;
    uint32_t nlex;
    const int32_t *sa;       /* optional prebuilt suffix array (NULL: built on the device) */
    /* Long-sentence mode (opt-in; SURVEY 8(f4): the reference rejects sentences of 255 tokens and more, ExtractPair.cu:2683).
     * Both NULL = the reference's byte positions.  Both set: per target token min/max aligned source position as 16-bit values
     * (0xFFFF = unaligned; ltar / rtar are then ignored), and every rlp word carries three more bits for L, three for R and two
     * for P in its low byte, L and R as position codes -- cgx_rlp_pack() of csrc/cgx_rules.h; cgx_corpus_load_opt builds them. */
    const uint16_t *ltar16, *rtar16;
} cgx_index_host;

/* ---- lifetime ---- */
cgx_ctx *cgx_create(int device);                       /* replaces suffixArraySearchInit (SuffixArray.cu:769) */
void cgx_destroy(cgx_ctx *ctx);                        /* replaces suffixArraySearchFinalize*, extractPairFinalize */
const char *cgx_last_error(cgx_ctx *ctx);
int cgx_set_option(cgx_ctx *ctx, const char *name, int64_t value); /* "k1_limit" (default 128: query tokens per sentence that are looked up, the reference's K1 launch width; 0 = no limit), "chunk_items", "force_host_lexicon", "async_write", "prealloc_text" (default 0; 1 = a caller that will submit several batches with async_write asks for both text buffers to be allocated at the first one), "use_bigrams" (default 1), "use_lex_hash" (default 1), "device_format" (default 1), "gz_level" (0 = plain grammar.<q>.s; 1..9 = grammar.<q>.s.gz) and "gz_device" (default 1: with the device formatter the gzip members are produced on the GPU and gz_level only switches them on; 0 = the host's zlib compresses the plain text at gz_level), "sub_batch" (queries per internal batch of cgx_extract_grammars*; 0 = automatic: at most "auto_batch_tokens" (default 300000) query tokens per internal batch; with async_write the writer of one sub-batch overlaps the GPU work of the next), "write_period" / "write_count" (with a period P > 0 only the files of queries g with g % P < count are written, g = index in the whole query list; rules are counted for all, and an internal batch without such a query lays out no text: full-count runs that sample their output), "src_blocks" (default 1; 0 = the lookups locate a sentence's target-side alignment bytes through the delimiter's alignment word instead of the source-addressed copy of the blocks: A/B and test hook), "hit_order" (default 0: the hit lists fetched with cgx_fetch are grouped by pattern and ordered by position bucket only, which is all extraction needs; 1 = sorted completely on the card), "gz_dynamic" (default 1: Huffman codes made per batch for the device-made .gz output; 0 = the fixed codes of RFC 1951), "lex_flat" (default 1: MaxLex as one task list per wave; 0 = one lane per lexicon line), "lex_bits" (default 1: presence bits asked before the pair keys of the lexical table), "win_table" (default 0; 1, set before the index is built or loaded = one aligned 128-byte window row per corpus position, 128 N bytes: measured, buys nothing) and "tile_order" (default 0) -- A/B switches, same results either way; "wide_hits2", "look_rec_cap", "pool_cap", "fault_inject", "append_slack", "append_guess_milli" (test hooks for the lookup output sizing) */

/* ---- index: upload, device suffix-array construction, frequent-pair precomputation ---- */
int cgx_upload_index(cgx_ctx *ctx, const cgx_index_host *ix);   /* H2D of the index (SuffixArray.cu:1396-1412, ExtractPair.cu:3279-3282) */
int cgx_build_sa(cgx_ctx *ctx);                                 /* replaces suffixArrayConstruct (SuffixArray.c:196-242) on the device */
int cgx_precompute(cgx_ctx *ctx);                               /* replaces preComputation + precomp kernel (SuffixArray.cu:1132-1340) */
/* multi-GPU: allocate an empty replica of given sizes, then move buffers device-to-device
 * (dir 0: index buffer -> dptr, dir 1: dptr -> index buffer) around a collective broadcast. */
typedef struct { uint32_t n, nt, nlex, nphits; int32_t last; uint32_t lex_nrow, lex_ntgt, reserved; } cgx_index_dims;   /* derived tables (pair hash, l-gram tables) are rebuilt by cgx_index_finalize, not shipped */
int cgx_index_shape(cgx_ctx *ctx, cgx_index_dims *dims);        /* sizes of a built index (root rank) */
int cgx_index_alloc(cgx_ctx *ctx, const cgx_index_dims *dims);  /* empty replica of the same sizes (other ranks) */
/* A second context of the SAME device over the same index: dst borrows every index array of src (nothing is copied; dst never frees them; src must
 * outlive dst's use of them, and neither may rebuild the index meanwhile).  Two contexts driven from two host threads keep two batches in flight
 * on one card: the kernels of a batch wait most of their wave cycles, and the card takes the other batch's waves in between (measured: 1.2x). */
int cgx_share_index(cgx_ctx *dst, const cgx_ctx *src);
int cgx_index_nbuffers(cgx_ctx *ctx);
int cgx_index_buffer(cgx_ctx *ctx, int i, const char **name, uint64_t *nbytes);
int cgx_index_d2d(cgx_ctx *ctx, int i, void *dptr, int dir);
int cgx_index_finalize(cgx_ctx *ctx);                           /* after the last d2d on a replica */
/* the built index as one file (what a replica receives; derived tables are rebuilt on load): a later start with the same corpus
 * (cgx_corpus_checksum) skips cgx_upload_index / cgx_build_sa / cgx_precompute.  The reference's own cache is commented out (SuffixArray.c:208-230).
 * cgx_index_load: CGX_ERR_IO if the file is missing / unreadable / fails its content check, CGX_ERR_STATE if it is of another version or corpus. */
int cgx_index_save(cgx_ctx *ctx, const char *path, uint64_t corpus_checksum);
int cgx_index_load(cgx_ctx *ctx, const char *path, uint64_t corpus_checksum);
/* one-time broadcast of every index buffer from rank `root` over an existing RCCL communicator */
int cgx_broadcast_index(cgx_ctx *ctx, void *nccl_comm, int root, int rank);

/* ---- per batch of query sentences ---- */
int cgx_upload_queries(cgx_ctx *ctx, const int32_t *qoff, int32_t nq, const int32_t *qtok, int32_t ntok); /* constructQryIndex output (Start.cu:50-132) */
int cgx_sa_lookup(cgx_ctx *ctx);        /* K1 + K2: SuffixArray.cu:402-767, 109-400 */
int cgx_gappy_search(cgx_ctx *ctx);     /* enumeration, sorts, lookups: SuffixArray.cu:1530-2256, GappyLook.cu:128-737 */
int cgx_make_blocks(cgx_ctx *ctx);      /* GenerateBlocks (ExtractPair.cu:2742-2903) on the device, after cgx_sa_lookup: distinct contiguous phrases in first-seen order ("blocks") and the per-query lists ("qb_off", "qb_ids") */
int cgx_set_blocks(cgx_ctx *ctx, cgx_block *blocks, uint32_t g);   /* the same, with a block list made by the caller's own GenerateBlocks; fills blocks[i].string_start = sa[start] */
int cgx_extract(cgx_ctx *ctx);          /* three extraction launches + sorts: ExtractPair.cu:3336-3670 */
int cgx_lexicon(cgx_ctx *ctx);          /* device lexicon + MaxLex: createLexicon*Fast (ExtractPair.c:515-1276) + lexicalTaskMaxEF; results "lex1" "lex2" "lex0" */
int cgx_lex_features(cgx_ctx *ctx, const cgx_lextask *tasks, uint32_t ntask, uint32_t n_onegap, uint32_t n_twogap,
                     float *max_fe, float *max_ef);             /* lexicalTaskMaxEF: ExtractPair.cu:2144-2432 */

/* ---- grammar text laid out on the device (replaces the fprintf loops of PrintResults.c:339-577) ----
 * vocabulary spellings as byte pools (word i = pool[off[i] .. off[i+1])), the host-libm score tables
 * (aa[302*302], CountEF[302], SampleCountF[302]; ExtractPair.c:652-656) and the per-query block lists. */
int cgx_upload_vocab(cgx_ctx *ctx, const char *spool, const uint32_t *soff, uint32_t ns, const char *tpool, const uint32_t *toff, uint32_t nt);
int cgx_upload_score_tables(cgx_ctx *ctx, const float *aa, const float *bb, const float *fs);
int cgx_set_query_blocks(cgx_ctx *ctx, const uint32_t *off, const uint32_t *ids);    /* CSR over the batch's queries */
int cgx_format(cgx_ctx *ctx, uint64_t *total_bytes, uint64_t *total_lines, int *slot); /* after cgx_lexicon; two text slots alternate; total_bytes = size of all grammar files together; slot == NULL: only count the rule lines (total_lines), lay out no text */
/* What a slot holds: the UNIQUE text of the batch (every lexicon line formatted once, `unique_bytes`), and the files as lists of
 * pieces of it: piece s = bytes [seg_off[s], seg_off[s] + seg_len[s]) of the unique text; the file of query q is the concatenation of
 * pieces qseg[q] .. qseg[q+1]-1 in that order (PrintResults.c:451-570 emission order) and is qtext[q+1] - qtext[q] bytes long. */
int cgx_text_info(cgx_ctx *ctx, int slot, uint64_t *unique_bytes, uint64_t *nseg, uint64_t *file_bytes);
/* With option "gz_level" > 0 (and "gz_device", default 1) the slot holds the same text as DEFLATE data made by the formatter itself (RFC 1951, fixed
 * Huffman codes, back-references taken from the line structure): every emission group is a block that ends on a byte, so a file is still the
 * concatenation of its pieces -- put between a gzip header (RFC 1952: 1f 8b 08 00 00 00 00 00 00 03) and the trailer 03 00 <CRC-32> <ISIZE>, whose
 * two values come from cgx_text_trailers.  grammar.<q>.s.gz is then one gzip member that zcat / gzread / Python's gzip read as the plain file
 * (the reference writes one fprintf per rule, PrintResults.c:407-577).  Offsets, lengths and sizes above are then compressed bytes
 * (file sizes include the 20 bytes of header and trailer).  Returns CGX_TEXT_PLAIN or CGX_TEXT_GZIP_PIECES, or < 0. */
#define CGX_TEXT_PLAIN 0
#define CGX_TEXT_GZIP_PIECES 1
int cgx_text_encoding(cgx_ctx *ctx, int slot);
int cgx_text_trailers(cgx_ctx *ctx, int slot, uint32_t *trl /* 2 * nq: CRC-32 and ISIZE of every file; CGX_TEXT_GZIP_PIECES slots only */);
int cgx_text_trailers_begin(cgx_ctx *ctx, int slot, uint32_t *trl, int reader);       /* the same copy, only enqueued on side stream `reader` */
int cgx_text_segments(cgx_ctx *ctx, int slot, uint64_t *qseg /* nq+1 */, uint64_t *seg_off /* nseg */, uint32_t *seg_len /* nseg */);
int cgx_text_segments_begin(cgx_ctx *ctx, int slot, uint64_t *qseg, uint64_t *seg_off, uint32_t *seg_len, int reader); /* the same copies, only enqueued on side stream `reader`; cgx_text_read_wait(reader) waits for them */
int cgx_text_offsets(cgx_ctx *ctx, int slot, uint64_t *qtext);                       /* nq+1 cumulative file sizes: file q has qtext[q+1] - qtext[q] bytes */
#define CGX_MAX_READERS 64
int cgx_text_read(cgx_ctx *ctx, int slot, uint64_t off, uint64_t bytes, void *dst, int reader); /* D2H of unique-text bytes on side stream `reader` (0..CGX_MAX_READERS-1), thread safe per reader */
int cgx_text_read_begin(cgx_ctx *ctx, int slot, uint64_t off, uint64_t bytes, void *dst, int reader); /* the same copy, only enqueued */
int cgx_text_read_wait(cgx_ctx *ctx, int reader);                                                      /* wait for everything enqueued on `reader` */
/* the file phase alone: grammar.<first+q>.s, q < nq, assembled in `outdir` by `nthreads` host threads from a unique text and piece lists held by the caller
 * (no GPU, no context needed; what the writer of cgx_extract_grammars* runs after the DMA).  *file_ms = mean busy time per thread */
int cgx_assemble_files(const char *utext, const uint64_t *qseg, const uint64_t *seg_off, const uint32_t *seg_len, int32_t nq, int32_t first,
                       const char *outdir, int nthreads, double *file_ms);
/* the same for a text of deflate pieces (CGX_TEXT_GZIP_PIECES): grammar.<first+q>.s.gz = gzip header, the pieces, 03 00, trl[2q] (CRC-32), trl[2q+1] (ISIZE);
 * encoding CGX_TEXT_PLAIN (trl ignored) is cgx_assemble_files */
int cgx_assemble_files_enc(const char *utext, const uint64_t *qseg, const uint64_t *seg_off, const uint32_t *seg_len, int32_t nq, int32_t first,
                           const char *outdir, int nthreads, double *file_ms, int encoding, const uint32_t *trl);
void *cgx_pinned_alloc(size_t bytes);
void cgx_pinned_free(void *p);

/* ---- results: copy a named device/host result into caller memory.
 * dst == NULL returns the size in bytes; otherwise returns bytes written, or < 0. Names:
 *   "sa" "tokstart" "freq" "pidx" "miss" "phit_start" "phit_len"
 *   "lm" "up" "down" "g1" "p1" "pid1" "s1" "hits1" "g2" "c2" "pid2" "s2" "hits2" "r0" "r1" "r2" "p1d" "c2d" "one2" "lex0" "lex1" "lex2" "rng0" "rng1" "rng2" (id -> first,last line; -1,-1 when empty) "counts"
 *   "blocks" "qb_off" "qb_ids" (after cgx_make_blocks), "qo_off" "qo_ids" "qt_off" "qt_ids" (after cgx_format: per query, the ascending ids of its
 *   one-gap / two-gap patterns, CSR; oneGapQueryWithID / twoGapQueryWithID, SuffixArray.cu:1658-1719, 2058-2097) */
int64_t cgx_fetch(cgx_ctx *ctx, const char *name, void *dst, int64_t cap);
/* same, but into page-locked memory owned by the context (DMA speed); the pointer stays valid until
 * cgx_pinned_next_batch has been called twice.  Returns CGX_ERR_NOMEM (and *out = NULL) when pinned memory is unavailable. */
int cgx_fetch_pinned(cgx_ctx *ctx, const char *name, void **out, int64_t *nbytes);
int cgx_pinned_next_batch(cgx_ctx *ctx);
/* last-stage timings in milliseconds (hipEvent): "sa_lookup" "gappy" "extract" "lex" "build_sa" "precompute";
 * "sa_lookup_kernel" is the batched interval-search kernel alone.  Other tallies come through the same call: "append_reruns" (lookup launches repeated
 * because the output capacity remembered from the previous batch was too small, cumulative), "fmt_unique_bytes" / "fmt_plain_unique_bytes" (the
 * unique text of the last batch as laid out / before compression), and the device memory the context holds, in bytes: "mem_index" "mem_derived"
 * ("mem_derived_ngram_tables" "mem_derived_layouts" "mem_derived_lex_hash") "mem_text" "mem_batch" "mem_cached" "mem_other" */
double cgx_stage_ms(cgx_ctx *ctx, const char *name);

/* ---- whole-path host driver (what the CLI calls): text files in, grammar files out ---- */
typedef struct cgx_corpus cgx_corpus;
cgx_corpus *cgx_corpus_load(const char *src, const char *tgt, const char *align, const char *lex, char *err, size_t errcap);
#define CGX_CORPUS_LONG_SENTENCES 1   /* accept sentence pairs of 255 tokens and more (source < 1024, target < 2040 tokens): positions wider than the reference's bytes; results for shorter sentences are unchanged */
cgx_corpus *cgx_corpus_load_opt(const char *src, const char *tgt, const char *align, const char *lex, int flags, char *err, size_t errcap);   /* cgx_corpus_load with options (strmatchcuda --long-sentences); the corpus and index caches record the mode */
int cgx_corpus_flags(const cgx_corpus *c);                  /* CGX_CORPUS_LONG_SENTENCES when the corpus carries wide positions (loaded with that flag, from a cache written in that mode, or built by cgx_corpus_from_ids16) */
/* cgx_corpus_from_ids for sentence pairs of 255 tokens and more (long-sentence mode; the reference refuses them, ExtractPair.cu:2683): the four alignment
 * tables as 16-bit words, 0xFFFF = not aligned; source sentences < 1024 tokens, target sentences < 2040.  NULL on a position or sentence out of that range. */
cgx_corpus *cgx_corpus_from_ids16(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent, const int32_t *tstr, uint32_t nt,
                                  const int32_t *tsentind, const uint16_t *lsrc, const uint16_t *rsrc, const uint16_t *ltar, const uint16_t *rtar,
                                  const cgx_lexkey *lexk, const cgx_lexval *lexv, uint32_t nlex);
void cgx_corpus_free(cgx_corpus *c);
uint64_t cgx_corpus_checksum(const cgx_corpus *c);               /* FNV-1a over every array and spelling of the loaded corpus */
int cgx_corpus_save(const cgx_corpus *c, const char *path);      /* the parsed corpus as one binary file (the reference's own index cache is commented out, SuffixArray.c:208-230) */
cgx_corpus *cgx_corpus_load_cache(const char *path, char *err, size_t errcap);   /* reads a file written by cgx_corpus_save; NULL + message if it is missing, truncated, of another version, fails its checksum or holds ids out of range */
int cgx_corpus_matches_sources(const cgx_corpus *c, const char *src, const char *tgt, const char *align, const char *lex);   /* 1: the cache was made from these four files as they are now (size, mtime); 0: stale; -1: a file cannot be examined */
int cgx_corpus_upload(cgx_ctx *ctx, const cgx_corpus *c);
/* runs lookup -> gappy search -> extraction -> features for the query file and writes
 * <outdir>/grammar.<q>.s; queries [q_begin, q_end) only (q_end < 0: all) for query sharding */
int cgx_extract_grammars(cgx_ctx *ctx, const cgx_corpus *c, const char *qryfile, const char *outdir,
                         int32_t q_begin, int32_t q_end, uint64_t *nrules);
/* query sharding, one process per GPU: shard `shard` of `nshard` contiguous query ranges balanced by token count (cgx_shard_bounds:
 * bounds[r] .. bounds[r+1]-1 are the queries of shard r; qoff = start offset of every query in the token array, ntok tokens in all) */
int cgx_shard_bounds(const int32_t *qoff, int32_t nq, int64_t ntok, int32_t world, int32_t *bounds /* world+1 */);
int cgx_extract_grammars_shard(cgx_ctx *ctx, const cgx_corpus *c, const char *qryfile, const char *outdir, int32_t shard, int32_t nshard, uint64_t *nrules);
/* same on an id-level batch (bench / tests): corpus may have no spellings, words print as s<id>/t<id> */
int cgx_extract_grammars_ids(cgx_ctx *ctx, const cgx_corpus *c, const int32_t *qoff, int32_t nq, const int32_t *qtok,
                             int32_t ntok, const char *outdir, int32_t first_query_index, uint64_t *nrules);
/* with option "async_write" the files of a batch are written in the background while the next batch runs on the GPU;
 * cgx_flush waits for them (also done by the next cgx_extract_grammars* call and by cgx_destroy) */
int cgx_flush(cgx_ctx *ctx);
cgx_corpus *cgx_corpus_from_ids(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent,
                                const int32_t *tstr, uint32_t nt, const int32_t *tsentind,
                                const uint8_t *lsrc, const uint8_t *rsrc, const uint8_t *ltar, const uint8_t *rtar,
                                const cgx_lexkey *lexk, const cgx_lexval *lexv, uint32_t nlex);
/* host stage timings of the last cgx_extract_grammars* call, milliseconds:
 * "blocks" "lists" "lexicon" "write" "total" */
double cgx_host_ms(cgx_ctx *ctx, const char *name);

#ifdef __cplusplus
}
#endif
#endif
