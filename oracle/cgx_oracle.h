/*
 * cgx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread) of the reference hot path
 *   suffix-array lookup -> gappy-phrase search -> rule extraction -> features -> grammar files
 * of hohoCode/cgx.  Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (cgx_amd/, libcgx_hip.so, bin/strmatchcuda) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - host stages (suffix array, lexicon/feature creation, grammar writer) are pinned
 *     against the REAL reference objects compiled from /root/reference/{SuffixArray,
 *     ExtractPair,PrintResults,Timer}.c (oracle/_ref, built by oracle/Makefile);
 *   - the CUDA kernels (.cu) cannot be built in this image (no nvcc / CUDA runtime /
 *     thrust-for-CUDA), the reference ships no tests or golden vectors, so the kernel
 *     stages are "parity unpinned": a line-by-line restatement only.
 */
#ifndef CGX_ORACLE_H
#define CGX_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* compile-time limits of the reference, ComTypes.h:42-65, ExtractPair.cu:16 */
#define ORC_MAX_SPAN 15
#define ORC_MAX_SPAN_PATTERN 15
#define ORC_MAX_SYMBOLS 5
#define ORC_MIN_GAP 1
#define ORC_TOP 100
#define ORC_SAMPLER 300
#define ORC_SAMPLER_ONEGAP 65
#define ORC_SAMPLER_TWOGAP 70
#define ORC_LONGEST_SRC 5
#define ORC_MAXSCORE 99
#define ORC_K1_THREADS 128 /* SuffixArray.cu:1374 - query tokens past 128 are never searched */

/* ---- wire structs: same packed layouts as ComTypes.h so that dumps can be handed to the
 *      real reference objects in oracle/_ref (data formats, not code) ---- */
#pragma pack(push, 1)
typedef struct { int32_t qrystart; uint8_t a_len, b_len, gap; } orc_gappy;                  /* ComTypes.h:142 */
typedef struct { int32_t pat[5]; uint8_t number; } orc_gappat;                              /* :194 */
typedef struct { uint32_t blockid, gap2; uint8_t c_len; } orc_twogappy;                     /* :151 */
typedef struct { int32_t pat[1]; uint8_t number; uint32_t blockid; } orc_twogappat;         /* :200 */
typedef struct { int32_t qrystart; uint8_t a_len, b_len, gap; uint32_t position;
                 int32_t sa_start, sa_end; } orc_gapsearch;                                 /* :168 */
typedef struct { uint32_t blockid, gap2; uint8_t c_len; uint32_t position;
                 int32_t sa_start, sa_end; } orc_twogapsearch;                              /* :158 */
typedef struct { uint32_t position, str_position; uint8_t length; } orc_hit1;               /* :179 */
typedef struct { uint32_t position, str_position; uint8_t length, length2; } orc_hit2;      /* :186 */
typedef struct { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1; } orc_rule1;       /* :224 */
typedef struct { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1, gap2, gap2_1; } orc_rule2; /* :233 */
typedef struct { int32_t tar_start; int32_t block; uint8_t tar_end; } orc_rule0;            /* :349 */
typedef struct { uint32_t start; uint8_t length; } orc_prehit;                              /* :419 */
typedef struct { uint32_t lexid; int32_t src[5]; uint8_t nsrc; uint32_t tstart;
                 uint8_t end, gap1, gap1_1, gap2, gap2_1; } orc_lextask;                    /* :376 */
#pragma pack(pop)
typedef struct { int32_t start, end, matchlen, string_start; } orc_block;                   /* :342 saind_t */
typedef struct { uint32_t start, end; } orc_prerange;                                       /* :322 */
typedef struct { int32_t up, down; } orc_range;                                             /* :93 result_t */
typedef struct { int32_t src, tgt; } orc_lexkey;                                            /* :355 */
typedef struct { float v1, v2; } orc_lexval;                                                /* :360 */

/* one lexicon line (red_dup_t, ComTypes.h:244) */
typedef struct {
    int32_t id;          /* converted id (blocknumber) */
    char *text;          /* "<src> ||| <tgt>" */
    int32_t f;           /* group size */
    int32_t fsample;     /* all_suffix_fsample (capped) */
    int32_t paircount;
    float aa, bb, fscore, maxlex_fe, maxlex_ef;
} orc_lexent;

/* ---- index (corpus side) ---- */
typedef struct {
    /* source side */
    uint32_t n;            /* toklen incl. delimiters + "1,last" sentinels (Start.cu:321-327) */
    int32_t *str;          /* n + 3 zero pads */
    uint8_t *P;            /* in-sentence position */
    int32_t *sentind;      /* nsent+1 */
    int32_t nsent;
    int32_t nsvocab;       /* distinctTokenCount = #words + 2 */
    char **svocab;         /* id -> spelling, ids 2.. */
    int32_t *sa;
    uint32_t *rlp;
    /* target side */
    uint32_t nt;
    int32_t *tstr;
    int32_t *tsentind;
    int32_t ntvocab;
    char **tvocab;
    uint8_t *ltar, *rtar;
    uint16_t *ltar16, *rtar16;   /* long-sentence mode only (orc_set_long_sentences): the same tables with 16-bit positions, 0xFFFF = not aligned */
    /* lexical table, sorted by (src,tgt) */
    uint32_t nlex;
    orc_lexkey *lexk;
    orc_lexval *lexv;
    /* precomputation (SuffixArray.cu:1132-1340) */
    int32_t freq[ORC_TOP];
    orc_prerange pidx[ORC_TOP * ORC_TOP];
    int32_t miss[ORC_TOP * ORC_TOP];
    orc_prehit *phits;
    uint32_t nphits;
    /* vocab hash (source) for query loading */
    void *shash;
    void *thash;
} orc_index;

/* ---- per-batch state: every intermediate the tests compare against ---- */
typedef struct {
    int32_t nq, ntok;
    int32_t *qoff;       /* nq (start offset of each query) */
    int32_t *qtok;       /* ntok */
    int32_t *tok2q;      /* ntok */
    /* stage 1: lookup (K1/K2) */
    int32_t *lm;         /* longestmatch per token */
    int32_t *up, *down;  /* [ntok*5] interval of q[t..t+l) at index t*5+(l-1), -1 when l>lm */
    /* one-gap enumeration */
    uint32_t e1; orc_gappy *g1; orc_gappat *p1;       /* sorted */
    uint32_t d1; orc_gapsearch *s1;
    uint32_t h1; orc_hit1 *hits1;
    /* two-gap */
    uint32_t e2; orc_twogappy *g2; orc_twogappat *p2;
    uint32_t d2; orc_twogapsearch *s2;
    uint32_t h2; orc_hit2 *hits2;
    /* per query id lists */
    uint32_t **qblocks; uint32_t *nqblocks;
    uint32_t **qone;    uint32_t *nqone;
    uint32_t **qtwo;    uint32_t *nqtwo;
    /* blocks */
    uint32_t g; orc_block *blocks; char **blockname;
    /* rules */
    uint32_t n0; orc_rule0 *r0;
    uint32_t n1; orc_rule1 *r1; uint32_t sep1;           /* [Xab/abX | aXb] */
    uint32_t n2; orc_rule2 *r2; uint32_t sep2a, sep2b;   /* [XabX | aXbXc | XaXb/aXbX] */
    /* lexicon */
    uint32_t nl1; orc_lexent *lex1; orc_range *rng1; uint32_t nrng1;
    uint32_t nl2; orc_lexent *lex2; orc_range *rng2; uint32_t nrng2;
    uint32_t nl0; orc_lexent *lex0; orc_range *rng0;
    uint32_t ntask; orc_lextask *tasks; float *task_fe, *task_ef;
    uint64_t nlines;     /* grammar lines written by the last orc_write_grammars */
    double t_lookup, t_gappy, t_extract, t_lexicon, t_lextask, t_write; /* seconds */
} orc_batch;

/* index construction from text files (Start.cu:142-380, ExtractPair.cu:2442-2554,2639-2739) */
orc_index *orc_index_load(const char *src, const char *tgt, const char *align, const char *lex);
/* index from id arrays (bench / tests); sa may be NULL (then built here). Arrays are copied. */
orc_index *orc_index_from_arrays(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent,
                                 const int32_t *tstr, uint32_t nt, const int32_t *tsentind,
                                 const uint8_t *lsrc, const uint8_t *rsrc, /* per source token L/R (255 = unaligned) */
                                 const uint8_t *ltar, const uint8_t *rtar,
                                 const orc_lexkey *lexk, const orc_lexval *lexv, uint32_t nlex,
                                 const int32_t *sa);
/* the same with the suffix array and the frequent-pair tables handed in (the layouts of cgx_fetch "sa" "freq" "pidx" "miss" "phit_start" "phit_len"); nothing is built */
orc_index *orc_index_from_arrays_pre(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent,
                                     const int32_t *tstr, uint32_t nt, const int32_t *tsentind,
                                     const uint8_t *lsrc, const uint8_t *rsrc, const uint8_t *ltar, const uint8_t *rtar,
                                     const orc_lexkey *lexk, const orc_lexval *lexv, uint32_t nlex, const int32_t *sa,
                                     const int32_t *freq, const uint32_t *pidx, const int32_t *miss, const uint32_t *phit_start, const uint8_t *phit_len, uint32_t nphits);
void orc_index_free(orc_index *ix);
void orc_set_long_sentences(int on);   /* before orc_index_load: accept sentences of 255+ tokens (source < 1024, target < 2040); default off = the reference's byte positions */
void orc_build_sa(const int32_t *str, uint32_t n, int32_t *sa);
int orc_precompute(orc_index *ix);

orc_batch *orc_batch_load(const orc_index *ix, const char *qryfile);
orc_batch *orc_batch_from_ids(const int32_t *qoff, int32_t nq, const int32_t *qtok, int32_t ntok);
void orc_batch_free(orc_batch *b);

/* the stages, in reference order */
void orc_sa_lookup(const orc_index *ix, orc_batch *b);        /* K1+K2, SuffixArray.cu:109-767 */
void orc_gappy_search(const orc_index *ix, orc_batch *b);     /* SuffixArray.cu:1530-2256 */
void orc_extract(const orc_index *ix, orc_batch *b);          /* ExtractPair.cu:3215-3670 */
void orc_features(const orc_index *ix, orc_batch *b);         /* ExtractPair.cu:3694-3982 + ExtractPair.c */
int  orc_write_grammars(const orc_batch *b, const char *outdir, int first_query_index); /* PrintResults.c:407-577 */
int  orc_run_all(const orc_index *ix, orc_batch *b, const char *outdir);
uint64_t orc_batch_times(const orc_batch *b, double *t6);   /* the six stage timers above, seconds; returns nlines */

/* helpers exported for unit tests */
int orc_check_gap(const orc_index *ix, uint32_t start, uint32_t ender);                     /* GappyLook.cu:43-126 */
int orc_sample_hit(int n, int sampler, int idx);                                            /* ExtractPair.cu:1143-1160 */
float orc_lex_lookup(const orc_index *ix, int32_t src, int32_t tgt, int which);             /* ExtractPair.cu:2108-2142 */

/* binary dump of every intermediate: [8-byte tag][u64 nbytes][payload]... */
int orc_dump(const orc_index *ix, const orc_batch *b, const char *path);

#ifdef __cplusplus
}
#endif
#endif
