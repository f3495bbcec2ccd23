/*
 * cgx_host.c -- host side (plain C) of the extractor: corpus / query / alignment / lexical
 * table loaders, distinct-phrase bookkeeping, lexicon + feature creation and the grammar
 * writer.  Everything between those stages runs on the GPU through the C ABI in cgx.h;
 * nothing here has a CPU fallback for the kernels.
 *
 * Reference behaviour followed (file:line in /root/reference):
 *   corpus tokenisation, ids, sentinels          Start.cu:142-380
 *   query tokenisation, OOV = -1                 Start.cu:50-132
 *   alignment -> RLP / L_tar / R_tar             ExtractPair.cu:2639-2739
 *   lexical table text format                    ExtractPair.cu:2442-2519
 *   distinct contiguous phrases (blocks)         ExtractPair.cu:2742-2903
 *   per-query pattern id lists                   SuffixArray.cu:1666-1719, 2056-2097
 *   lexicon + features                           ExtractPair.c:515-1276
 *   id -> lexicon ranges                         ExtractPair.cu:3743-3756, 3802-3816, 2082-2106
 *   grammar files                                PrintResults.c:339-577
 * Unlike the reference, the lexicon is keyed on integer target-symbol tuples; spellings are
 * only touched when a line is formatted.
 */
#define _GNU_SOURCE
#include "../../include/cgx.h"
#include "cgx_internal.h"
#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <zlib.h>

#define SAMPLER 300
#define LONGEST_SRC 5

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

/* ------------------------------------------------------------------ */
/* word -> id table                                                    */
/* ------------------------------------------------------------------ */
typedef struct { const char **key; int32_t *val; size_t cap, n; } wordmap;
static uint64_t hash_bytes(const char *s, size_t n) { uint64_t h = 0xcbf29ce484222325ull; for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ull; } return h; }
static int wordmap_init(wordmap *m, size_t cap) { m->cap = cap; m->n = 0; m->key = calloc(cap, sizeof *m->key); m->val = calloc(cap, sizeof *m->val); return m->key && m->val ? 0 : -1; }
static void wordmap_free(wordmap *m) { free(m->key); free(m->val); memset(m, 0, sizeof *m); }
static int32_t wordmap_get(const wordmap *m, const char *s, size_t len) {
    if (!m->cap) return -1;
    size_t i = hash_bytes(s, len) & (m->cap - 1);
    while (m->key[i]) { if (!strncmp(m->key[i], s, len) && m->key[i][len] == 0) return m->val[i]; i = (i + 1) & (m->cap - 1); }
    return -1;
}
static int wordmap_put(wordmap *m, const char *key, int32_t v) {
    if ((m->n + 1) * 2 > m->cap) {
        wordmap b; if (wordmap_init(&b, m->cap * 2)) return -1;
        for (size_t j = 0; j < m->cap; j++) if (m->key[j]) wordmap_put(&b, m->key[j], m->val[j]);
        wordmap_free(m); *m = b;
    }
    size_t i = hash_bytes(key, strlen(key)) & (m->cap - 1);
    while (m->key[i]) i = (i + 1) & (m->cap - 1);
    m->key[i] = key; m->val[i] = v; m->n++;
    return 0;
}

/* ------------------------------------------------------------------ */
/* corpus                                                              */
/* ------------------------------------------------------------------ */
struct cgx_corpus {
    uint32_t n, nt; int32_t nsent;
    int32_t *str, *tstr, *sentind, *tsentind;
    uint8_t *P;
    uint32_t *rlp; uint8_t *ltar, *rtar;
    int long_pos; uint16_t *ltar16, *rtar16;              /* long-sentence mode (cgx_corpus_load_opt, CGX_CORPUS_LONG_SENTENCES): wider positions, see cgx_rules.h */
    char **svocab, **tvocab; int32_t nsvocab, ntvocab;   /* id -> spelling, NULL entries when built from ids */
    uint32_t *svlen, *tvlen; uint32_t maxword;            /* spelling lengths (writer) */
    struct wslot { uint8_t len; char s[15]; } *svslot, *tvslot;   /* words of <= 15 bytes packed in 16-byte slots: one cache line serves four words */
    wordmap smap, tmap;
    cgx_lexkey *lexk; cgx_lexval *lexv; uint32_t nlex;
    uint64_t src_size[4], src_mtime[4];                   /* the four text files the corpus was parsed from (0: unknown), kept in the cache header */
};

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)sz + 2);
    if (!buf) { fclose(f); return NULL; }
    size_t got = fread(buf, 1, (size_t)sz, f); fclose(f);
    buf[got] = 0; *len = got;
    return buf;
}
#define GROW(arr, cnt, cap) do { if ((cnt) == (cap)) { (cap) = (cap) ? (cap) * 2 : 1024; (arr) = realloc((arr), (cap) * sizeof *(arr)); if (!(arr)) return -1; } } while (0)

/* One side of the bitext.  Lines end at '\n'; words are separated by single blanks only
 * (strtok(" ")); a word that starts with other white space ends the line (Start.cu:280).
 * Ids are handed out in order of first appearance (2, 3, ...), exactly as a sequential reader would,
 * but the file is tokenised by several threads: each parses a run of whole lines against a private
 * dictionary, the private dictionaries are merged in file order (which reproduces the sequential
 * numbering), then the threads rewrite their private ids and copy their piece into place. */
typedef struct { const char **key; uint32_t *klen; int32_t *val; size_t cap, n; } spanmap;   /* words as (pointer, length) into the file buffer */
static int spanmap_id(spanmap *m, const char *s, uint32_t len, int *added) {
    if ((m->n + 1) * 2 > m->cap) {
        size_t oc = m->cap, nc = oc ? oc * 2 : 4096;
        const char **ok = m->key; uint32_t *ol = m->klen; int32_t *ov = m->val;
        m->key = calloc(nc, sizeof *m->key); m->klen = malloc(nc * sizeof *m->klen); m->val = malloc(nc * sizeof *m->val);
        if (!m->key || !m->klen || !m->val) return -1;
        m->cap = nc;
        for (size_t j = 0; j < oc; j++) if (ok[j]) { size_t i = hash_bytes(ok[j], ol[j]) & (nc - 1); while (m->key[i]) i = (i + 1) & (nc - 1); m->key[i] = ok[j]; m->klen[i] = ol[j]; m->val[i] = ov[j]; }
        free(ok); free(ol); free(ov);
    }
    size_t i = hash_bytes(s, len) & (m->cap - 1);
    while (m->key[i]) { if (m->klen[i] == len && !memcmp(m->key[i], s, len)) { *added = 0; return m->val[i]; } i = (i + 1) & (m->cap - 1); }
    m->key[i] = s; m->klen[i] = len; m->val[i] = (int32_t)m->n + 2; m->n++; *added = 1;
    return m->val[i];
}
typedef struct {
    const char *buf; size_t begin, end; int want_P, rc;
    int32_t *tok; uint8_t *P; size_t ntok, ctok;              /* private ids (>= 2), 1 closes a line */
    int32_t *sent; size_t nsent, csent;                       /* tokens of this piece up to the end of each of its lines */
    spanmap map; const char **wkey; uint32_t *wlen; size_t nw, cw;   /* its new words in order of first appearance */
    int32_t *l2g;                                             /* private id -> final id (filled by the merge) */
    int32_t *str_out; uint8_t *P_out; int32_t *sent_out; size_t tok_off, sent_off;
} sidepiece;
static int piece_room(sidepiece *pc) {                      /* room for one more token */
    if (pc->ntok < pc->ctok) return 0;
    pc->ctok = pc->ctok ? pc->ctok * 2 : 1 << 16;
    pc->tok = realloc(pc->tok, pc->ctok * sizeof *pc->tok);
    if (pc->want_P) pc->P = realloc(pc->P, pc->ctok);
    return !pc->tok || (pc->want_P && !pc->P) ? -1 : 0;
}
static int piece_parse(sidepiece *pc) {
    const char *buf = pc->buf; size_t i = pc->begin; const size_t len = pc->end;
    while (i < len) {
        size_t e = i; while (e < len && buf[e] != '\n') e++;
        uint8_t local = 0; size_t p = i;
        for (;;) {
            while (p < e && buf[p] == ' ') p++;
            if (p >= e) break;
            size_t q = p; while (q < e && buf[q] != ' ') q++;
            if (isspace((unsigned char)buf[p])) break;
            int added; int32_t id = spanmap_id(&pc->map, buf + p, (uint32_t)(q - p), &added);
            if (id < 0) return -1;
            if (added) {
                if (pc->nw == pc->cw) {
                    pc->cw = pc->cw ? pc->cw * 2 : 1024;
                    pc->wkey = realloc(pc->wkey, pc->cw * sizeof *pc->wkey); pc->wlen = realloc(pc->wlen, pc->cw * sizeof *pc->wlen);
                    if (!pc->wkey || !pc->wlen) return -1;
                }
                pc->wkey[pc->nw] = buf + p; pc->wlen[pc->nw] = (uint32_t)(q - p); pc->nw++;
            }
            if (piece_room(pc)) return -1;
            pc->tok[pc->ntok] = id; if (pc->want_P) pc->P[pc->ntok] = local; pc->ntok++;
            local++; p = q;
        }
        if (piece_room(pc)) return -1;
        pc->tok[pc->ntok] = 1; if (pc->want_P) pc->P[pc->ntok] = 0; pc->ntok++;
        GROW(pc->sent, pc->nsent, pc->csent); pc->sent[pc->nsent++] = (int32_t)pc->ntok;
        i = e + 1;
    }
    return 0;
}
static void *piece_parse_job(void *arg) { sidepiece *pc = arg; pc->rc = piece_parse(pc); return NULL; }
static void *piece_place_job(void *arg) {
    sidepiece *pc = arg;
    for (size_t k = 0; k < pc->ntok; k++) pc->str_out[pc->tok_off + k] = pc->tok[k] == 1 ? 1 : pc->l2g[pc->tok[k]];
    if (pc->want_P && pc->ntok) memcpy(pc->P_out + pc->tok_off, pc->P, pc->ntok);      /* an empty piece has no P array */
    for (size_t k = 0; k < pc->nsent; k++) pc->sent_out[pc->sent_off + k] = (int32_t)(pc->tok_off + (size_t)pc->sent[k]);
    return NULL;
}
static void piece_free(sidepiece *pc) { free(pc->tok); free(pc->P); free(pc->sent); free(pc->map.key); free(pc->map.klen); free(pc->map.val); free(pc->wkey); free(pc->wlen); free(pc->l2g); }
static int nthreads_host(void);
#define SIDE_MAX_PIECES 32
/* files smaller than this are parsed by one thread (CGX_LOAD_PIECE_MIN overrides it: tests force the multi-piece path on small fixtures) */
static size_t piece_min_bytes(void) { const char *e = getenv("CGX_LOAD_PIECE_MIN"); long v = e ? atol(e) : 0; return v > 0 ? (size_t)v : (size_t)4 << 20; }
static int load_side(const char *path, int32_t **str_out, uint32_t *n_out, uint8_t **P_out, int32_t **sent_out, int32_t *nsent_out,
                     char ***vocab_out, int32_t *nvocab_out, wordmap *map) {
    double tr0 = now_ms(); const int trace = getenv("CGX_TRACE") != NULL;
    size_t len; char *buf = slurp(path, &len);
    if (!buf) return -1;
    double tr1 = now_ms();
    int np = nthreads_host() / 2; if (np > SIDE_MAX_PIECES) np = SIDE_MAX_PIECES; if (len < piece_min_bytes() || np < 1) np = 1;   /* two sides load at once */
    sidepiece pc[SIDE_MAX_PIECES]; memset(pc, 0, sizeof pc);
    size_t cut = 0;
    for (int k = 0; k < np; k++) {                            /* pieces start at line starts */
        pc[k].buf = buf; pc[k].want_P = P_out != NULL; pc[k].begin = cut;
        size_t e = k + 1 == np ? len : len / (size_t)np * (size_t)(k + 1);
        if (e < cut) e = cut;
        while (e < len && buf[e] != '\n') e++;
        if (e < len) e++;
        pc[k].end = cut = e;
    }
    pthread_t th[SIDE_MAX_PIECES]; int started[SIDE_MAX_PIECES] = {0}; int rc = 0;
    for (int k = 1; k < np; k++) started[k] = !pthread_create(&th[k], NULL, piece_parse_job, &pc[k]);
    piece_parse_job(&pc[0]);
    for (int k = 1; k < np; k++) { if (started[k]) pthread_join(th[k], NULL); else piece_parse_job(&pc[k]); }
    for (int k = 0; k < np; k++) if (pc[k].rc) rc = -1;
    double tr2 = now_ms();
    /* merge the dictionaries in file order: the id of a word is 2 + the number of distinct words seen before it */
    char **voc = NULL; size_t nv = 0, cv = 0; int32_t *str = NULL, *sent = NULL; uint8_t *P = NULL;
    if (!rc && wordmap_init(map, 1 << 16)) rc = -1;
    if (!rc) { GROW(voc, nv, cv); voc[nv++] = NULL; GROW(voc, nv, cv); voc[nv++] = NULL; }
    for (int k = 0; k < np && !rc; k++) {
        pc[k].l2g = malloc((pc[k].nw + 2) * sizeof *pc[k].l2g);
        if (!pc[k].l2g) { rc = -1; break; }
        for (size_t j = 0; j < pc[k].nw; j++) {
            int32_t id = wordmap_get(map, pc[k].wkey[j], pc[k].wlen[j]);
            if (id < 0) {
                id = (int32_t)map->n + 2;
                char *w = malloc((size_t)pc[k].wlen[j] + 1);
                if (!w) { rc = -1; break; }
                memcpy(w, pc[k].wkey[j], pc[k].wlen[j]); w[pc[k].wlen[j]] = 0;
                if (wordmap_put(map, w, id)) { rc = -1; break; }
                if (nv == cv) { cv = cv ? cv * 2 : 1024; voc = realloc(voc, cv * sizeof *voc); if (!voc) { rc = -1; break; } }
                voc[nv++] = w;
            }
            pc[k].l2g[j + 2] = id;
        }
    }
    double tr3 = now_ms();
    size_t ns = 0, nl = 0;
    for (int k = 0; k < np; k++) { pc[k].tok_off = ns; pc[k].sent_off = nl + 1; ns += pc[k].ntok; nl += pc[k].nsent; }
    if (!rc) {
        str = malloc((ns + 2) * sizeof *str); sent = malloc((nl + 1) * sizeof *sent); if (P_out) P = malloc(ns + 2);
        if (!str || !sent || (P_out && !P)) rc = -1;
    }
    if (!rc) {
        sent[0] = 0;
        for (int k = 0; k < np; k++) { pc[k].str_out = str; pc[k].P_out = P; pc[k].sent_out = sent; }
        for (int k = 1; k < np; k++) started[k] = !pthread_create(&th[k], NULL, piece_place_job, &pc[k]);
        piece_place_job(&pc[0]);
        for (int k = 1; k < np; k++) { if (started[k]) pthread_join(th[k], NULL); else piece_place_job(&pc[k]); }
        int32_t last = map->n ? (int32_t)map->n + 1 : -1;     /* id of the newest word, then one past it (Start.cu:300-312) */
        str[ns] = 1; str[ns + 1] = last + 1; if (P) { P[ns] = 0; P[ns + 1] = 0; }
        *str_out = str; *n_out = (uint32_t)(ns + 2); if (P_out) *P_out = P;
        *sent_out = sent; *nsent_out = (int32_t)nl; *vocab_out = voc; *nvocab_out = (int32_t)nv;
    } else { free(str); free(sent); free(P); for (size_t j = 2; j < nv; j++) free(voc[j]); free(voc); }
    if (trace) fprintf(stderr, "cgx: %s: read %.0f ms, tokenise (%d threads) %.0f ms, merge %.0f ms, place %.0f ms\n", path, tr1 - tr0, np, tr2 - tr1, tr3 - tr2, now_ms() - tr3);
    for (int k = 0; k < np; k++) piece_free(&pc[k]);
    free(buf);
    return rc;
}

/* cgx_rlp_pack of csrc/cgx_rules.h, restated for this C file: L / R < 0 = not aligned; position codes skip every value whose low byte is 255 */
static uint32_t rlp_pack_long(int L, int R, uint32_t P) {
    const uint32_t cl = L < 0 ? 255u : (uint32_t)L + (uint32_t)L / 255u, cr = R < 0 ? 255u : (uint32_t)R + (uint32_t)R / 255u;
    return ((cl & 255u) << 24) | ((cr & 255u) << 16) | ((P & 255u) << 8) | ((cl >> 8) << 5) | ((cr >> 8) << 2) | (P >> 8);
}
#define LONG_MAX_SRC 1024
#define LONG_MAX_TGT 2040
/* Ls / Rs: min / max aligned target position per source token, -1 = none */
static int pack_alignment_i(cgx_corpus *c, const int32_t *Ls, const int32_t *Rs) {
    c->rlp = calloc((size_t)c->n + 1, sizeof(uint32_t));
    if (!c->rlp) return -1;
    int q = 1;
    for (uint32_t i = 0; i + 1 < c->n; i++) {
        if (q <= c->nsent && (int32_t)i == c->sentind[q] - 1) { c->rlp[i] = (uint32_t)c->tsentind[q]; q++; }
        else if (!c->long_pos) c->rlp[i] = ((uint32_t)(Ls[i] < 0 ? 255 : Ls[i]) << 24) | ((uint32_t)(Rs[i] < 0 ? 255 : Rs[i]) << 16) | ((uint32_t)c->P[i] << 8);
        else {
            const uint32_t pp = q - 1 < c->nsent ? (uint32_t)((int32_t)i - c->sentind[q - 1]) : 0u;      /* the token's true position in its sentence */
            if (pp >= LONG_MAX_SRC) return -2;
            c->rlp[i] = rlp_pack_long(Ls[i], Rs[i], pp);
        }
    }
    return 0;
}
static int pack_alignment(cgx_corpus *c, const uint8_t *Ls, const uint8_t *Rs) {      /* byte tables (cgx_corpus_from_ids) */
    int32_t *l = malloc(((size_t)c->n + 1) * 4), *r = malloc(((size_t)c->n + 1) * 4); int rc = -1;
    if (l && r) { for (uint32_t i = 0; i < c->n; i++) { l[i] = Ls[i] == 255 ? -1 : Ls[i]; r[i] = Rs[i] == 255 ? -1 : Rs[i]; } rc = pack_alignment_i(c, l, r); }
    free(l); free(r);
    return rc;
}
/* Alignment lines are independent once their line number is known, and line q only touches the tokens of
 * sentence pair q: pieces of whole lines are parsed by several threads; the error of the earliest piece wins. */
typedef struct { cgx_corpus *c; const char *buf; size_t begin, end; int q0, rc; int32_t *Ls, *Rs, *lt, *rt; char err[160]; } alignpiece;
static void *align_piece_job(void *arg) {
    alignpiece *a = arg; cgx_corpus *c = a->c; const char *buf = a->buf; int32_t *Ls = a->Ls, *Rs = a->Rs, *lt = a->lt, *rt = a->rt;
    const int max_s = c->long_pos ? LONG_MAX_SRC : 255, max_t = c->long_pos ? LONG_MAX_TGT : 255;       /* the reference: 255 (ExtractPair.cu:2683) */
    size_t i = a->begin; const size_t len = a->end; int q = a->q0 - 1; int rc = CGX_OK;
    while (i < len && rc == CGX_OK) {
        size_t e = i; while (e < len && buf[e] != '\n') e++;
        q++;
        if (q >= c->nsent) { snprintf(a->err, sizeof a->err, "alignment file has more lines than the corpus"); rc = CGX_ERR_ARG; break; }
        size_t p = i; int have_s = 0, s = 0;
        for (;;) {                                      /* tokens separated by blanks and '-' (ExtractPair.cu:2657) */
            while (p < e && (buf[p] == ' ' || buf[p] == '-')) p++;
            if (p >= e) break;
            size_t t = p; while (t < e && buf[t] != ' ' && buf[t] != '-') t++;
            if (isspace((unsigned char)buf[p])) break;
            int val = 0;                                  /* atoi() of the token: optional '+', leading digits, anything else ends it */
            { size_t k = p; if (k < t && buf[k] == '+') k++; for (; k < t && buf[k] >= '0' && buf[k] <= '9'; k++) if (val < 100000000) val = val * 10 + (buf[k] - '0'); }
            p = t;
            if (!have_s) { s = val; have_s = 1; continue; }
            have_s = 0;
            if (s >= max_s || val >= max_t || s < 0 || val < 0) { snprintf(a->err, sizeof a->err, "Not possible, too long sentence"); rc = CGX_ERR_ALIGN_RANGE; break; }
            uint32_t si = (uint32_t)(c->sentind[q] + s), ti = (uint32_t)(c->tsentind[q] + val);
            if (si >= (uint32_t)c->sentind[q + 1] || ti >= (uint32_t)c->tsentind[q + 1]) { snprintf(a->err, sizeof a->err, "alignment link outside its sentence pair on line %d", q + 1); rc = CGX_ERR_ARG; break; }
            if (Ls[si] < 0 || Rs[si] < 0) Ls[si] = Rs[si] = val; else if (val > Rs[si]) Rs[si] = val; else if (val < Ls[si]) Ls[si] = val;
            if (lt[ti] < 0 || rt[ti] < 0) lt[ti] = rt[ti] = s; else if (s > rt[ti]) rt[ti] = s; else if (s < lt[ti]) lt[ti] = s;
        }
        if (rc == CGX_OK && have_s) { snprintf(a->err, sizeof a->err, "Not possible!"); rc = CGX_ERR_ALIGN_PAIR; }
        i = e + 1;
    }
    a->rc = rc;
    return NULL;
}
static int load_alignment(cgx_corpus *c, const char *path, char *err, size_t errcap) {
    size_t len; char *buf = slurp(path, &len);
    if (!buf) { snprintf(err, errcap, "Can not open reference file \"%s\"", path); return CGX_ERR_IO; }
    int32_t *Ls = malloc(((size_t)c->n + 1) * 4), *Rs = malloc(((size_t)c->n + 1) * 4), *lt = malloc(((size_t)c->nt + 1) * 4), *rt = malloc(((size_t)c->nt + 1) * 4);
    c->ltar = malloc((size_t)c->nt + 1); c->rtar = malloc((size_t)c->nt + 1);
    if (!Ls || !Rs || !lt || !rt || !c->ltar || !c->rtar) { free(Ls); free(Rs); free(lt); free(rt); free(buf); return CGX_ERR_NOMEM; }
    memset(Ls, 0xFF, ((size_t)c->n + 1) * 4); memset(Rs, 0xFF, ((size_t)c->n + 1) * 4); memset(lt, 0xFF, ((size_t)c->nt + 1) * 4); memset(rt, 0xFF, ((size_t)c->nt + 1) * 4);   /* -1 = none */
    int np = nthreads_host() / 2; if (np > SIDE_MAX_PIECES) np = SIDE_MAX_PIECES; if (len < piece_min_bytes() || np < 1) np = 1;   /* the lexical table loads at the same time */
    alignpiece pc[SIDE_MAX_PIECES]; size_t cut = 0; int line = 0;
    for (int k = 0; k < np; k++) {
        pc[k].c = c; pc[k].buf = buf; pc[k].Ls = Ls; pc[k].Rs = Rs; pc[k].lt = lt; pc[k].rt = rt; pc[k].rc = CGX_OK; pc[k].err[0] = 0; pc[k].begin = cut; pc[k].q0 = line;
        size_t e = k + 1 == np ? len : len / (size_t)np * (size_t)(k + 1);
        if (e < cut) e = cut;
        while (e < len && buf[e] != '\n') e++;
        if (e < len) e++;
        for (const char *p = buf + cut; (p = memchr(p, '\n', (size_t)(buf + e - p))) != NULL; p++) line++;
        if (e == len && e > cut && buf[e - 1] != '\n') line++;      /* last line without a newline */
        pc[k].end = cut = e;
    }
    pthread_t th[SIDE_MAX_PIECES]; int started[SIDE_MAX_PIECES] = {0};
    for (int k = 1; k < np; k++) started[k] = !pthread_create(&th[k], NULL, align_piece_job, &pc[k]);
    align_piece_job(&pc[0]);
    for (int k = 1; k < np; k++) { if (started[k]) pthread_join(th[k], NULL); else align_piece_job(&pc[k]); }
    int rc = CGX_OK;
    for (int k = 0; k < np && rc == CGX_OK; k++) if (pc[k].rc != CGX_OK) { rc = pc[k].rc; snprintf(err, errcap, "%s", pc[k].err); }
    free(buf);
    /* target-side tables: bytes as in the reference; 16-bit words beside them in long-sentence mode */
    for (uint32_t i = 0; i < c->nt; i++) { c->ltar[i] = (uint8_t)(lt[i] < 0 || lt[i] > 254 ? 255 : lt[i]); c->rtar[i] = (uint8_t)(rt[i] < 0 || rt[i] > 254 ? 255 : rt[i]); }
    if (rc == CGX_OK && c->long_pos) {
        c->ltar16 = malloc(((size_t)c->nt + 1) * 2); c->rtar16 = malloc(((size_t)c->nt + 1) * 2);
        if (!c->ltar16 || !c->rtar16) rc = CGX_ERR_NOMEM;
        else for (uint32_t i = 0; i < c->nt; i++) { c->ltar16[i] = (uint16_t)(lt[i] < 0 ? 0xFFFF : lt[i]); c->rtar16[i] = (uint16_t)(rt[i] < 0 ? 0xFFFF : rt[i]); }
    }
    if (rc == CGX_OK) { int pr = pack_alignment_i(c, Ls, Rs); if (pr == -2) { snprintf(err, errcap, "Not possible, too long sentence"); rc = CGX_ERR_ALIGN_RANGE; } else if (pr) rc = CGX_ERR_NOMEM; }
    free(Ls); free(Rs); free(lt); free(rt);
    return rc;
}
/* Lexical table: four white-space separated fields per entry, like `file >> a >> b >> v1 >> v2` (entries, not lines:
 * a piece boundary must fall between entries, so pieces are cut at line ends and a piece whose field count is not a
 * multiple of four makes the loader fall back to one piece).  Rows keep file order; the "Not Available" notices of
 * the reference are printed in file order after the pieces are joined. */
typedef struct { const cgx_corpus *c; const char *buf; size_t begin, end; cgx_lexkey *k; cgx_lexval *v; size_t n, cap; char *msg; size_t nmsg, cmsg; size_t fields; int rc; } lexpiece;
static int lexpiece_note(lexpiece *lp, const char *what, const char *w, size_t wl) {
    size_t need = strlen(what) + wl + 2;
    if (lp->nmsg + need + 1 > lp->cmsg) { size_t nc = lp->cmsg ? lp->cmsg * 2 : 4096; while (nc < lp->nmsg + need + 1) nc *= 2; lp->msg = realloc(lp->msg, nc); if (!lp->msg) return -1; lp->cmsg = nc; }
    lp->nmsg += (size_t)sprintf(lp->msg + lp->nmsg, "%s%.*s\n", what, (int)wl, w);
    return 0;
}
static void *lex_piece_job(void *arg) {
    lexpiece *lp = arg; const cgx_corpus *c = lp->c; const char *buf = lp->buf; size_t p = lp->begin; const size_t len = lp->end;
    const char *w[4]; size_t wl[4];
    for (;;) {
        int k = 0;
        while (k < 4) {
            while (p < len && isspace((unsigned char)buf[p])) p++;
            if (p >= len) break;
            w[k] = buf + p; size_t q = p; while (q < len && !isspace((unsigned char)buf[q])) q++;
            wl[k] = q - p; p = q; k++;
        }
        lp->fields += (size_t)k;
        if (k < 4) break;
        int32_t s = wordmap_get(&c->smap, w[0], wl[0]), t = wordmap_get(&c->tmap, w[1], wl[1]);
        int snull = wl[0] == 4 && !strncmp(w[0], "NULL", 4), tnull = wl[1] == 4 && !strncmp(w[1], "NULL", 4);
        if (s < 0 && !snull) { if (lexpiece_note(lp, "Ch Not Available!!! ", w[0], wl[0])) { lp->rc = CGX_ERR_NOMEM; return NULL; } continue; }
        if (t < 0 && !tnull) { if (lexpiece_note(lp, "En Not Available!!! ", w[1], wl[1])) { lp->rc = CGX_ERR_NOMEM; return NULL; } continue; }
        char f1[64], f2[64]; size_t a = wl[2] < 63 ? wl[2] : 63, b = wl[3] < 63 ? wl[3] : 63;
        memcpy(f1, w[2], a); f1[a] = 0; memcpy(f2, w[3], b); f2[b] = 0;
        if (lp->n == lp->cap) { lp->cap = lp->cap ? lp->cap * 2 : 4096; lp->k = realloc(lp->k, lp->cap * sizeof *lp->k); lp->v = realloc(lp->v, lp->cap * sizeof *lp->v); if (!lp->k || !lp->v) { lp->rc = CGX_ERR_NOMEM; return NULL; } }
        lp->k[lp->n].src = s < 0 ? -1 : s; lp->k[lp->n].tgt = t < 0 ? -1 : t;
        lp->v[lp->n].v1 = strtof(f1, NULL); lp->v[lp->n].v2 = strtof(f2, NULL); lp->n++;
    }
    return NULL;
}
static int load_lex(cgx_corpus *c, const char *path, char *err, size_t errcap) {
    size_t len; char *buf = slurp(path, &len);
    if (!buf) { snprintf(err, errcap, "The Word Possibility File is not Found!"); return CGX_ERR_IO; }
    int np = nthreads_host() / 2; if (np > SIDE_MAX_PIECES) np = SIDE_MAX_PIECES; if (len < piece_min_bytes() || np < 1) np = 1;   /* the alignment loads at the same time */
    lexpiece pc[SIDE_MAX_PIECES]; pthread_t th[SIDE_MAX_PIECES]; int started[SIDE_MAX_PIECES]; int rc = CGX_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        memset(pc, 0, sizeof pc); memset(started, 0, sizeof started);
        size_t cut = 0;
        for (int k = 0; k < np; k++) {
            pc[k].c = c; pc[k].buf = buf; pc[k].begin = cut;
            size_t e = k + 1 == np ? len : len / (size_t)np * (size_t)(k + 1);
            if (e < cut) e = cut;
            while (e < len && buf[e] != '\n') e++;
            if (e < len) e++;
            pc[k].end = cut = e;
        }
        for (int k = 1; k < np; k++) started[k] = !pthread_create(&th[k], NULL, lex_piece_job, &pc[k]);
        lex_piece_job(&pc[0]);
        for (int k = 1; k < np; k++) { if (started[k]) pthread_join(th[k], NULL); else lex_piece_job(&pc[k]); }
        int ragged = 0;
        for (int k = 0; k + 1 < np; k++) if (pc[k].fields % 4) ragged = 1;     /* an entry straddles a line end: only a sequential read parses it like `>>` */
        for (int k = 0; k < np; k++) if (pc[k].rc != CGX_OK) rc = pc[k].rc;
        if (!ragged || np == 1 || rc != CGX_OK) break;
        for (int k = 0; k < np; k++) { free(pc[k].k); free(pc[k].v); free(pc[k].msg); }
        np = 1;
    }
    size_t n = 0;
    for (int k = 0; k < np; k++) n += pc[k].n;
    if (rc == CGX_OK) {
        c->lexk = malloc((n + 1) * sizeof *c->lexk); c->lexv = malloc((n + 1) * sizeof *c->lexv);
        if (!c->lexk || !c->lexv) rc = CGX_ERR_NOMEM;
    }
    size_t at = 0;
    for (int k = 0; k < np; k++) {
        if (rc == CGX_OK) {
            if (pc[k].nmsg) fwrite(pc[k].msg, 1, pc[k].nmsg, stdout);
            if (pc[k].n) { memcpy(c->lexk + at, pc[k].k, pc[k].n * sizeof *c->lexk); memcpy(c->lexv + at, pc[k].v, pc[k].n * sizeof *c->lexv); at += pc[k].n; }
        }
        free(pc[k].k); free(pc[k].v); free(pc[k].msg);
    }
    free(buf);
    if (rc == CGX_OK) c->nlex = (uint32_t)n;
    return rc;
}

static int build_word_slots(cgx_corpus *c) {
    c->svslot = calloc((size_t)c->nsvocab + 1, sizeof *c->svslot); c->tvslot = calloc((size_t)c->ntvocab + 1, sizeof *c->tvslot);
    if (!c->svslot || !c->tvslot) return -1;
    for (int32_t i = 0; i < c->nsvocab; i++) { uint32_t L = c->svocab[i] ? c->svlen[i] : 0; if (c->svocab[i] && L <= 15) { c->svslot[i].len = (uint8_t)L; memcpy(c->svslot[i].s, c->svocab[i], L); } else c->svslot[i].len = 255; }
    for (int32_t i = 0; i < c->ntvocab; i++) { uint32_t L = c->tvocab[i] ? c->tvlen[i] : 0; if (c->tvocab[i] && L <= 15) { c->tvslot[i].len = (uint8_t)L; memcpy(c->tvslot[i].s, c->tvocab[i], L); } else c->tvslot[i].len = 255; }
    return 0;
}
/* FNV-1a over every array of the corpus and the spellings: equal corpora have equal checksums whatever the number of
 * loader threads (tests), and a caller can tell whether two files produced the same index input */
static uint64_t fnv_more(uint64_t h, const void *p, size_t n) { const unsigned char *b = p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; } return h; }
uint64_t cgx_corpus_checksum(const cgx_corpus *c) {
    if (!c) return 0;
    uint64_t h = 0xcbf29ce484222325ull;
    h = fnv_more(h, &c->n, 4); h = fnv_more(h, &c->nt, 4); h = fnv_more(h, &c->nsent, 4); h = fnv_more(h, &c->nlex, 4);
    h = fnv_more(h, c->str, (size_t)c->n * 4); h = fnv_more(h, c->tstr, (size_t)c->nt * 4);
    h = fnv_more(h, c->sentind, ((size_t)c->nsent + 1) * 4); h = fnv_more(h, c->tsentind, ((size_t)c->nsent + 1) * 4);
    if (c->rlp) h = fnv_more(h, c->rlp, (size_t)c->n * 4);
    if (c->ltar) h = fnv_more(h, c->ltar, c->nt);
    if (c->rtar) h = fnv_more(h, c->rtar, c->nt);
    if (c->ltar16) h = fnv_more(h, c->ltar16, (size_t)c->nt * 2);
    if (c->rtar16) h = fnv_more(h, c->rtar16, (size_t)c->nt * 2);
    if (c->lexk) h = fnv_more(h, c->lexk, (size_t)c->nlex * sizeof *c->lexk);
    if (c->lexv) h = fnv_more(h, c->lexv, (size_t)c->nlex * sizeof *c->lexv);
    for (int32_t i = 2; c->svocab && i < c->nsvocab; i++) if (c->svocab[i]) h = fnv_more(h, c->svocab[i], strlen(c->svocab[i]) + 1);
    for (int32_t i = 2; c->tvocab && i < c->ntvocab; i++) if (c->tvocab[i]) h = fnv_more(h, c->tvocab[i], strlen(c->tvocab[i]) + 1);
    return h;
}
void cgx_corpus_free(cgx_corpus *c) {
    if (!c) return;
    for (int32_t i = 0; c->svocab && i < c->nsvocab; i++) free(c->svocab[i]);
    for (int32_t i = 0; c->tvocab && i < c->ntvocab; i++) free(c->tvocab[i]);
    free(c->svocab); free(c->tvocab); free(c->svlen); free(c->tvlen); free(c->svslot); free(c->tvslot); wordmap_free(&c->smap); wordmap_free(&c->tmap);
    free(c->str); free(c->tstr); free(c->sentind); free(c->tsentind); free(c->P); free(c->rlp); free(c->ltar); free(c->rtar); free(c->ltar16); free(c->rtar16);
    free(c->lexk); free(c->lexv); free(c);
}

typedef struct { cgx_corpus *c; const char *path; int what; int rc; char *err; size_t errcap; int32_t nsent; } loadjob;
static void *load_side_job(void *arg) {
    loadjob *j = arg; cgx_corpus *c = j->c;
    if (j->what == 0) j->rc = load_side(j->path, &c->str, &c->n, &c->P, &c->sentind, &c->nsent, &c->svocab, &c->nsvocab, &c->smap);
    else if (j->what == 1) j->rc = load_side(j->path, &c->tstr, &c->nt, NULL, &c->tsentind, &j->nsent, &c->tvocab, &c->ntvocab, &c->tmap);
    else if (j->what == 2) j->rc = load_lex(c, j->path, j->err, j->errcap);
    else j->rc = load_alignment(c, j->path, j->err, j->errcap);
    return NULL;
}
cgx_corpus *cgx_corpus_load(const char *src, const char *tgt, const char *align, const char *lex, char *err, size_t errcap) {
    return cgx_corpus_load_opt(src, tgt, align, lex, 0, err, errcap);
}
cgx_corpus *cgx_corpus_load_opt(const char *src, const char *tgt, const char *align, const char *lex, int flags, char *err, size_t errcap) {
    char dummy[8]; if (!err) { err = dummy; errcap = sizeof dummy; }
    err[0] = 0;
    cgx_corpus *c = calloc(1, sizeof *c);
    if (!c) return NULL;
    c->long_pos = (flags & CGX_CORPUS_LONG_SENTENCES) != 0;
    /* the two sides of the bitext are parsed concurrently, then the lexical table and the alignment (both only
     * read what the first pair produced) */
    int32_t tn = 0; const int trace = getenv("CGX_TRACE") != NULL; double t0 = now_ms();
    loadjob js = {c, src, 0, 0, err, errcap, 0}, jt = {c, tgt, 1, 0, NULL, 0, 0};
    pthread_t th; int threaded = !pthread_create(&th, NULL, load_side_job, &jt);
    load_side_job(&js);
    if (threaded) pthread_join(th, NULL); else load_side_job(&jt);
    if (js.rc) { snprintf(err, errcap, "Can not open reference file \"%s\"", src); goto bad; }
    if (jt.rc) { snprintf(err, errcap, "Can not open reference file \"%s\"", tgt); goto bad; }
    tn = jt.nsent;
    if (trace) fprintf(stderr, "cgx: bitext parsed in %.0f ms (%u + %u tokens)\n", now_ms() - t0, c->n, c->nt);
    t0 = now_ms();
    if (tn != c->nsent) { snprintf(err, errcap, "source has %d lines, target %d", c->nsent, tn); goto bad; }
    {
        char err2[512]; err2[0] = 0;
        loadjob jl = {c, lex, 2, 0, err, errcap, 0}, ja = {c, align, 3, 0, err2, sizeof err2, 0};
        threaded = !pthread_create(&th, NULL, load_side_job, &ja);
        load_side_job(&jl);
        if (threaded) pthread_join(th, NULL); else load_side_job(&ja);
        if (jl.rc != CGX_OK) goto bad;                       /* same precedence as the sequential loader: lexical table first */
        if (ja.rc != CGX_OK) { snprintf(err, errcap, "%s", err2); goto bad; }
    }
    if (trace) fprintf(stderr, "cgx: lexical table + alignment parsed in %.0f ms\n", now_ms() - t0);
    c->svlen = calloc((size_t)c->nsvocab + 1, 4); c->tvlen = calloc((size_t)c->ntvocab + 1, 4); c->maxword = 16;
    if (!c->svlen || !c->tvlen) goto bad;
    for (int32_t i = 2; i < c->nsvocab; i++) { c->svlen[i] = (uint32_t)strlen(c->svocab[i]); if (c->svlen[i] > c->maxword) c->maxword = c->svlen[i]; }
    for (int32_t i = 2; i < c->ntvocab; i++) { c->tvlen[i] = (uint32_t)strlen(c->tvocab[i]); if (c->tvlen[i] > c->maxword) c->maxword = c->tvlen[i]; }
    if (build_word_slots(c)) goto bad;
    {
        const char *files[4] = {src, tgt, align, lex}; struct stat sb;
        for (int k = 0; k < 4; k++) if (!stat(files[k], &sb)) { c->src_size[k] = (uint64_t)sb.st_size + 1; c->src_mtime[k] = (uint64_t)sb.st_mtim.tv_sec * 1000000000ull + (uint64_t)sb.st_mtim.tv_nsec; }   /* size + 1: 0 stays "unknown", an empty file is known */
    }
    return c;
bad:
    cgx_corpus_free(c);
    return NULL;
}

cgx_corpus *cgx_corpus_from_ids(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent, const int32_t *tstr, uint32_t nt,
                                const int32_t *tsentind, const uint8_t *lsrc, const uint8_t *rsrc, const uint8_t *ltar, const uint8_t *rtar,
                                const cgx_lexkey *lexk, const cgx_lexval *lexv, uint32_t nlex) {
    cgx_corpus *c = calloc(1, sizeof *c);
    if (!c) return NULL;
    c->n = n; c->nt = nt; c->nsent = nsent; c->nlex = nlex;
    c->str = malloc((size_t)n * 4); c->tstr = malloc((size_t)nt * 4); c->sentind = malloc(((size_t)nsent + 1) * 4); c->tsentind = malloc(((size_t)nsent + 1) * 4);
    c->P = calloc(n, 1); c->ltar = malloc((size_t)nt + 1); c->rtar = malloc((size_t)nt + 1);
    c->lexk = malloc(((size_t)nlex + 1) * sizeof *c->lexk); c->lexv = malloc(((size_t)nlex + 1) * sizeof *c->lexv);
    if (!c->str || !c->tstr || !c->sentind || !c->tsentind || !c->P || !c->ltar || !c->rtar || !c->lexk || !c->lexv) { cgx_corpus_free(c); return NULL; }
    memcpy(c->str, str, (size_t)n * 4); memcpy(c->tstr, tstr, (size_t)nt * 4);
    memcpy(c->sentind, sentind, ((size_t)nsent + 1) * 4); memcpy(c->tsentind, tsentind, ((size_t)nsent + 1) * 4);
    memcpy(c->ltar, ltar, nt); memcpy(c->rtar, rtar, nt); memcpy(c->lexk, lexk, (size_t)nlex * sizeof *lexk); memcpy(c->lexv, lexv, (size_t)nlex * sizeof *lexv);
    for (int32_t q = 0; q < nsent; q++) for (int32_t i = sentind[q]; i < sentind[q + 1] - 1; i++) c->P[i] = (uint8_t)(i - sentind[q]);
    if (pack_alignment(c, lsrc, rsrc)) { cgx_corpus_free(c); return NULL; }
    c->maxword = 16;
    /* no spellings were given: words print as s<id> / t<id>; build them once */
    int32_t maxs = 0, maxt = 0;
    for (uint32_t i = 0; i < n; i++) if (str[i] > maxs) maxs = str[i];
    for (uint32_t i = 0; i < nt; i++) if (tstr[i] > maxt) maxt = tstr[i];
    c->nsvocab = maxs + 1; c->ntvocab = maxt + 1;
    c->svocab = calloc((size_t)c->nsvocab + 1, sizeof(char *)); c->tvocab = calloc((size_t)c->ntvocab + 1, sizeof(char *));
    c->svlen = calloc((size_t)c->nsvocab + 1, 4); c->tvlen = calloc((size_t)c->ntvocab + 1, 4);
    if (!c->svocab || !c->tvocab || !c->svlen || !c->tvlen) { cgx_corpus_free(c); return NULL; }
    char tmp[24];
    for (int32_t i = 2; i < c->nsvocab; i++) { c->svlen[i] = (uint32_t)snprintf(tmp, sizeof tmp, "s%d", i); c->svocab[i] = strdup(tmp); }
    for (int32_t i = 2; i < c->ntvocab; i++) { c->tvlen[i] = (uint32_t)snprintf(tmp, sizeof tmp, "t%d", i); c->tvocab[i] = strdup(tmp); }
    if (build_word_slots(c)) { cgx_corpus_free(c); return NULL; }
    return c;
}

/* The same for sentence pairs of 255 tokens and more (long-sentence mode): alignment positions as 16-bit words,
 * 0xFFFF = not aligned; source sentences < 1024 tokens, target sentences < 2040.  The corpus then behaves like one loaded with
 * CGX_CORPUS_LONG_SENTENCES (on a corpus the reference accepts it yields the same files as cgx_corpus_from_ids). */
cgx_corpus *cgx_corpus_from_ids16(const int32_t *str, uint32_t n, const int32_t *sentind, int32_t nsent, const int32_t *tstr, uint32_t nt,
                                  const int32_t *tsentind, const uint16_t *lsrc, const uint16_t *rsrc, const uint16_t *ltar, const uint16_t *rtar,
                                  const cgx_lexkey *lexk, const cgx_lexval *lexv, uint32_t nlex) {
    if (!str || !sentind || !tstr || !tsentind || !lsrc || !rsrc || !ltar || !rtar || (nlex && (!lexk || !lexv)) || nsent < 0) return NULL;
    for (int32_t q = 0; q < nsent; q++) if (sentind[q + 1] - sentind[q] - 1 >= LONG_MAX_SRC || tsentind[q + 1] - tsentind[q] - 1 >= LONG_MAX_TGT) return NULL;
    for (uint32_t i = 0; i < n; i++) if ((lsrc[i] != 0xFFFF && lsrc[i] >= LONG_MAX_TGT) || (rsrc[i] != 0xFFFF && rsrc[i] >= LONG_MAX_TGT)) return NULL;
    for (uint32_t i = 0; i < nt; i++) if ((ltar[i] != 0xFFFF && ltar[i] >= LONG_MAX_SRC) || (rtar[i] != 0xFFFF && rtar[i] >= LONG_MAX_SRC)) return NULL;
    /* byte views for the common constructor (not-aligned stays not-aligned, positions beyond a byte saturate: only the 16-bit tables are read in this mode) */
    uint8_t *b = malloc(2 * (size_t)n + 2 * (size_t)nt + 4);
    if (!b) return NULL;
    uint8_t *ls8 = b, *rs8 = b + n, *lt8 = b + 2 * (size_t)n, *rt8 = lt8 + nt;
    for (uint32_t i = 0; i < n; i++) { ls8[i] = (uint8_t)(lsrc[i] > 254 ? 255 : lsrc[i]); rs8[i] = (uint8_t)(rsrc[i] > 254 ? 255 : rsrc[i]); }
    for (uint32_t i = 0; i < nt; i++) { lt8[i] = (uint8_t)(ltar[i] > 254 ? 255 : ltar[i]); rt8[i] = (uint8_t)(rtar[i] > 254 ? 255 : rtar[i]); }
    cgx_corpus *c = cgx_corpus_from_ids(str, n, sentind, nsent, tstr, nt, tsentind, ls8, rs8, lt8, rt8, lexk, lexv, nlex);
    free(b);
    if (!c) return NULL;
    c->long_pos = 1;
    c->ltar16 = malloc(((size_t)nt + 1) * 2); c->rtar16 = malloc(((size_t)nt + 1) * 2);
    int32_t *l = malloc(((size_t)n + 1) * 4), *r = malloc(((size_t)n + 1) * 4);
    int rc = -1;
    if (c->ltar16 && c->rtar16 && l && r) {
        memcpy(c->ltar16, ltar, (size_t)nt * 2); memcpy(c->rtar16, rtar, (size_t)nt * 2);
        for (uint32_t i = 0; i < n; i++) { l[i] = lsrc[i] == 0xFFFF ? -1 : (int32_t)lsrc[i]; r[i] = rsrc[i] == 0xFFFF ? -1 : (int32_t)rsrc[i]; }
        free(c->rlp); c->rlp = NULL;
        rc = pack_alignment_i(c, l, r);                       /* the alignment words again, with position codes (cgx_rules.h) */
    }
    free(l); free(r);
    if (rc) { cgx_corpus_free(c); return NULL; }
    return c;
}
int cgx_corpus_flags(const cgx_corpus *c) { return c && c->long_pos ? CGX_CORPUS_LONG_SENTENCES : 0; }

/* ------------------------------------------------------------------ */
/* on-disk corpus cache: the parsed corpus as one binary file, so that later runs skip the text loaders
 * (the reference re-reads and re-tokenises its four text files on every start; its own index cache is
 * commented out, SuffixArray.c:208-230).  Native byte order; the header carries every count and the
 * loader checks the file size against them. */
/* ------------------------------------------------------------------ */
typedef struct { char magic[8]; uint64_t checksum; uint32_t n, nt, nsent, nlex, nsvocab, ntvocab, maxword, reserved; uint64_t sbytes, tbytes; uint64_t src_size[4], src_mtime[4]; } cachehdr;
static const char CACHE_MAGIC[8] = {'C', 'G', 'X', 'C', 'O', 'R', 'P', '4'};   /* 3: source fingerprint = size + 1 and nanosecond mtime; 4: header field `reserved` = flags (bit 0: long-sentence mode, the 16-bit target tables follow the byte tables) */
#define CACHE_LONG 1u
int cgx_corpus_save(const cgx_corpus *c, const char *path) {
    if (!c || !path || !c->rlp || !c->svocab || !c->tvocab || (c->long_pos && (!c->ltar16 || !c->rtar16))) return CGX_ERR_ARG;
    /* written under a private name and renamed into place: a reader (another --shard process started at the same time)
     * sees either no cache or a complete one, never a file with holes */
    size_t pl = strlen(path); char *tmp = malloc(pl + 40);
    if (!tmp) return CGX_ERR_NOMEM;
    snprintf(tmp, pl + 40, "%s.tmp.%ld", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f) { free(tmp); return CGX_ERR_IO; }
    cachehdr h; memset(&h, 0, sizeof h); memcpy(h.magic, CACHE_MAGIC, 8);
    h.checksum = cgx_corpus_checksum(c); h.n = c->n; h.nt = c->nt; h.nsent = (uint32_t)c->nsent; h.nlex = c->nlex;
    h.nsvocab = (uint32_t)c->nsvocab; h.ntvocab = (uint32_t)c->ntvocab; h.maxword = c->maxword; h.reserved = c->long_pos ? CACHE_LONG : 0u;
    memcpy(h.src_size, c->src_size, sizeof h.src_size); memcpy(h.src_mtime, c->src_mtime, sizeof h.src_mtime);
    for (int32_t i = 0; i < c->nsvocab; i++) h.sbytes += c->svocab[i] ? c->svlen[i] : 0;
    for (int32_t i = 0; i < c->ntvocab; i++) h.tbytes += c->tvocab[i] ? c->tvlen[i] : 0;
    int ok = fwrite(&h, sizeof h, 1, f) == 1;
#define PUT(ptr, count, size) do { if (ok && (count) && fwrite((ptr), (size), (count), f) != (size_t)(count)) ok = 0; } while (0)
    PUT(c->str, c->n, 4); PUT(c->tstr, c->nt, 4); PUT(c->sentind, (size_t)c->nsent + 1, 4); PUT(c->tsentind, (size_t)c->nsent + 1, 4);
    PUT(c->P, c->n, 1); PUT(c->rlp, c->n, 4); PUT(c->ltar, c->nt, 1); PUT(c->rtar, c->nt, 1);
    if (c->long_pos) { PUT(c->ltar16, c->nt, 2); PUT(c->rtar16, c->nt, 2); }
    PUT(c->lexk, c->nlex, sizeof *c->lexk); PUT(c->lexv, c->nlex, sizeof *c->lexv);
    /* spelling lengths (0 for the unused ids 0 and 1), then the spellings back to back */
    for (int32_t i = 0; ok && i < c->nsvocab; i++) { uint32_t L = c->svocab[i] ? c->svlen[i] : 0; PUT(&L, 1, 4); }
    for (int32_t i = 0; ok && i < c->ntvocab; i++) { uint32_t L = c->tvocab[i] ? c->tvlen[i] : 0; PUT(&L, 1, 4); }
    for (int32_t i = 0; ok && i < c->nsvocab; i++) if (c->svocab[i]) PUT(c->svocab[i], c->svlen[i], 1);
    for (int32_t i = 0; ok && i < c->ntvocab; i++) if (c->tvocab[i]) PUT(c->tvocab[i], c->tvlen[i], 1);
#undef PUT
    if (ok && (fflush(f) || fsync(fileno(f)))) ok = 0;
    if (fclose(f)) ok = 0;
    if (ok && rename(tmp, path)) ok = 0;
    if (!ok) (void)unlink(tmp);
    free(tmp);
    return ok ? CGX_OK : CGX_ERR_IO;
}
static int cache_words(FILE *f, int32_t nv, char ***voc_out, uint32_t **len_out) {
    char **voc = calloc((size_t)nv + 1, sizeof *voc); uint32_t *len = calloc((size_t)nv + 1, 4);
    *voc_out = voc; *len_out = len;
    if (!voc || !len) return -1;
    if (nv && fread(len, 4, (size_t)nv, f) != (size_t)nv) return -1;
    return 0;
}
cgx_corpus *cgx_corpus_load_cache(const char *path, char *err, size_t errcap) {
    char dummy[8]; if (!err) { err = dummy; errcap = sizeof dummy; }
    err[0] = 0;
    FILE *f = path ? fopen(path, "rb") : NULL;
    if (!f) { snprintf(err, errcap, "cannot open corpus cache \"%s\"", path ? path : ""); return NULL; }
    cachehdr h; cgx_corpus *c = NULL;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, CACHE_MAGIC, 8)) { snprintf(err, errcap, "\"%s\" is not a corpus cache of this version", path); fclose(f); return NULL; }
    uint64_t want = sizeof h + (uint64_t)h.n * 9 + (uint64_t)h.nt * 6 + ((uint64_t)h.nsent + 1) * 8 + (uint64_t)h.nlex * (sizeof(cgx_lexkey) + sizeof(cgx_lexval))
                  + ((uint64_t)h.nsvocab + h.ntvocab) * 4 + h.sbytes + h.tbytes + ((h.reserved & CACHE_LONG) ? (uint64_t)h.nt * 4 : 0);
    fseek(f, 0, SEEK_END); uint64_t have = (uint64_t)ftell(f); fseek(f, (long)sizeof h, SEEK_SET);
    if (have != want || h.n < 4 || h.nt < 2 || (h.reserved & ~CACHE_LONG)) { snprintf(err, errcap, "corpus cache \"%s\" is truncated or corrupt (%llu bytes, header says %llu)", path, (unsigned long long)have, (unsigned long long)want); fclose(f); return NULL; }
    c = calloc(1, sizeof *c);
    if (!c) { fclose(f); return NULL; }
    c->n = h.n; c->nt = h.nt; c->nsent = (int32_t)h.nsent; c->nlex = h.nlex; c->nsvocab = (int32_t)h.nsvocab; c->ntvocab = (int32_t)h.ntvocab; c->maxword = h.maxword;
    c->long_pos = (h.reserved & CACHE_LONG) != 0;
    int ok = 1;
#define GET(field, count, size) do { if (ok) { (field) = malloc((size_t)(count) * (size) + 16); if (!(field) || ((count) && fread((field), (size), (count), f) != (size_t)(count))) ok = 0; } } while (0)
    GET(c->str, c->n, 4); GET(c->tstr, c->nt, 4); GET(c->sentind, (size_t)c->nsent + 1, 4); GET(c->tsentind, (size_t)c->nsent + 1, 4);
    GET(c->P, c->n, 1); GET(c->rlp, c->n, 4); GET(c->ltar, c->nt, 1); GET(c->rtar, c->nt, 1);
    if (c->long_pos) { GET(c->ltar16, c->nt, 2); GET(c->rtar16, c->nt, 2); }
    GET(c->lexk, c->nlex, sizeof *c->lexk); GET(c->lexv, c->nlex, sizeof *c->lexv);
#undef GET
    if (ok && (cache_words(f, c->nsvocab, &c->svocab, &c->svlen) || cache_words(f, c->ntvocab, &c->tvocab, &c->tvlen))) ok = 0;
    if (ok && (wordmap_init(&c->smap, 1 << 16) || wordmap_init(&c->tmap, 1 << 16))) ok = 0;
    for (int side = 0; ok && side < 2; side++) {
        int32_t nv = side ? c->ntvocab : c->nsvocab; char **voc = side ? c->tvocab : c->svocab; uint32_t *len = side ? c->tvlen : c->svlen; wordmap *map = side ? &c->tmap : &c->smap;
        for (int32_t i = 0; ok && i < nv; i++) {
            if (i < 2 || !len[i]) { if (i >= 2) { snprintf(err, errcap, "corpus cache \"%s\": word %d has no spelling", path, i); ok = 0; } continue; }
            char *w = malloc((size_t)len[i] + 1);
            if (!w || fread(w, 1, len[i], f) != len[i]) { free(w); ok = 0; break; }
            w[len[i]] = 0; voc[i] = w;
            if (wordmap_put(map, w, i)) ok = 0;
        }
    }
    fclose(f);
    memcpy(c->src_size, h.src_size, sizeof c->src_size); memcpy(c->src_mtime, h.src_mtime, sizeof c->src_mtime);
    /* the content is not trusted: ids index the vocabularies and the lexical table on the device, so they are range-checked,
     * the sentence tables must be monotone, and the whole corpus must hash to the checksum the writer stored */
    if (ok) {
        const char *what = NULL;
        if (c->nsvocab < 2 || c->ntvocab < 2 || c->nsent < 0 || (uint64_t)c->nsent + 2 > c->n) what = "counts";
        /* valid ids are < nsvocab; only the closing sentinel (the last token, one past the newest word, Start.cu:321-327) may equal it */
        for (uint32_t i = 0; !what && i < c->n; i++) if (c->str[i] < 1 || c->str[i] > c->nsvocab || (c->str[i] == c->nsvocab && i + 1 != c->n)) what = "source token id";
        for (uint32_t i = 0; !what && i < c->nt; i++) if (c->tstr[i] < 1 || c->tstr[i] > c->ntvocab || (c->tstr[i] == c->ntvocab && i + 1 != c->nt)) what = "target token id";
        if (!what && (c->sentind[0] != 0 || c->tsentind[0] != 0)) what = "sentence table";
        for (int32_t q = 0; !what && q < c->nsent; q++) if (c->sentind[q + 1] <= c->sentind[q] || (uint32_t)c->sentind[q + 1] > c->n || c->tsentind[q + 1] <= c->tsentind[q] || (uint32_t)c->tsentind[q + 1] > c->nt) what = "sentence table";
        for (uint32_t i = 0; !what && i < c->nlex; i++) if (c->lexk[i].src < -1 || c->lexk[i].src >= c->nsvocab || c->lexk[i].tgt < -1 || c->lexk[i].tgt >= c->ntvocab) what = "lexical table id";
        for (uint32_t i = 0; !what && c->long_pos && i < c->nt; i++) if ((c->ltar16[i] != 0xFFFF && c->ltar16[i] >= LONG_MAX_SRC) || (c->rtar16[i] != 0xFFFF && c->rtar16[i] >= LONG_MAX_SRC)) what = "alignment position";
        if (!what && cgx_corpus_checksum(c) != h.checksum) what = "checksum";
        if (what) { snprintf(err, errcap, "corpus cache \"%s\" is corrupt (%s)", path, what); ok = 0; }
    }
    if (ok && build_word_slots(c)) ok = 0;
    if (!ok) { if (!err[0]) snprintf(err, errcap, "cannot read corpus cache \"%s\"", path); cgx_corpus_free(c); return NULL; }
    return c;
}
/* 1 when the corpus (as loaded from a cache) was parsed from exactly these four files as they are now (size and
 * modification time), 0 when one of them has changed or is unknown to the cache, -1 when one cannot be examined */
int cgx_corpus_matches_sources(const cgx_corpus *c, const char *src, const char *tgt, const char *align, const char *lex) {
    if (!c) return -1;
    const char *files[4] = {src, tgt, align, lex}; struct stat sb[4];
    for (int k = 0; k < 4; k++) if (!files[k] || stat(files[k], &sb[k])) return -1;
    for (int k = 0; k < 4; k++) if (!c->src_size[k] || c->src_size[k] != (uint64_t)sb[k].st_size + 1 || c->src_mtime[k] != (uint64_t)sb[k].st_mtim.tv_sec * 1000000000ull + (uint64_t)sb[k].st_mtim.tv_nsec) return 0;
    return 1;
}

int cgx_corpus_upload(cgx_ctx *ctx, const cgx_corpus *c) {
    if (!ctx || !c) return CGX_ERR_ARG;
    cgx_index_host ix; memset(&ix, 0, sizeof ix);
    ix.str = c->str; ix.n = c->n; ix.rlp = c->rlp; ix.tstr = c->tstr; ix.nt = c->nt; ix.ltar = c->ltar; ix.rtar = c->rtar;
    ix.lexk = c->lexk; ix.lexv = c->lexv; ix.nlex = c->nlex; ix.sa = NULL;
    if (c->long_pos) { ix.ltar16 = c->ltar16; ix.rtar16 = c->rtar16; }
    int rc = cgx_upload_index(ctx, &ix);
    if (rc == CGX_OK) rc = cgx_build_sa(ctx);
    if (rc == CGX_OK) rc = cgx_precompute(ctx);
    return rc;
}

/* ------------------------------------------------------------------ */
/* per batch host state                                                */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t *v; uint32_t n, cap; } idlist;
static int idlist_push(idlist *l, uint32_t x) { if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 8; l->v = realloc(l->v, l->cap * 4); if (!l->v) return -1; } l->v[l->n++] = x; return 0; }

typedef struct {                       /* one lexicon line (red_dup_t, ComTypes.h:244) */
    int32_t id; uint32_t rule; int32_t f, fsample, paircount;
    float aa, bb, fscore, fe, ef;
} lexent;
typedef struct { int32_t down, up; } range;

typedef struct {
    const cgx_corpus *c;
    int32_t nq, ntok; int32_t *qoff, *qtok;
    int32_t *lm, *up, *down;
    uint32_t g; cgx_block *blocks;
    idlist *qblocks, *qone, *qtwo;
    uint32_t e1, d1, e2, d2, h1;
    cgx_gappat *p1; cgx_gapsearch *s1; cgx_twogapsearch *s2; int32_t *c2; cgx_hit1 *hits1;
    uint32_t n0, n1, n2, sep1, sep2a, sep2b;
    cgx_rule0 *r0; cgx_rule1 *r1; cgx_rule2 *r2;
    uint32_t *pidx; int32_t *miss;
    lexent *lex0, *lex1, *lex2; uint32_t nl0, nl1, nl2;       /* exact host path only (hash-collision fallback) */
    cgx_lexent *L0, *L1, *L2;                                 /* lexicon lines as produced on the device */
    int L_pinned;                                             /* L* live in the context's pinned arena: not freed here */
    cgx_gappat *p1d; int32_t *c2d; uint32_t *one2;            /* per distinct pattern: symbols of aXb, token c and one-gap id of aXbXc */
    range *rng0, *rng1, *rng2;
    cgx_lextask *tasks; uint32_t ntask;
    int gz_level;                                             /* of the context this batch runs on (the host formatter may run later, on the writer thread) */
    int64_t write_period, write_count;
} batch;

static void batch_free(batch *b) {
    free(b->qoff); free(b->qtok); free(b->lm); free(b->up); free(b->down); free(b->blocks);
    for (int32_t q = 0; q < b->nq; q++) { if (b->qblocks) free(b->qblocks[q].v); if (b->qone) free(b->qone[q].v); if (b->qtwo) free(b->qtwo[q].v); }
    free(b->qblocks); free(b->qone); free(b->qtwo); free(b->p1); free(b->s1); free(b->s2); free(b->c2); free(b->hits1);
    free(b->r0); free(b->r1); free(b->r2); free(b->pidx); free(b->miss); free(b->lex0); free(b->lex1); free(b->lex2);
    free(b->rng0); free(b->rng1); free(b->rng2); free(b->tasks); if (!b->L_pinned) { free(b->L0); free(b->L1); free(b->L2); } free(b->p1d); free(b->c2d); free(b->one2);
}

static int fetch_alloc(cgx_ctx *ctx, const char *name, void **out, size_t elem, uint32_t *count) {
    int64_t bytes = cgx_fetch(ctx, name, NULL, 0);
    if (bytes < 0) return (int)bytes;
    *out = malloc((size_t)bytes + 16);
    if (!*out) return CGX_ERR_NOMEM;
    int64_t got = cgx_fetch(ctx, name, *out, bytes);
    if (got < 0) return (int)got;
    if (count) *count = (uint32_t)((size_t)bytes / elem);
    return CGX_OK;
}

/* the blocks and per-query block lists made by cgx_make_blocks, for the host formatter / host lexicon */
static int fetch_host_blocks(cgx_ctx *ctx, batch *b) {
    int rc; uint32_t g = 0, noff = 0, nids = 0; uint32_t *off = NULL, *ids = NULL;
    free(b->blocks); b->blocks = NULL;
    if ((rc = fetch_alloc(ctx, "blocks", (void **)&b->blocks, sizeof *b->blocks, &g)) != CGX_OK) return rc;
    b->g = g;
    if ((rc = fetch_alloc(ctx, "qb_off", (void **)&off, 4, &noff)) != CGX_OK) return rc;
    if ((rc = fetch_alloc(ctx, "qb_ids", (void **)&ids, 4, &nids)) != CGX_OK) { free(off); return rc; }
    b->qblocks = calloc((size_t)b->nq + 1, sizeof *b->qblocks);
    if (!b->qblocks || noff < (uint32_t)b->nq + 1) { free(off); free(ids); return CGX_ERR_NOMEM; }
    for (int32_t q = 0; q < b->nq && rc == CGX_OK; q++)
        for (uint32_t k = off[q]; k < off[q + 1] && k < nids; k++) if (idlist_push(&b->qblocks[q], ids[k])) { rc = CGX_ERR_NOMEM; break; }
    free(off); free(ids);
    return rc;
}

/* per-query lists of distinct pattern ids, ascending (checkDup loops, SuffixArray.cu:1707-1718, 2085-2096) */
static int pattern_lists(batch *b, const uint32_t *pid, const int32_t *tokpos, uint32_t n, const int32_t *tok2q, idlist **out) {
    idlist *l = calloc((size_t)b->nq + 1, sizeof *l);
    int64_t *seen = malloc(((size_t)b->nq + 1) * 8);
    if (!l || !seen) { free(l); free(seen); return CGX_ERR_NOMEM; }
    *out = l;                                                 /* owned by the batch from here on (batch_free releases the lists) */
    for (int32_t q = 0; q < b->nq; q++) seen[q] = -1;
    for (uint32_t i = 0; i < n; i++) { int32_t q = tok2q[tokpos[i]]; if (seen[q] != (int64_t)pid[i]) { seen[q] = pid[i]; if (idlist_push(&l[q], pid[i])) { free(seen); return CGX_ERR_NOMEM; } } }
    free(seen);
    return CGX_OK;
}

/* ------------------------------------------------------------------ */
/* lexicon: per id group, distinct target sides in first-occurrence order */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t *h; uint32_t *idx; uint32_t *gen; uint32_t cap, cur; } grpmap;
static int grp_init(grpmap *m) { m->cap = 4096; m->cur = 0; m->h = calloc(m->cap, 8); m->idx = calloc(m->cap, 4); m->gen = calloc(m->cap, 4); return m->h && m->idx && m->gen ? 0 : -1; }
static void grp_free(grpmap *m) { free(m->h); free(m->idx); free(m->gen); }

/* target side of a rule as a symbol tuple: words, -1 for [X,1], -2 for [X,2] */
static int target_symbols(const cgx_corpus *c, uint32_t t0, int end, int g1, int g1e, int g2, int g2e, int kind, int32_t *out) {
    int n = 0; uint32_t t1 = t0 + (uint32_t)end, a = t0 + (uint32_t)g1, b = t0 + (uint32_t)g1e, cs = t0 + (uint32_t)g2, ce = t0 + (uint32_t)g2e;
    for (uint32_t jj = t0; jj <= t1 && n < 40; jj++) {
        if (kind >= 1 && jj >= a && jj <= b) { out[n++] = -1; jj = b; }
        else if (kind >= 2 && jj >= cs && jj <= ce) { out[n++] = -2; jj = ce; }
        else out[n++] = c->tstr[jj];
    }
    return n;
}
static uint64_t hash_syms(const int32_t *s, int n) { uint64_t h = 0x9E3779B97F4A7C15ull; for (int i = 0; i < n; i++) { h ^= (uint32_t)s[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 29; } return h; }

typedef struct { const batch *b; int kind; } lexctx;   /* kind 0 contiguous, 1 one gap, 2 two gaps */
static int rule_symbols(const batch *b, int kind, uint32_t rule, int32_t *out) {
    if (kind == 0) return target_symbols(b->c, (uint32_t)b->r0[rule].tar_start, b->r0[rule].tar_end, 0, 0, 0, 0, 0, out);
    if (kind == 1) return target_symbols(b->c, b->r1[rule].tstart, b->r1[rule].end, b->r1[rule].gap1, b->r1[rule].gap1_1, 0, 0, 1, out);
    return target_symbols(b->c, b->r2[rule].tstart, b->r2[rule].end, b->r2[rule].gap1, b->r2[rule].gap1_1, b->r2[rule].gap2, b->r2[rule].gap2_1, 2, out);
}
static uint32_t grp_find(grpmap *m, const batch *b, int kind, const lexent *lex, const int32_t *sym, int n, uint64_t h) {
    uint32_t i = (uint32_t)h & (m->cap - 1); int32_t other[48];
    while (m->gen[i] == m->cur) {
        if (m->h[i] == h) { int k = rule_symbols(b, kind, lex[m->idx[i]].rule, other); if (k == n && !memcmp(other, sym, (size_t)n * 4)) return m->idx[i]; }
        i = (i + 1) & (m->cap - 1);
    }
    return UINT32_MAX;
}
static void grp_add(grpmap *m, uint64_t h, uint32_t idx) { uint32_t i = (uint32_t)h & (m->cap - 1); while (m->gen[i] == m->cur) i = (i + 1) & (m->cap - 1); m->gen[i] = m->cur; m->h[i] = h; m->idx[i] = idx; }

static int marker_fsample(const batch *b, uint32_t one) {          /* ExtractPair.c:895-908 */
    const cgx_gapsearch *s = &b->s1[one];
    int fs = 1 + s->sa_end - s->sa_start;
    if (fs == 1 && b->hits1[s->sa_start].length == 0) { uint32_t pre = b->hits1[s->sa_start].str_position; fs = (int)(1 - b->pidx[2 * pre] + b->pidx[2 * pre + 1] + (uint32_t)b->miss[pre]); }
    return fs;
}
static void scores(lexent *l, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        l[i].aa = -log10f((float)l[i].paircount / (float)l[i].fsample);
        l[i].bb = (float)log10((double)(1 + l[i].paircount));
        l[i].fscore = (float)log10((double)(1 + l[i].fsample));
    }
}
static const cgx_gappat *pat_of(const batch *b, uint32_t one) { return b->p1d ? &b->p1d[one] : &b->p1[b->s1[one].position]; }
static int32_t c_of(const batch *b, uint32_t two) { return b->c2d ? b->c2d[two] : b->c2[b->s2[two].position]; }
static uint32_t one_of(const batch *b, uint32_t two) { return b->one2 ? b->one2[two] : b->s2[two].blockid; }
static int pattern_src(const cgx_gappat *p, int32_t *src) { int n = 0; for (int j = 0; j < p->number; j++) if (p->pat[j] >= 0) src[n++] = p->pat[j]; return n; }
static int block_src(const batch *b, uint32_t bn, int32_t *src) { const cgx_block *k = &b->blocks[bn]; for (int s = 0; s < k->matchlen; s++) src[s] = b->c->str[k->string_start + s]; return k->matchlen; }

static range *make_ranges_dev(const cgx_lexent *l, uint32_t nl, uint32_t nid) {
    range *r = malloc(((size_t)nid + 1) * sizeof *r);
    if (!r) return NULL;
    for (uint32_t i = 0; i < nid; i++) r[i].down = r[i].up = -1;
    for (uint32_t i = 0; i < nl; i++) { if (i == 0 || l[i].id != l[i - 1].id) r[l[i].id].down = (int32_t)i; r[l[i].id].up = (int32_t)i; }
    return r;
}
static range *make_ranges(const lexent *l, uint32_t nl, uint32_t nid) {
    range *r = malloc(((size_t)nid + 1) * sizeof *r);
    if (!r) return NULL;
    for (uint32_t i = 0; i < nid; i++) r[i].down = r[i].up = -1;
    for (uint32_t i = 0; i < nl; i++) { if (i == 0 || l[i].id != l[i - 1].id) r[l[i].id].down = (int32_t)i; r[l[i].id].up = (int32_t)i; }
    return r;
}

/* converted id of rule i (ExtractPair.c:724-728, 1000-1006) */
static uint32_t rule_cid(const batch *b, int kind, uint32_t i) {
    const uint32_t G = b->g, D2 = b->d2;
    if (kind == 0) return (uint32_t)b->r0[i].block;
    if (kind == 1) return i < b->sep1 ? (uint32_t)b->r1[i].id : 2 * G + (uint32_t)b->r1[i].id;
    return i < b->sep2a ? (uint32_t)b->r2[i].id : i < b->sep2b ? G + (uint32_t)b->r2[i].id : G + D2 + (uint32_t)b->r2[i].id;
}
/* source words of the group with converted id cid (for the MaxLex task) */
static int group_src(const batch *b, int kind, uint32_t cid, int32_t *src) {
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2;
    if (kind == 0) return block_src(b, cid, src);
    if (kind == 1) return cid < 2 * G ? block_src(b, cid < G ? cid : cid - G, src) : pattern_src(pat_of(b, cid - 2 * G), src);
    if (cid < G) return block_src(b, cid, src);
    if (cid < G + D2) { int n = pattern_src(pat_of(b, one_of(b, cid - G)), src); src[n++] = c_of(b, cid - G); return n; }
    return pattern_src(pat_of(b, cid < G + D2 + D1 ? cid - G - D2 : cid - G - D2 - D1), src);
}
static int group_fsample(const batch *b, int kind, uint32_t cid) {
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2; int fs;
    if (kind == 0) fs = 1 + b->blocks[cid].end - b->blocks[cid].start;
    else if (kind == 1) { if (cid < 2 * G) { uint32_t r = cid >= G ? cid - G : cid; fs = 1 + b->blocks[r].end - b->blocks[r].start; } else fs = marker_fsample(b, cid - 2 * G); }
    else if (cid < G) fs = 1 + b->blocks[cid].end - b->blocks[cid].start;
    else if (cid < G + D2) fs = 1 + b->s2[cid - G].sa_end - b->s2[cid - G].sa_start;
    else fs = marker_fsample(b, cid < G + D2 + D1 ? cid - G - D2 : cid - G - D2 - D1);
    return fs > SAMPLER ? SAMPLER : fs;
}

typedef struct {
    const batch *b; int kind; uint32_t lo, hi;       /* rule range, aligned to group starts */
    lexent *lex; uint32_t nl, cap; cgx_lextask *tasks; int rc;
} lexjob;
static void *lex_worker(void *arg) {
    lexjob *j = arg; const batch *b = j->b; const int kind = j->kind;
    grpmap gm; int32_t sym[48], src[8]; int nsrc = 0; uint32_t cid = 0; int fsample = 0;
    j->rc = CGX_ERR_NOMEM; j->nl = 0; j->cap = j->hi - j->lo + 1;
    j->lex = malloc((size_t)j->cap * sizeof *j->lex); j->tasks = malloc((size_t)j->cap * sizeof *j->tasks);
    if (!j->lex || !j->tasks || grp_init(&gm)) return NULL;
    for (uint32_t i = j->lo; i < j->hi; i++) {
        uint32_t c = rule_cid(b, kind, i);
        if (i == j->lo || c != cid) { cid = c; gm.cur++; nsrc = group_src(b, kind, cid, src); fsample = group_fsample(b, kind, cid); }
        int n = rule_symbols(b, kind, i, sym); uint64_t h = hash_syms(sym, n);
        uint32_t hit = grp_find(&gm, b, kind, j->lex, sym, n, h);
        if (hit != UINT32_MAX) { j->lex[hit].paircount++; continue; }
        lexent *e = &j->lex[j->nl]; memset(e, 0, sizeof *e);
        e->id = (int32_t)cid; e->rule = i; e->paircount = 1; e->fsample = fsample;
        cgx_lextask *t = &j->tasks[j->nl]; memset(t, 0, sizeof *t);
        t->lexid = j->nl; t->nsrc = (uint8_t)nsrc; for (int q = 0; q < nsrc; q++) t->src[q] = src[q];
        if (kind == 0) { t->tstart = (uint32_t)b->r0[i].tar_start; t->end = b->r0[i].tar_end; }
        else if (kind == 1) { t->tstart = b->r1[i].tstart; t->end = b->r1[i].end; t->gap1 = b->r1[i].gap1; t->gap1_1 = b->r1[i].gap1_1; }
        else { t->tstart = b->r2[i].tstart; t->end = b->r2[i].end; t->gap1 = b->r2[i].gap1; t->gap1_1 = b->r2[i].gap1_1; t->gap2 = b->r2[i].gap2; t->gap2_1 = b->r2[i].gap2_1; }
        grp_add(&gm, h, j->nl); j->nl++;
    }
    grp_free(&gm);
    j->rc = CGX_OK;
    return NULL;
}
/* f = number of rules in the entry's id group (fsample_arr, ExtractPair.c:718-730) */
static void fill_group_sizes(const batch *b, int kind, lexent *lex, uint32_t nl, uint32_t nrules) {
    uint32_t i = 0;
    while (i < nl) {
        uint32_t k = i; while (k < nl && lex[k].id == lex[i].id) k++;
        uint32_t first = lex[i].rule, r = first, cid = (uint32_t)lex[i].id;   /* a group's first entry is made by its first rule */
        while (r < nrules && rule_cid(b, kind, r) == cid) r++;
        for (uint32_t q = i; q < k; q++) lex[q].f = (int32_t)(r - first);
        i = k;
    }
}
/* CPUs this process may actually use: online CPUs, its affinity mask and the cgroup CPU quota (cpu.max), whichever is smallest */
static int usable_cpus(void) {
    long c = sysconf(_SC_NPROCESSORS_ONLN); int n = c > 0 ? (int)c : 1;
    cpu_set_t set; if (sched_getaffinity(0, sizeof set, &set) == 0) { int k = CPU_COUNT(&set); if (k > 0 && k < n) n = k; }
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) { char q[64]; long long per = 0; if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") && per > 0) { long long k = (atoll(q) + per - 1) / per; if (k > 0 && k < n) n = (int)k; } fclose(f); }
    return n;
}
static int nthreads_host(void) {
    const char *e = getenv("CGX_THREADS"); int n = e ? atoi(e) : 0;
    if (n <= 0) { int c = usable_cpus(); n = c > 16 ? 16 : c; }
    return n < 1 ? 1 : n > 64 ? 64 : n;
}
/* The file phase is a page-cache copy that keeps every writer thread busy.  Under a CPU quota the whole process is frozen
 * for the rest of the scheduling period once the quota is used up -- the thread that feeds the GPU included -- so two CPUs'
 * worth of quota are left to it and to the HIP runtime's own threads: on the 16-CPU boxes 14 writers finish a file phase in
 * 348-357 ms, 16 in 348-394 ms with the GPU stages stretched by the freezes (tools/gpu_threads_ab.sh). */
static int writer_threads(void) {
    const char *e = getenv("CGX_THREADS"); int n = e ? atoi(e) : 0;
    if (n <= 0) { int c = usable_cpus(); n = c >= 8 ? c - 2 : c; if (n > 16) n = 16; }
    return n < 1 ? 1 : n > 64 ? 64 : n;
}
static int build_lex_kind(batch *b, int kind, uint32_t nrules, lexent **out, uint32_t *nout, uint32_t nid, range **rng, uint32_t *taskcap) {
    int nt = nthreads_host(); if ((uint32_t)nt > nrules / 4096 + 1) nt = (int)(nrules / 4096 + 1);
    lexjob jobs[64]; pthread_t th[64]; uint32_t cut[65]; cut[0] = 0;
    for (int t = 1; t < nt; t++) {                    /* cut points moved forward to the next group start */
        uint32_t c = (uint32_t)((uint64_t)nrules * (uint64_t)t / (uint64_t)nt);
        if (c < cut[t - 1]) c = cut[t - 1];
        while (c > 0 && c < nrules && rule_cid(b, kind, c) == rule_cid(b, kind, c - 1)) c++;
        cut[t] = c;
    }
    cut[nt] = nrules;
    for (int t = 0; t < nt; t++) { memset(&jobs[t], 0, sizeof jobs[t]); jobs[t].b = b; jobs[t].kind = kind; jobs[t].lo = cut[t]; jobs[t].hi = cut[t + 1]; }
    for (int t = 1; t < nt; t++) if (pthread_create(&th[t], NULL, lex_worker, &jobs[t])) return CGX_ERR_NOMEM;
    lex_worker(&jobs[0]);
    for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
    uint32_t total = 0;
    for (int t = 0; t < nt; t++) { if (jobs[t].rc != CGX_OK) return jobs[t].rc; total += jobs[t].nl; }
    lexent *lex = malloc(((size_t)total + 1) * sizeof *lex);
    if (!lex) return CGX_ERR_NOMEM;
    if (b->ntask + total > *taskcap) { *taskcap = b->ntask + total + 1; b->tasks = realloc(b->tasks, (size_t)*taskcap * sizeof *b->tasks); if (!b->tasks) return CGX_ERR_NOMEM; }
    uint32_t o = 0;
    for (int t = 0; t < nt; t++) {
        memcpy(lex + o, jobs[t].lex, (size_t)jobs[t].nl * sizeof *lex);
        for (uint32_t k = 0; k < jobs[t].nl; k++) { cgx_lextask x = jobs[t].tasks[k]; x.lexid += o; b->tasks[b->ntask + o + k] = x; }
        o += jobs[t].nl; free(jobs[t].lex); free(jobs[t].tasks);
    }
    b->ntask += total;
    fill_group_sizes(b, kind, lex, total, nrules);
    scores(lex, total);
    *out = lex; *nout = total;
    return (*rng = make_ranges(lex, total, nid)) ? CGX_OK : CGX_ERR_NOMEM;
}
static int build_lexicons(batch *b) {
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2; uint32_t taskcap = 0; int rc;
    /* task order = one gap, two gaps, contiguous (the shared lexicalTaskCounter, ExtractPair.cu:3708-3868) */
    if ((rc = build_lex_kind(b, 1, b->n1, &b->lex1, &b->nl1, 2 * G + D1, &b->rng1, &taskcap)) != CGX_OK) return rc;
    if ((rc = build_lex_kind(b, 2, b->n2, &b->lex2, &b->nl2, G + 2 * D1 + D2, &b->rng2, &taskcap)) != CGX_OK) return rc;
    return build_lex_kind(b, 0, b->n0, &b->lex0, &b->nl0, G, &b->rng0, &taskcap);
}

/* ------------------------------------------------------------------ */
/* grammar writer                                                      */
/* ------------------------------------------------------------------ */
typedef struct { char *p; size_t n, cap; } sbuf;
static int sb_need(sbuf *s, size_t extra) { if (s->n + extra + 1 > s->cap) { size_t nc = s->cap ? s->cap : 1 << 16; while (nc < s->n + extra + 1) nc *= 2; s->p = realloc(s->p, nc); if (!s->p) return -1; s->cap = nc; } return 0; }
/* The formatter below appends through a raw cursor; the caller reserves line_max() bytes per line first. */
/* used by tests/cpu_sim/f6test.c, which includes this file to check put_f6 against printf */
static int sb_f6(sbuf *s, float x) __attribute__((unused));
static size_t line_max(const cgx_corpus *c) { return 512 + 32 * ((size_t)c->maxword + 8); }
#define PUT_LIT(p, lit) do { memcpy((p), (lit), sizeof(lit) - 1); (p) += sizeof(lit) - 1; } while (0)
static const char DIG2[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
static inline char *put_uint(char *p, uint32_t v) {
    char t[12]; int k = 12;
    while (v >= 100) { uint32_t r = v % 100; v /= 100; k -= 2; memcpy(t + k, DIG2 + 2 * r, 2); }
    if (v >= 10) { k -= 2; memcpy(t + k, DIG2 + 2 * v, 2); } else t[--k] = (char)('0' + v);
    memcpy(p, t + k, (size_t)(12 - k));
    return p + (12 - k);
}
static inline char *put_word(char *p, const cgx_corpus *c, int target, int32_t id) {
    int32_t nv = target ? c->ntvocab : c->nsvocab;
    if (id >= 0 && id < nv) {
        const struct wslot *w = &(target ? c->tvslot : c->svslot)[id];
        if (w->len != 255) { memcpy(p, w->s, 15); return p + w->len; }          /* 15 bytes copied, len kept: the line buffer has headroom */
        char **voc = target ? c->tvocab : c->svocab;
        if (voc[id]) { uint32_t L = (target ? c->tvlen : c->svlen)[id]; memcpy(p, voc[id], L); return p + L; }
    }
    *p++ = target ? 't' : 's';                            /* no spelling known */
    if (id < 0) { *p++ = '-'; id = -id; }
    return put_uint(p, (uint32_t)id);
}
static char *put_block(char *p, const batch *b, uint32_t bn) {
    const cgx_block *k = &b->blocks[bn];
    for (int i = 0; i < k->matchlen; i++) { if (i) *p++ = ' '; p = put_word(p, b->c, 0, b->c->str[k->string_start + i]); }
    return p;
}
static char *put_pattern(char *p, const batch *b, uint32_t one, char gapdigit, int lead_space) {
    const cgx_gappat *pt = pat_of(b, one);
    for (int j = 0; j < pt->number; j++) {
        if (j || lead_space) *p++ = ' ';
        if (pt->pat[j] >= 0) p = put_word(p, b->c, 0, pt->pat[j]); else { PUT_LIT(p, "[X,"); *p++ = gapdigit; *p++ = ']'; }
    }
    return p;
}
/* source side of a rule group from its converted id (ExtractPair.c:743-796, 1021-1123) */
static char *put_source(char *p, const batch *b, int kind, uint32_t cid) {
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2;
    if (kind == 0) return put_block(p, b, cid);
    if (kind == 1) {
        if (cid < G) { PUT_LIT(p, "[X,1] "); return put_block(p, b, cid); }
        if (cid < 2 * G) { p = put_block(p, b, cid - G); PUT_LIT(p, " [X,1]"); return p; }
        return put_pattern(p, b, cid - 2 * G, '1', 0);
    }
    if (cid < G) { PUT_LIT(p, "[X,1] "); p = put_block(p, b, cid); PUT_LIT(p, " [X,2]"); return p; }
    if (cid < G + D2) { p = put_pattern(p, b, one_of(b, cid - G), '1', 0); PUT_LIT(p, " [X,2] "); return put_word(p, b->c, 0, c_of(b, cid - G)); }
    if (cid < G + D2 + D1) { PUT_LIT(p, "[X,1]"); return put_pattern(p, b, cid - G - D2, '2', 1); }
    p = put_pattern(p, b, cid - G - D2 - D1, '1', 0); PUT_LIT(p, " [X,2]"); return p;
}
static char *put_target(char *p, const batch *b, const cgx_lexent *e) {
    const cgx_corpus *c = b->c; int first = 1;
    uint32_t t0 = e->tstart, t1 = t0 + e->end, ga = t0 + e->gap1, gb = t0 + e->gap1_1, gc = t0 + e->gap2, gd = t0 + e->gap2_1;
    for (uint32_t jj = t0; jj <= t1; jj++) {             /* ExtractPair.c:813-837, 1141-1163 */
        if (!first) *p++ = ' ';
        first = 0;
        if (e->kind >= 1 && jj >= ga && jj <= gb) { PUT_LIT(p, "[X,1]"); jj = gb; }
        else if (e->kind >= 2 && jj >= gc && jj <= gd) { PUT_LIT(p, "[X,2]"); jj = gd; }
        else p = put_word(p, c, 1, c->tstr[jj]);
    }
    return p;
}
/* feature values depend only on (paircount, fsample) <= 300 each (ExtractPair.c:652-656): tabulated once with the host libm */
#define TABN 302
static float g_aa[TABN][TABN], g_bb[TABN], g_fs[TABN]; static int g_tab_ready;
static char g_aa_s[TABN][TABN][12], g_bb_s[TABN][12], g_fs_s[TABN][12];      /* the same values already formatted with %f */
static uint8_t g_aa_l[TABN][TABN], g_bb_l[TABN], g_fs_l[TABN];
static inline char *put_f6(char *p, float x);
static int sb_f6(sbuf *s, float x) { if (sb_need(s, 64)) return -1; s->n = (size_t)(put_f6(s->p + s->n, x) - s->p); return 0; }
static void score_tables(void) {
    if (g_tab_ready) return;
    for (int p = 0; p < TABN; p++) {
        g_bb[p] = (float)log10((double)(1 + p)); g_fs[p] = (float)log10((double)(1 + p));
        g_bb_l[p] = (uint8_t)(put_f6(g_bb_s[p], g_bb[p]) - g_bb_s[p]); g_fs_l[p] = (uint8_t)(put_f6(g_fs_s[p], g_fs[p]) - g_fs_s[p]);
        for (int f = 1; f < TABN; f++) { g_aa[p][f] = -log10f((float)p / (float)f); g_aa_l[p][f] = (uint8_t)(put_f6(g_aa_s[p][f], g_aa[p][f]) - g_aa_s[p][f]); }
    }
    g_tab_ready = 1;
}
/* "%f" of a float: the value times 10^6 is exact in double (24 + 14 significant bits), so
 * rounding it to nearest-even is exactly what printf does; odd cases fall back to snprintf. */
static inline char *put_f6(char *p, float x) {
    double v = (double)x;
    if (!(fabs(v) < 1e12)) return p + snprintf(p, 64, "%f", v);
    if (signbit(v)) { *p++ = '-'; v = -v; }
    uint64_t m = (uint64_t)llrint(v * 1e6);                 /* current rounding mode = to nearest even, like printf */
    uint32_t fp = (uint32_t)(m % 1000000u); uint64_t ip = m / 1000000u;
    if (ip < 0xFFFFFFFFull) p = put_uint(p, (uint32_t)ip);
    else { char tmp[24]; int k = 0; do { tmp[k++] = (char)('0' + ip % 10); ip /= 10; } while (ip); while (k) *p++ = tmp[--k]; }
    *p++ = '.';
    memcpy(p, DIG2 + 2 * (fp / 10000u), 2); memcpy(p + 2, DIG2 + 2 * (fp / 100u % 100u), 2); memcpy(p + 4, DIG2 + 2 * (fp % 100u), 2);
    return p + 6;
}
static int emit_range(sbuf *s, const batch *b, int kind, const cgx_lexent *lex, const range *rng, uint32_t id, uint64_t *lines) {
    if (rng[id].down == -1 || rng[id].up == -1) return 0;
    const size_t lmax = line_max(b->c);
    char srcbuf[4096]; size_t srclen = 0;                 /* every line of the range shares the source side */
    if (lmax < sizeof srcbuf) srclen = (size_t)(put_source(srcbuf, b, kind, (uint32_t)lex[rng[id].down].id) - srcbuf);
    for (int32_t i = rng[id].down; i <= rng[id].up; i++) {
        const cgx_lexent *e = &lex[i];
        if (i + 8 <= rng[id].up) __builtin_prefetch(&b->c->tstr[lex[i + 8].tstart]);   /* the target words sit at random places of a GB-sized array */
        if (i + 3 <= rng[id].up) { const cgx_lexent *n = &lex[i + 3]; const int32_t *tw = &b->c->tstr[n->tstart]; __builtin_prefetch(&b->c->tvslot[tw[0]]); __builtin_prefetch(&b->c->tvslot[tw[n->end]]); }
        if (sb_need(s, lmax)) return -1;
        char *p = s->p + s->n;
        const int tab = e->paircount < TABN && e->fsample < TABN && e->fsample > 0 && e->paircount > 0;
        const int pc = e->paircount, f = e->fsample;
        PUT_LIT(p, "[X] ||| ");
        if (srclen) { memcpy(p, srcbuf, srclen); p += srclen; } else p = put_source(p, b, kind, (uint32_t)e->id);
        PUT_LIT(p, " ||| "); p = put_target(p, b, e);
        PUT_LIT(p, " ||| EgivenFCoherent=");
        if (tab) { memcpy(p, g_aa_s[pc][f], 12); p += g_aa_l[pc][f]; PUT_LIT(p, " SampleCountF="); memcpy(p, g_fs_s[f], 12); p += g_fs_l[f]; PUT_LIT(p, " CountEF="); memcpy(p, g_bb_s[pc], 12); p += g_bb_l[pc]; }
        else {
            p = put_f6(p, -log10f((float)e->paircount / (float)e->fsample)); PUT_LIT(p, " SampleCountF="); p = put_f6(p, (float)log10((double)(1 + e->fsample)));
            PUT_LIT(p, " CountEF="); p = put_f6(p, (float)log10((double)(1 + e->paircount)));
        }
        PUT_LIT(p, " MaxLexFgivenE="); p = put_f6(p, e->fe); PUT_LIT(p, " MaxLexEgivenF="); p = put_f6(p, e->ef);
        PUT_LIT(p, " IsSingletonF="); *p++ = e->f == 1 ? '1' : '0'; PUT_LIT(p, " IsSingletonFE="); *p++ = e->paircount == 1 ? '1' : '0'; *p++ = '\n';
        s->n = (size_t)(p - s->p);
        (*lines)++;
    }
    return 0;
}
static int g_diag_format_only;
typedef struct { const batch *b; const char *outdir; int32_t first; int32_t *next; uint64_t lines; int rc, gz; int64_t period, count; } writejob;
/* options "write_period" / "write_count": with a period, only queries g (index in the whole query list) with g % period < count get a file */
static int file_selected(int64_t period, int64_t count, int64_t g) { return period <= 0 || g % period < count; }
static void *write_worker(void *arg) {
    writejob *w = arg; const batch *b = w->b;
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2;
    sbuf s; memset(&s, 0, sizeof s); char fn[4096]; uint64_t *lines = &w->lines;
    w->rc = CGX_OK;
    for (;;) {
        int32_t q = __atomic_fetch_add(w->next, 1, __ATOMIC_RELAXED);
        if (q >= b->nq) break;
        s.n = 0; int bad = 0;
        for (uint32_t k = 0; !bad && k < b->qblocks[q].n; k++) {
            uint32_t p = b->qblocks[q].v[k];
            bad = emit_range(&s, b, 1, b->L1, b->rng1, p + G, lines) || emit_range(&s, b, 1, b->L1, b->rng1, p, lines) ||
                  emit_range(&s, b, 2, b->L2, b->rng2, p, lines) || emit_range(&s, b, 0, b->L0, b->rng0, p, lines);
        }
        for (uint32_t k = 0; !bad && b->qone && k < b->qone[q].n; k++) {
            uint32_t id = b->qone[q].v[k];
            bad = emit_range(&s, b, 1, b->L1, b->rng1, 2 * G + id, lines) || emit_range(&s, b, 2, b->L2, b->rng2, G + D2 + id, lines) ||
                  emit_range(&s, b, 2, b->L2, b->rng2, G + D2 + D1 + id, lines);
        }
        for (uint32_t k = 0; !bad && b->qtwo && k < b->qtwo[q].n; k++) bad = emit_range(&s, b, 2, b->L2, b->rng2, G + b->qtwo[q].v[k], lines);
        if (bad) { w->rc = CGX_ERR_NOMEM; break; }
        snprintf(fn, sizeof fn, "%s/grammar.%d.s", w->outdir, w->first + q);
        /* overwrite in place and cut to length: same bytes as fopen(fn,"w"), but an existing file keeps its pages */
        if (g_diag_format_only || !file_selected(w->period, w->count, (int64_t)w->first + q)) continue;                   /* diagnostic (CGX_DIAG_FORMAT_ONLY=1): measure formatting without the file system */
        if (w->gz) {
            char mode[8]; snprintf(mode, sizeof mode, "wb%d", w->gz); strcat(fn, ".gz");
            gzFile f = gzopen(fn, mode);
            int bad_gz = !f || (s.n && gzwrite(f, s.p, (unsigned)s.n) != (int)s.n);
            if (f && gzclose(f) != Z_OK) bad_gz = 1;
            if (bad_gz) { w->rc = CGX_ERR_IO; break; }
            continue;
        }
        int fd = open(fn, O_WRONLY | O_CREAT, 0644);
        if (fd < 0) { w->rc = CGX_ERR_IO; break; }
        size_t off = 0; int bad_io = 0;
        while (off < s.n) { ssize_t k = write(fd, s.p + off, s.n - off); if (k <= 0) { bad_io = 1; break; } off += (size_t)k; }
        if (!bad_io && ftruncate(fd, (off_t)s.n)) bad_io = 1;
        close(fd);
        if (bad_io) { w->rc = CGX_ERR_IO; break; }
    }
    free(s.p);
    return NULL;
}
/* one file per query (PrintResults.c:434-446); queries are independent, so a pool of host threads formats them */
static int write_grammars(const batch *b, const char *outdir, int32_t first, uint64_t *lines) {
    int nt = nthreads_host(); if (nt > b->nq) nt = b->nq > 0 ? b->nq : 1;
    { const char *e = getenv("CGX_DIAG_FORMAT_ONLY"); g_diag_format_only = e && *e == '1'; }
    writejob jobs[64]; pthread_t th[64]; int32_t next = 0;
    for (int t = 0; t < nt; t++) { jobs[t].b = b; jobs[t].outdir = outdir; jobs[t].first = first; jobs[t].next = &next; jobs[t].lines = 0; jobs[t].rc = CGX_OK; jobs[t].gz = b->gz_level; jobs[t].period = b->write_period; jobs[t].count = b->write_count; }
    for (int t = 1; t < nt; t++) if (pthread_create(&th[t], NULL, write_worker, &jobs[t])) return CGX_ERR_NOMEM;
    write_worker(&jobs[0]);
    for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nt; t++) { if (jobs[t].rc != CGX_OK) return jobs[t].rc; *lines += jobs[t].lines; }
    return CGX_OK;
}

/* ------------------------------------------------------------------ */
/* the whole path for one batch of queries                             */
/* ------------------------------------------------------------------ */
/* ---- grammar text formatted on the device: the host only moves bytes into files ---- */
static int ensure_vocab(cgx_ctx *ctx, const cgx_corpus *c) {
    if (cgx__get_vocab_owner(ctx) == (const void *)c) return CGX_OK;
    int rc = CGX_ERR_NOMEM;
    uint32_t ns = (uint32_t)c->nsvocab, nt = (uint32_t)c->ntvocab;
    uint32_t *soff = malloc(((size_t)ns + 1) * 4), *toff = malloc(((size_t)nt + 1) * 4);
    if (!soff || !toff) return rc;
    size_t sb = 0, tbytes = 0;
    for (uint32_t i = 0; i < ns; i++) { soff[i] = (uint32_t)sb; sb += c->svocab && c->svocab[i] ? c->svlen[i] : 0; }
    soff[ns] = (uint32_t)sb;
    for (uint32_t i = 0; i < nt; i++) { toff[i] = (uint32_t)tbytes; tbytes += c->tvocab && c->tvocab[i] ? c->tvlen[i] : 0; }
    toff[nt] = (uint32_t)tbytes;
    char *sp = malloc(sb + 1), *tp = malloc(tbytes + 1);
    if (sp && tp) {
        for (uint32_t i = 0; i < ns; i++) if (soff[i + 1] > soff[i]) memcpy(sp + soff[i], c->svocab[i], soff[i + 1] - soff[i]);
        for (uint32_t i = 0; i < nt; i++) if (toff[i + 1] > toff[i]) memcpy(tp + toff[i], c->tvocab[i], toff[i + 1] - toff[i]);
        rc = cgx_upload_vocab(ctx, sp, soff, ns, tp, toff, nt);
        if (rc == CGX_OK) { score_tables(); rc = cgx_upload_score_tables(ctx, &g_aa[0][0], g_bb, g_fs); }
        if (rc == CGX_OK) cgx__set_vocab_owner(ctx, c);
    }
    free(sp); free(tp); free(soff); free(toff);
    return rc;
}
/* ---- writer -------------------------------------------------------------------------------------------------------
 * The device hands over the UNIQUE text of the batch (every lexicon line once) and, per query, the list of pieces of it
 * that make up its grammar file (cgx_text_info / cgx_text_segments_begin).  A batch's output then takes two phases:
 *   1. DMA: unique text + piece lists -> page-locked host memory (enqueued by the submitting thread on the context's
 *      side streams; no CPU work);
 *   2. host threads assemble the files: one pwritev per <= IOV_MAX pieces straight from the unique text into the page
 *      cache (a phrase that occurs in thousands of queries has its rules formatted once and sent over PCIe once).
 * Two host buffer sets alternate, so with async_write the DMA of batch k+1 runs while the files of batch k are being
 * assembled, and both overlap the GPU stages of batch k+2.  File phases run strictly in submission order. */
#include <sys/uio.h>
#include <limits.h>
#ifndef IOV_MAX
#define IOV_MAX 1024
#endif
#define MAX_WRITERS 64
#define COPY_PIECE ((uint64_t)64 << 20)                   /* unique text is copied in pieces of this size, round robin over the copy streams */
#define READERS_PER_SLOT 4                                /* reader ids of text slot s: s*4 .. s*4+2 text pieces, s*4+3 the piece lists */
typedef struct { char *utext; uint64_t utext_cap; uint64_t *segoff, *qseg; uint32_t *seglen, *trl; uint64_t segoff_cap, seglen_cap, qseg_cap, trl_cap; } hostbuf;
struct pending;
typedef struct {
    pthread_mutex_t m; pthread_cond_t cv;
    uint64_t next_seq, done_seq;                           /* file phases finish in submission order: done_seq counts them */
    struct pending *inflight[2]; int n;
    hostbuf hb[2];
} wstate;
typedef struct pending {
    wstate *ws; uint64_t seq; cgx_ctx *ctx; pthread_t th; int rc;
    batch *b;                                              /* host-formatter path: the batch whose lexicon the threads format */
    int dev, slot, hb; int32_t nq, first; char *outdir;
    uint64_t ubytes, nseg, fbytes, lines; double ms, wait_ms, file_ms;
    int gz, members; int64_t period, count;                 /* the options as they were when the batch was submitted (the writer thread runs later, beside the next batch) */
} pending;
static wstate *get_wstate(cgx_ctx *ctx) {
    wstate *ws = cgx__get_host_state(ctx);
    if (ws) return ws;
    ws = calloc(1, sizeof *ws);
    if (!ws) return NULL;
    pthread_mutex_init(&ws->m, NULL); pthread_cond_init(&ws->cv, NULL);
    cgx__set_host_state(ctx, ws);
    return ws;
}
int cgx__host_busy(cgx_ctx *ctx) { wstate *ws = cgx__get_host_state(ctx); return ws ? ws->n : 0; }
static int pinned_reserve(void **p, uint64_t *cap, uint64_t need, size_t elem) {
    if (need <= *cap) return 0;
    uint64_t nc = need + need / 4 + 1024;
    void *np = cgx_pinned_alloc((size_t)nc * elem);
    if (!np) return -1;
    cgx_pinned_free(*p); *p = np; *cap = nc;
    return 0;
}
/* Writer threads copy 30+ GB per batch from the page-locked unique text into the page cache: keep them on
 * the CPUs of the GPU's NUMA node, where the DMA lands.  Best effort (the list comes from sysfs). */
static void pin_to_device_node(cgx_ctx *ctx) {
    if (!cgx__option(ctx, "numa_pin")) return;
    char list[1024]; cgx__device_cpulist(ctx, list, sizeof list);
    cpu_set_t set; CPU_ZERO(&set); int any = 0;
    for (char *p = list; *p;) {
        char *e; long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); if (e == p) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { if (c >= 0) { CPU_SET((int)c, &set); any = 1; } }
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    if (any) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}
typedef struct {
    cgx_ctx *ctx; int tid, rc; int32_t nq, first; const char *outdir; int32_t *next_q;
    const char *utext; const uint64_t *qseg, *seg_off; const uint32_t *seg_len, *trl;
    double file_ms; uint64_t calls; int gz, members; int64_t period, count;     /* gz: host zlib level for plain text; members: the text already is deflate data (CGX_TEXT_GZIP_PIECES), trl its files' CRC-32 / ISIZE */
} devjob;
/* optional gzip output (option "gz_level" 1..9): grammar.<q>.s.gz, the same bytes through zlib's deflate */
static int write_one_file_gz(devjob *w, int32_t q) {
    char fn[4096], mode[8];
    snprintf(fn, sizeof fn, "%s/grammar.%d.s.gz", w->outdir, w->first + q); snprintf(mode, sizeof mode, "wb%d", w->gz);
    gzFile f = gzopen(fn, mode);
    if (!f) return CGX_ERR_IO;
    (void)gzbuffer(f, 1u << 18);
    int bad = 0;
    for (uint64_t s = w->qseg[q]; s < w->qseg[q + 1] && !bad; s++) if (gzwrite(f, w->utext + w->seg_off[s], w->seg_len[s]) != (int)w->seg_len[s]) bad = 1;
    if (gzclose(f) != Z_OK) bad = 1;
    return bad ? CGX_ERR_IO : CGX_OK;
}
/* Around the pieces of a text of deflate blocks (CGX_TEXT_GZIP_PIECES): the gzip header (RFC 1952 2.3: ID1 ID2 CM FLG MTIME XFL OS), and behind them
 * an empty final block (03 00), CRC-32 and ISIZE.  A file without rules is header + trailer: what gzclose writes for an empty file. */
static const unsigned char GZ_FILE_HEADER[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
static int write_one_file(devjob *w, int32_t q, struct iovec *iov) {
    if (w->gz && !w->members) return write_one_file_gz(w, q);
    char fn[4096];
    snprintf(fn, sizeof fn, w->members ? "%s/grammar.%d.s.gz" : "%s/grammar.%d.s", w->outdir, w->first + q);
    const uint64_t s0 = w->qseg[q], s1 = w->qseg[q + 1];
    /* overwrite in place and cut to length: same bytes as fopen(fn,"w"), but an existing file keeps its pages */
    int fd = open(fn, s0 == s1 && !w->members ? O_WRONLY | O_CREAT | O_TRUNC : O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return CGX_ERR_IO;
    int bad = 0; uint64_t pos = 0;
    unsigned char trailer[10] = {3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (w->members) { const uint32_t crc = w->trl[2 * (size_t)q], isz = w->trl[2 * (size_t)q + 1]; for (int k = 0; k < 4; k++) { trailer[2 + k] = (unsigned char)(crc >> (8 * k)); trailer[6 + k] = (unsigned char)(isz >> (8 * k)); } }
    int head = w->members, tail = w->members;                 /* still to be written */
    for (uint64_t s = s0; (s < s1 || head || tail) && !bad;) {
        int n = 0; uint64_t want = 0;
        if (head) { iov[n].iov_base = (void *)GZ_FILE_HEADER; iov[n].iov_len = sizeof GZ_FILE_HEADER; want += sizeof GZ_FILE_HEADER; n++; head = 0; }
        for (; s < s1 && n < IOV_MAX; s++, n++) { iov[n].iov_base = (void *)(w->utext + w->seg_off[s]); iov[n].iov_len = w->seg_len[s]; want += w->seg_len[s]; }
        if (tail && s == s1 && n < IOV_MAX) { iov[n].iov_base = trailer; iov[n].iov_len = sizeof trailer; want += sizeof trailer; n++; tail = 0; }
        uint64_t done = 0; int k = 0;
        while (done < want && !bad) {                       /* a short write resumes in the middle of a piece */
            ssize_t r = pwritev(fd, iov + k, n - k, (off_t)(pos + done)); w->calls++;
            if (r <= 0) { bad = 1; break; }
            done += (uint64_t)r;
            while (k < n && (uint64_t)r >= iov[k].iov_len) { r -= (ssize_t)iov[k].iov_len; k++; }
            if (k < n && r > 0) { iov[k].iov_base = (char *)iov[k].iov_base + r; iov[k].iov_len -= (size_t)r; }
        }
        pos += want;
    }
    if (!bad && pos && ftruncate(fd, (off_t)pos)) bad = 1;   /* an older, longer file of the same name is cut to the new length */
    close(fd);
    return bad ? CGX_ERR_IO : CGX_OK;
}
static void *dev_write_worker(void *arg) {
    devjob *w = arg; w->rc = CGX_OK; w->file_ms = 0; w->calls = 0;
    cgx__bind_thread(w->ctx);
    if (w->tid > 0) pin_to_device_node(w->ctx);               /* thread 0 is the batch's writer thread itself: already placed */
    struct iovec *iov = malloc(sizeof(struct iovec) * IOV_MAX);
    if (!iov) { w->rc = CGX_ERR_NOMEM; return NULL; }
    double t0 = now_ms();
    for (;;) {
        int32_t q = __atomic_fetch_add(w->next_q, 1, __ATOMIC_RELAXED);
        if (q >= w->nq) break;
        if (!file_selected(w->period, w->count, (int64_t)w->first + q)) continue;
        int rc = write_one_file(w, q, iov);
        if (rc != CGX_OK) { w->rc = rc; break; }
    }
    w->file_ms = now_ms() - t0;
    free(iov);
    return NULL;
}
/* phase 1, called by the thread that submits the batch: sizes the host buffers and enqueues every copy */
static int dev_copy_begin(pending *pw) {
    cgx_ctx *ctx = pw->ctx; hostbuf *hb = &pw->ws->hb[pw->hb]; int rc;
    if ((rc = cgx_text_info(ctx, pw->slot, &pw->ubytes, &pw->nseg, &pw->fbytes)) != CGX_OK) return rc;
    if (pinned_reserve((void **)&hb->utext, &hb->utext_cap, pw->ubytes + 64, 1) || pinned_reserve((void **)&hb->qseg, &hb->qseg_cap, (uint64_t)pw->nq + 2, 8) ||
        pinned_reserve((void **)&hb->segoff, &hb->segoff_cap, pw->nseg + 1, 8) || pinned_reserve((void **)&hb->seglen, &hb->seglen_cap, pw->nseg + 1, 4)) return CGX_ERR_NOMEM;
    const int r0 = pw->slot * READERS_PER_SLOT;
    const int64_t npieces = (int64_t)((pw->ubytes + COPY_PIECE - 1) / COPY_PIECE);
    for (int64_t pc = 0; pc < npieces; pc++) {
        const uint64_t lo = (uint64_t)pc * COPY_PIECE, n = pw->ubytes - lo < COPY_PIECE ? pw->ubytes - lo : COPY_PIECE;
        if ((rc = cgx_text_read_begin(ctx, pw->slot, lo, n, hb->utext + lo, r0 + (int)(pc % 3))) != CGX_OK) return rc;
    }
    if (pw->members) {                                       /* the files' CRC-32 / ISIZE travel with the piece lists (same reader: one wait covers both) */
        if (pinned_reserve((void **)&hb->trl, &hb->trl_cap, 2 * (uint64_t)pw->nq + 2, 4)) return CGX_ERR_NOMEM;
        if ((rc = cgx_text_trailers_begin(ctx, pw->slot, hb->trl, r0 + 3)) != CGX_OK) return rc;
    }
    return cgx_text_segments_begin(ctx, pw->slot, hb->qseg, hb->segoff, hb->seglen, r0 + 3);
}
/* phase 2 (after the copies have landed and the previous batch's files are complete) */
static int dev_write_files(pending *pw) {
    cgx_ctx *ctx = pw->ctx; const hostbuf *hb = &pw->ws->hb[pw->hb]; const int32_t nq = pw->nq;
    int nt = writer_threads(); if (nt > MAX_WRITERS) nt = MAX_WRITERS; if (nt > nq) nt = nq > 0 ? nq : 1;
    cgx__set_host_ms(ctx, "writer_threads", nt);             /* reported beside the timings (bench.py) */
    devjob jobs[MAX_WRITERS]; pthread_t th[MAX_WRITERS]; int started[MAX_WRITERS]; int32_t next = 0; int rc = CGX_OK;
    for (int t = 0; t < nt; t++) { memset(&jobs[t], 0, sizeof jobs[t]); jobs[t].ctx = ctx; jobs[t].tid = t; jobs[t].nq = nq; jobs[t].first = pw->first; jobs[t].outdir = pw->outdir; jobs[t].next_q = &next;
                                   jobs[t].utext = hb->utext; jobs[t].qseg = hb->qseg; jobs[t].seg_off = hb->segoff; jobs[t].seg_len = hb->seglen; jobs[t].trl = hb->trl; jobs[t].rc = CGX_OK; jobs[t].gz = pw->gz; jobs[t].members = pw->members; jobs[t].period = pw->period; jobs[t].count = pw->count; }
    started[0] = 1;
    for (int t = 1; t < nt; t++) started[t] = !pthread_create(&th[t], NULL, dev_write_worker, &jobs[t]);   /* a thread that cannot start: the others take its share */
    dev_write_worker(&jobs[0]);
    for (int t = 1; t < nt; t++) if (started[t]) pthread_join(th[t], NULL);
    double wr = 0; uint64_t calls = 0; int ran = 0;
    for (int t = 0; t < nt; t++) if (started[t]) { if (jobs[t].rc != CGX_OK && rc == CGX_OK) rc = jobs[t].rc; wr += jobs[t].file_ms; calls += jobs[t].calls; ran++; }
    pw->file_ms = wr / (ran ? ran : 1);
    if (getenv("CGX_TRACE"))
        fprintf(stderr, "cgx writer: unique text %.2f GB + %llu pieces on the host after %.1f ms; %d threads assembled %.2f GB of files in %.1f ms per thread (%.1f GB/s), %llu pwritev calls\n",
                pw->ubytes / 1e9, (unsigned long long)pw->nseg, pw->wait_ms, ran, pw->fbytes / 1e9, pw->file_ms, pw->fbytes / 1e6 / (pw->file_ms > 0 ? pw->file_ms : 1), (unsigned long long)calls);
    return rc;
}

/* The file phase by itself, for a caller that holds the unique text and the piece lists (cgx_text_info / cgx_text_read /
 * cgx_text_segments hand them out): assembles grammar.<first+q>.s for q < nq in `outdir` with `nthreads` threads, exactly as the
 * writer of cgx_extract_grammars* does (the same write_one_file).  Needs no GPU and no context: tools/rehearse_writers.py
 * replays one recorded batch from several processes at once to measure what one host can write. */
int cgx_assemble_files(const char *utext, const uint64_t *qseg, const uint64_t *seg_off, const uint32_t *seg_len, int32_t nq, int32_t first,
                       const char *outdir, int nthreads, double *file_ms) {
    return cgx_assemble_files_enc(utext, qseg, seg_off, seg_len, nq, first, outdir, nthreads, file_ms, CGX_TEXT_PLAIN, NULL);
}
int cgx_assemble_files_enc(const char *utext, const uint64_t *qseg, const uint64_t *seg_off, const uint32_t *seg_len, int32_t nq, int32_t first,
                           const char *outdir, int nthreads, double *file_ms, int encoding, const uint32_t *trl) {
    if (!utext || !qseg || !seg_off || !seg_len || nq < 0 || !outdir || (encoding != CGX_TEXT_PLAIN && encoding != CGX_TEXT_GZIP_PIECES) || (encoding == CGX_TEXT_GZIP_PIECES && !trl)) return CGX_ERR_ARG;
    int nt = nthreads < 1 ? 1 : nthreads > MAX_WRITERS ? MAX_WRITERS : nthreads; if (nt > nq) nt = nq > 0 ? nq : 1;
    devjob jobs[MAX_WRITERS]; pthread_t th[MAX_WRITERS]; int started[MAX_WRITERS]; int32_t next = 0; int rc = CGX_OK;
    for (int t = 0; t < nt; t++) { memset(&jobs[t], 0, sizeof jobs[t]); jobs[t].tid = t; jobs[t].nq = nq; jobs[t].first = first; jobs[t].outdir = outdir; jobs[t].next_q = &next;
                                   jobs[t].utext = utext; jobs[t].qseg = qseg; jobs[t].seg_off = seg_off; jobs[t].seg_len = seg_len; jobs[t].trl = trl; jobs[t].rc = CGX_OK; jobs[t].members = encoding == CGX_TEXT_GZIP_PIECES; }
    started[0] = 1;
    for (int t = 1; t < nt; t++) started[t] = !pthread_create(&th[t], NULL, dev_write_worker, &jobs[t]);
    dev_write_worker(&jobs[0]);
    for (int t = 1; t < nt; t++) if (started[t]) pthread_join(th[t], NULL);
    double wr = 0; int ran = 0;
    for (int t = 0; t < nt; t++) if (started[t]) { if (jobs[t].rc != CGX_OK && rc == CGX_OK) rc = jobs[t].rc; wr += jobs[t].file_ms; ran++; }
    if (file_ms) *file_ms = wr / (ran ? ran : 1);
    return rc;
}

/* number of grammar lines the writer will produce (PrintResults.c:451-570 walked without formatting) */
static uint64_t range_len(const range *r, uint32_t id) { return (r[id].down == -1 || r[id].up == -1) ? 0 : (uint64_t)(r[id].up - r[id].down + 1); }
static uint64_t count_lines(const batch *b) {
    const uint32_t G = b->g, D1 = b->d1, D2 = b->d2; uint64_t n = 0;
    for (int32_t q = 0; q < b->nq; q++) {
        for (uint32_t k = 0; k < b->qblocks[q].n; k++) { uint32_t p = b->qblocks[q].v[k]; n += range_len(b->rng1, p + G) + range_len(b->rng1, p) + range_len(b->rng2, p) + range_len(b->rng0, p); }
        for (uint32_t k = 0; b->qone && k < b->qone[q].n; k++) { uint32_t id = b->qone[q].v[k]; n += range_len(b->rng1, 2 * G + id) + range_len(b->rng2, G + D2 + id) + range_len(b->rng2, G + D2 + D1 + id); }
        for (uint32_t k = 0; b->qtwo && k < b->qtwo[q].n; k++) n += range_len(b->rng2, G + b->qtwo[q].v[k]);
    }
    return n;
}
static void *pending_main(void *arg) {
    pending *pw = arg; wstate *ws = pw->ws; double t = now_ms(); int rc = CGX_OK;
    cgx__bind_thread(pw->ctx); pin_to_device_node(pw->ctx);  /* this background thread doubles as writer 0 */
    if (pw->dev) {                                            /* phase 1 was enqueued by the submitter: wait for the DMA */
        for (int r = 0; r < READERS_PER_SLOT; r++) { int rw = cgx_text_read_wait(pw->ctx, pw->slot * READERS_PER_SLOT + r); if (rc == CGX_OK) rc = rw; }
        pw->wait_ms = now_ms() - t;
    }
    pthread_mutex_lock(&ws->m);                               /* the files of the previous batch first */
    while (ws->done_seq + 1 < pw->seq) pthread_cond_wait(&ws->cv, &ws->m);
    pthread_mutex_unlock(&ws->m);
    if (rc == CGX_OK) rc = pw->dev ? dev_write_files(pw) : write_grammars(pw->b, pw->outdir, pw->first, &pw->lines);
    pw->rc = rc; pw->ms = now_ms() - t;
    pthread_mutex_lock(&ws->m); ws->done_seq = pw->seq; pthread_cond_broadcast(&ws->cv); pthread_mutex_unlock(&ws->m);
    return NULL;
}
/* waits for the oldest batch in flight; its error, if any, is returned */
static int join_oldest(cgx_ctx *ctx, wstate *ws) {
    if (!ws || !ws->n) return CGX_OK;
    pending *pw = ws->inflight[0];
    pthread_join(pw->th, NULL);
    int rc = pw->rc;
    cgx__set_host_ms(ctx, "write", pw->ms);
    if (pw->dev) { cgx__set_host_ms(ctx, "write_wait_d2h", pw->wait_ms); cgx__set_host_ms(ctx, "write_file", pw->file_ms); }
    /* running totals over every batch joined so far: a caller that brackets a region with cgx_flush reads the region's own batches from
     * the differences (the figures above are those of the batch joined LAST, which with two batches in flight is not the one just submitted) */
    { double a = cgx_host_ms(ctx, "write_sum"), b = cgx_host_ms(ctx, "write_wait_d2h_sum"), c = cgx_host_ms(ctx, "write_file_sum"), d = cgx_host_ms(ctx, "batches_joined");
      cgx__set_host_ms(ctx, "write_sum", (a > 0 ? a : 0) + pw->ms); cgx__set_host_ms(ctx, "batches_joined", (d > 0 ? d : 0) + 1);
      if (pw->dev) { cgx__set_host_ms(ctx, "write_wait_d2h_sum", (b > 0 ? b : 0) + pw->wait_ms); cgx__set_host_ms(ctx, "write_file_sum", (c > 0 ? c : 0) + pw->file_ms); } }
    ws->inflight[0] = ws->inflight[1]; ws->inflight[1] = NULL; ws->n--;
    if (pw->b) { batch_free(pw->b); free(pw->b); }
    free(pw->outdir); free(pw);
    return rc;
}
/* hands a finished batch to the writer: device text in `slot` (dev) or the host formatter on batch b */
static int submit_output(cgx_ctx *ctx, batch *b, int dev, int slot, int32_t nq, const char *outdir, int32_t first) {
    wstate *ws = get_wstate(ctx); int rc = CGX_OK;
    if (!ws) return CGX_ERR_NOMEM;
    while (ws->n >= 2) { int r = join_oldest(ctx, ws); if (rc == CGX_OK) rc = r; }      /* at most two batches in flight: one copying, one writing */
    if (rc != CGX_OK) return rc;
    pending *pw = calloc(1, sizeof *pw);
    if (!pw) return CGX_ERR_NOMEM;
    pw->ws = ws; pw->ctx = ctx; pw->b = b; pw->dev = dev; pw->slot = slot; pw->nq = nq; pw->first = first; pw->outdir = strdup(outdir);
    pw->hb = ws->n && ws->inflight[0]->hb == 0 ? 1 : 0;      /* the buffer set the batch still in flight does not use */
    pw->gz = b ? b->gz_level : (int)cgx__option(ctx, "gz_level"); pw->period = b ? b->write_period : cgx__option(ctx, "write_period"); pw->count = b ? b->write_count : cgx__option(ctx, "write_count");
    pw->members = dev && cgx_text_encoding(ctx, slot) == CGX_TEXT_GZIP_PIECES;
    if (!pw->outdir) { free(pw); return CGX_ERR_NOMEM; }
    if (dev && (rc = dev_copy_begin(pw)) != CGX_OK) {         /* copies already enqueued target buffers that stay allocated: drain them */
        for (int r = 0; r < READERS_PER_SLOT; r++) (void)cgx_text_read_wait(ctx, slot * READERS_PER_SLOT + r);
        free(pw->outdir); free(pw); return rc;
    }
    pw->seq = ++ws->next_seq;
    if (pthread_create(&pw->th, NULL, pending_main, pw)) {
        for (int r = 0; dev && r < READERS_PER_SLOT; r++) (void)cgx_text_read_wait(ctx, slot * READERS_PER_SLOT + r);
        pthread_mutex_lock(&ws->m); ws->next_seq--; pthread_mutex_unlock(&ws->m);
        free(pw->outdir); free(pw); return CGX_ERR_NOMEM;
    }
    ws->inflight[ws->n++] = pw;
    return CGX_OK;
}

static int run_batch(cgx_ctx *ctx, const cgx_corpus *c, batch *b, const char *outdir, int32_t first, uint64_t *nrules, int *handed_off) {
    int rc; double t0 = now_ms(), t, tl = t0;
#define LAP(name) do { double n_ = now_ms(); cgx__set_host_ms(ctx, "t_" name, n_ - tl); tl = n_; } while (0)
    b->c = c;
    b->write_period = cgx__option(ctx, "write_period"); b->write_count = cgx__option(ctx, "write_count");
    if (outdir && b->write_period > 0) {                      /* a batch none of whose queries gets a file only counts its rules: no text layout, no DMA */
        int any = 0;
        for (int64_t g = first; g < (int64_t)first + b->nq && !any; ) { if (file_selected(b->write_period, b->write_count, g)) any = 1; else g += b->write_period - g % b->write_period; }
        if (!any) outdir = NULL;
    }
    if ((rc = cgx_upload_queries(ctx, b->qoff, b->nq, b->qtok, b->ntok)) != CGX_OK) return rc;
    if ((rc = cgx_sa_lookup(ctx)) != CGX_OK) return rc;
    LAP("upload_sa");
    /* GenerateBlocks on the device: distinct contiguous phrases + per-query lists (kept there for the formatter) */
    if ((rc = cgx_make_blocks(ctx)) != CGX_OK) return rc;
    {
        uint32_t counts0[16];
        if (cgx_fetch(ctx, "counts", counts0, sizeof counts0) < 0) return CGX_ERR_HIP;
        b->g = counts0[6];
    }
    LAP("blocks");
    const int devfmt = outdir && cgx__option(ctx, "device_format");
    b->gz_level = (int)cgx__option(ctx, "gz_level");
    if (devfmt && (rc = ensure_vocab(ctx, c)) != CGX_OK) return rc;
    LAP("qblocks");
    if ((rc = cgx_gappy_search(ctx)) != CGX_OK) return rc;
    LAP("gappy");
    if ((rc = cgx_extract(ctx)) != CGX_OK) return rc;
    LAP("extract");
    /* lexicon + MaxLex features on the device (host path only if a target-side hash collides) */
    int exact_host = 0;
    rc = cgx_lexicon(ctx);
    if (rc == CGX_ERR_STATE && strstr(cgx_last_error(ctx), "hash collision")) exact_host = 1; else if (rc != CGX_OK) return rc;
    cgx__set_host_ms(ctx, "exact_host_lexicon", exact_host);  /* 1: this batch took the exact host lexicon */
    LAP("lexicon");
    if (!outdir && !exact_host && cgx__option(ctx, "device_format")) {     /* nothing to write: only the number of rules is wanted */
        uint64_t nl = 0;
        if ((rc = cgx_format(ctx, NULL, &nl, NULL)) != CGX_OK) return rc;
        LAP("format");
        if (nrules) *nrules = nl;
        cgx__set_host_ms(ctx, "lists", 0); cgx__set_host_ms(ctx, "lexicon", 0); cgx__set_host_ms(ctx, "write", 0);
        cgx__set_host_ms(ctx, "total", now_ms() - t0);
        return CGX_OK;
    }
    if (devfmt && !exact_host) {
        /* the text of every file is laid out on the GPU; host threads only pull byte ranges and write them */
        uint64_t bytes = 0, nl = 0; int slot = 0;
        rc = cgx_format(ctx, &bytes, &nl, &slot);
        LAP("format");
        if (rc == CGX_OK) {
            if (nrules) *nrules = nl;
            fprintf(stderr, "Start Printing Gappy Phrases...\n");
            t = now_ms();
            if ((rc = submit_output(ctx, NULL, 1, slot, b->nq, outdir, first)) != CGX_OK) return rc;
            LAP("flush_wait");
            if (!cgx__option(ctx, "async_write") && (rc = cgx_flush(ctx)) != CGX_OK) return rc;
            cgx__set_host_ms(ctx, "lists", 0); cgx__set_host_ms(ctx, "lexicon", 0);
            cgx__set_host_ms(ctx, "total", now_ms() - t0);
            return CGX_OK;
        }
        if (rc != CGX_ERR_STATE) return rc;                   /* CGX_ERR_STATE: a line the device cannot represent -> host formatter below */
    }
    /* device results needed by the host stages */
    if ((rc = fetch_host_blocks(ctx, b)) != CGX_OK) return rc;
    uint32_t *pid1 = NULL, *pid2 = NULL; cgx_gappy *g1 = NULL; cgx_twogappy *g2 = NULL; uint32_t counts[16];
    if (cgx_fetch(ctx, "counts", counts, sizeof counts) < 0) return CGX_ERR_HIP;
    b->d1 = counts[1]; b->d2 = counts[4]; b->sep1 = counts[10]; b->sep2a = counts[11]; b->sep2b = counts[12];
    t = now_ms();
    {   /* per-query pattern lists; the scratch arrays of this block are released on every path out of it */
        int32_t *tok2q = NULL, *pos1 = NULL, *pos2 = NULL;
        if ((rc = fetch_alloc(ctx, "pid1", (void **)&pid1, 4, &b->e1)) || (rc = fetch_alloc(ctx, "g1", (void **)&g1, sizeof *g1, NULL)) ||
            (rc = fetch_alloc(ctx, "pid2", (void **)&pid2, 4, &b->e2)) || (rc = fetch_alloc(ctx, "g2", (void **)&g2, sizeof *g2, NULL)) ||
            (rc = fetch_alloc(ctx, "p1d", (void **)&b->p1d, sizeof *b->p1d, NULL)) || (rc = fetch_alloc(ctx, "c2d", (void **)&b->c2d, 4, NULL)) ||
            (rc = fetch_alloc(ctx, "one2", (void **)&b->one2, 4, NULL))) goto lists_done;
        tok2q = malloc(((size_t)b->ntok + 1) * 4); pos1 = malloc(((size_t)b->e1 + 1) * 4); pos2 = malloc(((size_t)b->e2 + 1) * 4);
        if (!tok2q || !pos1 || !pos2) { rc = CGX_ERR_NOMEM; goto lists_done; }
        for (int32_t q = 0; q < b->nq; q++) for (int32_t k = b->qoff[q]; k < b->qoff[q + 1]; k++) tok2q[k] = q;
        for (uint32_t i = 0; i < b->e1; i++) pos1[i] = g1[i].qrystart;
        for (uint32_t i = 0; i < b->e2; i++) pos2[i] = (int32_t)g2[i].gap2;
        if ((rc = pattern_lists(b, pid1, pos1, b->e1, tok2q, &b->qone)) == CGX_OK) rc = pattern_lists(b, pid2, pos2, b->e2, tok2q, &b->qtwo);
    lists_done:
        free(tok2q); free(pos1); free(pos2); free(pid1); free(pid2); free(g1); free(g2);
        if (rc != CGX_OK) return rc;
    }
    cgx__set_host_ms(ctx, "lists", now_ms() - t);
    t = now_ms();
    score_tables();
    uint32_t nl0 = 0, nl1 = 0, nl2 = 0;
    if (!exact_host) {
        int64_t by1 = 0, by2 = 0, by0 = 0;
        /* the lexicon lines go to one of TWO page-locked arenas that alternate between batches, and the host formatter of a
         * batch reads its arena until its files are written: the batch before the previous one must be finished before its
         * arena is filled again (up to two batches can be in flight) */
        { wstate *ws = cgx__get_host_state(ctx); while (ws && ws->n > 1) { int r = join_oldest(ctx, ws); if (r != CGX_OK) return r; } }
        cgx_pinned_next_batch(ctx);
        if (cgx_fetch_pinned(ctx, "lex2", (void **)&b->L2, &by2) == CGX_OK && cgx_fetch_pinned(ctx, "lex1", (void **)&b->L1, &by1) == CGX_OK &&
            cgx_fetch_pinned(ctx, "lex0", (void **)&b->L0, &by0) == CGX_OK) {
            b->L_pinned = 1; nl1 = (uint32_t)((size_t)by1 / sizeof *b->L1); nl2 = (uint32_t)((size_t)by2 / sizeof *b->L2); nl0 = (uint32_t)((size_t)by0 / sizeof *b->L0);
        } else {                                              /* no pinned memory: pageable copies */
            b->L_pinned = 0; b->L0 = b->L1 = b->L2 = NULL;
            if ((rc = fetch_alloc(ctx, "lex1", (void **)&b->L1, sizeof *b->L1, &nl1)) || (rc = fetch_alloc(ctx, "lex2", (void **)&b->L2, sizeof *b->L2, &nl2)) ||
                (rc = fetch_alloc(ctx, "lex0", (void **)&b->L0, sizeof *b->L0, &nl0))) return rc;
        }
    } else {
        /* exact host lexicon (ExtractPair.c:515-1276 restated on integer tuples) + MaxLex through the task ABI */
        if ((rc = fetch_alloc(ctx, "p1", (void **)&b->p1, sizeof *b->p1, NULL)) || (rc = fetch_alloc(ctx, "s1", (void **)&b->s1, sizeof *b->s1, NULL)) ||
            (rc = fetch_alloc(ctx, "hits1", (void **)&b->hits1, sizeof *b->hits1, &b->h1)) || (rc = fetch_alloc(ctx, "c2", (void **)&b->c2, 4, NULL)) ||
            (rc = fetch_alloc(ctx, "s2", (void **)&b->s2, sizeof *b->s2, NULL)) ||
            (rc = fetch_alloc(ctx, "r0", (void **)&b->r0, sizeof *b->r0, &b->n0)) || (rc = fetch_alloc(ctx, "r1", (void **)&b->r1, sizeof *b->r1, &b->n1)) ||
            (rc = fetch_alloc(ctx, "r2", (void **)&b->r2, sizeof *b->r2, &b->n2)) ||
            (rc = fetch_alloc(ctx, "pidx", (void **)&b->pidx, 4, NULL)) || (rc = fetch_alloc(ctx, "miss", (void **)&b->miss, 4, NULL))) return rc;
        if ((rc = build_lexicons(b)) != CGX_OK) return rc;
        float *fe = malloc(((size_t)b->ntask + 1) * 4), *ef = malloc(((size_t)b->ntask + 1) * 4);
        if (!fe || !ef) { free(fe); free(ef); return CGX_ERR_NOMEM; }
        if ((rc = cgx_lex_features(ctx, b->tasks, b->ntask, b->nl1, b->nl2, fe, ef)) != CGX_OK) { free(fe); free(ef); return rc; }
        nl0 = b->nl0; nl1 = b->nl1; nl2 = b->nl2;
        b->L0 = calloc((size_t)nl0 + 1, sizeof *b->L0); b->L1 = calloc((size_t)nl1 + 1, sizeof *b->L1); b->L2 = calloc((size_t)nl2 + 1, sizeof *b->L2);
        if (!b->L0 || !b->L1 || !b->L2) { free(fe); free(ef); return CGX_ERR_NOMEM; }   /* the L* that were allocated belong to the batch and go with it */
        for (uint32_t i = 0; i < b->ntask; i++) {
            int kind = i < nl1 ? 1 : i < nl1 + nl2 ? 2 : 0; uint32_t k = b->tasks[i].lexid;
            const lexent *e = kind == 1 ? &b->lex1[k] : kind == 2 ? &b->lex2[k] : &b->lex0[k];
            cgx_lexent *o = kind == 1 ? &b->L1[k] : kind == 2 ? &b->L2[k] : &b->L0[k];
            o->id = e->id; o->kind = (uint8_t)kind; o->f = (uint16_t)e->f; o->fsample = (uint16_t)e->fsample; o->paircount = (uint16_t)e->paircount; o->fe = fe[i]; o->ef = ef[i];
            o->tstart = b->tasks[i].tstart; o->end = b->tasks[i].end; o->gap1 = b->tasks[i].gap1; o->gap1_1 = b->tasks[i].gap1_1; o->gap2 = b->tasks[i].gap2; o->gap2_1 = b->tasks[i].gap2_1;
        }
        free(fe); free(ef);
    }
    b->nl0 = nl0; b->nl1 = nl1; b->nl2 = nl2;
    free(b->rng0); free(b->rng1); free(b->rng2); b->rng0 = b->rng1 = b->rng2 = NULL;
    if (!exact_host) {                                        /* ranges were built on the device next to the lexicon */
        if ((rc = fetch_alloc(ctx, "rng1", (void **)&b->rng1, sizeof *b->rng1, NULL)) || (rc = fetch_alloc(ctx, "rng2", (void **)&b->rng2, sizeof *b->rng2, NULL)) ||
            (rc = fetch_alloc(ctx, "rng0", (void **)&b->rng0, sizeof *b->rng0, NULL))) return rc;
    } else {
        b->rng1 = make_ranges_dev(b->L1, nl1, 2 * b->g + b->d1); b->rng2 = make_ranges_dev(b->L2, nl2, b->g + 2 * b->d1 + b->d2); b->rng0 = make_ranges_dev(b->L0, nl0, b->g);
        if (!b->rng0 || !b->rng1 || !b->rng2) return CGX_ERR_NOMEM;
    }
    cgx__set_host_ms(ctx, "lexicon", now_ms() - t);
    uint64_t lines = count_lines(b);
    if (nrules) *nrules = lines;
    if (outdir && cgx__option(ctx, "async_write")) {
        /* hand the finished batch to a background writer; the GPU is free for the next batch */
        fprintf(stderr, "Start Printing Gappy Phrases...\n");
        if ((rc = submit_output(ctx, b, 0, 0, b->nq, outdir, first)) != CGX_OK) return rc;
        *handed_off = 1;
        cgx__set_host_ms(ctx, "total", now_ms() - t0);
        return CGX_OK;
    }
    t = now_ms();
    if (outdir) {
        uint64_t wrote = 0;
        fprintf(stderr, "Start Printing Gappy Phrases...\n");
        if ((rc = write_grammars(b, outdir, first, &wrote)) != CGX_OK) return rc;
    }
    cgx__set_host_ms(ctx, "write", now_ms() - t);
    cgx__set_host_ms(ctx, "total", now_ms() - t0);
    return CGX_OK;
}

/* waits for every batch handed to the writer (async_write); the first error, if any, is returned here */
int cgx_flush(cgx_ctx *ctx) {
    if (!ctx) return CGX_ERR_ARG;
    wstate *ws = cgx__get_host_state(ctx); int rc = CGX_OK;
    while (ws && ws->n) { int r = join_oldest(ctx, ws); if (rc == CGX_OK) rc = r; }
    return rc;
}
void cgx__host_release(cgx_ctx *ctx) {
    (void)cgx_flush(ctx);
    wstate *ws = cgx__get_host_state(ctx);
    if (!ws) return;
    cgx__bind_thread(ctx);
    for (int k = 0; k < 2; k++) { cgx_pinned_free(ws->hb[k].utext); cgx_pinned_free(ws->hb[k].segoff); cgx_pinned_free(ws->hb[k].seglen); cgx_pinned_free(ws->hb[k].qseg); cgx_pinned_free(ws->hb[k].trl); }
    pthread_mutex_destroy(&ws->m); pthread_cond_destroy(&ws->cv);
    free(ws); cgx__set_host_state(ctx, NULL);
}

static int extract_ids_once(cgx_ctx *ctx, const cgx_corpus *c, const int32_t *qoff, int32_t nq, const int32_t *qtok, int32_t ntok,
                            const char *outdir, int32_t first, uint64_t *nrules) {
    batch *b = calloc(1, sizeof *b);
    if (!b) return CGX_ERR_NOMEM;
    b->nq = nq; b->ntok = ntok;
    b->qoff = malloc(((size_t)nq + 1) * 4); b->qtok = malloc(((size_t)ntok + 1) * 4);
    if (!b->qoff || !b->qtok) { batch_free(b); free(b); return CGX_ERR_NOMEM; }
    const int32_t base = nq ? qoff[0] : 0;                   /* sub-batches start in the middle of the caller's arrays */
    for (int32_t q = 0; q < nq; q++) b->qoff[q] = qoff[q] - base;
    b->qoff[nq] = ntok; if (ntok) memcpy(b->qtok, qtok + base, (size_t)ntok * 4);
    int handed_off = 0;
    int rc = run_batch(ctx, c, b, outdir, first, nrules, &handed_off);
    if (!handed_off) { batch_free(b); free(b); }
    return rc;
}
int cgx_extract_grammars_ids(cgx_ctx *ctx, const cgx_corpus *c, const int32_t *qoff, int32_t nq, const int32_t *qtok, int32_t ntok,
                             const char *outdir, int32_t first, uint64_t *nrules) {
    if (!ctx || !c || nq < 0 || ntok < 0 || (nq && !qoff) || (ntok && !qtok) || (ntok && !nq)) return CGX_ERR_ARG;
    /* Internal batches: `sub_batch` queries each if set; otherwise as many queries as fit AUTO_BATCH_TOKENS query
     * tokens (option "auto_batch_tokens", default 300 000 = about 12 k sentences), which bounds the device memory of a call whatever the size of the query file
     * (the text of 10 k queries alone is 30-40 GB, kept twice).  Queries are independent: any split gives the same files. */
    const int64_t sub = cgx__option(ctx, "sub_batch"), AUTO_BATCH_TOKENS = cgx__option(ctx, "auto_batch_tokens");
    if ((sub <= 0 || sub >= nq) && ntok <= AUTO_BATCH_TOKENS) return extract_ids_once(ctx, c, qoff, nq, qtok, ntok, outdir, first, nrules);
    uint64_t total = 0;
    /* the offsets may start anywhere in the caller's token array (a shard, cgx_extract_grammars_shard): `ntok` counts
     * the tokens of these nq queries, so the last query ends at qoff[0] + ntok */
    const int32_t end = qoff[0] + ntok;
    (void)cgx_set_option(ctx, "prealloc_text", 1);           /* several internal batches: both text slots are going to be needed */
    for (int32_t q0 = 0; q0 < nq;) {
        int32_t q1 = q0;
        if (sub > 0) q1 = q0 + (int32_t)sub < nq ? q0 + (int32_t)sub : nq;
        else { while (q1 < nq && ((q1 + 1 < nq ? qoff[q1 + 1] : end) - qoff[q0] <= AUTO_BATCH_TOKENS || q1 == q0)) q1++; }
        uint64_t n = 0;
        int32_t t1 = q1 < nq ? qoff[q1] : end;
        int rc = extract_ids_once(ctx, c, qoff + q0, q1 - q0, qtok, t1 - qoff[q0], outdir, first + q0, &n);
        if (rc != CGX_OK) return rc;
        total += n; q0 = q1;
    }
    if (nrules) *nrules = total;
    return CGX_OK;
}

/* constructQryIndex (Start.cu:74-111): query lines [q_begin, q_end) (q_end < 0: to the end) as token ids, OOV = -1 */
static int load_queries(const cgx_corpus *c, const char *qryfile, int32_t q_begin, int32_t q_end, int32_t **off_out, int32_t *nq_out, int32_t **tok_out, int32_t *ntok_out) {
    size_t len; char *buf = slurp(qryfile, &len);
    if (!buf) return CGX_ERR_IO;
    int32_t *off = NULL, *tok = NULL; size_t no = 0, co = 0, nt = 0, ct = 0; size_t i = 0; int32_t line = 0; int rc = CGX_OK;
    while (i < len && rc == CGX_OK) {
        size_t e = i; while (e < len && buf[e] != '\n') e++;
        int take = line >= q_begin && (q_end < 0 || line < q_end);
        if (take) { if (no == co) { co = co ? co * 2 : 256; int32_t *n = realloc(off, co * 4); if (!n) { rc = CGX_ERR_NOMEM; break; } off = n; } off[no++] = (int32_t)nt; }
        size_t p = i;
        for (; take;) {
            while (p < e && buf[p] == ' ') p++;
            if (p >= e) break;
            size_t q = p; while (q < e && buf[q] != ' ') q++;
            if (isspace((unsigned char)buf[p])) break;
            int32_t id = wordmap_get(&c->smap, buf + p, q - p);
            if (nt == ct) { ct = ct ? ct * 2 : 1024; int32_t *n = realloc(tok, ct * 4); if (!n) { rc = CGX_ERR_NOMEM; break; } tok = n; }
            tok[nt++] = id < 0 ? -1 : id;
            p = q;
        }
        line++; i = e + 1;
    }
    free(buf);
    if (rc != CGX_OK) { free(off); free(tok); return rc; }
    *off_out = off; *nq_out = (int32_t)no; *tok_out = tok; *ntok_out = (int32_t)nt;
    return CGX_OK;
}
int cgx_extract_grammars(cgx_ctx *ctx, const cgx_corpus *c, const char *qryfile, const char *outdir, int32_t q_begin, int32_t q_end, uint64_t *nrules) {
    if (!ctx || !c || !qryfile) return CGX_ERR_ARG;
    int32_t *off = NULL, *tok = NULL, nq = 0, ntok = 0;
    int rc = load_queries(c, qryfile, q_begin, q_end, &off, &nq, &tok, &ntok);
    if (rc != CGX_OK) return rc;
    rc = cgx_extract_grammars_ids(ctx, c, off, nq, tok, ntok, outdir, q_begin > 0 ? q_begin : 0, nrules);
    free(off); free(tok);
    return rc;
}
/* Query sharding policy (one process per GPU): contiguous query ranges with about equal TOKEN counts -- shard r ends
 * with the first query whose last token reaches r+1 shares of the tokens.  The same arithmetic as cgx_amd/shard.py
 * (bench.py); bounds[0] = 0, bounds[world] = nq. */
int cgx_shard_bounds(const int32_t *qoff, int32_t nq, int64_t ntok, int32_t world, int32_t *bounds) {
    if (nq < 0 || ntok < 0 || world < 1 || !bounds || (nq && !qoff)) return CGX_ERR_ARG;
    bounds[0] = 0; bounds[world] = nq;
    for (int32_t r = 1; r < world; r++) {
        const int64_t target = (int64_t)r * ntok / world;
        int32_t a = 0, z = nq;                                /* first query whose end offset is >= target */
        while (a < z) { int32_t m = a + (z - a) / 2; int64_t end = m + 1 < nq ? qoff[m + 1] : ntok; if (end < target) a = m + 1; else z = m; }
        bounds[r] = a > bounds[r - 1] ? a : bounds[r - 1];
    }
    return CGX_OK;
}
int cgx_extract_grammars_shard(cgx_ctx *ctx, const cgx_corpus *c, const char *qryfile, const char *outdir, int32_t shard, int32_t nshard, uint64_t *nrules) {
    if (!ctx || !c || !qryfile || nshard < 1 || shard < 0 || shard >= nshard) return CGX_ERR_ARG;
    int32_t *off = NULL, *tok = NULL, nq = 0, ntok = 0;
    int rc = load_queries(c, qryfile, 0, -1, &off, &nq, &tok, &ntok);
    if (rc != CGX_OK) return rc;
    int32_t *b = malloc(((size_t)nshard + 1) * sizeof *b);
    if (!b) { free(off); free(tok); return CGX_ERR_NOMEM; }
    rc = cgx_shard_bounds(off, nq, ntok, nshard, b);
    if (rc == CGX_OK) {
        const int32_t q0 = b[shard], q1 = b[shard + 1];
        const int32_t t0 = q0 < nq ? off[q0] : ntok, t1 = q1 < nq ? off[q1] : ntok;
        rc = cgx_extract_grammars_ids(ctx, c, off + q0, q1 - q0, tok, t1 - t0, outdir, q0, nrules);   /* offsets stay absolute: the callee rebases on off[q0] */
    }
    free(b); free(off); free(tok);
    return rc;
}
