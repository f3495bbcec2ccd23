/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL reference host stages, compiled by oracle/Makefile straight from
 * /root/reference/{SuffixArray,ExtractPair,PrintResults,Timer}.c into oracle/_ref/, on the
 * intermediates dumped by the CPU oracle (orc_dump).  It pins the oracle's restatement of
 *   - suffixArrayConstruct            (SuffixArray.c:196-242)          mode "sa"
 *   - createLexiconGappyFast / TwoGapFast / Fast (ExtractPair.c:515-1276)
 *     + print_query_GPU_Gappy         (PrintResults.c:407-577)         mode "grammar"
 * against the reference's own object code.  Nothing here is copied from the reference: the
 * file only includes its headers where they lie and calls its functions.  The small pieces
 * of host glue that live inside ExtractPairs_Large_Data_Gappy (a .cu file that cannot be
 * built here) -- the id -> lexicon range loops (ExtractPair.cu:3743-3756, 3802-3816), the
 * contiguous range lookup (:2082-2106) and the MaxLex scatter (:3965-3982) -- are restated.
 *
 * usage: ref_harness sa <dump>                 -> exit 0 when the reference SA == dumped SA
 *        ref_harness grammar <dump> <outdir>   -> writes grammar.<q>.s via the reference
 *        ref_harness time-sa <dump>            -> prints seconds of suffixArrayConstruct
 *        ref_harness time-grammar <dump> <outdir> -> as "grammar", prints seconds of createLexicon*Fast + print_query_GPU_Gappy
 */
#include "ComTypes.h"
#include "SuffixArray.h"
#include "ExtractPair.h"
#include "PrintResults.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

static_assert(sizeof(gappy) == 7 && sizeof(gapPattern) == 21 && sizeof(twogappy) == 9 && sizeof(twoGapPattern) == 9, "wire layout");
static_assert(sizeof(gappy_search) == 19 && sizeof(two_gappy_search) == 21 && sizeof(oneGapOnSA) == 9 && sizeof(twoGapOnSA) == 10, "wire layout");
static_assert(sizeof(rule_onegap) == 11 && sizeof(rule_twogap) == 13 && sizeof(res_phrase_t) == 9 && sizeof(lexicalTask) == 34, "wire layout");
static_assert(sizeof(saind_t) == 16 && sizeof(precomp_st_end) == 8 && sizeof(result_t) == 8 && sizeof(precompute_enu_3) == 5, "wire layout");

typedef std::map<std::string, std::vector<char> > dump_t;

static dump_t read_dump(const char *path) {
    dump_t d;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    char tag[9]; unsigned long long nb;
    while (fread(tag, 1, 8, f) == 8 && fread(&nb, 8, 1, f) == 1) {
        tag[8] = 0;
        std::vector<char> v(nb);
        if (nb && fread(v.data(), 1, nb, f) != nb) { fprintf(stderr, "short dump\n"); exit(2); }
        d[tag] = v;
    }
    fclose(f);
    return d;
}
template <class T> static T *arr(dump_t &d, const char *k) { if (!d.count(k)) { fprintf(stderr, "dump lacks %s\n", k); exit(2); } return (T *)d[k].data(); }
template <class T> static size_t cnt(dump_t &d, const char *k) { return d[k].size() / sizeof(T); }
static std::vector<char *> strings(dump_t &d, const char *k) {
    std::vector<char *> out; std::vector<char> &v = d[k];
    for (size_t i = 0; i < v.size();) { out.push_back(v.data() + i); i += strlen(v.data() + i) + 1; }
    return out;
}
static std::vector<std::vector<unsigned int> > lists(dump_t &d, const char *k, int nq) {
    std::vector<std::vector<unsigned int> > out(nq); unsigned int *p = arr<unsigned int>(d, k);
    for (int q = 0; q < nq; q++) { unsigned int n = *p++; out[q].assign(p, p + n); p += n; }
    return out;
}

static int run_sa(dump_t &d, bool timing) {
    unsigned int *hdr = arr<unsigned int>(d, "header"); unsigned int n = hdr[0];
    int *str = arr<int>(d, "str");                        /* n + 3 zero pads */
    ref_t ref; memset(&ref, 0, sizeof ref);
    ref.toklen = n; ref.str = str;
    ref.sa = (int *)malloc(sizeof(int) * n);
    ref.buf = (int *)calloc((size_t)n * 4 + 4, sizeof(int));
    std::vector<int> tmp(str, str + n + 3);               /* suffixArrayInt reads s[n..n+2] = 0 */
    int last = str[n - 1];
    clock_t t0 = clock();
    suffixArrayConstruct(&ref, last, tmp.data());
    double sec = (double)(clock() - t0) / CLOCKS_PER_SEC;
    if (timing) { printf("%.6f\n", sec); return 0; }
    int *sa = arr<int>(d, "sa");
    for (unsigned int i = 0; i < n; i++) if (sa[i] != ref.sa[i]) { printf("SA MISMATCH at %u: oracle %d reference %d\n", i, sa[i], ref.sa[i]); return 1; }
    printf("SA OK n=%u\n", n);
    return 0;
}

static double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static int run_grammar(dump_t &d, char *outdir, bool timing) {
    unsigned int *hdr = arr<unsigned int>(d, "header");
    int nq = (int)hdr[7]; unsigned int G = hdr[9], D1 = hdr[10], D2 = hdr[11], sep1 = hdr[12], sep2a = hdr[13], sep2b = hdr[14];
    int *str = arr<int>(d, "str"), *tstr = arr<int>(d, "tstr");
    std::vector<char *> sv = strings(d, "svocab"), tv = strings(d, "tvocab"), names = strings(d, "blkname");
    names.resize(G + 1);
    saind_t *blocks = arr<saind_t>(d, "blocks");
    rule_onegap *r1 = arr<rule_onegap>(d, "r1"); unsigned int n1 = (unsigned int)cnt<rule_onegap>(d, "r1");
    rule_twogap *r2 = arr<rule_twogap>(d, "r2"); unsigned int n2 = (unsigned int)cnt<rule_twogap>(d, "r2");
    res_phrase_t *r0 = arr<res_phrase_t>(d, "r0"); int n0 = (int)cnt<res_phrase_t>(d, "r0");
    gapPattern *p1 = arr<gapPattern>(d, "p1"); gapPattern2 *p2 = arr<gapPattern2>(d, "p2");
    gappy_search *s1 = arr<gappy_search>(d, "s1"); two_gappy_search *s2 = arr<two_gappy_search>(d, "s2");
    oneGapOnSA *hits1 = arr<oneGapOnSA>(d, "hits1");
    ref_t ref; memset(&ref, 0, sizeof ref);
    ref.precomp_index = arr<precomp_st_end>(d, "pidx"); ref.featureMissingCount = arr<int>(d, "miss");

    std::vector<lexicalTask> maxEF(n1 + n2 + n0 + 1);
    unsigned int ntask = 0;
    int nl1 = 0, nl2 = 0, nl0 = 0;
    std::vector<red_dup_t> buf1(n1 + 1), buf2(n2 + 1), buf0(n0 + 1);
    const double w0 = wall_s();
    red_dup_t *lex1 = createLexiconGappyFast(str, tstr, &nl1, blocks, maxEF.data(), names, (int)G, r1, r2, n1, n2,
                                             (int)sep1, (int)sep2a, (int)sep2b, D1, D2, p1, p2, tv.data(), sv.data(),
                                             buf1.data(), s1, s2, &ntask, hits1, &ref);
    red_dup_t *lex2 = createLexiconTwoGapFast(str, tstr, &nl2, blocks, maxEF.data(), names, (int)G, r1, r2, n1, n2,
                                              (int)sep1, (int)sep2a, (int)sep2b, D1, D2, p1, p2, tv.data(), sv.data(),
                                              buf2.data(), s1, s2, &ntask, hits1, &ref);
    hashtbl_aux *lexic = NULL; hash_lexicon *tc = NULL, *fc = NULL;
    red_dup_t *lex0 = createLexiconFast(n0, r0, str, tstr, &lexic, &tc, &fc, &nl0, blocks, maxEF.data(), names, (int)G,
                                        buf0.data(), tv.data(), &ntask);
    const double w_lex = wall_s() - w0;

    /* the oracle's lexical tasks must be the reference's, field for field */
    lexicalTask *otask = arr<lexicalTask>(d, "tasks"); size_t on = cnt<lexicalTask>(d, "tasks");
    if (on != ntask) { printf("TASK COUNT MISMATCH oracle %zu reference %u\n", on, ntask); return 1; }
    for (unsigned int i = 0; i < ntask; i++) {
        const lexicalTask &a = otask[i], &b = maxEF[i];
        bool same = a.fastSpeedId == b.fastSpeedId && a.sourcePatternCounter == b.sourcePatternCounter && a.targetStart == b.targetStart && a.end == b.end;
        for (int j = 0; same && j < a.sourcePatternCounter; j++) same = a.sourcePattern[j] == b.sourcePattern[j];
        if ((int)i < nl1 + nl2) same = same && a.gap1 == b.gap1 && a.gap1_1 == b.gap1_1;
        if ((int)i >= nl1 && (int)i < nl1 + nl2) same = same && a.gap2 == b.gap2 && a.gap2_1 == b.gap2_1;
        if (!same) { printf("TASK MISMATCH at %u\n", i); return 1; }
    }
    /* MaxLex values come from the oracle (the kernel is .cu); scatter as ExtractPair.cu:3965-3982 */
    float *fe = arr<float>(d, "task_fe"), *ef = arr<float>(d, "task_ef");
    for (unsigned int i = 0; i < ntask; i++) {
        red_dup_t *e = (int)i < nl1 ? &lex1[maxEF[i].fastSpeedId] : (int)i < nl1 + nl2 ? &lex2[maxEF[i].fastSpeedId] : &lex0[maxEF[i].fastSpeedId];
        e->MaxLexFgivenE = fe[i]; e->MaxLexEgivenF = ef[i];
    }
    /* id -> [down,up] ranges */
    std::vector<result_t> rg1(2 * G + D1 + 1), rg2(G + 2 * D1 + D2 + 1), rg0(G + 1);
    for (size_t i = 0; i < rg1.size(); i++) rg1[i].up = rg1[i].down = -1;
    for (size_t i = 0; i < rg2.size(); i++) rg2[i].up = rg2[i].down = -1;
    for (size_t i = 0; i < rg0.size(); i++) rg0[i].up = rg0[i].down = -1;
    for (int i = 0; i < nl1; i++) { if (i == 0 || lex1[i].blocknumber != lex1[i - 1].blocknumber) rg1[lex1[i].blocknumber].down = i; rg1[lex1[i].blocknumber].up = i; }
    for (int i = 0; i < nl2; i++) { if (i == 0 || lex2[i].blocknumber != lex2[i - 1].blocknumber) rg2[lex2[i].blocknumber].down = i; rg2[lex2[i].blocknumber].up = i; }
    for (int i = 0; i < nl0; i++) { if (i == 0 || lex0[i].blocknumber != lex0[i - 1].blocknumber) rg0[lex0[i].blocknumber].down = i; rg0[lex0[i].blocknumber].up = i; }

    std::vector<std::vector<unsigned int> > qb = lists(d, "qblocks", nq), q1 = lists(d, "qone", nq), q2 = lists(d, "qtwo", nq);
    const double w1 = wall_s();
    print_query_GPU_Gappy(qb, q1, q2, nq, rg0.data(), rg1.data(), rg2.data(), lex0, lex1, lex2, outdir, G, D1, D2);
    if (timing) { printf("%.6f\n", w_lex + (wall_s() - w1)); return 0; }
    printf("GRAMMAR OK lex %d/%d/%d tasks %u\n", nl1, nl2, nl0, ntask);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_harness sa|time-sa <dump> | grammar|time-grammar <dump> <outdir>\n"); return 2; }
    dump_t d = read_dump(argv[2]);
    if (!strcmp(argv[1], "sa")) return run_sa(d, false);
    if (!strcmp(argv[1], "time-sa")) return run_sa(d, true);
    if (!strcmp(argv[1], "grammar") && argc >= 4) return run_grammar(d, argv[3], false);
    if (!strcmp(argv[1], "time-grammar") && argc >= 4) return run_grammar(d, argv[3], true);
    return 2;
}
