/*
 * strmatchcuda.c -- command line of the MI355X extractor; same interface as the reference's
 * bin/strmatchcuda (Main.c:29-86):
 *   strmatchcuda [-h] [-l minmatchlen] [-t fingerlen] [-s timefile] <src> <query> <tgt> <align> <lex> <outdir>
 * Extra, optional: --device N (default 0), --shard i/n (this process handles the i-th of n
 * contiguous query shards balanced by token count; one process per GPU), --index-cache FILE (load the parsed
 * corpus from FILE and the built index -- suffix array, frequent-pair lists, tables -- from FILE.idx if they exist and
 * still match the text files; otherwise parse / build and write them for the next run), --gz LEVEL (1..9: write
 * grammar.<q>.s.gz through zlib instead of plain files), --sub-batch N (queries per internal batch; default: as many as
 * hold 300 000 query tokens.  Any split writes the same files), --long-sentences (accept sentence pairs of 255 tokens and more,
 * which the reference rejects: source < 1024, target < 2040 tokens; shorter sentences give the same files with or without it;
 * the caches remember the mode they were written in), --query-limit N (query tokens per sentence that are looked up: the reference's first kernel launches
 * 128 threads per query sentence, SuffixArray.cu:1374-1378, so tokens from the 129th on never match -- the default here too;
 * N = 0 lifts the limit, any other N sets it).
 */
#include "../../include/cgx.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>

static void print_help(void) {
    printf("\nGPU source codes for gappy extraction. Please check your input arguments.\n\n");
    exit(0);
}

int main(int argc, char **argv) {
    int minmatchlen = 1, fingerlen = 10, device = 0, shard = 0, nshard = 1, gz = 0, sub_batch = 0, long_sentences = 0, query_limit = -1; const char *timefile = NULL, *cache = NULL;
    /* pull the long options out first so getopt sees the reference's grammar only */
    char **av = malloc(sizeof(char *) * (size_t)(argc + 1)); int ac = 0;
    for (int i = 0; i < argc; i++) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--index-cache") && i + 1 < argc) cache = argv[++i];
        else if (!strcmp(argv[i], "--gz") && i + 1 < argc) { gz = atoi(argv[++i]); if (gz < 1 || gz > 9) print_help(); }
        else if (!strcmp(argv[i], "--long-sentences")) long_sentences = 1;
        else if (!strcmp(argv[i], "--query-limit") && i + 1 < argc) { query_limit = atoi(argv[++i]); if (query_limit < 0) print_help(); }
        else if (!strcmp(argv[i], "--sub-batch") && i + 1 < argc) { sub_batch = atoi(argv[++i]); if (sub_batch < 1) print_help(); }
        else if (!strcmp(argv[i], "--shard") && i + 1 < argc) { if (sscanf(argv[++i], "%d/%d", &shard, &nshard) != 2 || nshard < 1 || shard < 0 || shard >= nshard) print_help(); }
        else av[ac++] = argv[i];
    }
    av[ac] = NULL;
    int ch, errflg = 0;
    while (!errflg && (ch = getopt(ac, av, "hl:t:s:")) != EOF) {
        switch (ch) {
        case 'h': print_help(); break;
        case '?': fprintf(stderr, "Unknown option %c\n", optopt); errflg = 1; break;
        case 'l': minmatchlen = atoi(optarg); break;
        case 't': fingerlen = atoi(optarg); break;
        case 's': timefile = optarg; break;
        default: errflg = 1; break;
        }
    }
    if (optind != ac - 6 || errflg) print_help();
    if (fingerlen > 10 || fingerlen <= 0) { fprintf(stderr, "finger length must be between 1 and 10\n"); exit(0); }
    const char *src = av[optind], *qry = av[optind + 1], *tgt = av[optind + 2], *ali = av[optind + 3], *lex = av[optind + 4], *out = av[optind + 5];
    (void)timefile;                                       /* parsed, unused: recordTime is commented out in the reference (Start.cu:621) */
    fprintf(stderr, "reference file: %s\nquery file: %s\nref target file: %s\nalign file: %s\nMinimum match length: %d\n", src, qry, tgt, ali, minmatchlen);

    FILE *qf = fopen(qry, "r");
    if (!qf) { fprintf(stderr, "Can not open query file \"%s\"\n", qry); return 0; }   /* diskInit failure: start() returns (Start.cu:528) */
    fclose(qf);

    char err[512];
    cgx_corpus *corpus = cache ? cgx_corpus_load_cache(cache, err, sizeof err) : NULL;
    if (cache && corpus && cgx_corpus_matches_sources(corpus, src, tgt, ali, lex) == 0) {   /* a text file changed since the cache was written: parse again */
        fprintf(stderr, "strmatchcuda: corpus cache %s is older than the text files, rebuilding it\n", cache);
        cgx_corpus_free(corpus); corpus = NULL;
    }
    if (cache && corpus && (cgx_corpus_flags(corpus) & CGX_CORPUS_LONG_SENTENCES) != (long_sentences ? CGX_CORPUS_LONG_SENTENCES : 0)) {   /* parsed under the other position width */
        fprintf(stderr, "strmatchcuda: corpus cache %s was written %s --long-sentences, rebuilding it\n", cache, long_sentences ? "without" : "with");
        cgx_corpus_free(corpus); corpus = NULL;
    }
    if (cache && corpus) fprintf(stderr, "strmatchcuda: corpus read from cache %s\n", cache);
    if (!corpus) {
        corpus = cgx_corpus_load_opt(src, tgt, ali, lex, long_sentences ? CGX_CORPUS_LONG_SENTENCES : 0, err, sizeof err);
        if (corpus && cache && cgx_corpus_save(corpus, cache) != CGX_OK) fprintf(stderr, "strmatchcuda: could not write the corpus cache %s\n", cache);
    }
    if (!corpus) {
        if (!strncmp(err, "Not possible, too long", 22)) { printf("%s\n", err); return 1; }
        if (!strcmp(err, "Not possible!")) { printf("%s\n", err); return 0; }
        fprintf(stderr, "%s\n", err); return 0;
    }
    struct stat sb;
    if (stat(out, &sb) || !S_ISDIR(sb.st_mode)) {           /* PrintResults.c:443-446 */
        fprintf(stderr, "Please check your file directory address for grammar rule files output. It is not valid. Program Exits.\n");
        return 0;
    }
    cgx_ctx *ctx = cgx_create(device);
    if (!ctx) { fprintf(stderr, "strmatchcuda: no usable MI355X/HIP device %d\n", device); return 2; }
    int rc = CGX_ERR_IO; char *idx = NULL;
    if (cache) {                                             /* the built index next to the parsed corpus: <cache>.idx */
        idx = malloc(strlen(cache) + 8); sprintf(idx, "%s.idx", cache);
        rc = cgx_index_load(ctx, idx, cgx_corpus_checksum(corpus));
        if (rc == CGX_OK) fprintf(stderr, "strmatchcuda: index read from cache %s\n", idx);
    }
    if (rc != CGX_OK) {
        rc = cgx_corpus_upload(ctx, corpus);
        if (rc != CGX_OK) { fprintf(stderr, "strmatchcuda: %s\n", cgx_last_error(ctx)); return 2; }
        if (idx && cgx_index_save(ctx, idx, cgx_corpus_checksum(corpus)) != CGX_OK) fprintf(stderr, "strmatchcuda: could not write the index cache %s\n", idx);
    }
    free(idx);
    uint64_t nrules = 0;
    if (gz) (void)cgx_set_option(ctx, "gz_level", gz);       /* grammar.<q>.s.gz instead of grammar.<q>.s */
    if (sub_batch) (void)cgx_set_option(ctx, "sub_batch", sub_batch);
    if (query_limit >= 0) (void)cgx_set_option(ctx, "k1_limit", query_limit);   /* 0: no limit */
    (void)cgx_set_option(ctx, "async_write", 1);             /* large query files run as several internal batches: write batch k while batch k+1 is on the GPU */
    rc = nshard == 1 ? cgx_extract_grammars(ctx, corpus, qry, out, 0, -1, &nrules)
                     : cgx_extract_grammars_shard(ctx, corpus, qry, out, shard, nshard, &nrules);   /* contiguous shards balanced by token count */
    if (rc == CGX_OK) rc = cgx_flush(ctx);                   /* every file is on disk (or its error reported) before the summary */
    if (rc != CGX_OK) { fprintf(stderr, "strmatchcuda: %s (%d)\n", cgx_last_error(ctx), rc); return rc == CGX_ERR_IO ? 0 : 2; }
    fprintf(stderr, "strmatchcuda: %llu rules | index: suffix array %.1f ms, frequent pairs %.1f ms | last batch: lookup %.3f, blocks %.3f, gappy %.3f, extract %.3f, lexicon %.3f, text layout %.3f ms | files written in %.1f ms\n",
            (unsigned long long)nrules, cgx_stage_ms(ctx, "build_sa"), cgx_stage_ms(ctx, "precompute"), cgx_stage_ms(ctx, "sa_lookup"), cgx_stage_ms(ctx, "blocks"),
            cgx_stage_ms(ctx, "gappy"), cgx_stage_ms(ctx, "extract"), cgx_stage_ms(ctx, "lexicon"), cgx_stage_ms(ctx, "format"), cgx_host_ms(ctx, "write"));
    cgx_destroy(ctx); cgx_corpus_free(corpus); free(av);
    return 0;
}
