/*
 * oracle_main.c -- TEST INFRASTRUCTURE ONLY.
 * Command line of the CPU oracle; same six positionals as the reference's
 * bin/strmatchcuda (Main.c:35-61) plus optional "--dump <file>" for the intermediate dump and
 * "--long-sentences" (sentences of 255 tokens and more: positions of up to 1023 / 2039 instead of the reference's bytes).
 */
#include "cgx_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    const char *dump = NULL; const char *pos[6]; int np = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--long-sentences")) orc_set_long_sentences(1);   /* opt-in: wider positions (sentences of 255+ tokens) */
        else if (np < 6) pos[np++] = argv[i];
        else np++;
    }
    if (np != 6) { printf("\nCPU oracle for gappy extraction. Please check your input arguments.\n\n"); return 0; }
    orc_index *ix = orc_index_load(pos[0], pos[2], pos[3], pos[4]);
    orc_batch *b = orc_batch_load(ix, pos[1]);
    int rc = orc_run_all(ix, b, pos[5]);
    fprintf(stderr, "oracle: Q=%d T=%d G=%u D1=%u D2=%u H1=%u H2=%u rules=%u/%u/%u lines=%llu | lookup %.3fs gappy %.3fs extract %.3fs lexicon %.3fs lextask %.3fs write %.3fs\n",
            b->nq, b->ntok, b->g, b->d1, b->d2, b->h1, b->h2, b->n0, b->n1, b->n2, (unsigned long long)b->nlines,
            b->t_lookup, b->t_gappy, b->t_extract, b->t_lexicon, b->t_lextask, b->t_write);
    if (dump && orc_dump(ix, b, dump)) { fprintf(stderr, "cannot write %s\n", dump); rc = 1; }
    orc_batch_free(b); orc_index_free(ix);
    return rc ? 1 : 0;
}
