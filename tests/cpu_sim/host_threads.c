// TEST ONLY.  Harness for tools/tsan_host.sh: the host threads of the product (chunk-parallel loaders, file phase) under ThreadSanitizer; no GPU, no Python.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "cgx.h"
int main(int argc, char **argv) {
    const char *fx = argv[1]; char p[4][512]; const char *n[4] = {"corpus.f", "corpus.e", "corpus.a", "lex.txt"};
    for (int i = 0; i < 4; i++) snprintf(p[i], 512, "%s/%s", fx, n[i]);
    char err[256] = "";
    for (int rep = 0; rep < 3; rep++) {
        void *c = cgx_corpus_load_opt(p[0], p[1], p[2], p[3], 0, err, sizeof err);
        if (!c) { printf("load failed: %s\n", err); return 1; }
        printf("checksum %llx\n", (unsigned long long)cgx_corpus_checksum(c));
        cgx_corpus_free(c);
    }
    // file phase: 200 files of ~40 pieces each out of a 4 MB text
    const int nq = 200; const size_t T = 4u << 20; char *text = malloc(T); for (size_t i = 0; i < T; i++) text[i] = (char)('a' + i % 23);
    uint64_t *qseg = malloc((nq + 1) * 8); uint64_t *off = malloc(nq * 40 * 8); uint32_t *len = malloc(nq * 40 * 4); uint32_t s = 0; srand(1);
    for (int q = 0; q < nq; q++) { qseg[q] = s; for (int k = 0; k < 40; k++) { len[s] = 1 + rand() % 5000; off[s] = (uint64_t)rand() % (T - 6000); s++; } } qseg[nq] = s;
    double ms = 0;
    for (int rep = 0; rep < 2; rep++) { int rc = cgx_assemble_files(text, qseg, off, len, nq, 0, argv[2], 6, &ms); printf("assemble rc %d %.1f ms\n", rc, ms); if (rc) return 1; }
    return 0;
}
