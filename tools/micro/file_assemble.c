/* file_assemble -- how fast can T host threads assemble many files (a few MB each) from ~1.5 KB pieces of a large
 * in-memory text?  Three ways: (A) pwritev of <= 1024 pieces per call (what the writer does), (B) pieces memcpy'd into a
 * per-thread buffer, one pwrite per file, (C) mmap(MAP_SHARED|MAP_POPULATE) of the (existing) file + memcpy.
 * usage: file_assemble <dir> <threads> <files> <file_MB> [source_GB] [methods-mask] [stride] [max piece bytes, default 3000] [source on huge pages 0/1]
 *   (files are written three times: fresh, then twice in place; methods-mask bit m enables method m, default 7;
 *    stride > 0 pins thread t to CPU t*stride -- e.g. 4 on a 64-core socket puts 16 threads on 16 different core pairs / 8 CCDs;
 *    the process CPU time is printed so that bytes per CPU-second can be compared under a cgroup CPU quota) */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static char *src; static uint64_t src_bytes; static uint64_t *seg_off; static uint32_t *seg_len; static uint64_t *fseg; static int nfiles; static const char *dir; static int method; static int next_file;
static uint64_t rnd(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static int stride;
static void *worker(void *arg) {
    if (stride > 0) { cpu_set_t cs; CPU_ZERO(&cs); CPU_SET((int)(intptr_t)arg * stride, &cs); (void)pthread_setaffinity_np(pthread_self(), sizeof cs, &cs); }
    struct iovec iov[1024]; char fn[512]; char *buf = NULL; size_t cap = 0;
    for (;;) {
        int f = __atomic_fetch_add(&next_file, 1, __ATOMIC_RELAXED);
        if (f >= nfiles) break;
        snprintf(fn, sizeof fn, "%s/f%d", dir, f);
        uint64_t s0 = fseg[f], s1 = fseg[f + 1], total = 0;
        for (uint64_t s = s0; s < s1; s++) total += seg_len[s];
        int fd = open(fn, method == 2 ? O_RDWR | O_CREAT : O_WRONLY | O_CREAT, 0644);
        if (fd < 0) { perror(fn); exit(1); }
        if (method == 0) {
            uint64_t pos = 0;
            for (uint64_t s = s0; s < s1;) { int n = 0; uint64_t want = 0; for (; s < s1 && n < 1024; s++, n++) { iov[n].iov_base = src + seg_off[s]; iov[n].iov_len = seg_len[s]; want += seg_len[s]; }
                if (pwritev(fd, iov, n, (off_t)pos) != (ssize_t)want) { perror("pwritev"); exit(1); } pos += want; }
        } else if (method == 1) {
            if (cap < total) { free(buf); buf = malloc(total); cap = total; }
            uint64_t pos = 0; for (uint64_t s = s0; s < s1; s++) { memcpy(buf + pos, src + seg_off[s], seg_len[s]); pos += seg_len[s]; }
            if (pwrite(fd, buf, total, 0) != (ssize_t)total) { perror("pwrite"); exit(1); }
        } else {
            if (ftruncate(fd, (off_t)total)) { perror("ftruncate"); exit(1); }
            char *m = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); exit(1); }
            uint64_t pos = 0; for (uint64_t s = s0; s < s1; s++) { memcpy(m + pos, src + seg_off[s], seg_len[s]); pos += seg_len[s]; }
            munmap(m, total);
        }
        if (method != 2 && ftruncate(fd, (off_t)total)) { perror("ftruncate"); exit(1); }
        close(fd);
    }
    free(buf);
    return NULL;
}
int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: file_assemble <dir> <threads> <files> <file_MB> [source_GB]\n"); return 2; }
    dir = argv[1]; int T = atoi(argv[2]); nfiles = atoi(argv[3]); double fmb = atof(argv[4]); double sgb = argc > 5 ? atof(argv[5]) : 2.0; int mask = argc > 6 ? atoi(argv[6]) : 7; stride = argc > 7 ? atoi(argv[7]) : 0; uint32_t pmax = argc > 8 ? (uint32_t)atoi(argv[8]) : 3000;   /* pieces of 200 .. pmax bytes */
    int thp = argc > 9 ? atoi(argv[9]) : 0;   /* 1: the source text on transparent huge pages (madvise) */
    src_bytes = (uint64_t)(sgb * (1 << 30)); if (posix_memalign((void **)&src, 2u << 20, src_bytes)) return 1; if (thp) madvise(src, src_bytes, MADV_HUGEPAGE); memset(src, 'x', src_bytes);
    uint64_t per = (uint64_t)(fmb * 1e6), nseg_est = (uint64_t)nfiles * (per / 200 + 2), ns = 0, st = 88172645463325252ull;
    seg_off = malloc(nseg_est * 8); seg_len = malloc(nseg_est * 4); fseg = malloc(((size_t)nfiles + 1) * 8);
    for (int f = 0; f < nfiles; f++) { fseg[f] = ns; for (uint64_t b = 0; b < per;) { uint32_t l = 200 + (uint32_t)(rnd(&st) % (pmax - 200)); seg_len[ns] = l; seg_off[ns] = rnd(&st) % (src_bytes - 4096); ns++; b += l; } }
    fseg[nfiles] = ns;
    const char *names[3] = {"pwritev(1024 pieces)", "memcpy+pwrite", "mmap(MAP_POPULATE)+memcpy"};
    for (method = 0; method < 3; method++) for (int pass = 0; pass < 3 && ((mask >> method) & 1); pass++) {
        next_file = 0; pthread_t th[256]; double t0 = now();
        struct rusage r0, r1; getrusage(RUSAGE_SELF, &r0);
        for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
        for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
        double dt = now() - t0; getrusage(RUSAGE_SELF, &r1);
        double cpu = (r1.ru_utime.tv_sec - r0.ru_utime.tv_sec) + (r1.ru_utime.tv_usec - r0.ru_utime.tv_usec) * 1e-6 + (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) + (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec) * 1e-6;
        printf("%-28s pass %d (%s): %.2f GB in %.3f s = %.1f GB/s with %d threads (stride %d), %.2f CPU-s = %.2f GB per CPU-s, %llu pieces\n", names[method], pass, pass ? "in place" : "fresh or first rewrite", nfiles * fmb / 1e3, dt, nfiles * fmb / 1e3 / dt, T, stride, cpu, nfiles * fmb / 1e3 / cpu, (unsigned long long)ns);
    }
    return 0;
}
