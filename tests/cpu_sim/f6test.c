/* TEST ONLY: the writer's "%f" fast path (sb_f6 in cgx_host.c) against printf on many floats.
 * Includes the host translation unit to reach the static function; never calls the device ABI. */
#include "../../cgx_amd/csrc/cgx_host.c"
int main(void) {
    char b[64]; uint64_t bad = 0, n = 0; uint32_t st = 12345; sbuf s; memset(&s, 0, sizeof s);
    float specials[] = {0.0f, -0.0f, 1.0f / 128, 3.0f / 128, 0.5f, 99.0f, 495.0f, 1e-7f, 5e-7f, 4.9999999e-7f, 2.4771213f, 1e11f, 9.9999999e11f, 1e13f, INFINITY, -INFINITY};
    for (unsigned i = 0; i < sizeof specials / sizeof *specials; i++) { s.n = 0; sb_f6(&s, specials[i]); s.p[s.n] = 0; sprintf(b, "%f", (double)specials[i]); if (strcmp(s.p, b)) { printf("MISMATCH %a: %s vs %s\n", specials[i], s.p, b); bad++; } }
    for (uint64_t i = 0; i < 4000000ull; i++) {
        st = st * 1664525u + 1013904223u; uint32_t bits = st ^ (uint32_t)(i * 2654435761u); float x; memcpy(&x, &bits, 4);
        if (!(fabsf(x) < 1e9f)) continue;
        if ((i & 3) == 0) x = (float)((double)(bits >> 8) / 16777216.0 * 8.0);
        s.n = 0; sb_f6(&s, x); s.p[s.n] = 0; sprintf(b, "%f", (double)x); n++;
        if (strcmp(s.p, b)) { if (bad < 10) printf("MISMATCH %a: %s vs %s\n", x, s.p, b); bad++; }
    }
    for (int m = 1; m <= 24; m++) for (int k = 1; k < 4096; k += 2) { float x = (float)ldexp((double)k, -m); s.n = 0; sb_f6(&s, x); s.p[s.n] = 0; sprintf(b, "%f", (double)x); n++; if (strcmp(s.p, b)) { printf("TIE MISMATCH %a\n", x); bad++; } }
    printf("F6 %s checked %llu\n", bad ? "MISMATCH" : "OK", (unsigned long long)n);
    return bad != 0;
}
