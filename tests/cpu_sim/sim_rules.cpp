// sim_rules.cpp -- TEST ONLY.  Host build of the per-occurrence device functions in
// cgx_amd/csrc/cgx_rules.h (the same text the gfx950 kernels compile), driven over an oracle
// dump: every sampled occurrence of every block / pattern is pushed through
// cgx_extract_contig / _twogap / _onegap, the frequent-pair sweep through cgx_gap_ok and the
// lexical tasks through cgx_maxlex, and the results are compared with the oracle's arrays.
// This is the CPU sanitizer build of the kernel logic; the product never links it.
#include "../../cgx_amd/csrc/cgx_rules.h"
#include "../../oracle/cgx_oracle.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

typedef std::map<std::string, std::vector<char> > dump_t;
static dump_t read_dump(const char *path) {
    dump_t d; FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
    char tag[9]; unsigned long long nb;
    while (fread(tag, 1, 8, f) == 8 && fread(&nb, 8, 1, f) == 1) { tag[8] = 0; std::vector<char> v(nb); if (nb && fread(v.data(), 1, nb, f) != nb) exit(2); d[tag] = v; }
    fclose(f); return d;
}
template <class T> static const T *arr(dump_t &d, const char *k) { return (const T *)d[k].data(); }
template <class T> static size_t cnt(dump_t &d, const char *k) { return d[k].size() / sizeof(T); }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    dump_t d = read_dump(argv[1]);
    const uint32_t *hdr = arr<uint32_t>(d, "header");
    uint32_t n = hdr[0], G = hdr[9], D1 = hdr[10], D2 = hdr[11];
    cgx_view v{arr<int32_t>(d, "str"), arr<uint32_t>(d, "rlp"), arr<uint8_t>(d, "ltar"), arr<uint8_t>(d, "rtar"), n};
    const int32_t *sa = arr<int32_t>(d, "sa");
    const orc_block *blocks = arr<orc_block>(d, "blocks");
    const orc_gapsearch *s1 = arr<orc_gapsearch>(d, "s1"); const orc_twogapsearch *s2 = arr<orc_twogapsearch>(d, "s2");
    const orc_hit1 *h1 = arr<orc_hit1>(d, "hits1"); const orc_hit2 *h2 = arr<orc_hit2>(d, "hits2");
    const orc_prerange *pidx = arr<orc_prerange>(d, "pidx"); const orc_prehit *ph = arr<orc_prehit>(d, "phits");
    int bad = 0; unsigned guards = 0;

    // ---- the 64-byte target-side blocks (lr16 / lrs): built with cgx_lr_block_put, read with cgx_lr_block; for every start and
    // every length <= 16 the bytes a back-projection test sees must be those of the plain tables (normalised) ----
    {
        const size_t nt = d["ltar"].size();
        std::vector<uint8_t> blk(((nt + 64) / 8 + 8) * 64 + 64, 0xFF);
        uint8_t *tab = blk.data() + ((64 - ((uintptr_t)blk.data() & 63)) & 63);
        for (size_t w = 0; w < nt; w++) cgx_lr_block_put(tab, w, v.ltar[w], v.rtar[w]);
        for (size_t ts = 0; ts + 16 <= nt && bad == 0; ts++) {
            uint32_t a[5], b[5]; cgx_lr_block(tab, (int)ts, a, b);
            for (int k = 0; k < 16; k++) {
                const size_t w = ts + k; const int idx = (int)(ts & 3) + k;
                const uint8_t L = (uint8_t)(a[idx / 4] >> (8 * (idx & 3))), R = (uint8_t)(b[idx / 4] >> (8 * (idx & 3)));
                const bool none = v.ltar[w] == 255 || v.rtar[w] == 255;
                if (L != (none ? 255 : v.ltar[w]) || R != (none ? 0 : v.rtar[w])) { printf("lr block: word %zu of the span at %zu reads (%d,%d), tables say (%d,%d)\n", w, ts, L, R, v.ltar[w], v.rtar[w]); bad++; break; }
            }
        }
    }

    // ---- frequent-pair sweep (k_precomp) ----
    {
        const int32_t *freq = arr<int32_t>(d, "freq"); const int32_t *miss = arr<int32_t>(d, "miss");
        std::map<int32_t, int> rank; for (int j = 0; j < 100; j++) rank[freq[j]] = j;
        std::vector<std::tuple<uint32_t, uint32_t, int> > hits; std::vector<int> ms(10000, 0);
        for (uint32_t i = 0; i < n; i++) {
            auto ra = rank.find(v.str[i]); if (ra == rank.end() || v.str[i + 1] < 2) continue;
            for (int dd = 2; dd + 1 <= CGX_MAX_SPAN; dd++) {
                int32_t t = v.str[i + dd]; if (t < 2) break;
                auto rb = rank.find(t); if (rb == rank.end()) continue;
                uint32_t pair = (uint32_t)(ra->second * 100 + rb->second);
                if (cgx_gap_ok(v, i + 1, i + dd - 1)) hits.push_back(std::make_tuple(pair, i, dd)); else ms[pair]++;
            }
        }
        std::sort(hits.begin(), hits.end());
        size_t np = cnt<orc_prehit>(d, "phits");
        if (hits.size() != np) { printf("precomp: %zu hits vs oracle %zu\n", hits.size(), np); bad++; }
        else for (size_t i = 0; i < np; i++) if (std::get<1>(hits[i]) != ph[i].start || std::get<2>(hits[i]) != ph[i].length) { printf("precomp hit %zu differs\n", i); bad++; break; }
        for (int p = 0; p < 10000; p++) if (ms[p] != miss[p]) { printf("miss[%d] %d vs %d\n", p, ms[p], miss[p]); bad++; break; }
    }

    // ---- extraction ----
    std::vector<cgx_r0> A0; std::vector<cgx_r1> A1, C1; std::vector<cgx_r2> A2, B2, C2;
    for (uint32_t bn = 0; bn < G; bn++) {
        int nn = blocks[bn].end - blocks[bn].start + 1, S = nn < CGX_SAMPLER ? nn : CGX_SAMPLER;
        for (int k = 0; k < S; k++) {
            int x = cgx_sample_index(nn, CGX_SAMPLER, k);
            cgx_r0 a; cgx_r1 b, c; cgx_r2 e;
            if (cgx_extract_contig(v, (int32_t)bn, (int32_t)G, blocks[bn].matchlen, sa[blocks[bn].start + x], &a, &b, &c, &e)) guards++;
            if (a.valid) A0.push_back(a); if (b.valid) A1.push_back(b); if (c.valid) A1.push_back(c); if (e.valid) A2.push_back(e);
        }
    }
    for (uint32_t id = 0; id < D2; id++) {
        if (s2[id].sa_start == -1) continue;
        int nn = s2[id].sa_end - s2[id].sa_start + 1, S = nn < CGX_SAMPLER_TWOGAP ? nn : CGX_SAMPLER_TWOGAP;
        for (int k = 0; k < S; k++) {
            const orc_hit2 &h = h2[s2[id].sa_start + cgx_sample_index(nn, CGX_SAMPLER_TWOGAP, k)]; cgx_r2 r;
            if (cgx_extract_twogap(v, (int32_t)id, s1[s2[id].blockid].a_len, s1[s2[id].blockid].b_len, s2[id].c_len, h.str_position, h.length, h.length2, &r)) guards++;
            if (r.valid) B2.push_back(r);
        }
    }
    for (uint32_t id = 0; id < D1; id++) {
        if (s1[id].sa_start == -1) continue;
        int nn = s1[id].sa_end - s1[id].sa_start + 1; bool marker = nn == 1 && h1[s1[id].sa_start].length == 0; uint32_t base = 0;
        if (marker) { uint32_t pre = h1[s1[id].sa_start].str_position; base = pidx[pre].start; nn = (int)(pidx[pre].end - pidx[pre].start + 1); }
        int S = nn < CGX_SAMPLER_ONEGAP ? nn : CGX_SAMPLER_ONEGAP;
        for (int k = 0; k < S; k++) {
            int x = cgx_sample_index(nn, CGX_SAMPLER_ONEGAP, k); uint32_t cur; int fe;
            if (marker) { cur = ph[base + x].start; fe = ph[base + x].length; } else { cur = h1[s1[id].sa_start + x].str_position; fe = h1[s1[id].sa_start + x].length; }
            cgx_r1 a; cgx_r2 b, c;
            if (cgx_extract_onegap(v, (int32_t)id, (int32_t)D1, s1[id].a_len, s1[id].b_len, cur, fe, &a, &b, &c)) guards++;
            if (a.valid) C1.push_back(a); if (b.valid) C2.push_back(b); if (c.valid) C2.push_back(c);
        }
    }
    auto k1 = [](const cgx_r1 &r) { return std::make_tuple(r.id, r.tstart, r.end, r.gap1, r.gap1_1); };
    auto k2 = [](const cgx_r2 &r) { return std::make_tuple(r.id, r.tstart, r.end, r.gap1, r.gap1_1, r.gap2, r.gap2_1); };
    auto s1f = [&](std::vector<cgx_r1> &x) { std::sort(x.begin(), x.end(), [&](const cgx_r1 &a, const cgx_r1 &b) { return k1(a) < k1(b); }); };
    auto s2f = [&](std::vector<cgx_r2> &x) { std::sort(x.begin(), x.end(), [&](const cgx_r2 &a, const cgx_r2 &b) { return k2(a) < k2(b); }); };
    std::sort(A0.begin(), A0.end(), [](const cgx_r0 &a, const cgx_r0 &b) { return std::make_tuple(a.block, a.tar_start, a.tar_end) < std::make_tuple(b.block, b.tar_start, b.tar_end); });
    s1f(A1); s1f(C1); s2f(A2); s2f(B2); s2f(C2);
    const orc_rule0 *r0 = arr<orc_rule0>(d, "r0"); const orc_rule1 *r1 = arr<orc_rule1>(d, "r1"); const orc_rule2 *r2 = arr<orc_rule2>(d, "r2");
    size_t n0 = cnt<orc_rule0>(d, "r0"), n1 = cnt<orc_rule1>(d, "r1"), n2 = cnt<orc_rule2>(d, "r2");
    std::vector<cgx_r1> R1 = A1; R1.insert(R1.end(), C1.begin(), C1.end());
    std::vector<cgx_r2> R2 = A2; R2.insert(R2.end(), B2.begin(), B2.end()); R2.insert(R2.end(), C2.begin(), C2.end());
    if (A0.size() != n0 || R1.size() != n1 || R2.size() != n2 || A1.size() != hdr[12] || A2.size() != hdr[13] || A2.size() + B2.size() != hdr[14]) {
        printf("rule counts differ: %zu/%zu/%zu vs oracle %zu/%zu/%zu\n", A0.size(), R1.size(), R2.size(), n0, n1, n2); bad++;
    } else {
        for (size_t i = 0; i < n0; i++) if (A0[i].block != r0[i].block || A0[i].tar_start != r0[i].tar_start || A0[i].tar_end != r0[i].tar_end) { printf("r0[%zu] differs\n", i); bad++; break; }
        for (size_t i = 0; i < n1; i++) if (k1(R1[i]) != std::make_tuple(r1[i].id, r1[i].tstart, r1[i].end, r1[i].gap1, r1[i].gap1_1)) { printf("r1[%zu] differs\n", i); bad++; break; }
        for (size_t i = 0; i < n2; i++) if (k2(R2[i]) != std::make_tuple(r2[i].id, r2[i].tstart, r2[i].end, r2[i].gap1, r2[i].gap1_1, r2[i].gap2, r2[i].gap2_1)) { printf("r2[%zu] differs\n", i); bad++; break; }
    }

    // ---- lexical tasks ----
    {
        uint32_t nlex = hdr[5]; const orc_lexkey *lk = arr<orc_lexkey>(d, "lexk"); const orc_lexval *lv = arr<orc_lexval>(d, "lexv");
        std::vector<uint64_t> key(nlex); std::vector<float> v1(nlex), v2(nlex), n1v(nlex), n2v(nlex);
        for (uint32_t i = 0; i < nlex; i++) { key[i] = cgx_lexkey_pack(lk[i].src, lk[i].tgt); v1[i] = lv[i].v1; v2[i] = lv[i].v2; n1v[i] = -log10f(lv[i].v1); n2v[i] = -log10f(lv[i].v2); }
        uint32_t maxs = 0, maxt = 0;
        for (uint32_t i = 0; i < nlex; i++) { uint32_t a = (uint32_t)(key[i] >> 32), b = (uint32_t)key[i]; if (a > maxs) maxs = a; if (b > maxt) maxt = b; }
        uint32_t nrow = maxs + 1, ntgt = maxt;
        std::vector<uint32_t> row((size_t)nrow + 2, nlex); std::vector<int32_t> nullt((size_t)ntgt + 1, -1);
        for (uint32_t i = nlex; i-- > 0;) row[(size_t)(key[i] >> 32)] = i;
        for (size_t a = nrow; a-- > 0;) if (row[a] == nlex || row[a] > row[a + 1]) row[a] = row[a + 1];
        for (uint32_t i = 0; i < nlex && (key[i] >> 32) == 0; i++) { uint32_t b = (uint32_t)key[i]; if (b >= 1) nullt[b - 1] = (int32_t)i; }
        cgx_lexview t{key.data(), v1.data(), v2.data(), n1v.data(), n2v.data(), nlex, row.data(), nullt.data(), nrow, ntgt};
        const orc_lextask *tk = arr<orc_lextask>(d, "tasks"); size_t nt = cnt<orc_lextask>(d, "tasks");
        const float *fe = arr<float>(d, "task_fe"), *ef = arr<float>(d, "task_ef");
        size_t nl1 = cnt<int32_t>(d, "lex1_int") / 4, nl2 = cnt<int32_t>(d, "lex2_int") / 4;
        for (size_t i = 0; i < nt; i++) {
            int kind = i < nl1 ? 0 : i < nl1 + nl2 ? 1 : 2; float a, b; int32_t src[5]; for (int j = 0; j < 5; j++) src[j] = tk[i].src[j];
            cgx_maxlex(t, arr<int32_t>(d, "tstr"), src, tk[i].nsrc, tk[i].tstart, tk[i].end, tk[i].gap1, tk[i].gap1_1, tk[i].gap2, tk[i].gap2_1, kind, &a, &b);
            if (memcmp(&a, &fe[i], 4) || memcmp(&b, &ef[i], 4)) { printf("lex task %zu: %a/%a vs oracle %a/%a\n", i, a, b, fe[i], ef[i]); bad++; break; }
        }
    }
    printf("%s guards=%u rules=%zu/%zu/%zu\n", bad ? "SIM MISMATCH" : "SIM OK", guards, A0.size(), R1.size(), R2.size());
    return bad ? 1 : 0;
}
