// TEST ONLY: the formatter's DEFLATE output (cgx_amd/csrc/cgx_fmt.h) built for the host.
// Random lexicons over a random vocabulary are laid out group by group exactly as k_fmt_lines_gz / k_gz_groups / k_gz_files do it
// (plain count -> bit count -> scans -> bit-granular write -> CRC fold): every group's bytes, closed by an empty final block, are
// inflated by zlib as a raw deflate stream and compared with the plain text of the group; files made of random runs of groups
// between the gzip header and the trailer (03 00, CRC-32, ISIZE) are read back as gzip members.  Never calls the device ABI.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>
#include "../../cgx_amd/csrc/cgx_fmt.h"

static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
static uint32_t rr(uint32_t n) { return rnd() % n; }
// FNV-1a over every plain line and every byte of deflate output of the run: printed at the end, so that a change of the formatter can be
// checked for "same bytes as before" on the CPU (tests/test_cpu_sim.py pins the values of the formatter that made the golden files)
static uint64_t dig_plain = 1469598103934665603ull, dig_gz = 1469598103934665603ull;
static void dig(uint64_t &h, const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }

struct Case {
    std::vector<char> spool, tpool; std::vector<uint32_t> soff, toff;
    std::vector<int32_t> str, tstr; std::vector<cgx_block> blocks; std::vector<cgx_gappat> p1d; std::vector<int32_t> c2d; std::vector<uint32_t> one2;
    std::vector<float> aa, bb, fs;
    std::vector<cgx_lexent> lex[3]; std::vector<int32_t> rng[3];
    uint32_t G, D1, D2;
    fmt_view F;
};
static void make_vocab(std::vector<char> &pool, std::vector<uint32_t> &off, uint32_t n, bool utf8) {
    off.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        off[i] = (uint32_t)pool.size();
        if (rr(7) == 0) continue;                               // a word without a spelling: printed as s<id> / t<id>
        uint32_t len = 1 + rr(rr(4) ? 9 : 40);
        for (uint32_t k = 0; k < len; k++) { char c; if (utf8 && rr(9) == 0) c = (char)(0x80 + rr(0x7F)); else c = (char)('a' + rr(26)); pool.push_back(c); }
    }
    off[n] = (uint32_t)pool.size();
    for (int k = 0; k < 16; k++) pool.push_back('#');           // the word sinks read whole 8-byte words
}
static void make_case(Case &c, uint32_t G, uint32_t D1, uint32_t D2, uint32_t ns, uint32_t nt, bool utf8, uint32_t maxlines) {
    c.G = G; c.D1 = D1; c.D2 = D2;
    make_vocab(c.spool, c.soff, ns, utf8); make_vocab(c.tpool, c.toff, nt, utf8);
    // ids past the vocabulary print as numbers too: one, two and three digit groups, and beyond 10^6 (the general digit loop)
    auto any_id = [&](uint32_t nv) { const uint32_t k = rr(16); return (int32_t)(k == 0 ? nv + rr(900) : k == 1 ? nv + rr(99000) : k == 2 ? 999990 + rr(20) : k == 3 ? 1000000 + rr(2000000000u) : rr(nv + 3)); };
    c.str.resize(4096); for (auto &x : c.str) x = any_id(ns);
    c.tstr.resize(8192); for (auto &x : c.tstr) x = any_id(nt);
    c.blocks.resize(G); for (auto &b : c.blocks) { b.start = 0; b.end = 0; b.matchlen = 1 + (int32_t)rr(5); b.string_start = (int32_t)rr(4000); }
    c.p1d.resize(D1 ? D1 : 1);
    for (auto &p : c.p1d) { int a = 1 + (int)rr(2), bn = 1 + (int)rr(2); p.number = (uint8_t)(a + 1 + bn); int j = 0; for (int k = 0; k < a; k++) p.pat[j++] = (int32_t)rr(ns); p.pat[j++] = -1; for (int k = 0; k < bn; k++) p.pat[j++] = (int32_t)rr(ns); while (j < 5) p.pat[j++] = -2; }
    c.c2d.resize(D2 ? D2 : 1); c.one2.resize(D2 ? D2 : 1);
    for (uint32_t i = 0; i < D2; i++) { c.c2d[i] = (int32_t)rr(ns); c.one2[i] = D1 ? rr(D1) : 0; }
    c.aa.assign(FMT_TABN * FMT_TABN, 0.f); c.bb.assign(FMT_TABN, 0.f); c.fs.assign(FMT_TABN, 0.f);
    for (int p = 0; p < FMT_TABN; p++) { c.bb[p] = (float)log10((double)(1 + p)); c.fs[p] = (float)log10((double)(1 + p)); for (int f = 1; f < FMT_TABN; f++) c.aa[p * FMT_TABN + f] = p ? -log10f((float)p / (float)f) : 0.f; }
    const uint32_t nid[3] = {G, 2 * G + D1, G + 2 * D1 + D2};
    static const float vals[] = {0.30103f, 1.5f, 99.0f, 12.25f, 0.0f, 3.1415927f, 123.456f, 1e-7f, 7.0f, 2.4771213f};
    for (int kind = 0; kind < 3; kind++) {
        c.rng[kind].assign(2 * (size_t)nid[kind] + 2, -1);
        for (uint32_t id = 0; id < nid[kind]; id++) {
            uint32_t n = rr(3) == 0 ? 0 : 1 + rr(rr(5) == 0 ? maxlines : 6);
            if (!n) continue;
            c.rng[kind][2 * id] = (int32_t)c.lex[kind].size(); c.rng[kind][2 * id + 1] = (int32_t)(c.lex[kind].size() + n - 1);
            const uint16_t fsample = (uint16_t)(1 + rr(301));
            for (uint32_t k = 0; k < n; k++) {
                cgx_lexent e; memset(&e, 0, sizeof e);
                e.id = (int32_t)id; e.kind = (uint8_t)kind; e.tstart = rr(8000); e.end = (uint8_t)rr(rr(6) ? 4 : 15);
                e.gap1 = e.gap1_1 = e.gap2 = e.gap2_1 = 200;
                if (kind >= 1) { e.gap1 = (uint8_t)rr(e.end + 1); e.gap1_1 = (uint8_t)(e.gap1 + rr(e.end - e.gap1 + 1)); }
                if (kind >= 2 && e.gap1_1 + 2 <= e.end) { e.gap2 = (uint8_t)(e.gap1_1 + 2 + rr(e.end - e.gap1_1 - 1)); e.gap2_1 = (uint8_t)(e.gap2 + rr(e.end - e.gap2 + 1)); }
                e.fsample = fsample; e.paircount = (uint16_t)(rr(3) ? 1 : 1 + rr(fsample < 40 ? fsample : 40)); e.f = (uint16_t)(rr(2) ? 1 : 1 + rr(300));
                e.fe = rr(3) ? vals[rr(10)] : (float)rr(2000000) / 65536.0f; e.ef = rr(3) ? vals[rr(10)] : (float)rr(2000000) / 32768.0f;
                if (k && rr(4) == 0) { cgx_lexent p = c.lex[kind].back(); p.id = e.id; if (rr(2)) p.tstart = e.tstart; c.lex[kind].push_back(p); continue; }   // (nearly) the line before it once more
                c.lex[kind].push_back(e);
            }
        }
    }
    fmt_view &F = c.F;
    for (int k = 0; k < 3; k++) { F.lex[k] = c.lex[k].data(); F.rng[k] = c.rng[k].data(); }
    F.blocks = c.blocks.data(); F.p1d = c.p1d.data(); F.c2d = c.c2d.data(); F.one2 = c.one2.data(); F.str = c.str.data(); F.tstr = c.tstr.data();
    F.spool = c.spool.data(); F.tpool = c.tpool.data(); F.soff = c.soff.data(); F.toff = c.toff.data(); F.ns = ns; F.nt = nt;
    F.aa = c.aa.data(); F.bb = c.bb.data(); F.fs = c.fs.data(); F.G = G; F.D1 = D1; F.D2 = D2;
}

static int inflate_member(const unsigned char *src, size_t n, std::string &out, size_t *used, int wbits = 16 + MAX_WBITS) {
    z_stream s; memset(&s, 0, sizeof s);
    if (inflateInit2(&s, wbits) != Z_OK) return -1;
    s.next_in = (Bytef *)src; s.avail_in = (uInt)n;
    unsigned char buf[65536]; int rc;
    do { s.next_out = buf; s.avail_out = sizeof buf; rc = inflate(&s, Z_NO_FLUSH); if (rc != Z_OK && rc != Z_STREAM_END) { fprintf(stderr, "inflate: %d %s\n", rc, s.msg ? s.msg : ""); inflateEnd(&s); return -1; } out.append((char *)buf, sizeof buf - s.avail_out); } while (rc != Z_STREAM_END);
    *used = n - s.avail_in; inflateEnd(&s);
    return 0;
}

// the writing sink of the simulation: bits OR-ed into a zeroed buffer from a bit position on (what BitWordSink does with words and lanes)
struct BitBuf {
    unsigned char *p; uint64_t at; uint32_t n;
    void bits(uint32_t code, uint32_t len) { for (uint32_t i = 0; i < len; i++, at++) if ((code >> i) & 1u) p[at >> 3] |= (unsigned char)(1u << (at & 7)); n += len; }
    void align() { while (at & 7) { at++; n++; } }
};
static int run_case(uint32_t G, uint32_t D1, uint32_t D2, uint32_t ns, uint32_t nt, bool utf8, uint32_t maxlines, bool dynamic, uint64_t *nlines_out, uint64_t *plain_out, uint64_t *gz_out, uint64_t *hdr_out) {
    Case c; make_case(c, G, D1, D2, ns, nt, utf8, maxlines);
    const fmt_view &F = c.F;
    const uint32_t NC = 4 * G + 3 * D1 + D2, NG = G + D1 + D2;
    // canonical items -> lines (k_fmt_canon_lines / k_fmt_canon_ent)
    std::vector<uint32_t> cstart(NC + 1, 0), line_ent, line_c;
    for (uint32_t ci = 0; ci < NC; ci++) {
        int kind; uint32_t cid; canon_item(ci, G, D1, D2, &kind, &cid);
        cstart[ci] = (uint32_t)line_ent.size();
        const int32_t dn = F.rng[kind][2 * (size_t)cid], up = F.rng[kind][2 * (size_t)cid + 1];
        if (dn < 0 || up < 0) continue;
        for (int32_t l = dn; l <= up; l++) { line_ent.push_back(((uint32_t)kind << 30) | (uint32_t)l); line_c.push_back(ci); }
    }
    const uint32_t NU = (uint32_t)line_ent.size(); cstart[NC] = NU;
    uint32_t tab[GZ_TAB_WORDS]; gz_make_tables(tab);
    for (int t = 0; t < 200; t++) {                             // the group-wise CRC against zlib, every group length
        unsigned char buf[64]; uint32_t crc = 0xFFFFFFFFu; size_t at = 0; for (auto &c : buf) c = (unsigned char)rr(256);
        while (at < 56) { const uint32_t n = 1 + rr(8); uint64_t v; memcpy(&v, buf + at, 8); crc = gz_crc_group(tab, crc, v, n); at += n; }
        if ((crc ^ 0xFFFFFFFFu) != (uint32_t)crc32(0, buf, (uInt)at)) { fprintf(stderr, "gz_crc_group disagrees with zlib\n"); return 1; }
    }
    // pass 1: plain lengths; the plain text itself is the expectation
    std::vector<uint32_t> len_u(NU), tail_u(NU); std::vector<uint64_t> U(NU + 1, 0); std::vector<std::string> plain(NU);
    for (uint32_t l = 0; l < NU; l++) {
        const int kind = (int)(line_ent[l] >> 30); const cgx_lexent &e = F.lex[kind][line_ent[l] & 0x3FFFFFFFu];
        CountSink cs{0}; fmt_line(cs, F, kind, e, &tail_u[l]); len_u[l] = cs.n;
        std::vector<char> buf(cs.n + 16); MemSink ms{buf.data()}; if (!fmt_line(ms, F, kind, e)) { fprintf(stderr, "plain line not representable\n"); return 1; }
        if ((uint32_t)(ms.p - buf.data()) != cs.n) { fprintf(stderr, "plain count != plain write\n"); return 1; }
        plain[l].assign(buf.data(), cs.n); U[l + 1] = U[l] + cs.n;
    }
    auto group_of = [&](uint32_t l, uint32_t *l0, uint32_t *l1) { const uint32_t g = canon_group(line_c[l], G, D1); *l0 = cstart[canon_group_first(g, G, D1)]; *l1 = cstart[canon_group_first(g + 1, G, D1)]; return g; };
    auto place = [&](uint32_t l, gz_place &P) {
        uint32_t l0, l1; group_of(l, &l0, &l1);
        memset(&P, 0, sizeof P);
        P.first = l == l0; P.last = l + 1 == l1; P.same_item = l > cstart[line_c[l]];
        if (!P.first) { const int pk = (int)(line_ent[l - 1] >> 30); P.pe = &F.lex[pk][line_ent[l - 1] & 0x3FFFFFFFu]; P.prev_len = len_u[l - 1]; P.prev_tail = tail_u[l - 1]; }
    };
    // the batch's Huffman codes (gz_codes in cgx_format.inc): a tally of every 16th block of 256 lines, +1 for every symbol the text may hold
    gz_code C;
    if (dynamic) {
        unsigned int hist[GZ_NSYM] = {0};
        for (uint32_t l = 0; l < NU; l++) {
            if ((l / 256) % 16 != 0) continue;
            const int kind = (int)(line_ent[l] >> 30); const cgx_lexent &e = F.lex[kind][line_ent[l] & 0x3FFFFFFFu];
            gz_place pl; place(l, pl);
            BitCount bc{0}; GzSink<BitCount, false, GZ_TALLY> z(bc, nullptr, nullptr, hist);
            (void)fmt_line_gz(bc, z, F, &C, kind, e, pl);
        }
        unsigned char lit[256] = {0};
        for (const char *p = "[X] ||| ,12=.-?\n0123456789stEgivenFCoherent SampleCountF CountEF MaxLexFgivenE MaxLexEgivenF IsSingletonF IsSingletonFE"; *p; p++) lit[(unsigned char)*p] = 1;
        for (char ch : c.spool) lit[(unsigned char)ch] = 1;
        for (char ch : c.tpool) lit[(unsigned char)ch] = 1;
        uint64_t f[GZ_NSYM];
        for (int i = 0; i < GZ_NSYM; i++) f[i] = (uint64_t)hist[i] * 16 + ((i >= 256 || lit[i]) ? 1u : 0u);
        if (!gz_build_dynamic(f, C)) { fprintf(stderr, "the dynamic block header does not fit\n"); return 1; }
        for (int i = 0; i < GZ_NSYM; i++) if (f[i] && (C.sym[i] >> 16) == 0) { fprintf(stderr, "symbol %d has a frequency and no code\n", i); return 1; }
        if (C.hdr_bits > *hdr_out) *hdr_out = C.hdr_bits;
    } else { gz_build_fixed(C); *hdr_out = C.hdr_bits; }
    // pass 2: bits per line, their running sum, bytes and byte offsets of the groups (k_gz_group_bytes)
    std::vector<uint32_t> bits(NU); std::vector<uint64_t> P(NU + 1, 0), GO(NG + 1, 0);
    for (uint32_t l = 0; l < NU; l++) {
        const int kind = (int)(line_ent[l] >> 30); const cgx_lexent &e = F.lex[kind][line_ent[l] & 0x3FFFFFFFu];
        gz_place pl; place(l, pl);
        BitCount bc{0}; GzSink<BitCount, false, GZ_COUNT> z(bc, nullptr, C.sym);
        if (!fmt_line_gz(bc, z, F, &C, kind, e, pl)) { fprintf(stderr, "gz count: line %u not representable\n", l); return 1; }
        if (z.pos != len_u[l]) { fprintf(stderr, "gz sink saw %u characters, the plain line has %u\n", z.pos, len_u[l]); return 1; }
        if (bc.n < 32u) { fprintf(stderr, "line %u has %u bits: the writing sink wants more than a word\n", l, bc.n); return 1; }
        bits[l] = bc.n; P[l + 1] = P[l] + bc.n;
    }
    for (uint32_t g = 0; g < NG; g++) { const uint32_t l0 = cstart[canon_group_first(g, G, D1)], l1 = cstart[canon_group_first(g + 1, G, D1)]; GO[g + 1] = GO[g] + (l1 > l0 ? gz_group_bytes(P[l1] - P[l0]) : 0); }
    // pass 3: write at bit granularity, CRC contributions
    std::vector<unsigned char> text(GO[NG] + 64, 0); std::vector<uint32_t> contrib(NU);
    for (uint32_t l = 0; l < NU; l++) {
        const int kind = (int)(line_ent[l] >> 30); const cgx_lexent &e = F.lex[kind][line_ent[l] & 0x3FFFFFFFu];
        gz_place pl; place(l, pl); uint32_t l0, l1; const uint32_t g = group_of(l, &l0, &l1);
        BitBuf bo{text.data(), 8 * GO[g] + (P[l] - P[l0]), 0}; GzSink<BitBuf, true, GZ_WRITE> z(bo, tab, C.sym);
        if (!fmt_line_gz(bo, z, F, &C, kind, e, pl)) { fprintf(stderr, "gz write: line %u not representable\n", l); return 1; }
        if (bo.n != bits[l]) { fprintf(stderr, "gz count %u != gz write %u at line %u\n", bits[l], bo.n, l); return 1; }
        if (pl.last) { gz_stored(bo); if (bo.at != 8 * GO[g + 1]) { fprintf(stderr, "group %u ends at bit %llu, laid out to end at %llu\n", g, (unsigned long long)bo.at, (unsigned long long)(8 * GO[g + 1])); return 1; } }
        contrib[l] = gz_multmodp(gz_x8n(tab, U[l1] - U[l + 1]), z.crc ^ 0xFFFFFFFFu);
    }
    // groups (k_gz_groups): CRC fold, and every group by itself as a raw deflate stream closed by an empty final block
    std::vector<uint32_t> gcrc(NG, 0), glen(NG, 0), gpow(NG, 0);
    static const unsigned char FINAL[2] = {3, 0};
    for (uint32_t g = 0; g < NG; g++) {
        const uint32_t l0 = cstart[canon_group_first(g, G, D1)], l1 = cstart[canon_group_first(g + 1, G, D1)];
        if (l0 == l1) { if (GO[g + 1] != GO[g]) return 1; continue; }
        uint32_t crc = 0; for (uint32_t l = l0; l < l1; l++) crc ^= contrib[l];
        gcrc[g] = crc; glen[g] = (uint32_t)(U[l1] - U[l0]); gpow[g] = gz_x8n(tab, glen[g]);
        std::string want; for (uint32_t l = l0; l < l1; l++) want += plain[l];
        if ((uint32_t)crc32(0, (const Bytef *)want.data(), (uInt)want.size()) != crc) { fprintf(stderr, "CRC fold of group %u is wrong\n", g); return 1; }
        std::vector<unsigned char> one(text.begin() + (long)GO[g], text.begin() + (long)GO[g + 1]); one.insert(one.end(), FINAL, FINAL + 2);
        std::string got; size_t used = 0;
        if (inflate_member(one.data(), one.size(), got, &used, -MAX_WBITS)) { fprintf(stderr, "group %u does not inflate\n", g); return 1; }
        if (used != one.size()) { fprintf(stderr, "group %u: %zu bytes inflated, laid out as %zu\n", g, used, one.size()); return 1; }
        if (got != want) { fprintf(stderr, "group %u inflates to other text\n", g); return 1; }
    }
    // files (k_gz_files and the host writer): random runs of groups, in any order, between header and trailer
    static const unsigned char HDR[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    for (int f = 0; f < 12; f++) {
        std::vector<unsigned char> file(HDR, HDR + 10); std::string want; uint32_t crc = 0, len = 0;
        const uint32_t runs = f == 0 ? 0 : 1 + rr(6);
        for (uint32_t r = 0; r < runs && NG; r++) {
            const uint32_t g0 = rr(NG), g1 = g0 + 1 + rr(NG - g0 < 9 ? NG - g0 : 9);
            file.insert(file.end(), text.begin() + (long)GO[g0], text.begin() + (long)GO[g1]);
            for (uint32_t g = g0; g < g1; g++) {
                if (!glen[g]) continue;
                crc = gz_multmodp(gpow[g], crc) ^ gcrc[g]; len += glen[g];
                for (uint32_t l = cstart[canon_group_first(g, G, D1)]; l < cstart[canon_group_first(g + 1, G, D1)]; l++) want += plain[l];
            }
        }
        file.push_back(3); file.push_back(0);
        for (int k = 0; k < 4; k++) file.push_back((unsigned char)(crc >> (8 * k)));
        for (int k = 0; k < 4; k++) file.push_back((unsigned char)(len >> (8 * k)));
        std::string got; size_t used = 0;
        if (inflate_member(file.data(), file.size(), got, &used)) { fprintf(stderr, "file %d is not a gzip member (CRC / ISIZE / stream)\n", f); return 1; }
        if (used != file.size() || got != want) { fprintf(stderr, "file %d inflates to other text\n", f); return 1; }
    }
    for (uint32_t l = 0; l < NU; l++) dig(dig_plain, plain[l].data(), plain[l].size());
    dig(dig_gz, text.data(), (size_t)GO[NG]);
    *nlines_out += NU; *plain_out += U[NU]; *gz_out += GO[NG];
    return 0;
}

// gz_build_dynamic on its own: random frequency tables -- a handful of symbols, hundreds, weights from 1 to 10^9 side by side (the length
// limit of 15 bits then bites) -- must give complete codes of at most 15 bits whose block header zlib accepts: the header, every coded
// literal once, the end-of-block code and an empty final block inflate to exactly those literals.
static int fuzz_codes(int rounds) {
    for (int r = 0; r < rounds; r++) {
        uint64_t f[GZ_NSYM] = {0};
        const int style = (int)rr(4);
        for (int i = 0; i < GZ_NSYM; i++) {
            const bool must = i >= 256;                          // end of block, lengths, distances: always codable (gz_codes)
            const bool used = must || rr(style == 0 ? 40 : style == 1 ? 2 : 5) == 0;
            if (!used) continue;
            f[i] = style == 3 ? 1 + (uint64_t)rr(3) * rr(1000000) * (uint64_t)rr(1000) : 1 + (rr(3) ? 0 : (uint64_t)rr(100000));
        }
        f[(int)rr(256)] += 5; f[256] += 1;
        gz_code C;
        if (!gz_build_dynamic(f, C)) { fprintf(stderr, "fuzz %d: header does not fit (%u bits)\n", r, C.hdr_bits); return 1; }
        for (int part = 0; part < 2; part++) {                   // Kraft sum = 1, lengths <= 15
            uint64_t kraft = 0; const int a = part ? GZ_NLIT : 0, b = part ? GZ_NSYM : GZ_NLIT;
            for (int i = a; i < b; i++) { const uint32_t L = C.sym[i] >> 16; if ((f[i] != 0) != (L != 0) || L > 15) { fprintf(stderr, "fuzz %d: symbol %d has frequency %llu and length %u\n", r, i, (unsigned long long)f[i], L); return 1; } if (L) kraft += 1ull << (15 - L); }
            if (kraft != (1ull << 15)) { fprintf(stderr, "fuzz %d: %s code is not complete\n", r, part ? "distance" : "literal/length"); return 1; }
        }
        std::vector<unsigned char> buf(4096, 0); BitBuf bo{buf.data(), 0, 0}; std::string want;
        gz_block_header(bo, &C);
        for (int i = 0; i < 256; i++) if (f[i]) { bo.bits(C.sym[i] & 0xFFFFu, C.sym[i] >> 16); want.push_back((char)i); }
        bo.bits(C.sym[256] & 0xFFFFu, C.sym[256] >> 16);
        bo.bits(3u, 10);                                          // an empty final block with the fixed codes: BFINAL = 1, BTYPE = 01, end of block
        std::string got; size_t used = 0;
        if (inflate_member(buf.data(), (size_t)((bo.at + 7) / 8), got, &used, -MAX_WBITS) || got != want) { fprintf(stderr, "fuzz %d: zlib does not accept the block header (%u bits)\n", r, C.hdr_bits); return 1; }
    }
    return 0;
}

int main(int argc, char **argv) {
    if (fuzz_codes(3000)) { printf("GZ SIM FAILED: code construction\n"); return 1; }
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    for (int dyn = 0; dyn < 2; dyn++) {
        uint64_t nl = 0, pb = 0, gb = 0, hb = 0;
        for (int r = 0; r < rounds; r++) {
            if (run_case(40 + rr(40), 60 + rr(60), 50 + rr(50), 500, 700, (r & 1) != 0, r == 2 ? 1200 : 40, dyn != 0, &nl, &pb, &gb, &hb)) { printf("GZ SIM FAILED in round %d (%s codes)\n", r, dyn ? "dynamic" : "fixed"); return 1; }
        }
        if (run_case(3, 0, 0, 20, 20, false, 5, dyn != 0, &nl, &pb, &gb, &hb) || run_case(0, 2, 0, 20, 20, false, 5, dyn != 0, &nl, &pb, &gb, &hb) || run_case(0, 0, 0, 20, 20, false, 5, dyn != 0, &nl, &pb, &gb, &hb)) { printf("GZ SIM FAILED on a degenerate case (%s codes)\n", dyn ? "dynamic" : "fixed"); return 1; }
        printf("GZ SIM OK (%s codes, block header up to %llu bits): %llu lines, %llu bytes of text as %llu bytes of deflate blocks (%.3f)\n", dyn ? "dynamic" : "fixed", (unsigned long long)hb, (unsigned long long)nl, (unsigned long long)pb, (unsigned long long)gb, pb ? (double)gb / (double)pb : 0.0);
    }
    printf("GZ SIM DIGEST plain %016llx deflate %016llx\n", (unsigned long long)dig_plain, (unsigned long long)dig_gz);
    return 0;
}
