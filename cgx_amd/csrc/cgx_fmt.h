// cgx_fmt.h -- one grammar line as text, and the same line as DEFLATE symbols.
//
// Replaces the reference's fprintf loops (printGapMode / print_query_GPU_Gappy, PrintResults.c:339-577).  Everything here is
// a pure function of read-only tables, marked CGX_HD so the same text compiles for gfx950 (hipcc) and, for the CPU unit test
// under tests/cpu_sim only (gz_sim.cpp: the gzip members are inflated by zlib and compared with the plain text), for the host.
// The product links only the device instantiation (cgx_format.inc).
//
// "%f": value * 10^6 is exact in double for a float input, rounded to nearest-even, printed as integer part '.' six digits --
// bit-identical to glibc printf (tests/cpu_sim/f6test.c checks the same arithmetic on the host).  aa / bb / SampleCountF are
// taken from tables computed by the host libm (they depend on two small integers only).
#ifndef CGX_FMT_H
#define CGX_FMT_H
#include <stdint.h>
#include <math.h>
#include "../../include/cgx.h"
#include "cgx_rules.h"

#define FMT_TABN 302
struct fmt_view {
    const cgx_lexent *lex[3]; const int32_t *rng[3];           // [kind]: 0 contiguous, 1 one gap, 2 two gaps
    const cgx_block *blocks; const cgx_gappat *p1d; const int32_t *c2d; const uint32_t *one2;
    const int32_t *str, *tstr;
    const char *spool, *tpool; const uint32_t *soff, *toff; uint32_t ns, nt;
    const float *aa, *bb, *fs;                                  // host-libm tables [FMT_TABN*FMT_TABN], [FMT_TABN], [FMT_TABN]
    uint32_t G, D1, D2;
};
// Two sinks run the same formatting routine: one counts, one writes.  The writing sink stores whole 8-byte words
// wherever the text allows it (string literals, the ".dddddd" of a %f, word spellings): a line of ~200 characters
// costs ~45 store requests instead of ~200 single-byte ones, and the byte stores were what bounded the kernel.
// `wide(v, n)` stores the 8 bytes of v but advances only n <= 8: the surplus bytes land on positions this same lane
// overwrites next (every such field is followed by at least 8 more characters of the same line), never on another
// line.  Global memory accepts unaligned 8-byte stores on gfx950.
typedef uint64_t __attribute__((aligned(1), may_alias)) u64_unaligned;
struct CountSink {
    uint32_t n;
    CGX_HD void put(char) { n++; }
    CGX_HD void puts(const char *, uint32_t len) { n += len; }
    CGX_HD void wide(uint64_t, uint32_t len) { n += len; }
    CGX_HD void word(const char *, uint32_t len) { n += len; }
};
struct MemSink {
    char *p;
    CGX_HD void put(char c) { *p++ = c; }
    CGX_HD void puts(const char *s, uint32_t len) { for (uint32_t i = 0; i < len; i++) *p++ = s[i]; }
    CGX_HD void wide(uint64_t v, uint32_t len) { *(u64_unaligned *)p = v; p += len; }
    // a spelling from the padded word pool: 8 bytes at a time, the last chunk may carry bytes of the next pool entry
    CGX_HD void word(const char *s, uint32_t len) {
        uint32_t i = 0;
        for (; i + 8 <= len; i += 8) *(u64_unaligned *)(p + i) = *(const u64_unaligned *)(s + i);
        if (i < len) *(u64_unaligned *)(p + i) = *(const u64_unaligned *)(s + i);
        p += len;
    }
};

CGX_HD long long fmt_rn(double v) {                              // round to nearest, ties to even
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2ll_rn(v);
#else
    return llrint(v);
#endif
}
// decimal digits, most significant first, without a per-thread digit buffer (a private array here is turned into
// 252 bytes of LDS per thread by the compiler, which capped the kernel at two blocks per CU)
template <class S> CGX_HD void fmt_uint(S &o, uint64_t v) {
    uint64_t p = 1;
    while (v / p >= 10) p *= 10;
    for (; p; p /= 10) o.put((char)('0' + (int)((v / p) % 10)));
}
// A 32-bit number: the digit count from comparisons, the digits by constant divisions, least significant first, each put at
// its place in one or two 8-byte words that leave as one or two insertions.  This is how a word WITHOUT a spelling is written
// (its id after 's' / 't') -- every word of a corpus that was handed over as ids (cgx_extract_grammars_ids: the bench's corpora),
// so it is no cold path there; fmt_uint spends two 64-bit divisions per digit on it.  (The kernel's time did not move with this:
// 22.4 ms before and after -- DESIGN section 3 has the leave-one-out timings of the formatter.)
template <class S> CGX_HD void fmt_u32(S &o, uint32_t v) {
    const uint32_t n = 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
    uint64_t w0 = 0, w1 = 0;
    CGX_UNROLL
    for (uint32_t k = 0; k < 10; k++) {
        if (k < n) {
            const uint64_t d = (uint64_t)('0' + v % 10u); v /= 10u;
            const uint32_t at = n - 1u - k;                         // place counted from the most significant digit
            if (at < 8u) w0 |= d << (8u * at); else w1 |= d << (8u * (at - 8u));
        }
    }
    o.wide(w0, n < 8u ? n : 8u);
    if (n > 8u) o.wide(w1, n - 8u);
}
template <class S> CGX_HD void fmt_word(S &o, const char *pool, const uint32_t *off, uint32_t nwords, char fallback, int32_t id) {
    if (id >= 0 && (uint32_t)id < nwords) { uint32_t a = off[id], b = off[id + 1]; if (b > a) { o.word(pool + a, b - a); return; } }
    o.put(fallback); if (id < 0) { o.put('-'); id = -id; } fmt_u32(o, (uint32_t)id);
}
template <class S> CGX_HD bool fmt_f6(S &o, float x) {
    double v = (double)x;
    if (!(fabs(v) < 9.0e12)) { o.put('?'); return false; }      // never produced by this path; reported so the host formats instead
    if (signbit(v)) { o.put('-'); v = -v; }
    unsigned long long m = (unsigned long long)fmt_rn(v * 1e6);
    unsigned long long ip = m / 1000000ull; uint32_t fp = (uint32_t)(m % 1000000ull);
    if (ip < 10) o.put((char)('0' + (int)ip)); else fmt_uint(o, ip);
    uint64_t w = '.';                                           // ".dddddd" as one little-endian word (7 characters)
    CGX_UNROLL
    for (int i = 6; i >= 1; i--) { w |= (uint64_t)('0' + fp % 10) << (8 * i); fp /= 10; }
    o.wide(w, 7);
    return true;
}
// the number of characters fmt_f6 writes for x, without the divisions
CGX_HD uint32_t fmt_f6_len(float x) {
    double v = (double)x;
    if (!(fabs(v) < 9.0e12)) return 1u;
    uint32_t n = 8u;
    if (signbit(v)) { n++; v = -v; }
    const unsigned long long m = (unsigned long long)fmt_rn(v * 1e6);
    for (unsigned long long t = 10000000ull; m >= t; t *= 10ull) n++;      // m < 9.1e18: t stays below 1e19
    return n;
}
// a string literal: 8 bytes per store, the tail as one (overlapping) 8-byte store when the literal has >= 8 characters
template <class S, uint32_t N> CGX_HD void fmt_lit(S &o, const char (&s)[N]) {
    constexpr uint32_t n = N - 1;
    if (n < 4) { for (uint32_t i = 0; i < n; i++) o.put(s[i]); return; }     // " ||| " is always followed by three more characters of its line
    uint32_t i = 0;
    CGX_UNROLL
    for (; i + 8 <= n; i += 8) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)(uint8_t)s[i + k] << (8 * k); o.wide(v, 8); }
    if (i < n) { uint64_t v = 0; for (uint32_t k = 0; k < n - i; k++) v |= (uint64_t)(uint8_t)s[i + k] << (8 * k); o.wide(v, n - i); }
}
#define FMT_LIT(o, lit) fmt_lit(o, lit)
template <class S> CGX_HD void fmt_gap(S &o, char d) {      // "[X,d]": never the end of a line
    o.wide((uint64_t)'[' | ((uint64_t)'X' << 8) | ((uint64_t)',' << 16) | ((uint64_t)(uint8_t)d << 24) | ((uint64_t)']' << 32), 5);
}
template <class S> CGX_HD void fmt_block(S &o, const fmt_view &F, uint32_t bn) {
    cgx_block k = F.blocks[bn];
    for (int i = 0; i < k.matchlen; i++) { if (i) o.put(' '); fmt_word(o, F.spool, F.soff, F.ns, 's', F.str[k.string_start + i]); }
}
template <class S> CGX_HD void fmt_pattern(S &o, const fmt_view &F, uint32_t one, char gapdigit, bool lead_space) {
    const cgx_gappat *p = &F.p1d[one];                          // read in place: a private copy indexed by j would be spilled to LDS (promoted alloca)
    const int n = p->number;
    for (int j = 0; j < n; j++) {
        const int32_t sym = p->pat[j];
        if (j || lead_space) o.put(' ');
        if (sym >= 0) fmt_word(o, F.spool, F.soff, F.ns, 's', sym); else fmt_gap(o, gapdigit);
    }
}
// source side from the converted id (ExtractPair.c:743-796, 1021-1123)
template <class S> CGX_HD void fmt_source(S &o, const fmt_view &F, int kind, uint32_t cid) {
    const uint32_t G = F.G, D1 = F.D1, D2 = F.D2;
    if (kind == 0) { fmt_block(o, F, cid); return; }
    if (kind == 1) {
        if (cid < G) { fmt_gap(o, '1'); o.put(' '); fmt_block(o, F, cid); }
        else if (cid < 2 * G) { fmt_block(o, F, cid - G); o.put(' '); fmt_gap(o, '1'); }
        else fmt_pattern(o, F, cid - 2 * G, '1', false);
        return;
    }
    if (cid < G) { fmt_gap(o, '1'); o.put(' '); fmt_block(o, F, cid); o.put(' '); fmt_gap(o, '2'); }
    else if (cid < G + D2) { fmt_pattern(o, F, F.one2[cid - G], '1', false); o.put(' '); fmt_gap(o, '2'); o.put(' '); fmt_word(o, F.spool, F.soff, F.ns, 's', F.c2d[cid - G]); }
    else if (cid < G + D2 + D1) { fmt_gap(o, '1'); fmt_pattern(o, F, cid - G - D2, '2', true); }
    else { fmt_pattern(o, F, cid - G - D2 - D1, '1', false); o.put(' '); fmt_gap(o, '2'); }
}
// target side of a lexicon line (ExtractPair.c:813-837, 1141-1163)
template <class S> CGX_HD void fmt_target(S &o, const fmt_view &F, int kind, const cgx_lexent &e) {
    uint32_t t0 = e.tstart, t1 = t0 + e.end, ga = t0 + e.gap1, gb = t0 + e.gap1_1, gc = t0 + e.gap2, gd = t0 + e.gap2_1; bool first = true;
    for (uint32_t jj = t0; jj <= t1; jj++) {
        if (!first) o.put(' ');
        first = false;
        if (kind >= 1 && jj >= ga && jj <= gb) { fmt_gap(o, '1'); jj = gb; }
        else if (kind >= 2 && jj >= gc && jj <= gd) { fmt_gap(o, '2'); jj = gd; }
        else fmt_word(o, F.tpool, F.toff, F.nt, 't', F.tstr[jj]);
    }
}
CGX_HD void fmt_note(CountSink &o, uint32_t *at) { if (at) *at = o.n; }
template <class S> CGX_HD void fmt_note(S &, uint32_t *) {}
// tail_at (counting sink only): where the features start, " ||| EgivenFCoherent=" -- the gzip members refer to the previous line's tail by it
template <class S> CGX_HD bool fmt_line(S &o, const fmt_view &F, int kind, const cgx_lexent &e, uint32_t *tail_at = nullptr) {
    bool ok = true;
    FMT_LIT(o, "[X] ||| "); fmt_source(o, F, kind, (uint32_t)e.id); FMT_LIT(o, " ||| ");
    fmt_target(o, F, kind, e);
    fmt_note(o, tail_at);
    if (e.paircount >= FMT_TABN || e.fsample >= FMT_TABN || e.fsample == 0) ok = false;
    uint32_t pc = e.paircount < FMT_TABN ? e.paircount : 0, f = e.fsample < FMT_TABN ? e.fsample : 0;
    FMT_LIT(o, " ||| EgivenFCoherent="); ok &= fmt_f6(o, F.aa[pc * FMT_TABN + f]);
    FMT_LIT(o, " SampleCountF="); ok &= fmt_f6(o, F.fs[f]);
    FMT_LIT(o, " CountEF="); ok &= fmt_f6(o, F.bb[pc]);
    FMT_LIT(o, " MaxLexFgivenE="); ok &= fmt_f6(o, e.fe);
    FMT_LIT(o, " MaxLexEgivenF="); ok &= fmt_f6(o, e.ef);
    FMT_LIT(o, " IsSingletonF="); o.put(e.f == 1 ? '1' : '0');
    FMT_LIT(o, " IsSingletonFE="); o.put(e.paircount == 1 ? '1' : '0'); o.put('\n');
    return ok;
}

// ---- emission groups ----
// Every (kind, converted id) belongs to exactly one CANONICAL item: the four items a block always emits together (4p .. 4p+3),
// the three of a one-gap pattern (4G + 3id ..), the one of a two-gap pattern (4G + 3D1 + id), PrintResults.c:451-570.  The unique
// text is laid out in canonical order, so the items a file emits back to back are neighbours in it and leave as one piece.
CGX_HD void canon_item(uint32_t c, uint32_t G, uint32_t D1, uint32_t D2, int *kind, uint32_t *cid) {
    if (c < 4 * G) { const uint32_t p = c >> 2; const int sub = (int)(c & 3); *kind = sub <= 1 ? 1 : sub == 2 ? 2 : 0; *cid = sub == 0 ? p + G : p; return; }
    c -= 4 * G;
    if (c < 3 * D1) { const uint32_t id = c / 3; const int sub = (int)(c % 3); *kind = sub == 0 ? 1 : 2; *cid = sub == 0 ? 2 * G + id : sub == 1 ? G + D2 + id : G + D2 + D1 + id; return; }
    *kind = 2; *cid = G + (c - 3 * D1);
}
// group of canonical item c, and the first canonical item of group g (g = G + D1 + D2 gives the number of items)
CGX_HD uint32_t canon_group(uint32_t c, uint32_t G, uint32_t D1) { if (c < 4 * G) return c >> 2; c -= 4 * G; return c < 3 * D1 ? G + c / 3 : G + D1 + (c - 3 * D1); }
CGX_HD uint32_t canon_group_first(uint32_t g, uint32_t G, uint32_t D1) { if (g < G) return 4 * g; g -= G; return g < D1 ? 4 * G + 3 * g : 4 * G + 3 * D1 + (g - D1); }

// =====================================================================================================================
// The same line as DEFLATE symbols: grammar.<q>.s.gz without a compressor.
//
// SURVEY 8(f3) asks for optional gz output because at eight GPUs file I/O bounds the path; the reference writes one fprintf
// per rule (PrintResults.c:407-577).  A general compressor searches for repetitions; the formatter KNOWS them: a line repeats
// its predecessor's "[X] ||| source ||| ", every feature name, and most feature values.  So the formatter emits the symbols
// itself -- back-references (length, distance) where it knows the bytes stood one line earlier, literals elsewhere, the fixed
// Huffman code of RFC 1951 3.2.6 -- and no hash chains, no window, no second look at the text are needed.
//
//  * One DEFLATE BLOCK per emission group (the four items of a contiguous phrase, the three of a one-gap pattern, the one of a
//    two-gap pattern), followed by an empty stored block (zlib's "sync flush": its header pads to a byte boundary): a group's
//    bytes are a self-contained, byte-aligned stretch of a deflate stream -- a back-reference never leaves its group -- so a
//    grammar file is still a concatenation of pieces of the batch's unique text, and phrases shared by thousands of queries are
//    formatted, compressed and sent once.  The file is ONE gzip member (RFC 1952): ten header bytes, the groups, an empty final
//    block (03 00), CRC-32 and ISIZE -- header and trailer are put around the pieces by the host writer, the trailer's values
//    come from the device (k_gz_files).
//    (Round 4's first version made every group a gzip member of its own and every line a block of its own, ending on a byte:
//    18 bytes per group and 5-6 per line, a quarter of the output, were framing.)
//  * Inside a group the lines follow each other at BIT granularity: count pass (bits per line) -> scan -> bit offset of every
//    line; the writing sink of k_fmt_lines_gz shares a line's first and last 32-bit word with its neighbours.
//  * CRC-32: every line computes the CRC of its own text (group-wise, tables in LDS) and multiplies it by x^(8 * bytes that
//    follow it in the group) mod P (zlib's crc32_combine as a product of table entries); a group's CRC is the XOR of its lines'
//    contributions (k_gz_groups) and a file's the fold of its groups' (k_gz_files).
// What a line may refer to: its predecessor in the same group (prefix through the source side when both lines belong to the
// same item, "[X] ||| " otherwise; the tail from " ||| EgivenFCoherent=" on, name by name, a value only when it equals the
// predecessor's), and itself (" ||| " and, in a group's first line, " IsSingletonF" / " MaxLex").  Target words are literals.
// =====================================================================================================================
CGX_HD uint32_t gz_brev(uint32_t x, uint32_t nbits) {            // the low nbits of x, reversed (Huffman codes are packed starting with their most significant bit)
#if defined(__clang__)
    return __builtin_bitreverse32(x) >> (32u - nbits);
#else
    uint32_t r = 0; for (uint32_t i = 0; i < nbits; i++) r |= ((x >> i) & 1u) << (nbits - 1u - i); return r;
#endif
}
CGX_HD uint32_t gz_brev_bytes(uint32_t x) {                      // every byte of x reversed in place
#if defined(__clang__)
    return __builtin_bswap32(__builtin_bitreverse32(x));
#else
    x = ((x & 0xF0F0F0F0u) >> 4) | ((x & 0x0F0F0F0Fu) << 4); x = ((x & 0xCCCCCCCCu) >> 2) | ((x & 0x33333333u) << 2); return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
#endif
}
CGX_HD uint32_t gz_log2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }   // x > 0

// bit sinks: the counting pass only adds; the writing sinks (BitWordSink in cgx_format.inc, a plain bit buffer in tests/cpu_sim)
// pack LSB-first from the line's bit offset on
struct BitCount {
    uint32_t n;
    CGX_HD void bits(uint32_t, uint32_t len) { n += len; }
};

#define GZ_POLY 0xEDB88320u
// a(x) * b(x) mod P in the reflected representation (x^0 = 0x80000000), as zlib's multmodp
CGX_HD uint32_t gz_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ GZ_POLY : b >> 1;
    }
    return p;
}
// Tables, 12 x 256 words.  [k][x], k = 0..7: the CRC register after byte x and k zero bytes, started from 0 ("slicing": a group of
// n <= 8 bytes is folded in with n independent look-ups instead of n dependent ones; the kernel keeps these 8 KB in LDS).
// [8 + j][d] = x^(8 * d * 256^j) mod P, j = 0..3: the factors of x^(8 * n).
#define GZ_CRC_WORDS (8 * 256)
#define GZ_TAB_WORDS (12 * 256)
CGX_HD uint32_t gz_x8n(const uint32_t *tab, uint64_t nbytes) {   // x^(8 * nbytes) mod P, nbytes < 2^32
    uint32_t r = tab[GZ_CRC_WORDS + (nbytes & 255u)];
    CGX_UNROLL
    for (int j = 1; j < 4; j++) { const uint32_t d = (uint32_t)(nbytes >> (8 * j)) & 255u; if (d) r = gz_multmodp(tab[GZ_CRC_WORDS + 256 * j + d], r); }
    return r;
}
// the CRC register after the low n (1..8) bytes of v
CGX_HD uint32_t gz_crc_group(const uint32_t *tab, uint32_t crc, uint64_t v, uint32_t n) {
    const uint64_t w = v ^ (uint64_t)crc;
    uint32_t r = n < 4u ? crc >> (8u * n) : 0u;
    CGX_UNROLL
    for (uint32_t i = 0; i < 8u; i++) if (i < n) r ^= tab[((n - 1u - i) << 8) + ((uint32_t)(w >> (8u * i)) & 255u)];
    return r;
}
inline void gz_make_tables(uint32_t *tab) {                      // host only
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ GZ_POLY : c >> 1; tab[i] = c; }
    for (int k = 1; k < 8; k++) for (uint32_t i = 0; i < 256; i++) { const uint32_t c = tab[256 * (k - 1) + i]; tab[256 * k + i] = (c >> 8) ^ tab[c & 255u]; }
    uint32_t x8 = 0x80000000u; for (int k = 0; k < 8; k++) x8 = (x8 & 1u) ? (x8 >> 1) ^ GZ_POLY : x8 >> 1;    // x^8
    uint32_t step = x8;
    for (int j = 0; j < 4; j++) {
        uint32_t *t = tab + GZ_CRC_WORDS + 256 * j; t[0] = 0x80000000u;
        for (uint32_t d = 1; d < 256; d++) t[d] = gz_multmodp(t[d - 1], step);
        step = gz_multmodp(t[255], step);                        // x^(8 * 256^(j+1))
    }
}

// ---- the Huffman codes of a batch -------------------------------------------------------------------------------------
// Every group is ONE block, and every block of a batch uses the same codes: either the fixed codes of RFC 1951 3.2.6 or -- the
// default -- dynamic codes made for the batch from a tally of its own symbols (a sixteenth of its lines: k_fmt_lines_gz<GZ_TALLY>),
// whose description (3.2.7) every group carries in front.  The text is digits, a few punctuation marks and the words' letters:
// codes of 3-5 bits for the digits instead of 8, two-bit distance codes for the previous line instead of 5, at the price of a
// header of 50-60 bytes per group.
// Table entry: low 16 bits = the code, bit-reversed (DEFLATE packs Huffman codes from their most significant bit), bits 16.. = its
// length (0: the symbol has no code -- a literal byte no word of the vocabulary contains; a line that needs it is not representable).
#define GZ_NLIT 286
#define GZ_NDIST 30
#define GZ_NSYM (GZ_NLIT + GZ_NDIST)
#define GZ_HDR_WORDS 64
struct gz_code { uint32_t sym[GZ_NSYM]; uint32_t hdr_bits; uint32_t hdr[GZ_HDR_WORDS]; };   // hdr: BFINAL = 0, BTYPE and, for dynamic codes, the code lengths: the first bits of every group
#if defined(__HIP_DEVICE_COMPILE__)
#define GZ_TALLY_ADD(p) atomicAdd((p), 1u)
#else
#define GZ_TALLY_ADD(p) ((void)(*(p))++)
#endif
enum { GZ_COUNT = 0, GZ_WRITE = 1, GZ_TALLY = 2 };

// The deflate sink: the interface of the byte sinks above, plus match() / lit() to say what the bytes that follow are.
// MODE GZ_TALLY: symbols are counted in hist[GZ_NSYM] instead of being emitted.
template <class B, bool CRC, int MODE = GZ_WRITE> struct GzSink {
    B &b; const uint32_t *tab; const uint32_t *code; unsigned int *hist;
    uint32_t pos, mlen, mdist, crc; bool inm, bad;
    CGX_HD GzSink(B &bits, const uint32_t *crc_table, const uint32_t *codes, unsigned int *tally = nullptr) : b(bits), tab(crc_table), code(codes), hist(tally), pos(0), mlen(0), mdist(0), crc(0xFFFFFFFFu), inm(false), bad(false) {}
    CGX_HD void sym(uint32_t s) {
        if (MODE == GZ_TALLY) { GZ_TALLY_ADD(&hist[s]); return; }
        const uint32_t e = code[s];
        if ((e >> 16) == 0) bad = true;
        b.bits(e & 0xFFFFu, e >> 16);
    }
    CGX_HD void emit_match(uint32_t len, uint32_t dist) {        // 3 <= len <= 257, 1 <= dist <= 32768
        const uint32_t l = len - 3u; uint32_t s, eb = 0;
        if (l < 8u) s = 257u + l; else { eb = gz_log2(l) - 2u; s = 261u + 4u * eb + ((l >> eb) & 3u); }
        sym(s);
        if (eb && MODE != GZ_TALLY) b.bits(l & ((1u << eb) - 1u), eb);
        const uint32_t dd = dist - 1u; uint32_t dc, deb = 0;
        if (dd < 4u) dc = dd; else { const uint32_t k = gz_log2(dd); deb = k - 1u; dc = 2u * k + ((dd >> deb) & 1u); }
        sym(GZ_NLIT + dc);
        if (deb && MODE != GZ_TALLY) b.bits(dd & ((1u << deb) - 1u), deb);
    }
    CGX_HD void flush() { if (mlen) { if (mlen < 3u) bad = true; else emit_match(mlen, mdist); mlen = 0; } }
    // the bytes that follow stood `dist` bytes earlier in this group / are new
    CGX_HD void match(uint32_t dist) { if (inm && dist == mdist) return; flush(); inm = dist >= 1u && dist <= 32768u; mdist = dist; }
    CGX_HD void lit() { flush(); inm = false; }
    CGX_HD void feed(uint64_t v, uint32_t n) {                   // the low n (1..8) bytes of v
        pos += n;
#ifndef GZ_LOO
#define GZ_LOO 0      // leave-one-out builds for timing studies (output meaningless; the counting and the writing pass stay consistent): 1 no CRC-32, 2 literals as raw bytes (no code table), 3 both
#endif
        if (CRC && !(GZ_LOO & 1)) crc = gz_crc_group(tab, crc, v, n);
        if (inm) { mlen += n; if (mlen >= 258u) { emit_match(255u, mdist); mlen -= 255u; } }      // what stays pending is again >= 3
        else if (MODE == GZ_TALLY) { for (uint32_t i = 0; i < n; i++) GZ_TALLY_ADD(&hist[(uint32_t)(v >> (8u * i)) & 255u]); }
        else if (GZ_LOO & 2) { b.bits((uint32_t)v & (n >= 4u ? 0xFFFFFFFFu : (1u << (8u * n)) - 1u), n >= 4u ? 32u : 8u * n); if (n > 4u) b.bits((uint32_t)(v >> 32) & (n >= 8u ? 0xFFFFFFFFu : (1u << (8u * (n - 4u))) - 1u), 8u * (n - 4u)); }
        else {
            // literals two at a time: two codes are at most 30 bits, one insertion into the bit sink instead of two
            uint32_t i = 0;
            for (; i + 2u <= n; i += 2u) {
                const uint32_t e0 = code[(uint32_t)(v >> (8u * i)) & 255u], e1 = code[(uint32_t)(v >> (8u * i + 8u)) & 255u];
                const uint32_t l0 = e0 >> 16, l1 = e1 >> 16;
                if (l0 == 0 || l1 == 0) bad = true;
                b.bits((e0 & 0xFFFFu) | ((e1 & 0xFFFFu) << l0), l0 + l1);
            }
            if (i < n) sym((uint32_t)(v >> (8u * i)) & 255u);
        }
    }
    CGX_HD void put(char c) { feed((uint64_t)(uint8_t)c, 1); }
    CGX_HD void puts(const char *s, uint32_t len) { for (uint32_t i = 0; i < len; i++) put(s[i]); }
    CGX_HD void wide(uint64_t v, uint32_t len) { feed(v, len); }
    CGX_HD void word(const char *s, uint32_t len) {
        for (uint32_t i = 0; i < len; i += 8) { const uint32_t n = len - i < 8 ? len - i : 8; feed(*(const u64_unaligned *)(s + i), n); }
    }
};

// what a line knows about its place in its member
struct gz_place {
    bool first, last;        // first / last line of its emission group
    bool same_item;          // the line before it belongs to the same canonical item: same source side
    uint32_t prev_len, prev_tail;   // characters of the line before it, and where its features start (first == false)
    const cgx_lexent *pe;    // that line's lexicon entry: its feature values are fetched where they are compared, not held
};
CGX_HD bool gz_same_bits(float a, float b) { union { float f; uint32_t u; } x, y; x.f = a; y.f = b; return x.u == y.u; }
// one feature value: in the running match when it is the previous line's value, literal otherwise; the distance moves by the difference of the two lengths
template <class Z> CGX_HD bool gz_value(Z &z, float v, float pv, uint32_t &dist) {
    const uint32_t p0 = z.pos;
    if (!gz_same_bits(v, pv)) z.lit();
    const bool ok = fmt_f6(z, v);
    dist += (z.pos - p0) - fmt_f6_len(pv);
    return ok;
}
// The bits that follow a group's last line: the empty stored block that ends the group on a byte (BTYPE = 00 after at most seven
// bits of padding, LEN = 0, NLEN = ~0).  bytes = whole bytes a group of `raw` symbol bits (block header .. end of block) takes.
#define GZ_STORED_BITS 3u
CGX_HD uint64_t gz_group_bytes(uint64_t raw_bits) { return (raw_bits + GZ_STORED_BITS + 7u) / 8u + 4u; }
// What the host puts around a file's pieces (RFC 1952 2.3): ID1 ID2 CM FLG MTIME(4) XFL OS; and behind them an empty final block
// (BFINAL = 1, BTYPE = 01, end of block: 03 00), CRC-32 and ISIZE of the file's text.
#define GZ_FILE_HEADER_BYTES 10
#define GZ_FILE_TRAILER_BYTES 10
// the block header: BFINAL = 0, BTYPE, and the description of the batch's codes when they are dynamic
template <class B> CGX_HD void gz_block_header(B &b, const gz_code *C) {
    const uint32_t hb = C->hdr_bits;
    for (uint32_t k = 0; 32u * k < hb; k++) b.bits(hb - 32u * k >= 32u ? C->hdr[k] : C->hdr[k] & ((1u << (hb - 32u * k)) - 1u), hb - 32u * k >= 32u ? 32u : hb - 32u * k);
}
// One line of a group's block.  Bits: [block header, first line only] symbols [end of block, last line only] -- the caller
// appends the stored block (gz_stored) behind a last line, where the bit position is known.
// Returns false when the line cannot be represented (the batch is then formatted on the host).
template <class B, bool CRC, int MODE> CGX_HD bool fmt_line_gz(B &b, GzSink<B, CRC, MODE> &z, const fmt_view &F, const gz_code *C, int kind, const cgx_lexent &e, const gz_place &P) {
    bool ok = true;
    if (P.first && MODE != GZ_TALLY) gz_block_header(b, C);
    const bool prev = !P.first;
    if (prev && P.same_item) { z.match(P.prev_len); FMT_LIT(z, "[X] ||| "); fmt_source(z, F, kind, (uint32_t)e.id); FMT_LIT(z, " ||| "); }
    else {
        if (prev) z.match(P.prev_len); else z.lit();
        FMT_LIT(z, "[X] ||| "); z.lit(); fmt_source(z, F, kind, (uint32_t)e.id);
        z.match(z.pos - 3u); FMT_LIT(z, " ||| ");                // the " ||| " of this line's own "[X] ||| "
    }
    z.lit(); fmt_target(z, F, kind, e);
    if (e.paircount >= FMT_TABN || e.fsample >= FMT_TABN || e.fsample == 0) ok = false;
    const uint32_t pc = e.paircount < FMT_TABN ? e.paircount : 0, f = e.fsample < FMT_TABN ? e.fsample : 0;
    const float v0 = F.aa[pc * FMT_TABN + f], v1 = F.fs[f], v2 = F.bb[pc];
    const bool f1 = e.f == 1, f2 = e.paircount == 1;
    if (prev) {
        // the distance to the same place of the previous line: what is left of that line plus what this line has so far
        uint32_t dist = z.pos + P.prev_len - P.prev_tail;
        const uint32_t ppc = P.pe->paircount < FMT_TABN ? P.pe->paircount : 0, pf = P.pe->fsample < FMT_TABN ? P.pe->fsample : 0;
        z.match(dist); FMT_LIT(z, " ||| EgivenFCoherent="); ok &= gz_value(z, v0, F.aa[ppc * FMT_TABN + pf], dist);
        z.match(dist); FMT_LIT(z, " SampleCountF="); ok &= gz_value(z, v1, F.fs[pf], dist);
        z.match(dist); FMT_LIT(z, " CountEF="); ok &= gz_value(z, v2, F.bb[ppc], dist);
        z.match(dist); FMT_LIT(z, " MaxLexFgivenE="); ok &= gz_value(z, e.fe, P.pe->fe, dist);
        z.match(dist); FMT_LIT(z, " MaxLexEgivenF="); ok &= gz_value(z, e.ef, P.pe->ef, dist);
        z.match(dist); FMT_LIT(z, " IsSingletonF=");
        if (f1 != (P.pe->f == 1)) z.lit(); z.put(f1 ? '1' : '0');
        z.match(dist); FMT_LIT(z, " IsSingletonFE=");
        if (f2 != (P.pe->paircount == 1)) z.lit(); z.put(f2 ? '1' : '0'); z.put('\n');    // the line feed alone would be a match of one character: it goes the way of the flag
    } else {
        z.match(z.pos - 3u); FMT_LIT(z, " ||| "); z.lit(); FMT_LIT(z, "EgivenFCoherent="); ok &= fmt_f6(z, v0);
        FMT_LIT(z, " SampleCountF="); ok &= fmt_f6(z, v1);
        FMT_LIT(z, " CountEF="); ok &= fmt_f6(z, v2);
        FMT_LIT(z, " MaxLexFgivenE="); const uint32_t at = z.pos; ok &= fmt_f6(z, e.fe);
        z.match(z.pos - at + 15u); FMT_LIT(z, " MaxLex"); z.lit(); FMT_LIT(z, "EgivenF="); ok &= fmt_f6(z, e.ef);
        FMT_LIT(z, " IsSingletonF="); z.put(f1 ? '1' : '0');
        z.match(15u); FMT_LIT(z, " IsSingletonF"); z.lit(); z.put('E'); z.put('='); z.put(f2 ? '1' : '0'); z.put('\n');
    }
    z.lit();
    if (z.bad) ok = false;
    if (P.last) z.sym(256u);                                     // end of block
    if (z.bad) ok = false;
    return ok;
}
// ---- host: the code tables of a batch (host functions; the device reads the finished gz_code) ----
#include <vector>
#include <algorithm>
// Huffman code lengths of at most `limit` bits for the symbols with f > 0 (at least two of them): the plain construction, and
// while it comes out too deep the frequencies are halved (floor 1), which flattens the tree.
inline void gz_huff_lengths(const uint64_t *f, int n, int limit, uint8_t *len) {
    std::vector<uint64_t> w(f, f + n);
    for (;;) {
        struct node { uint64_t w; int l, r; };
        std::vector<node> t; std::vector<int> act;
        for (int i = 0; i < n; i++) { len[i] = 0; if (w[i]) { t.push_back({w[i], -1 - i, 0}); act.push_back((int)t.size() - 1); } }
        if (act.size() == 1) { len[-1 - t[act[0]].l] = 1; return; }
        auto cmp = [&](int a, int b) { return t[a].w != t[b].w ? t[a].w > t[b].w : a > b; };   // min-heap on weight (stable on creation order)
        std::make_heap(act.begin(), act.end(), cmp);
        while (act.size() > 1) {
            std::pop_heap(act.begin(), act.end(), cmp); const int a = act.back(); act.pop_back();
            std::pop_heap(act.begin(), act.end(), cmp); const int b = act.back(); act.pop_back();
            t.push_back({t[a].w + t[b].w, a, b}); act.push_back((int)t.size() - 1); std::push_heap(act.begin(), act.end(), cmp);
        }
        int deepest = 0;
        std::vector<std::pair<int, int>> st; st.push_back({act[0], 0});
        while (!st.empty()) {
            const std::pair<int, int> x = st.back(); st.pop_back();
            if (t[x.first].l < 0) { len[-1 - t[x.first].l] = (uint8_t)x.second; if (x.second > deepest) deepest = x.second; }
            else { st.push_back({t[x.first].l, x.second + 1}); st.push_back({t[x.first].r, x.second + 1}); }
        }
        if (deepest <= limit) return;
        for (int i = 0; i < n; i++) if (w[i]) w[i] = (w[i] + 1) / 2;
    }
}
// Lengths for an alphabet most of whose codable symbols were NOT seen in the tally (f == 1: the floor gz_codes gives every symbol the
// text may hold -- match lengths and distances that are possible but did not occur in the sample, letters of rare words).  Left to the
// plain construction they end up on the two or three deepest levels in no order, and the block header, which every group carries,
// spends three to four bits on each.  Here the seen symbols and ONE stand-in for all the others get Huffman lengths; the stand-in is
// then replaced by a complete subtree over the others, the shorter of its two depths on the lower symbol numbers: the unseen symbols
// have at most two lengths, in long runs, which the header's run-length code takes six at a time.
inline void gz_huff_lengths_rare(const uint64_t *f, int n, uint8_t *len) {
    std::vector<int> rare; std::vector<uint64_t> w(f, f + n);
    for (int i = 0; i < n; i++) if (f[i] == 1) rare.push_back(i);
    int seen = 0; for (int i = 0; i < n; i++) seen += f[i] > 1;
    if (rare.size() < 4 || seen < 1) { gz_huff_lengths(f, n, 15, len); return; }
    int k = 0; while ((1u << k) < rare.size()) k++;             // depth of the subtree: its leaves sit k - 1 and k levels below the stand-in
    for (int limit = 15 - k; ; ) {
        std::vector<uint64_t> g(w); for (int i : rare) g[i] = 0;
        g[rare[0]] = rare.size();                                // the stand-in: as heavy as the symbols it stands for
        gz_huff_lengths(g.data(), n, limit, len);
        const int base = len[rare[0]];
        const size_t shallow = ((size_t)1 << k) - rare.size();   // leaves at depth k - 1 (a complete tree: 2 * shallow + (m - shallow) = 2^k)
        for (size_t j = 0; j < rare.size(); j++) len[rare[j]] = (uint8_t)(base + (j < shallow ? k - 1 : k));
        return;
    }
}
// canonical codes (RFC 1951 3.2.2) for lengths len[0..n), bit-reversed for LSB-first packing: out[i] = code | len << 16
inline void gz_canonical(const uint8_t *len, int n, uint32_t *out) {
    uint32_t cnt[17] = {0}, next[17] = {0};
    for (int i = 0; i < n; i++) cnt[len[i]]++;
    cnt[0] = 0;
    uint32_t code = 0; for (int b = 1; b <= 16; b++) { code = (code + cnt[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < n; i++) out[i] = len[i] ? (gz_brev(next[len[i]]++, len[i]) | ((uint32_t)len[i] << 16)) : 0u;
}
struct gz_bitstr { uint32_t *w; uint32_t cap, n; bool over; void put(uint32_t v, uint32_t k) { for (uint32_t i = 0; i < k; i++, n++) { if ((n >> 5) >= cap) { over = true; return; } if ((v >> i) & 1u) w[n >> 5] |= 1u << (n & 31); } } };
// the fixed codes of RFC 1951 3.2.6
inline void gz_build_fixed(gz_code &C) {
    uint8_t ll[GZ_NLIT + 2], dl[GZ_NDIST];
    for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    for (int i = 0; i < GZ_NDIST; i++) dl[i] = 5;
    uint32_t tmp[288]; gz_canonical(ll, 288, tmp);
    for (int i = 0; i < GZ_NLIT; i++) C.sym[i] = tmp[i];
    gz_canonical(dl, GZ_NDIST, C.sym + GZ_NLIT);                  // (32 five-bit codes of which 30 are used: the canonical numbering of 30 equal lengths is 0..29 all the same)
    for (int k = 0; k < GZ_HDR_WORDS; k++) C.hdr[k] = 0;
    C.hdr[0] = 2u; C.hdr_bits = 3;                               // BFINAL = 0, BTYPE = 01
}
// Dynamic codes from symbol frequencies (every symbol the text can contain must have f > 0), and the block header that describes
// them (RFC 1951 3.2.7).  false: the header does not fit GZ_HDR_WORDS (never with 316 symbols) -- the caller falls back to the fixed codes.
inline bool gz_build_dynamic(const uint64_t *freq, gz_code &C) {
    uint8_t ll[GZ_NLIT], dl[GZ_NDIST];
    gz_huff_lengths_rare(freq, GZ_NLIT, ll); gz_huff_lengths_rare(freq + GZ_NLIT, GZ_NDIST, dl);
    gz_canonical(ll, GZ_NLIT, C.sym); gz_canonical(dl, GZ_NDIST, C.sym + GZ_NLIT);
    int nl = GZ_NLIT; while (nl > 257 && ll[nl - 1] == 0) nl--;
    int nd = GZ_NDIST; while (nd > 1 && dl[nd - 1] == 0) nd--;
    // the two length sequences as one, run-length coded with the code-length alphabet: (symbol, extra value)
    std::vector<uint8_t> seq(ll, ll + nl); seq.insert(seq.end(), dl, dl + nd);
    std::vector<std::pair<int, int>> rl;
    for (size_t i = 0; i < seq.size();) {
        size_t j = i; while (j < seq.size() && seq[j] == seq[i]) j++;
        size_t run = j - i;
        if (seq[i] == 0) {
            while (run >= 11) { const size_t k = run > 138 ? 138 : run; rl.push_back({18, (int)k - 11}); run -= k; }
            if (run >= 3) { rl.push_back({17, (int)run - 3}); run = 0; }
            while (run--) rl.push_back({0, 0});
        } else {
            rl.push_back({seq[i], 0}); run--;
            while (run >= 3) { const size_t k = run > 6 ? 6 : run; rl.push_back({16, (int)k - 3}); run -= k; }
            while (run--) rl.push_back({seq[i], 0});
        }
        i = j;
    }
    uint64_t cf[19] = {0}; for (auto &x : rl) cf[x.first]++;
    uint8_t cl[19]; gz_huff_lengths(cf, 19, 7, cl);
    { int used = 0; for (int i = 0; i < 19; i++) used += cl[i] != 0; if (used < 2) { for (int i = 0; i < 19 && used < 2; i++) if (!cl[i]) { cl[i] = 1; used++; } } }   // a complete code needs two symbols
    uint32_t cc[19]; gz_canonical(cl, 19, cc);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int nc = 19; while (nc > 4 && cl[order[nc - 1]] == 0) nc--;
    for (int k = 0; k < GZ_HDR_WORDS; k++) C.hdr[k] = 0;
    gz_bitstr H{C.hdr, GZ_HDR_WORDS, 0, false};
    H.put(0, 1); H.put(2, 2);                                    // BFINAL = 0, BTYPE = 10
    H.put((uint32_t)(nl - 257), 5); H.put((uint32_t)(nd - 1), 5); H.put((uint32_t)(nc - 4), 4);
    for (int k = 0; k < nc; k++) H.put(cl[order[k]], 3);
    for (auto &x : rl) { H.put(cc[x.first] & 0xFFFFu, cc[x.first] >> 16); if (x.first == 16) H.put((uint32_t)x.second, 2); else if (x.first == 17) H.put((uint32_t)x.second, 3); else if (x.first == 18) H.put((uint32_t)x.second, 7); }
    C.hdr_bits = H.n;
    return !H.over;
}

// behind a group's last line, in a sink that knows its bit position (align): the empty stored block
template <class B> CGX_HD void gz_stored(B &b) { b.bits(0, GZ_STORED_BITS); b.align(); b.bits(0, 16); b.bits(0xFFFFu, 16); }
#endif
