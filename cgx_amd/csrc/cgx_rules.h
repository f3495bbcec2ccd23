// cgx_rules.h -- per-occurrence alignment logic of the extraction and gap-validation kernels.
//
// Everything here is a pure function of the read-only index arrays, marked CGX_HD so the
// same text compiles for gfx950 (hipcc) and, for the CPU sanitizer/unit-test build under
// tests/cpu_sim only, for the host.  The product links only the device instantiation.
//
// Reference semantics followed (file:line in /root/reference):
//   gap validity            checkBoundaryGap          GappyLook.cu:43-126
//   tight back-projection   consistent                ExtractPair.cu:103-133
//   span helpers            checkBoundaryFast/Fast2   ExtractPair.cu:135-250
//   span with error codes   checkBoundary             ExtractPair.cu:252-342
//   ab / Xab / abX / XabX   extractConsistentPairs_Gappy   ExtractPair.cu:1055-1795
//   aXbXc                   extractConsistentPairs_TwoGap  ExtractPair.cu:891-1053
//   aXb / XaXb / aXbX       extractConsistentPairs_OneGap  ExtractPair.cu:351-889
//   sampling                ExtractPair.cu:1143-1160, 454-471, 955-972
//   MaxLex features         lexicalTaskMaxEF          ExtractPair.cu:2144-2432
#ifndef CGX_RULES_H
#define CGX_RULES_H
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define CGX_HD __host__ __device__ __forceinline__
#define CGX_UNROLL _Pragma("unroll")
#else
#define CGX_HD inline
#define CGX_UNROLL
#endif

#define CGX_MAX_SPAN 15
#define CGX_MAX_SYMBOLS 5
#define CGX_TOP 100
#define CGX_SAMPLER 300
#define CGX_SAMPLER_ONEGAP 65
#define CGX_SAMPLER_TWOGAP 70
#define CGX_MAXSCORE 99.0f

struct cgx_tok8 { int32_t tok; uint32_t rlp; };   // one corpus position: token id and its alignment word, side by side
struct cgx_view {            // read-only index arrays (device pointers on the GPU)
    const int32_t *str;      // source tokens
    const uint32_t *rlp;     // (L<<24)|(R<<16)|(P<<8) per source token; delimiter slot = target offset of next sentence
    const uint8_t *ltar;     // per target token: min aligned source position (in sentence), 255 = none
    const uint8_t *rtar;
    uint32_t n;
    // derived device-only layouts (built after upload, rebuilt on replicas):
    const cgx_tok8 *tok8;    // str and rlp interleaved: a 16-token window is ONE 128-byte run instead of two 64-byte runs in two arrays
    const uint8_t *lr16;     // ltar/rtar in overlapping 64-byte blocks, one per 8 target words (cgx_lr_block): the back-projection test
                             // of a <= 16-word span reads ONE block with three 16-byte loads instead of a run in each of two arrays
    const uint16_t *ltar16;  // long-sentence mode only (else null): ltar / rtar with 16-bit positions, 0xFFFF = none (then the byte tables and lr16 are unused)
    const uint16_t *rtar16;
    const uint8_t *lrs;      // lr16 blocks again, but addressed from the SOURCE side: the target words of the sentence whose source starts at
    uint32_t lrs_k;          // src0 sit at word lrs_k * src0 + (position in the target sentence).  Null when the corpus does not allow it.
    const void *win;         // the 16 tok8 entries from EVERY position on, as one 128-byte-aligned row per position (16x the bytes of tok8: what 288 GB are for):
                             // a lookup or extraction window is then ONE line of the L2 instead of 1.6-1.8 (cgx_index.inc build_layouts).  Null when it was not built.
};
// token / alignment word of corpus position k: from the interleaved array where it exists (one sector serves both), else from the two plain arrays
CGX_HD int32_t cgx_tok(const cgx_view &v, int64_t k) { return v.tok8 ? v.tok8[k].tok : v.str[k]; }
CGX_HD uint32_t cgx_rlpw(const cgx_view &v, int64_t k) { return v.tok8 ? v.tok8[k].rlp : v.rlp[k]; }

// running min/max of aligned target positions over a set of source tokens
#define CGX_NOPOS 0xFFFF     // "no position yet": above every position (the reference starts its unsigned chars at 255, above every position IT allows)
struct cgx_span {
    int lo, hi;              // CGX_NOPOS / 0 when empty
    CGX_HD void reset() { lo = CGX_NOPOS; hi = 0; }
    CGX_HD void add(int L, int R) { if (lo > L) lo = L; if (hi < R) hi = R; }
    CGX_HD bool empty() const { return lo > hi; }
};

// Alignment word of a source token: the reference layout (L<<24)|(R<<16)|(P<<8) -- min / max aligned target position and the
// token's own position, all inside the sentence, 255 = not aligned -- plus, in the opt-in long-sentence mode only (sentences
// of 255 tokens and more; the reference rejects them, ExtractPair.cu:2683), three more bits for L and for R and
// two for P in the low byte the reference leaves zero: bits 7..5, 4..2, 1..0.  L and R are stored as CODES that skip every
// value whose low byte is 255 (code = p + p / 255, p = low + 255 * high), so "low byte == 255" stays the not-aligned test and a
// reference-format word (high bits zero) decodes to exactly the values it always had.  cgx_L8 / cgx_R8 / cgx_P8 read the bytes
// alone: the inner loops of the lookup kernels use them when the index is in the reference format.
CGX_HD int cgx_L8(uint32_t w) { return (int)((w >> 24) & 0xFF); }
CGX_HD int cgx_R8(uint32_t w) { return (int)((w >> 16) & 0xFF); }
CGX_HD int cgx_P8(uint32_t w) { return (int)((w >> 8) & 0xFF); }
CGX_HD int cgx_L(uint32_t w) { return (int)(((w >> 24) & 0xFF) + 255u * ((w >> 5) & 7u)); }
CGX_HD int cgx_R(uint32_t w) { return (int)(((w >> 16) & 0xFF) + 255u * ((w >> 2) & 7u)); }
CGX_HD int cgx_P(uint32_t w) { return (int)(((w >> 8) & 0xFF) | ((w & 3u) << 8)); }
CGX_HD bool cgx_unaligned(uint32_t w) { return ((w >> 24) & 0xFF) == 255 || ((w >> 16) & 0xFF) == 255; }
CGX_HD uint32_t cgx_pos_code(uint32_t p) { return p + p / 255u; }                       // position -> code (low byte never 255)
CGX_HD uint32_t cgx_rlp_pack(int L, int R, uint32_t P) {                                  // L / R < 0: not aligned
    const uint32_t cl = L < 0 ? 255u : cgx_pos_code((uint32_t)L), cr = R < 0 ? 255u : cgx_pos_code((uint32_t)R);
    return ((cl & 255u) << 24) | ((cr & 255u) << 16) | ((P & 255u) << 8) | ((cl >> 8) << 5) | ((cr >> 8) << 2) | (P >> 8);
}
#define CGX_LONG_MAX_SRC 1024   // P has 10 bits
#define CGX_LONG_MAX_TGT 2040   // codes up to 2046

// sentence bookkeeping for the token at k: *src0 = index of the first token of its source
// sentence, returns the offset of its target sentence (GappyLook.cu:70-77)
CGX_HD int cgx_sentence(const cgx_view &v, int k, uint32_t w, int *src0) {
    int prev_delim = k - cgx_P(w) - 1;
    *src0 = prev_delim + 1;
    return prev_delim == -1 ? 0 : (int)cgx_rlpw(v, prev_delim);
}

// The block layout of lr16 and lrs: one 64-byte block per EIGHT words, holding the L bytes of the 24 words from its first one on,
// then their 24 R bytes (16 bytes unused).  A span of <= 16 words that starts in block i ends in block i, so the bytes a
// back-projection test needs -- a[i], b[i] = L and R of words (ts & ~3) + 4i .. + 3 -- come with THREE aligned 16-byte loads of one
// line.  (Blocks of 16 words without overlap made that ten 4-byte loads over two blocks; what a kernel pays the L1 for is load
// instructions per lane, hardly less for a line it already asked for: tools/micro/gather_coop)
// cgx_lr_block_put writes one word into the (up to three) blocks that hold it, normalised for cgx_tight_blocks: not aligned = (255, 0).
// Both compile for the host too (tests/cpu_sim checks the addressing against the plain tables).
struct __attribute__((aligned(16))) cgx_q16 { uint32_t x, y, z, w; };   // the blocks are 64-byte aligned
CGX_HD void cgx_lr_block_put(uint8_t *out, size_t w, uint8_t L, uint8_t R) {
    const bool none = L == 255 || R == 255;
    for (size_t t = 0; t < 3 && t <= (w >> 3); t++) {
        uint8_t *blk = out + (((w >> 3) - t) << 6); const size_t o = (w & 7) + 8 * t;
        blk[o] = none ? 255 : L; blk[24 + o] = none ? 0 : R;
    }
}
CGX_HD void cgx_lr_block(const uint8_t *tab, int ts, uint32_t (&a)[5], uint32_t (&b)[5]) {
    const cgx_q16 *p = (const cgx_q16 *)(tab + ((size_t)((uint32_t)ts >> 3) << 6));     // 64-byte aligned: three 16-byte loads on the device
    const cgx_q16 q0 = p[0], q1 = p[1], q2 = p[2];         // L of words 0..15 | L 16..23, R 0..7 | R 8..23
    const bool up = (ts & 4) != 0;                          // the span starts in the block's second dword
    a[0] = up ? q0.y : q0.x; a[1] = up ? q0.z : q0.y; a[2] = up ? q0.w : q0.z; a[3] = up ? q1.x : q0.w; a[4] = up ? q1.y : q1.x;
    b[0] = up ? q1.w : q1.z; b[1] = up ? q2.x : q1.w; b[2] = up ? q2.y : q2.x; b[3] = up ? q2.z : q2.y; b[4] = up ? q2.w : q2.z;
}

#if defined(__HIPCC__)
// cgx_tight on the lr16 blocks (also the source-addressed copy), te - ts <= 15.  The blocks hold the bytes NORMALISED: a target word
// that is not aligned has L = 255 and R = 0 (the tables it is built from say 255 / 255), so that it drops out of a minimum of
// L and a maximum of R by itself; the words past te are forced to the same values with two masks per dword.  What is left is a
// 16-way unsigned minimum and maximum, taken two 16-bit lanes at a time -- about 50 vector instructions where the byte-by-byte
// version with its three conditions per word took 150, and this routine is most of what the extraction kernels execute.
typedef unsigned short cgx_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool cgx_tight_blocks(const uint8_t *lr16, int ts, int te, int s_chk, int e_chk, int src0) {
    uint32_t a[5], b[5];
    cgx_lr_block(lr16, ts, a, b);
    const unsigned sh = (unsigned)ts & 3u; const int n = te - ts + 1;                    // words in the span, 1..16
    cgx_u16x2 mn = {0xFFFF, 0xFFFF}, mx = {0, 0};
    CGX_UNROLL
    for (int i = 0; i < 4; i++) {
        uint32_t l4 = __builtin_amdgcn_alignbyte(a[i + 1], a[i], sh), r4 = __builtin_amdgcn_alignbyte(b[i + 1], b[i], sh);
        const int k = n - 4 * i;                                                        // bytes of this dword inside the span
        const uint32_t m = k >= 4 ? 0xFFFFFFFFu : k <= 0 ? 0u : (1u << (8 * k)) - 1u;
        l4 |= ~m; r4 &= m;
        const uint32_t le = __builtin_amdgcn_perm(0u, l4, 0x0C020C00u), lo_ = __builtin_amdgcn_perm(0u, l4, 0x0C030C01u);   // bytes 0,2 and 1,3 as two 16-bit lanes
        const uint32_t re = __builtin_amdgcn_perm(0u, r4, 0x0C020C00u), ro_ = __builtin_amdgcn_perm(0u, r4, 0x0C030C01u);
        mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(__builtin_bit_cast(cgx_u16x2, le), __builtin_bit_cast(cgx_u16x2, lo_)));
        mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(__builtin_bit_cast(cgx_u16x2, re), __builtin_bit_cast(cgx_u16x2, ro_)));
    }
    const int lo = mn.x < mn.y ? mn.x : mn.y, hi = mx.x > mx.y ? mx.x : mx.y;
    // no aligned word at all: the byte-by-byte version ends with lo = CGX_NOPOS, hi = 0
    return src0 + (lo == 255 ? CGX_NOPOS : lo) == s_chk && src0 + hi == e_chk;
}
#endif
// Does the target span [ts,te] project back exactly onto source [s_chk,e_chk]?  Unaligned
// target words inside the span are ignored (ExtractPair.cu:103-133).
CGX_HD bool cgx_tight(const cgx_view &v, int ts, int te, int s_chk, int e_chk, int src0) {
    int lo = CGX_NOPOS, hi = 0;
    if (v.ltar16) {                                           // long-sentence mode: 16-bit tables, plain loop
        for (int k = ts; k <= te; k++) { const int L = v.ltar16[k], R = v.rtar16[k]; if (L == 0xFFFF || R == 0xFFFF) continue; if (lo > L) lo = L; if (hi < R) hi = R; }
        return src0 + lo == s_chk && src0 + hi == e_chk;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // Every caller has already rejected target spans of 15 words or more, so [ts,te] fits 16 bytes: on the device
    // both byte tables are read with five aligned dword loads each (they are padded) instead of a byte load per
    // word -- the extraction kernels were bound by the number of L1 requests, most of them issued here.
    if (te - ts < 16) {
        if (v.lr16) return cgx_tight_blocks(v.lr16, ts, te, s_chk, e_chk, src0);
        uint32_t a[5], b[5];
        const uint32_t *pl = (const uint32_t *)(v.ltar + (ts & ~3)), *pr = (const uint32_t *)(v.rtar + (ts & ~3));
        CGX_UNROLL
        for (int i = 0; i < 5; i++) { a[i] = pl[i]; b[i] = pr[i]; }
        const unsigned sh = (unsigned)ts & 3u; const int last = te - ts;
        CGX_UNROLL
        for (int i = 0; i < 4; i++) {
            const uint32_t l4 = __builtin_amdgcn_alignbyte(a[i + 1], a[i], sh), r4 = __builtin_amdgcn_alignbyte(b[i + 1], b[i], sh);
            CGX_UNROLL
            for (int k = 0; k < 4; k++) {
                const int L = (int)((l4 >> (8 * k)) & 0xFF), R = (int)((r4 >> (8 * k)) & 0xFF);
                if (4 * i + k <= last && L != 255 && R != 255) { if (lo > L) lo = L; if (hi < R) hi = R; }
            }
        }
        return src0 + lo == s_chk && src0 + hi == e_chk;
    }
#endif
    for (int k = ts; k <= te; k++) {
        int L = v.ltar[k], R = v.rtar[k];
        if (L == 255 || R == 255) continue;
        if (k == ts) { lo = L; hi = R; } else { if (lo > L) lo = L; if (hi < R) hi = R; }
    }
    return src0 + lo == s_chk && src0 + hi == e_chk;
}

// Source span [start,ender] accumulated into a target span.  Returns false when an edge
// token is unaligned (ExtractPair.cu:154-181).  *src0/*tb are set from the first token.
CGX_HD bool cgx_edge_span(const cgx_view &v, uint32_t start, uint32_t ender, cgx_span *sp, int *src0, int *tb) {
    sp->reset(); *src0 = 0; *tb = -1;
    for (uint32_t k = start; k <= ender; k++) {
        uint32_t w = cgx_rlpw(v, k);
        bool un = cgx_unaligned(w);
        if (un && (k == start || k == ender)) return false;
        if (un) continue;
        if (k == start) { *tb = cgx_sentence(v, (int)k, w, src0); sp->lo = cgx_L(w); sp->hi = cgx_R(w); }
        else sp->add(cgx_L(w), cgx_R(w));
    }
    return true;
}

// checkBoundaryGap: is [start,ender] a valid gap (edges aligned, target width < 15, tight)?
CGX_HD bool cgx_gap_ok(const cgx_view &v, uint32_t start, uint32_t ender) {
    cgx_span sp; int src0, tb;
    if (!cgx_edge_span(v, start, ender, &sp, &src0, &tb)) return false;
    if (sp.empty() || sp.hi - sp.lo >= CGX_MAX_SPAN) return false;
    return cgx_tight(v, sp.lo + tb, sp.hi + tb, (int)start, (int)ender, src0);
}

// checkBoundary: whole-phrase span with the reference's result codes
//   0 not consistent, 1 consistent, 2 first token unaligned, 3 last token unaligned, 4 both.
// *ts/*te receive the target span computed from the aligned tokens even when the code != 1.
CGX_HD int cgx_span_code(const cgx_view &v, uint32_t start, uint32_t ender, uint32_t *ts, uint32_t *te) {
    int lo = CGX_NOPOS, hi = 0, src0 = 0, tb = -1, wrong = 0;
    for (uint32_t k = start; k <= ender; k++) {
        uint32_t w = cgx_rlpw(v, k);
        bool un = cgx_unaligned(w);
        if (un && (k == start || k == ender)) {
            if (start == ender && wrong == 0) wrong = 4;
            else if (wrong == 0 && k == start) wrong = 2;
            else if (wrong == 0 && k == ender) wrong = 3;
            else wrong = 4;
            if (k == start) tb = cgx_sentence(v, (int)k, w, &src0);
        } else if (un) {
        } else if (k == start) { tb = cgx_sentence(v, (int)k, w, &src0); lo = cgx_L(w); hi = cgx_R(w); }
        else { if (lo > cgx_L(w)) lo = cgx_L(w); if (hi < cgx_R(w)) hi = cgx_R(w); }
    }
    *ts = (uint32_t)(lo + tb); *te = (uint32_t)(hi + tb);
    if (wrong) return wrong;
    if (lo <= hi && hi - lo < CGX_MAX_SPAN && cgx_tight(v, (int)*ts, (int)*te, (int)start, (int)ender, src0)) return 1;
    return 0;
}

// index of the k-th sampled occurrence out of n with cap S (n > S): ROUND(k * (float)n/(float)S)
CGX_HD int cgx_sample_index(int n, int S, int k) {
    if (n <= S) return k;
    float step = (float)n / (float)S;
    return (int)((double)((float)k * step) + 0.5);
}

// ---- outputs of one occurrence ----
struct cgx_r0 { int32_t tar_start, block; uint8_t tar_end; uint8_t valid; };
struct cgx_r1 { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1, valid; };
struct cgx_r2 { int32_t id; uint32_t tstart; uint8_t end, gap1, gap1_1, gap2, gap2_1, valid; };

CGX_HD void cgx_put1(cgx_r1 *o, int32_t id, uint32_t ts, uint32_t te, uint32_t gs, uint32_t ge) {
    o->id = id; o->tstart = ts; o->end = (uint8_t)(te - ts); o->gap1 = (uint8_t)(gs - ts); o->gap1_1 = (uint8_t)(ge - ts); o->valid = 1;
}
CGX_HD void cgx_put2(cgx_r2 *o, int32_t id, uint32_t ts, uint32_t te, uint32_t g1s, uint32_t g1e, uint32_t g2s, uint32_t g2e) {
    o->id = id; o->tstart = ts; o->end = (uint8_t)(te - ts); o->gap1 = (uint8_t)(g1s - ts); o->gap1_1 = (uint8_t)(g1e - ts);
    o->gap2 = (uint8_t)(g2s - ts); o->gap2_1 = (uint8_t)(g2e - ts); o->valid = 1;
}

// One sampled corpus occurrence (starting at source position cur, lm tokens) of contiguous
// block bnum: emits ab, the first consistent Xab and abX, and the first consistent XabX.
// Returns true when the reference thread would have taken one of its `printf; return;`
// exits (ExtractPair.cu:1306-1314 etc.), which also cancels its later strided occurrences.
CGX_HD bool cgx_extract_contig(const cgx_view &v, int32_t bnum, int32_t G, int lm, int cur,
                               cgx_r0 *o_ab, cgx_r1 *o_Xab, cgx_r1 *o_abX, cgx_r2 *o_XabX) {
    o_ab->valid = 0; o_Xab->valid = 0; o_abX->valid = 0; o_XabX->valid = 0;
    const int ender = cur + lm - 1;
    int src0 = 0, tb = -1;
    cgx_span body; body.reset();
    bool abX = true, Xab = true, XabX = true, ab = true, XabOpen = true, abXOpen = true;
    for (int k = cur; k <= ender; k++) {
        uint32_t w = cgx_rlpw(v, k);
        if (k == cur) tb = cgx_sentence(v, k, w, &src0);
        if (cgx_unaligned(w)) {
            if (k == cur || k == ender) { ab = false; if (k == cur) abXOpen = false; else XabOpen = false; }
        } else body.add(cgx_L(w), cgx_R(w));
    }
    if (body.empty() || body.hi - body.lo >= CGX_MAX_SPAN) { abX = Xab = XabX = ab = false; }
    if (ab && cgx_tight(v, body.lo + tb, body.hi + tb, cur, ender, src0)) {
        o_ab->tar_start = body.lo + tb; o_ab->tar_end = (uint8_t)(body.hi - body.lo); o_ab->block = bnum; o_ab->valid = 1;
    }
    if (lm + 1 > CGX_MAX_SYMBOLS) { abX = false; Xab = false; }
    if (lm + 2 > CGX_MAX_SYMBOLS) XabX = false;
    cgx_span left, right, other; left.reset(); right.reset(); other.reset();
    int leftLen = 0, rightLen = 0;                 // XabCount / abXCount: widest tight gap seen so far on each side
    uint32_t gs = 0, ge = 0, ts = 0, te = 0;
    for (int i = 1; lm + i <= CGX_MAX_SPAN && (abXOpen || XabOpen || XabX); i++) {
        if (Xab && cur - i >= 0 && cgx_tok(v, cur - i) >= 2) {                // widen the left gap by one token
            bool next = true;
            uint32_t w = cgx_rlpw(v, cur - i);
            if (cgx_unaligned(w)) { next = false; if (i == 1) { Xab = false; XabX = false; } }
            else left.add(cgx_L(w), cgx_R(w));
            if (next && left.empty()) return true;
            if (left.hi - left.lo >= CGX_MAX_SPAN) { next = false; Xab = false; }
            if (next) {
                gs = (uint32_t)(tb + left.lo); ge = (uint32_t)(tb + left.hi);
                next = cgx_tight(v, (int)gs, (int)ge, cur - i, cur - 1, src0);
                if (next) leftLen = i;
            }
            if (XabOpen && next) {
                ts = (uint32_t)(tb + (left.lo < body.lo ? left.lo : body.lo));
                te = (uint32_t)(tb + (left.hi < body.hi ? body.hi : left.hi));
                if (ts > te) return true;
                if (te - ts >= CGX_MAX_SPAN) { next = false; Xab = false; }
                if (next) next = cgx_tight(v, (int)ts, (int)te, cur - i, ender, src0);
                if (next) { cgx_put1(o_Xab, bnum, ts, te, gs, ge); XabOpen = false; }
            }
        } else Xab = false;

        if (abX && cgx_tok(v, ender + i) >= 2) {                              // widen the right gap by one token
            bool next = true;
            uint32_t w = cgx_rlpw(v, ender + i);
            if (cgx_unaligned(w)) { next = false; if (i == 1) { abX = false; XabX = false; } }
            else right.add(cgx_L(w), cgx_R(w));
            if (next && right.empty()) return true;
            if (right.hi - right.lo >= CGX_MAX_SPAN) { next = false; abX = false; }
            if (next) {
                gs = (uint32_t)(tb + right.lo); ge = (uint32_t)(tb + right.hi);
                next = cgx_tight(v, (int)gs, (int)ge, ender + 1, ender + i, src0);
                if (next) rightLen = i;
            }
            if (abXOpen && next) {
                ts = (uint32_t)(tb + (right.lo < body.lo ? right.lo : body.lo));
                te = (uint32_t)(tb + (right.hi < body.hi ? body.hi : right.hi));
                if (ts > te) return true;
                if (te - ts >= CGX_MAX_SPAN) { next = false; abX = false; }
                if (next) next = cgx_tight(v, (int)ts, (int)te, cur, ender + i, src0);
                if (next) { cgx_put1(o_abX, G + bnum, ts, te, gs, ge); abXOpen = false; }
            }
        } else abX = false;

        if (XabX && (abX || Xab)) {
            if (leftLen == i) {                    // the left gap of width i is tight: pair it with right gaps 1..rightLen
                other.reset();
                for (int ic = 1; XabX && ic <= rightLen; ic++) {
                    bool next = true;
                    if (ic + leftLen + lm <= CGX_MAX_SPAN) {
                        uint32_t w = cgx_rlpw(v, ender + ic);
                        if (cgx_unaligned(w)) { next = false; if (i == 1) return true; }
                        else other.add(cgx_L(w), cgx_R(w));
                    } else { next = false; ic = rightLen + 1; }
                    if (next && other.hi - other.lo >= CGX_MAX_SPAN) { next = false; ic = rightLen + 1; }
                    uint32_t g2s = 0, g2e = 0;
                    if (next) {
                        if (other.empty()) return true;
                        g2s = (uint32_t)(tb + other.lo); g2e = (uint32_t)(tb + other.hi);
                        next = cgx_tight(v, (int)g2s, (int)g2e, ender + 1, ender + ic, src0);
                    }
                    if (next) {
                        int lo = other.lo < left.lo ? other.lo : left.lo; if (lo > body.lo) lo = body.lo;
                        int hi = other.hi < left.hi ? left.hi : other.hi; if (hi < body.hi) hi = body.hi;
                        ts = (uint32_t)(tb + lo); te = (uint32_t)(tb + hi);
                        if (ts > te) return true;
                        if (te - ts >= CGX_MAX_SPAN) { next = false; ic = rightLen + 1; }
                        if (next) next = cgx_tight(v, (int)ts, (int)te, cur - leftLen, ender + ic, src0);
                        if (next) { cgx_put2(o_XabX, bnum, ts, te, (uint32_t)(tb + left.lo), (uint32_t)(tb + left.hi), g2s, g2e); XabX = false; }
                    }
                }
            }
            if (XabX && rightLen == i) {           // the right gap of width i is tight: pair it with left gaps 1..leftLen
                other.reset();
                for (int ic = 1; XabX && ic <= leftLen; ic++) {
                    bool next = true;
                    if (ic + rightLen + lm <= CGX_MAX_SPAN) {
                        uint32_t w = cgx_rlpw(v, cur - ic);
                        if (cgx_unaligned(w)) { next = false; if (i == 1) return true; }
                        else other.add(cgx_L(w), cgx_R(w));
                    } else { ic = leftLen + 1; next = false; }
                    if (next && other.hi - other.lo >= CGX_MAX_SPAN) { ic = leftLen + 1; next = false; }
                    uint32_t g1s = 0, g1e = 0;
                    if (next) {
                        if (other.empty()) return true;
                        g1s = (uint32_t)(tb + other.lo); g1e = (uint32_t)(tb + other.hi);
                        next = cgx_tight(v, (int)g1s, (int)g1e, cur - ic, cur - 1, src0);
                    }
                    if (next) {
                        int lo = other.lo < right.lo ? other.lo : right.lo; if (lo > body.lo) lo = body.lo;
                        int hi = other.hi < right.hi ? right.hi : other.hi; if (hi < body.hi) hi = body.hi;
                        ts = (uint32_t)(tb + lo); te = (uint32_t)(tb + hi);
                        if (ts > te) return true;
                        if (te - ts >= CGX_MAX_SPAN) { next = false; ic = leftLen + 1; }
                        if (next) next = cgx_tight(v, (int)ts, (int)te, cur - ic, ender + rightLen, src0);
                        if (next) { cgx_put2(o_XabX, bnum, ts, te, g1s, g1e, (uint32_t)(tb + right.lo), (uint32_t)(tb + right.hi)); XabX = false; }
                    }
                }
            }
        } else XabX = false;

        if (!XabX) { if (!Xab && XabOpen) XabOpen = false; if (!abX && abXOpen) abXOpen = false; }
    }
    return false;
}

// One sampled occurrence of aXbXc: (cur, firstEnd = offset of b's last token, secondEnd = offset of c's last token)
CGX_HD bool cgx_extract_twogap(const cgx_view &v, int32_t id, int a_len, int b_len, int c_len,
                               uint32_t cur, uint32_t firstEnd, uint32_t secondEnd, cgx_r2 *o) {
    o->valid = 0;
    cgx_span g1, g2; int src0, tb, tb2;
    if (!cgx_edge_span(v, cur + a_len, cur + firstEnd - b_len, &g1, &src0, &tb)) return true;
    if (g1.empty() || g1.hi - g1.lo >= CGX_MAX_SPAN) return true;
    if (!cgx_edge_span(v, cur + firstEnd + 1, cur + secondEnd - c_len, &g2, &src0, &tb2)) return true;
    if (g2.empty() || g2.hi - g2.lo >= CGX_MAX_SPAN) return true;
    uint32_t ts, te;
    if (cgx_span_code(v, cur, cur + secondEnd, &ts, &te) == 1)
        cgx_put2(o, id, ts, te, (uint32_t)(g1.lo + tb), (uint32_t)(g1.hi + tb), (uint32_t)(g2.lo + tb2), (uint32_t)(g2.hi + tb2));
    return false;
}

// One sampled occurrence of aXb, second half: given the gap's target span [g0s,g0e], the whole phrase's span [ts,te] and its
// consistency code, emits aXb, then the first consistent XaXb and aXbX (widening one token at a time on either side).
CGX_HD bool cgx_onegap_tail(const cgx_view &v, int32_t id, int32_t D1, int a_len, int b_len, uint32_t cur, int firstEnd, uint32_t ender,
                            int src0, int tb, uint32_t g0s, uint32_t g0e, uint32_t ts, uint32_t te, int code, cgx_r1 *o_aXb, cgx_r2 *o_XaXb, cgx_r2 *o_aXbX);
// One sampled occurrence of aXb: emits aXb, then the first consistent XaXb and aXbX.
CGX_HD bool cgx_extract_onegap(const cgx_view &v, int32_t id, int32_t D1, int a_len, int b_len,
                               uint32_t cur, int firstEnd, cgx_r1 *o_aXb, cgx_r2 *o_XaXb, cgx_r2 *o_aXbX) {
    o_aXb->valid = 0; o_XaXb->valid = 0; o_aXbX->valid = 0;
    if (cur + (uint32_t)firstEnd - (uint32_t)b_len > v.n) return true;
    const uint32_t ender = cur + (uint32_t)firstEnd;
    cgx_span gap; int src0, tb;
    if (!cgx_edge_span(v, cur + a_len, ender - b_len, &gap, &src0, &tb)) return true;
    if (gap.empty() || gap.hi - gap.lo >= CGX_MAX_SPAN) return true;
    const uint32_t g0s = (uint32_t)(gap.lo + tb), g0e = (uint32_t)(gap.hi + tb);
    uint32_t ts, te;
    int code = cgx_span_code(v, cur, ender, &ts, &te);
    return cgx_onegap_tail(v, id, D1, a_len, b_len, cur, firstEnd, ender, src0, tb, g0s, g0e, ts, te, code, o_aXb, o_XaXb, o_aXbX);
}
CGX_HD bool cgx_onegap_tail(const cgx_view &v, int32_t id, int32_t D1, int a_len, int b_len, uint32_t cur, int firstEnd, uint32_t ender,
                            int src0, int tb, uint32_t g0s, uint32_t g0e, uint32_t ts, uint32_t te, int code, cgx_r1 *o_aXb, cgx_r2 *o_XaXb, cgx_r2 *o_aXbX) {
    // whole-phrase span in sentence coordinates (the reference keeps it in unsigned chars; an empty span is caught by the guard below either way)
    int bodyLo = (int)(ts - (uint32_t)tb), bodyHi = (int)(te - (uint32_t)tb);
    bool left = !(code == 3 || code == 4), right = !(code == 2 || code == 4);
    if ((ts == 0 && te == 0) || bodyLo > bodyHi || g0s < ts || g0e > te) return true;
    if (code == 1) cgx_put1(o_aXb, id, ts, te, g0s, g0e);
    if (a_len + b_len + 2 > CGX_MAX_SYMBOLS) return false;
    cgx_span L, R; L.reset(); R.reset();
    for (int i = 1; firstEnd + 1 + i <= CGX_MAX_SPAN && (left || right); i++) {
        if (left && (int)(cur - (uint32_t)i) >= 0 && cgx_tok(v, cur - i) >= 2) {
            bool next = true;
            uint32_t w = cgx_rlpw(v, cur - i);
            if (cgx_unaligned(w)) { next = false; if (i == 1) left = false; }
            else L.add(cgx_L(w), cgx_R(w));
            if (next && L.empty()) return true;
            if (L.hi - L.lo >= CGX_MAX_SPAN) { next = false; left = false; }
            uint32_t gs = 0, ge = 0;
            if (next) { gs = (uint32_t)(tb + L.lo); ge = (uint32_t)(tb + L.hi); next = cgx_tight(v, (int)gs, (int)ge, (int)(cur - i), (int)cur - 1, src0); }
            if (next) {
                ts = (uint32_t)(tb + (L.lo < bodyLo ? L.lo : bodyLo)); te = (uint32_t)(tb + (L.hi < bodyHi ? bodyHi : L.hi));
                if (ts > te) return true;
                if (te - ts >= CGX_MAX_SPAN) { next = false; left = false; }
                if (next) next = cgx_tight(v, (int)ts, (int)te, (int)(cur - i), (int)ender, src0);
            }
            if (next) { cgx_put2(o_XaXb, id, ts, te, gs, ge, g0s, g0e); left = false; }
        } else left = false;
        if (right && cgx_tok(v, ender + i) >= 2) {
            bool next = true;
            uint32_t w = cgx_rlpw(v, ender + i);
            if (cgx_unaligned(w)) { next = false; if (i == 1) right = false; }
            else R.add(cgx_L(w), cgx_R(w));
            if (next && R.empty()) return true;
            if (R.hi - R.lo >= CGX_MAX_SPAN) { next = false; right = false; }
            uint32_t gs = 0, ge = 0;
            if (next) { gs = (uint32_t)(tb + R.lo); ge = (uint32_t)(tb + R.hi); next = cgx_tight(v, (int)gs, (int)ge, (int)ender + 1, (int)(ender + i), src0); }
            if (next) {
                ts = (uint32_t)(tb + (R.lo < bodyLo ? R.lo : bodyLo)); te = (uint32_t)(tb + (R.hi < bodyHi ? bodyHi : R.hi));
                if (ts > te) return true;
                if (te - ts >= CGX_MAX_SPAN) { next = false; right = false; }
                if (next) next = cgx_tight(v, (int)ts, (int)te, (int)cur, (int)(ender + i), src0);
            }
            if (next) { cgx_put2(o_aXbX, D1 + id, ts, te, g0s, g0e, gs, ge); right = false; }
        } else right = false;
    }
    return false;
}

// ---- lexical table: keys packed (src+1)<<32 | (tgt+1), sorted ascending ----
struct cgx_lexview {
    const uint64_t *key; const float *v1, *v2, *n1, *n2; uint32_t n;    // n1/n2 = -log10f(v1/v2), precomputed on the host
    const uint32_t *row;     // row[s] = first entry whose (src+1) >= s, s = 0..nrow; the table is sorted by (src+1, tgt+1)
    const int32_t *nullt;    // entry index of (NULL, tgt) per target id, -1 if absent
    uint32_t nrow, ntgt;     // valid src+1 values are < nrow; target ids are < ntgt
    // optional open-addressing hash key -> lowest entry index (device index only): one or two probes
    // instead of a binary search over the source word's row
    const uint64_t *hkey = nullptr; const uint32_t *hidx = nullptr; uint32_t hmask = 0; unsigned hshift = 0;
    // optional packed copies for MaxLex: the four values next to the key (one 32-byte slot per probe instead of
    // key + index + four arrays), and (NULL, tgt) as a direct {v1, n1} table (v1 < 0: absent)
    const struct cgx_lexslot *hslot = nullptr; const struct cgx_lexnull *nullv = nullptr;
    // optional presence bits: bit ((key * CGX_LEXBIT_MUL) >> pshift) is set for every key of the table -- a few megabytes that stay in the L2, asked
    // before the pair keys (48 MB at the bench's table: most probes of MaxLex are for pairs that are in no row)
    const uint32_t *pbits = nullptr; unsigned pshift = 0;
};
#define CGX_LEXBIT_MUL 0xD6E8FEB86659FD93ull
struct cgx_lexslot { uint64_t key; float v1, v2, n1, n2; uint64_t pad; };
struct cgx_lexnull { float v1, n1; };
CGX_HD uint64_t cgx_lexkey_pack(int32_t src, int32_t tgt) { return ((uint64_t)(uint32_t)(src + 1) << 32) | (uint32_t)(tgt + 1); }
// index of the (src,tgt) row or -1.  Replaces searchLexFile's binary search over the whole table
// (ExtractPair.cu:2108-2142): (NULL,tgt) is a direct table, other pairs go through the pair hash
// when the view has one, else a per-source row pointer narrows the search to that word's few
// translations.
CGX_HD int64_t cgx_lex_find(const cgx_lexview &t, int32_t src, int32_t tgt) {
    if (src < -1 || tgt < -1) return -1;
    if (src == -1 && tgt >= 0) return (uint32_t)tgt < t.ntgt ? (int64_t)t.nullt[tgt] : -1;
    uint32_t s = (uint32_t)(src + 1);
    if (s >= t.nrow) return -1;
    if (t.hkey) {
        const uint64_t want = cgx_lexkey_pack(src, tgt);
        if (want == 0) return -1;                              // (NULL,NULL) is never a row; 0 marks an empty slot
        uint32_t slot = (uint32_t)((want * 0x9E3779B97F4A7C15ull) >> t.hshift) & t.hmask;
        for (;;) { uint64_t k = t.hkey[slot]; if (k == want) return (int64_t)t.hidx[slot]; if (k == 0) return -1; slot = (slot + 1) & t.hmask; }
    }
    int64_t lo = t.row[s], hi = (int64_t)t.row[s + 1] - 1;
    const uint32_t want = (uint32_t)(tgt + 1);
    while (lo <= hi) { int64_t m = lo + ((hi - lo) >> 1); uint32_t x = (uint32_t)t.key[m]; if (want < x) hi = m - 1; else if (want > x) lo = m + 1; else return m; }
    return -1;
}

// values of the (src,tgt) row: v[0..3] = v1, v2, n1, n2.  false when the pair is not in the table.
// SLOTS: the caller knows the view has the packed slots (the MaxLex kernel is compiled for that case alone: with the general
// routine inlined at its three call sites the kernel carried the row search and the index hash along through every divergent loop)
template <bool SLOTS>
CGX_HD bool cgx_lex_get_t(const cgx_lexview &t, int32_t src, int32_t tgt, float *v) {
    if (src < -1 || tgt < -1) return false;
    if (SLOTS || t.hslot) {
        if (src == -1 && tgt >= 0) {
            if ((uint32_t)tgt >= t.ntgt) return false;
            const cgx_lexnull e = t.nullv[tgt];
            if (e.v1 < 0.0f) return false;
            v[0] = e.v1; v[1] = 0.0f; v[2] = e.n1; v[3] = 0.0f;   // only v1/n1 are read for (NULL, tgt)
            return true;
        }
        if ((uint32_t)(src + 1) >= t.nrow) return false;
        const uint64_t want = cgx_lexkey_pack(src, tgt);
        if (want == 0) return false;
        uint32_t slot = (uint32_t)((want * 0x9E3779B97F4A7C15ull) >> t.hshift) & t.hmask;
        for (;;) {
            const cgx_lexslot e = t.hslot[slot];
            if (e.key == want) { v[0] = e.v1; v[1] = e.v2; v[2] = e.n1; v[3] = e.n2; return true; }
            if (e.key == 0) return false;
            slot = (slot + 1) & t.hmask;
        }
    }
    const int64_t m = cgx_lex_find(t, src, tgt);
    if (m < 0) return false;
    v[0] = t.v1[m]; v[1] = t.v2[m]; v[2] = t.n1[m]; v[3] = t.n2[m];
    return true;
}
CGX_HD bool cgx_lex_get(const cgx_lexview &t, int32_t src, int32_t tgt, float *v) { return cgx_lex_get_t<false>(t, src, tgt, v); }

// MaxLexFgivenE / MaxLexEgivenF of one rule (kind 0: one gap, 1: two gaps, 2: contiguous).
// The reference adds -log10(max) per word in float; the log of every table value is
// precomputed by the host libm, so the kernel only picks the arg-max and adds.
// The reference walks the (source word, target word) pairs twice, once per direction
// (ExtractPair.cu:2145-2290).  Here every pair is looked up ONCE, target word outer, source word
// inner: the per-target maximum is a scalar, the per-source maxima are kept in a small array.
// Each maximum still sees its candidates in the reference's order (NULL first, then ascending
// position, strict '>'), and the two sums are still added in ascending word order, so the
// floats are identical.
#define CGX_MAXLEX_SRC 5
template <bool SLOTS>
CGX_HD void cgx_maxlex_t(const cgx_lexview &t, const int32_t *tstr, const int32_t *src, int nsrc, uint32_t tstart,
                         int end, int gap1, int gap1_1, int gap2, int gap2_1, int kind, float *fe, float *ef) {
    float fgivene = 0.0f, egivenf = 0.0f;
    const int t0 = (int)tstart, tend = t0 + end, g1s = t0 + gap1, g1e = t0 + gap1_1, g2s = t0 + gap2, g2e = t0 + gap2_1;
    if (nsrc <= CGX_MAXLEX_SRC) {
        float mx2[CGX_MAXLEX_SRC], ng2[CGX_MAXLEX_SRC]; bool first = true;
        CGX_UNROLL
        for (int j = 0; j < CGX_MAXLEX_SRC; j++) { mx2[j] = 0.0f; ng2[j] = 0.0f; }
        for (int jj = t0; jj <= tend; jj++) {
            bool outside = kind == 2 || ((jj < g1s || jj > g1e) && (kind == 0 || jj < g2s || jj > g2e));
            if (!outside) continue;
            const int32_t tw = tstr[jj];
            float mx1 = 0.0f, ng1 = 0.0f;
            float q[4];
            if (cgx_lex_get_t<SLOTS>(t, -1, tw, q) && q[0] > mx1) { mx1 = q[0]; ng1 = q[2]; }
        CGX_UNROLL
            for (int j = 0; j < CGX_MAXLEX_SRC; j++) {
                if (j >= nsrc) break;
                if (first && cgx_lex_get_t<SLOTS>(t, src[j], -1, q) && q[1] > mx2[j]) { mx2[j] = q[1]; ng2[j] = q[3]; }
                if (cgx_lex_get_t<SLOTS>(t, src[j], tw, q)) {
                    if (q[0] > mx1) { mx1 = q[0]; ng1 = q[2]; }
                    if (q[1] > mx2[j]) { mx2[j] = q[1]; ng2[j] = q[3]; }
                }
            }
            first = false;
            egivenf += (nsrc > 0 && mx1 > 0.0f) ? ng1 : CGX_MAXSCORE;
        }
        CGX_UNROLL
        for (int j = 0; j < CGX_MAXLEX_SRC; j++) if (j < nsrc) fgivene += mx2[j] > 0.0f ? ng2[j] : CGX_MAXSCORE;
        *fe = fgivene; *ef = egivenf;
        return;
    }
    for (int j = 0; j < nsrc; j++) {
        float mx = 0.0f, neg = 0.0f; bool first = true;
        for (int jj = t0; jj <= tend; jj++) {
            bool outside = kind == 2 || ((jj < g1s || jj > g1e) && (kind == 0 || jj < g2s || jj > g2e));
            if (!outside) continue;
            if (first) { int64_t m = cgx_lex_find(t, src[j], -1); if (m >= 0 && t.v2[m] > mx) { mx = t.v2[m]; neg = t.n2[m]; } first = false; }
            int64_t m = cgx_lex_find(t, src[j], tstr[jj]);
            if (m >= 0 && t.v2[m] > mx) { mx = t.v2[m]; neg = t.n2[m]; }
        }
        fgivene += mx > 0.0f ? neg : CGX_MAXSCORE;
    }
    for (int jj = t0; jj <= tend; jj++) {
        bool outside = kind == 2 || ((jj < g1s || jj > g1e) && (kind == 0 || jj < g2s || jj > g2e));
        if (!outside) continue;
        float mx = 0.0f, neg = 0.0f; bool first = true;
        for (int j = 0; j < nsrc; j++) {
            if (first) { int64_t m = cgx_lex_find(t, -1, tstr[jj]); if (m >= 0 && t.v1[m] > mx) { mx = t.v1[m]; neg = t.n1[m]; } first = false; }
            int64_t m = cgx_lex_find(t, src[j], tstr[jj]);
            if (m >= 0 && t.v1[m] > mx) { mx = t.v1[m]; neg = t.n1[m]; }
        }
        egivenf += mx > 0.0f ? neg : CGX_MAXSCORE;
    }
    *fe = fgivene; *ef = egivenf;
}
CGX_HD void cgx_maxlex(const cgx_lexview &t, const int32_t *tstr, const int32_t *src, int nsrc, uint32_t tstart,
                       int end, int gap1, int gap1_1, int gap2, int gap2_1, int kind, float *fe, float *ef) {
    cgx_maxlex_t<false>(t, tstr, src, nsrc, tstart, end, gap1, gap1_1, gap2, gap2_1, kind, fe, ef);
}

#endif
