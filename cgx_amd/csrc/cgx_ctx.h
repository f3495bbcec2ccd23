// cgx_ctx.h -- the context behind the opaque cgx_ctx handle (device half).
#ifndef CGX_CTX_H
#define CGX_CTX_H
#include <hip/hip_runtime.h>
#include "cgx_rules.h"
#define CGX_COPY_STREAMS 3
struct __attribute__((aligned(16))) cgx_ngslot { unsigned long long key; uint32_t lo, hi; };   // one 16-byte slot: a probe touches one sector
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/cgx.h"

struct cgx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    int k1_limit = 128;                 // K1 launches 128 threads per query sentence (SuffixArray.cu:1374-1378)
    bool async_write = false;           // grammar files of batch k are written by host threads while the GPU runs batch k+1 (cgx_flush joins)
    void *host_state = nullptr;         // owned by the host TU
    unsigned lex_hash_bits = 0;         // test hook: hash bits in the lexicon grouping key (0: all that fit beside the id)
    bool force_host_lexicon = false;    // test hook: take the exact host lexicon path
    uint64_t chunk_items = 1ull << 26;  // work items per lookup launch
    uint64_t append_slack = 65536;
    double look1_per_item = 2.0, look2_per_item = 2.0;   // output records per work item (capacity guess, adapts per batch)
    std::map<std::string, double> ms;   // stage timings

    // ---- index, resident for the life of the context ----
    uint32_t n = 0, nt = 0, nlex = 0, nphits = 0; int32_t last = 0;
    bool have_sa = false, have_pre = false;
    bool index_borrowed = false;        // the index arrays below belong to another context of this device (cgx_share_index): used, never freed
    int32_t *d_str = nullptr, *d_sa = nullptr, *d_tstr = nullptr;
    uint32_t *d_rlp = nullptr;
    uint8_t *d_ltar = nullptr, *d_rtar = nullptr;
    bool long_pos = false;              // long-sentence mode (cgx_rules.h): alignment words carry extra position bits, target-side tables are 16-bit
    uint16_t *d_ltar16 = nullptr, *d_rtar16 = nullptr;
    uint4 *d_win = nullptr; bool win_table = false;    // every position's 16-entry window as an aligned 128-byte row (cgx_view::win); option "win_table", OFF: measured, buys nothing (cgx_index.inc)
    cgx_tok8 *d_tok8 = nullptr; uint8_t *d_lr16 = nullptr;   // derived layouts (cgx_rules.h), built by build_layouts
    double append_ms = 0.0;             // GPU time of the launches of the append passes since the caller last zeroed it (append_pass)
    uint8_t *d_lrs = nullptr; uint32_t lrs_k = 0;            // lr16 blocks addressed from the source side (cgx_view::lrs); null / 0 when the corpus does not allow them
    int32_t *d_pos1 = nullptr;          // derived: the corpus positions of every token in ascending order, token by token (same buckets as the suffix array's one-token intervals: tokstart)
    uint64_t *d_lexkey = nullptr; float *d_lexv1 = nullptr, *d_lexv2 = nullptr, *d_lexn1 = nullptr, *d_lexn2 = nullptr;
    uint32_t *d_lexrow = nullptr; int32_t *d_lexnullt = nullptr; uint32_t lex_nrow = 0, lex_ntgt = 0;
    cgx_lexslot *d_lexslot = nullptr; cgx_lexnull *d_lexnullv = nullptr;
    uint64_t *d_lexhkey = nullptr; uint32_t *d_lexhidx = nullptr; uint32_t lex_hmask = 0; unsigned lex_hshift = 0;   // pair hash (derived, rebuilt on replicas)
    uint32_t *d_lexpbits = nullptr; unsigned lex_pshift = 0; bool lex_bits = true;   // presence bits in front of the pair keys (cgx_lexview::pbits); option "lex_bits"
    int32_t *d_tokstart = nullptr; int8_t *d_tokrank = nullptr; int32_t *d_freq = nullptr;
    uint32_t *d_pidx = nullptr; int32_t *d_miss = nullptr; uint32_t *d_phit_start = nullptr; uint8_t *d_phit_len = nullptr;
    cgx_ngslot *d_ng[4] = {nullptr, nullptr, nullptr, nullptr}; uint32_t ng_cap[4] = {0, 0, 0, 0}; unsigned ng_shift[4] = {0, 0, 0, 0};   // l-gram (l = 2..5) -> SA interval
    int gz_level = 0;                   // 1..9: grammar.<q>.s.gz instead of plain files
    bool gz_device = true;              // with gz_level and the device formatter: the formatter emits the gzip members itself (cgx_fmt.h); 0 = the host's zlib at gz_level compresses the plain text
    uint32_t *d_gztab = nullptr;        // CRC tables of the deflate output (GZ_TAB_WORDS words)
    struct gz_code *d_gzcode = nullptr; // the Huffman codes of the batch being formatted (cgx_fmt.h)
    bool gz_dynamic = true;             // dynamic Huffman codes made from a tally of the batch (0: the fixed codes of RFC 1951 3.2.6)
    uint8_t gz_lit[256] = {0};          // literal bytes a grammar line can hold (cgx_upload_vocab)
    unsigned int *d_rs_long = nullptr;  // run_sort: runs longer than the fix pass's LDS buffer met so far (slow path taken; cgx_stage_ms "run_sort_long_runs")
    bool text_gz[2] = {false, false};   // what the text slot holds: deflate pieces / plain text
    uint32_t *d_trl[2] = {nullptr, nullptr};   // deflate pieces: CRC-32 and ISIZE of every query's file (k_gz_files)
    bool occ_order = true;              // one-token driving phrases take their occurrences in corpus order (d_pos1) instead of suffix order (test / A-B hook)
    bool src_blocks = true;             // the lookups find a sentence's target-side bytes from its source start (d_lrs); 0 = through the delimiter's alignment word (test / A-B hook)
    bool use_layouts = true;            // test hook: 0 = window kernels read the plain str / rlp / ltar / rtar arrays (round-1 access pattern)
    int ngram_max = 5;                  // longest phrase answered from the l-gram tables (1: none, every l >= 2 by binary search)
    bool count_probes = false;          // cgx_sa_lookup also runs the probe-counting variant of its kernel (untimed; "sa_probe_*")
    bool prealloc_text = false;         // allocate both text slots at the first batch (set when more batches will follow)
    bool numa_pin = true;               // writer threads run on the CPUs of the GPU's NUMA node
    bool use_lex_hash = true;           // MaxLex pair lookups through the pair hash (0: binary search in the source word's row)
    int64_t auto_batch_tokens = 300000; // with sub_batch == 0: query tokens per internal batch (bounds device memory per call)
    int64_t sub_batch = 0;              // queries per internal batch of cgx_extract_grammars* (0 = all at once)
    int64_t write_period = 0, write_count = 0;   // write_period > 0: only the grammar files of queries g with g % write_period < write_count are written (rules are still counted for all); batches without such a query skip text layout and DMA
    int64_t fault_inject = 0;           // test hook: the n-th device allocation from now fails
    uint32_t pool_cap = 1u << 30;       // test hook: entries of the per-block append pool in use (clamped to POOL_N)
    uint32_t look_rec_cap = 65535;      // test hook: groups with more records than this read them from global memory
    bool wide_hits2 = false;            // test hook: take the path of two-gap pattern ids that do not fit beside the occurrence
    bool tile_order = false;            // 1: the tiles of k_look1 launched in corpus-region order (cgx_search.inc k_tile_region) -- measured in round 4: 29.8 -> 29.3 ms, not worth its sort; kept as an A/B switch
    int lex_flat = 1;                   // MaxLex as one task list per wave (k_lex_finish_flat); 0: one lane per line (k_lex_finish), the A/B switch; 2 (test hook): as 1, but the "not covered" answer is forced, so that the redo path runs
    bool hit_order = false;             // 1: the hit lists are sorted completely (what cgx_fetch "hits1" / "hits2" callers may want); 0: as far as extraction needs
    int32_t freq[100] = {0};

    // ---- batch ----
    int32_t nq = 0, ntok = 0;
    std::vector<int32_t> h_qoff, h_tok2q;
    int32_t *d_qoff = nullptr, *d_qtok = nullptr, *d_tok2q = nullptr;
    int32_t *d_lm = nullptr, *d_up = nullptr, *d_down = nullptr;
    uint32_t e1 = 0, d1 = 0, h1 = 0, e2 = 0, d2 = 0, h2 = 0;
    cgx_gappy *d_g1 = nullptr; cgx_gappat *d_p1 = nullptr; uint32_t *d_pid1 = nullptr; cgx_gapsearch *d_s1 = nullptr; cgx_hit1 *d_hits1 = nullptr;
    cgx_twogappy *d_g2 = nullptr; int32_t *d_c2 = nullptr; uint32_t *d_pid2 = nullptr; cgx_twogapsearch *d_s2 = nullptr; cgx_hit2 *d_hits2 = nullptr;
    // the hit lists as the stages use them (cgx_search.inc "hitview"): 64-bit keys grouped by pattern, ordered inside a pattern by the top bits of the
    // start position (okey >> hb) or completely (hit_order); d_hits1 / d_hits2 above are made from them on request (cgx_fetch)
    uint64_t *d_hk1 = nullptr, *d_hk2 = nullptr; uint32_t *d_hid2 = nullptr; unsigned hs1 = 4, hs2 = 4, hb1 = 0, hb2 = 0;
    cgx_gappat *d_p1d = nullptr; int32_t *d_c2d = nullptr; uint32_t *d_one2 = nullptr;   // per distinct pattern
    uint32_t g = 0; cgx_block *d_blocks = nullptr;
    uint32_t n0 = 0, n1 = 0, n2 = 0, sep1 = 0, sep2a = 0, sep2b = 0, guard_exits = 0;
    cgx_rule0 *d_r0 = nullptr; cgx_rule1 *d_r1 = nullptr; cgx_rule2 *d_r2 = nullptr;
    int32_t *d_rng0 = nullptr, *d_rng1 = nullptr, *d_rng2 = nullptr;   // id -> [first,last] lexicon line
    cgx_lexent *d_lex0 = nullptr, *d_lex1 = nullptr, *d_lex2 = nullptr; uint32_t nl0 = 0, nl1 = 0, nl2 = 0;

    void *arena[2] = {nullptr, nullptr}; size_t arena_cap[2] = {0, 0}, arena_used[2] = {0, 0}; int arena_sel = 0;   // pinned result arenas

    // ---- device text formatter ----
    char *d_spool = nullptr, *d_tpool = nullptr; uint32_t *d_soff = nullptr, *d_toff = nullptr; uint32_t vocab_ns = 0, vocab_nt = 0;
    float *d_aa = nullptr, *d_bb = nullptr, *d_fs = nullptr;
    uint32_t nqb = 0, nqo = 0, nqt = 0;                       // entries of d_qb_ids / d_qo_ids / d_qt_ids
    uint32_t *d_qb_off = nullptr, *d_qb_ids = nullptr, *d_qo_off = nullptr, *d_qo_ids = nullptr, *d_qt_off = nullptr, *d_qt_ids = nullptr;
    char *d_text[2] = {nullptr, nullptr}; size_t text_cap[2] = {0, 0}; uint64_t text_bytes[2] = {0, 0}; uint64_t *d_qtext[2] = {nullptr, nullptr}; int32_t text_nq[2] = {0, 0}; int text_sel = 0;
    uint64_t *d_seg_off[2] = {nullptr, nullptr}, *d_qseg[2] = {nullptr, nullptr}; uint32_t *d_seg_len[2] = {nullptr, nullptr}; uint64_t text_nseg[2] = {0, 0}, text_total[2] = {0, 0};   // pieces of the unique text per query
    hipStream_t copy_streams[CGX_COPY_STREAMS] = {nullptr};   // few, so that they do not share a hardware queue with `stream`
    hipEvent_t sync_ev = nullptr;                             // blocking-sync event for host waits on `stream`
    void *h_small = nullptr;                                  // 4 KB of page-locked host memory: where the counters a stage reads back land (a copy into pageable memory is staged inside the runtime, under its lock)
    hipEvent_t copy_done[CGX_MAX_READERS] = {nullptr};        // one per reader: its last enqueued copy
    const void *vocab_owner = nullptr;
    bool device_format = true;          // lay the grammar text out on the GPU (host formatter kept as the fallback)

    // ---- host-side stage timings of the whole-path driver ----
    std::map<std::string, double> host_ms;
};
#endif
