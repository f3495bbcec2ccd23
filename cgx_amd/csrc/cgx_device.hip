// cgx_device.hip -- gfx950 kernels and the device half of the C ABI (include/cgx.h).
//
// Design (MI355X-first):
//  * the whole corpus index stays resident in HBM for the life of the context: token ids,
//    suffix array, packed alignment words, target-side alignment bytes, lexical table,
//    a per-token SA bucket table and the frequent-pair lists;
//  * every variable-size result is produced by count -> scan -> fill (ordered compaction)
//    or by wave-aggregated appends followed by a radix sort on the full record, so no result
//    depends on atomic arrival order and nothing has a fixed capacity;
//  * sort / scan / reduce are rocPRIM; integer gathers only, no MFMA.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include "../../include/cgx.h"
#include "cgx_rules.h"
#include "cgx_ctx.h"
#include "cgx_internal.h"

// ------------------------------------------------------------------------------------
// error handling / memory helpers
// ------------------------------------------------------------------------------------
static int fail(cgx_ctx *c, int code, const char *what, hipError_t e) {
    snprintf(c->err, sizeof c->err, "%s: %s", what, e == hipSuccess ? "failed" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(ctx, CGX_ERR_HIP, #x, e_); } while (0)
#define TRY(x) do { int r_ = (x); if (r_ != CGX_OK) return r_; } while (0)

// Caching device allocator: hipMalloc/hipFree of multi-GB scratch cost 100s of ms per batch and
// hipFree synchronises the device.  Freed blocks are kept (grow-only pool, best fit within 2x)
// and reused by later stages/batches on the same stream; cgx_destroy returns them to the driver.
#include <unordered_map>
#include <map>
// Caching device allocator: multi-GB hipMalloc/hipFree calls cost up to a second each, and every batch needs the same
// sizes again.  One pool per device (a block is only ever reused on the device it came from), one lock for all of
// them (contexts on different devices may be driven from different threads).
#include <mutex>
// A freed block may still be in use by kernels queued on the stream of the context that freed it.  Reuse by the SAME
// stream is ordered behind them; a block handed to a context with another stream (two contexts on one device) first
// waits for the stream it was last used on.
struct DevPool {
    struct blk { size_t bytes; const void *owner; hipStream_t stream; };   // owner != null: a temporary of that context (see dalloc)
    struct idle { void *p; hipStream_t stream; };
    std::unordered_map<void *, blk> live;
    std::multimap<size_t, idle> cached;
    size_t cached_bytes = 0;
    void *get(size_t bytes, const void *owner, hipStream_t stream) {
        // best fit within 2x among the blocks last used on THIS stream (or on none); a block of another stream -- two contexts working on one
        // device at the same time -- would have to wait for that stream to drain, so it is taken only when the driver has no memory left
        auto other = cached.end();
        for (auto it = cached.lower_bound(bytes); it != cached.end() && it->first <= bytes * 2 + (1u << 20); ++it) {
            if (it->second.stream && it->second.stream != stream) { if (other == cached.end()) other = it; continue; }
            idle e = it->second; const size_t have = it->first;
            cached_bytes -= have; cached.erase(it);
            live[e.p] = blk{have, owner, stream};
            return e.p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) == hipSuccess) { live[p] = blk{bytes, owner, stream}; return p; }
        (void)hipGetLastError();
        if (other != cached.end()) {
            idle e = other->second; const size_t have = other->first;
            cached_bytes -= have; cached.erase(other);
            (void)hipStreamSynchronize(e.stream);
            live[e.p] = blk{have, owner, stream};
            return e.p;
        }
        if (hipMalloc(&p, bytes) != hipSuccess) { trim(); (void)hipGetLastError(); if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; }
        live[p] = blk{bytes, owner, stream};
        return p;
    }
    bool put(void *p) {
        auto it = live.find(p);
        if (it == live.end()) return false;
        cached.insert({it->second.bytes, idle{p, it->second.stream}}); cached_bytes += it->second.bytes; live.erase(it);
        return true;
    }
    size_t sweep(const void *owner) {                         // temporaries a failed call of `owner` left behind
        size_t n = 0;
        for (auto it = live.begin(); it != live.end();) {
            if (it->second.owner == owner) { cached.insert({it->second.bytes, idle{it->first, it->second.stream}}); cached_bytes += it->second.bytes; it = live.erase(it); n++; }
            else ++it;
        }
        return n;
    }
    void forget(hipStream_t stream) { for (auto &kv : cached) if (kv.second.stream == stream) kv.second.stream = nullptr; }   // the stream is about to be destroyed (already synchronised)
    void trim() { for (auto &kv : cached) (void)hipFree(kv.second.p); cached.clear(); cached_bytes = 0; }
};
#define CGX_MAX_DEVICES 64
static DevPool g_pools[CGX_MAX_DEVICES];
static std::mutex g_pool_lock;
static DevPool &pool_of(int device) { return g_pools[device < 0 || device >= CGX_MAX_DEVICES ? 0 : device]; }
static void *pool_get(int device, size_t bytes, const void *owner, hipStream_t stream) {
    std::lock_guard<std::mutex> g(g_pool_lock);
    return pool_of(device).get(bytes, owner, stream);         // the caller has made `device` current
}
static void pool_put(void *p) {
    std::lock_guard<std::mutex> g(g_pool_lock);
    for (int d = 0; d < CGX_MAX_DEVICES; d++) if (g_pools[d].put(p)) return;
    (void)hipFree(p);                                         // not ours (never happens): plain free
}
static void pool_trim(int device) {
    std::lock_guard<std::mutex> g(g_pool_lock);
    pool_of(device).trim();
}
// Every stage entry point starts with this: device temporaries of an earlier call that returned through TRY with an
// error are handed back to the pool.  A block is a temporary when the pointer that receives it is not a member of the
// context object (a local of the stage function); results live in context members and are left alone.
static void stage_enter(cgx_ctx *ctx) {
    std::lock_guard<std::mutex> g(g_pool_lock);
    size_t n = pool_of(ctx->device).sweep(ctx);
    if (n) ctx->ms["swept_temporaries"] += (double)n;
}
template <class T> static int dalloc(cgx_ctx *ctx, T **p, size_t count) {
    size_t bytes = (count ? count : 1) * sizeof(T);
    bytes = (bytes + 255) & ~(size_t)255;
    const bool member = (const char *)p >= (const char *)ctx && (const char *)p < (const char *)(ctx + 1);
    if (ctx->fault_inject > 0 && --ctx->fault_inject == 0) { *p = nullptr; return fail(ctx, CGX_ERR_NOMEM, "device allocation (injected fault)", hipErrorOutOfMemory); }
    *p = (T *)pool_get(ctx->device, bytes, member ? nullptr : (const void *)ctx, ctx->stream);
    if (!*p) return fail(ctx, CGX_ERR_NOMEM, "device allocation", hipErrorOutOfMemory);
    return CGX_OK;
}
template <class T> static void dfree(T *&p) { if (p) { pool_put((void *)p); p = nullptr; } }
static int dalloc_bytes(cgx_ctx *ctx, void **p, size_t bytes) { return dalloc(ctx, (char **)p, bytes); }
static void dfree_bytes(void *p) { if (p) pool_put(p); }
static inline unsigned nblocks(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
static int bits_for(uint64_t maxval) { int b = 1; while (b < 64 && (maxval >> b)) b++; return b; }

// Host waits sleep instead of spinning (blocking-sync event): the process's CPU share belongs to the writer threads.
static hipError_t stream_wait(cgx_ctx *ctx) {
    if (!ctx->sync_ev) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->sync_ev, ctx->stream);
    return e != hipSuccess ? e : hipEventSynchronize(ctx->sync_ev);
}
struct Timer {
    hipEvent_t a, b; hipStream_t s;
    Timer(hipStream_t st) : s(st) { (void)hipEventCreate(&a); (void)hipEventCreateWithFlags(&b, hipEventBlockingSync); (void)hipEventRecord(a, s); }
    double stop() { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); (void)hipEventDestroy(a); (void)hipEventDestroy(b); return ms; }
};

// rocPRIM wrappers (temporary storage allocated per call; all on ctx->stream)
template <class K, class V>
static int sort_pairs(cgx_ctx *ctx, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned b0, unsigned b1) {
    size_t tb = 0; void *tmp = nullptr;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, b0, b1, ctx->stream));
    TRY(dalloc_bytes(ctx, &tmp, tb ? tb : 1));
    hipError_t e = rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, b0, b1, ctx->stream);
    dfree_bytes(tmp);
    HIPCHK(e);
    return CGX_OK;
}
template <class K>
static int sort_keys(cgx_ctx *ctx, const K *kin, K *kout, size_t n, unsigned b0, unsigned b1) {
    size_t tb = 0; void *tmp = nullptr;
    HIPCHK(rocprim::radix_sort_keys(nullptr, tb, kin, kout, n, b0, b1, ctx->stream));
    TRY(dalloc_bytes(ctx, &tmp, tb ? tb : 1));
    hipError_t e = rocprim::radix_sort_keys(tmp, tb, kin, kout, n, b0, b1, ctx->stream);
    dfree_bytes(tmp);
    HIPCHK(e);
    return CGX_OK;
}
template <class In, class Out>
static int excl_scan(cgx_ctx *ctx, const In *in, Out *out, size_t n) {
    size_t tb = 0; void *tmp = nullptr;
    HIPCHK(rocprim::exclusive_scan(nullptr, tb, in, out, (Out)0, n, rocprim::plus<Out>(), ctx->stream));
    TRY(dalloc_bytes(ctx, &tmp, tb ? tb : 1));
    hipError_t e = rocprim::exclusive_scan(tmp, tb, in, out, (Out)0, n, rocprim::plus<Out>(), ctx->stream);
    dfree_bytes(tmp);
    HIPCHK(e);
    return CGX_OK;
}
template <class In, class Out>
static int incl_scan(cgx_ctx *ctx, const In *in, Out *out, size_t n) {
    size_t tb = 0; void *tmp = nullptr;
    HIPCHK(rocprim::inclusive_scan(nullptr, tb, in, out, n, rocprim::plus<Out>(), ctx->stream));
    TRY(dalloc_bytes(ctx, &tmp, tb ? tb : 1));
    hipError_t e = rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::plus<Out>(), ctx->stream);
    dfree_bytes(tmp);
    HIPCHK(e);
    return CGX_OK;
}
#define CGX_SMALL_COPY 4096
template <class T> static int d2h(cgx_ctx *ctx, T *dst, const T *src, size_t count) {
    const size_t bytes = count * sizeof(T);
    if (ctx->h_small && bytes <= CGX_SMALL_COPY) {               // the counters of a stage: through the context's page-locked word, so that the copy is a plain asynchronous DMA
        HIPCHK(hipMemcpyAsync(ctx->h_small, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(stream_wait(ctx));
        memcpy(dst, ctx->h_small, bytes);
        return CGX_OK;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(stream_wait(ctx));
    return CGX_OK;
}
template <class T> static int h2d(cgx_ctx *ctx, T *dst, const T *src, size_t count) {
    HIPCHK(hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(stream_wait(ctx));
    return CGX_OK;
}

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------
// last index i in [0,n] with off[i] <= w  (off is an exclusive-scan array with off[n] = total)
__device__ __forceinline__ uint32_t seg_of(const uint64_t *off, uint32_t n, uint64_t w) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t m = (lo + hi + 1) >> 1; if (off[m] <= w) lo = m; else hi = m - 1; }
    return lo;
}
// wave-aggregated append: returns the slot of this lane's record (valid lanes only)
__device__ __forceinline__ uint32_t wave_append(unsigned int *counter, bool valid) {
    unsigned long long m = __ballot(valid);
    uint32_t base = 0;
    int lane = (int)(threadIdx.x & 63);
    int leader = __ffsll((long long)m) - 1;
    if (m && lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = __shfl(base, leader < 0 ? 0 : leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// ---- single-pass variable-output append ---------------------------------------------
// Lanes that produce a variable number of u64 records put them into a per-block LDS pool (LDS
// atomic on the fill count).  At block-uniform points the pool is drained: ONE global atomic
// reserves room for the whole pool and the block copies it out coalesced -- a single output
// counter cannot absorb one atomic per wave (same-address atomics serialise in one L2 channel).
// A record that finds the pool full is appended directly.  The global counter keeps counting
// past `cap`, so the host learns the exact size needed when a launch overflows and can rerun
// it; nothing is written beyond `cap`.
#define POOL_N 1024
struct appender { uint64_t *out; uint64_t cap; unsigned long long *total; uint32_t pool_n; };   // pool_n <= POOL_N: pool entries in use (test hook)
struct pool_t { uint64_t buf[POOL_N]; unsigned int n, snap; unsigned long long base; };
__device__ __forceinline__ uint64_t lanes_reserve(unsigned long long *ctr) {        // callable from divergent code
    unsigned long long m = __ballot(1);
    int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(ctr, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void pool_put(pool_t &P, const appender &ap, uint64_t rec) {
    unsigned int slot = atomicAdd(&P.n, 1u);
    if (slot < ap.pool_n) P.buf[slot] = rec;
    else { uint64_t g = lanes_reserve(ap.total); if (g < ap.cap) ap.out[g] = rec; }
}
// block-uniform: every thread of the block must call it.  Drains when the pool is at least
// `threshold` full (0 = always).
__device__ __forceinline__ void pool_drain(pool_t &P, const appender &ap, unsigned int threshold) {
    __syncthreads();
    if (threadIdx.x == 0) P.snap = P.n;                      // one thread decides, so the branch below is uniform
    __syncthreads();
    unsigned int cnt = P.snap;
    if (cnt < threshold || cnt == 0) return;
    if (cnt > ap.pool_n) cnt = ap.pool_n;
    if (threadIdx.x == 0) P.base = atomicAdd(ap.total, (unsigned long long)cnt);
    __syncthreads();
    const unsigned long long base = P.base;
    for (unsigned int i = threadIdx.x; i < cnt; i += blockDim.x) { unsigned long long g = base + i; if (g < ap.cap) ap.out[g] = P.buf[i]; }
    __syncthreads();
    if (threadIdx.x == 0) P.n = 0;
    __syncthreads();
}

// growable device array of u64
struct dvec64 { uint64_t *p = nullptr; size_t n = 0, cap = 0; };
static int dvec_reserve(cgx_ctx *ctx, dvec64 &v, size_t need) {
    if (need <= v.cap) return CGX_OK;
    size_t nc = v.cap ? v.cap : 1024; while (nc < need) nc *= 2;
    uint64_t *np = nullptr; TRY(dalloc(ctx, &np, nc));
    if (v.n) HIPCHK(hipMemcpyAsync(np, v.p, v.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(stream_wait(ctx));
    dfree(v.p); v.p = np; v.cap = nc;
    return CGX_OK;
}

// Run a single-pass appending kernel over `units` launch units (in launches of at most `chunk`)
// that together cover W work items.
// `out` already holds out.n records; the kernel appends after them.  The capacity is a guess
// (`per_item` records per work item, remembered from the previous batch); if the launch counted
// more than fits it is rerun once with the exact size.
// `reset` runs before every attempt (for kernels with side effects besides the appended records).
template <class Launch, class Reset>
static int append_pass(cgx_ctx *ctx, uint64_t units, uint64_t chunk, uint64_t W, dvec64 &out, double *per_item, Launch launch, Reset reset) {
    unsigned long long *total = nullptr; TRY(dalloc(ctx, &total, 1));
    const size_t n0 = out.n;
    size_t want = n0 + (size_t)((double)W * *per_item) + ctx->append_slack;
    for (int attempt = 0; ; attempt++) {
        if (want > out.cap) { size_t keep = out.n; uint64_t *np = nullptr; TRY(dalloc(ctx, &np, want));
            if (keep) HIPCHK(hipMemcpyAsync(np, out.p, keep * 8, hipMemcpyDeviceToDevice, ctx->stream));
            HIPCHK(stream_wait(ctx)); dfree(out.p); out.p = np; out.cap = want; }
        unsigned long long init = n0; TRY(h2d(ctx, total, &init, 1));
        reset();
        appender ap{out.p, out.cap, total, ctx->pool_cap < POOL_N ? (ctx->pool_cap ? ctx->pool_cap : 1u) : POOL_N};
        // the launches alone between two events of their own (ctx->append_ms, summed over the attempts): a timer around the whole
        // pass also holds the host round trips before and after them, and in a run whose writer threads use up the process's CPU
        // quota the thread that feeds the GPU can be held there for tens of milliseconds (profiles/r3cd_cfg5_builds_...json)
        hipEvent_t ea = nullptr, eb = nullptr;
        bool timed = hipEventCreate(&ea) == hipSuccess;
        if (timed && hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); timed = false; }
        if (timed) (void)hipEventRecord(ea, ctx->stream);
        for (uint64_t w0 = 0; w0 < units; w0 += chunk) {
            uint64_t nw = units - w0 < chunk ? units - w0 : chunk;
            launch(w0, nw, ap);
        }
        if (timed) (void)hipEventRecord(eb, ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        unsigned long long got = 0; const int rc_got = launch_err == hipSuccess ? d2h(ctx, &got, total, 1) : CGX_ERR_HIP;      // d2h waits for the stream: both events have happened
        if (timed) { float ms = 0.0f; if (rc_got == CGX_OK && hipEventElapsedTime(&ms, ea, eb) == hipSuccess) ctx->append_ms += (double)ms; (void)hipEventDestroy(ea); (void)hipEventDestroy(eb); }
        if (launch_err != hipSuccess) { dfree(total); return fail(ctx, CGX_ERR_HIP, "append pass launch", launch_err); }
        if (rc_got != CGX_OK) { dfree(total); return rc_got; }
        if (got <= out.cap) { out.n = (size_t)got; break; }
        if (attempt) { snprintf(ctx->err, sizeof ctx->err, "append pass overflowed twice"); dfree(total); return CGX_ERR_STATE; }
        want = (size_t)got; ctx->ms["append_reruns"] += 1.0;   // the capacity guessed from the previous batch was too small: the launch runs once more with the exact size
    }
    if (W) { double r = (double)(out.n - n0) / (double)W * 1.25 + 0.05; if (r > *per_item || r < *per_item * 0.5) *per_item = r; }
    dfree(total);
    return CGX_OK;
}

template <class Launch>
static int append_pass(cgx_ctx *ctx, uint64_t units, uint64_t chunk, uint64_t W, dvec64 &out, double *per_item, Launch launch) {
    return append_pass(ctx, units, chunk, W, out, per_item, launch, [] {});
}

__global__ void k_iota(uint32_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (uint32_t)i; }
template <class T> __global__ void k_gather(const T *src, const uint32_t *perm, T *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) dst[i] = src[perm[i]];
}
template <class T> __global__ void k_head_flags(const T *keys, uint32_t *flags, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// ---- sorting inside short id runs --------------------------------------------------------------------------------------
// The rule and lexicon keys of a batch are "id-major with short runs": work items are laid out unit by unit (block by block,
// pattern by pattern), so the array is already ordered by its id (`major`, non-decreasing) and one id has at most RS_MAXRUN
// records (a unit has at most 300 / 65 / 70 sampled occurrences).  Only the order INSIDE each id run is missing, and the id
// need not even be part of the key.  A device-wide radix sort spends 7-10 passes over the whole array on that (round 2: 24 ms
// per batch for the rule keys, 16 ms for the lexicon keys).  Here, two kernels, one read and one write of the array:
//   k_runsort_block   a block takes RS_BLOCK consecutive records into LDS; every record counts the records of ITS run inside the block
//                     that sort before it (ties by position: stable) and goes to that rank.  Runs that lie inside one block --
//                     nearly all of them -- are done; the work per record is its run's length.  (1024 records per block: 256 made the
//                     first kernel 15 % faster and the second twice as slow, 15.6 ms per batch in all against 13.7.)
//   k_runsort_fix     one wave per block boundary that cuts a run: the run's pieces (each sorted by the first kernel) are loaded
//                     into LDS and ranked against each other.
// (Measured and dropped on the way, profiles/r3l, r3m: a bitonic network over 2048-record tiles with halos -- 66 barrier stages,
// 68 us per tile; a counting rank sort over such tiles -- per-thread tails on 300-record runs; rocPRIM's segmented radix sort --
// 5 ms per 6.7e7 keys in 1e7 segments.  All three were slower than the radix sort they were to replace.)
#define RS_BLOCK 1024
#define RS_MAXRUN 320
// TAG: the keys are narrower than RS_TAGBITS bits (or all ones, the extraction's "no rule"): a record's position in the block goes into the
// ten bits below its key, which makes the keys of a block distinct and turns "sorts before me, ties by position" into ONE 64-bit comparison
// per pair instead of three -- the ranking loop is this kernel's time (a record of a 300-record run makes 300 comparisons).
#define RS_TAGBITS 53
template <bool VAL, bool TAG>
__global__ __launch_bounds__(RS_BLOCK) void k_runsort_block(const uint32_t *__restrict__ major, const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, uint32_t n) {
    static_assert(RS_BLOCK == 1024, "the position tag is ten bits");
    __shared__ uint64_t sk[RS_BLOCK + 8]; __shared__ unsigned long long heads[RS_BLOCK / 64];
    __shared__ uint64_t sk2[TAG ? RS_BLOCK : 1];                  // TAG: every wave's share of every run, sorted (below)
    const uint32_t base = blockIdx.x * RS_BLOCK, i = base + threadIdx.x;
    const bool valid = i < n;
    const uint64_t k = valid ? kin[i] : ~0ull; const uint32_t m = valid ? major[i] : 0xFFFFFFFFu;
    uint32_t v = 0; if (VAL && valid) v = vin[i];
    // run heads: a record whose id differs from its left neighbour's (the first record of the block always).  One ballot per
    // wave gives every lane the bounds of its run without walking: the nearest head at or before it, the next head after it.
    const uint32_t mprev = (threadIdx.x == 0 || i == 0 || !valid) ? ~m : major[i - 1];       // i - 1 is in this block for threadIdx.x > 0: an L1 hit
    const unsigned long long hmask = __ballot(threadIdx.x == 0 || !valid || mprev != m);
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (lane == 0) heads[wave] = hmask;
    const uint64_t kt = TAG ? (k << 10) | (uint64_t)threadIdx.x : k;
    sk[threadIdx.x] = kt;
    __syncthreads();
    int lo = 0, hi = -1;
    if (valid) {
        // lo: highest head bit at or below this position, searching this wave's mask, then the waves to the left
        { int w = wave; unsigned long long hm = heads[w] & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
          while (hm == 0) { w--; hm = heads[w]; }              // wave 0 has bit 0 set: terminates
          lo = w * 64 + 63 - __clzll((long long)hm); }
        // hi: one before the next head above this position (or the end of the block / of the valid records)
        { int w = wave; unsigned long long hm = lane == 63 ? 0ull : (heads[w] >> (lane + 1)) << (lane + 1);
          while (hm == 0 && w + 1 < RS_BLOCK / 64) { w++; hm = heads[w]; }
          hi = hm ? w * 64 + __ffsll((long long)hm) - 2 : RS_BLOCK - 1; }
    }
    uint32_t rank = 0; const int me = (int)threadIdx.x;
    if (TAG) {
        // A run of 300 records lies in five or six waves.  Every record first ranks itself among the records of its run IN ITS OWN WAVE (at most
        // 64 comparisons) and goes to that place of a second copy, which so holds every wave's share of every run in order; its rank among the
        // shares of the other waves is then a bisection each (six steps): ~90 comparisons for a record of a 300-record run instead of 300, the
        // same as before for a run that lies in one wave.
        const int w0 = wave << 6, wlo = lo > w0 ? lo : w0, whi = hi < w0 + 63 ? hi : w0 + 63;
        if (valid) {
            int x = wlo;
            for (; x + 3 <= whi; x += 4) { const uint64_t o0 = sk[x], o1 = sk[x + 1], o2 = sk[x + 2], o3 = sk[x + 3]; rank += (uint32_t)(o0 < kt) + (uint32_t)(o1 < kt) + (uint32_t)(o2 < kt) + (uint32_t)(o3 < kt); }
            for (; x <= whi; x++) rank += (uint32_t)(sk[x] < kt);
            sk2[wlo + (int)rank] = kt;
        }
        __syncthreads();
        if (valid) {
            for (int w = lo >> 6; w <= (hi >> 6); w++) {
                if (w == wave) continue;
                const int plo = lo > (w << 6) ? lo : (w << 6), phi = hi < (w << 6) + 63 ? hi : (w << 6) + 63;
                int a = plo, b = phi + 1;
                while (a < b) { const int mid = (a + b) >> 1; if (sk2[mid] < kt) a = mid + 1; else b = mid; }
                rank += (uint32_t)(a - plo);
            }
        }
    } else if (valid) {
        int x = lo;
        for (; x + 3 <= hi; x += 4) {                                        // four independent LDS reads in flight
            const uint64_t o0 = sk[x], o1 = sk[x + 1], o2 = sk[x + 2], o3 = sk[x + 3];
            rank += (o0 < k || (o0 == k && x < me)) + (o1 < k || (o1 == k && x + 1 < me)) + (o2 < k || (o2 == k && x + 2 < me)) + (o3 < k || (o3 == k && x + 3 < me));
        }
        for (; x <= hi; x++) { const uint64_t o = sk[x]; rank += (o < k || (o == k && x < me)) ? 1u : 0u; }
    }
    if (valid) { kout[base + (uint32_t)lo + rank] = k; if (VAL) vout[base + (uint32_t)lo + rank] = v; }
}
template <bool VAL>
__global__ __launch_bounds__(64) void k_runsort_fix(const uint32_t *__restrict__ major, const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                     uint64_t *__restrict__ key, uint32_t *__restrict__ val, uint32_t n, unsigned int *__restrict__ nlong) {
    __shared__ uint64_t sk[RS_MAXRUN]; __shared__ uint32_t sv[RS_MAXRUN];
    const uint32_t p = (blockIdx.x + 1) * RS_BLOCK;            // the boundary between record p-1 and record p
    if (p >= n) return;
    const uint32_t m = major[p];
    if (major[p - 1] != m) return;                               // no run is cut here
    const int lane = (int)threadIdx.x;
    // run start: the first record of the id (wave-parallel: every lane looks at one record per round; a short run ends the walk
    // in the first rounds, a long one is followed to its end -- no length is assumed)
    uint32_t rs = p;
    for (uint32_t back = 0; ; back += 64) {
        const bool same = p >= 1 + back + (uint32_t)lane && major[p - 1 - back - (uint32_t)lane] == m;
        const unsigned long long b = __ballot(same);
        const int run = b == ~0ull ? 64 : __ffsll((long long)~b) - 1;   // lanes 0..run-1 still belong to the id
        rs = p - back - (uint32_t)run;
        if (run < 64) break;
    }
    if ((rs / RS_BLOCK + 1) * RS_BLOCK != p) return;             // a run cut by several boundaries is handled at the first of them
    uint32_t re = p;                                             // one past the last record of the id
    for (uint32_t fwd = 0; ; fwd += 64) {
        const uint32_t q = p + fwd + (uint32_t)lane; const bool same = q < n && major[q] == m;
        const unsigned long long b = __ballot(same);
        const int run = b == ~0ull ? 64 : __ffsll((long long)~b) - 1;
        re = p + fwd + (uint32_t)run;
        if (run < 64) break;
    }
    const uint32_t len = re - rs;
    if (len > RS_MAXRUN) {
        // Not a short run (no caller produces one today: static_assert below).  Still sorted, only slowly: the block pass permuted
        // the records of the run inside each block, so the INPUT arrays hold the same records between rs and re and nobody writes
        // them; every record is ranked against the whole run there (ties by input position: stable) and written to its place.
        if (lane == 0) atomicAdd(nlong, 1u);
        for (uint32_t j = rs + (uint32_t)lane; j < re; j += 64) {
            const uint64_t k = kin[j]; uint32_t rank = 0;
            for (uint32_t x = rs; x < re; x++) { const uint64_t o = kin[x]; rank += (o < k || (o == k && x < j)) ? 1u : 0u; }
            key[rs + rank] = k; if (VAL) val[rs + rank] = vin[j];
        }
        return;
    }
    for (uint32_t j = (uint32_t)lane; j < len; j += 64) { sk[j] = key[rs + j]; if (VAL) sv[j] = val[rs + j]; }
    __syncthreads();
    // The block pass has sorted the run's records on either side of the boundary (stably): two sorted pieces, A = [0, na) and B = [na, len),
    // and a stable merge puts a record of A behind the records of B that are smaller, a record of B behind those of A that are not
    // greater -- a bisection each instead of a comparison with every record of the run.
    const uint32_t na = p - rs;
    for (uint32_t j = (uint32_t)lane; j < len; j += 64) {
        const uint64_t k = sk[j]; uint32_t lo, hi, rank;
        if (j < na) { lo = na; hi = len; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sk[mid] < k) lo = mid + 1; else hi = mid; } rank = j + (lo - na); }
        else { lo = 0; hi = na; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sk[mid] <= k) lo = mid + 1; else hi = mid; } rank = (j - na) + lo; }
        key[rs + rank] = k; if (VAL) val[rs + rank] = sv[j];
    }
}
// The fast path of the fix pass holds a run in LDS: every caller's runs fit (an id has at most as many records as a unit has samples)
static_assert(CGX_SAMPLER <= RS_MAXRUN && CGX_SAMPLER_ONEGAP <= RS_MAXRUN && CGX_SAMPLER_TWOGAP <= RS_MAXRUN, "run_sort: a unit's samples must fit the fix pass's LDS run");
// keys (and an optional payload) sorted inside every run of equal `major`, whatever the run lengths (runs of more than RS_MAXRUN
// records that cross a block boundary take the slow path of the fix pass and are counted in ctx->d_rs_long, "run_sort_long_runs");
// kout/vout must not alias the inputs
// keybits: the callers' promise that every key is below 2^keybits or all ones (64: no promise)
static int run_sort(cgx_ctx *ctx, const uint32_t *major, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, size_t n, unsigned keybits = 64) {
    if (n == 0) return CGX_OK;
    if (n > 0xFFFFFF00ull) return fail(ctx, CGX_ERR_NOMEM, "run_sort: too many records", hipSuccess);
    if ((const void *)kin == (const void *)kout || (vin && vin == vout)) return fail(ctx, CGX_ERR_ARG, "run_sort: output aliases input", hipSuccess);
    if (!ctx->d_rs_long) { TRY(dalloc(ctx, &ctx->d_rs_long, 1)); HIPCHK(hipMemsetAsync(ctx->d_rs_long, 0, 4, ctx->stream)); }
    const unsigned nb = nblocks(n, RS_BLOCK);
    const bool tag = keybits <= RS_TAGBITS;
    if (vin) { if (tag) k_runsort_block<true, true><<<nb, RS_BLOCK, 0, ctx->stream>>>(major, kin, vin, kout, vout, (uint32_t)n); else k_runsort_block<true, false><<<nb, RS_BLOCK, 0, ctx->stream>>>(major, kin, vin, kout, vout, (uint32_t)n); }
    else { if (tag) k_runsort_block<false, true><<<nb, RS_BLOCK, 0, ctx->stream>>>(major, kin, nullptr, kout, nullptr, (uint32_t)n); else k_runsort_block<false, false><<<nb, RS_BLOCK, 0, ctx->stream>>>(major, kin, nullptr, kout, nullptr, (uint32_t)n); }
    if (nb > 1) {
        if (vin) k_runsort_fix<true><<<nb - 1, 64, 0, ctx->stream>>>(major, kin, vin, kout, vout, (uint32_t)n, ctx->d_rs_long);
        else k_runsort_fix<false><<<nb - 1, 64, 0, ctx->stream>>>(major, kin, nullptr, kout, nullptr, (uint32_t)n, ctx->d_rs_long);
    }
    HIPCHK(hipGetLastError());
    return CGX_OK;
}
// test hook (tests/test_gpu_parity.py): host arrays through run_sort; returns the number of long runs the fix pass met
extern "C" int64_t cgx__test_run_sort_bits(cgx_ctx *ctx, const uint32_t *major, const uint64_t *key, const uint32_t *val, uint64_t *key_out, uint32_t *val_out, int64_t n, int keybits);
extern "C" int64_t cgx__test_run_sort(cgx_ctx *ctx, const uint32_t *major, const uint64_t *key, const uint32_t *val, uint64_t *key_out, uint32_t *val_out, int64_t n) {
    return cgx__test_run_sort_bits(ctx, major, key, val, key_out, val_out, n, 64);
}
extern "C" int64_t cgx__test_run_sort_bits(cgx_ctx *ctx, const uint32_t *major, const uint64_t *key, const uint32_t *val, uint64_t *key_out, uint32_t *val_out, int64_t n, int keybits) {
    if (!ctx || n < 0 || (n && (!major || !key || !key_out)) || (val && !val_out) || keybits < 1 || keybits > 64) return CGX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device)); stage_enter(ctx);
    uint32_t *dm = nullptr, *dv = nullptr, *dvo = nullptr; uint64_t *dk = nullptr, *dko = nullptr;
    TRY(dalloc(ctx, &dm, (size_t)n)); TRY(dalloc(ctx, &dk, (size_t)n)); TRY(dalloc(ctx, &dko, (size_t)n));
    if (val) { TRY(dalloc(ctx, &dv, (size_t)n)); TRY(dalloc(ctx, &dvo, (size_t)n)); }
    unsigned int before = 0, after = 0;
    if (n) {
        TRY(h2d(ctx, dm, major, (size_t)n)); TRY(h2d(ctx, dk, key, (size_t)n)); if (val) TRY(h2d(ctx, dv, val, (size_t)n));
        if (ctx->d_rs_long) TRY(d2h(ctx, &before, ctx->d_rs_long, 1));
        TRY(run_sort(ctx, dm, dk, dv, dko, dvo, (size_t)n, (unsigned)keybits));
        TRY(d2h(ctx, key_out, dko, (size_t)n)); if (val) TRY(d2h(ctx, val_out, dvo, (size_t)n));
        TRY(d2h(ctx, &after, ctx->d_rs_long, 1));
    }
    dfree(dm); dfree(dk); dfree(dko); dfree(dv); dfree(dvo);
    return (int64_t)(after - before);
}

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
extern "C" cgx_ctx *cgx_create(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "cgx: no HIP device %d available (found %d); there is no CPU fallback\n", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    cgx_ctx *c = new cgx_ctx();
    c->device = device;
    {   // compute stream at the highest priority: the text copies of the previous batch run beside it on side streams.
        // CGX_COMPUTE_PRIORITY=normal: default priority (streams of ONE priority share few hardware queues: two contexts that are to work on
        // the card at the same time must not both sit in the high-priority queue)
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char *pr = getenv("CGX_COMPUTE_PRIORITY");
        if (pr && !strcmp(pr, "normal")) { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; } }
        else if (hipStreamCreateWithPriority(&c->stream, hipStreamDefault, hi) != hipSuccess) { (void)hipGetLastError(); if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return nullptr; } }
    }
    if (hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { (void)hipGetLastError(); c->sync_ev = nullptr; }
    if (hipHostMalloc(&c->h_small, CGX_SMALL_COPY, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->h_small = nullptr; }
    return c;
}
static void free_batch(cgx_ctx *c) {
    dfree(c->d_qoff); dfree(c->d_qtok); dfree(c->d_tok2q); dfree(c->d_lm); dfree(c->d_up); dfree(c->d_down);
    dfree(c->d_g1); dfree(c->d_p1); dfree(c->d_pid1); dfree(c->d_s1); dfree(c->d_hits1); dfree(c->d_hk1);
    dfree(c->d_g2); dfree(c->d_c2); dfree(c->d_pid2); dfree(c->d_s2); dfree(c->d_hits2); dfree(c->d_hk2); dfree(c->d_hid2); dfree(c->d_p1d); dfree(c->d_c2d); dfree(c->d_one2);
    dfree(c->d_qb_off); dfree(c->d_qb_ids); dfree(c->d_qo_off); dfree(c->d_qo_ids); dfree(c->d_qt_off); dfree(c->d_qt_ids); c->nqo = c->nqt = 0;
    dfree(c->d_blocks); dfree(c->d_r0); dfree(c->d_r1); dfree(c->d_r2); dfree(c->d_lex0); dfree(c->d_lex1); dfree(c->d_lex2); dfree(c->d_rng0); dfree(c->d_rng1); dfree(c->d_rng2); c->nl0 = c->nl1 = c->nl2 = 0;
    c->e1 = c->d1 = c->h1 = c->e2 = c->d2 = c->h2 = c->g = c->n0 = c->n1 = c->n2 = c->sep1 = c->sep2a = c->sep2b = 0;
    c->guard_exits = 0;
}
// the index members of a context, as one list: what free_index gives back and what cgx_share_index lends
#define CGX_INDEX_POINTERS(X) \
    X(d_win) X(d_str) X(d_sa) X(d_rlp) X(d_tstr) X(d_ltar) X(d_rtar) X(d_ltar16) X(d_rtar16) X(d_tok8) X(d_lr16) X(d_lrs) X(d_pos1) \
    X(d_lexkey) X(d_lexv1) X(d_lexv2) X(d_lexn1) X(d_lexn2) X(d_lexrow) X(d_lexnullt) X(d_lexhkey) X(d_lexhidx) X(d_lexslot) X(d_lexnullv) X(d_lexpbits) \
    X(d_tokstart) X(d_tokrank) X(d_freq) X(d_pidx) X(d_miss) X(d_phit_start) X(d_phit_len) X(d_ng[0]) X(d_ng[1]) X(d_ng[2]) X(d_ng[3])
static void free_index(cgx_ctx *c) {
    if (c->index_borrowed) {                                    // another context's arrays: forget them
#define X(m) c->m = nullptr;
        CGX_INDEX_POINTERS(X)
#undef X
        c->index_borrowed = false;
    }
    dfree(c->d_win); dfree(c->d_str); dfree(c->d_sa); dfree(c->d_rlp); dfree(c->d_tstr); dfree(c->d_ltar); dfree(c->d_rtar); dfree(c->d_ltar16); dfree(c->d_rtar16); c->long_pos = false; dfree(c->d_tok8); dfree(c->d_lr16); dfree(c->d_lrs); c->lrs_k = 0; dfree(c->d_pos1);
    dfree(c->d_lexkey); dfree(c->d_lexv1); dfree(c->d_lexv2); dfree(c->d_lexn1); dfree(c->d_lexn2); dfree(c->d_lexrow); dfree(c->d_lexnullt); dfree(c->d_lexhkey); dfree(c->d_lexhidx); dfree(c->d_lexslot); dfree(c->d_lexnullv); dfree(c->d_lexpbits); c->lex_hmask = 0;
    dfree(c->d_tokstart); dfree(c->d_tokrank); dfree(c->d_freq); dfree(c->d_pidx); dfree(c->d_miss);
    dfree(c->d_phit_start); dfree(c->d_phit_len); for (int k = 0; k < 4; k++) { dfree(c->d_ng[k]); c->ng_cap[k] = 0; }
    c->n = c->nt = c->nlex = c->nphits = 0; c->have_sa = c->have_pre = false;
}
// Two contexts of one device over ONE index (include/cgx.h): dst borrows every index array of src and the scalars that describe them.
extern "C" int cgx_share_index(cgx_ctx *dst, const cgx_ctx *src) {
    if (!dst || !src || dst == src) return CGX_ERR_ARG;
    cgx_ctx *ctx = dst;
    if (dst->device != src->device) return fail(ctx, CGX_ERR_ARG, "cgx_share_index: the two contexts are on different devices", hipSuccess);
    if (!src->d_str || !src->have_sa || !src->have_pre) return fail(ctx, CGX_ERR_STATE, "cgx_share_index: the lending context has no complete index", hipSuccess);
    HIPCHK(hipSetDevice(dst->device)); stage_enter(dst);
    HIPCHK(hipStreamSynchronize(src->stream));                   // whatever built the index has finished
    free_batch(dst); free_index(dst);
#define X(m) dst->m = src->m;
    CGX_INDEX_POINTERS(X)
#undef X
    dst->n = src->n; dst->nt = src->nt; dst->nlex = src->nlex; dst->nphits = src->nphits; dst->last = src->last; dst->have_sa = src->have_sa; dst->have_pre = src->have_pre;
    dst->long_pos = src->long_pos; dst->lrs_k = src->lrs_k; dst->lex_nrow = src->lex_nrow; dst->lex_ntgt = src->lex_ntgt; dst->lex_hmask = src->lex_hmask; dst->lex_hshift = src->lex_hshift; dst->lex_pshift = src->lex_pshift;
    for (int k = 0; k < 4; k++) { dst->ng_cap[k] = src->ng_cap[k]; dst->ng_shift[k] = src->ng_shift[k]; }
    memcpy(dst->freq, src->freq, sizeof dst->freq);
    for (const char *k : {"src_blocks_factor", "win_table_gb", "ngram_table_bytes"}) { auto it = src->ms.find(k); if (it != src->ms.end()) dst->ms[k] = it->second; }
    dst->index_borrowed = true;
    return CGX_OK;
}
extern "C" void cgx_destroy(cgx_ctx *c) {
    if (!c) return;
    cgx__host_release(c);
    (void)hipSetDevice(c->device);
    free_batch(c); free_index(c);
    for (int a = 0; a < 2; a++) if (c->arena[a]) { (void)hipHostFree(c->arena[a]); c->arena[a] = nullptr; }
    for (int a = 0; a < 2; a++) { dfree(c->d_text[a]); dfree(c->d_qtext[a]); dfree(c->d_seg_off[a]); dfree(c->d_seg_len[a]); dfree(c->d_qseg[a]); dfree(c->d_trl[a]); }
    dfree(c->d_spool); dfree(c->d_soff); dfree(c->d_tpool); dfree(c->d_toff); dfree(c->d_aa); dfree(c->d_bb); dfree(c->d_fs); dfree(c->d_gztab); dfree(c->d_gzcode); dfree(c->d_rs_long);
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    if (c->h_small) (void)hipHostFree(c->h_small);
    for (int r = 0; r < CGX_COPY_STREAMS; r++) if (c->copy_streams[r]) (void)hipStreamDestroy(c->copy_streams[r]);
    for (int r = 0; r < CGX_MAX_READERS; r++) if (c->copy_done[r]) (void)hipEventDestroy(c->copy_done[r]);
    (void)hipStreamSynchronize(c->stream);
    stage_enter(c);
    { std::lock_guard<std::mutex> g(g_pool_lock); pool_of(c->device).forget(c->stream); }
    pool_trim(c->device);
    (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" const char *cgx_last_error(cgx_ctx *c) { return c ? c->err : "null context"; }
extern "C" int cgx_set_option(cgx_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return CGX_ERR_ARG;
    if (!strcmp(name, "k1_limit")) { c->k1_limit = value <= 0 || value > 0x7FFFFFFF ? 0x7FFFFFFF : (int)value; return CGX_OK; }   /* 0 = no limit; the reference's K1 stops at 128 tokens per query */
    if (!strcmp(name, "device_format")) { c->device_format = value != 0; return CGX_OK; }
    if (!strcmp(name, "use_bigrams")) { c->ngram_max = value != 0 ? 5 : 1; return CGX_OK; }       /* round-1 name: 0 = every l >= 2 by binary search */
    if (!strcmp(name, "ngram_tables")) { if (value < 1 || value > 5) return CGX_ERR_ARG; c->ngram_max = (int)value; return CGX_OK; }
    if (!strcmp(name, "gz_level")) { if (value < 0 || value > 9) return CGX_ERR_ARG; c->gz_level = (int)value; return CGX_OK; }
    if (!strcmp(name, "gz_device")) { c->gz_device = value != 0; return CGX_OK; }
    if (!strcmp(name, "gz_dynamic")) { c->gz_dynamic = value != 0; return CGX_OK; }
    if (!strcmp(name, "use_layouts")) { c->use_layouts = value != 0; return CGX_OK; }
    if (!strcmp(name, "occ_order")) { c->occ_order = value != 0; return CGX_OK; }
    if (!strcmp(name, "src_blocks")) { c->src_blocks = value != 0; return CGX_OK; }
    if (!strcmp(name, "count_probes")) { c->count_probes = value != 0; return CGX_OK; }
    if (!strcmp(name, "numa_pin")) { c->numa_pin = value != 0; return CGX_OK; }
    if (!strcmp(name, "prealloc_text")) { c->prealloc_text = value != 0; return CGX_OK; }
    if (!strcmp(name, "use_lex_hash")) { c->use_lex_hash = value != 0; return CGX_OK; }
    if (!strcmp(name, "wide_hits2")) { c->wide_hits2 = value != 0; return CGX_OK; }
    if (!strcmp(name, "hit_order")) { c->hit_order = value != 0; return CGX_OK; }
    if (!strcmp(name, "tile_order")) { c->tile_order = value != 0; return CGX_OK; }
    if (!strcmp(name, "lex_flat")) { if (value < 0 || value > 2) return CGX_ERR_ARG; c->lex_flat = (int)value; return CGX_OK; }
    if (!strcmp(name, "lex_bits")) { c->lex_bits = value != 0; return CGX_OK; }
    if (!strcmp(name, "win_table")) { c->win_table = value != 0; return CGX_OK; }       /* before the index is built or loaded: 1 = build the window table (cgx_view::win; 128 bytes per corpus position) */
    if (!strcmp(name, "write_period")) { if (value < 0) return CGX_ERR_ARG; c->write_period = value; return CGX_OK; }
    if (!strcmp(name, "write_count")) { if (value < 0) return CGX_ERR_ARG; c->write_count = value; return CGX_OK; }
    if (!strcmp(name, "fault_inject")) { c->fault_inject = value; return CGX_OK; }
    if (!strcmp(name, "pool_cap")) { if (value < 1) return CGX_ERR_ARG; c->pool_cap = (uint32_t)(value > POOL_N ? POOL_N : value); return CGX_OK; }
    if (!strcmp(name, "look_rec_cap")) { if (value < 0) return CGX_ERR_ARG; c->look_rec_cap = (uint32_t)(value > 65535 ? 65535 : value); return CGX_OK; }
    if (!strcmp(name, "sub_batch")) { if (value < 0) return CGX_ERR_ARG; c->sub_batch = value; return CGX_OK; }
    if (!strcmp(name, "auto_batch_tokens")) { if (value < 1) return CGX_ERR_ARG; c->auto_batch_tokens = value; return CGX_OK; }
    if (!strcmp(name, "append_slack")) { if (value < 0) return CGX_ERR_ARG; c->append_slack = (uint64_t)value; return CGX_OK; }
    if (!strcmp(name, "append_guess_milli")) { if (value < 0) return CGX_ERR_ARG; c->look1_per_item = c->look2_per_item = (double)value / 1000.0; return CGX_OK; }
    if (!strcmp(name, "async_write")) { c->async_write = value != 0; return CGX_OK; }
    if (!strcmp(name, "lex_hash_bits")) { if (value < 0 || value > 63) return CGX_ERR_ARG; c->lex_hash_bits = (unsigned)value; return CGX_OK; }
    if (!strcmp(name, "force_host_lexicon")) { c->force_host_lexicon = value != 0; return CGX_OK; }
    if (!strcmp(name, "chunk_items")) { if (value < 1024) return CGX_ERR_ARG; c->chunk_items = (uint64_t)value; return CGX_OK; }
    snprintf(c->err, sizeof c->err, "unknown option %s", name);
    return CGX_ERR_ARG;
}
extern "C" int64_t cgx__option(cgx_ctx *c, const char *name) {
    if (!c || !name) return 0;
    if (!strcmp(name, "async_write")) return (int64_t)c->async_write;
    if (!strcmp(name, "device_format")) return (int64_t)c->device_format;
    if (!strcmp(name, "sub_batch")) return c->sub_batch;
    if (!strcmp(name, "auto_batch_tokens")) return c->auto_batch_tokens;
    if (!strcmp(name, "numa_pin")) return (int64_t)c->numa_pin;
    if (!strcmp(name, "gz_level")) return (int64_t)c->gz_level;
    if (!strcmp(name, "gz_device")) return (int64_t)c->gz_device;
    if (!strcmp(name, "write_period")) return c->write_period;
    if (!strcmp(name, "write_count")) return c->write_count;
    return 0;
}
// "local_cpulist" of the GPU's PCI function (the CPUs of its NUMA node), or an empty string
extern "C" void cgx__device_cpulist(cgx_ctx *c, char *buf, size_t cap) {
    if (!buf || !cap) return;
    buf[0] = 0;
    if (!c) return;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, c->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return;
    if (!fgets(buf, (int)cap, f)) buf[0] = 0;
    fclose(f);
}
extern "C" void cgx__bind_thread(cgx_ctx *c) { if (c) (void)hipSetDevice(c->device); }
extern "C" const void *cgx__get_vocab_owner(cgx_ctx *c) { return c ? c->vocab_owner : nullptr; }
extern "C" void cgx__set_vocab_owner(cgx_ctx *c, const void *p) { if (c) c->vocab_owner = p; }
extern "C" void cgx__set_host_state(cgx_ctx *c, void *p) { if (c) c->host_state = p; }
extern "C" void *cgx__get_host_state(cgx_ctx *c) { return c ? c->host_state : nullptr; }
extern "C" void cgx__set_host_ms(cgx_ctx *c, const char *name, double ms) { if (c && name) c->host_ms[name] = ms; }
extern "C" double cgx_host_ms(cgx_ctx *c, const char *name) {
    if (!c || !name) return -1;
    auto it = c->host_ms.find(name);
    return it == c->host_ms.end() ? -1.0 : it->second;
}
// Device memory by purpose, bytes ("mem_*" names of cgx_stage_ms): what the context holds right now, looked up block by block in
// the allocator.  index = what a replica receives; derived = tables rebuilt from it on every rank (interleaved layouts, target-side
// blocks, l-gram tables, pair hash); text = the two text slots and their piece lists; batch = results of the last batch kept for
// cgx_fetch and the formatter; cached = idle blocks the allocator keeps for the next batch (lookup outputs, sort buffers: the
// high-water mark of the temporaries); other = live blocks of other contexts on this device.
static void memory_report(cgx_ctx *c) {
    std::lock_guard<std::mutex> g(g_pool_lock);
    DevPool &P = pool_of(c->device);
    auto sz = [&](const void *p) -> double { if (!p) return 0.0; auto it = P.live.find((void *)p); return it == P.live.end() ? 0.0 : (double)it->second.bytes; };
    double index = 0, derived = 0, text = 0, batch = 0;
    for (const void *p : {(const void *)c->d_str, (const void *)c->d_sa, (const void *)c->d_rlp, (const void *)c->d_tstr, (const void *)c->d_ltar, (const void *)c->d_rtar, (const void *)c->d_ltar16, (const void *)c->d_rtar16,
                          (const void *)c->d_lexkey, (const void *)c->d_lexv1, (const void *)c->d_lexv2, (const void *)c->d_lexn1, (const void *)c->d_lexn2, (const void *)c->d_lexrow, (const void *)c->d_lexnullt,
                          (const void *)c->d_tokstart, (const void *)c->d_tokrank, (const void *)c->d_freq, (const void *)c->d_pidx, (const void *)c->d_miss, (const void *)c->d_phit_start, (const void *)c->d_phit_len,
                          (const void *)c->d_spool, (const void *)c->d_soff, (const void *)c->d_tpool, (const void *)c->d_toff, (const void *)c->d_aa, (const void *)c->d_bb, (const void *)c->d_fs, (const void *)c->d_gztab}) index += sz(p);
    double ngram = 0; for (int k = 0; k < 4; k++) ngram += sz(c->d_ng[k]);
    const double layouts = sz(c->d_tok8) + sz(c->d_lr16) + sz(c->d_lrs) + sz(c->d_pos1) + sz(c->d_win), lexhash = sz(c->d_lexslot) + sz(c->d_lexnullv) + sz(c->d_lexhkey) + sz(c->d_lexhidx) + sz(c->d_lexpbits);
    derived = ngram + layouts + lexhash;
    for (int a = 0; a < 2; a++) text += sz(c->d_text[a]) + sz(c->d_qtext[a]) + sz(c->d_seg_off[a]) + sz(c->d_seg_len[a]) + sz(c->d_qseg[a]) + sz(c->d_trl[a]);
    for (const void *p : {(const void *)c->d_qoff, (const void *)c->d_qtok, (const void *)c->d_tok2q, (const void *)c->d_lm, (const void *)c->d_up, (const void *)c->d_down,
                          (const void *)c->d_g1, (const void *)c->d_p1, (const void *)c->d_pid1, (const void *)c->d_s1, (const void *)c->d_hits1, (const void *)c->d_g2, (const void *)c->d_c2, (const void *)c->d_pid2, (const void *)c->d_s2, (const void *)c->d_hits2,
                          (const void *)c->d_hk1, (const void *)c->d_hk2, (const void *)c->d_hid2,
                          (const void *)c->d_p1d, (const void *)c->d_c2d, (const void *)c->d_one2, (const void *)c->d_blocks, (const void *)c->d_r0, (const void *)c->d_r1, (const void *)c->d_r2,
                          (const void *)c->d_rng0, (const void *)c->d_rng1, (const void *)c->d_rng2, (const void *)c->d_lex0, (const void *)c->d_lex1, (const void *)c->d_lex2,
                          (const void *)c->d_qb_off, (const void *)c->d_qb_ids, (const void *)c->d_qo_off, (const void *)c->d_qo_ids, (const void *)c->d_qt_off, (const void *)c->d_qt_ids}) batch += sz(p);
    double live = 0; for (auto &kv : P.live) live += (double)kv.second.bytes;
    c->ms["mem_index"] = index; c->ms["mem_derived"] = derived; c->ms["mem_derived_ngram_tables"] = ngram; c->ms["mem_derived_layouts"] = layouts; c->ms["mem_derived_lex_hash"] = lexhash;
    c->ms["mem_text"] = text; c->ms["mem_batch"] = batch; c->ms["mem_cached"] = (double)P.cached_bytes; c->ms["mem_other"] = live - index - derived - text - batch;
}
extern "C" double cgx_stage_ms(cgx_ctx *c, const char *name) {
    if (!c || !name) return -1;
    if (!strncmp(name, "mem_", 4)) memory_report(c);
    if (!strcmp(name, "run_sort_long_runs")) {                  // runs the fix pass of run_sort had to take the slow way (none with today's callers)
        unsigned int v = 0;
        if (c->d_rs_long && (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(&v, c->d_rs_long, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || stream_wait(c) != hipSuccess)) return -1.0;
        return (double)v;
    }
    auto it = c->ms.find(name);
    return it == c->ms.end() ? -1.0 : it->second;
}

// The stages live in .inc files that are part of this translation unit (shared static helpers above):
#include "cgx_index.inc"      // index upload, suffix array, frequent-pair precomputation, bigram table, replica plumbing
#include "cgx_search.inc"     // query upload, batched SA interval search, one-/two-gap enumeration and corpus lookups (cgx_sa_lookup, cgx_gappy_search)
#include "cgx_extract.inc"    // rule extraction (three launch families) and GenerateBlocks on the device (cgx_extract, cgx_make_blocks, cgx_set_blocks)
#include "cgx_lexicon.inc"    // lexical features, device lexicon, pinned result arenas (cgx_lex_features, cgx_lexicon)

// ------------------------------------------------------------------------------------
// result fetch
// ------------------------------------------------------------------------------------
extern "C" int64_t cgx_fetch(cgx_ctx *ctx, const char *name, void *dst, int64_t cap) {
    if (!ctx || !name) return CGX_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return CGX_ERR_HIP;
    const void *src = nullptr; int64_t bytes = -1;
    uint32_t counts[16] = { ctx->e1, ctx->d1, ctx->h1, ctx->e2, ctx->d2, ctx->h2, ctx->g, ctx->n0, ctx->n1, ctx->n2, ctx->sep1, ctx->sep2a, ctx->sep2b,
                            ctx->nphits, ctx->guard_exits, (uint32_t)ctx->last };
    std::string s(name);
    // the reference's hit records are made from the device's keys when somebody asks for them
    if (s == "hits1" && dst && !ctx->d_hits1 && ctx->h1 && ctx->d_hk1) {
        if (dalloc(ctx, &ctx->d_hits1, ctx->h1) != CGX_OK) return CGX_ERR_NOMEM;
        k_unpack_hits1<<<nblocks(ctx->h1, 256), 256, 0, ctx->stream>>>(hits1_view(ctx), ctx->h1, ctx->d_hits1);
    }
    if (s == "hits2" && dst && !ctx->d_hits2 && ctx->h2 && ctx->d_hk2) {
        if (dalloc(ctx, &ctx->d_hits2, ctx->h2) != CGX_OK) return CGX_ERR_NOMEM;
        k_unpack_hits2<<<nblocks(ctx->h2, 256), 256, 0, ctx->stream>>>(hits2_view(ctx), ctx->h2, ctx->d_hits2);
    }
#define ENT(nm, ptr, cnt, T) if (s == nm) { src = (ptr); bytes = (int64_t)(cnt) * (int64_t)sizeof(T); }
    ENT("sa", ctx->d_sa, ctx->n, int32_t) ENT("tokstart", ctx->d_tokstart, (size_t)ctx->last + 3, int32_t)
    ENT("freq", ctx->d_freq, CGX_TOP, int32_t) ENT("pidx", ctx->d_pidx, 2 * CGX_TOP * CGX_TOP, uint32_t) ENT("miss", ctx->d_miss, CGX_TOP * CGX_TOP, int32_t)
    ENT("phit_start", ctx->d_phit_start, ctx->nphits, uint32_t) ENT("phit_len", ctx->d_phit_len, ctx->nphits, uint8_t)
    ENT("lm", ctx->d_lm, ctx->ntok, int32_t) ENT("up", ctx->d_up, (size_t)ctx->ntok * 5, int32_t) ENT("down", ctx->d_down, (size_t)ctx->ntok * 5, int32_t)
    ENT("g1", ctx->d_g1, ctx->e1, cgx_gappy) ENT("p1", ctx->d_p1, ctx->e1, cgx_gappat) ENT("pid1", ctx->d_pid1, ctx->e1, uint32_t)
    ENT("s1", ctx->d_s1, ctx->d1, cgx_gapsearch) ENT("hits1", ctx->d_hits1, ctx->h1, cgx_hit1)
    ENT("g2", ctx->d_g2, ctx->e2, cgx_twogappy) ENT("c2", ctx->d_c2, ctx->e2, int32_t) ENT("pid2", ctx->d_pid2, ctx->e2, uint32_t)
    ENT("s2", ctx->d_s2, ctx->d2, cgx_twogapsearch) ENT("hits2", ctx->d_hits2, ctx->h2, cgx_hit2)
    ENT("p1d", ctx->d_p1d, ctx->d1, cgx_gappat) ENT("c2d", ctx->d_c2d, ctx->d2, int32_t) ENT("one2", ctx->d_one2, ctx->d2, uint32_t)
    ENT("rng0", ctx->d_rng0, ctx->d_rng0 ? 2 * (size_t)ctx->g : 0, int32_t) ENT("rng1", ctx->d_rng1, ctx->d_rng1 ? 2 * ((size_t)2 * ctx->g + ctx->d1) : 0, int32_t)
    ENT("rng2", ctx->d_rng2, ctx->d_rng2 ? 2 * ((size_t)ctx->g + 2 * (size_t)ctx->d1 + ctx->d2) : 0, int32_t)
    ENT("lex0", ctx->d_lex0, ctx->nl0, cgx_lexent) ENT("lex1", ctx->d_lex1, ctx->nl1, cgx_lexent) ENT("lex2", ctx->d_lex2, ctx->nl2, cgx_lexent)
    ENT("r0", ctx->d_r0, ctx->n0, cgx_rule0) ENT("r1", ctx->d_r1, ctx->n1, cgx_rule1) ENT("r2", ctx->d_r2, ctx->n2, cgx_rule2)
    ENT("blocks", ctx->d_blocks, ctx->g, cgx_block) ENT("qb_off", ctx->d_qb_off, ctx->d_qb_off ? (size_t)ctx->nq + 1 : 0, uint32_t) ENT("qb_ids", ctx->d_qb_ids, ctx->nqb, uint32_t)
    ENT("qo_off", ctx->d_qo_off, ctx->d_qo_off ? (size_t)ctx->nq + 1 : 0, uint32_t) ENT("qo_ids", ctx->d_qo_ids, ctx->nqo, uint32_t)      /* per-query pattern lists: after cgx_format */
    ENT("qt_off", ctx->d_qt_off, ctx->d_qt_off ? (size_t)ctx->nq + 1 : 0, uint32_t) ENT("qt_ids", ctx->d_qt_ids, ctx->nqt, uint32_t)
#undef ENT
    if (s == "counts") { bytes = sizeof counts; if (!dst) return bytes; if (cap < bytes) return CGX_ERR_ARG; memcpy(dst, counts, sizeof counts); return bytes; }
    if (bytes < 0) { snprintf(ctx->err, sizeof ctx->err, "unknown result %s", name); return CGX_ERR_ARG; }
    if (!dst) return bytes;
    if (cap < bytes) { snprintf(ctx->err, sizeof ctx->err, "buffer too small for %s", name); return CGX_ERR_ARG; }
    if (bytes && !src) { snprintf(ctx->err, sizeof ctx->err, "%s not computed yet", name); return CGX_ERR_STATE; }
    if (bytes) {
        if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return CGX_ERR_HIP;
        if (stream_wait(ctx) != hipSuccess) return CGX_ERR_HIP;
    }
    return bytes;
}
