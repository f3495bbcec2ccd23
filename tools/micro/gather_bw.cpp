// gather_bw -- what the window-walking kernels are bound by: the rate at which an MI355X serves RANDOM small reads from a
// table far larger than the Infinity Cache.  Every lane reads `run` bytes (16..256) at a random run-aligned offset of a
// 4 GiB buffer; reported: useful bytes/s and 64-byte sectors/s.  The streaming figure (same kernel, consecutive offsets)
// is printed beside it.  hipcc --offload-arch=gfx950 -O3 gather_bw.cpp -o gather_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int RUN16, bool RANDOM>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ buf, uint64_t nrun, uint64_t total, uint32_t *out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; uint32_t acc = 0;
    for (; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = RANDOM ? ((i * 0x9E3779B97F4A7C15ull) >> 20) % nrun : i % nrun;
        const uint4 *p = buf + r * RUN16;
#pragma unroll
        for (int k = 0; k < RUN16; k++) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the same reads with ONE read per thread (no grid-stride loop): how fast can the chip start and retire short-lived waves?
template <int RUN16, int LDS_BYTES>
__global__ __launch_bounds__(256) void k_gather_once(const uint4 *__restrict__ buf, uint64_t nrun, uint64_t total, uint32_t *out) {
    __shared__ uint32_t pad[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; uint32_t acc = 0;
    if (LDS_BYTES) pad[threadIdx.x] = (uint32_t)i;
    if (i < total) {
        uint64_t r = ((i * 0x9E3779B97F4A7C15ull) >> 20) % nrun;
        const uint4 *p = buf + r * RUN16;
#pragma unroll
        for (int k = 0; k < RUN16; k++) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (LDS_BYTES) acc ^= pad[(threadIdx.x + 1) & 255];
    if (acc == 0x12345678u) out[0] = acc;
}
template <int RUN16, int LDS_BYTES> static int run_once(const uint4 *buf, uint64_t bytes, uint32_t *out) {
    const uint64_t nrun = bytes / (16ull * RUN16), total = 1ull << 28;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    k_gather_once<RUN16, LDS_BYTES><<<(unsigned)(total / 256 / 16), 256>>>(buf, nrun, total / 16, out);
    CHK(hipEventRecord(a));
    k_gather_once<RUN16, LDS_BYTES><<<(unsigned)(total / 256), 256>>>(buf, nrun, total, out);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    printf("{\"pattern\": \"random, one read per thread (short-lived waves), %d bytes of LDS per block\", \"run_bytes\": %d, \"reads\": %llu, \"ms\": %.3f, \"sector64_GBps\": %.1f, \"reads_per_s\": %.3e, \"waves_per_s\": %.3e}\n",
           LDS_BYTES, 16 * RUN16, (unsigned long long)total, ms, (double)total * ((16 * RUN16 + 63) / 64) * 64.0 / ms / 1e6, (double)total / (ms * 1e-3), (double)total / 64 / (ms * 1e-3));
    return 0;
}
template <int RUN16, bool RANDOM> static int run(const uint4 *buf, uint64_t bytes, uint32_t *out, const char *what) {
    const uint64_t nrun = bytes / (16ull * RUN16), total = 1ull << 28 >> (RUN16 > 4 ? 2 : 0);
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    k_gather<RUN16, RANDOM><<<256 * 16, 256>>>(buf, nrun, total >> 4, out);       // warm-up
    CHK(hipEventRecord(a));
    k_gather<RUN16, RANDOM><<<256 * 16, 256>>>(buf, nrun, total, out);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    const double useful = (double)total * 16.0 * RUN16, sectors = (double)total * ((16 * RUN16 + 63) / 64);
    printf("{\"pattern\": \"%s\", \"run_bytes\": %d, \"reads\": %llu, \"ms\": %.3f, \"useful_GBps\": %.1f, \"sector64_GBps\": %.1f, \"reads_per_s\": %.3e}\n",
           what, 16 * RUN16, (unsigned long long)total, ms, useful / ms / 1e6, sectors * 64.0 / ms / 1e6, (double)total / (ms * 1e-3));
    return 0;
}
int main() {
    const uint64_t bytes = 4ull << 30; uint4 *buf; uint32_t *out;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 4)); CHK(hipMemset(buf, 1, bytes));
    if (run<1, true>(buf, bytes, out, "random") || run<4, true>(buf, bytes, out, "random") || run<8, true>(buf, bytes, out, "random") || run<16, true>(buf, bytes, out, "random") ||
        run<4, false>(buf, bytes, out, "sequential") || run<16, false>(buf, bytes, out, "sequential") ||
        run_once<4, 0>(buf, bytes, out) || run_once<4, 24576>(buf, bytes, out) || run_once<8, 24576>(buf, bytes, out)) return 1;
    return 0;
}
