// gather_coop -- does it matter WHO reads the 128 bytes of a random run?  gather_bw (the round-2 figure: 1.28e10 random 128-byte
// runs per second) lets every lane read its own run with eight consecutive 16-byte loads: one load instruction touches 64 different
// lines, and the seven later loads of a lane hit a line whose miss is still pending.  Here eight neighbouring lanes read one run
// together (lane k the k-th 16 bytes), eight runs per instruction, so that a line is asked for by ONE instruction.  Both kernels keep
// eight loads per lane in flight and read the same runs; `align8` places the runs at random 8-byte offsets (what a lookup window is).
//   hipcc --offload-arch=gfx950 -O3 gather_coop.cpp -o gather_coop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct __attribute__((aligned(8))) q16_a8 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint64_t run_of(uint64_t i, uint64_t nslot) { return ((i * 0x9E3779B97F4A7C15ull) >> 20) % nslot; }
// MODE 0: lane-private (eight 16-byte loads per lane, one run per lane); 1: eight lanes per run, eight runs per lane-octet in flight
template <int MODE>
__global__ __launch_bounds__(256) void k_runs(const char *__restrict__ buf, uint64_t nslot, uint64_t slot_bytes, uint64_t total, uint32_t *out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63; uint32_t acc = 0;
    for (uint64_t i0 = tid - lane; i0 < total; i0 += nthr) {          // a wave handles runs i0 .. i0+63 per trip
        if (MODE == 0) {
            const q16_a8 *p = (const q16_a8 *)(buf + run_of(i0 + lane, nslot) * slot_bytes);
#pragma unroll
            for (int k = 0; k < 8; k++) { q16_a8 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        } else {
            q16_a8 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = *((const q16_a8 *)(buf + run_of(i0 + 8 * j + (lane >> 3), nslot) * slot_bytes) + (lane & 7));
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE> static int run(const char *buf, uint64_t bytes, uint64_t slot_bytes, uint32_t *out, const char *what, int blocks_per_cu) {
    const uint64_t nslot = (bytes - 256) / slot_bytes, total = 1ull << 26;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    k_runs<MODE><<<256 * blocks_per_cu, 256>>>(buf, nslot, slot_bytes, total >> 4, out);
    CHK(hipEventRecord(a));
    k_runs<MODE><<<256 * blocks_per_cu, 256>>>(buf, nslot, slot_bytes, total, out);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    printf("{\"pattern\": \"%s\", \"run_bytes\": 128, \"run_offsets\": \"multiples of %llu bytes\", \"blocks_per_cu\": %d, \"runs\": %llu, \"ms\": %.3f, \"useful_GBps\": %.1f, \"runs_per_s\": %.3e}\n",
           what, (unsigned long long)slot_bytes, blocks_per_cu, (unsigned long long)total, ms, (double)total * 128.0 / ms / 1e6, (double)total / (ms * 1e-3));
    fflush(stdout);
    return 0;
}
int main() {
    const uint64_t bytes = 4ull << 30; char *buf; uint32_t *out;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 4)); CHK(hipMemset(buf, 1, bytes));
    for (int bpc = 4; bpc <= 16; bpc *= 2) {
        if (run<0>(buf, bytes, 128, out, "random, one run per lane (eight 16-byte loads each)", bpc)) return 1;
        if (run<1>(buf, bytes, 128, out, "random, eight lanes per run (one 16-byte load each), eight runs in flight per lane", bpc)) return 1;
        if (run<0>(buf, bytes, 8, out, "random, one run per lane (eight 16-byte loads each)", bpc)) return 1;
        if (run<1>(buf, bytes, 8, out, "random, eight lanes per run (one 16-byte load each), eight runs in flight per lane", bpc)) return 1;
    }
    return 0;
}
