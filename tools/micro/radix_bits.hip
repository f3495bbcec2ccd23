// radix_bits -- how fast does rocPRIM's onesweep radix sort order N 64-bit keys (keys only) with 8, 9 or 10 bits per pass?
// The hit sorts of a batch order 4.8e8 keys of 57-60 significant bits: 8 passes at 8 bits, 7 at 9, 6 at 10.
// usage: radix_bits [N] [bits]      prints one JSON line per configuration
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_fill(uint64_t *p, size_t n, unsigned bits) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567; x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29; x *= 0xff51afd7ed558ccdull; x ^= x >> 32;
    p[i] = bits >= 64 ? x : x & ((1ull << bits) - 1);
}
__global__ void k_check(const uint64_t *p, size_t n, unsigned long long *bad, unsigned long long *sum) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i >= n) return;
    if (i && p[i - 1] > p[i]) atomicAdd(bad, 1ull);
    if ((i & 1023) == 0) atomicAdd(sum, (unsigned long long)p[i]);
}
template <class Cfg> static void run(const char *name, const uint64_t *src, uint64_t *a, uint64_t *b, size_t n, unsigned bits) {
    size_t tb = 0; void *tmp = nullptr;
    CK(rocprim::radix_sort_keys<Cfg>(nullptr, tb, a, b, n, 0, bits, 0));
    CK(hipMalloc(&tmp, tb ? tb : 1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemcpy(a, src, n * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(e0, 0)); CK(rocprim::radix_sort_keys<Cfg>(tmp, tb, a, b, n, 0, bits, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long *d, h[2] = {0, 0}; CK(hipMalloc(&d, 16)); CK(hipMemset(d, 0, 16));
    k_check<<<(unsigned)((n + 255) / 256), 256>>>(b, n, d, d + 1); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("{\"config\": \"%s\", \"keys\": %zu, \"key_bits\": %u, \"ms\": %.3f, \"Gkeys_per_s\": %.2f, \"out_of_order\": %llu, \"checksum\": %llu}\n", name, n, bits, best, n / best / 1e6, h[0], h[1]);
    CK(hipFree(tmp)); CK(hipFree(d));
}
int main(int argc, char **argv) {
    size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 480000000ull; unsigned bits = argc > 2 ? (unsigned)atoi(argv[2]) : 57;
    uint64_t *src, *a, *b; CK(hipMalloc(&src, n * 8)); CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    k_fill<<<(unsigned)((n + 255) / 256), 256>>>(src, n, bits); CK(hipDeviceSynchronize());
    using namespace rocprim;
    run<default_config>("library default (8 bits per pass)", src, a, b, n, bits);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<512, 12>, kernel_config<512, 12>, 9, block_radix_rank_algorithm::match>>>("512 x 12, 9 bits, match", src, a, b, n, bits);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 6>, kernel_config<1024, 6>, 10, block_radix_rank_algorithm::match>>>("1024 x 6, 10 bits, match", src, a, b, n, bits);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 6>, kernel_config<1024, 6>, 8, block_radix_rank_algorithm::match>>>("1024 x 6, 8 bits, match", src, a, b, n, bits);
    return 0;
}
