// D2H bandwidth probe: pinned host buffers, S streams, piece size P, total T bytes.  hipcc -O2 d2h_bw.cpp -o d2h_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    const size_t T = 8ull << 30;
    char *d; CK(hipMalloc(&d, T)); CK(hipMemset(d, 1, T));
    for (int S : {1, 2, 3, 4, 8, 16}) for (size_t P : {size_t(2) << 20, size_t(8) << 20, size_t(64) << 20}) {
        std::vector<hipStream_t> st(S); std::vector<char *> h(S * 2);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &p : h) CK(hipHostMalloc(&p, P, hipHostMallocDefault));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        size_t i = 0;
        for (size_t off = 0; off < T; off += P, i++) CK(hipMemcpyAsync(h[i % (S * 2)], d + off, P, hipMemcpyDeviceToHost, st[i % S]));
        CK(hipDeviceSynchronize());
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("streams %2d piece %3zu MiB: %.1f GB/s\n", S, P >> 20, T / s / 1e9);
        for (auto &s2 : st) CK(hipStreamDestroy(s2));
        for (auto &p : h) CK(hipHostFree(p));
    }
    return 0;
}
