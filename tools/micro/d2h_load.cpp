// D2H bandwidth while T host threads copy page-locked buffers into tmpfs files (what the writer threads do).
// hipcc -O2 d2h_load.cpp -lpthread -o d2h_load ; ./d2h_load [threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <atomic>
#include <thread>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16; const size_t P = 16u << 20, TOTAL = 16ull << 30;
    char *d; CK(hipMalloc(&d, 4ull << 30)); CK(hipMemset(d, 1, 4ull << 30));
    for (int load = 0; load <= 1; load++) {
        std::atomic<bool> stop{false}; std::atomic<size_t> written{0};
        std::vector<std::thread> th;
        if (load) for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            char *src; if (hipHostMalloc(&src, P, hipHostMallocDefault) != hipSuccess) return;
            char fn[64]; snprintf(fn, sizeof fn, "/dev/shm/d2hload_%d", t); int fd = open(fn, O_WRONLY | O_CREAT, 0644);
            for (size_t i = 0; !stop; i++) { if (pwrite(fd, src, P, (off_t)((i % 64) * P)) != (ssize_t)P) break; written += P; }
            close(fd); unlink(fn); hipHostFree(src);
        });
        if (load) sleep(1);
        const int S = 3; hipStream_t st[S]; char *h[S * 3];
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto &p : h) CK(hipHostMalloc(&p, P, hipHostMallocDefault));
        size_t w0 = written; double t0 = now(); size_t i = 0;
        for (size_t off = 0; off < TOTAL; off += P, i++) CK(hipMemcpyAsync(h[i % (S * 3)], d + (off % (4ull << 30)), P, hipMemcpyDeviceToHost, st[i % S]));
        CK(hipDeviceSynchronize());
        double dt = now() - t0;
        printf("%s: D2H %.1f GB/s; host threads wrote %.1f GB/s meanwhile\n", load ? "with 16 writers" : "idle host     ", TOTAL / dt / 1e9, (written - w0) / dt / 1e9);
        stop = true; for (auto &x : th) x.join();
        for (auto &s : st) CK(hipStreamDestroy(s));
        for (auto &p : h) CK(hipHostFree(p));
    }
    return 0;
}
