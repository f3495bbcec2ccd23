// Can D2H copies land directly in tmpfs page-cache pages?  mmap a /dev/shm file, hipHostRegister the
// mapping, copy into it, unregister.  Reports per-phase rates with T threads.  hipcc -O2 reg_file.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8; const size_t P = (argc > 2 ? atoi(argv[2]) : 16) * (size_t)1 << 20; const int N = argc > 3 ? atoi(argv[3]) : 32;
    char *d; if (hipMalloc(&d, P * 4) != hipSuccess) return 1; hipMemset(d, 0x41, P * 4);
    for (int round = 0; round < 2; round++) {               // round 0: fresh files, round 1: overwrite in place
        std::atomic<int> fail{0}; std::vector<double> treg(T), tcopy(T), tunreg(T), tmap(T);
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int i = 0; i < N; i++) {
                char fn[256]; snprintf(fn, sizeof fn, "/dev/shm/regtest_%d_%d", t, i);
                double a = now();
                int fd = open(fn, O_RDWR | O_CREAT, 0644); if (fd < 0 || ftruncate(fd, P)) { fail++; return; }
                void *p = mmap(nullptr, P, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0); if (p == MAP_FAILED) { fail++; return; }
                double b = now(); tmap[t] += b - a;
                if (hipHostRegister(p, P, hipHostRegisterDefault) != hipSuccess) { fail++; (void)hipGetLastError(); munmap(p, P); close(fd); return; }
                double c = now(); treg[t] += c - b;
                if (hipMemcpyAsync(p, d + (size_t)(i % 4) * P, P, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { fail++; return; }
                double e = now(); tcopy[t] += e - c;
                hipHostUnregister(p); munmap(p, P); close(fd);
                tunreg[t] += now() - e;
            }
        });
        for (auto &x : th) x.join();
        double wall = now() - t0;
        printf("round %d threads %d piece %zu MiB x %d: fail %d wall %.3f s -> %.1f GB/s | per-thread s: map %.3f reg %.3f copy %.3f unreg %.3f\n", round, T, P >> 20, N, fail.load(), wall, (double)T * N * P / wall / 1e9, tmap[0], treg[0], tcopy[0], tunreg[0]);
    }
    // verify one file, then clean up
    { char buf[16]; int fd = open("/dev/shm/regtest_0_0", O_RDONLY); ssize_t r = fd >= 0 ? pread(fd, buf, 16, P - 16) : -1; printf("verify: %zd bytes, first %02x\n", r, r > 0 ? (unsigned char)buf[0] : 0); if (fd >= 0) close(fd); }
    for (int t = 0; t < T; t++) for (int i = 0; i < N; i++) { char fn[256]; snprintf(fn, sizeof fn, "/dev/shm/regtest_%d_%d", t, i); unlink(fn); }
    return 0;
}
