#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>
#include <linux/falloc.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t total = 300u << 20;
    const int nt = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<char> src(total, 'x');
    for (int rep = 0; rep < 3; ++rep) {   // pwrite, one thread / nt threads at their offsets
        for (int par = 0; par < 2; ++par) {
            const char* path = "/dev/shm/iot_test.bin";
            unlink(path);
            int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
            double t0 = now();
            const size_t per = 1 << 20;
            const int k_n = par ? nt : 1;
            std::vector<std::thread> th;
            for (int k = 0; k < k_n; ++k) th.emplace_back([&, k] { for (size_t o = k * per; o < total; o += k_n * per) { ssize_t r = pwrite(fd, src.data() + o, std::min(per, total - o), o); (void)r; } });
            for (auto& x : th) x.join();
            close(fd);
            printf("pwrite %d thread(s): %.1f ms\n", k_n, (now() - t0) * 1e3);
            unlink(path);
        }
    }
    for (int mode = 0; mode < 4; ++mode) for (int rep = 0; rep < 3; ++rep) {
        const char* path = "/dev/shm/iot_test.bin";
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        ftruncate(fd, total);
        if (mode == 1) fallocate(fd, 0, 0, total);
        int flags = MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0);
        char* mp = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, flags, fd, 0);
        if (mode == 3) madvise(mp, total, MADV_POPULATE_WRITE);
        double t1 = now();
        std::vector<std::thread> th;
        const size_t per = 1 << 20;   // 1-MB parts dealt round robin
        for (int k = 0; k < nt; ++k) th.emplace_back([&, k] { for (size_t o = k * per; o < total; o += nt * per) memcpy(mp + o, src.data() + o, std::min(per, total - o)); });
        for (auto& x : th) x.join();
        double t2 = now();
        munmap(mp, total);
        close(fd);
        double t3 = now();
        printf("mode %d (%s): prep %.1f ms copy %.1f ms unmap %.1f ms total %.1f ms\n", mode, mode == 0 ? "ftruncate+mmap" : mode == 1 ? "fallocate" : mode == 2 ? "MAP_POPULATE" : "MADV_POPULATE_WRITE", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
        unlink(path);
    }
}
