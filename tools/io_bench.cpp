// Host I/O primitives behind `movi query` (round 5, profiles/r05_cli_path.txt): how fast can this box (a) get a read file that sits in
// the page cache in front of N parser threads -- a private file mapping (page faults) against pread() into an anonymous buffer, fresh
// or already touched --, and (b) put a BPF file of a few hundred MB into the page cache -- one write() stream against N threads
// pwrite()-ing large disjoint ranges of a fallocate()d file, a shared mapping, and O_DIRECT.
//   io_bench read  FILE            (FILE: any file of 100 MB+, e.g. the bench's FASTA)
//   io_bench write PATH [MB]       (PATH is created and removed)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static void par(int nt, F f) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=] { f(t); });
    for (auto &x : th) x.join();
}

static size_t count_nl(const char *p, size_t n) {
    size_t c = 0;
    const char *e = p + n;
    while (p < e) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        ++c;
        p = q + 1;
    }
    return c;
}

static int bench_read(const char *path) {
    const int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) { perror(path); return 1; }
    const size_t n = (size_t)sb.st_size;
    {   // make sure the file is in the page cache
        std::vector<char> tmp(8u << 20);
        size_t o = 0;
        while (o < n) { const ssize_t k = pread(fd, tmp.data(), tmp.size(), (off_t)o); if (k <= 0) break; o += (size_t)k; }
    }
    printf("# %s: %.1f MB, in the page cache\n", path, n / 1e6);
    for (int nt : {1, 2, 4, 8, 12}) {
        // (1) a fresh private mapping, the threads count newlines in disjoint ranges (first touch = file-mapping faults)
        for (int adv = 0; adv < 2; ++adv) {
            const double t0 = now();
            char *m = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); return 1; }
            if (adv) madvise(m, n, MADV_SEQUENTIAL);
            std::vector<size_t> c((size_t)nt);
            par(nt, [&, m](int t) { const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt; c[(size_t)t] = count_nl(m + a, b - a); });
            const double t1 = now();
            munmap(m, n);
            size_t tot = 0;
            for (size_t x : c) tot += x;
            printf("read  mmap%-11s threads %2d  %7.2f ms  %6.2f GB/s  (%zu lines)\n", adv ? "+sequential" : "", nt, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, tot);
        }
        // (2) pread into an anonymous buffer: fresh (first touch inside the copy), fresh with huge pages asked for, and touched before
        for (int flavour = 0; flavour < 3; ++flavour) {
            char *buf = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (buf == MAP_FAILED) { perror("mmap anon"); return 1; }
            if (flavour >= 1) madvise(buf, n, MADV_HUGEPAGE);
            if (flavour == 2) par(nt, [&, buf](int t) { const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt; memset(buf + a, 1, b - a); });
            std::vector<size_t> c((size_t)nt);
            const double t0 = now();
            par(nt, [&, buf](int t) {
                const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
                size_t o = a;
                while (o < b) { const ssize_t k = pread(fd, buf + o, std::min<size_t>(4u << 20, b - o), (off_t)o); if (k <= 0) abort(); o += (size_t)k; }
            });
            const double t1 = now();
            par(nt, [&, buf](int t) { const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt; c[(size_t)t] = count_nl(buf + a, b - a); });
            const double t2 = now();
            munmap(buf, n);
            printf("read  pread %-10s threads %2d  %7.2f ms  %6.2f GB/s  + scan %6.2f ms\n", flavour == 0 ? "fresh" : flavour == 1 ? "fresh+thp" : "touched", nt,
                   (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
        }
    }
    close(fd);
    return 0;
}

static int bench_write(const char *path, size_t mb) {
    const size_t total = mb << 20;
    char *src = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (src == MAP_FAILED) return 1;
    for (size_t i = 0; i < total; i += 8) *(uint64_t *)(src + i) = i * 0x9E3779B97F4A7C15ull;
    printf("# %s: %zu MB per file\n", path, mb);
    for (int rep = 0; rep < 2; ++rep) {
        {   // one write() stream, 8 MiB at a time (what output.cpp does)
            unlink(path);
            const double t0 = now();
            const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
            size_t o = 0;
            while (o < total) { const size_t k = std::min<size_t>(8u << 20, total - o); if (write(fd, src + o, k) != (ssize_t)k) { perror("write"); return 1; } o += k; }
            close(fd);
            const double t1 = now();
            printf("write stream                   %8.2f ms  %6.2f GB/s\n", (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
        }
        for (int fal = 0; fal < 2; ++fal)
            for (int nt : {2, 4, 8}) {   // N threads, one large disjoint range each, pwrite 4 MiB at a time; with and without fallocate first
                unlink(path);
                const double t0 = now();
                const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
                if (fal && posix_fallocate(fd, 0, (off_t)total)) perror("fallocate");
                par(nt, [&](int t) {
                    size_t o = total * (size_t)t / (size_t)nt;
                    const size_t e = total * (size_t)(t + 1) / (size_t)nt;
                    while (o < e) { const size_t k = std::min<size_t>(4u << 20, e - o); if (pwrite(fd, src + o, k, (off_t)o) != (ssize_t)k) abort(); o += k; }
                });
                close(fd);
                const double t1 = now();
                printf("write pwrite x%d %-13s %8.2f ms  %6.2f GB/s\n", nt, fal ? "(fallocated)" : "", (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
            }
        for (int nt : {1, 4, 8}) {       // a shared mapping of the fallocate()d file, N threads memcpy into disjoint ranges
            unlink(path);
            const double t0 = now();
            const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
            if (ftruncate(fd, (off_t)total)) return 1;
            char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { perror("mmap shared"); return 1; }
            par(nt, [&, m](int t) { const size_t a = total * (size_t)t / (size_t)nt, b = total * (size_t)(t + 1) / (size_t)nt; memcpy(m + a, src + a, b - a); });
            munmap(m, total);
            close(fd);
            const double t1 = now();
            printf("write shared mapping x%d        %8.2f ms  %6.2f GB/s\n", nt, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
        }
        for (int nt : {1, 4}) {          // O_DIRECT, aligned 8 MiB slabs (src is page-aligned)
            unlink(path);
            const double t0 = now();
            const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC | O_DIRECT, 0644);
            if (fd < 0) { printf("write O_DIRECT x%d: open failed (%s)\n", nt, strerror(errno)); continue; }
            bool bad = false;
            par(nt, [&](int t) {
                size_t o = (total * (size_t)t / (size_t)nt) & ~((size_t)(8u << 20) - 1);
                const size_t e = t + 1 == nt ? total : (total * (size_t)(t + 1) / (size_t)nt) & ~((size_t)(8u << 20) - 1);
                while (o < e) { const size_t k = std::min<size_t>(8u << 20, e - o); if (pwrite(fd, src + o, k, (off_t)o) != (ssize_t)k) { bad = true; return; } o += k; }
            });
            close(fd);
            const double t1 = now();
            if (bad) printf("write O_DIRECT x%d: pwrite failed (%s)\n", nt, strerror(errno));
            else printf("write O_DIRECT x%d               %8.2f ms  %6.2f GB/s\n", nt, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
        }
    }
    unlink(path);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "read")) return bench_read(argv[2]);
    if (argc >= 3 && !strcmp(argv[1], "write")) return bench_write(argv[2], argc > 3 ? (size_t)atol(argv[3]) : 320);
    fprintf(stderr, "usage: io_bench read FILE | io_bench write PATH [MB]\n");
    return 2;
}
