// parse_bench.cpp -- the FASTA / FASTQ parser of `movi query` alone (movi_amd/host/reads.cpp): chunks per second and where
// the time goes, by worker count.  build: g++ -O2 -std=c++17 -o tools/parse_bench tools/parse_bench.cpp movi_amd/host/reads.cpp -lpthread
// run: tools/parse_bench reads.fa [threads ...]
#include "../movi_amd/host/reads.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
using namespace movi_host;
int main(int argc, char **argv) {
    if (argc < 2) return 1;
    int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { perror("open"); return 1; }
    const char *p = (const char *)mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    std::vector<unsigned> ts;
    for (int i = 2; i < argc; i++) ts.push_back((unsigned)atoi(argv[i]));
    if (ts.empty()) ts = {1, 4, 8, 16};
    const uint64_t chunk = getenv("MOVI_CHUNK_BASES") ? strtoull(getenv("MOVI_CHUNK_BASES"), nullptr, 10) : 1ull << 25;   // (small chunks: many scan-ahead windows)
    ReadSet rs[3];                                        // circulating, as in the command: warm after the first pass
    for (unsigned T : ts)
        for (int rep = 0; rep < 3; rep++) {
            BatchReader rd(p, (size_t)sb.st_size, 64, T);
            int k = 0;
            uint64_t bases = 0, reads = 0;
            auto t0 = std::chrono::steady_clock::now();
            while (rd.next_chunk(rs[k % 3], chunk, chunk < (1ull << 25) ? 1 : 1ull << 15, 1ull << 30)) { bases += rs[k % 3].bases.size(); reads += rs[k % 3].size(); k++; }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const BatchReader::PhaseTimes &pt = rd.phase_times();
            printf("T=%-2u chunks %d reads %lu bases %lu: %.4f s = %.2f Gbases/s  (newline scan %.4f, batch cut %.4f, lengths %.4f, copy %.4f; %lu reads cut in bulk, %lu in closed form)\n",
                   T, k, (unsigned long)reads, (unsigned long)bases, dt, bases / dt / 1e9, pt.prescan, pt.cut, pt.lengths, pt.copy, (unsigned long)pt.bulk_reads, (unsigned long)pt.closed_reads);
        }
    return 0;
}
