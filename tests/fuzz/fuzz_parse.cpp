// fuzz_parse.cpp -- CPU-only fuzz harness for movi_index_parse, built with ASan + UBSan (sanitizers run on the CPU build
// only: GPU ASan is not available on the pool).  Every truncation of the header and tail regions, sampled truncations of
// the rows, and random 64-bit field smashing; each candidate lives in an exact-size heap block so that any over-read is
// a sanitizer report, and whatever the parser ACCEPTS must be self-consistent (rows and side tables inside the image).
// build + run: tools/fuzz_parse.sh [iterations per image]   (test: tests/test_abi_cpu.py::test_parse_fuzz_sanitized)
// usage: fuzz_parse <iterations> image.movi [...]
#include "../../include/movi_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v(n); if (fread(v.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return v;
}
int main(int argc, char **argv) {
    std::mt19937_64 rng(12345);
    long ok = 0, bad = 0;
    if (argc < 3) { fprintf(stderr, "usage: fuzz_parse <iterations> image.movi [...]\n"); return 2; }
    const int iterations = atoi(argv[1]);
    for (int a = 2; a < argc; a++) {
        const std::vector<uint8_t> img = slurp(argv[a]);
        movi_index_desc_t d; size_t off, nb;
        if (movi_index_parse(img.data(), img.size(), &d, &off, &nb) != MOVI_OK) { printf("seed image %s rejected: %s\n", argv[a], movi_last_error()); return 1; }
        const size_t tail = img.size() - (off + nb);
        // 1. every truncation of the header region and of the tail, plus sampled truncations of the rows
        for (size_t cut = 0; cut < img.size(); cut += (cut < off + 64 || cut + tail + 64 >= img.size()) ? 1 : 4099) {
            std::vector<uint8_t> t(img.begin(), img.begin() + cut);        // exact-size heap copy: ASan sees any over-read
            (movi_index_parse(t.data(), t.size(), &d, &off, &nb) == MOVI_OK ? ok : bad)++;
        }
        // 2. random 64-bit field smashing in the header region and the tail
        for (int it = 0; it < iterations; it++) {
            std::vector<uint8_t> t(img);
            const int nmut = 1 + (int)(rng() % 3);
            for (int m = 0; m < nmut; m++) {
                size_t region = rng() % 2 ? (size_t)(rng() % 2300) : img.size() - 1 - (size_t)(rng() % (tail + 64));
                if (region + 8 > t.size()) region = t.size() - 8;
                uint64_t v;
                switch (rng() % 6) {
                    case 0: v = 0; break;
                    case 1: v = ~0ull; break;
                    case 2: v = 1ull << (rng() % 64); break;
                    case 3: v = (1ull << (rng() % 64)) - 1; break;
                    case 4: memcpy(&v, &t[region], 8); v += (1ull << 61); break;
                    default: v = rng(); break;
                }
                memcpy(&t[region], &v, 1 + rng() % 8);
            }
            size_t o2, n2;
            if (movi_index_parse(t.data(), t.size(), &d, &o2, &n2) == MOVI_OK) {
                ok++;
                // whatever is accepted must be self-consistent: the rows and every side table lie inside the image
                if (o2 > t.size() || n2 > t.size() - o2) { printf("accepted image with rows outside the buffer\n"); return 1; }
                const uint8_t *lo = t.data(), *hi = t.data() + t.size();
                auto inside = [&](const void *p, size_t bytes) { return !p || ((const uint8_t *)p >= lo && bytes <= (size_t)(hi - (const uint8_t *)p)); };
                if (!inside(d.id_blocks, (size_t)d.n_blocks * d.alphabet_size * 4) || !inside(d.tally_ids, (size_t)d.n_tally * d.alphabet_size * 5) ||
                    !inside(d.separator_thresholds, (size_t)d.n_separator_thresholds * 8) || !inside(d.separator_map, (size_t)d.n_separator_map * 16)) {
                    printf("accepted image with a side table outside the buffer\n"); return 1;
                }
                if (d.r == 0 || d.r >= (1ull << 36) || d.end_bwt_idx >= d.r || n2 / d.r > 8) { printf("accepted image with bad r\n"); return 1; }
            } else bad++;
        }
    }
    printf("fuzz ok: %ld accepted, %ld rejected, no sanitizer report\n", ok, bad);
    return 0;
}
