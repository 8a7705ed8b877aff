// bpf_writer_driver.cpp -- test driver (tests/test_bpf_writer_cpu.py): writes synthetic records through movi_host::BpfWriter, in chunks, so
// that the bytes can be compared with an independent serialisation of the record format (src/utils.cpp:202-246).
// usage: bpf_writer_driver <out.bpf> <seed> <n_records> <max_len> <n_chunks> [big_every]
#include "../../movi_amd/host/output.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace movi_host;
int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const uint64_t seed = strtoull(argv[2], nullptr, 10), n = strtoull(argv[3], nullptr, 10), max_len = strtoull(argv[4], nullptr, 10),
                   chunks = strtoull(argv[5], nullptr, 10), big_every = argc > 6 ? strtoull(argv[6], nullptr, 10) : 0;
    std::mt19937_64 rng(seed);
    BpfWriter w;
    w.open(argv[1], 16);
    std::vector<std::string> ids(n);
    std::vector<std::vector<uint16_t>> pml(n);
    for (uint64_t i = 0; i < n; i++) {
        ids[i] = "read" + std::to_string(i) + std::string((size_t)(rng() % 5), ' ');
        uint64_t len = rng() % (max_len + 1);
        if (big_every && i % big_every == big_every - 1) len = (1u << 19) + rng() % (1u << 19);   // payloads of 1 MiB and more: written from the array
        pml[i].resize(len);
        for (auto &v : pml[i]) v = (uint16_t)rng();
    }
    for (uint64_t c = 0; c < chunks; c++) {
        std::vector<BpfWriter::Record> recs;
        for (uint64_t i = n * c / chunks; i < n * (c + 1) / chunks; i++) recs.push_back(BpfWriter::Record{ids[i], pml[i].data(), pml[i].size()});
        w.append(recs);
    }
    w.close();
    // the independent serialisation, to <out>.expect
    std::string e = std::string(argv[1]) + ".expect";
    FILE *f = fopen(e.c_str(), "wb");
    const uint8_t h[12] = {0x00, 0x46, 0x50, 0x42, 1, 0, 0, 16, 0, 0, 0, 0};
    fwrite(h, 1, 12, f);
    for (uint64_t i = 0; i < n; i++) {
        const uint16_t idl = (uint16_t)ids[i].size();
        const uint64_t len = pml[i].size();
        fwrite(&idl, 2, 1, f); fwrite(ids[i].data(), 1, idl, f); fwrite(&len, 8, 1, f);
        if (len) fwrite(pml[i].data(), 2, len, f);
    }
    fclose(f);
    return 0;
}
