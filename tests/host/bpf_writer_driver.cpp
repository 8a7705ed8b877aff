// bpf_writer_driver.cpp -- test driver (tests/test_bpf_writer_cpu.py): writes synthetic records through movi_host::BpfWriter, in chunks, so
// that the bytes can be compared with an independent serialisation of the record format (src/utils.cpp:202-246).
// usage: bpf_writer_driver <out.bpf> <seed> <n_records> <max_len> <n_chunks> [big_every [threads]]
// threads > 0: through BpfWriter::append(Chunk, pool) -- the chunk's arrays, a shuffled record order, slabs gathered by the pool --
// with every third chunk through append(records) in between (the two paths must keep the file's order).
#include "../../movi_amd/host/output.hpp"
#include "../../movi_amd/host/reads.hpp"
#include <algorithm>
#include <memory>
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
    const unsigned threads = argc > 7 ? (unsigned)strtoul(argv[7], nullptr, 10) : 0;
    std::unique_ptr<WorkerPool> pool;
    if (threads) pool.reset(new WorkerPool(threads));
    std::vector<uint64_t> file_order;                                 // record i of the file = read file_order[i]
    for (uint64_t c = 0; c < chunks; c++) {
        const uint64_t a = n * c / chunks, b = n * (c + 1) / chunks;
        if (threads && c % 3 != 2) {
            std::vector<uint64_t> offsets{0}, id_off{0};
            std::vector<uint16_t> vals;
            std::vector<uint8_t> id_bytes;
            std::vector<uint32_t> order;
            for (uint64_t i = a; i < b; i++) {
                vals.insert(vals.end(), pml[i].begin(), pml[i].end());
                offsets.push_back(vals.size());
                id_bytes.insert(id_bytes.end(), ids[i].begin(), ids[i].end());
                id_off.push_back(id_bytes.size());
                order.push_back((uint32_t)(i - a));
            }
            std::shuffle(order.begin(), order.end(), rng);
            for (uint32_t k : order) file_order.push_back(a + k);
            w.append(BpfWriter::Chunk{order.data(), order.size(), offsets.data(), vals.data(), id_off.data(), id_bytes.data()}, *pool);
        } else {
            std::vector<BpfWriter::Record> recs;
            for (uint64_t i = a; i < b; i++) { recs.push_back(BpfWriter::Record{ids[i], pml[i].data(), pml[i].size()}); file_order.push_back(i); }
            w.append(recs);
        }
    }
    w.close();
    // the independent serialisation, to <out>.expect
    std::string e = std::string(argv[1]) + ".expect";
    FILE *f = fopen(e.c_str(), "wb");
    const uint8_t h[12] = {0x00, 0x46, 0x50, 0x42, 1, 0, 0, 16, 0, 0, 0, 0};
    fwrite(h, 1, 12, f);
    for (uint64_t i : file_order) {
        const uint16_t idl = (uint16_t)ids[i].size();
        const uint64_t len = pml[i].size();
        fwrite(&idl, 2, 1, f); fwrite(ids[i].data(), 1, idl, f); fwrite(&len, 8, 1, f);
        if (len) fwrite(pml[i].data(), 2, len, f);
    }
    fclose(f);
    return 0;
}
