// output.hpp -- byte-exact writers for everything `movi query` / `movi view` emit.
//
//   BPFHeader / output_base_stats / output_binary  include/utils.hpp:64-82, src/utils.cpp:202-246
//   --stdout PML text   include/move_query.hpp:33-37 + src/utils.cpp:214-219
//   output_counts       src/utils.cpp:248-256
//   output_read         src/utils.cpp:291-294 (--filter)
//   Classifier          src/classifier.cpp:24-63 (header line), :99-143 (classify)
//   EmpNullDatabase::deserialize  src/emperical_null_database.cpp:105-126
//   view()              src/movi.cpp:402-503
#pragma once
#include <cstdint>
#include <fstream>
#include <ostream>
#include <string>
#include <string_view>
#include <vector>

#include "options.hpp"

namespace movi_host {

class WorkerPool;                            // reads.hpp

const uint32_t kBpfMagic = 0x42504600u;      // "BPF\0", include/utils.hpp:26

// BPF file: 12-byte header, then per read u16 id_len | id | u64 n | n x u16 (emission order = last base first).
// The BPF file of one query, written through a raw descriptor: records are appended a chunk at a time (see append()
// in output.cpp for what was measured).
class BpfWriter {
public:
    BpfWriter() = default;
    BpfWriter(const BpfWriter &) = delete;
    BpfWriter &operator=(const BpfWriter &) = delete;
    ~BpfWriter();
    void open(const std::string &path, uint8_t entry_size);           // creates / truncates, writes the header
    bool is_open() const { return fd_ >= 0; }
    struct Record { std::string_view id; const uint16_t *pml; uint64_t n; };
    void append(const std::vector<Record> &records);                  // in the given order
    // A chunk's records straight from its arrays (ReadSet's and the result vector): record k is read order[k], its id
    // id_bytes[id_off[i] .. id_off[i + 1]), its values pml[offsets[i] .. offsets[i + 1]).  The same bytes as append(records); the
    // records are gathered into slabs by the pool's threads while a thread of the writer's own write()s the slab before
    // (output.cpp has the measurements).  Returns once the last slab is handed over: errors of the write surface in the next
    // call or in close().
    struct Chunk { const uint32_t *order; size_t n; const uint64_t *offsets; const uint16_t *pml; const uint64_t *id_off; const uint8_t *id_bytes; };
    void append(const Chunk &c, WorkerPool &pool);
    void close();
    struct Times { double gather = 0, wait = 0, write = 0; };         // append(Chunk): gathering, waiting for a free slab; the write()s
    Times times() const;

private:
    struct Async;
    void drain();
    int fd_ = -1;
    std::string path_;
    std::vector<uint8_t> buf_;
    Async *async_ = nullptr;
    double gather_s_ = 0, wait_s_ = 0;
};

// `>id\n` + values in read order, each followed by a space, + `\n`
void write_stdout_pmls(std::ostream &out, std::string_view id, const uint16_t *pml, uint64_t n);
void append_stdout_pmls(std::string &txt, std::string_view id, const uint16_t *pml, uint64_t n);   // the same text, appended
void write_count_line(std::ostream &out, std::string_view id, uint64_t query_length, uint64_t matched, uint64_t count);
void append_count_line(std::string &txt, std::string_view id, uint64_t query_length, uint64_t matched, uint64_t count);   // the same line, appended

class Classifier {
public:
    // Reads DIR/movi.pml.nulldb; returns max_value_thr = max(percentile, 3) + 1.
    size_t load_null_db(const std::string &index_dir, const std::string &query_type, bool verbose);
    void write_report_header(std::ostream &out) const;
    // Bins of bin_width over the PML vector in emission order; the last bin absorbs a
    // remainder shorter than bin_width.  Writes the report line unless out == nullptr.
    bool classify(std::string_view read_name, const uint16_t *pml, uint64_t n, size_t bin_width, std::ostream *out) const;
    uint16_t max_value_thr = 0;
};

int view_bpf(const Options &o, std::ostream &out);

}  // namespace movi_host
