// reads.hpp -- FASTA/FASTQ input with the reference's exact batching and read-id
// rules, plus the host-side emulation of its strand scheduler, so that output
// records come out in the order `movi query -t1` produces them.
//
//   BatchLoader::loadBatch     src/batch_loader.cpp:26-89   (batch boundaries)
//   BatchLoader::grabNextRead  src/batch_loader.cpp:91-143  (id rule, sequence assembly)
//   ReadProcessor::process_latency_hiding  src/read_processor.cpp:641-730 (record order)
#pragma once
#include <cstdint>
#include <istream>
#include <string>
#include <vector>

namespace movi_host {

struct ReadSet {
    std::vector<std::string> ids;       // header.substr(1, pos of first " \t\r"): keeps that whitespace char
    std::vector<uint8_t> bases;         // concatenated sequences
    std::vector<uint64_t> offsets;      // n+1
    std::vector<uint32_t> batch_of;     // reference batch index of each read
    size_t size() const { return ids.size(); }
    uint64_t len(size_t i) const { return offsets[i + 1] - offsets[i]; }
};

// Reads up to `max_bases` bases worth of whole reference batches from `in`
// (at least one batch).  Returns false when the input is exhausted and nothing
// was read.  `min_reads` is 4*strands in prefetch mode, 1 with --no-prefetch
// (src/movi.cpp:283, :326).  Throws std::runtime_error on malformed input with the
// reference's messages.
// Block-buffered line source with std::istream's good()/peek()/getline() state semantics
// (the batch cut of the reference depends on them), ~GB/s instead of std::getline's ~0.3 GB/s.
class LineSource {
public:
    explicit LineSource(std::istream &in) : in_(in), buf_(1u << 24) {}
    bool good() const { return !eof_; }
    int peek();                                            // EOF sets the eof state, like istream::peek
    bool getline(const char *&p, size_t &n);               // span valid until the next call

private:
    bool fill();
    std::istream &in_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false, drained_ = false;
};

class BatchReader {
public:
    BatchReader(std::istream &in, size_t min_reads) : src_(in), min_reads_(min_reads) {}
    // Whole reference batches until `max_bases` bases are held -- and, for long reads, until `min_reads`
    // reads or `hard_max_bases` bases are (one GPU lane walks one read: a chunk needs reads, not bases).
    bool next_chunk(ReadSet &out, uint64_t max_bases, uint64_t min_reads = 0, uint64_t hard_max_bases = 0);

private:
    struct Span { size_t off, len; };
    bool load_batch();
    LineSource src_;
    std::string arena_;                 // the lines of the current reference batch, back to back
    std::vector<Span> lines_;
    size_t min_reads_;
    int format_ = -1;                   // -1 unknown, 0 FASTA, 1 FASTQ
    uint32_t batch_counter_ = 0;
};

// Order in which ReadProcessor emits the reads of `rs` with `strands` strands and one
// thread: within each reference batch, strands 0..S-1 take the first S reads; every round
// each live strand consumes `1` unit of its read; a strand that finishes writes its record at
// once and takes the batch's next read.  `cost[i]` = rounds read i needs (its length for PML).
std::vector<uint32_t> strand_order(const ReadSet &rs, const std::vector<uint64_t> &cost, size_t strands);

}  // namespace movi_host
