// reads.hpp -- FASTA/FASTQ input with the reference's exact batching and read-id
// rules, plus the host-side emulation of its strand scheduler, so that output
// records come out in the order `movi query -t1` produces them.
//
//   BatchLoader::loadBatch     src/batch_loader.cpp:26-89   (batch boundaries)
//   BatchLoader::grabNextRead  src/batch_loader.cpp:91-143  (id rule, sequence assembly)
//   ReadProcessor::process_latency_hiding  src/read_processor.cpp:641-730 (record order)
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <new>
#include <string>
#include <vector>

namespace movi_host {

// A byte buffer that can grow WITHOUT initialising its bytes (std::vector value-initialises on resize: for a 1 GB chunk
// that is one single-threaded pass of page faults before the parallel fill even starts).  Grow-only capacity.
class RawBytes {
public:
    RawBytes() = default;
    RawBytes(const RawBytes &) = delete;
    RawBytes &operator=(const RawBytes &) = delete;
    ~RawBytes() { std::free(p_); }
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    uint8_t *begin() { return p_; }
    uint8_t *end() { return p_ + n_; }
    void clear() { n_ = 0; }
    void resize_uninitialized(size_t n) {                   // contents are NOT kept across a growth
        if (n > cap_) {
            std::free(p_);
            cap_ = n + (n >> 4) + 64;
            p_ = static_cast<uint8_t *>(std::malloc(cap_));
            if (!p_) { cap_ = 0; n_ = 0; throw std::bad_alloc(); }
        }
        n_ = n;
    }
    void assign(const RawBytes &o) {
        resize_uninitialized(o.n_);
        if (o.n_) std::memcpy(p_, o.p_, o.n_);
    }

private:
    uint8_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

struct ReadSet {
    std::vector<std::string> ids;       // header.substr(1, pos of first " \t\r"): keeps that whitespace char
    RawBytes bases;                     // concatenated sequences
    std::vector<uint64_t> offsets;      // n+1
    std::vector<uint32_t> batch_of;     // reference batch index of each read
    size_t size() const { return ids.size(); }
    uint64_t len(size_t i) const { return offsets[i + 1] - offsets[i]; }
};

// Line source with std::istream's good()/peek()/getline() state semantics (the batch cut of the reference depends
// on them).  Two forms: block-buffered over a stream (~GB/s instead of std::getline's ~0.3 GB/s; spans valid until the
// next call), or over a memory range (an mmap'ed read file: spans stay valid, nothing is copied).
class LineSource {
public:
    explicit LineSource(std::istream &in) : in_(&in), buf_(1u << 24) {}
    LineSource(const char *mem, size_t bytes) : mem_(mem), end_(bytes), drained_(true) {}
    bool stable() const { return mem_ != nullptr; }         // spans outlive the next call
    bool good() const { return !eof_; }
    int peek();                                            // EOF sets the eof state, like istream::peek
    bool getline(const char *&p, size_t &n);

private:
    const char *data() const { return mem_ ? mem_ : buf_.data(); }
    bool fill();
    std::istream *in_ = nullptr;
    const char *mem_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false, drained_ = false;
};

// Reads whole reference batches (BatchLoader::loadBatch) until a chunk is full.  Returns false when the input is
// exhausted and nothing was read.  `min_reads` of the constructor is 4*strands in prefetch mode, 1 with --no-prefetch
// (src/movi.cpp:283, :326).  Throws std::runtime_error on malformed input with the reference's messages.
// Two phases per chunk: (1) sequential -- cut lines and batches exactly as the reference does, validate headers, note
// where every read's id and sequence lines are (no byte of sequence is copied; with a memory-mapped input none is even
// touched except by memchr); (2) parallel -- ids, lengths and the concatenated bases are filled in by `threads` workers.
class BatchReader {
public:
    // size_hint: bytes of input when known (a regular file), 0 otherwise -- only used to reserve memory once
    BatchReader(std::istream &in, size_t min_reads, uint64_t size_hint = 0, unsigned threads = 0)
        : src_(in), min_reads_(min_reads), size_hint_(size_hint), threads_(threads) {}
    BatchReader(const char *mem, size_t bytes, size_t min_reads, unsigned threads = 0)
        : src_(mem, bytes), mem_(mem), min_reads_(min_reads), size_hint_(bytes), threads_(threads) {}
    // Whole reference batches until `max_bases` bases are held -- and, for long reads, until `min_reads`
    // reads or `hard_max_bases` bases are (one GPU lane walks one read: a chunk needs reads, not bases).
    bool next_chunk(ReadSet &out, uint64_t max_bases, uint64_t min_reads = 0, uint64_t hard_max_bases = 0);

private:
    struct Span { size_t off, len; };
    struct Rec { size_t hdr, seq_first, seq_end; uint32_t batch; };   // line indexes of one read
    bool load_batch(size_t &first_line);
    const char *line(size_t i) const { return (mem_ ? mem_ : arena_.data()) + lines_[i].off; }
    LineSource src_;
    const char *mem_ = nullptr;         // memory-mapped input: spans point into it
    std::string arena_;                 // stream input: the lines of the current CHUNK, back to back
    std::vector<Span> lines_;           // lines of the current chunk
    std::vector<Rec> recs_;
    size_t min_reads_;
    uint64_t size_hint_ = 0;
    unsigned threads_ = 0;              // 0 = hardware concurrency (at most 16)
    int format_ = -1;                   // -1 unknown, 0 FASTA, 1 FASTQ
    uint32_t batch_counter_ = 0;
};

// Order in which ReadProcessor emits the reads of `rs` with `strands` strands and one
// thread: within each reference batch, strands 0..S-1 take the first S reads; every round
// each live strand consumes `1` unit of its read; a strand that finishes writes its record at
// once and takes the batch's next read.  `cost[i]` = rounds read i needs (its length for PML).
std::vector<uint32_t> strand_order(const ReadSet &rs, const std::vector<uint64_t> &cost, size_t strands);

}  // namespace movi_host
