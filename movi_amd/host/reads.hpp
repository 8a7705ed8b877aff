// reads.hpp -- FASTA/FASTQ input with the reference's exact batching and read-id
// rules, plus the host-side emulation of its strand scheduler, so that output
// records come out in the order `movi query -t1` produces them.
//
//   BatchLoader::loadBatch     src/batch_loader.cpp:26-89   (batch boundaries)
//   BatchLoader::grabNextRead  src/batch_loader.cpp:91-143  (id rule, sequence assembly)
//   ReadProcessor::process_latency_hiding  src/read_processor.cpp:641-730 (record order)
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

namespace movi_host {

// A byte buffer that can grow WITHOUT initialising its bytes (std::vector value-initialises on resize: for a 1 GB chunk
// that is one single-threaded pass of page faults before the parallel fill even starts).  Grow-only capacity.
// The memory comes from malloc unless set_allocator() names another source -- `movi query` hands the chunk buffers
// page-locked memory (movi_host_alloc) so that the engine's host entry points overlap their transfers with the walk.
class RawBytes {
public:
    typedef void *(*AllocFn)(size_t);
    typedef void (*FreeFn)(void *);
    RawBytes() = default;
    RawBytes(const RawBytes &) = delete;
    RawBytes &operator=(const RawBytes &) = delete;
    ~RawBytes() { release(); }
    void set_allocator(AllocFn a, FreeFn f) { release(); alloc_ = a; free_ = f; }   // drops what is held
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    uint8_t *begin() { return p_; }
    uint8_t *end() { return p_ + n_; }
    void clear() { n_ = 0; }
    void resize_uninitialized(size_t n) {                   // contents are NOT kept across a growth
        if (n > cap_) {
            release();
            cap_ = n + (n >> 4) + 64;
            p_ = static_cast<uint8_t *>(alloc_ ? alloc_(cap_) : std::malloc(cap_));
            if (!p_) { cap_ = 0; n_ = 0; throw std::bad_alloc(); }
        }
        n_ = n;
    }
    void assign(const RawBytes &o) {
        resize_uninitialized(o.n_);
        if (o.n_) std::memcpy(p_, o.p_, o.n_);
    }

private:
    void release() {
        if (p_) { if (free_) free_(p_); else std::free(p_); }
        p_ = nullptr;
        n_ = cap_ = 0;
    }
    uint8_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    AllocFn alloc_ = nullptr;
    FreeFn free_ = nullptr;
};

// A growable array of plain structs whose resize leaves the new elements uninitialised (they are about to be filled by
// several workers at once: value-initialising 2 M line records first is a single-threaded pass over 32 MB).
template <typename T>
class PodVec {
public:
    PodVec() = default;
    PodVec(const PodVec &) = delete;
    PodVec &operator=(const PodVec &) = delete;
    ~PodVec() { std::free(p_); }
    T *data() { return p_; }
    const T *data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    size_t capacity() const { return cap_; }
    void reserve(size_t n) {
        if (n <= cap_) return;
        T *q = static_cast<T *>(std::realloc(p_, n * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p_ = q;
        cap_ = n;
    }
    void resize_uninitialized(size_t n) { reserve(n); n_ = n; }   // contents below the old size are kept
    void push_back(const T &v) {
        if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 1024);
        p_[n_++] = v;
    }

private:
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

// One chunk of reads.  Ids and bases are two byte arenas with offsets: no per-read allocation (a million std::strings
// per chunk were a quarter of the parser's time, and ids longer than 15 characters -- every SRR id -- an allocation each).
struct ReadSet {
    RawBytes id_bytes;                  // the ids back to back: header.substr(1, pos of first " \t\r") keeps that whitespace char
    std::vector<uint64_t> id_off;       // n+1
    RawBytes bases;                     // concatenated sequences
    std::vector<uint64_t> offsets;      // n+1
    std::vector<uint32_t> batch_of;     // reference batch index of each read
    size_t size() const { return batch_of.size(); }
    uint64_t len(size_t i) const { return offsets[i + 1] - offsets[i]; }
    std::string_view id(size_t i) const {
        return std::string_view(reinterpret_cast<const char *>(id_bytes.data()) + id_off[i], (size_t)(id_off[i + 1] - id_off[i]));
    }
};

// The hardware threads that share a last-level cache with the calling thread (sysfs); empty if unknown.
std::vector<int> llc_siblings();

// A few worker threads that stay around between chunks (two thread launches per phase and chunk were ~2 ms of a 10 ms chunk).
class WorkerPool {
public:
    explicit WorkerPool(unsigned threads);
    ~WorkerPool();
    unsigned size() const { return (unsigned)th_.size() + 1; }
    // the calling thread joins the pool's cache domain (a thread other than the one that built the pool is going to drive it)
    void adopt_owner();
    void run(unsigned parts, const std::function<void(unsigned)> &fn);   // fn(0 .. parts-1), the caller works too; returns when all are done

private:
    void loop();
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned parts_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    cpu_set_t owner_mask_;                // the owning thread's affinity before the pool narrowed it (restored by ~WorkerPool)
    pthread_t owner_{};
    bool have_owner_mask_ = false, pinned_owner_ = false;
    cpu_set_t pool_set_;                  // the cache domain the pool's threads are pinned to (valid if pinned_)
    bool pinned_ = false;
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
    int peek() {                                           // EOF sets the eof state, like istream::peek
        if (pos_ < end_) return (unsigned char)data()[pos_];
        return peek_slow();
    }
    bool getline(const char *&p, size_t &n) {
        if (nl_i_ < cur_.nl.size() || next_window()) {     // a newline found by the scan-ahead
            const size_t at = cur_.nl[nl_i_];
            next_first_ = cur_.first[nl_i_++];             // ... and the byte behind it: the next line's first character
            p = mem_ + pos_;
            n = at - pos_;
            pos_ = at + 1;
            have_next_first_ = true;
            return true;
        }
        have_next_first_ = false;
        return getline_slow(p, n);
    }
    // first character of the line at the cursor without touching the input, when the scan knows it (else peek())
    int peek_first() {
        if (have_next_first_ && pos_ < end_) return next_first_;
        return peek();
    }
    size_t prescanned_lines() const { return cur_.nl.size() - nl_i_; }
    // memory form: the complete lines the scan has found from the cursor on -- line t (t < count) spans [t ? nl[t-1] + 1 : pos,
    // nl[t]) and starts with first0 (t = 0; -1: unknown) or first[t-1]; first[t] is what the line behind it starts with (0 at
    // the end of the input).  Valid until the next getline() / consume().
    struct Ahead { const size_t *nl; const uint8_t *first; size_t count, pos; int first0; };
    Ahead ahead() const {
        Ahead a;
        a.nl = cur_.nl.data() + nl_i_;
        a.first = cur_.first.data() + nl_i_;
        a.count = cur_.nl.size() - nl_i_;
        a.pos = pos_;
        a.first0 = (have_next_first_ && pos_ < end_) ? (int)next_first_ : (pos_ < end_ ? (int)(unsigned char)mem_[pos_] : -1);
        return a;
    }
    void consume(size_t lines) {                           // lines <= ahead().count: as many getline() calls
        if (lines == 0) return;
        const size_t i = nl_i_ + lines - 1;
        pos_ = cur_.nl[i] + 1;
        next_first_ = cur_.first[i];
        have_next_first_ = true;
        nl_i_ += lines;
    }
    ~LineSource();
    // memory form only: newlines are found ahead of the parser, a window of `bytes` bytes at a time -- the first window by
    // all workers of `pool` at once, every later one by a helper thread while the parser cuts, measures and copies the chunk
    // before it (the helper also takes the page faults of a fresh file mapping off the parser's path); getline() then takes
    // the line ends from the list instead of running memchr line by line
    void prescan(size_t bytes, WorkerPool &pool);

private:
    const char *data() const { return mem_ ? mem_ : buf_.data(); }
    bool fill();
    int peek_slow();
    bool getline_slow(const char *&p, size_t &n);
    std::istream *in_ = nullptr;
    const char *mem_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false, drained_ = false;
    // A scanned window of the input: newline offsets in [from, to), ascending, and the byte after each of them (0 at the end
    // of the input): with those the sequential batch cut never touches the input itself -- reading one byte per line
    // streamed most of the file through one core (15 ms per 160 MB).
    struct Window { std::vector<size_t> nl; std::vector<uint8_t> first; size_t from = 0, to = 0; };
public:
    double scan_busy_s() const { return scan_busy_s_; }
    double scan_wait_s() const { return scan_wait_s_; }
private:
    bool next_window();                 // cur_ is used up: adopt the window scanned ahead (waits for it), scan the one after
    void scan_ahead();
    struct Pending { Window w; std::thread th; double busy_s = 0; };
    Window cur_;
    std::deque<std::unique_ptr<Pending>> ahead_;   // windows being scanned behind cur_, oldest first
    size_t ahead_to_ = 0;               // where the last window handed to a helper ends
    unsigned scan_depth_ = 1;           // windows kept in flight (prescan() raises it once the parser is running)
    double scan_busy_s_ = 0, scan_wait_s_ = 0;   // the helper's own time per window, summed; what next_window() waited for it
    size_t window_bytes_ = 0;
    size_t nl_i_ = 0;                   // next unused entry of cur_
    uint8_t next_first_ = 0;
    bool have_next_first_ = false;
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
    // Round 5 -- memory-mapped input only, BEFORE the first next_chunk and possibly on another thread than the one that will call
    // it: build the worker pool, scan the first window (what the first next_chunk does first: it also takes the file mapping's page
    // faults), size the circulating ReadSets and the reader's own line / record tables for a chunk of `max_bases` and let the POOL
    // touch their pages -- first touch by the pinned workers keeps the memory on their NUMA node.  `movi query` does this while the
    // index loads: the first three chunks of a run used to be parsed into fresh memory (chunk 1: 7.4 ms, chunks 2 - 3: 3.9 - 4.3 ms,
    // warm: 2.8 ms; tools/r05_cli.sh).  The thread that then calls next_chunk calls adopt_pool() first.
    void warm_up(ReadSet *const *sets, unsigned n_sets, uint64_t max_bases, double lines_per_byte = 0.0);
    double warm_input_seconds() const { return warm_input_s_; }   // seconds warm_up spent reading the input (the first window's scan): part of the read-processing clock
    void adopt_pool() { if (pool_) pool_->adopt_owner(); }
    // seconds spent in the parser's phases so far (movi query --verbose, tools/parse_bench.cpp)
    struct PhaseTimes { double prescan = 0, cut = 0, lengths = 0, copy = 0, scan_busy = 0, scan_wait = 0; uint64_t bulk_reads = 0, closed_reads = 0, reads = 0; };   // scan_*: the scan-ahead helper's own time / what the cut waited for it (inside `cut`)   // bulk_reads: cut by cut_ahead (closed_reads of them: batches by the closed form)
    PhaseTimes phase_times() const { PhaseTimes t = times_; t.scan_busy = src_.scan_busy_s(); t.scan_wait = src_.scan_wait_s(); return t; }

private:
    struct Span { uint64_t off; uint32_t len; uint8_t first; };   // 16 bytes; first = the line's first character (0: empty line)
    struct Rec { uint32_t hdr, seq_first, seq_end, batch; };          // line indexes of one read (a chunk holds < 2^32 lines)
    bool load_batch(size_t &first_line);
    bool detect_format();
    bool cut_ahead(uint64_t max_bases, uint64_t min_reads, uint64_t hard_max_bases, uint64_t &approx_bases);
    const char *line(size_t i) const { return (mem_ ? mem_ : arena_.data()) + lines_[i].off; }
    LineSource src_;
    const char *mem_ = nullptr;         // memory-mapped input: spans point into it
    std::string arena_;                 // stream input: the lines of the current CHUNK, back to back
    PodVec<Span> lines_;                // lines of the current chunk
    PodVec<Rec> recs_;
    std::vector<uint32_t> hdr_line_, rec_batch_;   // cut_ahead's scratch: header line of each read, its batch
    std::vector<uint64_t> batch_pending_;          // ... and, closed-form cut, the sequence bytes of each batch
    size_t min_reads_;
    uint64_t size_hint_ = 0;
    unsigned threads_ = 0;              // 0 = hardware concurrency (at most 16)
    std::unique_ptr<WorkerPool> pool_;
    void make_pool();
    PhaseTimes times_;
    double warm_input_s_ = 0;
    size_t skip_until_ = 0;   // cut_ahead makes no pass while the input position is before this byte (an irregular line close ahead)
    size_t bad_at_ = 0;       // the byte behind an irregular line a pass has found and the cut has not crossed yet (0: none known)
    size_t lines_cap_ = 0;    // lines a pass may consider (0: the chunk's reach): small behind an irregular line, doubling with clean passes
    const bool no_fast_cut_ = std::getenv("MOVI_NO_FAST_CUT") != nullptr;   // every batch line by line (tests: both cuts must agree)
    int format_ = -1;                   // -1 unknown, 0 FASTA, 1 FASTQ
    uint32_t batch_counter_ = 0;
};

// Order in which ReadProcessor emits the reads of `rs` with `strands` strands and one
// thread: within each reference batch, strands 0..S-1 take the first S reads; every round
// each live strand consumes `1` unit of its read; a strand that finishes writes its record at
// once and takes the batch's next read.  `cost[i]` = rounds read i needs (its length for PML).
std::vector<uint32_t> strand_order(const ReadSet &rs, const std::vector<uint64_t> &cost, size_t strands, WorkerPool *pool = nullptr);

}  // namespace movi_host
