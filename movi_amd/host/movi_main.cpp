// movi_main.cpp -- the `movi` host binary of the MI355X engine: `movi query` and
// `movi view` with the reference's command line, input rules and output bytes
// (driver: src/movi.cpp:221-400 query(), :402-503 view(), :561-748 main()).
//
// All query arithmetic happens on the GPU behind the C-ABI of include/movi_hip.h; this
// file only parses, batches, orders and writes.  One binary serves both index modes (the
// reference ships one binary per mode and a launcher, src/movi_launcher.cpp:244-254).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/movi_hip.h"
#include "nulldb.hpp"
#include "options.hpp"
#include "output.hpp"
#include "reads.hpp"

using namespace movi_host;

namespace {

struct EngineError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void check(int rc, const char *what) {
    if (rc != MOVI_OK) throw EngineError(std::string(what) + ": " + movi_last_error());
}

const char *index_type_name(uint32_t mode) {                          // program(), src/utils.cpp:10-40
    return mode == MOVI_MODE_REGULAR_THRESHOLDS ? "regular-thresholds"
           : mode == MOVI_MODE_SAMPLED_THRESHOLDS ? "sampled-thresholds"
           : mode == MOVI_MODE_SAMPLED ? "sampled"
           : mode == MOVI_MODE_REGULAR ? "regular"
           : mode == MOVI_MODE_BLOCKED ? "blocked" : "blocked-thresholds";
}

// Contiguous shards of [0, n) balanced by bases.
std::vector<size_t> shard_bounds(const ReadSet &rs, int parts) {
    std::vector<size_t> b(1, 0);
    const uint64_t total = rs.offsets.back();
    for (int p = 1; p < parts; p++) {
        const uint64_t target = total * p / parts;
        size_t i = std::lower_bound(rs.offsets.begin(), rs.offsets.end(), target) - rs.offsets.begin();
        b.push_back(std::min(std::max(i, b.back()), rs.size()));
    }
    b.push_back(rs.size());
    return b;
}

// Rounds the strand scheduler spends on a read in prefetch mode (see reads.hpp).  PML: one
// round per base.  Count: ReadProcessor::backward_search (src/read_processor.cpp:1096-1175)
// returns in the round of the last LF when it stops at position 0 or at an illegal base, and
// one round later when the interval became empty.
uint64_t count_rounds(const ReadSet &rs, size_t i, uint64_t matched, const uint8_t *code_of) {
    const uint64_t len = rs.len(i);
    const uint8_t *R = rs.bases.data() + rs.offsets[i];
    if (len == 0 || code_of[R[len - 1]] == 0xFF) return 1;
    if (matched >= len) return std::max<uint64_t>(len - 1, 1);
    if (code_of[R[len - matched - 1]] == 0xFF) return std::max<uint64_t>(matched - 1, 1);
    return std::max<uint64_t>(matched, 1);
}

// ZML: one round per ReadProcessor::backward_search call (src/read_processor.cpp:1096-1175 driven
// by :667-714).  A round either extends the phrase by one base (and also ends it there when the
// next base is illegal or position 0 is reached) or finds the interval empty / the first
// extension impossible and ends the phrase without moving.  Derived from the match lengths:
// base p-1 extended the phrase of p iff z[p-1] == z[p] + 1 (or both sit at the u16 clamp).
uint64_t zml_rounds(const ReadSet &rs, size_t i, const uint16_t *z, const uint8_t *code_of) {
    const int64_t len = (int64_t)rs.len(i);
    const uint8_t *R = rs.bases.data() + rs.offsets[i];
    auto legal = [&](int64_t p) { return code_of[R[p]] != 0xFF; };
    auto val = [&](int64_t p) { return z[len - 1 - p]; };
    int64_t pos = len - 1;
    while (pos >= 0 && !legal(pos)) pos--;                             // reset_backward_search at load
    if (pos < 0) return 1;
    uint64_t rounds = 0;
    bool first = true;
    for (;;) {
        rounds++;
        if (pos == 0) return rounds;                                   // first iteration at position 0 ends the read
        bool ended = false;
        if (first && !legal(pos - 1)) ended = true;                    // :1119-1121, no LF in this round
        first = false;
        if (!ended) {
            const bool extended = legal(pos - 1) && (val(pos - 1) == (uint16_t)(val(pos) + 1) ||
                                                     (val(pos) == 65535 && val(pos - 1) == 65535));
            if (!extended) {
                ended = true;                                          // empty interval: :1142-1146
            } else {
                pos--;
                if (pos == 0) return rounds;                           // :1148-1156
                if (!legal(pos - 1)) ended = true;                     // :1159-1163, same round
            }
        }
        if (ended) {                                                   // :688-700
            pos--;
            while (pos >= 0 && !legal(pos)) pos--;
            if (pos < 0) return rounds + 1;                            // one more round on the empty range
            first = true;
        }
    }
}

// ---- the three-stage host pipeline of `movi query`: parse (thread) -> GPU calls (caller) -> order + write (thread).
// Three Jobs circulate; each carries a chunk of reads and everything the engine returns for it, so the parser can fill
// chunk k+1 and the writer can drain chunk k-1 while the GPU works on chunk k.  Record order is untouched: chunks are
// parsed, processed and written strictly in input order.
// Result buffer of one job: grow-only, never zero-filled (the engine writes every entry) -- but its pages are TOUCHED,
// by several threads, when it is allocated: a device-to-host copy into memory that has never been touched takes the
// driver's page-fault path and ran at a third of the PCIe rate (1 Gbase of 10 kbp reads: 0.24 s of GPU calls, 0.10 s
// with resident pages).
struct MlBuf {
    uint16_t *p = nullptr;
    size_t cap = 0;
    bool pinned = false;                                              // page-locked (movi_host_alloc): the engine overlaps its transfers
    ~MlBuf() { release(); }
    void release() {
        if (p) { if (pinned) movi_host_free(p); else std::free(p); }
        p = nullptr;
        cap = 0;
    }
    uint16_t *data() const { return p; }
    void ensure(size_t n, bool want_pinned) {
        if (n <= cap) return;
        release();
        cap = n + (n >> 4);
        const size_t bytes = ((cap * sizeof(uint16_t) + (2u << 20) - 1) >> 21) << 21;
        void *q = nullptr;
        if (want_pinned && movi_host_alloc(bytes, &q) == MOVI_OK && q) {   // resident and pinned as it comes
            p = static_cast<uint16_t *>(q);
            pinned = true;
            return;
        }
        pinned = false;
        if (posix_memalign(&q, 2u << 20, bytes) != 0 || !q) { cap = 0; throw std::bad_alloc(); }
        p = static_cast<uint16_t *>(q);
        madvise(q, bytes, MADV_HUGEPAGE);                              // fewer, larger faults where THP is on
        const unsigned T = bytes >= (64u << 20) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        auto touch = [&](unsigned t) {
            volatile uint8_t *b = static_cast<volatile uint8_t *>(q);
            for (size_t off = bytes / T * t, end = t + 1 == T ? bytes : bytes / T * (t + 1); off < end; off += 4096) b[off] = 0;
        };
        if (T == 1) touch(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; t++) th.emplace_back(touch, t);
            for (auto &x : th) x.join();
        }
    }
};

void *pinned_alloc(size_t bytes) {
    void *q = nullptr;
    return movi_host_alloc(bytes, &q) == MOVI_OK ? q : nullptr;
}
void pinned_free(void *q) { movi_host_free(q); }

// The chunks' read buffers: page-locked up to the size of an ordinary chunk -- allocated by the parser's warm-up while the index loads --,
// plain memory beyond it (a chunk of long reads grows to a GB: page-locking that much inside the run costs more than it gives).
// Why (tools/r05_stall.sh, the engine's MOVI_TRACE_HOST_CALLS=1): the upload of a chunk's 33.5 MB from PAGEABLE memory takes 0.6 ms when
// nothing else runs, 1.1 - 2.9 ms beside the parser's threads, and now and then 20 - 30 ms (one run in five on some boxes: the runtime's
// pageable path against the parser's page faults); from page-locked memory it is one direct DMA.  25 runs each on one box, ms per
// 150 Mbases: --no-output 14.6 - 38.7 (median 18.4) -> 12.4 - 19.9 (13.5); with the BPF file 32.0 - 69.8 (35.7) -> 31.0 - 44.9 (32.8).
// The calls are kept whole ("host_overlap" 0: the engine would otherwise cut a call on page-locked buffers into overlapped pieces,
// which chunk-sized calls never earn back -- the MOVI_PINNED=1 measurement above).  MOVI_PINNED=0: pageable buffers (A/B).
constexpr size_t kPinnedChunkLimit = 96u << 20;
std::mutex g_pinned_m;
std::vector<void *> g_pinned_ptrs;
void *chunk_alloc(size_t bytes) {
    if (bytes <= kPinnedChunkLimit) {
        void *q = pinned_alloc(bytes);
        if (q) { std::lock_guard<std::mutex> g(g_pinned_m); g_pinned_ptrs.push_back(q); return q; }
    }
    return std::malloc(bytes);
}
void chunk_free(void *q) {
    {
        std::lock_guard<std::mutex> g(g_pinned_m);
        auto it = std::find(g_pinned_ptrs.begin(), g_pinned_ptrs.end(), q);
        if (it != g_pinned_ptrs.end()) { g_pinned_ptrs.erase(it); movi_host_free(q); return; }
    }
    std::free(q);
}

struct Job {
    ReadSet rs;
    MlBuf pml;                                                        // PML / ZML values, emission order per read
    std::vector<uint16_t> log_ff, log_scan;                           // --logs: per-base fast-forwards / scan rows
    std::vector<uint64_t> matched, counts;                            // --count
    std::vector<uint8_t> err;                                         // per-read error byte
    RawBytes original;                                                // reads as given (--filter after --ignore-illegal-chars 1)
    std::vector<uint32_t> bins_above, bins_below;                     // verdict-only classification
    std::vector<uint64_t> bins_sum;
    bool verdict_only = false;
};

template <typename T>
class HandOff {                                                       // unbounded FIFO between two pipeline stages
public:
    void push(T v) {
        { std::lock_guard<std::mutex> g(m_); q_.push_back(v); }
        cv_.notify_all();
    }
    // false once close() was called and the queue is drained -- or at once after abandon()
    bool pop(T &v) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty() || closed_ || abandoned_; });
        if (abandoned_ || q_.empty()) return false;
        v = q_.front();
        q_.erase(q_.begin());
        return true;
    }
    void close() {
        { std::lock_guard<std::mutex> g(m_); closed_ = true; }
        cv_.notify_all();
    }
    void abandon() {
        { std::lock_guard<std::mutex> g(m_); abandoned_ = true; }
        cv_.notify_all();
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<T> q_;
    bool closed_ = false, abandoned_ = false;
};

// A regular read file is memory-mapped: the parser then cuts lines and batches without copying a byte, and its worker
// threads copy the sequences straight from the page cache into the chunk; pipes and stdin go through the stream.
struct InputMapping {
    void *p = MAP_FAILED;
    size_t n = 0;
    ~InputMapping() { if (p != MAP_FAILED) munmap(p, n); }
};
std::unique_ptr<BatchReader> open_reader(const std::string &path, std::istream &in, size_t min_reads, InputMapping &map) {
    if (path != "-" && !std::getenv("MOVI_NO_MMAP")) {
        const int fd = open(path.c_str(), O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            map.n = (size_t)sb.st_size;
            map.p = mmap(nullptr, map.n, PROT_READ, MAP_PRIVATE, fd, 0);
            // (populating the mapping ahead of the parser from a helper thread -- MADV_POPULATE_READ, 16 MB at a time -- was
            // measured and dropped: it contends with the parser's own faults for the address-space lock; scan 14 -> 23 ms)
            if (map.p != MAP_FAILED) madvise(map.p, map.n, MADV_SEQUENTIAL);
        }
        if (fd >= 0) close(fd);
    }
    if (map.p != MAP_FAILED) return std::unique_ptr<BatchReader>(new BatchReader(static_cast<const char *>(map.p), map.n, min_reads));
    return std::unique_ptr<BatchReader>(new BatchReader(in, min_reads));
}

int run_query(const Options &o) {
    int n_dev = 0;
    check(movi_device_count(&n_dev), "no usable GPU");
    if (n_dev < 1) throw EngineError("no usable GPU: the MI355X engine has no CPU fallback");
    // test hook for 1-GPU boxes: MOVI_SHARE_GPU=1 puts every logical GPU of --gpus N on --device
    const bool share_gpu = std::getenv("MOVI_SHARE_GPU") && std::string(std::getenv("MOVI_SHARE_GPU")) == "1";
    auto dev_of = [&](int g) { return share_gpu ? o.device : o.device + g; };
    if (!share_gpu && o.device + o.gpus > n_dev)
        throw EngineError("requested devices " + std::to_string(o.device) + ".." + std::to_string(o.device + o.gpus - 1) +
                          " but only " + std::to_string(n_dev) + " visible");
    auto t0 = std::chrono::steady_clock::now();
    // Round 5: the read file is mapped and the parser WARMED while the index loads (BatchReader::warm_up: the worker pool, the first
    // window's newline scan -- and with it the mapping's page faults --, the three circulating chunks' buffers sized and first
    // touched by the pool's pinned threads).  A run of 1 M x 150 bp is 4.5 chunks, three of them used to be parsed into fresh memory:
    // chunk 1 took 7.4 ms, chunks 2 - 3 3.9 - 4.3 ms, a warm chunk 2.8 ms (tools/r05_cli.sh).  (Touching the buffers from helper
    // threads of THIS thread instead made the parse slower, 0.024 - 0.028 s against 0.020 s: first touch by threads on another NUMA node
    // than the parser's pinned pool.)  Round 6 (advisor finding): the warm-up no longer reads the input -- round 5's scanned the first
    // window while the index loaded, outside the "processing the reads" clock the reference keeps its whole parse inside -- : its tables
    // are sized from the line density of the file's first MiB (the probe below), and the first window's scan, page faults included, is
    // the first chunk's own, inside the clock.  What stays outside is set-up of the engine's host side -- the worker pool, sizing and
    // page-locking the three circulating chunks' buffers --, like the device staging reserved with the index.  (Should a warm-up ever scan
    // -- no line density known --, its seconds are added to the reported time.)  MOVI_NO_WARM_PARSER=1: A/B.
    double warm_seconds = 0;
    std::ifstream file_in;
    std::istream *in = &std::cin;
    InputMapping map;
    std::unique_ptr<BatchReader> reader_ptr;
    Job jobs[3];
    uint64_t chunk_bases = 1ull << 25, chunk_min_reads = 1ull << 15, chunk_hard_max = 1ull << 30;
    if (const char *e = std::getenv("MOVI_CHUNK_BASES")) {             // test hook: many small chunks
        chunk_bases = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
        chunk_min_reads = 1;
    }
    const std::string pinned_env = std::getenv("MOVI_PINNED") ? std::getenv("MOVI_PINNED") : "";
    const bool pin_buffers = pinned_env == "1";
    bool pin_chunks = false;                                           // read buffers page-locked at warm-up, one direct upload per chunk (chunk_alloc)
    std::thread warmer;
    struct WarmJoin { std::thread &t; ~WarmJoin() { if (t.joinable()) t.join(); } } warm_join{warmer};
    if (o.read_file != "-") {
        file_in.open(o.read_file.c_str());
        if (file_in.good()) {                                          // (a missing file is reported where it always was: after the index)
            in = &file_in;
            reader_ptr = open_reader(o.read_file, *in, o.prefetch ? 4 * o.strands : 1, map);   // src/movi.cpp:283, :326
            const bool warm = map.p != MAP_FAILED && !pin_buffers && !std::getenv("MOVI_NO_WARM_PARSER") && !std::getenv("MOVI_CHUNK_BASES");
            // (only where the warm-up allocates them, outside the run -- and only for short lines: a chunk of long reads outgrows the
            // warmed buffers at once, and giving page-locked memory back inside the run cost 100 k x 10 kbp 20 %: 0.094 - 0.135 -> 0.134 - 0.159 s,
            // tools/r05_pin_ab.sh)
            bool short_lines = false;
            double lines_per_byte = 0.0;                               // (of the file's first MiB: sizes the warm-up's tables without a scan of the input)
            if (warm) {
                const char *m = static_cast<const char *>(map.p);
                const size_t probe = std::min<size_t>(map.n, 1u << 20);
                size_t nl = 0;
                for (const char *q = m; (q = static_cast<const char *>(std::memchr(q, '\n', (size_t)(m + probe - q)))) != nullptr; ++q) nl++;
                short_lines = nl * 1024 >= probe;                      // at least a line per KiB (the scan-ahead's own test for long reads)
                lines_per_byte = probe ? (double)(nl + 1) / (double)probe : 0.0;
            }
            pin_chunks = warm && short_lines && pinned_env != "0";
            if (pin_chunks)
                for (Job &j : jobs) j.rs.bases.set_allocator(chunk_alloc, chunk_free);
            if (warm)
                warmer = std::thread([&] {
                    ReadSet *sets[3] = {&jobs[0].rs, &jobs[1].rs, &jobs[2].rs};
                    try { reader_ptr->warm_up(sets, 3, chunk_bases, lines_per_byte); } catch (...) { /* no memory for it: the chunks allocate as they come */ }
                    warm_seconds = reader_ptr->warm_input_seconds();           // (read after the join) the part that read the input
                });
        }
    }
    std::vector<movi_index_t *> handles((size_t)o.gpus, nullptr);
    struct Closer {
        std::vector<movi_index_t *> &h;
        ~Closer() { for (auto *x : h) movi_index_destroy(x); }
    } closer{handles};
    if (!o.gpus_given) {
        check(movi_index_load(o.device, o.index_dir.c_str(), &handles[0]), "loading the index");
    } else if (!share_gpu) {
        // --gpus N: the index file is mapped once, its rows cross PCIe once (to the first GPU) and reach the others
        // through ONE RCCL broadcast over xGMI; every GPU then builds its resident layout (include/movi_hip.h)
        std::vector<int> devs((size_t)o.gpus);
        for (int g = 0; g < o.gpus; g++) devs[(size_t)g] = dev_of(g);
        check(movi_index_load_replicated(o.index_dir.c_str(), devs.data(), o.gpus, handles.data()), "replicating the index");
    } else {
        // MOVI_SHARE_GPU=1 (test hook for 1-GPU boxes: every logical GPU of --gpus N is the same device, which RCCL cannot
        // serve as N ranks): N independent loads, so that the read sharding and the per-GPU host threads can be exercised
        for (int g = 0; g < o.gpus; g++) check(movi_index_load(dev_of(g), o.index_dir.c_str(), &handles[(size_t)g]), "loading the index");
    }
    if (o.seg_len >= 0)
        for (auto *hd : handles) check(movi_set_option(hd, "seg_len", o.seg_len), "--seg-len");
    // (the ZML parse does not walk on the look-ahead rows unless "zml_ahead" asks for it: `--zml --ahead-rows 1` builds nothing)
    if (o.zml && o.ahead_rows == 1)
        std::cerr << "[movi] --ahead-rows 1 is ignored with --zml: the ZML parse walks on the plain rows (the engine's \"zml_ahead\" option is opt-in and no faster).\n";
    if (o.ahead_rows >= 0 && !(o.zml && o.ahead_rows == 1))
        for (auto *hd : handles) check(movi_set_option(hd, "ahead_rows", o.ahead_rows), "--ahead-rows");
    if (pin_chunks)
        for (auto *hd : handles) check(movi_set_option(hd, "host_overlap", 0), "host_overlap");
    // (the command's own pipeline keeps the host's cores busy -- parser pool, record order, BPF gather and write --: the PML vector comes
    // down as it is instead of as reset masks for the engine's worker threads to expand beside them; tools/r06_l.sh: GPU calls 13 - 15 ms
    // against 35 - 40 ms in three runs of five.  MOVI_PML_VIA_MASK=1 forces the mask route: the parity suite's re-run.)
    if (!std::getenv("MOVI_PML_VIA_MASK")) {
        const char *hm = std::getenv("MOVI_HOST_MASKS");                      // (A/B: -1 = the engine's own policy, 1 = every call)
        for (auto *hd : handles) (void)movi_set_option(hd, "host_masks", hm ? std::atoi(hm) : 0);
    }
    // Round 5: the handles' derived tables -- top-of-walk / interval table, look-ahead rows (16 bytes per row: 16 GB and ~0.3 s of
    // allocation + build for a 1 B-row index), row-start checkpoints -- are part of LOADING THE INDEX (movi_index_prepare), not of the
    // first chunk's GPU call: rounds 3 - 4 built them inside the read-processing clock, behind the first chunk's parse, which a
    // 14 M-row index hides and a 1 B-row one does not (bench.py big_table.cli_path: 0.31 s of "processing" for 0.03 s of work).
    for (auto *hd : handles) {                                         // (errors here are not the query's: the real calls report)
        if (o.pml && o.logs) continue;                                 // --logs runs on the first kernel, which uses none of the derived tables
        (void)movi_index_prepare(hd, o.pml ? MOVI_PREPARE_PML : (o.zml ? MOVI_PREPARE_ZML : MOVI_PREPARE_COUNT), nullptr, nullptr);
        // ... and so is the device staging of a chunk's host call (three hipMallocs: 1.2 ms of the first chunk's 3 ms call otherwise):
        // a chunk's bases with the slack of its last batch, its result vector when one comes back, reads down to 64 bases long
        const int64_t cb = (int64_t)std::min<uint64_t>(chunk_bases + (chunk_bases >> 3), 1ull << 31) / (o.gpus > 0 ? o.gpus : 1);
        // (a reservation that does not fit is not an error: the staging then grows inside the first call, as it did before round 5)
        bool reserved = movi_set_option(hd, "reserve_host_bases", cb) == MOVI_OK;
        reserved = movi_set_option(hd, "reserve_host_reads", cb / 64) == MOVI_OK && reserved;
        if (o.ml() && (o.write_output_allowed() || o.classify) && !(o.pml && o.classify && !o.write_output_allowed()))
            reserved = movi_set_option(hd, "reserve_host_results", cb) == MOVI_OK && reserved;
        if (!reserved && o.verbose) std::cerr << "[movi] The device staging could not be reserved up front (" << movi_last_error() << "); it grows with the first chunk.\n";
    }
    movi_index_desc_t desc;
    check(movi_index_get_desc(handles[0], &desc), "index description");
    const std::string index_type = index_type_name(desc.mode);
    std::cerr << "[movi] The " << index_type << " index is being used (r = " << desc.r << ").\n";
    std::cerr << "[movi] Time measured for loading the index: "
              << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s\n";

    // input: file or stdin (setup_input_file, src/movi.cpp:106-118) -- opened above, while the index loaded
    if (o.read_file != "-" && !file_in.good()) throw std::runtime_error("The input file " + o.read_file + " does not exist.");

    // outputs (open_output_files, src/utils.cpp:319-384)
    Classifier classifier;
    // stream buffers first: they must outlive the streams that flush through them on destruction
    std::vector<char> out_buf2(1u << 20);
    std::ofstream report_file, matches_file;
    std::unique_ptr<WorkerPool> bpf_pool;                             // the writer stage's helper threads (record order, BPF gather; built by the writer thread, outlives mls_file)
    BpfWriter mls_file;
    matches_file.rdbuf()->pubsetbuf(out_buf2.data(), (std::streamsize)out_buf2.size());
    std::ostream *report = nullptr;
    if (o.classify) {
        classifier.load_null_db(o.index_dir, o.query_type(), o.verbose);
        if (!o.filter) {                                              // src/classifier.cpp:39-61
            if (!o.write_stdout) {
                const std::string name = o.read_file + "." + index_type + "." + o.query_type() + ".report";
                std::cerr << "[movi] Report file name: " << name << "\n";
                report_file.open(name);
                report = &report_file;
            } else {
                report = &std::cout;
            }
            classifier.write_report_header(*report);
        }
    }
    const bool open_files = (!o.write_stdout || o.classify) && o.write_output_allowed();
    if (open_files) {
        std::string prefix = !o.out_file.empty() ? o.out_file : o.read_file + "." + index_type;
        prefix += "." + o.query_type();
        if (o.ml()) {
            mls_file.open(prefix + ".bpf", 16);
        } else {
            matches_file.open(prefix + ".matches");
            if (!matches_file.good()) throw std::runtime_error("Failed to open the output file: " + prefix + ".matches");
        }
    }

    // --logs (src/utils.cpp:376-382: opened with the other output files, written by output_logs :268-289 per read)
    const bool logs = o.logs && o.pml && open_files;
    std::ofstream costs_file, scans_file, ff_file;
    if (logs) {
        std::string prefix = (!o.out_file.empty() ? o.out_file : o.read_file + "." + index_type) + "." + o.query_type();
        costs_file.open(prefix + ".costs");
        scans_file.open(prefix + ".scans");
        ff_file.open(prefix + ".fastforwards");
        if (!costs_file.good() || !scans_file.good() || !ff_file.good()) throw std::runtime_error("Failed to open the log files: " + prefix + ".costs/.scans/.fastforwards");
    } else if (o.logs) {
        std::cerr << "[movi] --logs is only collected for PML queries that write their output to files; ignored here.\n";
    }

    if (warmer.joinable()) warmer.join();
    auto t1 = std::chrono::steady_clock::now();
    if (!reader_ptr) reader_ptr = open_reader(o.read_file, *in, o.prefetch ? 4 * o.strands : 1, map);   // (stdin) src/movi.cpp:283, :326
    BatchReader &reader = *reader_ptr;
    // chunks of >= 2^25 bases and >= 2^15 reads (long reads: up to 2^30 bases): one GPU lane walks one read, so a chunk needs
    // READS to fill the lanes (2^25 bases of 150 bp reads = 224 k reads: more than the 147 k lanes the PML kernel keeps
    // resident) -- but the host stages (text parsing ~1.4 GB/s, BPF writing) are what bounds the command, and they only overlap
    // the GPU calls and each other across chunks: better several half-filled launches than one full one
    uint64_t reads_done = 0, bases_done = 0;
    double gpu_seconds = 0;

    // Page-locked chunk buffers (MOVI_PINNED=1): with the reads and the result vector in page-locked memory the engine's
    // *_host entry points cut a call into pieces whose upload, walk and download overlap (include/movi_hip.h).  Measured
    // in this command and NOT the default (profiles/r03_cli_path.txt): its chunks (2^25 bases of short reads, 2^15 long
    // reads) are a quarter of what the overlapped path needs to pay -- each piece must still fill the GPU -- and a run makes
    // a handful of calls, so the path's one-time set-up (six streams with staging) is never earned back: GPU calls of
    // 1 M x 150 bp 0.021 s pageable, 0.084 s page-locked; 100 k x 10 kbp 0.15 s / 0.31 s.  (Round 5: page-locked read buffers with
    // the calls NOT cut into pieces -- one direct upload per chunk -- 8.1 - 9.5 against 8.6 - 9.9 ms of GPU calls per 150 Mbases, the
    // command 12.0 - 13.1 against 12.5 - 12.6 ms, and with the BPF file 50 - 57 against 32 - 36 ms: dropped.  tools/r05_scan.sh)  The command's own pipeline
    // (parse | GPU calls | order + write, three chunks in flight) already overlaps the transfers with the other stages.
    uint64_t input_bytes = 0;
    if (map.p != MAP_FAILED) input_bytes = map.n;
    auto pin_this_chunk = [&](uint64_t) { return pin_buffers; };
    HandOff<Job *> free_q, parsed_q, done_q;
    (void)input_bytes;
    if (pin_buffers)
        for (Job &j : jobs) j.rs.bases.set_allocator(pinned_alloc, pinned_free);
    for (Job &j : jobs) free_q.push(&j);
    double parse_seconds = 0, write_seconds = 0;
    std::vector<double> chunk_parse_s, chunk_gpu_s;                    // --verbose: the first chunks' stage times one by one
    std::vector<BatchReader::PhaseTimes> chunk_phases;                 // (cumulative parser phases after each chunk)
    std::exception_ptr parse_error, write_error;
    std::mutex err_m;

    // (round 5, measured and dropped: keeping the parser's threads off the cache domain of the GPU-call and writer threads --
    // inside the command every parser phase runs 2 x slower than in tools/parse_bench -- changed nothing: 0.027 - 0.036 s against
    // 0.026 s per 150 Mbases.  tools/r05_cli.sh)
    // ---- stage 1: parse
    std::thread parser([&] {
        try {
            reader.adopt_pool();                                       // (a pool built by the warm-up: this thread joins its cache domain)
            Job *j = nullptr;
            // (round 5, measured and dropped: smaller FIRST chunks -- a quarter, then half of the steady size, to shorten the
            // pipeline's fill -- tripled the GPU stage, 0.011 -> 0.035 s per 150 Mbases: every growth of a chunk re-allocates the
            // engine's device staging, and hipFree waits for the device.  tools/r05_cli.sh.  Tried again with the staging reserved up
            // front ("reserve_host_*"): nothing to see through the box's own noise, 15 runs each, BPF 30 - 40 against 30 - 39 and 32 - 35
            // against 34 - 37 ms, --no-output worse in one round and better in the other)
            while (free_q.pop(j)) {
                const auto tp = std::chrono::steady_clock::now();
                const bool more = reader.next_chunk(j->rs, chunk_bases, chunk_min_reads, chunk_hard_max);
                const double dtp = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
                parse_seconds += dtp;
                if (chunk_parse_s.size() < 64) { chunk_parse_s.push_back(dtp); chunk_phases.push_back(reader.phase_times()); }
                if (!more) break;
                parsed_q.push(j);
            }
        } catch (...) {
            std::lock_guard<std::mutex> g(err_m);
            parse_error = std::current_exception();
        }
        parsed_q.close();
    });

    // ---- stage 3: record order + writers (everything that touches the output streams lives on this thread)
    const bool bpf_serial = std::getenv("MOVI_BPF_SERIAL") != nullptr;  // A/B: the one-thread gather-and-write loop (BpfWriter::append(records))
    unsigned bpf_threads = 4;                                          // (8 measured the same: the stage is bound by its write()s)
    if (const char *e = std::getenv("MOVI_BPF_THREADS")) bpf_threads = (unsigned)std::max(1, std::atoi(e));
    auto write_job = [&](Job &job) {
        ReadSet &rs = job.rs;
        const size_t n = rs.size();
        const bool verdict_only = job.verdict_only;
        // nothing to write for this chunk (`--no-output` without a report or a filter): no record order needed either
        if (!o.write_output_allowed() && !(o.classify && (o.filter || report))) return;
        // record order: strand scheduler emulation in prefetch mode, file order otherwise
        if (!bpf_pool && n >= 4096) bpf_pool.reset(new WorkerPool(bpf_threads));   // this stage's helpers (order, BPF gather)
        std::vector<uint32_t> order;
        if (o.prefetch) {
            std::vector<uint64_t> cost(n);
            auto cost_of = [&](size_t i) -> uint64_t {
                return o.pml ? rs.len(i)
                     : o.zml ? zml_rounds(rs, i, job.pml.data() + rs.offsets[i], desc.code_of)
                             : count_rounds(rs, i, job.matched[i], desc.code_of);
            };
            if (bpf_pool) {
                const unsigned P = bpf_pool->size() * 2u;
                bpf_pool->run(P, [&](unsigned p) { for (size_t i = n * p / P, e = n * (p + 1) / P; i < e; i++) cost[i] = cost_of(i); });
            } else {
                for (size_t i = 0; i < n; i++) cost[i] = cost_of(i);
            }
            order = strand_order(rs, cost, o.strands, bpf_pool.get());
        } else {
            order.resize(n);
            for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
        }
        std::vector<BpfWriter::Record> bpf;                           // the chunk's records, in emission order
        const bool to_bpf = o.ml() && o.write_output_allowed() && !o.write_stdout_enabled();
        if (to_bpf && !o.classify && !logs && !bpf_serial) {
            // the plain BPF file: records gathered from the chunk's arrays by a pool of this thread's, written behind it (output.cpp)
            if (!bpf_pool) bpf_pool.reset(new WorkerPool(bpf_threads));
            mls_file.append(BpfWriter::Chunk{order.data(), n, rs.offsets.data(), job.pml.data(), rs.id_off.data(), rs.id_bytes.data()}, *bpf_pool);
            return;
        }
        if (to_bpf) bpf.reserve(n);
        if (o.ml() && !o.classify && o.write_output_allowed() && o.write_stdout_enabled() && rs.bases.size() >= (1u << 22)) {
            // plain `--stdout`: the text of a chunk is formatted by worker threads (ranges balanced by bases), written in order
            const unsigned T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            std::vector<std::string> txt(T);
            std::vector<size_t> cut(T + 1, n);
            cut[0] = 0;
            uint64_t acc = 0;
            unsigned t = 1;
            for (size_t k = 0; k < n && t < T; k++) {
                acc += rs.len(order[k]);
                while (t < T && acc >= rs.bases.size() * (uint64_t)t / T) cut[t++] = k + 1;
            }
            std::vector<std::thread> th;
            for (unsigned u = 0; u < T; u++)
                th.emplace_back([&, u] {
                    for (size_t k = cut[u]; k < cut[u + 1]; k++) {
                        const uint32_t i = order[k];
                        append_stdout_pmls(txt[u], rs.id(i), job.pml.data() + rs.offsets[i], rs.len(i));
                    }
                });
            for (auto &x : th) x.join();
            for (unsigned u = 0; u < T; u++) std::cout.write(txt[u].data(), (std::streamsize)txt[u].size());
            return;
        }
        std::string count_txt;
        if (!o.ml() && o.write_output_allowed()) count_txt.reserve(n * 32);
        for (uint32_t i : order) {
            const uint64_t len = rs.len(i);
            if (o.ml()) {
                const uint16_t *p = verdict_only ? nullptr : job.pml.data() + rs.offsets[i];
                if (o.classify) {                                     // write_mls, src/read_processor.cpp:565-578
                    const bool found = verdict_only
                        ? (job.bins_above[i] / (job.bins_above[i] + job.bins_below[i] + 0.0) > 0.50)      // classifier.cpp:119
                        : classifier.classify(rs.id(i), p, len, o.bin_width, o.write_output_allowed() ? report : nullptr);
                    if (o.filter && !o.no_output && (found != o.invert)) {
                        const uint8_t *seq = (job.original.empty() ? rs.bases.data() : job.original.data()) + rs.offsets[i];
                        std::cout << ">" << rs.id(i) << "\n";
                        std::cout.write(reinterpret_cast<const char *>(seq), (std::streamsize)len);
                        std::cout << "\n";
                    }
                }
                if (o.write_output_allowed()) {
                    if (o.write_stdout_enabled()) write_stdout_pmls(std::cout, rs.id(i), p, len);
                    else bpf.push_back(BpfWriter::Record{rs.id(i), p, len});
                    if (logs) {                                       // output_logs, src/utils.cpp:268-289
                        // costs: the wall-clock nanoseconds a CPU strand spent per base -- nothing a GPU lane has; zeros
                        std::string tc, ts, tf;
                        for (std::string *t : {&tc, &ts, &tf}) { t->push_back('>'); t->append(rs.id(i)); t->push_back('\n'); }
                        const uint16_t *sc = job.log_scan.data() + rs.offsets[i], *ff = job.log_ff.data() + rs.offsets[i];
                        for (uint64_t k = 0; k < len; k++) {
                            tc += "0 ";
                            ts += std::to_string(sc[k]); ts.push_back(' ');
                            tf += std::to_string(ff[k]); tf.push_back(' ');
                        }
                        costs_file << tc << "\n";
                        scans_file << ts << "\n";
                        ff_file << tf << "\n";
                    }
                }
            } else if (o.write_output_allowed()) {
                append_count_line(count_txt, rs.id(i), len, job.matched[i], job.counts[i]);
            }
        }
        if (!count_txt.empty()) {                                     // the chunk's count lines in one write
            std::ostream &out = o.write_stdout_enabled() ? static_cast<std::ostream &>(std::cout) : matches_file;
            out.write(count_txt.data(), (std::streamsize)count_txt.size());
        }
        if (to_bpf) mls_file.append(bpf);
    };
    std::thread writer([&] {
        Job *j = nullptr;
        while (done_q.pop(j)) {
            bool failed;
            { std::lock_guard<std::mutex> g(err_m); failed = write_error != nullptr; }
            if (!failed) {
                try {
                    const auto tw = std::chrono::steady_clock::now();
                    write_job(*j);
                    write_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                } catch (...) {
                    std::lock_guard<std::mutex> g(err_m);
                    write_error = std::current_exception();
                }
            }
            free_q.push(j);
        }
    });
    // whatever happens below, the two threads are released and joined before the streams and jobs go away
    struct Joiner {
        HandOff<Job *> &free_q, &done_q;
        std::thread &parser, &writer;
        ~Joiner() {
            free_q.abandon();                                         // parser: stop taking jobs
            done_q.close();                                           // writer: drain what was handed over, then stop
            if (parser.joinable()) parser.join();
            if (writer.joinable()) writer.join();
        }
    } joiner{free_q, done_q, parser, writer};

    // ---- warm-up, while the parser works on the first chunk (the derived tables were built with the index, above):
    // one one-read query of the same kind through the host entry point: what the FIRST host call does once -- the handle's
    // device staging, the page-locked block of its small results, the walk kernels' code object (a translation unit of its own
    // since round 5: the builders' launches do not load it) -- cost the first chunk 20 - 30 ms without it (tools/r05_cli.sh:
    // GPU calls of 1 M x 150 bp 0.035 - 0.045 s with the preparation alone, 0.012 - 0.014 s with the warm-up call).
    {
        const uint8_t wb[32] = {'A','C','G','T','A','C','G','T','A','C','G','T','A','C','G','T','A','C','G','T','A','C','G','T','A','C','G','T','A','C','G','T'};
        const uint64_t wo[2] = {0, 32};
        uint16_t wp[32];
        uint64_t wm = 0, wc = 0;
        uint32_t wa = 0, wbl = 0;
        uint64_t wsum = 0;
        uint8_t we = 0;
        for (auto *hd : handles) {                                     // (errors here are not the query's: the real calls report)
            if (o.pml && o.logs) continue;                             // --logs runs on the first kernel, which uses none of the derived tables
            if (o.pml && o.classify && !o.write_output_allowed())
                (void)movi_pml_classify_host(hd, wb, wo, 1, (uint32_t)o.bin_width, classifier.max_value_thr, &wa, &wbl, &wsum, &we, nullptr);
            else if (o.pml) (void)movi_pml_host(hd, wb, wo, 1, wp, &we, nullptr);
            else if (o.zml) (void)movi_zml_host(hd, wb, wo, 1, wp, &we, nullptr);
            else (void)movi_count_host(hd, wb, wo, 1, &wm, &wc, &we, nullptr);
        }
    }

    // ---- stage 2: the GPU calls, in input order
    Job *jp = nullptr;
    while (parsed_q.pop(jp)) {
        Job &job = *jp;
        ReadSet &rs = job.rs;
        const size_t n = rs.size();
        { std::lock_guard<std::mutex> g(err_m); if (write_error) std::rethrow_exception(write_error); }
        if (n == 0) { free_q.push(jp); continue; }
        if (o.reverse)                                                // src/read_processor.cpp:49-51
            for (size_t i = 0; i < n; i++) std::reverse(rs.bases.begin() + rs.offsets[i], rs.bases.begin() + rs.offsets[i + 1]);
        job.original.clear();                                         // --filter echoes the read as given
        if (o.ignore_illegal_chars == 1) {                            // check_alphabet, src/move_structure.cpp:389-395
            if (o.filter) job.original.assign(rs.bases);
            for (auto &c : rs.bases)
                if (desc.code_of[c] == 0xFF) c = 'A';
        }
        // --classify with --filter / --no-output needs verdicts only: the bins are reduced on the
        // GPU and the PML vectors never cross PCIe
        const bool verdict_only = o.pml && o.classify && !o.write_output_allowed();   // PML only: ZML takes the host bins
        job.verdict_only = verdict_only;
        job.bins_above.assign(verdict_only ? n : 0, 0);
        job.bins_below.assign(verdict_only ? n : 0, 0);
        job.bins_sum.assign(verdict_only ? n : 0, 0);
        // `--no-output` without classification: the walk runs, nothing comes back (movi_pml_host with a NULL vector)
        const bool walk_only = o.ml() && !o.classify && !o.write_output_allowed();
        job.pml.ensure(o.ml() && !verdict_only && !walk_only ? rs.bases.size() : 0, pin_this_chunk(rs.bases.size() * 2));
        if (logs) { job.log_ff.resize(rs.bases.size()); job.log_scan.resize(rs.bases.size()); }
        job.matched.assign(o.count ? n : 0, 0);
        job.counts.assign(o.count ? n : 0, 0);
        job.err.assign(n, 0);
        const std::vector<size_t> sb = shard_bounds(rs, o.gpus);
        std::vector<std::string> errors((size_t)o.gpus);
        auto tg = std::chrono::steady_clock::now();
        auto work = [&](int g) {
            const size_t a = sb[g], b = sb[g + 1];
            if (a == b) return;
            int rc;
            if (verdict_only)
                rc = movi_pml_classify_host(handles[g], rs.bases.data(), rs.offsets.data() + a, b - a, (uint32_t)o.bin_width,
                                            classifier.max_value_thr, job.bins_above.data() + a, job.bins_below.data() + a,
                                            job.bins_sum.data() + a, job.err.data() + a, nullptr);
            else if (o.pml && logs)
                rc = movi_pml_logs_host(handles[g], rs.bases.data(), rs.offsets.data() + a, b - a, job.pml.data(), job.log_ff.data(),
                                        job.log_scan.data(), job.err.data() + a, nullptr);
            else if (o.pml)
                rc = movi_pml_host(handles[g], rs.bases.data(), rs.offsets.data() + a, b - a, walk_only ? nullptr : job.pml.data(),
                                   job.err.data() + a, nullptr);
            else if (o.zml)
                rc = movi_zml_host(handles[g], rs.bases.data(), rs.offsets.data() + a, b - a, walk_only ? nullptr : job.pml.data(),
                                   job.err.data() + a, nullptr);
            else
                rc = movi_count_host(handles[g], rs.bases.data(), rs.offsets.data() + a, b - a, job.matched.data() + a,
                                     job.counts.data() + a, job.err.data() + a, nullptr);
            if (rc != MOVI_OK) errors[g] = movi_last_error();
        };
        if (o.gpus == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int g = 0; g < o.gpus; g++) th.emplace_back(work, g);
            for (auto &t : th) t.join();
        }
        { const double dtg = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count(); gpu_seconds += dtg; if (chunk_gpu_s.size() < 64) chunk_gpu_s.push_back(dtg); }
        for (const auto &e : errors)
            if (!e.empty()) throw EngineError(e);
        reads_done += n;
        bases_done += rs.bases.size();
        done_q.push(jp);
    }
    // end of input (or a parse error, which surfaces after every earlier chunk has been written)
    done_q.close();
    writer.join();
    free_q.abandon();
    parser.join();
    if (write_error) std::rethrow_exception(write_error);
    if (parse_error) std::rethrow_exception(parse_error);
    mls_file.close();
    // (the parser's warm-up touched the input before t1: its seconds count as read processing, although they ran beside the index load)
    const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() + warm_seconds;
    std::cerr << "[movi] " << reads_done << " reads are processed.\n";
    std::cerr << "[movi] Time measured for processing the reads: " << total << " s (" << bases_done << " bases; GPU calls "
              << gpu_seconds << " s; input scanned ahead of the clock and added to it: " << warm_seconds << " s)\n";
    if (o.verbose) {
        const BatchReader::PhaseTimes pt = reader.phase_times();
        std::cerr << "[movi] Parser phases: newline scan " << pt.prescan << " s, batch cut " << pt.cut << " s, lengths " << pt.lengths
                  << " s, copy " << pt.copy << " s; " << pt.bulk_reads << " of " << pt.reads << " reads cut in bulk (" << pt.closed_reads << " by the closed form)\n";
    }
    if (o.verbose) {                                                   // chunk by chunk: the first chunk's parse runs alone, the others beside the GPU calls
        std::cerr << "[movi] Chunks: parse";
        for (double x : chunk_parse_s) std::cerr << " " << x;
        std::cerr << " s; GPU calls";
        for (double x : chunk_gpu_s) std::cerr << " " << x;
        std::cerr << " s\n";
        std::cerr << "[movi] Chunk phases (cut [of which waiting for the scan-ahead] / lengths / copy | the scan-ahead helper's own time, ms):";
        BatchReader::PhaseTimes prev;
        for (const auto &p : chunk_phases) {
            std::cerr << " " << (p.cut - prev.cut) * 1e3 << "[" << (p.scan_wait - prev.scan_wait) * 1e3 << "]/" << (p.lengths - prev.lengths) * 1e3 << "/"
                      << (p.copy - prev.copy) * 1e3 << "|" << (p.scan_busy - prev.scan_busy) * 1e3;
            prev = p;
        }
        std::cerr << "\n";
    }
    if (o.verbose && bpf_pool) {
        const BpfWriter::Times bt = mls_file.times();
        std::cerr << "[movi] BPF writer: gather " << bt.gather << " s, waiting for a free slab " << bt.wait << " s (" << bpf_pool->size()
                  << " threads); write() " << bt.write << " s on its own thread\n";
    }
    if (o.verbose)                                                     // the three pipeline stages run side by side: the slowest one bounds the command
        std::cerr << "[movi] Stage times: parse " << parse_seconds << " s, GPU calls " << gpu_seconds << " s, order + write "
                  << write_seconds << " s (page-locked chunk buffers: " << (pin_buffers ? "yes" : (pin_chunks ? "reads" : "no")) << ")\n";
    std::cout.flush();
    return 0;
}

// `movi null`: Classifier::generate_null_statistics (src/classifier.cpp:12-22, driver src/movi.cpp:732-736).
int run_null(const Options &o) {
    const std::string pattern_file = o.index_dir + "/null_reads.fasta";
    if (o.gen_reads) {
        unsigned seed = (unsigned)std::time(nullptr);                 // srand(time(0)), src/utils.cpp:431
        if (const char *e = std::getenv("MOVI_NULL_SEED")) seed = (unsigned)std::strtoul(e, nullptr, 10);
        const size_t n = generate_null_reads(o.ref_file, pattern_file, seed);
        std::cerr << "[movi] " << n << " null reads written to " << pattern_file << "\n";
    }
    movi_index_t *h = nullptr;
    check(movi_index_load(o.device, o.index_dir.c_str(), &h), "loading the index");
    struct Closer { movi_index_t *h; ~Closer() { movi_index_destroy(h); } } closer{h};
    const std::vector<std::string> seqs = read_fasta_sequences(pattern_file);
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets(1, 0);
    for (const std::string &s : seqs) {
        bases.insert(bases.end(), s.begin(), s.end());
        offsets.push_back(bases.size());
    }
    std::vector<uint16_t> ml(bases.size());
    check(o.zml ? movi_zml_host(h, bases.data(), offsets.data(), seqs.size(), ml.data(), nullptr, nullptr)
                : movi_pml_host(h, bases.data(), offsets.data(), seqs.size(), ml.data(), nullptr, nullptr),
          "null statistics");
    const std::vector<uint64_t> values(ml.begin(), ml.end());        // ml_stats, emission order per read
    const NullStats st = compute_null_stats(values);
    const std::string name = o.index_dir + "/movi." + o.query_type() + ".nulldb";
    write_null_db(name, st, values);
    std::cerr << "[movi] Null database statistics: mean_null_stat: " << st.mean << " percentile_value: " << st.percentile_value
              << " (" << st.num_values << " values) -> " << name << "\n";
    return 0;
}

// `movi plan`: the host-side batching + record order, without any GPU work.
int run_plan(const Options &o) {
    std::ifstream file_in;
    std::istream *in = &std::cin;
    if (o.read_file != "-") {
        file_in.open(o.read_file.c_str());
        if (!file_in.good()) throw std::runtime_error("The input file " + o.read_file + " does not exist.");
        in = &file_in;
    }
    InputMapping map;
    std::unique_ptr<BatchReader> reader_ptr = open_reader(o.read_file, *in, o.prefetch ? 4 * o.strands : 1, map);
    BatchReader &reader = *reader_ptr;
    ReadSet rs;
    uint64_t plan_chunk = 1ull << 28;
    if (const char *e = std::getenv("MOVI_CHUNK_BASES")) plan_chunk = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));   // test hook
    std::unique_ptr<WorkerPool> plan_pool;                             // test hook: the record order by a pool, as `movi query`'s writer stage computes it
    if (const char *e = std::getenv("MOVI_PLAN_THREADS")) plan_pool.reset(new WorkerPool((unsigned)std::max(1, std::atoi(e))));
    while (reader.next_chunk(rs, plan_chunk)) {
        std::vector<uint64_t> cost(rs.size());
        for (size_t i = 0; i < rs.size(); i++) cost[i] = rs.len(i);
        std::vector<uint32_t> order;
        if (o.prefetch) order = strand_order(rs, cost, o.strands, plan_pool.get());
        else for (size_t i = 0; i < rs.size(); i++) order.push_back((uint32_t)i);
        for (uint32_t i : order) std::cout << rs.batch_of[i] << "\t" << rs.id(i) << "\t" << rs.len(i) << "\n";
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    std::ios::sync_with_stdio(false);
    try {
        Options o = parse_args(argc, argv);
        if (o.command == "help") {
            std::cerr << usage();
            return 0;
        }
        if (o.command == "view") return view_bpf(o, std::cout);
        if (o.command == "plan") return run_plan(o);
        if (o.command == "null") return run_null(o);
        if (o.command == "build") return run_build(o);
        return run_query(o);
    } catch (const UsageError &e) {
        std::cerr << "Error parsing command line options: " << e.what() << "\n" << usage();
        return 1;
    } catch (const std::exception &e) {                               // src/movi.cpp:744-747
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}
