#include "output.hpp"
#include "reads.hpp"
#include <charconv>

#include <cerrno>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace movi_host {

// append(Chunk)'s other half: a ring of slabs, filled by the caller's pool, written in order by one thread
struct BpfWriter::Async {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    enum { kMaxSlabs = 16 };
    RawBytes slab[kMaxSlabs];
    bool full[kMaxSlabs] = {false};
    unsigned n_slabs = 4, fill_next = 0, write_next = 0, queued = 0;
    uint64_t reserved_to = 0, written_to = 0;                          // fallocate()d / written file offsets (reserve_ahead)
    bool reserve = true;
    bool stop = false;
    bool abort = false;                                                // the destructor during unwinding: queued slabs are dropped, not written
    bool failed = false;                                               // a write() has failed: later slabs are handed back unwritten
    std::exception_ptr err;
    double write_s = 0;
};

BpfWriter::~BpfWriter() {
    if (async_) {                                                      // (an exception is unwinding: what is queued is dropped -- the thread leaves at once)
        { std::lock_guard<std::mutex> g(async_->m); async_->stop = true; async_->abort = true; }
        async_->cv.notify_all();
        if (async_->th.joinable()) async_->th.join();
        // blocks reserved ahead of the writes go back (best effort: the file is known to be incomplete)
        if (fd_ >= 0 && async_->reserved_to > async_->written_to && ::ftruncate(fd_, (off_t)async_->written_to) != 0) { /* nothing more to do */ }
        delete async_;
        async_ = nullptr;
    }
    // reached with the descriptor still open only when an exception is unwinding past the writer (close() is the normal
    // way out and throws on error); a close() error here cannot be thrown, so it is at least said
    if (fd_ >= 0 && ::close(fd_) != 0) std::fprintf(stderr, "[movi] Failed to write the output file: %s\n", path_.c_str());
}

void BpfWriter::open(const std::string &path, uint8_t entry_size) {
    path_ = path;
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd_ < 0) throw std::runtime_error("Failed to open the output file: " + path);
    // struct BPFHeader {u32 magic; u8 major, minor, patch; u8 entry_size; u16 reserved;} is written with sizeof == 12:
    // bytes 10-11 are struct padding (uninitialised in the reference, zero here).  include/utils.hpp:64-82, version
    // numbers include/version.hpp:12-14.
    uint8_t h[12] = {0};
    std::memcpy(h, &kBpfMagic, 4);
    h[4] = 1; h[5] = 0; h[6] = 0;
    h[7] = entry_size;
    ssize_t w;
    do { w = ::write(fd_, h, 12); } while (w < 0 && errno == EINTR);
    if (w != 12) throw std::runtime_error("Failed to write the output file: " + path);
}

// every slab handed over so far is in the file (or its error is thrown here)
void BpfWriter::drain() {
    if (!async_) return;
    std::unique_lock<std::mutex> g(async_->m);
    async_->cv.wait(g, [&] { return async_->queued == 0 || async_->err; });
    if (async_->err) { std::exception_ptr e = async_->err; async_->err = nullptr; std::rethrow_exception(e); }
}

void BpfWriter::close() {
    if (async_) {
        drain();
        { std::lock_guard<std::mutex> g(async_->m); async_->stop = true; }
        async_->cv.notify_all();
        async_->th.join();
    }
    if (fd_ >= 0 && async_ && async_->reserved_to > async_->written_to && ::ftruncate(fd_, (off_t)async_->written_to) != 0) {   // blocks reserved past the end go back
        ::close(fd_);
        fd_ = -1;
        throw std::runtime_error("Failed to write the output file: " + path_);
    }
    if (fd_ >= 0 && ::close(fd_) != 0) { fd_ = -1; throw std::runtime_error("Failed to write the output file: " + path_); }
    fd_ = -1;
}

BpfWriter::Times BpfWriter::times() const {
    Times t;
    t.gather = gather_s_;
    t.wait = wait_s_;
    if (async_) { std::lock_guard<std::mutex> g(async_->m); t.write = async_->write_s; }
    return t;
}

// Round 5, second pass.  tools/io_bench.cpp on the GPU box: `write()` of 320 MB into the page cache of /tmp 31 ms = 10.8 GB/s from one
// thread; N threads pwrite()-ing large disjoint ranges 31 - 37 ms, 26 - 28 ms into fallocate()d blocks -- the inode lock: no scaling --;
// a shared mapping 46 - 153 ms; O_DIRECT 42 - 57 ms.  So the write() is 30 ms of the 54 ms this stage took for 1 M x 150 bp reads; the
// other 24 were the record order, the vector of Records and the gather, all on the writing thread.  Here the gather is the pool's (a
// chunk's records cut into slabs, each slab into byte-balanced parts), the write() another thread's, and the stage costs what the
// write()s cost -- which depends on WHERE the slab is when it is written (tools/r05_bpf.sh, write() seconds for 317 MB / 2 GB):
//   ring of 2 x 32 MiB 0.029 / 0.20, 4 x 8 MiB 0.028 - 0.030 / 0.18, 8 x 4 MiB 0.028 / 0.18 | 4 x 4 MiB 0.021 - 0.023 / 0.13 - 0.14,
//   8 x 2 MiB 0.021 - 0.026 / 0.13 - 0.14, 16 x 1 MiB 0.022 - 0.025: a ring of 16 MiB stays in the last-level cache of the pool's
// domain (32 MB on that host; the writing thread joins the domain) and write() copies from there.  Blocks reserved ahead (below): another
// 10 %.  The command, fresh output file, three runs each: 1 M x 150 bp 0.054 - 0.063 -> 0.032 - 0.035 s (2.4 - 2.8 -> 4.3 - 4.7 Gbases/s),
// 100 k x 10 kbp 0.253 - 0.255 -> 0.194 - 0.213 s; pool of 4 threads as good as 8.  (What round 5's first attempt -- two unpinned helper
// threads gathering 8 MiB slabs ahead of the writer, slower than the loop above -- got wrong: where its threads and its slabs were.)
// Re-opening an EXISTING output file costs another 30 ms in close(): ext4 flushes a file that was truncated and rewritten when it is
// closed (auto_da_alloc) -- the reference pays the same; the measurements above write fresh files.
void BpfWriter::append(const Chunk &c, WorkerPool &pool) {
    using clk = std::chrono::steady_clock;
    static const size_t kSlabBytes = [] {                              // (MOVI_BPF_SLAB_BYTES: the tests cut slabs smaller than a record)
        const char *e = std::getenv("MOVI_BPF_SLAB_BYTES");
        const long long v = e ? std::atoll(e) : 0;
        return v > 0 ? (size_t)v : (size_t)(4u << 20);
    }();
    if (c.n == 0) return;
    if (!async_) {
        async_ = new Async();
        Async *a = async_;
        if (const char *e = std::getenv("MOVI_BPF_SLABS")) a->n_slabs = (unsigned)std::min<long>(Async::kMaxSlabs, std::max<long>(2, std::atol(e)));
        a->reserve = std::getenv("MOVI_BPF_NO_RESERVE") == nullptr;
        a->written_to = (uint64_t)::lseek(fd_, 0, SEEK_CUR);
        a->th = std::thread([this, a, &pool] {
            pool.adopt_owner();
            std::unique_lock<std::mutex> g(a->m);
            for (;;) {
                a->cv.wait(g, [&] { return a->stop || a->full[a->write_next]; });
                if (a->abort || !a->full[a->write_next]) return;       // stop with nothing queued -- or the destructor during unwinding: whatever is queued is dropped
                const unsigned s = a->write_next;
                if (a->failed) {                                       // a write() has failed: nothing more goes into a file known to be bad
                    a->full[s] = false;
                    a->queued -= 1;
                    a->write_next = (a->write_next + 1u) % a->n_slabs;
                    a->cv.notify_all();
                    continue;
                }
                g.unlock();
                const auto t0 = clk::now();
                std::exception_ptr err;
                const uint8_t *q = a->slab[s].data();
                size_t len = a->slab[s].size();
                // blocks reserved ahead of the writes, 256 MiB at a time, the file's size left alone (what is left over goes back in
                // close()): a write() into reserved blocks skips the per-page delayed-allocation bookkeeping -- 320 MB in 26 instead of
                // 31 ms (tools/io_bench.cpp).  A file system that cannot do it is not asked again.
                if (a->reserve && a->written_to + len > a->reserved_to) {
                    const uint64_t from = std::max(a->reserved_to, a->written_to), more = std::max<uint64_t>(256u << 20, len);
                    if (::fallocate(fd_, FALLOC_FL_KEEP_SIZE, (off_t)from, (off_t)more) == 0) a->reserved_to = from + more;
                    else a->reserve = false;
                }
                a->written_to += len;
                while (len) {
                    const ssize_t w = ::write(fd_, q, len);
                    if (w < 0 && errno == EINTR) continue;
                    if (w < 0) {
                        err = std::make_exception_ptr(std::runtime_error("Failed to write the output file: " + path_));
                        a->written_to -= len;                          // (what did not reach the file)
                        break;
                    }
                    q += w;
                    len -= (size_t)w;
                }
                const double dt = std::chrono::duration<double>(clk::now() - t0).count();
                g.lock();
                a->write_s += dt;
                if (err && !a->err) a->err = err;
                if (err) a->failed = true;
                a->full[s] = false;
                a->queued -= 1;
                a->write_next = (a->write_next + 1u) % a->n_slabs;
                a->cv.notify_all();
            }
        });
    }
    Async *a = async_;
    auto rec_bytes = [&](size_t k) -> uint64_t {
        const uint32_t i = c.order[k];
        return 10u + (uint64_t)(uint16_t)(c.id_off[i + 1] - c.id_off[i]) + 2u * (c.offsets[i + 1] - c.offsets[i]);
    };
    // byte offset of every record in the chunk's image (two-pass prefix sum by the pool)
    std::vector<uint64_t> at(c.n + 1);
    const unsigned P = std::max(1u, std::min<unsigned>(pool.size() * 2u, (unsigned)std::min<size_t>(c.n, 1u << 16)));
    std::vector<uint64_t> part_sum(P + 1, 0);
    const auto tg0 = clk::now();
    pool.run(P, [&](unsigned p) {
        uint64_t sum = 0;
        for (size_t k = c.n * p / P, e = c.n * (p + 1) / P; k < e; k++) { at[k] = sum; sum += rec_bytes(k); }
        part_sum[p + 1] = sum;
    });
    for (unsigned p = 0; p < P; p++) part_sum[p + 1] += part_sum[p];
    pool.run(P, [&](unsigned p) {
        const uint64_t base = part_sum[p];
        if (base) for (size_t k = c.n * p / P, e = c.n * (p + 1) / P; k < e; k++) at[k] += base;
    });
    at[c.n] = part_sum[P];
    gather_s_ += std::chrono::duration<double>(clk::now() - tg0).count();
    size_t k0 = 0;
    while (k0 < c.n) {
        // the slab: records k0 .. k1 - 1, at least one, up to kSlabBytes
        size_t k1 = (size_t)(std::upper_bound(at.begin() + (ptrdiff_t)k0, at.end(), at[k0] + kSlabBytes) - at.begin());
        if (k1 > 0) k1 -= 1;
        if (k1 <= k0) k1 = k0 + 1;
        if (k1 > c.n) k1 = c.n;
        const uint64_t lo = at[k0], bytes = at[k1] - lo;
        const unsigned s = a->fill_next;
        {
            const auto tw0 = clk::now();
            std::unique_lock<std::mutex> g(a->m);
            a->cv.wait(g, [&] { return !a->full[s] || a->err; });
            if (a->err) { std::exception_ptr e = a->err; a->err = nullptr; std::rethrow_exception(e); }
            wait_s_ += std::chrono::duration<double>(clk::now() - tw0).count();
        }
        const auto t0 = clk::now();
        a->slab[s].resize_uninitialized((size_t)bytes);
        uint8_t *dst = a->slab[s].data();
        const unsigned Q = std::max(1u, std::min<unsigned>(pool.size() * 2u, (unsigned)std::min<size_t>(k1 - k0, 1u << 16)));
        pool.run(Q, [&](unsigned p) {
            // parts balanced by bytes: part p takes the records that START in its share of the slab
            auto cut = [&](unsigned j) -> size_t {
                if (j == 0) return k0;
                if (j >= Q) return k1;
                return (size_t)(std::lower_bound(at.begin() + (ptrdiff_t)k0, at.begin() + (ptrdiff_t)k1, lo + bytes * j / Q) - at.begin());
            };
            for (size_t k = cut(p), e = cut(p + 1); k < e; k++) {
                const uint32_t i = c.order[k];
                uint8_t *q = dst + (at[k] - lo);
                const uint16_t idl = (uint16_t)(c.id_off[i + 1] - c.id_off[i]);      // src/utils.cpp:222
                const uint64_t n = c.offsets[i + 1] - c.offsets[i];
                std::memcpy(q, &idl, 2);
                std::memcpy(q + 2, c.id_bytes + c.id_off[i], idl);
                std::memcpy(q + 2 + idl, &n, 8);                                      // output_binary :204-210
                if (n) std::memcpy(q + 10 + idl, c.pml + c.offsets[i], n * 2);
            }
        });
        gather_s_ += std::chrono::duration<double>(clk::now() - t0).count();
        {
            std::lock_guard<std::mutex> g(a->m);
            a->full[s] = true;
            a->queued += 1;
            a->fill_next = (a->fill_next + 1u) % a->n_slabs;
        }
        a->cv.notify_all();
        k0 = k1;
    }
}

// Measured on the GPU box (2 GB of PMLs for 100 k x 10 kbp reads / 317 MB for 1 M x 150 bp; processing time of the whole
// command, +-10 % run to run): records gathered in a multi-MiB buffer and written with few large write()s 0.60 / 0.11 s;
// pwritev of (head, payload) iovec pairs from 8 threads 0.55 / 0.16 s (buffered writes to ONE file serialise on the inode
// lock, and 2 M small iovecs cost more than copying them); one write() per 20 KB payload 0.77 / 0.10 s (system calls are
// expensive there); T threads copying into a MAP_SHARED mapping of the grown file 1.40 / 0.22 s (page faults on the file
// mapping are slow there).  Round 5 (tools/r05_cli.sh): the slabs gathered by two helper threads a few slabs AHEAD of the one
// thread that write()s them in order -- the gather beside the write instead of before it -- 0.067 - 0.075 s against 0.053 -
// 0.063 s for this loop: the write() into the page cache is the whole cost (317 MB in ~55 ms = 5.7 GB/s, one thread, under the
// inode lock) and the helpers only competed with it; dropped.  Kept: one thread, an 8 MiB buffer, only payloads of 1 MiB and
// more written straight from the result array.
void BpfWriter::append(const std::vector<Record> &records) {
    drain();                                                           // (slabs of append(Chunk) still on their way come first)
    if (buf_.empty()) buf_.resize(8u << 20);
    size_t fill = 0;
    auto write_all = [&](const void *p, size_t len) {
        const uint8_t *q = static_cast<const uint8_t *>(p);
        while (len) {
            const ssize_t w = ::write(fd_, q, len);
            if (w < 0 && errno == EINTR) continue;                       // a signal, not an error: the ofstream this replaced retried too
            if (w < 0) throw std::runtime_error("Failed to write the output file: " + path_);
            q += w;
            len -= (size_t)w;
        }
    };
    auto flush = [&] { if (fill) { write_all(buf_.data(), fill); fill = 0; } };
    auto put = [&](const void *p, size_t len) {
        if (len > buf_.size() - fill) flush();
        std::memcpy(buf_.data() + fill, p, len);
        fill += len;
    };
    for (const Record &r : records) {
        const uint16_t idl = static_cast<uint16_t>(r.id.length());      // src/utils.cpp:222
        put(&idl, 2);
        put(r.id.data(), idl);
        put(&r.n, 8);                                                    // output_binary :204-210
        const size_t bytes = r.n * 2;
        if (bytes >= (1u << 20)) { flush(); write_all(r.pml, bytes); }
        else if (bytes) put(r.pml, bytes);
    }
    flush();
    if (async_) async_->written_to = (uint64_t)::lseek(fd_, 0, SEEK_CUR);   // (nothing is queued: drain() above)
}

void append_stdout_pmls(std::string &txt, std::string_view id, const uint16_t *pml, uint64_t n) {
    // add_ml appends " " + reversed digits per value and the whole string is reversed once at the
    // end: the net effect is the values in read order, each followed by one space.
    txt.push_back('>');
    txt += id;
    txt.push_back('\n');
    char buf[8];
    for (uint64_t i = n; i-- > 0;) {
        unsigned v = pml[i];
        int k = 0;
        do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) txt.push_back(buf[--k]);
        txt.push_back(' ');
    }
    txt.push_back('\n');
}

void write_stdout_pmls(std::ostream &out, std::string_view id, const uint16_t *pml, uint64_t n) {
    std::string line;
    line.reserve(id.size() + n * 3 + 3);
    append_stdout_pmls(line, id, pml, n);
    out << line;
}

void write_count_line(std::ostream &out, std::string_view id, uint64_t query_length, uint64_t matched, uint64_t count) {
    out << id << "\t" << matched << "/" << query_length << "\t" << count << "\n";
}

void append_count_line(std::string &txt, std::string_view id, uint64_t query_length, uint64_t matched, uint64_t count) {
    char num[24];
    auto put = [&](uint64_t v) {
        const auto r = std::to_chars(num, num + sizeof(num), v);
        txt.append(num, (size_t)(r.ptr - num));
    };
    txt.append(id.data(), id.size());
    txt.push_back('\t');
    put(matched);
    txt.push_back('/');
    put(query_length);
    txt.push_back('\t');
    put(count);
    txt.push_back('\n');
}

size_t Classifier::load_null_db(const std::string &index_dir, const std::string &query_type, bool verbose) {
    const std::string name = index_dir + "/movi." + query_type + ".nulldb";      // emperical_null_database.cpp:107
    std::ifstream in(name, std::ios::in | std::ios::binary);
    if (!in.good()) throw std::runtime_error("Failed to open the null database: " + name);
    uint64_t num_values = 0, percentile_value = 0;
    double mean_null_stat = 0;
    in.read(reinterpret_cast<char *>(&num_values), 8);
    in.read(reinterpret_cast<char *>(&mean_null_stat), 8);
    in.read(reinterpret_cast<char *>(&percentile_value), 8);
    if (!in.good()) throw std::runtime_error("Truncated null database: " + name);
    const size_t min_matching_length = 3;                             // MIN_MATCHING_LENGTH
    max_value_thr = static_cast<uint16_t>(std::max<size_t>(percentile_value, min_matching_length) + 1);
    if (verbose)
        std::cerr << "Null database statistics: mean_null_stat: " << mean_null_stat
                  << " percentile_value: " << percentile_value << " num_values: " << num_values << "\n";
    return max_value_thr;
}

void Classifier::write_report_header(std::ostream &out) const {      // src/classifier.cpp:51-60
    out.precision(4);
    out << std::setw(30) << std::left << "read id:"
        << std::setw(15) << std::left << "status:"
        << std::setw(19) << std::left << "avg max-value (thr="
        << std::setw(2) << std::left << max_value_thr
        << std::setw(5) << std::left << "):"
        << std::setw(12) << std::left << "above thr:"
        << std::setw(12) << std::left << "below thr:" << std::endl;
}

bool Classifier::classify(std::string_view read_name, const uint16_t *pml, uint64_t n, size_t bin_width,
                          std::ostream *out) const {                  // src/classifier.cpp:99-143
    size_t sum_max_bin_values = 0, bins = 0;
    size_t bins_above = 0, bins_below = 0;
    size_t start_pos = 0, end_pos = 0;
    while (start_pos < n) {
        end_pos = (start_pos + bin_width < n) ? start_pos + bin_width : n;
        if (n - end_pos < bin_width) end_pos = n;                     // avoids small regions at the end of read
        const uint16_t max_val = *std::max_element(pml + start_pos, pml + end_pos);
        if (max_val >= max_value_thr) bins_above++;
        else bins_below++;
        sum_max_bin_values += max_val;
        bins++;
        start_pos += (end_pos - start_pos);
    }
    const bool read_found = (bins_above / (bins_above + bins_below + 0.0) > 0.50);
    if (out) {
        out->precision(3);
        *out << std::setw(30) << std::left << read_name
             << std::setw(15) << std::left << (read_found ? "FOUND" : "NOT_PRESENT")
             << std::setw(26) << std::left << (sum_max_bin_values + 0.0) / bins
             << std::setw(12) << std::left << bins_above
             << std::setw(12) << std::left << bins_below << "\n";
    }
    return read_found;
}

static int view_bpf_stream(const Options &o, std::ostream &out) {    // src/movi.cpp:402-503, record by record
    std::ifstream f(o.bpf_file, std::ios::in | std::ios::binary);
    if (!f.good()) throw std::runtime_error("Failed to open the MLS file: " + o.bpf_file);
    uint8_t entry_size = 32;
    if (!o.no_header) {
        uint8_t h[12];
        f.read(reinterpret_cast<char *>(h), 12);
        uint32_t magic;
        std::memcpy(&magic, h, 4);
        if (!f.good() || magic != kBpfMagic) throw std::runtime_error("Invalid BPF header.");
        if (h[4] != 1) throw std::runtime_error("Invalid BPF version.");
        entry_size = h[7];
    } else if (o.small_bpf) {
        entry_size = 16;
    } else if (o.large_bpf) {
        entry_size = 64;
    }
    if (entry_size != 16 && entry_size != 32 && entry_size != 64) throw std::runtime_error("Invalid BPF entry size.");
    std::vector<uint8_t> buf;
    std::string line;
    while (true) {
        uint16_t st_length = 0;
        f.read(reinterpret_cast<char *>(&st_length), 2);
        if (f.eof()) break;
        std::string read_name(st_length, '\0');
        f.read(&read_name[0], st_length);
        read_name.erase(std::find(read_name.begin(), read_name.end(), '\0'), read_name.end());
        out << ">" << read_name << "\n";
        uint64_t n = 0;
        f.read(reinterpret_cast<char *>(&n), 8);
        const size_t w = entry_size / 8;
        buf.resize(n * w);
        f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)(n * w));
        line.clear();
        for (uint64_t i = n; i-- > 0;) {
            uint64_t v = 0;
            std::memcpy(&v, buf.data() + i * w, w);
            line += std::to_string(v);
            line.push_back(' ');
        }
        out << line << "\n";
    }
    return 0;
}


// `movi view`: the same text, produced from a memory-mapped file -- records are indexed sequentially (three small reads
// each), then formatted by worker threads in batches of ~64 MB of values and written in file order.  Any irregularity
// (not a regular file, a truncated record) falls back to the record-by-record reader above, which behaves like the
// reference on such input.  2 GB of PMLs: 7.5 s -> ~1 s on the 256-core host.
int view_bpf(const Options &o, std::ostream &out) {
    const int fd = ::open(o.bpf_file.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Failed to open the MLS file: " + o.bpf_file);
    struct stat sb;
    void *m = MAP_FAILED;
    size_t bytes = 0;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
        bytes = (size_t)sb.st_size;
        m = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    }
    ::close(fd);
    if (m == MAP_FAILED) return view_bpf_stream(o, out);
    struct Unmap { void *p; size_t n; ~Unmap() { ::munmap(p, n); } } unmap{m, bytes};
    const uint8_t *p = static_cast<const uint8_t *>(m);
    size_t pos = 0;
    uint8_t entry_size = 32;
    if (!o.no_header) {
        if (bytes < 12) return view_bpf_stream(o, out);
        uint32_t magic;
        std::memcpy(&magic, p, 4);
        if (magic != kBpfMagic) throw std::runtime_error("Invalid BPF header.");
        if (p[4] != 1) throw std::runtime_error("Invalid BPF version.");
        entry_size = p[7];
        pos = 12;
    } else if (o.small_bpf) {
        entry_size = 16;
    } else if (o.large_bpf) {
        entry_size = 64;
    }
    if (entry_size != 16 && entry_size != 32 && entry_size != 64) throw std::runtime_error("Invalid BPF entry size.");
    const size_t w = entry_size / 8;
    struct Rec { size_t name, name_len, vals; uint64_t n; };
    std::vector<Rec> recs;
    while (pos < bytes) {
        if (bytes - pos < 2) return view_bpf_stream(o, out);
        uint16_t idl;
        std::memcpy(&idl, p + pos, 2);
        if (bytes - pos < 2u + idl + 8u) return view_bpf_stream(o, out);
        uint64_t n;
        std::memcpy(&n, p + pos + 2 + idl, 8);
        const size_t vals = pos + 2 + idl + 8;
        if (n > (bytes - vals) / w) return view_bpf_stream(o, out);
        recs.push_back(Rec{pos + 2, idl, vals, n});
        pos = vals + n * w;
    }
    auto format = [&](size_t a, size_t b, std::string &txt) {
        size_t need = 0;
        for (size_t i = a; i < b; i++) need += recs[i].name_len + 3 + recs[i].n * (w == 2 ? 4 : 8);
        txt.clear();
        txt.reserve(need);
        char digits[24];
        for (size_t i = a; i < b; i++) {
            const Rec &r = recs[i];
            txt.push_back('>');
            const char *name = reinterpret_cast<const char *>(p + r.name);
            const void *nul = std::memchr(name, 0, r.name_len);            // the reference erases from the first NUL on
            txt.append(name, nul ? (size_t)(static_cast<const char *>(nul) - name) : r.name_len);
            txt.push_back('\n');
            for (uint64_t k = r.n; k-- > 0;) {                               // stored last base first: printed in read order
                uint64_t v = 0;
                std::memcpy(&v, p + r.vals + k * w, w);
                int d = 0;
                do { digits[d++] = (char)('0' + v % 10); v /= 10; } while (v);
                while (d) txt.push_back(digits[--d]);
                txt.push_back(' ');
            }
            txt.push_back('\n');
        }
    };
    const unsigned T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const uint64_t batch_vals = 32u << 20;                                 // values per batch, all threads together
    std::vector<std::string> txt(T);
    size_t i = 0;
    while (i < recs.size()) {
        size_t j = i;
        uint64_t vals = 0;
        while (j < recs.size() && (vals < batch_vals || j == i)) vals += recs[j++].n;
        if (T == 1 || vals < (1u << 20)) {
            format(i, j, txt[0]);
            out.write(txt[0].data(), (std::streamsize)txt[0].size());
        } else {
            std::vector<size_t> cut(T + 1, j);                              // ranges balanced by values
            cut[0] = i;
            uint64_t acc = 0;
            unsigned t = 1;
            for (size_t k = i; k < j && t < T; k++) {
                acc += recs[k].n;
                while (t < T && acc >= vals * t / T) cut[t++] = k + 1;
            }
            std::vector<std::thread> th;
            for (unsigned u = 0; u < T; u++) th.emplace_back([&, u] { format(cut[u], cut[u + 1], txt[u]); });
            for (auto &x : th) x.join();
            for (unsigned u = 0; u < T; u++) out.write(txt[u].data(), (std::streamsize)txt[u].size());
        }
        i = j;
    }
    return 0;
}

}  // namespace movi_host
