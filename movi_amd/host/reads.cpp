#include "reads.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <pthread.h>
#include <sched.h>
#include <chrono>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <thread>

namespace movi_host {

// The hardware threads that share a last-level cache with the calling thread, from sysfs ("0-7,128-135"); empty if unknown.
std::vector<int> llc_siblings() {
    std::vector<int> cpus;
    const int cpu = sched_getcpu();
    if (cpu < 0) return cpus;
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE *f = fopen(path, "r");
    if (!f) return cpus;
    char buf[4096];
    if (fgets(buf, sizeof(buf), f)) {
        for (char *p = buf; *p;) {
            char *e;
            long a = strtol(p, &e, 10), b = a;
            if (e == p) break;
            if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back((int)c);
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',' ) break;
        }
    }
    fclose(f);
    return cpus;
}

// The pool's threads and the thread that owns it are kept on ONE last-level-cache domain.  The parser's sequential phase
// writes the line and record tables that the workers read in the parallel phases and rewrites them for the next chunk:
// with the workers spread over the sockets of a 2 x 64-core host every one of those rewrites first had to pull its cache
// line back from another chiplet or socket, and the sequential phase ran 2 x slower with workers than without
// (profiles/r03_cli_path.txt).  MOVI_NO_AFFINITY=1 leaves the threads where the scheduler puts them.
WorkerPool::WorkerPool(unsigned threads) {
    std::vector<int> cpus;
    if (threads > 1 && !getenv("MOVI_NO_AFFINITY")) cpus = llc_siblings();
    // only the CPUs this thread may run on anyway (a cpuset may grant fewer than the cache domain lists); the owner's own
    // mask is put back when the pool goes (an embedder's main thread -- `movi plan`, tools/parse_bench -- must not stay
    // confined to one cache domain, nor hand that mask down to every thread it starts later)
    have_owner_mask_ = pthread_getaffinity_np(pthread_self(), sizeof(owner_mask_), &owner_mask_) == 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    size_t usable = 0;
    for (int c : cpus)
        if (!have_owner_mask_ || CPU_ISSET(c, &owner_mask_)) { CPU_SET(c, &set); usable++; }
    const bool pin = usable >= 2;
    if (pin) { pool_set_ = set; pinned_ = true; }
    if (pin) {
        owner_ = pthread_self();
        pinned_owner_ = pthread_setaffinity_np(owner_, sizeof(set), &set) == 0 && have_owner_mask_;
        if (threads > usable) threads = (unsigned)usable;
    }
    for (unsigned t = 1; t < threads; t++) {
        th_.emplace_back([this] { loop(); });
        if (pin) (void)pthread_setaffinity_np(th_.back().native_handle(), sizeof(set), &set);
    }
}

WorkerPool::~WorkerPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
    // (only from the owner itself: in `movi query` the owner is the parser thread, which is gone by the time the reader is
    // destroyed -- its handle must not be touched)
    if (pinned_owner_ && pthread_equal(pthread_self(), owner_)) (void)pthread_setaffinity_np(owner_, sizeof(owner_mask_), &owner_mask_);
}

void WorkerPool::adopt_owner() {
    if (pinned_) (void)pthread_setaffinity_np(pthread_self(), sizeof(pool_set_), &pool_set_);
}

void WorkerPool::loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> g(m_);
    for (;;) {
        cv_.wait(g, [&] { return stop_ || (gen_ != seen && next_ < parts_); });
        if (stop_) return;
        seen = gen_;
        while (next_ < parts_) {
            const unsigned k = next_++;
            g.unlock();
            (*fn_)(k);
            g.lock();
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
}

void WorkerPool::run(unsigned parts, const std::function<void(unsigned)> &fn) {
    if (parts == 0) return;
    if (parts == 1 || th_.empty()) { for (unsigned k = 0; k < parts; k++) fn(k); return; }
    std::unique_lock<std::mutex> g(m_);
    fn_ = &fn; parts_ = parts; next_ = 0; pending_ = parts; gen_++;
    cv_.notify_all();
    while (next_ < parts_) {                                           // the caller takes parts too
        const unsigned k = next_++;
        g.unlock();
        fn(k);
        g.lock();
        --pending_;
    }
    done_cv_.wait(g, [&] { return pending_ == 0; });
    fn_ = nullptr;
}

// Newlines of [p, e): offsets relative to `base` appended to v, the byte behind each (0 at `fin`) to f.  32 bytes per
// compare where AVX2 is there (FASTA lines are short: one memchr call per 10-byte header line cost more than its scan).
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static const char *scan_newlines_avx2(const char *p, const char *e, const char *base, const char *fin,
                                                                       std::vector<size_t> &v, std::vector<uint8_t> &f) {
    const __m256i nl = _mm256_set1_epi8('\n');
    while (p + 32 <= e) {
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p)), nl));
        while (m) {
            const char *q = p + __builtin_ctz(m);
            m &= m - 1;
            v.push_back((size_t)(q - base));
            f.push_back(q + 1 < fin ? (uint8_t)q[1] : (uint8_t)0);
        }
        p += 32;
    }
    return p;
}
#endif
static void scan_newlines(const char *p, const char *e, const char *base, const char *fin, std::vector<size_t> &v, std::vector<uint8_t> &f) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) p = scan_newlines_avx2(p, e, base, fin, v, f);
#endif
    while (p < e) {
        const void *q = std::memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        p = static_cast<const char *>(q) + 1;
        v.push_back((size_t)(p - 1 - base));
        f.push_back(p < fin ? (uint8_t)*p : (uint8_t)0);
    }
}

LineSource::~LineSource() {
    for (auto &p : ahead_)
        if (p->th.joinable()) p->th.join();
}

// Keep scan_depth_ windows in flight behind cur_, one helper thread each (a window is ~40 MB).  One scanner takes 5 - 6.5 ms per
// window of a mapping it is the first to touch; a warm parser consumes a window in ~3 ms, so with ONE window in flight the cut of
// the third and fourth chunk of a 1 M x 150 bp run waited 2 - 3 and 1 - 2 ms for it (tools/r05_scan.sh: the wait and the helper's own
// time are in --verbose's per-chunk line).  More scanners on the same window did not help -- four take 3.7 - 4.5 ms, not 1.4:
// their page faults queue up behind the GPU-call thread's page pinning on the address-space lock -- but two windows scanned side
// by side do: each helper still needs its 5 - 6 ms, started one window earlier.
void LineSource::scan_ahead() {
    if (window_bytes_ == 0) return;
    if (ahead_.empty()) ahead_to_ = cur_.to;
    while (ahead_.size() < scan_depth_ && ahead_to_ < end_) {
        std::unique_ptr<Pending> p(new Pending());
        p->w.from = ahead_to_;
        p->w.to = std::min(end_, ahead_to_ + window_bytes_);
        ahead_to_ = p->w.to;
        // long lines (the window in use has fewer than one per KiB: long reads): four scanners -- there the scan of a GB-sized
        // chunk by one thread was what the parser waited for (0.09 of 0.13 s on 100 k x 10 kbp) and the few line ends are joined
        // in no time; short lines: one scanner writes the list in place
        unsigned parts = (cur_.to > cur_.from && cur_.nl.size() * 1024 < cur_.to - cur_.from) ? 4u : 1u;
        static const unsigned forced = [] { const char *e = std::getenv("MOVI_SCAN_PARTS"); return e ? (unsigned)std::max(1, std::atoi(e)) : 0u; }();
        if (forced) parts = forced;
        Pending *q = p.get();
        q->th = std::thread([this, q, parts] {
            const auto t0 = std::chrono::steady_clock::now();
            Window &w = q->w;
            const size_t from = w.from, to = w.to;
            if (parts == 1) {
                w.nl.reserve((to - from) / 64 + 16);
                w.first.reserve((to - from) / 64 + 16);
                scan_newlines(mem_ + from, mem_ + to, mem_, mem_ + end_, w.nl, w.first);
            } else {
                std::vector<std::vector<size_t>> nl(parts);
                std::vector<std::vector<uint8_t>> first(parts);
                std::vector<std::thread> th;
                for (unsigned t = 0; t < parts; t++)
                    th.emplace_back([&, t] {
                        const size_t a = from + (to - from) * t / parts, b = from + (to - from) * (t + 1) / parts;
                        scan_newlines(mem_ + a, mem_ + b, mem_, mem_ + end_, nl[t], first[t]);
                    });
                for (auto &x : th) x.join();
                for (unsigned t = 0; t < parts; t++) {
                    w.nl.insert(w.nl.end(), nl[t].begin(), nl[t].end());
                    w.first.insert(w.first.end(), first[t].begin(), first[t].end());
                }
            }
            q->busy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
        ahead_.push_back(std::move(p));
    }
}

bool LineSource::next_window() {
    if (ahead_.empty()) return false;
    {
        const auto tw = std::chrono::steady_clock::now();
        ahead_.front()->th.join();
        scan_wait_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
    }
    scan_busy_s_ += ahead_.front()->busy_s;
    cur_ = std::move(ahead_.front()->w);
    ahead_.pop_front();
    nl_i_ = 0;
    while (nl_i_ < cur_.nl.size() && cur_.nl[nl_i_] < pos_) nl_i_++;   // (a line longer than a window was finished the slow way)
    scan_ahead();
    return nl_i_ < cur_.nl.size() || next_window();
}

void LineSource::prescan(size_t bytes, WorkerPool &pool) {
    if (!mem_) return;
    window_bytes_ = bytes;
    if (cur_.to == 0 && cur_.nl.empty() && ahead_.empty()) {          // the first window: all workers, now
        const size_t from = pos_, to = std::min(end_, pos_ + bytes);
        if (to > from) {
            const unsigned T = (to - from) >= (1u << 22) ? pool.size() : 1;
            std::vector<std::vector<size_t>> part(T);
            std::vector<std::vector<uint8_t>> part_first(T);
            pool.run(T, [&](unsigned t) {
                const size_t a = from + (to - from) * t / T, b = from + (to - from) * (t + 1) / T;
                part[t].reserve((b - a) / 64 + 16);
                part_first[t].reserve((b - a) / 64 + 16);
                scan_newlines(mem_ + a, mem_ + b, mem_, mem_ + end_, part[t], part_first[t]);
            });
            for (unsigned t = 0; t < T; t++) {
                cur_.nl.insert(cur_.nl.end(), part[t].begin(), part[t].end());
                cur_.first.insert(cur_.first.end(), part_first[t].begin(), part_first[t].end());
            }
            cur_.from = from;
            cur_.to = to;
            nl_i_ = 0;
        }
    }
    else {
        // (every later call: the parser is running -- from now on two windows in flight; the first call, which may be the warm-up's
        // while the index still loads, starts one: nothing more of the input is read ahead of the clock than before)
        static const unsigned depth = [] { const char *e = std::getenv("MOVI_SCAN_DEPTH"); return e ? (unsigned)std::min(8, std::max(1, std::atoi(e))) : 2u; }();
        scan_depth_ = depth;
    }
    scan_ahead();                                                      // keep the windows behind the current one in flight
}

bool LineSource::fill() {
    if (drained_) return false;
    if (pos_ > 0) {
        std::memmove(buf_.data(), buf_.data() + pos_, end_ - pos_);
        end_ -= pos_;
        pos_ = 0;
    }
    if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);             // a single line longer than the buffer
    in_->read(buf_.data() + end_, (std::streamsize)(buf_.size() - end_));
    const size_t got = (size_t)in_->gcount();
    end_ += got;
    if (got == 0) drained_ = true;
    return got > 0;
}

int LineSource::peek_slow() {
    if (pos_ == end_ && !fill()) { eof_ = true; return std::char_traits<char>::eof(); }
    return (unsigned char)data()[pos_];
}

bool LineSource::getline_slow(const char *&p, size_t &n) {
    for (;;) {
        const char *base = data() + pos_;
        const void *nl = std::memchr(base, '\n', end_ - pos_);
        if (nl) {
            n = (size_t)(static_cast<const char *>(nl) - base);
            p = base;
            pos_ += n + 1;
            return true;
        }
        if (!fill()) break;
    }
    if (pos_ == end_) { eof_ = true; return false; }                   // nothing left: getline fails
    p = data() + pos_;                                                 // last line without a newline:
    n = end_ - pos_;                                                   // returned, and eof is set
    pos_ = end_;
    eof_ = true;
    return true;
}

// One reference batch = the lines loadBatch would put into its stringstream
// (src/batch_loader.cpp:50-87): lines are read until BOTH >= 1000 "bases" and >= min_reads
// reads are covered.  FASTQ: a read is counted every 4 lines with (record bytes)/2 bases;
// FASTA: a read is counted when the NEXT line starts with '>' with (record bytes) bases.
// The batch's lines are appended to lines_ from index first_line on.
bool BatchReader::detect_format() {
    if (format_ >= 0) return true;
    if (!src_.good()) return false;
    int c = src_.peek();
    if (c == '>') format_ = 0;
    else if (c == '@') format_ = 1;
    else if (c == std::char_traits<char>::eof()) return false;
    else throw std::runtime_error("unrecognized input query file type - expects FASTA or FASTQ.");
    return true;
}

bool BatchReader::load_batch(size_t &first_line) {
    first_line = lines_.size();
    if (!detect_format()) return false;
    size_t bases = 0, reads = 0, nlines = 0, record = 0;
    const size_t num_bases = 1000;
    bool valid = false;
    while (src_.good() && (bases < num_bases || reads < min_reads_)) {
        const char *p;
        size_t n;
        const int first = src_.peek_first();               // this line's first character (EOF state untouched by the fast path)
        if (!src_.getline(p, n)) {
            if (format_ == 1 && nlines % 4 == 0) return valid || lines_.size() > first_line;
            if (format_ == 0 && nlines % 2 == 0) return valid || lines_.size() > first_line;
            // the reference returns false here and drops the partial batch (:57-63)
            lines_.resize_uninitialized(first_line);
            return false;
        }
        nlines++;
        record += n;
        valid = true;
        const uint8_t fc = n ? (uint8_t)(first >= 0 ? first : p[0]) : (uint8_t)0;
        if (n > 0xFFFFFFFFull) throw std::runtime_error("a line of the query file is longer than 4 GiB");
        if (lines_.size() >= 0xFFFFFFF0ull) throw std::runtime_error("more than 2^32 lines in one chunk of the query file");
        if (mem_) {
            lines_.push_back(Span{(uint64_t)(p - mem_), (uint32_t)n, fc});
        } else {
            lines_.push_back(Span{(uint64_t)arena_.size(), (uint32_t)n, fc});
            arena_.append(p, n);
        }
        if (format_ == 1) {
            if (nlines % 4 == 0) { bases += record / 2; record = 0; reads++; }
        } else if (src_.peek_first() == '>') {
            bases += record; record = 0; reads++;
        }
    }
    return valid;
}

// The batch cut over the lines the scan-ahead has already found, without walking them one by one (memory-mapped input).
// load_batch + the grabNextRead loop of next_chunk spend ~10 ns per line on a well-formed file deciding what a handful of
// array operations decide for a whole chunk: where the records start (FASTA: the lines that begin with '>'; FASTQ: every
// fourth line), how many bytes each has, and -- one running sum over the RECORDS -- where loadBatch's "1000 bases and
// min_reads reads" rule ends each batch and the chunk's budget ends the chunk.  Workers find the headers and fill lines_ /
// recs_; only the running sum is sequential (~2 ns per read).
// Taken only where it provably does what the line-by-line cut does: every line in reach non-empty (an empty header ends a
// batch early, src/batch_loader.cpp:99) and shorter than 4 GiB, every header longer than 2 characters and -- FASTQ --
// starting with '@'; whole batches only; the last record in reach (whose end the scan cannot vouch for) and anything
// irregular are left to the line-by-line cut, which then starts exactly where a batch would start anyway.
// An irregular line does not send everything in reach to the line-by-line cut (which takes ~7 reads per call: the check
// would run over the same million lines again for every batch, 0.011 -> 0.39 s per chunk with ONE trailing blank line): the
// records before it are cut in bulk, the cut goes line by line across it -- no pass is made while the input position is
// within 4096 lines of it (skip_until_) -- and in bulk again behind it.
bool BatchReader::cut_ahead(uint64_t max_bases, uint64_t min_reads, uint64_t hard_max_bases, uint64_t &approx_bases) {
    if (!mem_ || format_ < 0 || no_fast_cut_) return false;
    const LineSource::Ahead A = src_.ahead();
    if (A.pos < skip_until_) return false;
    // no further than the chunk can possibly reach (the passes below run over every line considered)
    const uint64_t budget = approx_bases < max_bases ? max_bases - approx_bases : (approx_bases < hard_max_bases ? hard_max_bases - approx_bases : 0);
    const size_t reach = A.pos + (size_t)std::min<uint64_t>(budget + (budget >> 2) + (1u << 20), 1ull << 40);
    size_t M = (size_t)(std::upper_bound(A.nl, A.nl + A.count, reach) - A.nl);
    // an irregular line a pass has already found (bad_at_ = the byte behind it) bounds this one too: the passes below run over
    // every line considered, and a batch that ended well short of that line would otherwise send them over the same lines up to it
    // -- and beyond -- again; after an irregular line the reach starts small and doubles with every clean pass (lines_cap_), so the
    // passes over an input with an irregular line every few thousand lines cost what lies between those lines, not the chunk
    if (bad_at_ && A.pos >= bad_at_) bad_at_ = 0;
    if (bad_at_) M = std::min(M, (size_t)(std::lower_bound(A.nl, A.nl + M, bad_at_ - 1) - A.nl));
    if (lines_cap_) M = std::min(M, lines_cap_);
    const size_t L0 = lines_.size(), R0 = recs_.size();
    if (M < 4096 || L0 + M > 0xFFFFFFF0ull) {
        if (bad_at_ && M < 4096) skip_until_ = bad_at_;                // the known line is close by: line by line across it
        return false;
    }
    auto beg = [&](size_t t) -> size_t { return t ? A.nl[t - 1] + 1 : A.pos; };
    auto fc = [&](size_t t) -> int { return t ? (int)A.first[t - 1] : A.first0; };
    const char head = format_ == 1 ? '@' : '>';
    if (fc(0) != head) return false;
    const unsigned T = pool_->size();
    // ---- workers: the records' header lines, and is everything regular?
    std::vector<size_t> n_hdr(T + 1, 0);
    std::vector<size_t> stop(T, 0);                                    // first irregular line of each worker's share (its end: none)
    size_t R = 0;                                                      // complete records in reach (the last one is not)
    // the regular prefix ends at line `first_bad`: no pass again until the line-by-line cut has crossed it
    bool clean = true;
    auto cut_short = [&](size_t first_bad) {
        skip_until_ = first_bad >= 4096 ? (size_t)0 : A.nl[first_bad] + 1;    // (a long prefix is cut now; the next call sees the line close by)
        bad_at_ = A.nl[first_bad] + 1;
        lines_cap_ = 16384;
        clean = false;
        M = first_bad;
        return M >= 4096;
    };
    if (format_ == 0) {
        const size_t M0 = M;
        pool_->run(T, [&](unsigned w) {
            const size_t a = M0 * w / T, b = M0 * (w + 1) / T;
            size_t c = 0, t = a;
            for (; t < b; t++) {
                const size_t len = A.nl[t] - beg(t);
                if (len == 0 || len > 0xFFFFFFFFull) break;
                if (fc(t) == '>') { if (len <= 2) break; c++; }
            }
            n_hdr[w + 1] = c;
            stop[w] = t;
        });
        for (unsigned w = 0; w < T; w++) {
            n_hdr[w + 1] += n_hdr[w];
            if (stop[w] < M0 * (w + 1) / T) {                          // the first irregular line in reach
                for (unsigned v = w + 1; v < T; v++) n_hdr[v + 1] = n_hdr[w + 1];
                if (!cut_short(stop[w])) return false;
                break;
            }
        }
        const size_t H = n_hdr[T];
        if (H < 2) return false;
        hdr_line_.resize(H);
        pool_->run(T, [&](unsigned w) {
            const size_t a = M0 * w / T, b = std::min(M, M0 * (w + 1) / T);
            size_t c = n_hdr[w];
            for (size_t t = a; t < b; t++)
                if (fc(t) == '>') hdr_line_[c++] = (uint32_t)t;
        });
        R = H - 1;
    } else {
        R = M / 4;
        if (R < 2) return false;
        const size_t Rr = R;
        pool_->run(T, [&](unsigned w) {
            const size_t a = Rr * w / T, b = Rr * (w + 1) / T;
            size_t r = a;
            for (; r < b; r++) {
                const size_t t = 4 * r, len = A.nl[t] - beg(t);
                bool irregular = fc(t) != '@' || len <= 2;
                for (size_t u = t; u < t + 4; u++)
                    if (A.nl[u] - beg(u) > 0xFFFFFFFFull) irregular = true;
                if (irregular) break;
            }
            stop[w] = r;
        });
        for (unsigned w = 0; w < T; w++)
            if (stop[w] < Rr * (w + 1) / T) {                          // the first irregular record in reach
                if (!cut_short(4 * stop[w])) return false;
                R = M / 4;
                break;
            }
        R -= 1;
    }
    if (clean && lines_cap_) lines_cap_ = lines_cap_ >= (1u << 26) ? 0 : lines_cap_ * 2;   // a clean pass: the next one may look further
    auto hdr_of = [&](size_t r) -> size_t { return format_ == 0 ? (size_t)hdr_line_[r] : 4 * r; };
    // ---- the running sum: loadBatch's rule over whole records, the chunk's budget over whole batches
    uint64_t bases = 0, pending = 0;
    size_t reads = 0, accepted = 0;
    uint32_t b = batch_counter_;
    const uint32_t b_first = b;
    // Closed form where it provably holds (round 5): the rule ends a batch behind the first record at which BOTH 1000 "bases" and
    // min_reads reads are reached (src/batch_loader.cpp:50-87).  If every record in reach counts for at least ceil(1000 / min_reads)
    // bases, min_reads records always reach 1000 and no fewer can end a batch: every batch is EXACTLY min_reads records.  The workers
    // check that and sum each batch's sequence bytes; what stays sequential is the chunk's budget over the batches (~3.5 k of them
    // per chunk of 150 bp reads instead of 224 k records: 0.45 -> 0.02 ms).  Anything else (min_reads 1 with --no-prefetch, a
    // record too short) takes the loop below.
    const size_t m = min_reads_;
    bool closed = false;
    if (m >= 2 && R >= m && !std::getenv("MOVI_NO_CLOSED_CUT")) {
        const uint64_t need = (1000 + (uint64_t)m - 1) / (uint64_t)m;
        const size_t nb = R / m;                                       // whole batches in reach
        batch_pending_.resize(nb);
        std::vector<uint8_t> short_rec(T, 0);
        pool_->run(T, [&](unsigned w) {
            for (size_t j = nb * w / T, je = nb * (w + 1) / T; j < je; j++) {
                uint64_t pend = 0;
                for (size_t r = j * m; r < (j + 1) * m; r++) {
                    const size_t h = hdr_of(r), e = hdr_of(r + 1);
                    const uint64_t bytes = (uint64_t)(A.nl[e - 1] - beg(h)) - (uint64_t)(e - h - 1);
                    if (format_ == 0) { if (bytes < need) short_rec[w] = 1; pend += bytes - (uint64_t)(A.nl[h] - beg(h)); }
                    else { if (bytes / 2 < need) short_rec[w] = 1; pend += (uint64_t)(A.nl[h + 1] - beg(h + 1)); }
                }
                batch_pending_[j] = pend;
            }
        });
        closed = true;
        for (unsigned w = 0; w < T; w++) closed = closed && !short_rec[w];
        if (closed)
            for (size_t j = 0; j < nb; j++) {
                approx_bases += batch_pending_[j];
                accepted = (j + 1) * m;
                b++;
                if (!(approx_bases < max_bases || (R0 + accepted < min_reads && approx_bases < hard_max_bases))) break;
            }
    }
    if (!closed) rec_batch_.resize(R);
    for (size_t r = 0; r < R && !closed; r++) {
        const size_t h = hdr_of(r), e = hdr_of(r + 1);                 // lines [h, e)
        const uint64_t bytes = (uint64_t)(A.nl[e - 1] - beg(h)) - (uint64_t)(e - h - 1);   // the lines' lengths, newlines excluded
        rec_batch_[r] = b;
        reads++;
        if (format_ == 0) { bases += bytes; pending += bytes - (uint64_t)(A.nl[h] - beg(h)); }
        else { bases += bytes / 2; pending += (uint64_t)(A.nl[h + 1] - beg(h + 1)); }
        if (bases >= 1000 && reads >= min_reads_) {                    // the batch ends behind this record
            approx_bases += pending;
            pending = 0; bases = 0; reads = 0;
            accepted = r + 1;
            b++;
            if (!(approx_bases < max_bases || (R0 + accepted < min_reads && approx_bases < hard_max_bases))) break;
        }
    }
    if (accepted == 0) return false;
    batch_counter_ = b;
    // ---- workers: lines_ and recs_ of the accepted batches
    const size_t C = hdr_of(accepted);                                 // lines consumed
    lines_.resize_uninitialized(L0 + C);
    recs_.resize_uninitialized(R0 + accepted);
    pool_->run(T, [&](unsigned w) {
        for (size_t t = C * w / T, te = C * (w + 1) / T; t < te; t++) {
            const size_t at = beg(t);
            lines_[L0 + t] = Span{(uint64_t)at, (uint32_t)(A.nl[t] - at), (uint8_t)(A.nl[t] > at ? fc(t) : 0)};
        }
        for (size_t r = accepted * w / T, re = accepted * (w + 1) / T; r < re; r++) {
            const uint32_t h = (uint32_t)(L0 + hdr_of(r));
            const uint32_t rb = closed ? b_first + (uint32_t)(r / m) : rec_batch_[r];
            recs_[R0 + r] = format_ == 0 ? Rec{h, h + 1, (uint32_t)(L0 + hdr_of(r + 1)), rb} : Rec{h, h + 1, h + 2, rb};
        }
    });
    src_.consume(C);
    times_.bulk_reads += accepted;
    if (closed) times_.closed_reads += accepted;
    return true;
}

// (std::isspace of the "C" locale -- space, \t \n \v \f \r -- spelled out: the call per line end was a tenth of the fill)
static inline bool is_space_c(unsigned char c) { return c == ' ' || (unsigned)(c - 9u) < 5u; }
static inline size_t rstrip_len(const char *p, size_t n) {
    while (n > 0 && is_space_c(static_cast<unsigned char>(p[n - 1]))) n--;
    return n;
}

void BatchReader::make_pool() {
    if (pool_) return;
    unsigned want = threads_ ? threads_ : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = std::getenv("MOVI_PARSE_THREADS")) want = (unsigned)std::max(1, std::atoi(e));   // (tuning hook: tools/r05_cli.sh)
    pool_.reset(new WorkerPool(want));
}

void BatchReader::warm_up(ReadSet *const *sets, unsigned n_sets, uint64_t max_bases, double lines_per_byte) {
    if (!mem_ || size_hint_ == 0) return;
    make_pool();
    const size_t window = (size_t)std::min<uint64_t>(max_bases + (max_bases >> 2) + (1u << 20), 1ull << 32);
    // Round 6 (advisor finding): the warm-up no longer READS THE INPUT when the caller can say how dense its lines are (lines_per_byte > 0,
    // from movi_main's probe of the file's first MiB): the first window's newline scan -- with the mapping's page faults -- belongs to the
    // command's read-processing clock and is left to the first next_chunk().  What remains here is set-up: the pool, and the tables and the
    // circulating chunks' buffers sized for the estimated number of lines and first touched by the pool's pinned threads.
    size_t L;
    if (lines_per_byte > 0) {
        L = (size_t)((double)std::min<uint64_t>(window, size_hint_) * lines_per_byte * 1.05) + 1024;
    } else {
        const auto t_in = std::chrono::steady_clock::now();
        src_.prescan(window, *pool_);                                  // exactly the first next_chunk's call
        warm_input_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();   // (the caller adds it to its clock)
        L = src_.prescanned_lines();
    }
    const size_t in_reach = (size_t)std::min<uint64_t>(window, size_hint_);
    // a chunk holds at most the bytes in reach of sequence, at most L / 2 records and their ids (<= header bytes: bounded by the same bytes;
    // an eighth of them covers ids of 18 characters on 150 bp reads -- a bigger chunk grows its buffers as before)
    const size_t n_rec = L / 2 + 2, id_bytes = std::min<size_t>(in_reach / 8 + 4096, 64u << 20);
    lines_.reserve(L + 1024);
    recs_.reserve(L / 2 + 1024);
    struct Region { uint8_t *p; size_t n; };
    std::vector<Region> regions;
    regions.push_back(Region{reinterpret_cast<uint8_t *>(lines_.data()), lines_.capacity() * sizeof(Span)});
    regions.push_back(Region{reinterpret_cast<uint8_t *>(recs_.data()), recs_.capacity() * sizeof(Rec)});
    for (unsigned k = 0; k < n_sets; k++) {
        ReadSet &rs = *sets[k];
        rs.bases.resize_uninitialized(in_reach);
        rs.id_bytes.resize_uninitialized(id_bytes);
        regions.push_back(Region{rs.bases.data(), in_reach});
        regions.push_back(Region{rs.id_bytes.data(), id_bytes});
        // (the offset / batch arrays are std::vectors: value-initialised here, on this thread -- 4.5 MB per set -- instead of inside the first chunks)
        rs.offsets.assign(n_rec, 0);
        rs.id_off.assign(n_rec, 0);
        rs.batch_of.assign(n_rec, 0);
    }
    const unsigned T = pool_->size();
    pool_->run(T, [&](unsigned t) {
        for (const Region &r : regions) {
            volatile uint8_t *b = r.p;
            for (size_t off = r.n / T * t & ~(size_t)4095, end = t + 1 == T ? r.n : r.n / T * (t + 1); off < end; off += 4096) b[off] = 0;
        }
    });
    for (unsigned k = 0; k < n_sets; k++) { sets[k]->bases.clear(); sets[k]->id_bytes.clear(); }
}

bool BatchReader::next_chunk(ReadSet &out, uint64_t max_bases, uint64_t min_reads, uint64_t hard_max_bases) {
    // (the offset / batch arrays keep their size across chunks: a ReadSet circulates, every element is rewritten below, and
    // value-initialising 20 MB of them per million reads was 2 ms of a 25 ms chunk)
    out.id_bytes.clear(); out.bases.clear();
    arena_.clear();
    lines_.clear();
    recs_.clear();
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    make_pool();
    // the newlines of about a chunk's worth of input, found by all workers at once (the rest, if the chunk turns out
    // longer, line by line as before)
    if (mem_) src_.prescan((size_t)std::min<uint64_t>(max_bases + (max_bases >> 2) + (1u << 20), 1ull << 32), *pool_);
    const double t_scanned = now();
    times_.prescan += t_scanned - t_begin;
    lines_.reserve(src_.prescanned_lines() + 1024);
    recs_.reserve(src_.prescanned_lines() / 2 + 1024);
    // ---- phase 1 (sequential): batches, headers, where each read's sequence lines are
    bool any = false;
    uint64_t approx_bases = 0;                                         // sequence-line bytes, trailing whitespace included
    while (approx_bases < max_bases || (recs_.size() < min_reads && approx_bases < hard_max_bases)) {
        // whole batches straight from the scanned lines where the input is regular (cut_ahead) ...
        if (detect_format() && cut_ahead(max_bases, min_reads, hard_max_bases, approx_bases)) { any = true; continue; }
        // ... else one reference batch, line by line
        size_t p = 0;
        if (!load_batch(p)) break;
        any = true;
        const uint32_t b = batch_counter_++;
        // grabNextRead over the batch (src/batch_loader.cpp:91-143)
        const size_t nl = lines_.size();
        while (p < nl) {
            const size_t hn = lines_[p].len;
            const char h0 = (char)lines_[p].first;
            if (hn == 0) break;                                        // ":99 an empty line" ends the batch
            if (format_ == 1 && h0 != '@')
                throw std::runtime_error(std::string("Incorrect FASTQ entry, it should start with '@' but found ") + h0);
            if (format_ == 0 && h0 != '>')
                throw std::runtime_error(std::string("Incorrect FASTA entry, it should start with '>' but found ") + h0);
            if (hn <= 2) throw std::runtime_error("header line is missing an id. invalid query cannot be processed.");
            Rec r{(uint32_t)p, (uint32_t)(p + 1), (uint32_t)(p + 1), b};
            p++;
            if (format_ == 1) {
                if (p >= nl) break;
                if (p + 2 >= nl) break;                                // '+' line and qualities must exist
                r.seq_end = (uint32_t)(p + 1);
                approx_bases += lines_[p].len;
                p += 3;
            } else {
                while (p < nl && (lines_[p].len == 0 || lines_[p].first != '>')) {
                    approx_bases += lines_[p].len;
                    p++;
                }
                r.seq_end = (uint32_t)p;
            }
            recs_.push_back(r);
        }
    }
    const double t_cut = now();
    times_.cut += t_cut - t_scanned;
    if (!any) {
        out.id_off.assign(1, 0); out.offsets.assign(1, 0); out.batch_of.clear();
        return false;
    }
    // ---- phase 2 (parallel): id and stripped sequence lengths, then ids and bases copied to their final offsets
    const size_t n = recs_.size();
    times_.reads += n;
    out.batch_of.resize(n);
    out.offsets.resize(n + 1);
    out.id_off.resize(n + 1);
    out.offsets[0] = 0;
    out.id_off[0] = 0;
    unsigned T = pool_->size();
    if (approx_bases < (1u << 22) || n < 64) T = 1;                    // not worth waking the workers
    auto for_ranges = [&](const std::function<void(size_t, size_t)> &fn) {
        // ranges of equal read counts: line offsets are monotone in the input, so the ranges are balanced by bytes too
        pool_->run(T, [&](unsigned t) { fn(n * t / T, n * (t + 1) / T); });
    };
    // pass A: every worker measures its range and leaves RUNNING sums inside it (offsets[i + 1] = bases of the range's reads up
    // to and including i); the ranges' totals are then chained (T additions) and pass B adds each range's base as it copies
    std::vector<uint64_t> base_bases(T + 1, 0), base_ids(T + 1, 0);
    for_ranges([&](size_t a, size_t b) {
        uint64_t run_b = 0, run_i = 0;
        for (size_t i = a; i < b; i++) {
            const Rec &r = recs_[i];
            const char *hdr = line(r.hdr);
            const size_t hn = lines_[r.hdr].len;
            size_t id_len = hn;                                        // find_first_of(" \t\r", 1)
            for (size_t k = 1; k < hn; k++)
                if (hdr[k] == ' ' || hdr[k] == '\t' || hdr[k] == '\r') { id_len = k; break; }
            // substr(1, id_len): id_len is used as a LENGTH, so the whitespace char is kept
            run_i += std::min(id_len, hn - 1);
            out.id_off[i + 1] = run_i;
            out.batch_of[i] = r.batch;
            for (size_t l = r.seq_first; l < r.seq_end; l++) run_b += rstrip_len(line(l), lines_[l].len);
            out.offsets[i + 1] = run_b;
        }
    });
    for (unsigned t = 0; t < T; t++) {                                 // range t = reads [n t / T, n (t + 1) / T)
        const size_t b = n * (t + 1) / T;
        const size_t a = n * t / T;
        base_bases[t + 1] = base_bases[t] + (b > a ? out.offsets[b] : 0);
        base_ids[t + 1] = base_ids[t] + (b > a ? out.id_off[b] : 0);
    }
    const double t_len = now();
    times_.lengths += t_len - t_cut;
    const uint64_t total = base_bases[T];
    out.bases.resize_uninitialized(total);                             // first touched by the workers below, in parallel
    out.id_bytes.resize_uninitialized(base_ids[T]);
    pool_->run(T, [&](unsigned t) {
        const size_t a = n * t / T, b = n * (t + 1) / T;
        const uint64_t bb = base_bases[t], bi = base_ids[t];
        uint64_t prev_b = 0, prev_i = 0;                               // running sums of the range up to read i - 1
        for (size_t i = a; i < b; i++) {
            const Rec &r = recs_[i];
            const uint64_t end_b = out.offsets[i + 1], end_i = out.id_off[i + 1];
            std::memcpy(out.id_bytes.data() + bi + prev_i, line(r.hdr) + 1, (size_t)(end_i - prev_i));
            uint8_t *dst = out.bases.data() + bb + prev_b;
            if (r.seq_end - r.seq_first == 1) {                        // one sequence line: its stripped length is the read's, measured in pass A
                std::memcpy(dst, line(r.seq_first), (size_t)(end_b - prev_b));
            } else {
                for (size_t l = r.seq_first; l < r.seq_end; l++) {
                    const size_t sn = rstrip_len(line(l), lines_[l].len);
                    std::memcpy(dst, line(l), sn);
                    dst += sn;
                }
            }
            out.offsets[i + 1] = bb + end_b;                           // final, absolute offsets
            out.id_off[i + 1] = bi + end_i;
            prev_b = end_b;
            prev_i = end_i;
        }
    });
    times_.copy += now() - t_len;
    return true;
}

std::vector<uint32_t> strand_order(const ReadSet &rs, const std::vector<uint64_t> &cost, size_t strands, WorkerPool *pool) {
    const size_t n = rs.size();
    std::vector<uint32_t> order(n);
    // the records of batch [i, j) fill order[i .. j): the batches are independent of each other, so a pool takes ranges of them
    // (1 M x 150 bp: 3 ms of a 12 ms chunk on one thread -- while the writer behind it sat idle)
    auto range = [&](size_t lo, size_t hi) {                           // reads [lo, hi), both on batch boundaries
        typedef std::pair<uint64_t, uint32_t> Ev;                      // (finish round, strand)
        std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> pq;  // (empty again after every batch)
        std::vector<uint32_t> cur(strands, 0);
        size_t i = lo;
        while (i < hi) {
            size_t j = i;
            while (j < hi && rs.batch_of[j] == rs.batch_of[i]) j++;
            size_t next = i, at = i;
            for (uint32_t s = 0; s < strands && next < j; s++, next++) {
                cur[s] = (uint32_t)next;
                pq.push(Ev(cost[next] ? cost[next] : 1, s));
            }
            while (!pq.empty()) {
                const Ev e = pq.top();
                pq.pop();
                order[at++] = cur[e.second];
                if (next < j) {
                    cur[e.second] = (uint32_t)next;
                    pq.push(Ev(e.first + (cost[next] ? cost[next] : 1), e.second));
                    next++;
                }
            }
            i = j;
        }
    };
    const unsigned P = pool && n >= 4096 ? pool->size() * 2u : 1u;
    if (P == 1) { range(0, n); return order; }
    std::vector<size_t> cut(P + 1, n);                                 // part boundaries moved up to the next batch boundary
    cut[0] = 0;
    for (unsigned p = 1; p < P; p++) {
        size_t k = std::max(cut[p - 1], n * p / P);
        while (k > 0 && k < n && rs.batch_of[k] == rs.batch_of[k - 1]) k++;
        cut[p] = k;
    }
    pool->run(P, [&](unsigned p) { range(cut[p], cut[p + 1]); });
    return order;
}

}  // namespace movi_host
