#include "output.hpp"
#include <charconv>

#include <cerrno>
#include <cstdio>

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace movi_host {

BpfWriter::~BpfWriter() {
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

void BpfWriter::close() {
    if (fd_ >= 0 && ::close(fd_) != 0) { fd_ = -1; throw std::runtime_error("Failed to write the output file: " + path_); }
    fd_ = -1;
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
