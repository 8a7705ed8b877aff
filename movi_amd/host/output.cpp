#include "output.hpp"

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <stdexcept>

namespace movi_host {

void write_bpf_header(std::ostream &f, uint8_t entry_size) {
    // struct BPFHeader {u32 magic; u8 major, minor, patch; u8 entry_size; u16 reserved;} is
    // written with sizeof == 12: bytes 10-11 are struct padding (uninitialised in the reference,
    // zero here).  include/utils.hpp:64-82, version numbers include/version.hpp:12-14.
    uint8_t h[12] = {0};
    std::memcpy(h, &kBpfMagic, 4);
    h[4] = 1; h[5] = 0; h[6] = 0;
    h[7] = entry_size;
    f.write(reinterpret_cast<const char *>(h), 12);
}

void write_bpf_record(std::ostream &f, const std::string &id, const uint16_t *pml, uint64_t n) {
    uint16_t st_length = static_cast<uint16_t>(id.length());          // src/utils.cpp:222
    f.write(reinterpret_cast<const char *>(&st_length), 2);
    f.write(id.data(), st_length);
    f.write(reinterpret_cast<const char *>(&n), 8);                   // output_binary :204-210
    f.write(reinterpret_cast<const char *>(pml), (std::streamsize)(n * 2));
}

void write_stdout_pmls(std::ostream &out, const std::string &id, const uint16_t *pml, uint64_t n) {
    // add_ml appends " " + reversed digits per value and the whole string is reversed once at the
    // end: the net effect is the values in read order, each followed by one space.
    std::string line;
    line.reserve(n * 3 + 1);
    char buf[8];
    for (uint64_t i = n; i-- > 0;) {
        unsigned v = pml[i];
        int k = 0;
        do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (k) line.push_back(buf[--k]);
        line.push_back(' ');
    }
    out << ">" << id << "\n" << line << "\n";
}

void write_count_line(std::ostream &out, const std::string &id, uint64_t query_length, uint64_t matched, uint64_t count) {
    out << id << "\t" << matched << "/" << query_length << "\t" << count << "\n";
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

bool Classifier::classify(const std::string &read_name, const uint16_t *pml, uint64_t n, size_t bin_width,
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

int view_bpf(const Options &o, std::ostream &out) {                   // src/movi.cpp:402-503
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

}  // namespace movi_host
