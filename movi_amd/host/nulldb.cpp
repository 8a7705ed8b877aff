#include "nulldb.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace movi_host {

std::vector<std::string> read_fasta_sequences(const std::string &path) {
    std::ifstream in(path);
    if (!in.good()) throw std::runtime_error("Failed to open the fasta file: " + path);
    std::vector<std::string> seqs;
    std::string line;
    bool in_record = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            seqs.emplace_back();
            in_record = true;
        } else if (in_record) {
            seqs.back() += line;
        }
    }
    return seqs;
}

size_t generate_null_reads(const std::string &ref_fasta, const std::string &out_path, unsigned seed) {
    const size_t chunk = 150, num_null_reads = 800, null_read_bound = 1000;   // include/utils.hpp:165-167
    std::srand(seed);
    const std::vector<std::string> seqs = read_fasta_sequences(ref_fasta);
    std::ofstream out(out_path);
    if (!out.good()) throw std::runtime_error("Failed to open the null reads file: " + out_path);
    size_t total = 0;
    bool go = total < null_read_bound;
    for (const std::string &s : seqs) {
        if (!go) break;
        const size_t reads_to_grab = total >= num_null_reads ? 25 : 100;       // downsample once enough exist
        for (size_t i = 0; i < reads_to_grab && go && s.size() > chunk; i++) {
            const size_t at = (size_t)std::rand() % (s.size() - chunk);
            std::string read = s.substr(at, chunk);
            if (read.find('N') == std::string::npos) {
                std::reverse(read.begin(), read.end());
                out << ">read_" << total << "\n" << read << "\n";
                total++;
                go = total < null_read_bound;
            }
        }
        if (s.size() <= chunk) {                                               // short record: taken whole
            std::string read = s;
            std::reverse(read.begin(), read.end());
            out << ">read_" << total << "\n" << read << "\n";
            total++;
        }
    }
    return total;
}

NullStats compute_null_stats(const std::vector<uint64_t> &values) {
    if (values.empty()) throw std::runtime_error("The null reads produced no matching lengths.");
    NullStats st;
    st.num_values = values.size();
    double sum = 0.0;
    for (uint64_t v : values) sum += (double)v;
    st.mean = sum / (double)values.size();
    std::vector<uint64_t> sorted(values);
    std::sort(sorted.begin(), sorted.end());
    uint64_t largest = 0, cur = sorted[0];
    size_t occ = 0;
    for (uint64_t x : sorted) {
        if (x == cur) {
            occ++;
        } else {
            if (occ >= 5) largest = cur;
            cur = x;
            occ = 1;
        }
    }
    if (occ >= 5) largest = cur;
    st.percentile_value = largest;
    return st;
}

void write_null_db(const std::string &path, const NullStats &st, const std::vector<uint64_t> &values) {
    std::ofstream out(path, std::ios::out | std::ios::binary);
    if (!out.good()) throw std::runtime_error("Failed to open the null database for writing: " + path);
    out.write(reinterpret_cast<const char *>(&st.num_values), 8);
    out.write(reinterpret_cast<const char *>(&st.mean), 8);
    out.write(reinterpret_cast<const char *>(&st.percentile_value), 8);
    out.write(reinterpret_cast<const char *>(values.data()), (std::streamsize)(values.size() * 8));
}

}  // namespace movi_host
