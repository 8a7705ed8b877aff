#include "reads.hpp"

#include <cctype>
#include <queue>
#include <stdexcept>

namespace movi_host {

// One reference batch = the lines loadBatch would put into its stringstream
// (src/batch_loader.cpp:50-87): lines are read until BOTH >= 1000 "bases" and >= min_reads
// reads are covered.  FASTQ: a read is counted every 4 lines with (record bytes)/2 bases;
// FASTA: a read is counted when the NEXT line starts with '>' with (record bytes) bases.
bool BatchReader::load_batch(std::vector<std::string> &lines) {
    lines.clear();
    if (format_ < 0) {
        if (!in_.good()) return false;
        int c = in_.peek();
        if (c == '>') format_ = 0;
        else if (c == '@') format_ = 1;
        else if (c == std::char_traits<char>::eof()) return false;
        else throw std::runtime_error("unrecognized input query file type - expects FASTA or FASTQ.");
    }
    size_t bases = 0, reads = 0, nlines = 0, record = 0;
    const size_t num_bases = 1000;
    std::string buf;
    bool valid = false;
    while (in_.good() && (bases < num_bases || reads < min_reads_)) {
        if (!std::getline(in_, buf)) {
            if (format_ == 1 && nlines % 4 == 0) return valid || !lines.empty();
            if (format_ == 0 && nlines % 2 == 0) return valid || !lines.empty();
            // the reference returns false here and drops the partial batch (:57-63)
            lines.clear();
            return false;
        }
        nlines++;
        record += buf.size();
        valid = true;
        if (format_ == 1) {
            if (nlines % 4 == 0) { bases += record / 2; record = 0; reads++; }
        } else if (in_.peek() == '>') {
            bases += record; record = 0; reads++;
        }
        lines.push_back(buf);
    }
    return valid;
}

static void strip_trailing_space(std::string &s) {
    while (!s.empty() && std::isspace(static_cast<unsigned char>(s.back()))) s.pop_back();
}

bool BatchReader::next_chunk(ReadSet &out, uint64_t max_bases) {
    out.ids.clear(); out.bases.clear(); out.offsets.assign(1, 0); out.batch_of.clear();
    std::vector<std::string> lines;
    bool any = false;
    while (out.bases.size() < max_bases) {
        if (!load_batch(lines)) break;
        any = true;
        const uint32_t b = batch_counter_++;
        // grabNextRead over the batch (src/batch_loader.cpp:91-143)
        size_t p = 0;
        while (p < lines.size()) {
            const std::string &hdr = lines[p];
            if (hdr.empty()) break;                                    // ":99 an empty line" ends the batch
            if (format_ == 1 && hdr[0] != '@')
                throw std::runtime_error(std::string("Incorrect FASTQ entry, it should start with '@' but found ") + hdr[0]);
            if (format_ == 0 && hdr[0] != '>')
                throw std::runtime_error(std::string("Incorrect FASTA entry, it should start with '>' but found ") + hdr[0]);
            if (hdr.size() <= 2) throw std::runtime_error("header line is missing an id. invalid query cannot be processed.");
            size_t id_len = hdr.find_first_of(" \t\r", 1);
            if (id_len == std::string::npos) id_len = hdr.size();
            std::string id = hdr.substr(1, id_len);                    // NB: a length, so the whitespace char is kept
            p++;
            std::string seq;
            if (format_ == 1) {
                if (p >= lines.size()) break;
                seq = lines[p++];
                strip_trailing_space(seq);
                if (p + 1 >= lines.size()) break;                      // '+' line and qualities must exist
                p += 2;
            } else {
                while (p < lines.size() && (lines[p].empty() || lines[p][0] != '>')) {
                    std::string l = lines[p++];
                    strip_trailing_space(l);
                    seq += l;
                }
            }
            out.ids.push_back(std::move(id));
            out.bases.insert(out.bases.end(), seq.begin(), seq.end());
            out.offsets.push_back(out.bases.size());
            out.batch_of.push_back(b);
        }
    }
    return any;
}

std::vector<uint32_t> strand_order(const ReadSet &rs, const std::vector<uint64_t> &cost, size_t strands) {
    std::vector<uint32_t> order;
    order.reserve(rs.size());
    size_t i = 0;
    const size_t n = rs.size();
    typedef std::pair<uint64_t, uint32_t> Ev;                          // (finish round, strand)
    while (i < n) {
        size_t j = i;
        while (j < n && rs.batch_of[j] == rs.batch_of[i]) j++;
        // batch = reads [i, j)
        std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> pq;
        std::vector<uint32_t> cur(strands, 0);
        size_t next = i;
        for (uint32_t s = 0; s < strands && next < j; s++, next++) {
            cur[s] = (uint32_t)next;
            pq.push(Ev(cost[next] ? cost[next] : 1, s));
        }
        while (!pq.empty()) {
            Ev e = pq.top();
            pq.pop();
            order.push_back(cur[e.second]);
            if (next < j) {
                cur[e.second] = (uint32_t)next;
                pq.push(Ev(e.first + (cost[next] ? cost[next] : 1), e.second));
                next++;
            }
        }
        i = j;
    }
    return order;
}

}  // namespace movi_host
