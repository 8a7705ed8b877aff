// build_index.cpp -- index constructor for modes 2 / 3 / 5 / 6 / 7 / 8 (bench + test tooling; SURVEY section 8(f) "next #2").
//
// FASTA (or a synthetic pangenome) -> cleaned text with reverse complements -> suffix array (SA-IS) ->
// BWT + LCP (Kasai) -> per-run thresholds -> move rows -> `index.movi` bytes.  Linear time, so that
// the "real-structure" workload of SURVEY section 8(d)(i) -- a 64-genome pangenome with ~10 M runs, 640 Mbp of
// text -- can be built in minutes.  What is computed follows the reference (file:line under
// /root/reference):
//   src/prepare_ref.cpp:39-58            cleaning ("not upper-case ACGT -> 'A'"), reverse complement per record
//   pfp-thresholds (external, tag `movi`) BWT with terminator byte 0; threshold of a run = leftmost minimum LCP
//                                         between the end of the previous run of that character and the run start
//   src/move_structure_build.cpp:328-396 row boundaries: character change, threshold position, MAX_RUN_LENGTH
//   :74-121, :449-692                    LF of run heads -> (id, offset); :694-731 base intervals
//   :807-935                             threshold bits (incl. the '$'-row-is-an-'A'-row quirk, :823)
//   :939-1074                            blocked ids;  src/move_structure_io.cpp:435-469 file layout
// Checked byte-for-byte against tests/golden/index_*/index.movi (whose sizes are the reference's
// known answers 948119 / 711733, tests/test_build.cpp:37,53) in tests/test_build_tool.py.
//
// usage: build_index fasta <ref.fasta> <mode 2|3|5|6|7|8> <out_dir> [separators]     ("separators" = movi build --separators:
//            every record and reverse complement followed by %; sizes 948232 / 711854 B, tests/test_build.cpp:79,95)
//        build_index pangenome <ancestor_len> <n_genomes> <snp_rate> <seed> <mode> <out_dir> [n_reads read_len sub_rate]
//            [n_reads2 read_len2 sub_rate2]
//            (also writes <out_dir>/text.bin and reads.bin (and reads2.bin): fixed-length substrings of the text
//            with substitutions)
//        build_index reads <text.bin> <n_reads> <read_len> <sub_rate> <seed> <out_file>
//        build_index pangenome <ancestor_len> <n_genomes> <snp_rate> <seed> <mode> <out_dir> text-only     (writes text.bin only)
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <vector>

typedef int32_t sa_t;

// ------------------------------------------------------------------ SA-IS (induced sorting), own implementation
namespace sais {

template <typename T>
static void get_counts(const T *s, sa_t *c, sa_t n, sa_t k) {
    for (sa_t i = 0; i < k; i++) c[i] = 0;
    for (sa_t i = 0; i < n; i++) c[s[i]]++;
}
static void get_buckets(const sa_t *c, sa_t *b, sa_t k, bool end) {
    sa_t sum = 0;
    if (end) for (sa_t i = 0; i < k; i++) { sum += c[i]; b[i] = sum; }
    else for (sa_t i = 0; i < k; i++) { sum += c[i]; b[i] = sum - c[i]; }
}

template <typename T>
static void induce(const std::vector<bool> &is_s, sa_t *sa, const T *s, sa_t *c, sa_t *b, sa_t n, sa_t k) {
    get_buckets(c, b, k, false);                      // L-type, left to right
    for (sa_t i = 0; i < n; i++) {
        sa_t j = sa[i] - 1;
        if (sa[i] > 0 && !is_s[j]) sa[b[s[j]]++] = j;
    }
    get_buckets(c, b, k, true);                       // S-type, right to left
    for (sa_t i = n - 1; i >= 0; i--) {
        sa_t j = sa[i] - 1;
        if (sa[i] > 0 && is_s[j]) sa[--b[s[j]]] = j;
    }
}

// s[n-1] must be the unique smallest symbol (sentinel).
template <typename T>
static void build(const T *s, sa_t *sa, sa_t n, sa_t k) {
    std::vector<bool> is_s(n);
    is_s[n - 1] = true;
    for (sa_t i = n - 2; i >= 0; i--) is_s[i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && is_s[i + 1]);
    auto is_lms = [&](sa_t i) { return i > 0 && is_s[i] && !is_s[i - 1]; };
    std::vector<sa_t> c(k), b(k);
    get_counts(s, c.data(), n, k);
    // stage 1: sort LMS substrings
    get_buckets(c.data(), b.data(), k, true);
    for (sa_t i = 0; i < n; i++) sa[i] = -1;
    for (sa_t i = 1; i < n; i++) if (is_lms(i)) sa[--b[s[i]]] = i;
    induce(is_s, sa, s, c.data(), b.data(), n, k);
    // compact sorted LMS substrings, name them
    sa_t n1 = 0;
    for (sa_t i = 0; i < n; i++) if (is_lms(sa[i])) sa[n1++] = sa[i];
    for (sa_t i = n1; i < n; i++) sa[i] = -1;
    sa_t name = 0, prev = -1;
    for (sa_t i = 0; i < n1; i++) {
        sa_t pos = sa[i];
        bool diff = false;
        if (prev < 0) diff = true;
        else {
            for (sa_t d = 0;; d++) {
                if (s[pos + d] != s[prev + d] || is_s[pos + d] != is_s[prev + d]) { diff = true; break; }
                if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
            }
        }
        if (diff) { name++; prev = pos; }
        sa[n1 + pos / 2] = name - 1;
    }
    for (sa_t i = n - 1, j = n - 1; i >= n1; i--) if (sa[i] >= 0) sa[j--] = sa[i];
    // stage 2: solve the reduced problem
    sa_t *sa1 = sa, *s1 = sa + n - n1;
    if (name < n1) {
        build<sa_t>(s1, sa1, n1, name);
    } else {
        for (sa_t i = 0; i < n1; i++) sa1[s1[i]] = i;
    }
    // stage 3: induce the result
    get_buckets(c.data(), b.data(), k, true);
    for (sa_t i = 1, j = 0; i < n; i++) if (is_lms(i)) s1[j++] = i;
    for (sa_t i = 0; i < n1; i++) sa1[i] = s1[sa1[i]];
    for (sa_t i = n1; i < n; i++) sa[i] = -1;
    for (sa_t i = n1 - 1; i >= 0; i--) {
        sa_t j = sa[i];
        sa[i] = -1;
        sa[--b[s[j]]] = j;
    }
    induce(is_s, sa, s, c.data(), b.data(), n, k);
}

}  // namespace sais

// ------------------------------------------------------------------------------------------------ helpers
static const int alphamap_3[4][4] = {{3, 0, 1, 2}, {0, 3, 1, 2}, {0, 1, 3, 2}, {0, 1, 2, 3}};   // src/utils.cpp:5-8

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void append_clean(std::vector<uint8_t> &text, const std::string &seq, bool separators = false) {
    // src/prepare_ref.cpp:39-58: the test uses the ORIGINAL byte, so lower case also becomes 'A';
    // with separators (:61-66) the record and its reverse complement are each followed by '%'
    const size_t a = text.size();
    for (char ch : seq) text.push_back((ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? (uint8_t)ch : (uint8_t)'A');
    const size_t b = text.size();
    if (separators) text.push_back('%');
    for (size_t i = b; i-- > a;) {
        uint8_t c = text[i];
        text.push_back(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A');
    }
    if (separators) text.push_back('%');
}

static void put64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }

// ------------------------------------------------------------------------------- text -> index.movi bytes
static std::vector<uint8_t> build_index(std::vector<uint8_t> &text, int mode) {
    const bool with_thresholds = mode != 5 && mode != 3 && mode != 2;     // sampled / regular / blocked: USE_THRESHOLDS is off
    text.push_back(0);                                                    // terminator
    const sa_t n = (sa_t)text.size();
    if ((uint64_t)text.size() >= (1ull << 31)) { fprintf(stderr, "text too long for 32-bit suffix array\n"); exit(1); }
    std::vector<sa_t> sa(n);
    fprintf(stderr, "[build_index] n = %d, building the suffix array...\n", n);
    sais::build<uint8_t>(text.data(), sa.data(), n, 256);
    // LCP (Kasai): lcp[i] = lcp(suffix sa[i-1], suffix sa[i])
    fprintf(stderr, "[build_index] LCP...\n");
    std::vector<sa_t> rank(n), lcp(n, 0);
    for (sa_t i = 0; i < n; i++) rank[sa[i]] = i;
    {
        sa_t h = 0;
        for (sa_t i = 0; i < n; i++) {
            sa_t r = rank[i];
            if (r > 0) {
                sa_t j = sa[r - 1];
                while (i + h < n && j + h < n && text[i + h] == text[j + h]) h++;
                lcp[r] = h;
                if (h > 0) h--;
            } else h = 0;
        }
    }
    std::vector<sa_t>().swap(rank);
    std::vector<uint8_t> bwt(n);
    for (sa_t i = 0; i < n; i++) bwt[i] = sa[i] ? text[sa[i] - 1] : 0;
    std::vector<sa_t>().swap(sa);
    // thresholds per original run + the split bit vector (fill_bits_by_thresholds :733-746)
    fprintf(stderr, "[build_index] thresholds + rows...\n");
    std::vector<bool> hard(n + 1, false);
    hard[0] = true;
    std::vector<uint64_t> thr;                                            // per original run
    {
        const sa_t INF = 0x7fffffff;
        sa_t curmin[256], arg[256];
        bool seen[256];
        for (int c = 0; c < 256; c++) { curmin[c] = INF; arg[c] = 0; seen[c] = false; }
        const int syms[6] = {0, '%', 'A', 'C', 'G', 'T'};
        for (sa_t i = 0; i < n; i++) {
            if (i > 0) for (int c : syms) if (lcp[i] < curmin[c]) { curmin[c] = lcp[i]; arg[c] = i; }
            const uint8_t c = bwt[i];
            if (i == 0 || bwt[i - 1] != c) {
                hard[i] = true;
                const uint64_t t = seen[c] ? (uint64_t)arg[c] : 0;
                thr.push_back(t);
            }
            seen[c] = true;
            curmin[c] = INF;                                              // the range restarts after this position
        }
    }
    if (with_thresholds) for (uint64_t t : thr) hard[t] = true;           // rows split at thresholds (:733-746)
    std::vector<sa_t>().swap(lcp);
    const uint64_t original_r = thr.size();
    const uint32_t maxrun = mode == 6 ? 2047 : (mode == 7 ? 511 : (mode == 3 ? 4095 : 1023));   // move_row_configs.hpp:51,101,135,117,31,72
    // rows (:328-396)
    std::vector<uint64_t> all_p;
    {
        uint64_t start = 0;
        for (uint64_t i = 1; i <= (uint64_t)n; i++) {
            if (i == (uint64_t)n || hard[i] || i - start == maxrun) { all_p.push_back(start); start = i; }
        }
    }
    const uint64_t r = all_p.size();
    all_p.push_back((uint64_t)n);
    // alphabet (build_alphabet :428-447)
    uint64_t alphamap[256], cnt_all[256] = {0};
    for (sa_t i = 0; i < n; i++) cnt_all[bwt[i]]++;
    std::vector<uint8_t> alphabet;
    std::vector<uint64_t> counts;
    for (int c = 0; c < 256; c++) alphamap[c] = 256;
    for (int c = 1; c < 256; c++) if (cnt_all[c]) { alphamap[c] = alphabet.size(); alphabet.push_back((uint8_t)c); counts.push_back(cnt_all[c]); }
    const size_t sigma = alphabet.size();
    // MoveStructure::use_separator, src/move_structure.cpp:547-552: five symbols led by '%'
    const int sep = (sigma == 5 && alphabet[0] == '%') ? 1 : 0;
    if (sigma < 1 || (sigma > 4 && !sep)) { fprintf(stderr, "only DNA alphabets (<= 4 symbols, or '%%' + ACGT) are in scope\n"); exit(1); }
    std::vector<uint8_t> code(r);
    std::vector<uint16_t> lens(r), doff(r);
    std::vector<uint64_t> dest(r);
    uint64_t end_bwt_idx = 0;
    for (uint64_t i = 0; i < r; i++) {
        const uint8_t h = bwt[all_p[i]];
        lens[i] = (uint16_t)(all_p[i + 1] - all_p[i]);
        if (h == 0) { end_bwt_idx = i; code[i] = 0; }                      // set_c: alphamap[0] == 256 shifts out
        else code[i] = (uint8_t)alphamap[h];
    }
    {   // LF of run heads (LF_heads, src/move_structure.cpp:515-523); destinations are monotone per character
        uint64_t C[5], rk[5] = {0, 0, 0, 0, 0}, ptr[5] = {0, 0, 0, 0, 0};
        C[0] = 1;
        for (size_t a = 1; a < sigma; a++) C[a] = C[a - 1] + counts[a - 1];
        for (uint64_t i = 0; i < r; i++) {
            if (i == end_bwt_idx) { dest[i] = 0; doff[i] = 0; continue; }
            const int a = code[i];
            const uint64_t lf = C[a] + rk[a];
            rk[a] += lens[i];
            uint64_t p = ptr[a];
            while (all_p[p + 1] <= lf) p++;
            ptr[a] = p;
            dest[i] = p;
            doff[i] = (uint16_t)(lf - all_p[p]);
        }
    }
    // base intervals (:694-731)
    std::vector<uint64_t> first_runs(1, 0), first_offsets(1, 0), last_runs(1, 0), last_offsets(1, 0);
    {
        uint64_t cc = 1;
        for (size_t a = 0; a < sigma; a++) {
            const uint64_t lr = last_runs.back(), lo = last_offsets.back();
            if (lo + 1 >= lens[lr]) { first_runs.push_back(lr + 1); first_offsets.push_back(0); }
            else { first_runs.push_back(lr); first_offsets.push_back(lo + 1); }
            cc += counts[a];
            const uint64_t k = std::lower_bound(all_p.begin(), all_p.begin() + r, cc) - all_p.begin();   // rows starting before cc
            last_runs.push_back(k - 1);
            last_offsets.push_back(cc - all_p[k - 1] - 1);
        }
    }
    // threshold bits (:807-935).  With separators (:826-831, :836-858, :912-921) no threshold is kept FOR the
    // separator; a row OF the separator (and the '$' row, whose character field decodes as one) gets an explicit
    // 4-value entry in separators_thresholds, appended in descending row order, row 0 last.
    std::vector<uint8_t> tbits(r, 0);
    uint64_t end_thr[4] = {0, 0, 0, 0};
    std::vector<std::array<uint16_t, 4>> sep_thr;
    std::vector<std::pair<uint64_t, uint64_t>> sep_map;
    {
        std::vector<uint64_t> at(sigma, (uint64_t)n);
        uint64_t thr_i = original_r - 1;
        for (uint64_t i = r - 1; with_thresholds && i > 0; --i) {
            const int rc = code[i];                                       // '$' row has c == 0 -> behaves as 'A' (:823)
            if (sep && rc == 0) {
                sep_thr.push_back({0, 0, 0, 0});
                sep_map.emplace_back(i, sep_thr.size() - 1);
            }
            for (size_t j = 0; j < sigma; j++) {
                if ((int)j == rc) {
                    at[j] = thr[thr_i];
                } else {
                    if (sep && j == 0) continue;                          // :849-852
                    const uint64_t cur = at[j];
                    int bit;
                    uint64_t val;
                    if (cur >= all_p[i] + lens[i]) { val = lens[i]; bit = 1; }
                    else if (cur <= all_p[i]) { val = 0; bit = 0; }
                    else { val = cur - all_p[i]; bit = -1; }              // strictly inside the row (:869-871)
                    if (i == end_bwt_idx) end_thr[j - sep] = val;
                    else if (sep && rc == 0) sep_thr.back()[j - 1] = (uint16_t)val;
                    else {
                        if (bit < 0) { fprintf(stderr, "threshold strictly inside a row\n"); exit(1); }
                        tbits[i] |= (uint8_t)(bit << alphamap_3[rc - sep][j - sep]);
                    }
                }
            }
            if (code[i] != code[i - 1] || i == end_bwt_idx || i - 1 == end_bwt_idx) thr_i--;
        }
        if (!with_thresholds) {
            // no thresholds of any kind
        } else if (sep && code[0] == 0) {                                 // :917-920
            sep_thr.push_back({0, 0, 0, 0});
            sep_map.emplace_back(0, sep_thr.size() - 1);
        } else {
            tbits[0] = 0;
        }
    }
    // blocked ids (:939-1074)
    std::vector<uint32_t> blocked, id_blocks;
    uint64_t n_blocks = 0, block_size = mode == 2 ? 1ull << 22 : 1ull << 20;   // move_row_configs.hpp:73 / :102
    uint64_t max_allowed = mode == 2 ? (1ull << 24) - 1 : (1ull << 22) - 1;     // :74 / :103
    if (mode == 8 || mode == 2) {
        blocked.resize(r);
        for (;;) {
            n_blocks = (r + block_size - 1) / block_size;
            id_blocks.assign(sigma * n_blocks, 0);
            uint64_t last[5] = {0, 0, 0, 0, 0};
            bool ok = true;
            for (uint64_t i = 0; i < r && ok; i++) {
                if (i % block_size == 0) for (size_t a = 0; a < sigma; a++) id_blocks[a * n_blocks + i / block_size] = (uint32_t)last[a];
                if (i == end_bwt_idx) { blocked[i] = 0; continue; }
                const int c = code[i];
                const uint64_t adj = dest[i] - first_runs[c + 1];
                const uint64_t b = adj - id_blocks[c * n_blocks + i / block_size];
                if (b > max_allowed) { ok = false; break; }
                blocked[i] = (uint32_t)b;
                last[c] = adj;
            }
            if (ok) break;
            block_size /= 2;
            max_allowed = ((max_allowed + 1) / 2) - 1;
        }
    }
    // sampled ids (mode 7; src/move_structure_build.cpp:486-496, :571-596, :677-682): every 20 rows the destination id
    // of the latest run of each character; a character not seen yet gets the id of its first run once it shows up
    const uint64_t tally_cp = 20;                                         // movi_options.hpp:257
    std::vector<uint64_t> tally;
    uint64_t n_tally = 0;
    if (mode == 7 || mode == 5) {
        n_tally = r / tally_cp + 2;
        tally.assign(sigma * n_tally, 0);
        std::vector<uint64_t> cur(sigma, r);
        for (uint64_t i = 0; i < r; i++) {
            if (i != end_bwt_idx) {
                const int a = code[i];
                if (cur[a] == r) for (uint64_t t = 0; t <= i / tally_cp; t++) tally[a * n_tally + t] = dest[i];
                cur[a] = dest[i];
            }
            if (i % tally_cp == 0) for (size_t a = 0; a < sigma; a++) tally[a * n_tally + i / tally_cp] = cur[a];
        }
        for (size_t a = 0; a < sigma; a++) tally[a * n_tally + n_tally - 1] = cur[a];
    }
    // serialize (src/move_structure_io.cpp:435-469)
    std::vector<uint8_t> o;
    o.reserve(2215 + r * 8 + 512 + id_blocks.size() * 4);
    o.resize(48, 0);
    const uint32_t magic = 0x4D4F5649u;
    memcpy(&o[0], &magic, 4);
    o[4] = 2; o[5] = 0; o[6] = 0; o[7] = (uint8_t)mode; o[8] = 0;
    const uint64_t hdr[4] = {(uint64_t)n, r, original_r, end_bwt_idx};
    memcpy(&o[16], hdr, 32);
    for (int i = 0; i < 4; i++) put64(o, end_thr[i]);
    for (int i = 0; i < 8; i++) put64(o, 0);
    put64(o, 256);
    for (int c = 0; c < 256; c++) put64(o, alphamap[c]);
    put64(o, sigma);
    for (uint8_t a : alphabet) o.push_back(a);
    o.push_back(0); o.push_back(0); o.push_back(0);                       // u16 nt_splitting, bool constant
    for (uint64_t i = 0; i < r; i++) {
        const uint32_t t = tbits[i];
        uint16_t w[4];
        if (mode == 6 || mode == 3) {                                      // mode 3: 12-bit n / offset, no threshold bits (t == 0)
            const uint64_t d = dest[i];
            w[0] = (uint16_t)(d & 0xFFFF);
            w[1] = (uint16_t)((d >> 16) & 0xFFFF);
            w[2] = (uint16_t)(lens[i] | (((t >> 1) & 1) << 11) | (((t >> 2) & 1) << 12) | ((uint32_t)code[i] << 13));
            w[3] = (uint16_t)(doff[i] | ((t & 1) << 11) | ((uint32_t)(d >> 32) << 12));
            const uint8_t *p = reinterpret_cast<const uint8_t *>(w);
            o.insert(o.end(), p, p + 8);
        } else if (mode == 2) {                                            // blocked: 24-bit id, MoveRow::set_id src/move_row.cpp:213-225
            const uint32_t b = blocked[i];
            w[0] = (uint16_t)(b & 0xFFFF);
            w[1] = (uint16_t)(lens[i] | (((b >> 16) & 0x3F) << 10));
            w[2] = (uint16_t)(doff[i] | ((uint32_t)code[i] << 10) | ((b >> 22) << 14));
            const uint8_t *p = reinterpret_cast<const uint8_t *>(w);
            o.insert(o.end(), p, p + 6);
        } else if (mode == 5) {                                            // sampled, no thresholds: configs :107-118
            o.push_back((uint8_t)(lens[i] & 0xFF));
            o.push_back((uint8_t)(doff[i] & 0xFF));
            o.push_back((uint8_t)((doff[i] >> 8) | ((lens[i] >> 8) << 2) | ((uint32_t)code[i] << 4)));
        } else if (mode == 7) {                                            // move_row.hpp:122-127, configs :120-136
            o.push_back((uint8_t)(lens[i] & 0xFF));
            o.push_back((uint8_t)(doff[i] & 0xFF));
            o.push_back((uint8_t)((doff[i] >> 8) | ((lens[i] >> 8) << 1) | ((uint32_t)code[i] << 2) | ((t & 7) << 5)));
        } else {
            const uint32_t b = blocked[i];
            w[0] = (uint16_t)(b & 0xFFFF);
            w[1] = (uint16_t)(lens[i] | ((b >> 16) << 10));
            w[2] = (uint16_t)(doff[i] | ((uint32_t)code[i] << 10) | ((t & 1) << 13) | (((t >> 1) & 1) << 14) | (((t >> 2) & 1) << 15));
            const uint8_t *p = reinterpret_cast<const uint8_t *>(w);
            o.insert(o.end(), p, p + 6);
        }
    }
    if (mode == 7 || mode == 5) {                                         // write_tally_table, io.cpp:328-336
        const uint32_t cp32 = (uint32_t)tally_cp;
        for (int b = 0; b < 4; b++) o.push_back((uint8_t)(cp32 >> (8 * b)));
        put64(o, n_tally);
        for (uint64_t v : tally) for (int b = 0; b < 5; b++) o.push_back((uint8_t)(v >> (8 * b)));   // MoveTally: u32 low | u8 high
    }
    for (int i = 0; i < 3; i++) put64(o, 0);
    put64(o, counts.size());
    for (uint64_t c : counts) put64(o, c);
    put64(o, last_runs.size());
    for (uint64_t v : last_runs) put64(o, v);
    for (uint64_t v : last_offsets) put64(o, v);
    for (uint64_t v : first_runs) put64(o, v);
    for (uint64_t v : first_offsets) put64(o, v);
    if (mode == 8 || mode == 2) {
        put64(o, n_blocks);
        const uint8_t *p = reinterpret_cast<const uint8_t *>(id_blocks.data());
        o.insert(o.end(), p, p + id_blocks.size() * 4);
        put64(o, block_size);
    }
    if (sep && with_thresholds) {                                         // write_separators_thresholds, io.cpp:399-413
        put64(o, sep_thr.size());
        for (const auto &t : sep_thr) for (int k = 0; k < 4; k++) { o.push_back((uint8_t)(t[k] & 0xFF)); o.push_back((uint8_t)(t[k] >> 8)); }
        std::sort(sep_map.begin(), sep_map.end());                        // the reference walks an unordered_map; ascending rows here
        put64(o, sep_map.size());
        for (const auto &kv : sep_map) { put64(o, kv.first); put64(o, kv.second); }
    }
    fprintf(stderr, "[build_index] n = %d, original_r = %llu, r = %llu, n/r = %.2f, index %zu bytes\n", n,
            (unsigned long long)original_r, (unsigned long long)r, (double)n / r, o.size());
    text.pop_back();
    return o;
}

[[maybe_unused]] static void write_file(const std::string &path, const std::vector<uint8_t> &data) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(data.data()), (std::streamsize)data.size());
    if (!f.good()) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
}

// Fixed-length substrings of the text with substitutions (rate sub_rate) and 0.1 % 'N'.
[[maybe_unused]] static void draw_reads(const std::vector<uint8_t> &text, std::vector<uint8_t> &reads, uint64_t n_reads, uint64_t read_len,
                       double sub_rate, uint64_t sr) {
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint64_t pos = splitmix64(sr) % (text.size() - read_len);
        memcpy(&reads[i * read_len], &text[pos], read_len);
        for (uint64_t k = 0; k < read_len; k++) {
            const double u = (double)(splitmix64(sr) >> 11) * (1.0 / 9007199254740992.0);
            if (u < 0.001) reads[i * read_len + k] = 'N';
            else if (u < 0.001 + sub_rate) reads[i * read_len + k] = "ACGT"[splitmix64(sr) & 3];
        }
    }
}

// FASTA -> the indexed text (prepare_ref: every record forward + reverse complement, cleaned; with `separators` a '%'
// after every sequence).  Returns false when the file cannot be read.
static bool text_from_fasta(const std::string &path, bool separators, std::vector<uint8_t> &text) {
    std::ifstream in(path);
    if (!in.good()) return false;
    std::string line, seq;
    bool have = false;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ')) line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (have) append_clean(text, seq, separators);
            seq.clear();
            have = true;
        } else if (have) seq += line;
    }
    if (have) append_clean(text, seq, separators);
    return true;
}

// `movi build` of the host CLI (movi_amd/host/build_cmd.cpp includes this file with MOVI_BUILD_INDEX_NO_MAIN):
// FASTA file -> OUT_DIR/index.movi of the given type.  Returns a message on failure, "" on success.
std::string movi_build_index_from_fasta(const std::string &fasta, int mode, const std::string &out_dir, bool separators) {
    std::vector<uint8_t> text;
    if (!text_from_fasta(fasta, separators, text)) return "cannot open " + fasta;
    if (text.empty()) return "no sequence in " + fasta;
    if ((uint64_t)text.size() + 1 >= (1ull << 31)) return "text too long for the in-memory constructor (2^31 characters, reverse complements included)";
    mkdir(out_dir.c_str(), 0777);
    const std::vector<uint8_t> img = build_index(text, mode);
    std::ofstream f(out_dir + "/index.movi", std::ios::binary);
    f.write(reinterpret_cast<const char *>(img.data()), (std::streamsize)img.size());
    f.close();
    if (!f.good()) return "cannot write " + out_dir + "/index.movi";
    return "";
}

#ifndef MOVI_BUILD_INDEX_NO_MAIN
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: see the header of tools/build_index.cpp\n"); return 1; }
    const std::string cmd = argv[1];
    std::vector<uint8_t> text;
    int mode = 6;
    std::string out_dir;
    uint64_t n_reads = 0, read_len = 0, seed = 1;
    double sub_rate = 0;
    if (cmd == "fasta" && argc >= 5) {
        mode = atoi(argv[3]);
        out_dir = argv[4];
        const bool separators = argc >= 6 && std::string(argv[5]) == "separators";   // movi build --separators
        if (!text_from_fasta(argv[2], separators, text)) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    } else if (cmd == "pangenome" && argc >= 8) {
        const uint64_t anc_len = strtoull(argv[2], nullptr, 10), n_genomes = strtoull(argv[3], nullptr, 10);
        const double snp = atof(argv[4]);
        seed = strtoull(argv[5], nullptr, 10);
        mode = atoi(argv[6]);
        out_dir = argv[7];
        if (argc >= 11) { n_reads = strtoull(argv[8], nullptr, 10); read_len = strtoull(argv[9], nullptr, 10); sub_rate = atof(argv[10]); }
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + 7;
        std::string anc(anc_len, 'A');
        for (auto &c : anc) c = "ACGT"[splitmix64(st) & 3];
        // a shallow star phylogeny: every genome = ancestor + its own SNPs, plus SNPs shared by a random half
        for (uint64_t g = 0; g < n_genomes; g++) {
            std::string s = anc;
            uint64_t sg = seed * 1315423911ull + g * 2654435761ull;
            const uint64_t k = (uint64_t)(snp * anc_len);
            for (uint64_t i = 0; i < k; i++) {
                const uint64_t pos = splitmix64(sg) % anc_len;
                s[pos] = "ACGT"[splitmix64(sg) & 3];
            }
            uint64_t sh = seed * 77 + 5;                                   // shared variants: same stream for all genomes
            for (uint64_t i = 0; i < k; i++) {
                const uint64_t pos = splitmix64(sh) % anc_len;
                const char alt = "ACGT"[splitmix64(sh) & 3];
                const uint64_t carriers = splitmix64(sh);
                if ((carriers >> (g & 63)) & 1) s[pos] = alt;
            }
            append_clean(text, s);
        }
    } else if (cmd == "reads" && argc >= 8) {
        // build_index reads <text.bin> <n_reads> <read_len> <sub_rate> <seed> <out_file>
        std::ifstream in(argv[2], std::ios::binary);
        if (!in.good()) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        in.seekg(0, std::ios::end);
        text.resize((size_t)in.tellg());
        in.seekg(0);
        in.read(reinterpret_cast<char *>(text.data()), (std::streamsize)text.size());
        n_reads = strtoull(argv[3], nullptr, 10); read_len = strtoull(argv[4], nullptr, 10); sub_rate = atof(argv[5]);
        seed = strtoull(argv[6], nullptr, 10);
        std::vector<uint8_t> reads(n_reads * read_len);
        draw_reads(text, reads, n_reads, read_len, sub_rate, seed * 31 + 99);
        write_file(argv[7], reads);
        return 0;
    } else {
        fprintf(stderr, "usage: see the header of tools/build_index.cpp\n");
        return 1;
    }
    if (mode != 2 && mode != 3 && mode != 5 && mode != 6 && mode != 7 && mode != 8) { fprintf(stderr, "mode must be 2, 3, 5, 6, 7 or 8\n"); return 1; }
    mkdir(out_dir.c_str(), 0777);
    if (cmd == "pangenome" && argc == 9 && std::string(argv[8]) == "text-only") {   // only <out_dir>/text.bin (seconds): lets `reads` draw from a cached index's text
        write_file(out_dir + "/text.bin", text);
        return 0;
    }
    if (cmd == "pangenome") write_file(out_dir + "/text.bin", text);       // lets `reads` draw more reads later
    for (int set = 0; set < 2; set++) {                                    // optional second read set: argv[11..13] -> reads2.bin
        if (set == 1) {
            if (cmd != "pangenome" || argc < 14) break;
            n_reads = strtoull(argv[11], nullptr, 10); read_len = strtoull(argv[12], nullptr, 10); sub_rate = atof(argv[13]);
        }
        if (!n_reads) continue;
        std::vector<uint8_t> reads(n_reads * read_len);
        draw_reads(text, reads, n_reads, read_len, sub_rate, seed * 31 + 99 + set);
        write_file(out_dir + (set ? "/reads2.bin" : "/reads.bin"), reads);
    }
    const std::vector<uint8_t> img = build_index(text, mode);
    write_file(out_dir + "/index.movi", img);
    return 0;
}
#endif  // MOVI_BUILD_INDEX_NO_MAIN
