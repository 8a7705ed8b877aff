#include "options.hpp"

#include <cstdlib>
#include <map>
#include <set>

namespace movi_host {

namespace {

struct Spec {
    const char *long_name;
    char short_name;     // 0 = none
    bool takes_value;
};

// Flags of the actions we serve (src/movi_parser.cpp:82-223).  Flags of the reference that
// select features outside this engine's scope are recognised so that the error is explicit.
const Spec kSpecs[] = {
    {"index", 'i', true},        {"read", 'r', true},       {"out-file", 'o', true},
    {"threads", 't', true},      {"strands", 's', true},    {"pml", 0, false},
    {"count", 0, false},         {"classify", 0, false},    {"filter", 0, false},
    {"invert", 'v', false},      {"stdout", 0, false},      {"no-output", 0, false},
    {"no-prefetch", 'n', false}, {"reverse", 0, false},     {"verbose", 0, false},
    {"bin-width", 0, true},      {"ignore-illegal-chars", 0, true},
    {"bpf", 0, true},            {"small-bpf", 0, false},   {"large-bpf", 0, false},
    {"no-header", 0, false},     {"help", 'h', false},      {"gpus", 0, true},
    {"device", 0, true},         {"type", 0, true},         {"debug", 'd', false},
    {"logs", 0, false},          {"mmap", 0, false},        {"gen-reads", 0, false},
    {"fasta", 'f', true},          {"separators", 0, false},  {"seg-len", 0, true},
    {"ahead-rows", 0, true},
    // recognised but unsupported query types / features
    {"zml", 0, false},           {"mem", 0, false},         {"rpml", 0, false},
    {"kmer", 0, false},          {"kmer-count", 0, false},  {"sa-entries", 0, false},
    {"multi-classify", 0, false}, {"ftab-k", 0, true},      {"multi-ftab", 0, false},
    {"k-length", 'k', true},     {"min-mem-length", 'l', true},
};

const Spec *find_long(const std::string &n) {
    for (const Spec &s : kSpecs)
        if (n == s.long_name) return &s;
    return nullptr;
}
const Spec *find_short(char c) {
    for (const Spec &s : kSpecs)
        if (s.short_name && s.short_name == c) return &s;
    return nullptr;
}

long to_int(const std::string &name, const std::string &v) {
    char *end = nullptr;
    long x = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || (end && *end)) throw UsageError("Argument '" + v + "' failed to parse for option '" + name + "'");
    return x;
}

}  // namespace

std::string usage() {
    return "movi (MI355X engine): movi query -i DIR -r FILE|- [-o PREFIX] [--pml|--zml|--count] [--classify] [--filter [-v]]\n"
           "                      [--stdout] [--no-output] [-s N] [-t N] [-n] [--reverse] [--bin-width N]\n"
           "                      [--ignore-illegal-chars 1] [--gpus N] [--device D] [--seg-len N] [--ahead-rows 0|1] [--verbose]\n"
           "       movi view --bpf FILE\n"
           "       movi null -i DIR [--gen-reads -f REF.fasta] [--pml|--zml]\n"
           "       movi build -i DIR -f REF.fasta [--type regular-thresholds|blocked-thresholds|sampled-thresholds|regular|blocked|sampled]\n"
           "                  [--separators]\n";
}

Options parse_args(int argc, char **argv) {
    Options o;
    std::map<std::string, std::vector<std::string>> seen;
    std::vector<std::string> positional;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string name = a.substr(2), val;
            bool has_val = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
            const Spec *s = find_long(name);
            if (!s) throw UsageError("Option '" + name + "' does not exist");
            if (s->takes_value && !has_val) {
                if (i + 1 >= argc) throw UsageError("Option '" + name + "' is missing an argument");
                val = argv[++i];
            }
            seen[s->long_name].push_back(val);
        } else if (a.size() >= 2 && a[0] == '-' && a != "-") {
            // short options: -t1, -t 1, grouped booleans -nv
            for (size_t p = 1; p < a.size(); p++) {
                const Spec *s = find_short(a[p]);
                if (!s) throw UsageError(std::string("Option '") + a[p] + "' does not exist");
                if (s->takes_value) {
                    std::string val = a.substr(p + 1);
                    if (val.empty()) {
                        if (i + 1 >= argc) throw UsageError(std::string("Option '") + s->long_name + "' is missing an argument");
                        val = argv[++i];
                    }
                    seen[s->long_name].push_back(val);
                    break;
                }
                seen[s->long_name].push_back("");
            }
        } else {
            positional.push_back(a);
        }
    }
    auto has = [&](const char *n) { return seen.count(n) > 0; };
    auto val = [&](const char *n) { return seen[n].back(); };
    if (has("help") || positional.empty()) { o.command = "help"; return o; }
    o.command = positional[0];
    o.verbose = has("verbose");
    o.logs = has("logs");
    o.no_header = has("no-header");
    if (o.command == "query") {
        // src/movi_parser.cpp:341, :431-434
        if (seen["index"].size() != 1 || seen["read"].size() != 1)
            throw UsageError("Please include one index directory and one read file.");
        o.index_dir = val("index");
        o.read_file = val("read");
        if (has("out-file")) o.out_file = val("out-file");
        for (const char *bad : {"mem", "rpml", "kmer", "kmer-count", "sa-entries", "multi-classify", "ftab-k",
                                "multi-ftab"})
            if (has(bad))
                throw UsageError(std::string("--") + bad + " is not supported by the MI355X engine (PML, ZML and count queries "
                                 "on regular-thresholds / blocked-thresholds indexes only)");
        // --mmap (src/movi_parser.cpp: "Use memory mapping to read the index") is accepted and implied: movi_index_load
        // always maps the file and uploads the rows straight from the page cache
                if (has("bin-width")) o.bin_width = (size_t)to_int("bin-width", val("bin-width"));
        // movi_parser.cpp:353-355 applies set_count, set_zml, set_pml in this order and each setter
        // clears the other query types (movi_options.hpp:108-110), so the last one applied wins
        // (the "only specify count or pml" check at :407-410 can never fire)
        if (has("count")) { o.count = true; o.pml = false; o.zml = false; }
        if (has("zml")) { o.zml = true; o.pml = false; o.count = false; }
        if (has("pml")) { o.pml = true; o.count = false; o.zml = false; }
        o.classify = has("classify");
        o.filter = has("filter");
        if (o.filter) o.classify = true;                          // set_filter(), movi_options.hpp:122-125
        o.invert = has("invert");
        o.reverse = has("reverse");
        if (has("ignore-illegal-chars")) {
            long v = to_int("ignore-illegal-chars", val("ignore-illegal-chars"));
            if (v != 1 && v != 2)
                throw UsageError("ignore-illegal-chars should be either 1 (set illegal chars to 'A') or 2 (set illegal chars to a random char).");
            if (v == 2)
                throw UsageError("--ignore-illegal-chars 2 (random substitution) is not reproducible and is not supported; use 1");
            o.ignore_illegal_chars = (int)v;
        }
        if (has("no-prefetch")) o.prefetch = false;
        if (has("strands")) o.strands = (size_t)to_int("strands", val("strands"));
        if (has("threads")) o.threads = (size_t)to_int("threads", val("threads"));
        if (o.strands == 0) o.strands = 1;
        o.write_stdout = has("stdout");
        o.no_output = has("no-output");
        if (has("gpus")) { o.gpus = (int)to_int("gpus", val("gpus")); o.gpus_given = true; }
        if (has("device")) o.device = (int)to_int("device", val("device"));
        if (has("seg-len")) o.seg_len = (long)to_int("seg-len", val("seg-len"));
        if (has("ahead-rows")) {
            o.ahead_rows = (int)to_int("ahead-rows", val("ahead-rows"));
            if (o.ahead_rows < 0 || o.ahead_rows > 1) throw UsageError("--ahead-rows must be 0 or 1");
        }
        if (o.gpus < 1) throw UsageError("--gpus must be >= 1");
        if (o.classify && o.count) throw UsageError("--classify needs PML or ZML queries");
    } else if (o.command == "plan") {
        // host-only helper (no GPU): prints how the reads are batched and in which order
        // their records will be emitted, one `batch<TAB>id<TAB>length` line per read.
        if (seen["read"].size() != 1) throw UsageError("Please include one read file.");
        o.read_file = val("read");
        if (has("no-prefetch")) o.prefetch = false;
        if (has("strands")) o.strands = (size_t)to_int("strands", val("strands"));
        if (o.strands == 0) o.strands = 1;
    } else if (o.command == "null") {
        // src/movi_parser.cpp:524-541
        if (seen["index"].size() != 1) throw UsageError("Please specify the index directory file.");
        o.index_dir = val("index");
        if (has("gen-reads")) {
            o.gen_reads = true;
            if (!has("fasta")) throw UsageError("Please specify the reference fasta file.");
            o.ref_file = val("fasta");
        }
        if (has("zml")) { o.zml = true; o.pml = false; }
        if (has("pml")) { o.pml = true; o.zml = false; }
        if (has("device")) o.device = (int)to_int("device", val("device"));
    } else if (o.command == "build") {
        // src/movi_parser.cpp:441-470: one index directory and one reference
        if (seen["index"].size() != 1) throw UsageError("Please specify the index directory file.");
        if (seen["fasta"].size() != 1) throw UsageError("Please specify the reference fasta file.");
        o.index_dir = val("index");
        o.ref_file = val("fasta");
        if (has("type")) o.index_type = val("type");
        o.separators = has("separators");
    } else if (o.command == "view") {
        if (seen["bpf"].size() != 1) throw UsageError("Please specify one mls file.");
        o.bpf_file = val("bpf");
        o.small_bpf = has("small-bpf");
        o.large_bpf = has("large-bpf");
    } else {
        throw UsageError("The '" + o.command + "' action is not part of the MI355X engine (query, view, null and build only); use "
                         "the reference movi for inspect / color / ftab.");
    }
    return o;
}

}  // namespace movi_host
