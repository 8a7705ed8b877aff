// options.hpp -- `movi query` / `movi view` command line, mirroring the reference's
// flag table (src/movi_parser.cpp:82-223, handling :340-439) and defaults
// (include/movi_options.hpp:225-294) for the actions this engine implements.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace movi_host {

struct Options {
    std::string command;          // "query" | "view" | "null" | "plan"
    std::string index_dir;        // -i / --index
    std::string read_file;        // -r / --read ("-" = stdin)
    std::string out_file;         // -o / --out-file
    std::string bpf_file;         // view --bpf
    std::string ref_file;         // null --gen-reads -f/--fasta; build -f/--fasta
    std::string index_type = "regular-thresholds";   // build --type (DEFAULT_INDEX_TYPE, src/movi_launcher.cpp:17)
    bool separators = false;      // build --separators
    bool gen_reads = false;       // null --gen-reads
    bool pml = true;              // default query type (movi_options.hpp:243)
    bool count = false;
    bool zml = false;             // --zml: Ziv-Merhav cross parse lengths, same outputs as PML
    bool classify = false;
    bool filter = false;
    bool invert = false;
    bool write_stdout = false;    // --stdout
    bool no_output = false;
    bool prefetch = true;         // -n / --no-prefetch clears it (affects record order only)
    bool reverse = false;
    bool verbose = false;
    bool no_header = false;       // view: headerless BPF
    bool small_bpf = false, large_bpf = false;
    int ignore_illegal_chars = 0; // 1 = substitute 'A'; 2 (random) is rejected
    size_t strands = 16;
    size_t threads = 1;
    size_t bin_width = 150;
    int gpus = 1;                 // extension: --gpus N shards the reads across N devices
    bool logs = false;            // --logs: <prefix>.costs / .scans / .fastforwards next to the PML output (src/utils.cpp:376-382)
    bool gpus_given = false;      //   (given explicitly, N == 1 included: the index goes through the RCCL replication path)
    int device = 0;               // extension: --device D
    int ahead_rows = -1;          // extension: --ahead-rows 0|1: look-ahead rows off / built whatever the device's free memory says (default: the engine's policy)
    long seg_len = -1;            // extension: --seg-len N: segment length of the segment-parallel long-read walk (0 = off; default: the engine's)

    // derived predicates, same names as the reference (movi_options.hpp:57-58)
    bool write_output_allowed() const { return !no_output && !filter; }
    bool write_stdout_enabled() const { return write_stdout && !classify; }
    bool ml() const { return pml || zml; }                                // per-base matching lengths
    std::string query_type() const { return count ? "count" : (zml ? "zml" : "pml"); }   // src/utils.cpp:47-67
};

struct UsageError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Throws UsageError with the reference's message texts where they exist.
Options parse_args(int argc, char **argv);
std::string usage();
int run_build(const Options &o);  // build_cmd.cpp

}  // namespace movi_host
