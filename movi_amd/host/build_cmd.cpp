// build_cmd.cpp -- `movi build`: FASTA -> DIR/index.movi, in memory, without the reference's external pipeline
// (prepare_ref + pfp-thresholds + movi-<type> build, src/movi_launcher.cpp:190-242).  The constructor is the one the tests
// hold to the reference's index-size known answers (tools/build_index.cpp: SA-IS, Kasai LCP, leftmost-minimum
// thresholds, rows, blocked / sampled ids), compiled into the host binary; texts up to 2^31 characters.
#define MOVI_BUILD_INDEX_NO_MAIN 1
#include "../../tools/build_index.cpp"

#include "options.hpp"

namespace movi_host {

int run_build(const Options &o) {
    static const struct { const char *name; int mode; } kTypes[] = {                   // src/movi_launcher.cpp:56-66
        {"regular-thresholds", 6}, {"blocked-thresholds", 8}, {"sampled-thresholds", 7},
        {"regular", 3}, {"blocked", 2}, {"sampled", 5},
    };
    int mode = -1;
    for (const auto &t : kTypes)
        if (o.index_type == t.name) mode = t.mode;
    if (mode < 0)
        throw UsageError("index type '" + o.index_type + "' is not supported (regular-thresholds, blocked-thresholds, "
                         "sampled-thresholds, regular, blocked, sampled)");
    const std::string err = movi_build_index_from_fasta(o.ref_file, mode, o.index_dir, o.separators);
    if (!err.empty()) throw std::runtime_error(err);
    std::cerr << "[movi] The " << o.index_type << " index is written to " << o.index_dir << "/index.movi\n";
    return 0;
}

}  // namespace movi_host
