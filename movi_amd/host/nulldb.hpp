// nulldb.hpp -- `movi null`: the empirical null database the classifier reads its threshold from.
//
//   parse_null_reads                      src/utils.cpp:427-475   (random 150-bp chunks of the reference, reversed)
//   EmpNullDatabase::generate_stats       src/emperical_null_database.cpp:16-45  (PML / ZML of every null read)
//   EmpNullDatabase::compute_stats        src/emperical_null_database.cpp:47-92
//   EmpNullDatabase::serialize            src/emperical_null_database.cpp:94-104
//
// The matching lengths come from the GPU (movi_pml_host / movi_zml_host); this file only draws the
// reads, reduces the statistics and writes DIR/movi.<pml|zml>.nulldb.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace movi_host {

// FASTA records the way kseq hands them out: the sequence lines concatenated, characters as in the file.
std::vector<std::string> read_fasta_sequences(const std::string &path);

// Writes `out_path` (">read_<k>\n<reversed chunk>\n") and returns the number of null reads.  `seed` seeds
// std::rand like the reference's srand(time(0)).
size_t generate_null_reads(const std::string &ref_fasta, const std::string &out_path, unsigned seed);

struct NullStats {
    uint64_t num_values = 0;
    double mean = 0;
    uint64_t percentile_value = 0;          // "largest common value": the largest value that occurs >= 5 times
};

// values: every matching length of every null read, in the order the reads were processed.
NullStats compute_null_stats(const std::vector<uint64_t> &values);
void write_null_db(const std::string &path, const NullStats &st, const std::vector<uint64_t> &values);

}  // namespace movi_host
