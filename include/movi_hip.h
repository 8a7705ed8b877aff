/*
 * movi_hip.h -- C-ABI of libmovi_hip.so, the MI355X (gfx950) query engine.
 *
 * The reference (mohsenzakeri/Movi) has no FFI seam: its CLI driver calls C++
 * members directly.  This header declares, as plain C, exactly the calls a
 * Movi maintainer would bind in place of those members for the PML / count
 * query path (see INTEGRATION.md for the reference-side stub):
 *
 *   reference member (file:line under /root/reference)           replaced by
 *   ------------------------------------------------------------------------
 *   MoveStructure::deserialize      src/move_structure_io.cpp:471-511   movi_index_load / movi_index_create
 *   std::vector<MoveRow> rlbwt + side tables  include/move_structure.hpp:339-380   movi_index_desc_t
 *   ReadProcessor::process_latency_hiding (PML)  src/read_processor.cpp:641-730   movi_pml_host / movi_pml_device
 *   MoveStructure::query_pml        src/move_structure_query.cpp:234-474  movi_pml_host / movi_pml_device
 *   MoveStructure::query_backward_search  src/move_structure_search.cpp:340-352   movi_count_host / movi_count_device
 *   ReadProcessor::backward_search + compute_match_count  src/read_processor.cpp:610-620,1096-1175   movi_count_host / movi_count_device
 *   Classifier::classify (bins)     src/classifier.cpp:99-143           movi_classify_device / movi_pml_classify_device / movi_pml_classify_host
 *   MoveStructure::query_zml        src/move_structure_query.cpp:690-785  movi_zml_host / movi_zml_device
 *   MoveQuery::add_ml / matching_lens  include/move_query.hpp:26-38 (filled by process_char, src/read_processor.cpp:193-215)
 *                                                                       movi_pml_mask_device / movi_pml_mask_host (one reset bit per base)
 *                                                                       + movi_pml_expand_device / movi_pml_expand_host (bits -> u16 vector)
 *
 * Conventions: every entry point returns an int status (MOVI_OK == 0) and never
 * throws; movi_last_error() gives the message for the calling thread.  One
 * movi_index_t per GPU; calls on one handle are serialised by the caller;
 * distinct handles may be driven from distinct host threads / processes.
 * Pointers named d_* are device pointers on the handle's GPU, h_* host pointers.
 * No torch / C++ types appear in any signature.
 */
#ifndef MOVI_HIP_H
#define MOVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOVI_OK              0
#define MOVI_ERR_ARG        -1   /* bad argument                                        */
#define MOVI_ERR_FORMAT     -2   /* not a v2 index.movi of mode 5 / 6 / 7 / 8, or unsupported */
#define MOVI_ERR_IO         -3   /* file could not be read                               */
#define MOVI_ERR_HIP        -4   /* a HIP runtime call failed (message has the code)     */
#define MOVI_ERR_NO_DEVICE  -5   /* no usable gfx950 device: there is NO CPU fallback    */
#define MOVI_ERR_INVARIANT  -6   /* a walk hit one of the reference's "this should not
                                    happen" throws (move_structure.cpp:63-65,72-75;
                                    move_structure_query.cpp:582-598); per-read flags say which */

#define MOVI_MODE_REGULAR_THRESHOLDS 6   /* 8-byte rows, include/move_row.hpp:131-142 */
#define MOVI_MODE_BLOCKED_THRESHOLDS 8   /* 6-byte rows, include/move_row.hpp:128-142.  On upload get_id (blocked id + check
                                            point + first_runs, src/move_structure.cpp:91-102) is evaluated once per row and
                                            the rows rewritten in the mode-6 layout: one resident layout, one set of kernels */
#define MOVI_MODE_SAMPLED_THRESHOLDS 7   /* 3-byte rows without ids + sampled id table, move_row.hpp:122-127.  On upload
                                            the ids are recovered once, on the GPU, by MoveStructure::get_id
                                            (src/move_structure.cpp:104-283) and the rows rewritten in the mode-6
                                            layout: queries run the regular-thresholds kernels on identical rows */

#define MOVI_MODE_SAMPLED 5              /* like 7 without thresholds (10-bit lengths, move_row_configs.hpp:107-118).  Count and
                                            ZML queries only: PML on an index without thresholds repositions RANDOMLY in the
                                            reference (reposition_randomly) and is refused (MOVI_ERR_ARG) */

#define MOVI_MODE_REGULAR 3              /* 8-byte rows like 6 with 12-bit lengths and no thresholds (move_row_configs.hpp:21-32):
                                            resident as stored.  Count and ZML only, as for MOVI_MODE_SAMPLED */
#define MOVI_MODE_BLOCKED 2              /* 6-byte rows like 8 with a 24-bit blocked id and no thresholds (:54-75); expanded at
                                            upload to the MOVI_MODE_REGULAR layout.  Count and ZML only */

typedef struct movi_index movi_index_t;

/* The in-memory result of MoveStructure::deserialize for modes 6 / 8: the packed
 * row table (bytes exactly as stored in index.movi) plus the small side tables. */
typedef struct movi_index_desc {
    uint32_t mode;                    /* MOVI_MODE_*                                   */
    uint32_t alphabet_size;           /* <= 4 DNA symbols, or 5 = '%' + ACGT (movi build --separators) */
    uint64_t r;                       /* number of move rows                           */
    uint64_t length;                  /* BWT length n                                  */
    uint64_t end_bwt_idx;             /* row holding the terminator                    */
    uint64_t end_bwt_idx_thresholds[4];
    uint8_t  alphabet[8];             /* code -> ASCII                                 */
    uint8_t  code_of[256];            /* ASCII -> alphamap code (0..3; 1..4 with separators), 0xFF = illegal in a
                                         read (not in the alphabet, or the separator: check_alphabet, move_structure.cpp:383-397) */
    uint64_t first_runs[8], first_offsets[8], last_runs[8], last_offsets[8];  /* k = alphabet_size+1 used */
    uint64_t n_blocks;                /* mode 8: id_blocks is [alphabet_size][n_blocks] */
    uint64_t block_size;              /* mode 8                                        */
    const uint32_t *id_blocks;        /* mode 8, host pointer; NULL for mode 6          */
    /* Separators indexes only (alphabet "%ACGT"): the explicit thresholds of the separator's rows,
     * MoveStructure::separators_thresholds / separators_thresholds_map (include/move_structure.hpp:344-346)
     * exactly as read_separators_thresholds (src/move_structure_io.cpp:415-433) finds them in the file.
     * Host pointers (may be unaligned, into the parsed image); copied by movi_index_create*. */
    uint64_t n_separator_thresholds;
    const void *separator_thresholds;     /* n x ThresholdsRow { uint16_t values[4] }       */
    uint64_t n_separator_map;
    const void *separator_map;            /* n x { uint64_t row, uint64_t entry }           */
    /* Sampled-thresholds indexes only (mode 7): MoveStructure::tally_checkpoints / tally_ids
     * (include/move_structure.hpp:361-363) as read_tally_table (src/move_structure_io.cpp:338-349) finds them:
     * [alphabet_size][n_tally] 5-byte MoveTally entries (u32 low | u8 high, include/move_row.hpp:13-40). */
    uint32_t tally_checkpoints;
    uint32_t reserved_;
    uint64_t n_tally;
    const void *tally_ids;                /* host pointer, may be unaligned                 */
} movi_index_desc_t;

/* Per-query statistics (whole batch), for the algorithmic-bytes formula. */
typedef struct movi_query_stats {
    uint64_t bases;                   /* sum of read lengths                           */
    uint64_t fast_forwards;           /* rows stepped over in fast_forward             */
    uint64_t scans;                   /* rows stepped over in reposition scans / interval shrink */
    uint64_t repositions;             /* bases that took the mismatch branch           */
    uint64_t errors;                  /* reads whose error flag is set                 */
    uint64_t lane_steps;              /* lane state machines only: iterations in which a lane had work ...   */
    uint64_t wave_steps;              /* ... and iterations run by its wavefront: SIMT efficiency =
                                       * lane_steps / (64 * wave_steps); 0 / 0 from the other kernels        */
    uint64_t segments;                /* PML, segment-parallel path ("seg_len"): segments the reads were cut into (0: path not taken) */
    uint64_t rewalked;                /* ... and reads walked again from end to end because a boundary did not fall into step       */
} movi_query_stats_t;

const char *movi_last_error(void);
int movi_version(void);
int movi_device_count(int *count);

/* ---- index ------------------------------------------------------------------ */

/* Parse DIR/index.movi (fallback DIR/movi_index.bin, as
 * src/move_structure_io.cpp:16-32) or the file itself on the host and upload it. */
int movi_index_load(int device, const char *index_dir_or_file, movi_index_t **out);

/* Parse an index.movi image already in host memory (no upload): fills desc and
 * the offset/size of the row table inside the image.  Pure host code. */
int movi_index_parse(const void *h_image, size_t image_bytes, movi_index_desc_t *desc,
                     size_t *rows_offset, size_t *rows_bytes);

/* Upload host tables.  h_rows = r * row_bytes bytes as in the file. */
int movi_index_create(int device, const movi_index_desc_t *desc, const void *h_rows,
                      movi_index_t **out);

/* Adopt a row table that already lives on the device (e.g. received by an RCCL
 * broadcast into a caller-owned buffer).  The caller keeps ownership of d_rows
 * and must keep it alive until movi_index_destroy (modes 7 / 8: the rows are copied
 * -- expanded to the regular-thresholds layout, see MOVI_MODE_SAMPLED_THRESHOLDS --
 * so the buffer may be released after the call). */
int movi_index_create_from_device_rows(int device, const movi_index_desc_t *desc,
                                       const void *d_rows, movi_index_t **out);

/* One index on several GPUs of this node (`movi query --gpus N`; BASELINE north_star: "the shared index broadcast once
 * via RCCL over xGMI").  Replaces N runs of MoveStructure::deserialize (src/move_structure_io.cpp:471-511; the
 * reference has no multi-device form).  The file-format rows (8 / 6 / 3 bytes per row) cross PCIe ONCE, to
 * devices[0]; one RCCL broadcast -- single process: ncclCommInitAll, then ncclBroadcast per device inside
 * ncclGroupStart / ncclGroupEnd, root devices[0] -- carries them to the other GPUs over xGMI; every GPU then builds
 * its own resident layout (blocked / sampled types: expanded on that GPU).  out[i] is the handle on devices[i];
 * devices must be distinct.  n == 1 takes the same path (a communicator of one rank).  librccl.so.1 is bound when the
 * first of these calls is made (dlopen: 570 MB of code objects that the single-GPU paths never need); if it cannot be
 * loaded the call fails with MOVI_ERR_HIP -- there is no fallback to N uploads.
 * movi_index_load_replicated: the same from DIR/index.movi (mapped, as movi_index_load). */
int movi_index_replicate(const movi_index_desc_t *desc, const void *h_rows, const int *devices, int n,
                         movi_index_t **out);
int movi_index_load_replicated(const char *index_dir_or_file, const int *devices, int n, movi_index_t **out);

/* Build the handle's derived tables NOW instead of inside the first query: `what` = MOVI_PREPARE_PML (top-of-walk table,
 * 256 MB at K = 12; look-ahead rows, 16 bytes per row, where the device has room for them and half as much again, never more
 * than a quarter of the device by itself; round 6: the deep rows, 21.33 bytes per row, beside them for real-text tables of at most
 * 50 M rows -- "deep_rows") | MOVI_PREPARE_COUNT (row-start checkpoints, 8 bytes per 32 rows; interval table,
 * 256 MB; nothing else: since round 5 the count query's default is the lane state machine on the PLAIN rows -- the look-ahead rows
 * are built for it only under "count_variant" 0, the base-synchronous kernel of rounds 1 - 4, and then where a sample of the table
 * says the search will use them) | MOVI_PREPARE_ZML (nothing: accepted for symmetry).  Honours the options set before it ("kmer_k", "ftab_k", "ahead_rows": a table the caller built or switched
 * off is left alone).  Waits for the builders; *derived_bytes (optional) = bytes of device memory the handle's derived tables
 * hold afterwards (movi_index_info "derived_bytes").  After it the *_device entry points of those queries allocate nothing
 * and build nothing on this handle (batches of long reads still grow the segment workspace on their first call): they can be
 * captured into a HIP graph without a warm-up call -- a look-ahead copy that was declined for lack of device memory is asked for
 * again by the next movi_index_prepare call only, never from inside a query.  Not calling it is fine: the first query does the
 * same, lazily (and then retries a declined copy every 64 calls). */
#define MOVI_PREPARE_PML 1u
#define MOVI_PREPARE_COUNT 2u
#define MOVI_PREPARE_ZML 4u
int movi_index_prepare(movi_index_t *ix, uint32_t what, void *stream, uint64_t *derived_bytes);

int movi_index_destroy(movi_index_t *ix);
int movi_index_get_desc(const movi_index_t *ix, movi_index_desc_t *desc);   /* id_blocks = NULL */
/* Device pointer + size of the resident row table.  Mode 6: the file's bytes.  Modes 7 / 8: the expanded table, r x 8
 * bytes in the regular-thresholds layout (not what movi_index_create_from_device_rows takes for a mode-7 / 8
 * descriptor: broadcast the file bytes for that). */
int movi_index_device_rows(const movi_index_t *ix, const void **d_rows, size_t *bytes);

/* ---- PML -------------------------------------------------------------------- */

/* Reads are the concatenated ASCII bases; offsets has n_reads+1 entries (bytes).
 * out_pml[offsets[i] + k] = PML of base (len_i - 1 - k) of read i, i.e. emission
 * order (last base first), the order MoveQuery::matching_lens / the BPF record
 * hold (include/move_query.hpp:26-38, src/utils.cpp:202-246), u16-clamped.
 * d_read_err (optional, n_reads bytes): 0, or the code of the reference throw the walk
 * ran into (1 LF destination >= r, 2 >= 65535 fast-forwards, 3/4 no run below/above);
 * such a read reports all-zero PMLs.
 * d_read_order (optional, n_reads u32): a permutation of the reads; lane slot t works on read
 * d_read_order[t].  Results stay indexed by read.  A hook for callers with their own
 * scheduling; sorting by length measured no gain on MI355X, so the host entry
 * points pass NULL.  At most 2^32 reads per call, each shorter than 2^32 bases.
 * n_bases: the *_device entry points size device scratch from it, so it must be >= offsets[n_reads], and the batch
 * must start at offsets[0] == 0 (pass a sub-batch as its own d_bases / d_offsets / d_out_pml pointers with offsets
 * rebased to 0, not as a window into a larger offsets array).  A batch that breaks this is still answered correctly
 * -- the segment plan checks the offsets on the device and stands down -- but only by one lane per read.
 * Asynchronous on `stream` (a hipStream_t, NULL = the null stream) -- except that a batch of long reads that
 * qualifies for the segment-parallel walk ("seg_len" below) makes the call wait for a short probe of the batch
 * (under a millisecond, a 4-byte read-back) before it enqueues the walk.  Callers that capture the stream into a
 * graph or pipeline several streams set "seg_probe" = 2 and state the verdict themselves ("seg_verdict"): nothing
 * is read back then and the call never waits.  The FIRST PML (count) query on a handle also builds the handle's derived
 * tables -- top-of-walk ("kmer_k") / interval ("ftab_k") table, look-ahead rows ("ahead_rows"): allocations, a few ms of
 * kernels, a wait -- unless those options were set beforehand: make one warm-up call (or set the options) before capturing.
 * One query call per handle at a time: the counters behind movi_last_stats, the segment workspace and what
 * movi_last_launch reports belong to the handle, so two *_device calls on one handle must not be in flight together
 * even on different streams (the *_host entry points pipeline internally with per-chunk copies of all three). */
int movi_pml_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                    uint64_t n_reads, uint64_t n_bases, uint16_t *d_out_pml, uint8_t *d_read_err,
                    const uint32_t *d_read_order, void *stream);

/* Same, host buffers in and out; uploads, runs, downloads, synchronises.
 * stats may be NULL.  With h_bases and h_out_pml in PAGE-LOCKED memory (movi_host_alloc /
 * movi_host_register below) the call is overlapped: the reads are cut into chunks that travel through six
 * slots with their own streams -- up to three chunks going up or being walked while one comes down -- so that
 * upload, walks and download run at the same time (results are identical; DESIGN.md has the rates).  Pageable
 * buffers take the synchronous path -- unless the call is big (>= 2^27 bases and >= 3 x 2^15 reads): then it page-locks the
 * caller's buffers for its duration (hipHostRegister / hipHostUnregister: practically free for touched memory) and overlaps
 * all the same ("host_autopin" 0 turns that off).  h_out_pml == NULL: the walk runs and its error bytes and counters come back,
 * but no vector crosses PCIe (`movi query --no-output`: the reference computes and discards, src/movi.cpp:268-389). */
int movi_pml_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets,
                  uint64_t n_reads, uint16_t *h_out_pml, uint8_t *h_read_err,
                  movi_query_stats_t *stats);

/* `movi query --logs` (src/movi_parser.cpp:95; MoveQuery::add_fastforward / add_scan, include/move_query.hpp:51-53, collected
 * by ReadProcessor::process_char src/read_processor.cpp:99-121 and written by output_logs src/utils.cpp:268-289): the PML
 * query with two more u16 values per base, in emission order like the PMLs -- h_fastforwards[offsets[i] + k] = rows the
 * LF from base k to base k + 1 fast-forwarded over (the read's last entry repeats the one before it, as the reference's
 * strand does when it writes the read out; a read of one base: 0), h_scans[offsets[i] + k] = rows base k's reposition
 * scanned.  Runs the first, base-synchronous kernel: a diagnostic path, not a fast one.  (The third file of --logs, the
 * per-base wall-clock costs of a CPU strand, has no counterpart on a GPU lane; the CLI writes zeros.) */
int movi_pml_logs_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                       uint16_t *h_out_pml, uint16_t *h_fastforwards, uint16_t *h_scans, uint8_t *h_read_err,
                       movi_query_stats_t *stats);

/* Device-side counters of the last movi_pml_device / movi_count_device call on
 * this handle; synchronises `stream` first. */
int movi_last_stats(movi_index_t *ix, void *stream, movi_query_stats_t *stats);

/* What the last query call on this handle launched: which kernel the launch policy picked (the name as rocprofv3
 * prints it, template arguments included) and with which shape.  The policy lives in the library, so measurement code
 * asks instead of assuming (bench.py's roofline.kernel). */
typedef struct movi_launch_info {
    char kernel[96];
    int32_t variant;                  /* PML: 0, 1, 14 ("pml_variant"); ZML: 0, 1; count: 0, 1 ("count_variant")     */
    int32_t block_threads;
    int32_t waves_per_cu;             /* cap on resident wavefronts per CU that was applied (0 = none)             */
    int32_t segmented;                /* 1 = the segment-parallel plan ran around that kernel                       */
    int32_t idx64;                    /* 1 = the 64-bit row-index instantiation                                     */
    int32_t staged;                   /* > 0: every lane keeps the next `staged` bases of its read in LDS ("stage_reads";
                                         336 at the default occupancy cap, 256 on the look-ahead rows); 0: no staging */
    int32_t ahead;                    /* 1 = the walk ran on the look-ahead rows ("ahead_rows") */
    int32_t reserved_;
} movi_launch_info_t;
int movi_last_launch(const movi_index_t *ix, movi_launch_info_t *info);
/* Diagnostic (process-wide): the first call switches a log of the walk kernel's launches on; every call copies the DISTINCT
 * kernel names launched since the previous call into buf (one per line, NUL-terminated, truncated to cap) and clears the
 * log; *needed (optional) = bytes a complete copy takes.  Unlike movi_last_launch it sees the K1 / K3 launches of the
 * segment-parallel plan too: tests/test_kernel_coverage_gpu.py holds every instantiation the library was built with to the
 * oracle through it. */
int movi_launch_log(char *buf, size_t cap, size_t *needed);

/* What the handle holds in HBM besides the row table, and what its builders measured (no reference counterpart; the
 * derived tables of "kmer_k" / "ahead_rows" / "ftab_k" are built by the first query that can use them, so this is how a
 * caller sees their cost).  Keys: "rows_bytes" (the resident row table), "kmer_bytes", "ftab_bytes", "ahead_rows_bytes",
 * "ckpt_bytes" (0 = not built), "derived_bytes" (their sum), "ahead_no_ff" (share of the table's BWT
 * positions that reach their LF target without a fast-forward, tallied when the look-ahead rows are built: 0.83 on the
 * pangenome BWT, 0.51 on a uniformly random run sequence; -1 = not tallied yet).  Unknown key: MOVI_ERR_ARG. */
int movi_index_info(const movi_index_t *ix, const char *key, double *value);

/* ---- PML as reset masks (round 6) ----------------------------------------------- */

/* What the PML path PRODUCES is one bit per base.  process_char either increments match_len or zeroes it
 * (src/read_processor.cpp:193-215: match -> match_len + 1; mismatch + reposition or illegal character -> 0) and
 * MoveQuery::add_ml (include/move_query.hpp:26-38) records min(match_len, 65535): PML[k] = reset(k) ? 0 : PML[k - 1] + 1,
 * the run length since the last reset.  These entry points hand over the reset bits instead of the u16 vector -- 1/16 of
 * the bytes over PCIe and into host memory -- and the two expanders turn them back into exactly the vector movi_pml_* write
 * (u16 clamp included; MoveQuery::matching_lens is filled from them: INTEGRATION.md).
 *
 * Layout: 32-bit words.  Bit (k % 32) of word  W(i) + k / 32,  W(i) = ((first_base + offsets[i]) >> 5) - (first_base >> 5) + i,
 * belongs to emission step k of read i (k = 0: the read's LAST base, as in movi_pml_device); 1 = its PML is 0.  Every read
 * starts a word of its own and no prefix sum over the reads is needed (floor((o + l) / 32) + 1 >= floor(o / 32) + ceil(l / 32));
 * bits of a read's last word beyond its length are 0; words between two reads' ranges ("gap words") are unspecified.
 * first_base = position of this batch's first base in the caller's whole read set (0 for a stand-alone batch): consecutive
 * sub-batches of one read set then write word for word what one call over the whole set would -- sub-batch (first read f, first
 * base b) starts at word (b >> 5) + f of the whole.  movi_pml_mask_words: words the batch's array must hold.
 * A read that hit one of the reference's throws (d_read_err) reports every base as a reset, i.e. all-zero PMLs.
 * The default walk writes the words itself (one 4-byte store per 32 bases); batches it hands to another path -- long reads walked
 * segment-parallel, tables of fewer than 8 rows, "stage_reads" 0 -- write their vector to device scratch of the handle (2 bytes per
 * base, grow-only: those calls may allocate) and pack it.  Other arguments and conventions as movi_pml_device. */
int movi_pml_mask_words(uint64_t n_reads, uint64_t n_bases, uint64_t first_base, uint64_t *n_words);
int movi_pml_mask_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                         uint64_t n_reads, uint64_t n_bases, uint64_t first_base, uint32_t *d_mask_words,
                         uint8_t *d_read_err, const uint32_t *d_read_order, void *stream);
/* masks -> the u16 vector of movi_pml_device, on the device (streaming: 2 bytes written per base) */
int movi_pml_expand_device(movi_index_t *ix, const uint32_t *d_mask_words, const uint64_t *d_offsets,
                           uint64_t n_reads, uint64_t n_bases, uint64_t first_base, uint16_t *d_out_pml, void *stream);
/* Host buffers in, masks out: h_mask_words holds movi_pml_mask_words(n_reads, offsets[n_reads] - offsets[0], 0) words, laid out
 * with first_base = 0 relative to offsets[0].  Only 1/8 byte per base comes back over PCIe. */
int movi_pml_mask_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets,
                       uint64_t n_reads, uint32_t *h_mask_words, uint8_t *h_read_err, movi_query_stats_t *stats);
/* masks -> u16 vector on the HOST: pure CPU code (no device needed), n_threads worker threads (0 = three quarters of
 * the CPUs the process may use -- its affinity mask capped by the cgroup's CPU quota --, at most 24).  h_out_pml[offsets[i] + k] as movi_pml_host writes it; offsets need not start at 0 (the masks are laid out
 * relative to offsets[0], as movi_pml_mask_host writes them). */
int movi_pml_expand_host(const uint32_t *h_mask_words, const uint64_t *h_offsets, uint64_t n_reads,
                         uint16_t *h_out_pml, int n_threads);

/* ---- binary classification bins ------------------------------------------------ */

/* The per-read reduction of Classifier::classify (src/classifier.cpp:99-143) over PML vectors
 * that are already on the device: bins of bin_width PMLs in emission order, the last bin
 * absorbing a remainder shorter than bin_width; per read the number of bins whose maximum is
 * >= max_value_thr (bins_above) / below it, and the sum of the bin maxima.  The verdict is
 * FOUND iff bins_above / (bins_above + bins_below) > 0.5; the report's average is
 * sum_max / (bins_above + bins_below).  max_value_thr = max(percentile, 3) + 1 from
 * DIR/movi.pml.nulldb (src/classifier.cpp:30-33). */
int movi_classify_device(movi_index_t *ix, const uint16_t *d_pml, const uint64_t *d_offsets,
                         uint64_t n_reads, uint32_t bin_width, uint32_t max_value_thr,
                         uint32_t *d_bins_above, uint32_t *d_bins_below, uint64_t *d_sum_max,
                         void *stream);

/* PML query with the reduction above FUSED into the walk (each lane keeps its running bin maximum
 * in registers): one kernel, and with d_out_pml == NULL no PML vector is written at all -- what
 * `movi query --classify --filter / --no-output` needs.  d_out_pml != NULL writes the vectors too
 * (`--classify` with BPF output).  Other arguments as movi_pml_device / movi_classify_device. */
int movi_pml_classify_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                             uint64_t n_reads, uint64_t n_bases, uint32_t bin_width,
                             uint32_t max_value_thr, uint16_t *d_out_pml, uint32_t *d_bins_above,
                             uint32_t *d_bins_below, uint64_t *d_sum_max, uint8_t *d_read_err,
                             const uint32_t *d_read_order, void *stream);

/* Host buffers in, bins out (16 bytes back per read instead of 2 per base); uses the fused kernel
 * without a PML vector. */
int movi_pml_classify_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets,
                           uint64_t n_reads, uint32_t bin_width, uint32_t max_value_thr,
                           uint32_t *h_bins_above, uint32_t *h_bins_below, uint64_t *h_sum_max,
                           uint8_t *h_read_err, movi_query_stats_t *stats);

/* ---- count (backward search) -------------------------------------------------- */

/* Per read: matched = len - pos_on_r and count = rows in the last non-empty
 * interval, the two numbers src/utils.cpp:248-256 prints (`--no-prefetch`
 * semantics, src/move_structure_search.cpp:340-352). */
int movi_count_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint64_t *d_matched, uint64_t *d_count,
                      uint8_t *d_read_err, const uint32_t *d_read_order, void *stream);

int movi_count_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets,
                    uint64_t n_reads, uint64_t *h_matched, uint64_t *h_count, uint8_t *h_read_err,
                    movi_query_stats_t *stats);

/* ---- page-locked host memory --------------------------------------------------- */

/* The reference keeps reads and results in std::string / std::vector of its MoveQuery objects
 * (include/move_query.hpp:14-60); a caller that wants the *_host entry points to overlap their
 * transfers with the walk keeps them in page-locked memory instead: either allocated here
 * (hipHostMalloc) or its own buffers registered once (hipHostRegister) and reused across calls.
 * The *_host entry points detect it per call (movi_pml_host / movi_zml_host: h_bases and the
 * result vector; movi_count_host / movi_pml_classify_host: h_bases -- their per-read results are
 * small and may stay pageable). */
int movi_host_alloc(size_t bytes, void **out);
int movi_host_free(void *p);
int movi_host_register(void *p, size_t bytes);
int movi_host_unregister(void *p);

/* ---- tuning ------------------------------------------------------------------- */

/* Kernel variant / launch knobs, for A/B measurement (bench.py --variant).
 * Unknown keys return MOVI_ERR_ARG.  Keys: "pml_variant" (-1 = auto, 0 = first kernel (the one `--logs` runs on),
 * 1 = base-synchronous packed I/O, 14 = the lane state machine over row windows, software-pipelined: what auto
 * selects.  7 / 10 / 13 -- the row-at-a-time state machine, the hop-by-hop advance, lane refill -- were A/B variants that
 * never earned a default and were removed in round 5: DESIGN.md section 3),
 * "block_threads" (0 = auto, 64, 128, 192 or 256: the kernels' launch bound), "waves_per_cu"
 * (0 = the launch policy's cap, else at most this many wavefronts resident per CU), "idx64" (1 = run the kernel instantiations for
 * tables of 2^32 rows and more, whatever the size: a test hook), "release_scratch" (any value: frees the
 * device staging buffers that the *_host entry points keep, grow-only, across calls), "host_overlap" (0: a *_host call is never cut into overlapped pieces, whatever memory
 * its buffers are in -- one upload, the walk, one download; for a caller that pipelines chunk-sized calls itself and keeps its reads in page-locked
 * memory for the direct upload, as `movi query` does; default 1), "reserve_host_bases" / "reserve_host_results" /
 * "reserve_host_reads" (that staging reserved up front instead of inside the first big call: the reads of a synchronous *_host call of
 * up to this many bases, its u16 result vector, the per-read buffers of up to this many reads; movi_index_info "host_staging_bytes"
 * tells what is held), "pipe_chunk_bases"
 * (bases per chunk of the overlapped host path, 0 = its own policy: a test hook), "seg_len" (PML: batches whose mean
 * read length is at least twice this many bases are walked segment-parallel -- every read cut into segments of about
 * seg_len bases walked by their own lanes, stitched where the walks fall into step, reads that do not walked again:
 * identical results; ZML parses likewise; default 2048, 0 = off, else a multiple of 32; batches too small to fill the GPU
 * even so get shorter segments, down to 512), "seg_probe" (1, the default: an eligible batch
 * is cut into segments only if a probe of some of its reads finds that walks started mid-read fall into step within a
 * few hundred bases -- noisy long reads do, reads with 0.1 % errors and less do not and are better off with one lane
 * per read --; 0 = cut whatever the probe would say: a test hook; 2 = no probe, no length reduction, nothing read back:
 * "seg_verdict" (1 = cut, 0 = do not, the default) decides and the *_device calls stay asynchronous), "kmer_k"
 * (top-of-walk table: every walk starts in the same state, so its state after the last K bases of a read is a function
 * of those K bases -- one 16-byte table lookup replaces the first K row gathers of every read and segment; left alone the
 * first PML query on a DNA *-thresholds index builds the K = 12 table (256 MB, a few ms: that one call waits for it);
 * 0 = no table, 1..12 = build that one now), "stage_reads" (1, the default: every lane of the default walk copies the
 * next stretch of its read -- 256 bases at the occupancy cap of the look-ahead rows, 336 in small launches -- into the block's LDS and
 * takes its bases from there instead of re-fetching the read's cache line for every 16 bases; long reads roll through the
 * stretch; 0 = off: A/B),
 * "ahead_rows" (look-ahead rows: a second copy of the table, 16 bytes per row, in which each row's 128-byte line also holds
 * what the rows' LF targets look like -- character, length, offset, their own target --, so that a base that matches at the
 * target without a fast-forward is resolved, and its PML emitted, without fetching the target: two bases per gather on
 * real reads (+40 % on the cache-resident pangenome table); the count query walks both ends of its interval on them the
 * same way (+20 %; built by itself, the copy serves the count query only on tables whose positions mostly reach their LF
 * target without a fast-forward -- the builder tallies it --, which BWTs of real text do and random run sequences do
 * not).  Left alone, the first PML query builds them wherever the device has room for the copy and half as much again
 * (with the pair-shared gathers below they pay at every table size measured, 14 M to 1 B rows), the first count query where
 * a sample of the table says its search will use them.  1 = build now,
 * 0 = none (freed).  (Entries two rows deep -- "chain rows", three bases per gather -- were built in round 4, bit-exact, and
 * measured 10 - 38 % slower: profiles/r04_chain_rows.txt; removed),
 * "repo_hints" (1, the default: in the look-ahead copy of a table of fewer than 2^32 - 1 rows the rows' spare bits hold, per
 * threshold slot, how many rows beyond its window's edge the nearest run of that base lies -- a mismatch whose scan leaves the
 * window then gathers the scan's END in its next iteration instead of walking there window by window; 0 = ignore them: A/B),
 * "count_variant" (-1, the default: the count query runs as a lane state machine over row windows -- zml_kernel_flat<..., CNT = 1>,
 * by pairs of lanes on tables of 2 GB and more ("pair_loads") -- wherever it can: tables of 8 rows and more, batches of 16 bases
 * and more; 0 = count_kernel_v0, the base-synchronous kernel of rounds 1 - 4 (on the look-ahead rows where the table's statistic
 * admits them); 1 = the state machine or an error),
 * "inwin_repo" (1, the default: a reposition whose target run is one of the row window's other rows is resolved in
 * the iteration that sees the mismatch; 0 = off: A/B),
 * "pair_loads" (pair-shared gathers: the two lanes of a pair fetch each row window together, each lane one 16-byte half of
 * it in the same load instruction, halves exchanged through DPP -- one translation request and one 32-byte access where a
 * lane's own two loads are two of each; -1, the default: on for walked tables of 2 GB and more, where translation requests
 * bound the walk -- 1 B rows 32 -> 44 Gbases/s together with the look-ahead rows; the ZML parse fetches its two windows per
 * iteration the same way there: 16.8 -> 25.0 --; 1 / 0 = always / never: A/B),
 * "out_ring" (the PML kernels' PMLs leave through a ring in LDS -- one 2-byte LDS write per PML, a finished group of 16 as
 * two 16-byte stores -- instead of being packed in registers: a ninth fewer vector instructions per iteration; -1, the
 * default: batches whose mean read length is at least 1024, the shape that is bound by its own instruction stream (100 k x
 * 10 kbp: +5.7 %); 1 / 0 = wherever the block's LDS holds it / never: A/B),
 * "classify_fused" (movi_pml_classify_device with a PML vector: -1, the default: reads of mean length >= 1024 are walked
 * first and their bins reduced from the resident vectors by a wavefront per read -- faster than bins fused into the walk
 * there --, shorter ones fused; 1 = always fused, 0 = always two passes),
 * "zml_ahead" (1: the ZML parse walks on the look-ahead rows where they exist -- a third fewer iterations, no faster: off
 * by default),
 * "pml_via_mask" (round 6; PML as reset masks.  movi_pml_device: the walk writes one bit per base and every wavefront expands its reads'
 * words into the u16 vector itself when its walks are over -- -1, the default: batches of short reads (mean length < 1024: c2 78.4 ->
 * 86.7 Gbases/s, 1 B rows 42.3 -> 46.0); 1 = wherever the walk can write masks; 0 = the walk writes the vector itself (register packer /
 * LDS ring), "host_masks" (movi_pml_host: -1, the default: a call of >= 2^22 bases brings only the masks down and expands them into the
 * caller's vector on host worker threads beside the walks of the later chunks -- 1/16 of the bytes over PCIe, no page-locking of the
 * vector: 22.7 -> 33 - 34 Gbases/s on 1 M x 150 bp; 1 = every call; 2 = both ways down side by side, "host_mask_share" percent
 * (default 70) of the bases as masks, the rest as the vector itself by DMA into a page-locked vector; 0 = never masks: a caller whose
 * own threads are busy, like `movi query`),
 * "fused_expand" (1, the default; 0 = the expansion by kernels of their own behind the walk: A/B), "reserve_device_masks" (device scratch
 * for the mask words of movi_pml_device calls of up to this many bases, reserved now instead of inside the first such call),
 * "host_threads" (worker threads of the host-side expansion, 0 = three quarters of the CPUs the process may use -- affinity mask capped by the cgroup's quota --, at most 24),
 * "deep_rows" (round 6: a third layout of the table for the PML walk of short reads -- 21.33 bytes per row, windows of three rows, every
 * row with what the walk reads at its LF target AND at that row's target: up to three bases per gather.  Left alone, the first PML query
 * (or movi_index_prepare) builds them, beside the look-ahead rows, for tables of at most 50 M rows whose positions mostly reach their LF
 * target without a fast-forward (real text); 1 = build now (tables of fewer than 2^28 - 1 rows), 0 = none (freed).  "ahead_rows" 0 / 1
 * frees them too: it is a statement about what the walk runs on), "deep" (-1, the default: batches whose mean read length is below
 * 1024 walk on the deep rows where the handle holds them; 0 = never, 1 = always, the segment plan's launches included: A/B),
 * "ftab_k" (the count query's interval table -- the backward-search interval after the last K bases of a read by one lookup,
 * the reference's own ftab (src/move_structure_search.cpp:66-167) put to work for --count; left alone the first count query
 * on a DNA index builds the K = 12 table (256 MB); 0 = none, 1..12 = build that one now). */
int movi_set_option(movi_index_t *ix, const char *key, int64_t value);

/* ---- ZML (Ziv-Merhav cross parse) ---------------------------------------------- */

/* `movi query --zml`: MoveStructure::query_zml (src/move_structure_query.cpp:690-785) without
 * multi-classify.  Same buffers, layout, error and ordering conventions as movi_pml_device:
 * out_zml[offsets[i] + k] = match length recorded for base (len_i - 1 - k) of read i = the
 * number of bases of its greedy backward-search phrase to its right, u16-clamped; illegal
 * bases give 0 and end the phrase.  These are the values of the reference's --no-prefetch
 * path; its strand scheduler appends one extra entry to a read whose first base is illegal
 * (reset_backward_search skips to pos -1, src/read_processor.cpp:1006-1013, and the caller adds
 * again, :704), which this engine does not reproduce.  Batches of long reads are parsed segment-parallel like PML walks
 * ("seg_len"; the call then waits for a short probe of the batch before it enqueues the parse). */
int movi_zml_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                    uint64_t n_reads, uint64_t n_bases, uint16_t *d_out_zml, uint8_t *d_read_err,
                    const uint32_t *d_read_order, void *stream);

int movi_zml_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets,
                  uint64_t n_reads, uint16_t *h_out_zml, uint8_t *h_read_err,
                  movi_query_stats_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* MOVI_HIP_H */
