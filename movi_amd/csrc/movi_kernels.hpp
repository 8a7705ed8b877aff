// movi_kernels.hpp -- device-side index view shared by the kernels and the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace movi {

constexpr int kPrefixShift = 5;   // one BWT-position checkpoint every 32 rows (count path)

// What every kernel needs of the index, passed by value as a kernel argument
// (lives in the kernarg segment: scalar loads, no per-lane traffic).
struct DevIndex {
    const uint8_t *rows;          // r * row_bytes, packed exactly as in index.movi (mode 7: widened to 4 B per row)
    const uint32_t *id_blocks;    // mode 8: [alphabet][n_blocks]
    const uint8_t *code_of;       // 256 bytes: ASCII -> code 0..3 (1..4 with separators), 0xFF = illegal
    const uint64_t *row_start_ckpt; // BWT position of row 32*j (count path), r/32+1 entries
    uint64_t r;
    uint64_t end_bwt_idx;
    uint64_t n_blocks;
    uint64_t block_size;
    uint32_t block_shift;         // log2(block_size) when it is a power of two (always, in practice), else 0xFFFFFFFF
    uint32_t sigma;               // alphabet size: 4 (ACGT; fewer for reduced test alphabets) or 5 ('%' + ACGT)
    uint64_t end_thr[4];          // end_bwt_idx_thresholds, by DNA character
    uint64_t first_runs[6], first_offsets[6], last_runs[6], last_offsets[6];
    // `movi build --separators` indexes (MoveStructure::use_separator, src/move_structure.cpp:547-552): code 0 is
    // the separator '%', DNA codes are 1..4, and the rows OF the separator keep explicit thresholds in a side
    // table (separators_thresholds[separators_thresholds_map[idx]], src/move_structure_query.cpp:540-541) --
    // here sorted by row: sep_rows[k] <-> sep_vals[k] = 4 x u16 packed as (v0 | v1 << 16, v2 | v3 << 16)
    uint32_t sep;                 // 0 / 1
    uint32_t n_sep;
    const uint64_t *sep_rows;
    const uint2 *sep_vals;
    // sampled-thresholds (mode 7): tally_ids[character][checkpoint], widened from the file's 40-bit entries
    const uint64_t *tally;
    uint64_t tally_len;
    uint64_t tally_cp;            // rows between checkpoints (movi build --checkpoint, default 20)
    // 1: row indexes fit 32 bits (r < 2^32 - 1) -> the kernels' uint32_t instantiations; the "idx64" option clears it so
    // that tests can run the 64-bit instantiations (tables beyond 4 G rows) on small indexes
    uint32_t idx32;
    // Top-of-walk table ("kmer_k" option; 0 = none): every walk starts in the same state (row r-1), so its state after
    // the last K bases of the read depends on those K bases alone.  kmer[code of the K-mer] = that state (after the LF
    // towards base K, before its fast-forward), which of the K bases matched (their PMLs follow from that) and the
    // fast-forwards / scan rows the K steps took: one 16-byte lookup instead of K dependent row gathers.  The reference's
    // analogue is the ftab of its k-mer queries (src/move_structure_search.cpp:66-167, 203-259), there for intervals.
    // The same idea for the count query (the reference's own ftab, src/move_structure_search.cpp:66-167, 203-259): the
    // backward-search interval after the last K bases of a read is a function of those K bases.  ftab[code of the K-mer]:
    // x = run_start[31:0]; y = run_end[31:0]; z = run_start[35:32] | run_end[35:32] << 4 | offset_start << 8 | offset_end << 20
    // (12 bits each); w = fast-forwards (15 bits) | scan rows << 15 (16 bits) | valid << 31.  valid = 0: the interval of this K-mer
    // is empty before K bases are consumed (or a step throws): such reads take the ordinary search from their last base.
    uint32_t ftab_k;
    uint32_t pad2_;
    const uint4 *ftab;
    uint32_t stage_lds;           // set per launch by launch_pml: bytes of dynamic LDS per lane for read staging (0 = none; 336 at cap 7)
    uint32_t mask_phase;          // set per launch, reset-mask output (pml_kernel_flatp<..., RING = 2>): the low 5 bits of the position of the batch's first base in the caller's whole read set (0 for a whole call)
    uint32_t inwin;               // set per launch: 1 = a reposition whose target is one of the window's rows is resolved in the same iteration
    uint32_t kmer_k;
    const uint4 *kmer;            // 4^K entries: x = row[31:0]; y = row[35:32] | off << 4 (12 bits) | match mask << 16 (K bits) |
                                  // valid << 31; z = fast-forwards; w = scan rows.  valid = 0: one of the K steps hit one of the
                                  // reference's throws -- such reads take the ordinary walk and report it
    // Look-ahead rows ("ahead_rows" option; nullptr = none): a second copy of the table in which every row carries, in the
    // SAME 128-byte line, what the row its LF points to looks like -- so that a step whose next base matches there without a
    // fast-forward (the common case on real reads) is taken without fetching that row: two bases per gather.  The count query's
    // two interval ends use the same entries (count_kernel_v0<6, 1>).
    // Layout: line L = rows 8L .. 8L+7 (64 bytes) followed by their 8 look-ahead entries (64 bytes); one more line at
    // byte rows2_tail holds rows r-4 .. r-1 and their entries (the walk's last, pulled-back window).  Entry of row i, with
    // j = id(i), j2 = id(j): x = j2[31:0]; y = n(j) (11 bits) | offset(j) << 11 (11 bits) | c(j) << 22 (3 bits) |
    // j2[35:32] << 25 | valid << 31.  valid = 0 when j or j2 is not a row (corrupt table): such steps are taken one by one.
    const uint8_t *rows2;
    uint64_t rows2_tail;
    // 1: the count query walks on them too.  It gains where walks that follow the text mostly arrive at their LF target
    // without a fast-forward (pangenome BWTs: 0.83 of the positions, +20 %) and loses where they do not (uniformly random
    // run sequences: 0.51, -12 %: the copy's lines hold half as many rows for the interval shrink) -- decided from that
    // ratio, which the builder tallies, when the copy is built by itself; a caller who asks for the copy gets it for both.
    uint32_t rows2_count;
    // Round 5 -- REPOSITION HINTS in the look-ahead copy of a table with fewer than 2^32 - 1 rows (hints = 1).  There the four
    // high id bits of a row (y[31:28]) and of its entry (y[28:25]) are zero and two entry bits (y[30:29]) were free: ten spare
    // bits per row.  Nine of them hold, for each of the row's three threshold slots k (= alphamap_3[c(row)][a], the same slot
    // that holds the threshold bit of read base a): how many rows BEYOND THE EDGE OF THE ROW'S WINDOW, in the direction the
    // threshold bit gives (0 = down, 1 = up: get_thresholds is n or 0 and offset < n, src/move_structure.cpp:295-309), the
    // nearest run of base a lies -- 1 .. 7, or 0 for "inside the window, further than 7 rows, nowhere, or this is the '$' row
    // (whose direction depends on the offset)".  A mismatch whose scan (reposition_up / _down, src/move_structure_query.cpp:188-232)
    // would leave the window then gathers the TARGET's window in its next iteration instead of walking there window by window.
    // hint of slot k = (h >> 3 k) & 7 with h = row.y >> 28 | (entry.y >> 25 & 0x3F) << 4.
    // In this form ids in rows2 are 32 bits wide: x alone (a row whose id is not a row reads x = 0xFFFFFFFF), and every
    // reader of rows2 masks the hint bits off (tab_row / tab_entry / the walk kernel).
    uint32_t hints;
    // set per launch: width of a hint field the walk looks at -- 3 = use the hints (2 on the deep rows), 0 = ignore them ("repo_hints" 0: A/B)
    uint32_t hint_w;
    // Round 6 -- DEEP ROWS ("deep_rows" option; nullptr = none; tables of fewer than 2^28 - 1 rows): a copy of the table in which every row
    // carries what the walk reads at its LF target j = id(i) AND at j2 = id(j) -- up to three bases per gather -- packed to 21.33 bytes
    // per row (round 4's chain rows held the same in 32 bytes per row and fell out of the Infinity Cache: profiles/r04_chain_rows.txt).
    // Window q = rows 3q .. 3q + 2 = 64 bytes at byte 64 q (two windows per 128-byte line, six rows per line); 16 dwords: row s at
    // dwords 5s .. 5s + 4, its ten extra bits at [10s + 9 : 10s] of dword 15.  Row i, with j = id(i), j2 = id(j), j3 = id(j2):
    //   D0 = j (28 bits; 0x0FFFFFFF = not a row) | thr0 << 28 | thr1 << 29 | thr2 << 30
    //   D1 = n(i) | off(i) << 11 | c(i) << 22 | c(j) << 25 | c(j2) << 28          (c(j) = 7: no entry for j -- j or j2 is not a row; c(j2) = 7 likewise for j2 / j3)
    //   D2 = j2 (28 bits) | hints[3:0] << 28
    //   D3 = n(j) | off(j) << 11 | n(j2)[9:0] << 22
    //   D4 = j3 (28 bits) | n(j2)[10] << 28 | off(j2)[2:0] << 29
    //   X  = off(j2)[10:3] | hints[5:4] << 8
    // hints: two bits per threshold slot k (bits 2k + 1 : 2k): rows beyond the edge of the row's three-row window at which the nearest run
    // of the slot's base lies in the direction the threshold bit gives, 1 .. 3; 0 = inside the window, further, nowhere, or the '$' row.
    // The rows of the last window beyond r - 1 are padding (c = 7, n = 0, never reached).
    const uint8_t *rows3;
    // set per launch, reset-mask output (RING = 2) in blocks of one wavefront: non-null = every wavefront, when its 64 walks are over, turns
    // its reads' mask words into their u16 PML vectors here (through a tile in the LDS its reads were staged in: pml_kernel_flatp's tail)
    // -- the vector output at the mask walk's instruction count, the expansion hidden under the other wavefronts' gathers
    uint16_t *expand_out;
};

// Device counters of one query call.
struct DevStats {
    unsigned long long fast_forwards;
    unsigned long long scans;
    unsigned long long repositions;
    unsigned long long errors;
    // SIMT efficiency of the lane state machines: iterations in which a lane had work / 64 x wave iterations
    unsigned long long lane_steps;
    unsigned long long wave_steps;
    unsigned long long pad0_;
    unsigned long long segments;       // segment-parallel PML: segments walked, reads walked again
    unsigned long long rewalked;
    unsigned long long pad_;
};

constexpr int kCountCapWaves = 16;   // the count kernel's cap on cache-resident tables (launch_count)
constexpr int kCapWaves = 7;   // resident wavefronts per CU of the lane state machine on big batches (round 2: 9, optimum 8-10; round 3, with the reads staged in LDS and the top-of-walk table: 6-8, profiles/r03_occupancy_sweep.txt)
constexpr int kCapWavesAhead = 9;    // ... when the walk runs on the look-ahead rows: fewer lines per base, more walks in flight pay (profiles/r03_ahead_rows_ab.txt)
constexpr int kCapWavesDeep = 13;    // ... on the deep rows: fewer lines per base again and more instructions per iteration (c2, cap 9 / 11 / 13 / 14 / 16: vector out 71.8 / 75.7 / 78.4 / 78.7 / 77.9, reset masks out 87.3 / 89.9 / 89.4 / 89.5 / 87.8 Gbases/s: profiles/r06_deep_rows.txt)
constexpr uint32_t kOutRingBytes = 4096;          // pml_kernel_flatp<..., RING = 1>: the ring in the block's dynamic LDS its PMLs leave through (32 per lane)
constexpr uint64_t kOutRingReadLen = 1024;        // ... on by itself for batches whose mean read length is at least this (launch_pml)
constexpr uint32_t kTallySlots = 512;             // pairs of u64 counters a builder's tally is spread over (d_tally: 2 * kTallySlots u64, zeroed)
constexpr size_t kZmlStageBytes = 10240;          // zml_kernel_flat: dynamic LDS per one-wavefront block for its staged reads (160 bases per lane; 16 wavefronts per CU)
constexpr uint64_t kDeepReadLen = 1024;           // launch_pml: batches whose mean read length is below this walk on the deep rows (where the handle holds them)
constexpr uint64_t kPairLoadBytes = 2ull << 30;   // walked tables of this size and more: pair-shared gathers (launch_pml)

struct LaunchCfg {
    int block_threads = 0;   // 0 = auto: 64 for the PML and count kernels and the ZML state machine (finest dispatch grain), 256 for the base-synchronous ZML kernel
    // -1 auto; 0 first kernel (plain I/O; serves --logs), 1 base-synchronous packed I/O, 14 = the lane state machine over
    // row windows, software-pipelined (what auto picks).  (7 / 10 / 13 -- the row-at-a-time state machine, the hop-by-hop
    // advance, lane refill -- were A/B variants that never earned a default; removed in round 5.)
    int pml_variant = -1;
    int zml_variant = -1;  // -1 auto; 0 base-synchronous kernel, 1 lane state machine
    int count_variant = -1; // -1 auto (launch_count); 0 count_kernel_v0 (base-synchronous), 1 the lane state machine (zml_kernel_flat<..., CNT = 1>)
    int num_cus = 256;
    int waves_per_cu = 0;  // 0 = auto (the state machine on big batches: kCapWaves; else no cap); else cap resident waves per CU by padding the block's LDS allocation
    int seg_len = 2048;    // PML: batches whose mean read length is >= 2 x seg_len are walked segment-parallel (0 = never) ...
    int seg_probe = 1;     // ... if a probe of the batch finds that walks started mid-read fall into step quickly (0 = always: tests;
                           // 2 = no probe and no read-back at all, the caller's seg_verdict decides: the launch stays asynchronous)
    int seg_verdict = 0;   // seg_probe == 2: 1 = cut eligible batches, 0 = one lane per read
    int stage_reads = 1;   // every lane keeps the next stretch of its read in the block's LDS (rolling for long reads); 0 = off: A/B
    int inwin = 1;         // repositions inside the window resolved in the same iteration (0 = off: A/B)
    int out_ring = -1;     // PMLs out through a ring in LDS: -1 = batches of long reads (launch_pml), 0 / 1 = never / wherever it fits (A/B)
    int classify_fused = -1; // movi_pml_classify_*: -1 auto, 1 = vector + bins fused into the walk, 0 = the walk, then classify_kernel over the vectors
    int pair_loads = -1;   // the lanes of a pair fetch their row windows together (pml_kernel_flatp<..., PSH = 1>): -1 auto (tables of 2 GB and more), 0 never, 1 always
    int hints = 1;         // 1: mismatches whose scan leaves the row window jump by the reposition hints of the look-ahead rows (DevIndex::hints); 0 = off: A/B
    int zml_ahead = 0;     // 1: zml_kernel_flat<6, T, 0, 1> on the look-ahead rows where they exist (a third fewer iterations, no faster: opt-in)
    int fused_expand = 1;  // 1: a mask walk whose caller wants the vector expands its wavefronts' reads itself (DevIndex::expand_out); 0 = pml_expand_* kernels behind the walk: A/B
    int deep = -1;         // the PML walk on the deep rows (DevIndex::rows3) where the handle holds them: -1 = batches of short reads (mean length < kDeepReadLen), 0 never, 1 always
};

// What a launch_* call actually launched (movi_last_launch): the policy lives in the launchers, so they say what they picked.
struct LaunchInfo {
    char kernel[96] = {0};   // the dominant kernel's name as rocprofv3 prints it (template arguments included)
    int variant = -1;        // PML: 0, 1, 14; ZML: 0, 1; count: 0
    int block_threads = 0;
    int waves_per_cu = 0;    // resident-wavefront cap applied (0 = none)
    int segmented = 0;       // 1 = the segment-parallel plan ran (K1 + stitch + finalize around the named kernel)
    int idx64 = 0;           // 1 = the 64-bit row-index instantiation
    int ahead = 0;           // 1 = the walk ran on the look-ahead rows (two bases per gather where the next base matches)
    int staged = 0;          // > 0: every lane keeps the next `staged` bases of its read in LDS (pml_kernel_flatp<..., STG = 1>)
};

// Classifier::classify bins (src/classifier.cpp:99-143) fused into the PML kernels: per read the number of
// bins whose maximum is >= thr / < thr and the sum of the bin maxima.
struct ClsArgs {
    uint32_t bin_width = 0;                // 0 = no classification
    uint32_t thr = 0;
    uint32_t *above = nullptr;
    uint32_t *below = nullptr;
    uint64_t *sum_max = nullptr;
    // `movi query --logs` (src/read_processor.cpp:99-121,586-596, src/utils.cpp:268-289): per base, in emission order, the
    // fast-forwards of the LF that led to the NEXT base (entry k = LF into base k + 1; the last entry repeats the one before
    // it, as the reference's strand does when it writes the read out) and the rows the base's reposition scanned.
    // u16, truncated like MoveQuery's vectors.  Non-null: the first, base-synchronous kernel runs (pml_kernel<6, 0>).
    uint16_t *log_ff = nullptr;
    uint16_t *log_scan = nullptr;
};

// ---- segment-parallel long reads (PML) -------------------------------------------------------------------------
// A long read is cut into segments of about seg_len bases, every segment walked by its own lane from the state every
// read starts in (K1); then one lane per segment BOUNDARY continues the walk of the segment before across the
// boundary until it is in step -- same row, offset and match length at one of the checkpoints the speculative lane
// left every 32 bases -- with what that lane did (K2): from there on the two walks are the same walk.  A read all of
// whose boundaries fell into step is exact; any other is walked again from end to end (K3).  tests/studies/sync_study.py: on
// 10 kbp reads with 8 % / 1 % / 0.1 % substitutions a walk started mid-read is in step after a median of 11 / 80 / 607
// bases (maximum 107 / 665 / 4540).
struct SegCkpt { uint64_t idx; uint32_t off, ml, ff, scan, repo, pad_; };   // state after a base (before the LF to the next) + the segment's counters so far
struct SegFin { uint64_t idx; uint32_t off, ml; };                           // state after a segment's last base
struct SegTot { uint32_t ff, scan, repo, flag; };                            // a segment's counters; flag = kErr* of the speculative walk
struct SegArgs {
    const uint64_t *seg_in = nullptr;     // per segment: byte offset of its bases ...
    const uint64_t *seg_out = nullptr;    //   ... element offset of its first PML (emission order) ...
    const uint32_t *seg_len = nullptr;    //   ... and its length
    const uint64_t *n_seg = nullptr;      // device: number of segments
    SegCkpt *ckpt = nullptr;              // [(seg_out + k) >> 5] for every k of a segment with k % 32 == 31
    SegFin *fin = nullptr;
    SegTot *tot = nullptr;
    const uint8_t *read_fail = nullptr;   // K3: reads to walk again
    const uint32_t *go = nullptr;         // device: 1 = walk the segments; 0 = the probe advised against it: every read goes to K3
};

// The same for ZML (launch_zml_segmented): the state is the backward-search interval, its `open` flag and the match length.
struct ZSegCkpt { uint64_t rs, re; uint32_t os, oe, ml, open, ff, scan; };
struct ZSegFin { uint64_t rs, re; uint32_t os, oe, ml, open; };
struct ZSegArgs {
    const uint64_t *seg_in = nullptr;
    const uint64_t *seg_out = nullptr;
    const uint32_t *seg_len = nullptr;
    const uint64_t *n_seg = nullptr;
    ZSegCkpt *ckpt = nullptr;
    ZSegFin *fin = nullptr;
    SegTot *tot = nullptr;
    const uint8_t *read_fail = nullptr;
};

// One launch of the walk kernel (pml_kernel_flatp, movi_walk.hpp): its arguments plus the run-time choices that pick the
// instantiation.  The launchers (launch_pml, launch_pml_segmented) fill it; the movi_walk*_u32 / _u64 translation units turn
// it into a launch and name the kernel (every template argument) in `info`.
struct WalkLaunch {
    dim3 grid, block;
    size_t dyn_lds = 0;
    hipStream_t stream = nullptr;
    DevIndex ix;
    const uint8_t *bases = nullptr;
    const uint64_t *offs = nullptr;
    uint64_t n = 0;                       // reads (SEG 0 / 2) or an upper bound of the segments (SEG 1: the kernel reads the count)
    uint16_t *out = nullptr;
    uint8_t *err = nullptr;
    DevStats *stats = nullptr;
    const uint32_t *order = nullptr;
    ClsArgs cls;
    SegArgs seg;
    int cls_mode = 0;                     // 0 = PML vector, 1 = vector + classification bins, 2 = bins only
    int sep = 0, stg = 0, ahd = 0, psh = 0, ring = 0;   // separators index / reads staged through LDS / look-ahead rows / pair-shared gathers / PMLs out through the LDS ring (1) or as reset masks (2: `out` holds 32-bit words)
};
// Diagnostic: when switched on (movi_launch_log), every launch of the walk kernel notes its name -- the K1 / K3 launches of the
// segment plan included, which LaunchInfo (the dominant kernel only) does not show.
void note_walk_launch(const char *kernel_name);
size_t take_launch_log(char *buf, size_t cap);   // switches the log on; returns the bytes the names (one per line) take; clears it
hipError_t launch_walk_u32(const WalkLaunch &L, LaunchInfo *info);
hipError_t launch_walk_u64(const WalkLaunch &L, LaunchInfo *info);
hipError_t launch_walkseg_u32(int seg, const WalkLaunch &L, LaunchInfo *info);   // seg: 1 = segments (K1), 2 = re-walked reads (K3)
hipError_t launch_walkseg_u64(int seg, const WalkLaunch &L, LaunchInfo *info);
hipError_t preload_walk_u32();            // load the translation unit's code object now (movi_index_prepare)
hipError_t preload_walk_u64();
hipError_t preload_walkseg_u32();
hipError_t preload_walkseg_u64();

// Device workspace of the segmented path, owned by whoever owns the stream (the handle; a pipeline slot): grow-only.
struct SegWorkspace {
    void *buf = nullptr;
    size_t cap = 0;
};

// Reset masks instead of the PML vector (round 6; movi_pml_mask_device): PML[k] = reset(k) ? 0 : PML[k - 1] + 1, so one bit per base
// says everything the u16 vector does.  words: bit k % 32 of word ((offsets[i] + phase) >> 5) + i + k / 32 = 1 iff step k of read i
// reset match_len (its PML is 0); phase = the low five bits of the batch's first base's position in the caller's whole read set
// (sub-batches of one call then concatenate word for word).  The default walk writes them natively (pml_kernel_flatp<..., RING = 2>);
// every other path (segment-parallel long reads, the base-synchronous kernels) writes its vector to tmp_pml (n_bases u16, which the
// caller provides whenever pml_mask_needs_tmp says so) and pml_to_mask_kernel packs it.
struct MaskArgs {
    uint32_t *words = nullptr;
    uint32_t phase = 0;
    uint16_t *tmp_pml = nullptr;
    // non-null: the caller wants the u16 VECTOR here and `words` is scratch (movi_pml_device through masks): the mask walk expands its
    // wavefronts' reads itself where it can (one-wavefront blocks, reads in order, a 16-byte aligned vector), pml_expand_* kernels do
    // it behind the walk otherwise; a batch whose path has no mask output of its own simply writes the vector (no masks at all)
    uint16_t *expand_out = nullptr;
};
uint64_t pml_mask_words(uint64_t n_reads, uint64_t n_bases, uint32_t phase);          // words a batch's masks take (gap words included)
bool pml_mask_needs_tmp(const DevIndex &ix, const LaunchCfg &cfg, uint64_t n_reads, uint64_t n_bases, bool have_seg_ws);
bool pml_vector_via_masks(const DevIndex &ix, const LaunchCfg &cfg, uint64_t n_reads, uint64_t n_bases, bool have_seg_ws, bool ordered);   // movi_pml_device's policy: the vector of this batch through reset masks
// u16 vector <-> masks, streaming (one lane per read; a wavefront per read from a mean length of 2048 bases)
hipError_t launch_pml_expand(const uint32_t *d_words, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t phase,
                             uint16_t *d_out, hipStream_t stream);
// n 8-byte words from page-locked host memory (or anywhere the device can read) to the device by a kernel on `stream`: small blocks that
// must not queue behind bulk transfers on the copy engine (run_pipelined's per-chunk offsets)
hipError_t launch_copy_words(uint64_t *d_dst, const uint64_t *src, uint64_t n, hipStream_t stream);
hipError_t launch_pml_to_mask(const uint16_t *d_pml, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t phase,
                              uint32_t *d_words, hipStream_t stream);

// d_out == nullptr is allowed when cls.bin_width != 0: verdict bins only, no PML vector is written.
// seg_ws != nullptr allows the segment-parallel path for batches of long reads (cfg.seg_len); ragged_hint: 1 / 0 = the
// caller knows that the longest read is / is not more than 1.5 x the mean, -1 = it does not know (device offsets only).
hipError_t launch_pml(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                      const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream,
                      const ClsArgs &cls = ClsArgs(), SegWorkspace *seg_ws = nullptr, int ragged_hint = -1,
                      int *seg_verdict = nullptr, LaunchInfo *info = nullptr, const MaskArgs &mask = MaskArgs());

hipError_t launch_count(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                        uint64_t n_reads, uint64_t *d_matched, uint64_t *d_count, uint8_t *d_err,
                        DevStats *d_stats, const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream,
                        LaunchInfo *info = nullptr, uint64_t n_bases = 0);   // n_bases: the batch's size if the caller knows it (the state machine needs >= 16)

hipError_t launch_zml(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                      const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream,
                      SegWorkspace *seg_ws = nullptr, int ragged_hint = -1, int *seg_verdict = nullptr,
                      LaunchInfo *info = nullptr);

// d_err (optional): reads flagged there report no bins (0, 0, 0), like the fused kernels.
// n_bases (optional): the batch's size, if the caller knows it -- batches of long reads (mean >= 1024) take a wavefront per read.
hipError_t launch_classify(const uint16_t *d_pml, const uint64_t *d_offsets, uint64_t n_reads, uint32_t bin_width,
                           uint32_t thr, uint32_t *d_above, uint32_t *d_below, uint64_t *d_sum, hipStream_t stream,
                           const uint8_t *d_err = nullptr, uint64_t n_bases = 0);

// Mode 7: ix = a mode-7 view (widened rows + tally table); writes r mode-6 rows (8 bytes each) with the ids recovered by get_id.
hipError_t expand_sampled_rows(int mode, const DevIndex &ix, void *d_rows6, hipStream_t stream);   // mode 7 or 5
// Modes 8 / 2: ix = a blocked view (raw 6-byte rows + id_blocks); writes r 8-byte rows (regular-thresholds / regular
// layout) with the ids get_id reconstructs.
hipError_t expand_blocked_rows(int mode, const DevIndex &ix, void *d_rows6, hipStream_t stream);

// Mode 7: 3-byte file rows -> one dword per row (d_wide: r * 4 bytes + 16 of slack).
hipError_t widen_rows(const uint8_t *d_packed, uint64_t r, uint32_t *d_wide, hipStream_t stream);

// Fills the 4^K entries of the top-of-walk table (DevIndex::kmer) by walking every K-mer from the start state with the
// plain base-synchronous automaton.  ix.kmer / ix.kmer_k of `ix` are ignored; K in [1, 12]; thresholds types (kmode 6) only.
hipError_t build_kmer_table(const DevIndex &ix, uint32_t K, uint4 *d_table, hipStream_t stream);
// Look-ahead rows (DevIndex::rows2): ahead_rows_bytes(r) bytes at d_rows2, written by one kernel (a gather of id(id(i))
// per row); *tail = DevIndex::rows2_tail.  Thresholds types (kmode 6: the PML walk's rows), r >= 8.
uint64_t ahead_rows_bytes(uint64_t r);
// Deep rows (DevIndex::rows3): deep_rows_bytes(r) bytes at d_rows3 (zeroed by the call), one kernel (three dependent gathers per row).
// Thresholds types (kmode 6), 8 <= r < 2^28 - 1.
uint64_t deep_rows_bytes(uint64_t r);
bool deep_rows_eligible(uint64_t r);
hipError_t build_deep_rows(int kmode, const DevIndex &ix, uint8_t *d_rows3, hipStream_t stream);
bool ahead_rows_hinted(uint64_t r);    // the copy of a table of r rows carries reposition hints and 32-bit ids (DevIndex::hints)
hipError_t build_ahead_rows(int kmode, const DevIndex &ix, uint8_t *d_rows2, uint64_t *tail, hipStream_t stream,
                            unsigned long long *d_tally = nullptr);   // d_tally: 2 * kTallySlots zeroed u64 (tally_add spreads the atomics by block; the caller sums the pairs), optional
// sum of d_tally[2 s] / sum of d_tally[2 s + 1] over the kTallySlots pairs (2 * kTallySlots zeroed u64) = share of the BWT positions of every stride-th row that reach their LF
// target without a fast-forward (no_ff_share_kernel): the launch policy's statistic, without building the copy
hipError_t tally_no_ff_share(int kmode, const DevIndex &ix, uint64_t stride, unsigned long long *d_tally, hipStream_t stream);

// Fills the 4^K entries of the count query's interval table (DevIndex::ftab); mode = resident layout (6 or 3).
hipError_t build_ftab(int mode, const DevIndex &ix, uint32_t K, uint4 *d_table, hipStream_t stream);

// Fills ckpt[j] = BWT position of row (j << kPrefixShift), j = 0 .. ceil(r/32).
hipError_t build_row_start_ckpt(int mode, const uint8_t *d_rows, uint64_t r, uint64_t *d_ckpt,
                                hipStream_t stream);

}  // namespace movi
