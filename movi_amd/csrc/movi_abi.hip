// movi_abi.hip -- the extern "C" boundary of libmovi_hip.so (see include/movi_hip.h).
// Host-side C++ only: index.movi parsing, device residency, launches, transfers.
// There is deliberately NO CPU compute path here: without a HIP device every
// query entry point fails with MOVI_ERR_NO_DEVICE / MOVI_ERR_HIP.
#include "../../include/movi_hip.h"
#include "movi_kernels.hpp"
#include "movi_expand_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>           // types and prototypes only: librccl.so.1 is bound at first use (movi_index_replicate)
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

using namespace movi;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e) + " (hipError " + std::to_string((int)e) + ")";
    // the failure is REPORTED here: the thread's sticky last-error is cleared so that a later launch on this thread (callers may
    // tolerate this failure -- e.g. a staging reservation that did not fit) does not read it back as its own (hipGetLastError)
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? MOVI_ERR_NO_DEVICE : MOVI_ERR_HIP;
}
#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);        \
    } while (0)

constexpr uint32_t kMoviMagic = 0x4D4F5649u;   // include/utils.hpp:29

bool mode_supported(uint32_t m) {
    return m == MOVI_MODE_REGULAR_THRESHOLDS || m == MOVI_MODE_BLOCKED_THRESHOLDS || m == MOVI_MODE_SAMPLED_THRESHOLDS ||
           m == MOVI_MODE_SAMPLED || m == MOVI_MODE_REGULAR || m == MOVI_MODE_BLOCKED;
}
bool mode_blocked(uint32_t m) { return m == MOVI_MODE_BLOCKED_THRESHOLDS || m == MOVI_MODE_BLOCKED; }
bool mode_sampled(uint32_t m) { return m == MOVI_MODE_SAMPLED_THRESHOLDS || m == MOVI_MODE_SAMPLED; }
// USE_THRESHOLDS, include/utils.hpp:145
bool mode_has_thresholds(uint32_t m) {
    return m == MOVI_MODE_REGULAR_THRESHOLDS || m == MOVI_MODE_BLOCKED_THRESHOLDS || m == MOVI_MODE_SAMPLED_THRESHOLDS;
}
size_t mode_row_bytes(uint32_t m) {             // MoveRow::row_size, include/move_row.hpp:104-120
    return (m == MOVI_MODE_REGULAR_THRESHOLDS || m == MOVI_MODE_REGULAR) ? 8 : (mode_blocked(m) ? 6 : 3);
}

struct Reader {
    const uint8_t *p;
    size_t n, pos = 0;
    // overflow-safe: `len` comes from file fields and may be anything
    bool get(void *dst, size_t len) {
        if (len > n - pos) return false;
        memcpy(dst, p + pos, len);
        pos += len;
        return true;
    }
    bool skip(size_t len) {
        if (len > n - pos) return false;
        pos += len;
        return true;
    }
};

}  // namespace

struct movi_index {
    int device = 0;
    movi_index_desc_t desc{};
    std::vector<uint32_t> id_blocks_host;
    std::vector<uint64_t> sep_rows_host;         // separators indexes: rows of the separator, ascending
    std::vector<uint64_t> sep_vals_host;         //   their ThresholdsRow (4 x u16 in one u64)
    std::vector<uint16_t> sep_thr_raw;           //   the file's two tables, kept for movi_index_get_desc-style round trips
    std::vector<uint64_t> sep_map_raw;
    uint64_t *d_sep_rows = nullptr;
    uint64_t *d_sep_vals = nullptr;
    std::vector<uint64_t> tally_host;            // mode 7: tally_ids widened to u64, [alphabet][n_tally]
    uint64_t *d_tally = nullptr;
    uint8_t *d_rows = nullptr;
    bool owns_rows = false;
    size_t rows_bytes = 0;
    uint8_t *d_code_of = nullptr;
    uint32_t *d_id_blocks = nullptr;
    uint64_t *d_ckpt = nullptr;      // built lazily by the first count query
    uint4 *d_kmer = nullptr;         // top-of-walk table ("kmer_k" option), 16 << 2K bytes
    int kmer_auto = 12;              // K of the table the first PML query builds by itself (0 = none: "kmer_k" 0)
    uint8_t *d_rows2 = nullptr;      // look-ahead rows ("ahead_rows" option), 16 bytes per row
    uint8_t *d_rows3 = nullptr;      // deep rows ("deep_rows" option), 21.33 bytes per row: what the PML walk runs on where the policy builds them
    int deep_auto = 1;               // 1: the first PML query builds them for tables of at most kDeepAutoRows rows (instead of the look-ahead rows)
    double ahead_no_ff = 0.0;        // share of the table's positions that arrive at their LF target without a fast-forward (build_ahead)
    bool ahead_tallied = false;
    bool count_declined_ahead = false;   // the count query's auto-build found the copy not worth keeping (rows2_count == 0): do not build it per call
    int ahead_auto = 1;              // 1: the first PML query builds them when the device has room for them (ahead_wanted)
    int ahead_retry_in = 0;          // the copy was declined for lack of device memory: calls until the device is asked again (not every call)
    bool prepared = false;           // movi_index_prepare has run: query calls build and allocate nothing any more (a declined copy is retried by movi_index_prepare only)
    uint4 *d_ftab = nullptr;         // the count query's interval table ("ftab_k" option), 16 << 2K bytes
    int ftab_auto = 12;              // K of the table the first count query builds by itself (0 = none)
    DevStats *d_stats = nullptr;
    DevIndex dev{};
    int kmode = 0;                   // row layout the kernels run on: desc.mode, except 6 for sampled-thresholds (expanded)
    LaunchCfg cfg;
    // device staging of the *_host entry points: grow-only, kept across calls (a hipMalloc / hipFree pair per call and
    // buffer cost more than the copies themselves); released by movi_index_destroy or movi_set_option("release_scratch")
    enum { kBases = 0, kOffs, kErr, kOut, kA, kB, kS, kMask, kTmp, kScratchSlots };   // (kMask: a chunk's reset-mask words; kTmp: the u16 vector of a mask call whose path has no mask output of its own)
    void *scratch[kScratchSlots] = {};
    size_t scratch_cap[kScratchSlots] = {};
    SegWorkspace seg_ws;             // segment-parallel long reads (launch_pml): device workspace of the handle's own calls
    uint64_t *h_rel = nullptr;       // synchronous path: a chunk's relative offsets, page-locked and kept (a fresh 2 MB vector per call was page
    size_t h_rel_cap = 0;            // faults + a staged copy: ~0.3 ms of a 2 ms call); entries
    // the overlapped form of the *_host entry points (page-locked caller buffers, movi_host_alloc): kPipeSlots chunks in
    // flight, each on its own stream with its own device staging, counters and a small page-locked block for what
    // travels with a chunk (relative offsets up; error bytes, per-read results and counters down)
    struct PipeSlot {
        hipStream_t s = nullptr;
        hipEvent_t ev = nullptr;                 // the slot's walk has finished
        hipEvent_t ev_up = nullptr;              // the slot's chunk has arrived
        void *d[kScratchSlots] = {};
        size_t cap[kScratchSlots] = {};
        DevStats *d_stats = nullptr;
        SegWorkspace seg_ws;                     // segment-parallel long reads: this slot's device workspace
        uint8_t *h = nullptr;
        size_t h_cap = 0;
    };
    enum { kPipeSlots = 6, kPipeAhead = 3 };   // slots; chunks going up or being walked while one comes down (2: -14 %, 4: the same)
    PipeSlot pipe[kPipeSlots];
    hipStream_t pipe_up = nullptr;   // every chunk's upload, in order (uploads on separate streams share the link and all arrive late)
    std::vector<hipEvent_t> pipe_ev; // chunk i of a call has arrived (calls whose reads go up ahead of the loop into scratch[kBases])
    uint64_t pipe_chunk_bases = 0;   // test hook ("pipe_chunk_bases"): chunk size of the overlapped path, 0 = its policy
    LaunchInfo last_launch;          // what the last query call launched (movi_last_launch)
    bool host_autopin = true;        // big *_host calls on pageable buffers page-lock them for the call ("host_autopin")
    bool host_overlap = true;        // page-locked buffers take the overlapped path ("host_overlap" 0: one upload, the walk, one download)
    bool seg_seen = false;           // the last PML / ZML host call on long reads was walked segment-parallel (chunk policy below)
    // the segment plan's probe verdict of the last batch of long reads, kept for the next calls on batches of the same shape (a stream of
    // batches of one kind of reads): the probe is a 0.4 ms read-back in front of a 5 ms walk (profiles/r06_few_long_reads.txt)
    int seg_cache_verdict = -1, seg_cache_left = 0;
    uint32_t seg_cache_key = 0;
    uint64_t reserved_result_bases = 0, reserved_reads = 0;   // "reserve_host_results" / "reserve_host_reads": what the mask words' scratch is reserved for
    int pml_via_mask = -1;           // "pml_via_mask": the walk's u16 vector through reset masks that its wavefronts expand themselves (-1: batches of short reads)
    int host_mask_share = 70;        // "host_mask_share": percent of a mixed call's bases that come down as masks (the rest as the vector itself)
    int host_masks = -1;             // "host_masks": movi_pml_host brings reset masks down and expands them on host worker threads (-1: calls of >= 2^22 bases into a pageable vector)
    int host_threads = 0;            // "host_threads": workers of the host-side expansion (0 = host_threads_default())
};

static void release_scratch(movi_index *ix);
namespace { hipError_t grow(void **p, size_t *cap, size_t bytes); }
// The upload stream of the overlapped host paths.  Streams of one priority share a handful of hardware queues, and a chunk's stream that
// landed on the upload stream's queue had its walk queued behind the copies' completion packets (every sixth chunk waited for uploads that
// were not its own: 0.1 ms of a dry upload stream each time, profiles/r06_host_path.txt).  A stream of another priority gets a queue of
// its own kind.
static hipError_t create_upload_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}   // grow-only device staging (defined with the host paths)

extern "C" {

const char *movi_last_error(void) { return g_err.c_str(); }
int movi_version(void) { return 100; }

int movi_device_count(int *count) {
    if (!count) return fail(MOVI_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail_hip(e, "hipGetDeviceCount"); }
    *count = n;
    return MOVI_OK;
}

// MoveStructure::deserialize, src/move_structure_io.cpp:471-511: v2 header
// (include/utils.hpp:32-61) | end_bwt_idx thresholds / next_down / next_up (3 x 4 x u64,
// :145-151) | alphamap (:153-157) | alphabet (:159-169) | u16 nt_splitting, bool constant
// (:171-172) | r rows (:361-397) | three overflow tables (:219-237) | counts and the
// base intervals (:269-287) | mode 8: id_blocks + block_size (:305-324) | with separators: their
// thresholds and the row -> entry map (:415-433).
int movi_index_parse(const void *h_image, size_t image_bytes, movi_index_desc_t *desc,
                     size_t *rows_offset, size_t *rows_bytes) {
    if (!h_image || !desc) return fail(MOVI_ERR_ARG, "NULL argument");
    Reader rd{static_cast<const uint8_t *>(h_image), image_bytes};
    uint8_t hdr[48];
    if (!rd.get(hdr, 48)) return fail(MOVI_ERR_FORMAT, "index image shorter than the 48-byte header");
    uint32_t magic;
    memcpy(&magic, hdr, 4);
    if (magic != kMoviMagic)
        return fail(MOVI_ERR_FORMAT, "invalid magic number in header: not a Movi 2.x index file");
    memset(desc, 0, sizeof(*desc));
    desc->mode = hdr[7];
    if (!mode_supported(desc->mode))
        return fail(MOVI_ERR_FORMAT, "index mode " + std::to_string(desc->mode) + " is not supported (only blocked=2, regular=3, "
                                         "sampled=5, regular-thresholds=6, sampled-thresholds=7 and blocked-thresholds=8; the "
                                         "Movi-1 style types large / constant / split are not)");
    memcpy(&desc->length, hdr + 16, 8);
    memcpy(&desc->r, hdr + 24, 8);
    memcpy(&desc->end_bwt_idx, hdr + 40, 8);
    // ids are 36 bits in every supported row layout (move_row_configs.hpp:34-51), so r < 2^36 -- which also keeps
    // every product of r below with a row size far from wrapping a u64
    if (desc->r == 0 || desc->r >= (1ull << 36) || desc->end_bwt_idx >= desc->r)
        return fail(MOVI_ERR_FORMAT, "corrupt header (r / end_bwt_idx)");
    if (!rd.get(desc->end_bwt_idx_thresholds, 32) || !rd.skip(64)) return fail(MOVI_ERR_FORMAT, "truncated index (basic data)");
    uint64_t amap_n = 0;
    if (!rd.get(&amap_n, 8) || amap_n != 256) return fail(MOVI_ERR_FORMAT, "unexpected alphamap size");
    uint64_t amap[256];
    if (!rd.get(amap, sizeof(amap))) return fail(MOVI_ERR_FORMAT, "truncated index (alphamap)");
    uint64_t asz = 0;
    if (!rd.get(&asz, 8) || asz == 0 || asz > 8) return fail(MOVI_ERR_FORMAT, "unexpected alphabet size");
    if (!rd.get(desc->alphabet, asz)) return fail(MOVI_ERR_FORMAT, "truncated index (alphabet)");
    // MoveStructure::use_separator, src/move_structure.cpp:547-552: five symbols led by '%'
    const bool sep = asz == 5 && desc->alphabet[0] == '%';
    if (asz > 4 && !sep)
        return fail(MOVI_ERR_FORMAT, "alphabet has " + std::to_string(asz) +
                                         " symbols: only DNA (<= 4) and separator ('%' + ACGT) indexes are supported");
    desc->alphabet_size = (uint32_t)asz;
    for (int c = 0; c < 256; c++) desc->code_of[c] = (c < 128 && amap[c] < asz) ? (uint8_t)amap[c] : 0xFF;
    if (sep) desc->code_of[(int)'%'] = 0xFF;               // check_alphabet, move_structure.cpp:384-388
    if (!rd.skip(3)) return fail(MOVI_ERR_FORMAT, "truncated index (flags)");
    const size_t row_b = mode_row_bytes(desc->mode);
    const size_t roff = rd.pos;
    if (desc->r > (image_bytes - rd.pos) / row_b || !rd.skip(desc->r * row_b))
        return fail(MOVI_ERR_FORMAT, "truncated index (move rows)");
    const bool sampled = mode_sampled(desc->mode);
    if (sampled) {                                         // read_tally_table, move_structure_io.cpp:338-349
        if (!rd.get(&desc->tally_checkpoints, 4) || desc->tally_checkpoints == 0)
            return fail(MOVI_ERR_FORMAT, "truncated index (tally checkpoints)");
        if (!rd.get(&desc->n_tally, 8)) return fail(MOVI_ERR_FORMAT, "truncated index (tally table)");
        desc->tally_ids = rd.p + rd.pos;
        if (desc->n_tally > image_bytes / (asz * 5) || !rd.skip(desc->n_tally * asz * 5))
            return fail(MOVI_ERR_FORMAT, "truncated index (tally table)");
        if (desc->n_tally < desc->r / desc->tally_checkpoints + 2)
            return fail(MOVI_ERR_FORMAT, "tally table does not cover the rows");
    }
    for (int t = 0; t < 3; t++) {
        uint64_t sz = 0;
        if (!rd.get(&sz, 8)) return fail(MOVI_ERR_FORMAT, "truncated index (overflow tables)");
        if (sz != 0) return fail(MOVI_ERR_FORMAT, "non-empty overflow table in a mode 6/8 index");
    }
    uint64_t csz = 0;
    if (!rd.get(&csz, 8) || csz > 8 || !rd.skip(csz * 8)) return fail(MOVI_ERR_FORMAT, "truncated index (counts)");
    uint64_t k = 0;
    if (!rd.get(&k, 8) || k != asz + 1) return fail(MOVI_ERR_FORMAT, "unexpected base-interval table size");
    if (!rd.get(desc->last_runs, k * 8) || !rd.get(desc->last_offsets, k * 8) ||
        !rd.get(desc->first_runs, k * 8) || !rd.get(desc->first_offsets, k * 8))
        return fail(MOVI_ERR_FORMAT, "truncated index (base intervals)");
    if (mode_blocked(desc->mode)) {
        if (!rd.get(&desc->n_blocks, 8)) return fail(MOVI_ERR_FORMAT, "truncated index (id blocks)");
        // the id_blocks payload stays in the image; callers locate it via desc->id_blocks
        desc->id_blocks = reinterpret_cast<const uint32_t *>(rd.p + rd.pos);
        if (desc->n_blocks > image_bytes / (asz * 4) || !rd.skip(desc->n_blocks * asz * 4))
            return fail(MOVI_ERR_FORMAT, "truncated index (id blocks)");
        desc->block_size = desc->mode == MOVI_MODE_BLOCKED ? 4194304 : 1048576;   // BLOCK_SIZE, move_row_configs.hpp:73 / :102
        uint64_t bs = 0;
        if (rd.get(&bs, 8)) desc->block_size = bs;         // move_structure_io.cpp:321-323
        // n_blocks * block_size >= r without the product (either factor is a file field)
        if (desc->block_size == 0 || desc->n_blocks == 0 || (desc->r - 1) / desc->block_size >= desc->n_blocks)
            return fail(MOVI_ERR_FORMAT, "id blocks do not cover the table");
    }
    if (sep && mode_has_thresholds(desc->mode)) {          // read_separators_thresholds, move_structure_io.cpp:415-433 (USE_THRESHOLDS)
        if (!rd.get(&desc->n_separator_thresholds, 8)) return fail(MOVI_ERR_FORMAT, "truncated index (separator thresholds)");
        desc->separator_thresholds = rd.p + rd.pos;
        if (desc->n_separator_thresholds > image_bytes / 8 || !rd.skip(desc->n_separator_thresholds * 8))
            return fail(MOVI_ERR_FORMAT, "truncated index (separator thresholds)");
        if (!rd.get(&desc->n_separator_map, 8)) return fail(MOVI_ERR_FORMAT, "truncated index (separator map)");
        desc->separator_map = rd.p + rd.pos;
        if (desc->n_separator_map > image_bytes / 16 || !rd.skip(desc->n_separator_map * 16))
            return fail(MOVI_ERR_FORMAT, "truncated index (separator map)");
    }
    if (rows_offset) *rows_offset = roff;
    if (rows_bytes) *rows_bytes = desc->r * row_b;
    return MOVI_OK;
}

static int finish_create(movi_index *ix) {
    const movi_index_desc_t &d = ix->desc;
    HIP_TRY(hipMalloc(&ix->d_code_of, 256));
    HIP_TRY(hipMemcpy(ix->d_code_of, d.code_of, 256, hipMemcpyHostToDevice));
    if (mode_blocked(d.mode)) {
        const size_t nb = (size_t)d.n_blocks * d.alphabet_size;
        if (nb == 0 || ix->id_blocks_host.size() != nb) return fail(MOVI_ERR_ARG, "the blocked modes need id_blocks");
        HIP_TRY(hipMalloc(&ix->d_id_blocks, nb * 4));
        HIP_TRY(hipMemcpy(ix->d_id_blocks, ix->id_blocks_host.data(), nb * 4, hipMemcpyHostToDevice));
    }
    const bool sep = d.alphabet_size == 5;
    if (sep && !ix->sep_rows_host.empty()) {
        const size_t ns = ix->sep_rows_host.size();
        HIP_TRY(hipMalloc(&ix->d_sep_rows, ns * 8));
        HIP_TRY(hipMalloc(&ix->d_sep_vals, ns * 8));
        HIP_TRY(hipMemcpy(ix->d_sep_rows, ix->sep_rows_host.data(), ns * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ix->d_sep_vals, ix->sep_vals_host.data(), ns * 8, hipMemcpyHostToDevice));
    }
    if (mode_sampled(d.mode)) {
        if (ix->tally_host.empty()) return fail(MOVI_ERR_ARG, "the sampled modes need tally_ids");
        HIP_TRY(hipMalloc(&ix->d_tally, ix->tally_host.size() * 8));
        HIP_TRY(hipMemcpy(ix->d_tally, ix->tally_host.data(), ix->tally_host.size() * 8, hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
    ix->cfg.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *pl = getenv("MOVI_PAIR_LOADS")) ix->cfg.pair_loads = atoi(pl) > 0 ? 1 : (atoi(pl) < 0 ? -1 : 0);   // A/B and test hook: every handle's default for "pair_loads"
    HIP_TRY(hipMalloc(&ix->d_stats, sizeof(DevStats)));
    HIP_TRY(hipMemset(ix->d_stats, 0, sizeof(DevStats)));
    DevIndex &v = ix->dev;
    v.rows = ix->d_rows;
    v.id_blocks = ix->d_id_blocks;
    v.code_of = ix->d_code_of;
    v.row_start_ckpt = nullptr;
    v.r = d.r;
    v.end_bwt_idx = d.end_bwt_idx;
    v.n_blocks = d.n_blocks;
    v.block_size = d.block_size ? d.block_size : 1;
    v.block_shift = 0xFFFFFFFFu;
    if ((v.block_size & (v.block_size - 1)) == 0) {
        v.block_shift = 0;
        while ((1ull << v.block_shift) < v.block_size) v.block_shift++;
    }
    v.sigma = d.alphabet_size;
    v.sep = sep ? 1u : 0u;
    v.n_sep = (uint32_t)ix->sep_rows_host.size();
    v.sep_rows = ix->d_sep_rows;
    v.sep_vals = reinterpret_cast<const uint2 *>(ix->d_sep_vals);
    v.tally = ix->d_tally;
    v.tally_len = d.n_tally;
    v.tally_cp = d.tally_checkpoints ? d.tally_checkpoints : 1;
    v.idx32 = d.r < 0xFFFFFFFFull ? 1u : 0u;
    for (int i = 0; i < 4; i++) v.end_thr[i] = d.end_bwt_idx_thresholds[i];
    for (int i = 0; i < 6; i++) {
        v.first_runs[i] = d.first_runs[i];
        v.first_offsets[i] = d.first_offsets[i];
        v.last_runs[i] = d.last_runs[i];
        v.last_offsets[i] = d.last_offsets[i];
    }
    // ---- resident row layout: ONE for every index type.  Blocked- and sampled-* tables are expanded to regular-thresholds
    // rows once, on the GPU, by the reference's get_id (needs the complete device view above: get_id reads first_runs /
    // id_blocks / tally).  Blocked: aligned 8-byte rows with the id inside instead of 2-byte-aligned 6-byte rows + a
    // check-point lookup per LF (count +7-10 %, ZML +28 %, PML +2-4 %, measured); sampled: see expand_sampled_kernel.
    // (the threshold-less `regular` / `blocked` types keep 12-bit lengths: their resident layout is the `regular` one, kmode 3)
    ix->kmode = (int)d.mode;
    if (d.mode != MOVI_MODE_REGULAR_THRESHOLDS && d.mode != MOVI_MODE_REGULAR) {
        const bool blocked = mode_blocked(d.mode);
        uint8_t *rows6 = nullptr;
        HIP_TRY(hipMalloc(&rows6, (size_t)d.r * 8 + 16));
        hipError_t e = hipMemset(rows6, 0, (size_t)d.r * 8 + 16);
        if (e == hipSuccess) e = blocked ? expand_blocked_rows((int)d.mode, v, rows6, nullptr) : expand_sampled_rows((int)d.mode, v, rows6, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { (void)hipFree(rows6); return fail_hip(e, "expanding the rows to the resident layout"); }
        if (ix->owns_rows && ix->d_rows) (void)hipFree(ix->d_rows);       // the file-format copy (an adopted buffer stays the caller's)
        if (ix->d_tally) (void)hipFree(ix->d_tally);                      // sampled: only get_id needed the checkpoints
        ix->d_tally = nullptr;
        ix->d_rows = rows6;
        ix->owns_rows = true;
        v.rows = rows6;
        v.tally = nullptr;
        ix->kmode = d.mode == MOVI_MODE_BLOCKED ? MOVI_MODE_REGULAR : MOVI_MODE_REGULAR_THRESHOLDS;
    }
    return MOVI_OK;
}

static int check_desc(const movi_index_desc_t *desc) {
    if (!desc) return fail(MOVI_ERR_ARG, "desc is NULL");
    if (!mode_supported(desc->mode)) return fail(MOVI_ERR_ARG, "unsupported mode");
    if (mode_sampled(desc->mode) &&
        (!desc->tally_ids || desc->tally_checkpoints == 0 || desc->n_tally < desc->r / desc->tally_checkpoints + 2))
        return fail(MOVI_ERR_ARG, "mode 7 needs tally_checkpoints / tally_ids covering the rows");
    if (desc->r == 0 || desc->end_bwt_idx >= desc->r) return fail(MOVI_ERR_ARG, "bad r / end_bwt_idx");
    if (desc->r >= (1ull << 36)) return fail(MOVI_ERR_ARG, "2^36 rows or more: ids are 36 bits in every row layout");
    if (mode_blocked(desc->mode) &&
        (desc->block_size == 0 || desc->n_blocks == 0 || (desc->r - 1) / desc->block_size >= desc->n_blocks ||
         desc->n_blocks > (1ull << 36)))
        return fail(MOVI_ERR_ARG, "id blocks do not cover the table");
    if (desc->alphabet_size == 0 || desc->alphabet_size > 5) return fail(MOVI_ERR_ARG, "alphabet_size must be 1..5");
    if (desc->alphabet_size == 5) {
        if (desc->alphabet[0] != '%') return fail(MOVI_ERR_ARG, "a 5-symbol alphabet must be '%' + ACGT (separators)");
        if ((desc->n_separator_thresholds && !desc->separator_thresholds) || (desc->n_separator_map && !desc->separator_map))
            return fail(MOVI_ERR_ARG, "separator tables missing");
        if (desc->n_separator_map > 0xFFFFFFFFull) return fail(MOVI_ERR_ARG, "too many separator rows");
    }
    return MOVI_OK;
}

static movi_index *new_handle(int device, const movi_index_desc_t *desc) {
    movi_index *ix = new movi_index();
    ix->device = device;
    ix->desc = *desc;
    if (mode_blocked(desc->mode) && desc->id_blocks)
        ix->id_blocks_host.assign(desc->id_blocks, desc->id_blocks + (size_t)desc->n_blocks * desc->alphabet_size);
    ix->desc.id_blocks = nullptr;
    if (desc->alphabet_size == 5) {
        // separators_thresholds[separators_thresholds_map[row]] flattened into two arrays sorted by row
        const size_t nt = (size_t)desc->n_separator_thresholds, nm = (size_t)desc->n_separator_map;
        ix->sep_thr_raw.resize(nt * 4);
        if (nt) memcpy(ix->sep_thr_raw.data(), desc->separator_thresholds, nt * 8);
        ix->sep_map_raw.resize(nm * 2);
        if (nm) memcpy(ix->sep_map_raw.data(), desc->separator_map, nm * 16);
        std::vector<std::pair<uint64_t, uint64_t>> kv(nm);
        for (size_t e = 0; e < nm; e++) kv[e] = {ix->sep_map_raw[2 * e], ix->sep_map_raw[2 * e + 1]};
        std::sort(kv.begin(), kv.end());
        for (const auto &e : kv) {
            uint64_t packed = 0;                          // a key pointing past the table reads as zeros
            if (e.second < nt) memcpy(&packed, ix->sep_thr_raw.data() + e.second * 4, 8);
            ix->sep_rows_host.push_back(e.first);
            ix->sep_vals_host.push_back(packed);
        }
    }
    ix->desc.separator_thresholds = nullptr;
    ix->desc.separator_map = nullptr;
    if (mode_sampled(desc->mode)) {
        const size_t ne = (size_t)desc->n_tally * desc->alphabet_size;
        const uint8_t *p = static_cast<const uint8_t *>(desc->tally_ids);
        ix->tally_host.resize(ne);
        for (size_t e = 0; e < ne; e++) {                 // MoveTally::get, include/move_row.hpp:28-38
            uint32_t right;
            memcpy(&right, p + e * 5, 4);
            ix->tally_host[e] = (uint64_t)right | ((uint64_t)p[e * 5 + 4] << 32);
        }
    }
    ix->desc.tally_ids = nullptr;
    ix->rows_bytes = (size_t)desc->r * mode_row_bytes(desc->mode);
    // MOVI_PML_VIA_MASK = 0 / 1: the handle's initial "pml_via_mask" (the test suite re-runs its PML parity files with 1: every PML
    // vector then comes from reset masks, expanded on the device or by the host's worker threads)
    if (const char *e = getenv("MOVI_PML_VIA_MASK")) {
        if (e[0] == '0' || e[0] == '1') { ix->pml_via_mask = e[0] - '0'; ix->host_masks = e[0] - '0'; }
    }
    return ix;
}

// Mode 7: the resident table is the file's 3-byte rows widened to one aligned dword each (kernels: load_row<7>);
// `packed` is a device buffer with the file bytes.  The handle owns the widened copy.
static hipError_t adopt_widened(movi_index *ix, const uint8_t *d_packed) {
    uint32_t *wide = nullptr;
    hipError_t e = hipMalloc(&wide, (size_t)ix->desc.r * 4 + 16);
    if (e != hipSuccess) return e;
    e = hipMemset(wide, 0, (size_t)ix->desc.r * 4 + 16);
    if (e == hipSuccess) e = widen_rows(d_packed, ix->desc.r, wide, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(wide); return e; }
    ix->d_rows = reinterpret_cast<uint8_t *>(wide);
    ix->owns_rows = true;
    return hipSuccess;
}

// Rows that sit in a fresh file mapping (movi_index_load): every page the copy reads is a minor fault away, and a
// pageable hipMemcpy takes those faults one by one on the calling thread.  So the rows go up in 64 MiB pieces while a
// few helper threads touch the pages of the next piece.
static hipError_t upload_from_mapping(uint8_t *d_rows, const uint8_t *h_rows, size_t bytes) {
    constexpr size_t kPiece = 64ull << 20;
    constexpr unsigned kTouchers = 4;
    auto touch = [&](size_t a, size_t b) {
        std::vector<std::thread> th;
        const size_t span = (b - a + kTouchers - 1) / kTouchers;
        for (unsigned t = 0; t < kTouchers; t++) {
            const size_t lo = a + t * span, hi = std::min(b, lo + span);
            if (lo >= hi) break;
            th.emplace_back([=]() {
                volatile uint8_t sink = 0;
                for (size_t o = lo; o < hi; o += 4096) sink = sink + h_rows[o];
                (void)sink;
            });
        }
        for (auto &x : th) x.join();
    };
    touch(0, std::min(bytes, kPiece));
    for (size_t off = 0; off < bytes; off += kPiece) {
        const size_t n = std::min(kPiece, bytes - off), next = off + kPiece;
        std::thread ahead;
        if (next < bytes) ahead = std::thread([&, next]() { touch(next, std::min(bytes, next + kPiece)); });
        hipError_t e = hipMemcpy(d_rows + off, h_rows + off, n, hipMemcpyHostToDevice);
        if (ahead.joinable()) ahead.join();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// DIR/index.movi (open_index_read, src/move_structure_io.cpp:16-41: then DIR/movi_index.bin; or the file itself) mapped,
// not read: the header and side tables are parsed in place and the row table -- all but a few kB of the file -- goes
// from the page cache to the GPU without a copy into a process buffer (the reference's --mmap; an 8 GB index no longer
// passes through an 8 GB std::vector first).
struct MappedIndex {
    void *p = MAP_FAILED;
    size_t n = 0;
    int open_index(const char *path) {
        std::string cand[3] = {std::string(path) + "/index.movi", std::string(path) + "/movi_index.bin", std::string(path)};
        int fd = -1;
        struct stat sb{};
        for (auto &c : cand) {
            fd = open(c.c_str(), O_RDONLY | O_CLOEXEC);
            if (fd < 0) continue;
            // size and type of the file that was OPENED (a stat() of the path before the open could describe another file)
            if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) break;
            close(fd);
            fd = -1;
        }
        if (fd < 0) return fail(MOVI_ERR_IO, std::string("Failed to open the index file at: ") + path);
        n = (size_t)sb.st_size;
        if (n == 0) { close(fd); return fail(MOVI_ERR_IO, std::string("Failed to read the index file at: ") + path); }
        p = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return fail(MOVI_ERR_IO, std::string("Failed to read the index file at: ") + path);
        (void)madvise(p, n, MADV_WILLNEED);
        return MOVI_OK;
    }
    ~MappedIndex() {
        if (p != MAP_FAILED) { const std::string keep = g_err; munmap(p, n); g_err = keep; }
    }
};

static int index_create(int device, const movi_index_desc_t *desc, const void *h_rows, movi_index_t **out, bool from_mapping) {
    if (!out || !h_rows) return fail(MOVI_ERR_ARG, "NULL argument");
    *out = nullptr;
    int rc = check_desc(desc);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    movi_index *ix = new_handle(device, desc);
    hipError_t e = hipMalloc(&ix->d_rows, ix->rows_bytes + 16);
    if (e == hipSuccess)
        e = from_mapping ? upload_from_mapping(ix->d_rows, static_cast<const uint8_t *>(h_rows), ix->rows_bytes)
                         : hipMemcpy(ix->d_rows, h_rows, ix->rows_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { movi_index_destroy(ix); return fail_hip(e, "uploading the move rows"); }
    ix->owns_rows = true;
    if (mode_sampled(desc->mode)) {
        uint8_t *packed = ix->d_rows;
        ix->d_rows = nullptr;
        e = adopt_widened(ix, packed);
        (void)hipFree(packed);
        if (e != hipSuccess) { movi_index_destroy(ix); return fail_hip(e, "widening the 3-byte rows"); }
    }
    rc = finish_create(ix);
    if (rc) { std::string keep = g_err; movi_index_destroy(ix); g_err = keep; return rc; }
    *out = ix;
    return MOVI_OK;
}

int movi_index_create(int device, const movi_index_desc_t *desc, const void *h_rows, movi_index_t **out) {
    return index_create(device, desc, h_rows, out, false);
}

int movi_index_create_from_device_rows(int device, const movi_index_desc_t *desc, const void *d_rows,
                                       movi_index_t **out) {
    if (!out || !d_rows) return fail(MOVI_ERR_ARG, "NULL argument");
    *out = nullptr;
    int rc = check_desc(desc);
    if (rc) return rc;
    if ((reinterpret_cast<uintptr_t>(d_rows) & 7u) != 0) return fail(MOVI_ERR_ARG, "d_rows must be 8-byte aligned");
    HIP_TRY(hipSetDevice(device));
    movi_index *ix = new_handle(device, desc);
    ix->d_rows = const_cast<uint8_t *>(static_cast<const uint8_t *>(d_rows));
    ix->owns_rows = false;
    if (mode_sampled(desc->mode)) {   // private copy; the caller's buffer is not kept
        ix->d_rows = nullptr;
        hipError_t e = adopt_widened(ix, static_cast<const uint8_t *>(d_rows));
        if (e != hipSuccess) { movi_index_destroy(ix); return fail_hip(e, "widening the 3-byte rows"); }
    }
    rc = finish_create(ix);
    if (rc) { std::string keep = g_err; movi_index_destroy(ix); g_err = keep; return rc; }
    *out = ix;
    return MOVI_OK;
}

int movi_index_load(int device, const char *path, movi_index_t **out) {
    if (!path || !out) return fail(MOVI_ERR_ARG, "NULL argument");
    *out = nullptr;
    MappedIndex m;
    int rc = m.open_index(path);
    if (rc) return rc;
    movi_index_desc_t desc;
    size_t roff = 0, rbytes = 0;
    rc = movi_index_parse(m.p, m.n, &desc, &roff, &rbytes);
    if (rc == MOVI_OK) rc = index_create(device, &desc, static_cast<const uint8_t *>(m.p) + roff, out, true);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------- one index on several GPUs: RCCL broadcast
namespace {

// librccl.so.1 is 570 MB of code objects for every collective, datatype and topology; a process that serves one GPU
// (every path but this one) never needs it, and mapping it costs more than a small query.  So it is bound here, when
// the first replicated index is made -- by name, with the image's own header for the types.  No RCCL, no replica.
struct Rccl {
    void *so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load() {
        if (so) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
        }
        if (!so) { const char *m = dlerror(); err = std::string("librccl.so.1 could not be loaded: ") + (m ? m : "?"); return false; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(so, n);
            if (!p && err.empty()) err = std::string("librccl.so.1 lacks ") + n;
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!err.empty()) { dlclose(so); so = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;

int replicate(const movi_index_desc_t *desc, const void *h_rows, const int *devices, int n, movi_index_t **out,
              bool from_mapping) {
    if (!out || !h_rows || !devices) return fail(MOVI_ERR_ARG, "NULL argument");
    if (n < 1 || n > 64) return fail(MOVI_ERR_ARG, "the number of devices must be in [1, 64]");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int rc = check_desc(desc);
    if (rc) return rc;
    int visible = 0;
    HIP_TRY(hipGetDeviceCount(&visible));
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= visible)
            return fail(MOVI_ERR_NO_DEVICE, "device " + std::to_string(devices[i]) + " requested but only " + std::to_string(visible) + " visible");
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return fail(MOVI_ERR_ARG, "the devices of a replicated index must be distinct");
    }
    // One GPU: nothing crosses xGMI, so a missing librccl is no reason to fail (`movi query --gpus 1`); when the library is
    // there the one-rank communicator still runs -- the same calls as N ranks, which is how 1-GPU boxes test this path.
    const bool have_rccl = g_rccl.load();
    if (!have_rccl && n > 1) return fail(MOVI_ERR_HIP, g_rccl.err);
    const size_t rows_bytes = (size_t)desc->r * mode_row_bytes(desc->mode);
    std::vector<uint8_t *> d_rows((size_t)n, nullptr);
    std::vector<hipStream_t> streams((size_t)n, nullptr);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    bool comms_up = false;
    auto cleanup = [&](bool keep_rows) {
        const std::string keep = g_err;
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(devices[i]);
            if (streams[i]) (void)hipStreamDestroy(streams[i]);
            if (comms_up && comms[i]) (void)g_rccl.CommDestroy(comms[i]);
            if (!keep_rows && d_rows[i]) (void)hipFree(d_rows[i]);
        }
        g_err = keep;
    };
    auto hip_fail = [&](hipError_t e, const char *what) { const int c = fail_hip(e, what); cleanup(false); return c; };
    auto nccl_fail = [&](ncclResult_t r, const char *what) {
        const int c = fail(MOVI_ERR_HIP, std::string(what) + ": " + g_rccl.GetErrorString(r));
        cleanup(false);
        return c;
    };
    // the table's only trip over PCIe: host (or file mapping) -> devices[0]
    for (int i = 0; i < n; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipMalloc(&d_rows[i], rows_bytes + 16);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) return hip_fail(e, "allocating the move rows");
    }
    {
        hipError_t e = hipSetDevice(devices[0]);
        if (e == hipSuccess)
            e = from_mapping ? upload_from_mapping(d_rows[0], static_cast<const uint8_t *>(h_rows), rows_bytes)
                             : hipMemcpy(d_rows[0], h_rows, rows_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "uploading the move rows");
    }
    // ... and from there to every other GPU: one broadcast over xGMI, all ranks driven by this process.
    // (librccl prints a version banner with printf when the first communicator comes up; the caller's stdout may be the
    // query's output -- `movi query --stdout` -- so stdout points at stderr while RCCL initialises)
    struct StdoutToStderr {
        int saved = -1;
        StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
        ~StdoutToStderr() { fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
    };
    if (have_rccl) {
        ncclResult_t r;
        {
            StdoutToStderr quiet;
            r = g_rccl.CommInitAll(comms.data(), n, devices);
        }
        if (r != ncclSuccess) return nccl_fail(r, "ncclCommInitAll");
        comms_up = true;
        r = g_rccl.GroupStart();
        if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            (void)hipSetDevice(devices[i]);
            r = g_rccl.Broadcast(d_rows[i], d_rows[i], rows_bytes, ncclUint8, 0, comms[i], streams[i]);
        }
        const ncclResult_t rg = g_rccl.GroupEnd();
        if (r != ncclSuccess) return nccl_fail(r, "ncclBroadcast");
        if (rg != ncclSuccess) return nccl_fail(rg, "ncclGroupEnd");
    }
    for (int i = 0; i < n; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamSynchronize(streams[i]);
        if (e != hipSuccess) return hip_fail(e, "the RCCL broadcast of the move rows");
    }
    // every GPU builds its own resident layout from the file-format rows it now holds
    for (int i = 0; i < n; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e != hipSuccess) { for (int j = 0; j < i; j++) { movi_index_destroy(out[j]); out[j] = nullptr; d_rows[j] = nullptr; } return hip_fail(e, "hipSetDevice"); }
        movi_index *ix = new_handle(devices[i], desc);
        ix->d_rows = d_rows[i];
        ix->owns_rows = true;
        d_rows[i] = nullptr;                                   // the handle's from here on
        int rci = MOVI_OK;
        if (mode_sampled(desc->mode)) {
            uint8_t *packed = ix->d_rows;
            ix->d_rows = nullptr;
            e = adopt_widened(ix, packed);
            (void)hipFree(packed);
            if (e != hipSuccess) rci = fail_hip(e, "widening the 3-byte rows");
        }
        if (rci == MOVI_OK) rci = finish_create(ix);
        if (rci != MOVI_OK) {
            const std::string keep = g_err;
            movi_index_destroy(ix);
            for (int j = 0; j < i; j++) { movi_index_destroy(out[j]); out[j] = nullptr; }
            g_err = keep;
            cleanup(false);
            return rci;
        }
        out[i] = ix;
    }
    cleanup(true);
    return MOVI_OK;
}

}  // namespace

extern "C" {

int movi_index_replicate(const movi_index_desc_t *desc, const void *h_rows, const int *devices, int n, movi_index_t **out) {
    return replicate(desc, h_rows, devices, n, out, false);
}

int movi_index_load_replicated(const char *path, const int *devices, int n, movi_index_t **out) {
    if (!path || !out || !devices) return fail(MOVI_ERR_ARG, "NULL argument");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    MappedIndex m;
    int rc = m.open_index(path);
    if (rc) return rc;
    movi_index_desc_t desc;
    size_t roff = 0, rbytes = 0;
    rc = movi_index_parse(m.p, m.n, &desc, &roff, &rbytes);
    if (rc == MOVI_OK) rc = replicate(&desc, static_cast<const uint8_t *>(m.p) + roff, devices, n, out, true);
    return rc;
}

int movi_index_destroy(movi_index_t *ix) {
    if (!ix) return MOVI_OK;
    (void)hipSetDevice(ix->device);
    if (ix->owns_rows && ix->d_rows) (void)hipFree(ix->d_rows);
    if (ix->d_code_of) (void)hipFree(ix->d_code_of);
    if (ix->d_id_blocks) (void)hipFree(ix->d_id_blocks);
    if (ix->d_sep_rows) (void)hipFree(ix->d_sep_rows);
    if (ix->d_sep_vals) (void)hipFree(ix->d_sep_vals);
    if (ix->d_tally) (void)hipFree(ix->d_tally);
    if (ix->d_ckpt) (void)hipFree(ix->d_ckpt);
    if (ix->d_kmer) (void)hipFree(ix->d_kmer);
    if (ix->d_ftab) (void)hipFree(ix->d_ftab);
    if (ix->d_rows2) (void)hipFree(ix->d_rows2);
    if (ix->d_rows3) (void)hipFree(ix->d_rows3);
    if (ix->d_stats) (void)hipFree(ix->d_stats);
    release_scratch(ix);
    delete ix;
    return MOVI_OK;
}

}  // extern "C"

// Top-of-walk table (DevIndex::kmer): 16 << 2K bytes, filled by one kernel; the call waits for it -- chunks of the
// overlapped host path walk on other streams.
static bool kmer_eligible(const movi_index *ix) {
    return ix->kmode == MOVI_MODE_REGULAR_THRESHOLDS && ix->dev.sigma - ix->dev.sep == 4 && ix->desc.r >= 8;
}
static int build_kmer(movi_index *ix, uint32_t K, hipStream_t s) {
    const size_t bytes = (size_t)16 << (2 * K);
    HIP_TRY(hipMalloc(&ix->d_kmer, bytes));
    hipError_t e = build_kmer_table(ix->dev, K, ix->d_kmer, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(ix->d_kmer); ix->d_kmer = nullptr; return fail_hip(e, "building the top-of-walk table"); }
    ix->dev.kmer = ix->d_kmer;
    ix->dev.kmer_k = K;
    return MOVI_OK;
}

// Look-ahead rows (DevIndex::rows2): a second copy of the table, 16 bytes per row, built by itself by the first PML query
// wherever the device has room for it.  History of the rule: round 3 built it up to 100 M rows only -- on a uniformly random
// table, the worst case for it (the base after the LF fast-forwards more often than not), it was +14 % at 20 M rows, +8 % at
// 100 M, -9 % at 200 M and -37 % at 1 B, where the two extra 16-byte loads per step cost more translation requests than the
// skipped rows save (profiles/r03_ahead_rows_threshold.txt); round 4 first let the table's own statistic (share of positions
// that arrive at their LF target without a fast-forward: 0.83 / 0.75 on real BWTs, 0.51 on random ones) extend it to 256 M
// rows of real text (113 M rows: 42.3 -> 53.5 Gbases/s, profiles/r04_real_100M.txt).  The pair-shared gathers then removed
// the translation cost the size rule was about (launch_pml: on for walked tables of 2 GB and more): on the look-ahead rows
// the random table now runs 45.4 against 38.5 Gbases/s at 200 M rows, 43.4 / 36.3 at 350 M, 44.4 / 34.8 at 700 M and
// 42.0 - 44.2 / 34.6 at 1 B (without the pairs: 21.4), the real 226 M-row BWT 50.8 / 38.6
// (profiles/r04_pair_shared_gathers.txt) -- so the copy pays at every size measured, and only memory decides.
// The statistic still steers the count query (kAheadCountRatio below) and is reported (movi_index_info "ahead_no_ff").
static bool ahead_eligible(const movi_index *ix) {
    return ix->kmode == MOVI_MODE_REGULAR_THRESHOLDS && ix->desc.r >= 8 && (ix->desc.r >> 36) == 0;
}
// A builder's tally: kTallySlots pairs of counters on the device (movi_kernels.hip: tally_add), added up here.
constexpr size_t kTallyBytes = 2u * kTallySlots * sizeof(unsigned long long);
static hipError_t read_tally(const unsigned long long *d_tally, unsigned long long (&sum)[2], hipStream_t s) {
    std::vector<unsigned long long> h(2u * kTallySlots);
    hipError_t e = hipMemcpyAsync(h.data(), d_tally, kTallyBytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    sum[0] = sum[1] = 0;
    if (e == hipSuccess) for (uint32_t i = 0; i < kTallySlots; ++i) { sum[0] += h[2 * i]; sum[1] += h[2 * i + 1]; }
    return e;
}
// The statistic over a sample of the table's rows (every 16th), before anything is built.
static void sample_no_ff(movi_index *ix, hipStream_t s) {
    if (ix->ahead_tallied) return;
    unsigned long long *d_tally = nullptr, h_tally[2] = {0, 0};
    hipError_t e = hipMalloc(&d_tally, kTallyBytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_tally, 0, kTallyBytes, s);
    if (e == hipSuccess) e = tally_no_ff_share(ix->kmode, ix->dev, 16, d_tally, s);
    if (e == hipSuccess) e = read_tally(d_tally, h_tally, s);
    if (d_tally) (void)hipFree(d_tally);
    if (e != hipSuccess) { (void)hipGetLastError(); return; }
    ix->ahead_no_ff = h_tally[1] ? (double)h_tally[0] / (double)h_tally[1] : 0.0;
    ix->ahead_tallied = true;
}
// Should the first query build the look-ahead rows by itself?  Yes, if they leave room for the query's own buffers.
static bool ahead_wanted(movi_index *ix) {
    const uint64_t bytes = ahead_rows_bytes(ix->desc.r);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    // (never more than a quarter of the device by itself: a handle's derived tables are the caller's HBM too -- "ahead_rows" 1
    // builds them whatever their size, movi_index_info reports them)
    return bytes <= total_b / 4 && free_b > bytes + std::max<uint64_t>(2ull << 30, bytes / 2);
}
// by_itself: built by the size policy, not on request -- then the count query uses the copy only where the table's own
// statistic says it pays (DevIndex::rows2_count)
constexpr double kAheadCountRatio = 0.67;
static int build_ahead(movi_index *ix, hipStream_t s, bool by_itself) {
    HIP_TRY(hipMalloc(&ix->d_rows2, ahead_rows_bytes(ix->desc.r)));
    uint64_t tail = 0;
    unsigned long long *d_tally = nullptr, h_tally[2] = {0, 0};
    hipError_t e = hipMalloc(&d_tally, kTallyBytes);
    if (e == hipSuccess) e = hipMemsetAsync(d_tally, 0, kTallyBytes, s);
    if (e == hipSuccess) e = build_ahead_rows(ix->kmode, ix->dev, ix->d_rows2, &tail, s, d_tally);
    if (e == hipSuccess) e = read_tally(d_tally, h_tally, s);
    if (d_tally) (void)hipFree(d_tally);
    if (e != hipSuccess) { (void)hipFree(ix->d_rows2); ix->d_rows2 = nullptr; return fail_hip(e, "building the look-ahead rows"); }
    ix->dev.rows2 = ix->d_rows2;
    ix->dev.rows2_tail = tail;
    ix->dev.hints = ahead_rows_hinted(ix->desc.r) ? 1u : 0u;   // (the copy's ids are 32 bits wide and its spare bits hold reposition hints)
    ix->ahead_no_ff = h_tally[1] ? (double)h_tally[0] / (double)h_tally[1] : 0.0;
    ix->ahead_tallied = true;
    ix->dev.rows2_count = (!by_itself || ix->ahead_no_ff >= kAheadCountRatio) ? 1u : 0u;
    return MOVI_OK;
}

// Deep rows (DevIndex::rows3; round 6): the PML walk's layout for tables of fewer than 2^28 - 1 rows -- 1.33 x the bytes of the look-ahead
// rows for three bases per gather instead of two.  Built by itself up to kDeepAutoRows rows (the copy then stays within reach of the
// Infinity Cache and the per-CU TLBs: profiles/r06_deep_rows.txt has the sizes measured), instead of the look-ahead rows, which the PML
// walk no longer needs then; "deep_rows" 1 / 0 builds / frees them at any eligible size.
constexpr uint64_t kDeepAutoRows = 48ull << 20;             // 50 M rows: a copy of 1 GiB
static bool deep_eligible(const movi_index *ix) {
    return ix->kmode == MOVI_MODE_REGULAR_THRESHOLDS && deep_rows_eligible(ix->desc.r);
}
static int build_deep(movi_index *ix, hipStream_t s) {
    HIP_TRY(hipMalloc(&ix->d_rows3, deep_rows_bytes(ix->desc.r)));
    hipError_t e = build_deep_rows(ix->kmode, ix->dev, ix->d_rows3, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(ix->d_rows3); ix->d_rows3 = nullptr; return fail_hip(e, "building the deep rows"); }
    ix->dev.rows3 = ix->d_rows3;
    return MOVI_OK;
}

// The count query's interval table (DevIndex::ftab): any DNA index, thresholds or not.
static bool ftab_eligible(const movi_index *ix) {
    return (ix->kmode == MOVI_MODE_REGULAR_THRESHOLDS || ix->kmode == MOVI_MODE_REGULAR) && ix->dev.sigma - ix->dev.sep == 4;
}
static int build_ftab_table(movi_index *ix, uint32_t K, hipStream_t s) {
    const size_t bytes = (size_t)16 << (2 * K);
    HIP_TRY(hipMalloc(&ix->d_ftab, bytes));
    hipError_t e = build_ftab(ix->kmode, ix->dev, K, ix->d_ftab, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(ix->d_ftab); ix->d_ftab = nullptr; return fail_hip(e, "building the count query's interval table"); }
    ix->dev.ftab = ix->d_ftab;
    ix->dev.ftab_k = K;
    return MOVI_OK;
}

// The derived tables of the PML walk -- top-of-walk table (256 MB at K = 12, a few ms), look-ahead rows (16 B per row) --:
// built by movi_index_prepare, or by the first PML query on the handle.  Nothing here fails the query: a table there is no room
// for (or a device in trouble, which the walk's own launch will report) is done without.
static void ensure_pml_tables(movi_index *ix, hipStream_t s, bool from_prepare = false) {
    if (ix->kmer_auto > 0 && !ix->d_kmer && kmer_eligible(ix)) {
        if (build_kmer(ix, (uint32_t)ix->kmer_auto, s) != MOVI_OK) {
            (void)hipGetLastError();
            ix->kmer_auto = 0;
        }
    }
    // (after movi_index_prepare the header's promise holds: query calls allocate nothing and build nothing -- a copy that was declined
    // for lack of memory is asked for again by the next explicit movi_index_prepare, never from inside a query, which may be under
    // stream capture or in the middle of a streaming run)
    if (ix->ahead_auto > 0 && !ix->d_rows2 && ahead_eligible(ix) && (from_prepare || !ix->prepared)) {
        // (declined for lack of memory: the device is asked again after 64 calls, not in every one -- hipMemGetInfo per chunk of
        // a streaming run -- and two processes sharing a GPU that both pass the check and then collide switch the auto-build off)
        if (ix->ahead_retry_in > 0) ix->ahead_retry_in -= 1;
        else if (!ahead_wanted(ix)) ix->ahead_retry_in = 63;
        else if (build_ahead(ix, s, true) != MOVI_OK) {
            (void)hipGetLastError();
            ix->ahead_auto = 0;
        }
    }
    // The deep rows, beside the look-ahead rows (batches of short reads walk on the one, long reads on the other: launch_pml): tables of up
    // to kDeepAutoRows rows whose positions mostly reach their LF target without a fast-forward -- real text: the builder of the
    // look-ahead rows has tallied it, else a sample does -- three bases per gather need two such arrivals in a row (uniformly random
    // run sequences, 0.51: 62.4 -> 61.2 Gbases/s; the pangenome BWT, 0.83: 80.3 -> 89.6 with reset masks out)
    if (ix->deep_auto > 0 && !ix->d_rows3 && deep_eligible(ix) && ix->desc.r <= kDeepAutoRows && (from_prepare || !ix->prepared)) {
        if (!ix->ahead_tallied) sample_no_ff(ix, s);
        size_t free_b = 0, total_b = 0;
        const uint64_t bytes = deep_rows_bytes(ix->desc.r);
        if (ix->ahead_tallied && ix->ahead_no_ff < kAheadCountRatio) {
            ix->deep_auto = 0;                                 // (the table's own statistic declines, for good)
        } else if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (2ull << 30) || build_deep(ix, s) != MOVI_OK) {
            (void)hipGetLastError();
            ix->deep_auto = 0;                                 // (no room, or a device in trouble: the walk stays on the look-ahead rows)
        }
    }
}

extern "C" {

int movi_index_get_desc(const movi_index_t *ix, movi_index_desc_t *desc) {
    if (!ix || !desc) return fail(MOVI_ERR_ARG, "NULL argument");
    *desc = ix->desc;
    desc->id_blocks = nullptr;
    desc->separator_thresholds = nullptr;
    desc->separator_map = nullptr;
    desc->tally_ids = nullptr;
    return MOVI_OK;
}

int movi_index_device_rows(const movi_index_t *ix, const void **d_rows, size_t *bytes) {
    if (!ix || !d_rows || !bytes) return fail(MOVI_ERR_ARG, "NULL argument");
    *d_rows = ix->d_rows;
    // blocked- / sampled-thresholds: the resident table is the EXPANDED one (r rows of the 8-byte regular-thresholds layout)
    *bytes = ix->kmode != (int)ix->desc.mode ? (size_t)ix->desc.r * 8 : ix->rows_bytes;
    return MOVI_OK;
}

int movi_set_option(movi_index_t *ix, const char *key, int64_t value) {
    if (!ix || !key) return fail(MOVI_ERR_ARG, "NULL argument");
    if (!strcmp(key, "pml_variant")) {
        if (value != -1 && value != 0 && value != 1 && value != 14)
            return fail(MOVI_ERR_ARG, "pml_variant must be -1 (auto), 0, 1 or 14");
        ix->cfg.pml_variant = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "zml_variant")) {
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "zml_variant must be -1 (auto), 0 or 1");
        ix->cfg.zml_variant = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "count_variant")) {                     // A/B: -1 = the launch policy, 0 = count_kernel_v0, 1 = the lane state machine
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "count_variant must be -1 (auto), 0 or 1");
        ix->cfg.count_variant = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "idx64")) {                             // test hook: run the 64-bit-index kernel instantiations
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "idx64 must be 0 or 1");
        ix->dev.idx32 = (value == 0 && ix->desc.r < 0xFFFFFFFFull) ? 1u : 0u;
        return MOVI_OK;
    }
    if (!strcmp(key, "block_threads")) {
        // every query kernel is compiled with __launch_bounds__(256)
        if (value != 0 && value != 64 && value != 128 && value != 192 && value != 256)
            return fail(MOVI_ERR_ARG, "block_threads must be 0 (auto), 64, 128, 192 or 256");
        ix->cfg.block_threads = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "classify_fused")) {                    // -1 auto, 1: vector + bins in one kernel, 0: walk, then a streaming pass over the vectors
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "classify_fused must be -1, 0 or 1");
        ix->cfg.classify_fused = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "pair_loads")) {                        // the lanes of a pair fetch their row windows together (A/B)
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "pair_loads must be -1 (auto), 0 or 1");
        ix->cfg.pair_loads = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "zml_ahead")) {                         // 1: the ZML state machine walks on the look-ahead rows where they exist (A/B: measured no faster)
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "zml_ahead must be 0 or 1");
        ix->cfg.zml_ahead = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "out_ring")) {                          // A/B: the PML kernels' output ring in LDS (-1 = the launch policy)
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "out_ring must be -1, 0 or 1");
        ix->cfg.out_ring = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "repo_hints")) {                        // A/B: mismatches whose scan leaves the row window jump by the look-ahead rows' reposition hints
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "repo_hints must be 0 or 1");
        ix->cfg.hints = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "inwin_repo")) {                        // A/B: repositions inside the row window resolved in the same iteration
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "inwin_repo must be 0 or 1");
        ix->cfg.inwin = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "release_scratch")) {                   // give back the device staging the *_host entry points keep
        (void)hipSetDevice(ix->device);
        release_scratch(ix);
        return MOVI_OK;
    }
    // device staging of the synchronous *_host calls reserved up front (it is grow-only and otherwise grows inside the first big call:
    // three hipMallocs, ~1.2 ms of a 3 ms call of 2^25 bases): the reads of a call of up to `value` bases / its u16 result vector /
    // the per-read buffers of up to `value` reads
    if (!strcmp(key, "reserve_host_bases") || !strcmp(key, "reserve_host_results") || !strcmp(key, "reserve_host_reads")) {
        if (value < 0 || (uint64_t)value > (1ull << 40)) return fail(MOVI_ERR_ARG, std::string(key) + " out of range");
        HIP_TRY(hipSetDevice(ix->device));
        const size_t v = (size_t)value;
        if (!strcmp(key, "reserve_host_bases")) {
            HIP_TRY(grow(&ix->scratch[movi_index::kBases], &ix->scratch_cap[movi_index::kBases], v));
        } else if (!strcmp(key, "reserve_host_results")) {
            HIP_TRY(grow(&ix->scratch[movi_index::kOut], &ix->scratch_cap[movi_index::kOut], v * 2));
            // (round 6: a PML vector is written through reset masks on the device -- their words, for as many reads as "reserve_host_reads" says)
            ix->reserved_result_bases = std::max<uint64_t>(ix->reserved_result_bases, v);
            HIP_TRY(grow(&ix->scratch[movi_index::kMask], &ix->scratch_cap[movi_index::kMask],
                         (size_t)pml_mask_words(ix->reserved_reads, ix->reserved_result_bases, 31u) * 4));
        } else {                                             // "reserve_host_reads"
            ix->reserved_reads = std::max<uint64_t>(ix->reserved_reads, v);
            if (ix->reserved_result_bases)
                HIP_TRY(grow(&ix->scratch[movi_index::kMask], &ix->scratch_cap[movi_index::kMask],
                             (size_t)pml_mask_words(ix->reserved_reads, ix->reserved_result_bases, 31u) * 4));
            HIP_TRY(grow(&ix->scratch[movi_index::kOffs], &ix->scratch_cap[movi_index::kOffs], (v + 1) * 8));
            HIP_TRY(grow(&ix->scratch[movi_index::kErr], &ix->scratch_cap[movi_index::kErr], v));
            if (ix->h_rel_cap < v + 1) {
                if (ix->h_rel) (void)hipHostFree(ix->h_rel);
                ix->h_rel = nullptr;
                ix->h_rel_cap = 0;
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ix->h_rel), (v + 1) * 8, hipHostMallocDefault));
                ix->h_rel_cap = v + 1;
            }
        }
        return MOVI_OK;
    }
    if (!strcmp(key, "pipe_chunk_bases")) {                  // test hook: many small chunks through the overlapped host path
        if (value < 0) return fail(MOVI_ERR_ARG, "pipe_chunk_bases must be >= 0");
        ix->pipe_chunk_bases = (uint64_t)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "seg_len")) {                           // segment-parallel long reads (PML): 0 = off
        if (value != 0 && (value < 32 || value > (1 << 24) || (value & 31) != 0))
            return fail(MOVI_ERR_ARG, "seg_len must be 0 (off) or a multiple of 32 in [32, 2^24]");
        ix->cfg.seg_len = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "seg_probe")) {                         // 0: segment eligible batches whatever the probe would say (tests)
        if (value < 0 || value > 2) return fail(MOVI_ERR_ARG, "seg_probe must be 0, 1 or 2");
        ix->cfg.seg_probe = (int)value;                      // 2: the caller's "seg_verdict" decides, nothing is read back
        return MOVI_OK;
    }
    if (!strcmp(key, "seg_verdict")) {
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "seg_verdict must be 0 or 1");
        ix->cfg.seg_verdict = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "pml_via_mask")) {                      // the walk's u16 vector through reset masks its wavefronts expand themselves (-1: the policy)
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "pml_via_mask must be -1, 0 or 1");
        ix->pml_via_mask = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "host_masks")) {                        // movi_pml_host: masks down + expansion on host worker threads (-1: the policy)
        if (value < -1 || value > 2) return fail(MOVI_ERR_ARG, "host_masks must be -1, 0, 1 or 2");
        ix->host_masks = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "host_mask_share")) {                   // mixed calls of movi_pml_host: percent of the bases that take the mask route
        if (value < 0 || value > 100) return fail(MOVI_ERR_ARG, "host_mask_share must be 0 .. 100");
        ix->host_mask_share = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "fused_expand")) {                      // A/B: 0 = a mask walk whose caller wants the vector leaves the expansion to the pml_expand_* kernels
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "fused_expand must be 0 or 1");
        ix->cfg.fused_expand = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "reserve_device_masks")) {              // device scratch for the mask words of movi_pml_device calls of up to `value` bases (and as many reads / 16)
        if (value < 0 || (uint64_t)value > (1ull << 40)) return fail(MOVI_ERR_ARG, "reserve_device_masks out of range");
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(grow(&ix->scratch[movi_index::kMask], &ix->scratch_cap[movi_index::kMask], (size_t)pml_mask_words((uint64_t)value / 16 + 1, (uint64_t)value, 0) * 4));
        return MOVI_OK;
    }
    if (!strcmp(key, "host_threads")) {                      // workers of the host-side mask expansion (0 = as many as the process may run on)
        if (value < 0 || value > 256) return fail(MOVI_ERR_ARG, "host_threads must be 0 .. 256");
        ix->host_threads = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "host_autopin")) {                      // 0: pageable buffers always take the synchronous path (A/B)
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "host_autopin must be 0 or 1");
        ix->host_autopin = value != 0;
        return MOVI_OK;
    }
    if (!strcmp(key, "host_overlap")) {                      // 0: the *_host calls never cut themselves into overlapped pieces (a caller
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "host_overlap must be 0 or 1");   // that pipelines chunk-sized calls itself)
        ix->host_overlap = value != 0;
        return MOVI_OK;
    }
    if (!strcmp(key, "stage_reads")) {                       // A/B: reads of short-read wavefronts staged through LDS
        if (value != 0 && value != 1) return fail(MOVI_ERR_ARG, "stage_reads must be 0 or 1");
        ix->cfg.stage_reads = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "kmer_k")) {                            // top-of-walk table: 0 = none, K in [1, 12] = build it now
        if (value < 0 || value > 12) return fail(MOVI_ERR_ARG, "kmer_k must be in [0, 12]");
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());                     // no walk may be reading the table that goes away
        ix->dev.kmer_k = 0;
        ix->dev.kmer = nullptr;
        if (ix->d_kmer) (void)hipFree(ix->d_kmer);
        ix->d_kmer = nullptr;
        ix->kmer_auto = 0;                                   // the caller's choice from here on
        if (value == 0) return MOVI_OK;
        if (!kmer_eligible(ix))
            return fail(MOVI_ERR_ARG, "the top-of-walk table serves PML walks on DNA (ACGT) *-thresholds indexes only");
        return build_kmer(ix, (uint32_t)value, nullptr);
    }
    if (!strcmp(key, "ahead_rows")) {                        // look-ahead rows: 0 = none (freed), 1 = build them now
        if (value < 0 || value > 1) return fail(MOVI_ERR_ARG, "ahead_rows must be 0 or 1");
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());                     // no walk may be reading the copy that goes away
        ix->dev.rows2 = nullptr;
        ix->dev.rows2_tail = 0;
        ix->dev.rows2_count = 0;
        ix->dev.hints = 0;
        if (ix->d_rows2) (void)hipFree(ix->d_rows2);
        ix->d_rows2 = nullptr;
        ix->ahead_auto = 0;                                  // the caller's choice from here on ...
        ix->dev.rows3 = nullptr;                             // ... and it is about what the PML walk runs on: the deep rows, which would take
        if (ix->d_rows3) (void)hipFree(ix->d_rows3);         // precedence over either answer, go (and are not built by themselves any more;
        ix->d_rows3 = nullptr;                               // "deep_rows" 1 brings them back)
        ix->deep_auto = 0;
        if (value == 0) return MOVI_OK;
        if (!ahead_eligible(ix)) return fail(MOVI_ERR_ARG, "look-ahead rows serve PML walks on *-thresholds indexes only");
        return build_ahead(ix, nullptr, false);
    }
    if (!strcmp(key, "deep_rows")) {                         // deep rows: 0 = none (freed), 1 = build them now
        if (value < 0 || value > 1) return fail(MOVI_ERR_ARG, "deep_rows must be 0 or 1");
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());                     // no walk may be reading the copy that goes away
        ix->dev.rows3 = nullptr;
        if (ix->d_rows3) (void)hipFree(ix->d_rows3);
        ix->d_rows3 = nullptr;
        ix->deep_auto = 0;                                   // the caller's choice from here on
        if (value == 0) return MOVI_OK;
        if (!deep_eligible(ix)) return fail(MOVI_ERR_ARG, "deep rows serve PML walks on *-thresholds indexes of fewer than 2^28 - 1 rows only");
        return build_deep(ix, nullptr);
    }
    if (!strcmp(key, "deep")) {                              // the walk on the deep rows the handle holds: -1 = batches of short reads, 0 never, 1 always (A/B)
        if (value < -1 || value > 1) return fail(MOVI_ERR_ARG, "deep must be -1, 0 or 1");
        ix->cfg.deep = (int)value;
        return MOVI_OK;
    }
    if (!strcmp(key, "ftab_k")) {                            // count query's interval table: 0 = none, K in [1, 12] = build it now
        if (value < 0 || value > 12) return fail(MOVI_ERR_ARG, "ftab_k must be in [0, 12]");
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());
        ix->dev.ftab_k = 0;
        ix->dev.ftab = nullptr;
        if (ix->d_ftab) (void)hipFree(ix->d_ftab);
        ix->d_ftab = nullptr;
        ix->ftab_auto = 0;
        if (value == 0) return MOVI_OK;
        if (!ftab_eligible(ix)) return fail(MOVI_ERR_ARG, "the count query's interval table serves DNA (ACGT) indexes only");
        return build_ftab_table(ix, (uint32_t)value, nullptr);
    }
    if (!strcmp(key, "waves_per_cu")) {
        if (value < 0 || value > 32) return fail(MOVI_ERR_ARG, "waves_per_cu must be in [0,32]");
        ix->cfg.waves_per_cu = (int)value;
        return MOVI_OK;
    }
    return fail(MOVI_ERR_ARG, std::string("unknown option: ") + key);
}

// ---------------------------------------------------------------------------- PML

// mask (optional; PML only): reset masks out instead of the vector -- mask->words / phase from the caller; a path without a mask output
// of its own writes its vector to *tmp_p first (grow-only device scratch: the handle's, or a pipeline slot's).
static int ml_device(bool zml, movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                     uint64_t n_bases, uint16_t *d_out, uint8_t *d_read_err, const uint32_t *d_read_order, void *stream,
                     const ClsArgs &cls = ClsArgs(), DevStats *d_stats = nullptr, SegWorkspace *seg_ws = nullptr,
                     int ragged_hint = -1, int *seg_verdict = nullptr, const MaskArgs *mask = nullptr, void **tmp_p = nullptr,
                     size_t *tmp_cap = nullptr) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (!d_stats) d_stats = ix->d_stats;                     // (the pipelined host path counts per chunk in flight ...
    if (!seg_ws) seg_ws = &ix->seg_ws;                       //  ... and keeps a segment workspace per chunk in flight)
    if (!zml && !mode_has_thresholds(ix->desc.mode))
        return fail(MOVI_ERR_ARG, "PML needs thresholds: on a `regular`, `blocked` or `sampled` index the reference repositions "
                                  "randomly (reposition_randomly), which cannot be reproduced; use --zml or --count, or a "
                                  "*-thresholds index");
    if (n_reads == 0) return MOVI_OK;
    const bool bins_only = cls.bin_width != 0 && !d_out;
    const bool masks = mask && mask->words;
    if (masks && (zml || cls.bin_width != 0 || cls.log_ff)) return fail(MOVI_ERR_ARG, "reset masks: plain PML queries only");
    if (!d_offsets || (n_bases && (!d_bases || (!d_out && !bins_only && !masks)))) return fail(MOVI_ERR_ARG, "NULL device buffer");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(DevStats), s));
    if (n_reads > 0xFFFFFFFFull) return fail(MOVI_ERR_ARG, "more than 2^32 reads in one call");
    // the first PML query on a handle builds its derived tables -- unless movi_index_prepare did (--logs runs on the first
    // kernel, which uses none of them)
    if (!zml && cls.log_ff == nullptr) ensure_pml_tables(ix, s);
    if (zml)
        HIP_TRY(launch_zml(ix->kmode, ix->dev, d_bases, d_offsets, n_reads, n_bases, d_out, d_read_err, d_stats,
                           d_read_order, ix->cfg, s, seg_ws, ragged_hint, seg_verdict, &ix->last_launch));
    else {
        MaskArgs m;
        if (masks) {
            m = *mask;
            if (!m.expand_out && pml_mask_needs_tmp(ix->dev, ix->cfg, n_reads, n_bases, seg_ws != nullptr)) {   // (expand_out: such a batch writes the vector itself)
                if (!tmp_p) { tmp_p = &ix->scratch[movi_index::kTmp]; tmp_cap = &ix->scratch_cap[movi_index::kTmp]; }
                HIP_TRY(grow(tmp_p, tmp_cap, n_bases * 2));
                m.tmp_pml = static_cast<uint16_t *>(*tmp_p);
            }
        }
        // (device-pointer calls: the probe's verdict is remembered for the next 15 calls on batches of the same shape -- mean read length
        // and read count to a factor of two --: such calls then enqueue the walk without the probe and its read-back.  Either verdict
        // gives the same answers; a stale one costs speed for at most those calls.  "seg_probe" 0 / 2 never probe anyway.)
        int cached = -1;
        const bool use_cache = seg_verdict == nullptr && ix->cfg.seg_probe == 1 && n_bases / n_reads >= 2ull * (uint64_t)std::max(ix->cfg.seg_len, 1);
        const uint32_t key = use_cache ? ((uint32_t)(63 - __builtin_clzll(n_bases / n_reads)) << 8) | (uint32_t)(63 - __builtin_clzll(n_reads)) : 0u;
        if (use_cache) {
            if (ix->seg_cache_left > 0 && ix->seg_cache_key == key) { cached = ix->seg_cache_verdict; ix->seg_cache_left -= 1; }
            seg_verdict = &cached;
        }
        const int before = cached;
        HIP_TRY(launch_pml(ix->kmode, ix->dev, d_bases, d_offsets, n_reads, n_bases, d_out, d_read_err, d_stats,
                           d_read_order, ix->cfg, s, cls, seg_ws, ragged_hint, seg_verdict, &ix->last_launch, m));
        if (use_cache && before < 0 && cached >= 0) { ix->seg_cache_verdict = cached; ix->seg_cache_key = key; ix->seg_cache_left = 15; }
    }
    return MOVI_OK;
}

int movi_pml_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                    uint64_t n_bases, uint16_t *d_out_pml, uint8_t *d_read_err, const uint32_t *d_read_order,
                    void *stream) {
    if (ix && n_reads && d_out_pml && d_offsets && mode_has_thresholds(ix->desc.mode) && ix->pml_via_mask != 0) {
        // The vector THROUGH RESET MASKS: the walk writes one bit per base (a ninth fewer vector instructions per iteration than the
        // u16 packer, a third of the write traffic) and every wavefront, when its 64 walks are over, expands its reads' words into the
        // vector itself -- the expansion runs under the other wavefronts' gathers (pml_kernel_flatp's tail; "fused_expand" 0: the
        // pml_expand_* kernels behind the walk).  "pml_via_mask" 1: always; -1, the default: where the walk runs on the deep rows -- three
        // emissions per iteration: c2 78.4 -> 86.7 Gbases/s; the 1 B-row table 42.3 -> 46.0 --, i.e. batches of short reads; long reads keep
        // the ring in LDS, which is as good there (c3: 16.76 against 16.66 ms).  The mask words live in device scratch of the handle
        // (grow-only: "reserve_device_masks" reserves it, and movi_index_prepare's promise of no allocation inside a query holds from the
        // first call of each size on).
        HIP_TRY(hipSetDevice(ix->device));
        hipStream_t s = static_cast<hipStream_t>(stream);
        ensure_pml_tables(ix, s);
        if (ix->pml_via_mask > 0 || pml_vector_via_masks(ix->dev, ix->cfg, n_reads, n_bases, true, d_read_order != nullptr)) {
            HIP_TRY(grow(&ix->scratch[movi_index::kMask], &ix->scratch_cap[movi_index::kMask], (size_t)pml_mask_words(n_reads, n_bases, 0) * 4));
            MaskArgs m;
            m.words = static_cast<uint32_t *>(ix->scratch[movi_index::kMask]);
            m.phase = 0;
            m.expand_out = d_out_pml;
            return ml_device(false, ix, d_bases, d_offsets, n_reads, n_bases, nullptr, d_read_err, d_read_order, stream, ClsArgs(), nullptr, nullptr,
                             -1, nullptr, &m);
        }
    }
    return ml_device(false, ix, d_bases, d_offsets, n_reads, n_bases, d_out_pml, d_read_err, d_read_order, stream);
}

int movi_pml_mask_words(uint64_t n_reads, uint64_t n_bases, uint64_t first_base, uint64_t *n_words) {
    if (!n_words) return fail(MOVI_ERR_ARG, "n_words is NULL");
    *n_words = pml_mask_words(n_reads, n_bases, (uint32_t)(first_base & 31u));
    return MOVI_OK;
}

int movi_pml_mask_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                         uint64_t n_bases, uint64_t first_base, uint32_t *d_mask_words, uint8_t *d_read_err,
                         const uint32_t *d_read_order, void *stream) {
    if (n_reads && !d_mask_words) return fail(MOVI_ERR_ARG, "NULL device buffer");
    MaskArgs m;
    m.words = d_mask_words;
    m.phase = (uint32_t)(first_base & 31u);
    return ml_device(false, ix, d_bases, d_offsets, n_reads, n_bases, nullptr, d_read_err, d_read_order, stream, ClsArgs(), nullptr,
                     nullptr, -1, nullptr, &m);
}

int movi_pml_expand_device(movi_index_t *ix, const uint32_t *d_mask_words, const uint64_t *d_offsets, uint64_t n_reads,
                           uint64_t n_bases, uint64_t first_base, uint16_t *d_out_pml, void *stream) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) return MOVI_OK;
    if (!d_mask_words || !d_offsets || (n_bases && !d_out_pml)) return fail(MOVI_ERR_ARG, "NULL device buffer");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_pml_expand(d_mask_words, d_offsets, n_reads, n_bases, (uint32_t)(first_base & 31u), d_out_pml,
                              static_cast<hipStream_t>(stream)));
    return MOVI_OK;
}

int movi_zml_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                    uint64_t n_bases, uint16_t *d_out_zml, uint8_t *d_read_err, const uint32_t *d_read_order,
                    void *stream) {
    return ml_device(true, ix, d_bases, d_offsets, n_reads, n_bases, d_out_zml, d_read_err, d_read_order, stream);
}

int movi_last_launch(const movi_index_t *ix, movi_launch_info_t *info) {
    if (!ix || !info) return fail(MOVI_ERR_ARG, "NULL argument");
    memset(info, 0, sizeof(*info));
    static_assert(sizeof(info->kernel) == sizeof(ix->last_launch.kernel), "kernel name buffers differ");
    memcpy(info->kernel, ix->last_launch.kernel, sizeof(info->kernel));
    info->variant = ix->last_launch.variant;
    info->block_threads = ix->last_launch.block_threads;
    info->waves_per_cu = ix->last_launch.waves_per_cu;
    info->segmented = ix->last_launch.segmented;
    info->idx64 = ix->last_launch.idx64;
    info->staged = ix->last_launch.staged;
    info->ahead = ix->last_launch.ahead;
    return MOVI_OK;
}

int movi_launch_log(char *buf, size_t cap, size_t *needed) {
    const size_t n = take_launch_log(buf, cap);
    if (needed) *needed = n + 1;
    return MOVI_OK;
}

int movi_index_info(const movi_index_t *ix, const char *key, double *value) {
    if (!ix || !key || !value) return fail(MOVI_ERR_ARG, "NULL argument");
    const double rows = ix->kmode != (int)ix->desc.mode ? (double)ix->desc.r * 8.0 : (double)ix->rows_bytes;
    const double kmer = ix->d_kmer ? (double)((size_t)16 << (2 * ix->dev.kmer_k)) : 0.0;
    const double ftab = ix->d_ftab ? (double)((size_t)16 << (2 * ix->dev.ftab_k)) : 0.0;
    const double ahead = ix->d_rows2 ? (double)ahead_rows_bytes(ix->desc.r) : 0.0;
    const double deep = ix->d_rows3 ? (double)deep_rows_bytes(ix->desc.r) : 0.0;
    const double ckpt = ix->d_ckpt ? (double)((ix->desc.r >> kPrefixShift) + 2) * 8.0 : 0.0;
    if (!strcmp(key, "rows_bytes")) *value = rows;
    else if (!strcmp(key, "kmer_bytes")) *value = kmer;
    else if (!strcmp(key, "ftab_bytes")) *value = ftab;
    else if (!strcmp(key, "ahead_rows_bytes")) *value = ahead;
    else if (!strcmp(key, "deep_rows_bytes")) *value = deep;
    else if (!strcmp(key, "ckpt_bytes")) *value = ckpt;
    else if (!strcmp(key, "derived_bytes")) *value = kmer + ftab + ahead + deep + ckpt;
    else if (!strcmp(key, "ahead_no_ff")) *value = ix->ahead_tallied ? ix->ahead_no_ff : -1.0;
    else if (!strcmp(key, "host_staging_bytes")) {           // device staging the synchronous *_host calls hold at the moment
        double b = 0.0;
        for (int k = 0; k < movi_index::kScratchSlots; k++) b += (double)ix->scratch_cap[k];
        *value = b;
    }
    else return fail(MOVI_ERR_ARG, std::string("unknown info key: ") + key);
    return MOVI_OK;
}

int movi_last_stats(movi_index_t *ix, void *stream, movi_query_stats_t *stats) {
    if (!ix || !stats) return fail(MOVI_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    DevStats h{};
    HIP_TRY(hipMemcpy(&h, ix->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    stats->bases = 0;                      // not tracked on the device: the *_host entry points fill it in
    stats->fast_forwards = h.fast_forwards;
    stats->scans = h.scans;
    stats->repositions = h.repositions;
    stats->errors = h.errors;
    stats->lane_steps = h.lane_steps;
    stats->wave_steps = h.wave_steps;
    stats->segments = h.segments;
    stats->rewalked = h.rewalked;
    return MOVI_OK;
}

}  // extern "C"

namespace {

// Reads are cut into chunks so that the staging buffers stay bounded whatever the input size:
// about kChunkBases bases per launch -- but never so few READS that the GPU runs empty.  One lane
// walks one read, so a chunk of long reads (2^28 bases = 27 k reads of 10 kbp) would leave most of
// the 524 k lane slots idle and the walk latency-bound; such chunks grow until they hold
// kMinChunkReads reads or kMaxChunkBases bases (3 B of device memory per base, 6 GiB).
constexpr uint64_t kChunkBases = 1ull << 28;
constexpr uint64_t kMinChunkReads = 1ull << 18;
constexpr uint64_t kMaxChunkBases = 1ull << 31;
// The overlapped path (page-locked caller buffers) wants several chunks per call -- chunk i+1 goes up and chunk i-1 comes
// down while chunk i is walked, and the kernels of neighbouring chunks share the GPU -- so it cuts finer: about an
// eighth of the call, between 2^22 and 2^28 bases, and at least 2^15 reads (a download can only start when a walk
// has finished, so chunks must be short; the lanes in flight are those of the kPipeAhead chunks being walked, and a
// read takes as long as it takes however few lanes walk beside it -- long reads are best cut into just those
// kPipeAhead chunks: 100 k x 10 kbp as 3 x 33 k reads 57 ms, as 8 x 12.5 k reads 85 ms, synchronous 74 ms).
constexpr uint64_t kPipeMinBases = 1ull << 22;
constexpr uint64_t kPipeMinReads = 1ull << 15;
constexpr uint64_t kPipeTargetChunks = 8;
constexpr uint64_t kPipeMaskChunks = 16;

hipError_t grow(void **p, size_t *cap, size_t bytes) {
    if (bytes < 8) bytes = 8;
    if (*cap < bytes) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
        const size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(p, want);
        if (e != hipSuccess) return e;
        *cap = want;
    }
    return hipSuccess;
}

// Where one chunk of a *_host call lives: the stream it runs on, its counters, its device staging (the handle's in the
// synchronous path, a pipeline slot's in the overlapped one) and, overlapped only, page-locked room for per-read results.
struct ChunkCtx {
    movi_index *ix = nullptr;
    hipStream_t s = nullptr;
    DevStats *d_stats = nullptr;
    void **d = nullptr;
    size_t *cap = nullptr;
    uint8_t *h_small = nullptr;
    SegWorkspace *seg_ws = nullptr;
    int ragged_hint = -1;            // the chunk's longest read is (1) / is not (0) more than 1.5 x its mean: launch_pml's segment policy
    int *seg_verdict = nullptr;      // one probe per call: the first chunk's verdict serves the others (launch_pml_segmented)
    uint64_t first = 0, b0 = 0;      // the chunk: its first read and the position of its first base in the caller's arrays
    HostPool::Group *grp = nullptr;  // overlapped path: the slot's group of worker-pool tasks (what reads the slot's page-locked block)
    bool async = false;
    hipError_t alloc(int slot, size_t bytes, void **out) {
        hipError_t e = grow(&d[slot], &cap[slot], bytes);
        *out = d[slot];
        return e;
    }
    // device -> host: straight into the caller's buffer (synchronous path, or a page-locked destination) ...
    hipError_t down(void *h_dst, const void *d_src, size_t bytes) {
        if (!bytes) return hipSuccess;
        return async ? hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s)
                     : hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost);
    }
    // ... or, overlapped path with a destination that may be pageable (a copy into pageable memory would hold the host
    // until the kernel is done): into the slot's page-locked block at `small_off`; harvest() moves it on
    hipError_t down_small(void *h_dst, size_t small_off, const void *d_src, size_t bytes) {
        return down(async ? static_cast<void *>(h_small + small_off) : h_dst, d_src, bytes);
    }
};

// Page-locked (hipHostMalloc / hipHostRegister) host memory?  Pageable pointers are unknown to the runtime.
bool is_pinned(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

}  // namespace

static void release_scratch(movi_index *ix) {
    for (int k = 0; k < movi_index::kScratchSlots; k++) {
        if (ix->scratch[k]) (void)hipFree(ix->scratch[k]);
        ix->scratch[k] = nullptr;
        ix->scratch_cap[k] = 0;
    }
    if (ix->pipe_up) {
        (void)hipStreamSynchronize(ix->pipe_up);
        (void)hipStreamDestroy(ix->pipe_up);
        ix->pipe_up = nullptr;
    }
    for (hipEvent_t e : ix->pipe_ev) (void)hipEventDestroy(e);
    ix->pipe_ev.clear();
    if (ix->seg_ws.buf) (void)hipFree(ix->seg_ws.buf);
    ix->seg_ws = SegWorkspace();
    if (ix->h_rel) (void)hipHostFree(ix->h_rel);
    ix->h_rel = nullptr;
    ix->h_rel_cap = 0;
    for (auto &sl : ix->pipe) {
        if (sl.s) (void)hipStreamSynchronize(sl.s);
        if (sl.seg_ws.buf) (void)hipFree(sl.seg_ws.buf);
        sl.seg_ws = SegWorkspace();
        for (int k = 0; k < movi_index::kScratchSlots; k++) {
            if (sl.d[k]) (void)hipFree(sl.d[k]);
            sl.d[k] = nullptr;
            sl.cap[k] = 0;
        }
        if (sl.d_stats) (void)hipFree(sl.d_stats);
        sl.d_stats = nullptr;
        if (sl.h) (void)hipHostFree(sl.h);
        sl.h = nullptr;
        sl.h_cap = 0;
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        sl.ev = nullptr;
        if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
        sl.ev_up = nullptr;
        if (sl.s) (void)hipStreamDestroy(sl.s);
        sl.s = nullptr;
    }
}

namespace {

// The offsets are the caller's: every *_host entry point validates them all before they size a chunk, an
// allocation or a copy (and before the device is touched, so the check is testable without one).
int check_offsets(const uint64_t *h_offsets, uint64_t n_reads) {
    // one branch-free pass (a decreasing pair wraps to >= 2^63, an overlong read is >= 2^32: either way high bits);
    // the slow pass below only runs to name the first offender
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n_reads; i++) bad |= (h_offsets[i + 1] - h_offsets[i]) >> 32;
    if (!bad) return MOVI_OK;
    for (uint64_t i = 0; i < n_reads; i++) {
        if (h_offsets[i + 1] < h_offsets[i]) return fail(MOVI_ERR_ARG, "read offsets are not non-decreasing");
        if (h_offsets[i + 1] - h_offsets[i] > 0xFFFFFFFFull) return fail(MOVI_ERR_ARG, "a read is longer than 2^32 - 1 bases");
    }
    return MOVI_OK;
}

void add_stats(movi_query_stats_t *acc, uint64_t nb, const DevStats &h) {
    acc->bases += nb;
    acc->fast_forwards += h.fast_forwards;
    acc->scans += h.scans;
    acc->repositions += h.repositions;
    acc->errors += h.errors;
    acc->lane_steps += h.lane_steps;
    acc->wave_steps += h.wave_steps;
    acc->segments += h.segments;
    acc->rewalked += h.rewalked;
}

// One kind of query behind a *_host entry point:
//   launch(ctx, d_bases, d_offs, nr, nb, d_err)  allocates the chunk's result staging from ctx and enqueues the kernel;
//   fetch(ctx, first, nr, b0, nb)                enqueues the results' way back (ctx.down / ctx.down_small);
//   harvest(h_small, first, nr, group)           overlapped path only, after the chunk's stream has drained: per-read
//                                                results from the page-locked block to the caller's arrays;
//   small_bytes                                  page-locked bytes per read that fetch / harvest use.
template <typename Launch, typename Fetch, typename Harvest>
int run_chunked(movi_index *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                uint8_t *h_read_err, movi_query_stats_t *stats, Launch launch, Fetch fetch, Harvest, size_t, size_t) {
    if (stats) memset(stats, 0, sizeof(*stats));
    int seg_verdict = -1;
    ChunkCtx ctx;
    ctx.ix = ix;
    ctx.seg_verdict = &seg_verdict;
    ctx.d_stats = ix->d_stats;
    ctx.d = ix->scratch;
    ctx.cap = ix->scratch_cap;
    ctx.seg_ws = &ix->seg_ws;
    uint64_t first = 0;
    while (first < n_reads) {
        uint64_t last = first + 1;
        while (last < n_reads) {
            const uint64_t nb_next = h_offsets[last + 1] - h_offsets[first];
            if (nb_next <= kChunkBases || (last - first < kMinChunkReads && nb_next <= kMaxChunkBases)) ++last;
            else break;
        }
        const uint64_t nr = last - first;
        const uint64_t b0 = h_offsets[first], nb = h_offsets[last] - b0;
        struct { void *p; } d_bases{}, d_offs{}, d_err{};
        static const bool trace = getenv("MOVI_TRACE_HOST_CALLS") != nullptr;   // diagnostic: where a synchronous host call's time goes, to stderr
        double tr[8] = {0};
        auto stamp = [&](int k) { if (trace) tr[k] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        stamp(0);
        HIP_TRY(ctx.alloc(movi_index::kBases, nb, &d_bases.p));
        HIP_TRY(ctx.alloc(movi_index::kOffs, (nr + 1) * 8, &d_offs.p));
        HIP_TRY(ctx.alloc(movi_index::kErr, nr, &d_err.p));
        if (ix->h_rel_cap < nr + 1) {
            if (ix->h_rel) (void)hipHostFree(ix->h_rel);
            ix->h_rel = nullptr;
            ix->h_rel_cap = 0;
            const size_t want = (size_t)(nr + 1) + (size_t)((nr + 1) >> 3);
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ix->h_rel), want * 8, hipHostMallocDefault));
            ix->h_rel_cap = want;
        }
        uint64_t *rel = ix->h_rel;                            // (read by the copy below before the call returns: hipMemcpy is synchronous)
        uint64_t longest = 0;
        for (uint64_t i = 0; i <= nr; i++) {
            rel[i] = h_offsets[first + i] - b0;
            if (i && rel[i] - rel[i - 1] > longest) longest = rel[i] - rel[i - 1];
        }
        ctx.ragged_hint = (nr && longest * 2 > (nb / nr) * 3) ? 1 : 0;
        // No length sort here: on ragged batches handing the lanes out longest-first measured
        // slightly SLOWER (31.1 vs 32.8 Gbases/s, log-normal lengths) -- the walk is bound by the
        // memory system, not by lane occupancy, and the dispatcher already refills whole blocks.
        stamp(1);
        if (nb) HIP_TRY(hipMemcpy(d_bases.p, h_bases + b0, nb, hipMemcpyHostToDevice));
        stamp(2);
        HIP_TRY(hipMemcpy(d_offs.p, rel, (nr + 1) * 8, hipMemcpyHostToDevice));
        stamp(3);
        ctx.first = first;
        ctx.b0 = b0;
        int rc = launch(ctx, static_cast<const uint8_t *>(d_bases.p), static_cast<const uint64_t *>(d_offs.p), nr, nb,
                        static_cast<uint8_t *>(d_err.p));
        if (rc) return rc;
        stamp(4);
        movi_query_stats_t st{};
        rc = movi_last_stats(ix, nullptr, &st);
        if (rc) return rc;
        stamp(5);
        rc = fetch(ctx, first, nr, b0, nb);
        if (rc) return rc;
        stamp(6);
        if (h_read_err) HIP_TRY(hipMemcpy(h_read_err + first, d_err.p, nr, hipMemcpyDeviceToHost));
        stamp(7);
        if (trace)
            fprintf(stderr, "[movi_hip] host call: %llu reads %llu bases | staging + offsets %.3f | bases up %.3f | offsets up %.3f | launch %.3f | walk + counters %.3f | "
                            "results down %.3f | error bytes down %.3f ms\n", (unsigned long long)nr, (unsigned long long)nb, (tr[1] - tr[0]) * 1e3, (tr[2] - tr[1]) * 1e3,
                    (tr[3] - tr[2]) * 1e3, (tr[4] - tr[3]) * 1e3, (tr[5] - tr[4]) * 1e3, (tr[6] - tr[5]) * 1e3, (tr[7] - tr[6]) * 1e3);
        if (stats) {
            stats->bases += nb;
            stats->fast_forwards += st.fast_forwards;
            stats->scans += st.scans;
            stats->repositions += st.repositions;
            stats->errors += st.errors;
            stats->lane_steps += st.lane_steps;
            stats->wave_steps += st.wave_steps;
            stats->segments += st.segments;
            stats->rewalked += st.rewalked;
        }
        first = last;
    }
    return MOVI_OK;
}

// The same call with the caller's bases (and, for PML / ZML, result vector) in page-locked memory: kPipeSlots chunks in
// flight, each on its own stream -- upload, walk, download -- so that the three overlap across chunks: the call then
// costs about what its slowest leg does (2 B per base coming down: ~28 Gbases/s on a 56 GB/s link) instead of the
// sum of the three.
// A chunk's download is enqueued only once its walk HAS finished (the host waits on an event), never behind it as a
// dependent copy: a copy that waits for a kernel sits in its SDMA ring as a poll packet and holds up every copy queued
// after it -- the next chunk's upload included.  Enqueued that way (the first version) the timeline was strictly serial,
// chunk after chunk (rocprofv3 --memory-copy-trace), and the call no faster than the synchronous one.
template <typename Launch, typename Fetch, typename Harvest>
int run_pipelined(movi_index *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                  uint8_t *h_read_err, movi_query_stats_t *stats, Launch launch, Fetch fetch, Harvest harvest,
                  size_t small_bytes, size_t mask_word_bytes, bool vector_down) {
    if (stats) memset(stats, 0, sizeof(*stats));
    movi_query_stats_t acc{};
    const uint64_t total = h_offsets[n_reads] - h_offsets[0];
    // (calls whose results -- all or part of them -- come down as reset masks for the host's worker pool are cut twice as fine: the pool's
    // work arrives earlier and more evenly; 1 M x 150 bp: masks alone 22.9 -> 32.4 Gbases/s, both ways down 30 -> 33.9; finer still and
    // the walks no longer fill the GPU: profiles/r06_host_path.txt)
    uint64_t target = total / (mask_word_bytes ? kPipeMaskChunks : kPipeTargetChunks);
    target = target < kPipeMinBases ? kPipeMinBases : (target > kChunkBases ? kChunkBases : target);
    uint64_t min_reads = kPipeMinReads;
    // Long reads are cut into few, big chunks because a read takes its time however few lanes walk beside it -- unless the
    // reads are walked segment-parallel, which the host only knows afterwards: a caller whose last call was (a stream of
    // batches of the same kind of reads) gets chunks of >= 2^12 reads from then on, whose downloads start earlier.
    if (ix->seg_seen && ix->cfg.seg_len > 0 && total / n_reads >= 2ull * (uint64_t)ix->cfg.seg_len) min_reads = 1ull << 12;
    if (ix->pipe_chunk_bases) { target = ix->pipe_chunk_bases; min_reads = 1; }
    struct Chunk { uint64_t first, nr, b0, nb; };
    std::vector<Chunk> chunks;
    for (uint64_t first = 0; first < n_reads;) {
        // the longest run of reads from `first` with at most `target` bases (the offsets are non-decreasing: checked) --
        // at least one read, at least min_reads of them if that stays below kMaxChunkBases
        const uint64_t *lo = h_offsets + first + 1, *end = h_offsets + n_reads + 1;
        // (the first download can only start when the first walk is over: the first two chunks are a quarter and
        // half the size)
        uint64_t tgt = target;
        // (not where masks come down: a chunk's masks are 1/16 of its vector, and sixteen equal chunks ran 8 % faster than the same with a
        // half-size second one)
        if (!ix->pipe_chunk_bases && !mask_word_bytes && chunks.size() < 2 && (target >> (2 - chunks.size())) >= kPipeMinBases)
            tgt = target >> (2 - chunks.size());
        uint64_t last = (uint64_t)(std::upper_bound(lo, end, h_offsets[first] + tgt) - h_offsets) - 1;
        if (last - first < min_reads) {
            const uint64_t cap = (uint64_t)(std::upper_bound(lo, end, h_offsets[first] + kMaxChunkBases) - h_offsets) - 1;
            last = std::min(first + min_reads, cap);
        }
        if (last <= first) last = first + 1;
        if (last > n_reads) last = n_reads;
        chunks.push_back({first, last - first, h_offsets[first], h_offsets[last] - h_offsets[first]});
        first = last;
    }
    uint8_t *all = nullptr;                                    // the call's reads in one device buffer (set below, ahead of the loop), or chunk by chunk
    constexpr int S = movi_index::kPipeSlots;
    struct InFlight { int stage = 0; Chunk c{}; } fl[S];       // 0 free, 1 walking (upload + kernel enqueued), 2 coming down
    HostPool::Group grp[S];                                    // host-side work a slot's harvest has handed to the worker pool (it reads the slot's page-locked block)
    // layout of a slot's page-locked block
    auto off_err = [](uint64_t nr) { return (size_t)(nr + 1) * 8; };
    auto off_small = [&](uint64_t nr) { return (off_err(nr) + (size_t)nr + 15) & ~(size_t)15; };
    // (mask_word_bytes = 4: the block also holds the chunk's reset-mask words, behind the per-read results)
    auto off_stats = [&](uint64_t nr, uint64_t nb) {
        return (off_small(nr) + (size_t)nr * small_bytes + (mask_word_bytes ? (size_t)pml_mask_words(nr, nb, 31u) * mask_word_bytes : 0) + 15) & ~(size_t)15;
    };
    int seg_verdict = -1;
    auto ctx_of = [&](int k, uint64_t nr) {
        movi_index::PipeSlot &sl = ix->pipe[k];
        ChunkCtx ctx;
        ctx.ix = ix;
        ctx.seg_verdict = &seg_verdict;
        ctx.s = sl.s;
        ctx.d_stats = sl.d_stats;
        ctx.d = sl.d;
        ctx.cap = sl.cap;
        ctx.h_small = sl.h + off_small(nr);
        ctx.seg_ws = &sl.seg_ws;
        ctx.async = true;
        ctx.first = fl[k].c.first;
        ctx.b0 = fl[k].c.b0;
        ctx.grp = &grp[k];
        return ctx;
    };
    // a chunk's stream has drained: counters, error bytes, per-read results
    auto finish = [&](int k) -> int {
        movi_index::PipeSlot &sl = ix->pipe[k];
        InFlight &f = fl[k];
        if (f.stage == 0) return MOVI_OK;
        f.stage = 0;
        HIP_TRY(hipStreamSynchronize(sl.s));
        DevStats h;
        memcpy(&h, sl.h + off_stats(f.c.nr, f.c.nb), sizeof(h));
        add_stats(&acc, f.c.nb, h);
        if (h_read_err) memcpy(h_read_err + f.c.first, sl.h + off_err(f.c.nr), f.c.nr);
        harvest(sl.h + off_small(f.c.nr), f.c.first, f.c.nr, &grp[k]);
        return MOVI_OK;
    };
    // chunk c goes up into slot k and is walked
    auto up = [&](const Chunk &c, int k, size_t ci) -> int {
        movi_index::PipeSlot &sl = ix->pipe[k];
        if (int rc = finish(k)) return rc;
        HostPool::get().wait(&grp[k]);                       // (the slot's block is about to be rewritten)
        if (!sl.s) HIP_TRY(hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking));
        if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        if (!sl.ev_up) HIP_TRY(hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming));
        if (!ix->pipe_up) HIP_TRY(create_upload_stream(&ix->pipe_up));
        if (!sl.d_stats) HIP_TRY(hipMalloc(&sl.d_stats, sizeof(DevStats)));
        const size_t hb = off_stats(c.nr, c.nb) + sizeof(DevStats);
        if (sl.h_cap < hb) {
            if (sl.h) (void)hipHostFree(sl.h);
            sl.h = nullptr;
            sl.h_cap = 0;
            const size_t want = hb + (hb >> 2);
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sl.h), want, hipHostMallocDefault));
            sl.h_cap = want;
        }
        fl[k].c = c;
        ChunkCtx ctx = ctx_of(k, c.nr);
        struct { void *p; } d_bases{}, d_offs{}, d_err{};
        if (all) d_bases.p = all + (c.b0 - chunks[0].b0);
        else HIP_TRY(ctx.alloc(movi_index::kBases, c.nb, &d_bases.p));
        HIP_TRY(ctx.alloc(movi_index::kOffs, (c.nr + 1) * 8, &d_offs.p));
        HIP_TRY(ctx.alloc(movi_index::kErr, c.nr, &d_err.p));
        uint64_t *rel = reinterpret_cast<uint64_t *>(sl.h);
        uint64_t longest = 0;
        for (uint64_t i = 0; i <= c.nr; i++) {
            rel[i] = h_offsets[c.first + i] - c.b0;
            if (i && rel[i] - rel[i - 1] > longest) longest = rel[i] - rel[i - 1];
        }
        ctx.ragged_hint = (c.nr && longest * 2 > (c.nb / c.nr) * 3) ? 1 : 0;
        fl[k].c = c;
        fl[k].stage = 1;                                     // from here on the streams hold work that touches caller memory
        // (the upload stream carries the bases alone, copy behind copy: with the chunk's offsets in between, every chunk left it idle for
        // ~40 us -- a fifth of a 9.4 MB copy; the offsets go up on the chunk's own stream, ahead of its walk: profiles/r06_host_path.txt)
        if (!all) {
            if (c.nb) HIP_TRY(hipMemcpyAsync(d_bases.p, h_bases + c.b0, c.nb, hipMemcpyHostToDevice, ix->pipe_up));
            HIP_TRY(hipEventRecord(sl.ev_up, ix->pipe_up));
            HIP_TRY(hipMemcpyAsync(d_offs.p, rel, (c.nr + 1) * 8, hipMemcpyHostToDevice, sl.s));
        } else {
            // (by a kernel that reads the slot's page-locked block, not by the copy engine: that one has the whole call's reads queued)
            HIP_TRY(launch_copy_words(static_cast<uint64_t *>(d_offs.p), rel, c.nr + 1, sl.s));
        }
        HIP_TRY(hipStreamWaitEvent(sl.s, all ? ix->pipe_ev[ci] : sl.ev_up, 0));
        if (int rc = launch(ctx, static_cast<const uint8_t *>(d_bases.p), static_cast<const uint64_t *>(d_offs.p), c.nr, c.nb,
                            static_cast<uint8_t *>(d_err.p)))
            return rc;
        HIP_TRY(hipEventRecord(sl.ev, sl.s));
        return MOVI_OK;
    };
    // slot k's walk is over: its results start their way down
    auto down = [&](int k) -> int {
        movi_index::PipeSlot &sl = ix->pipe[k];
        const Chunk &c = fl[k].c;
        HIP_TRY(hipEventSynchronize(sl.ev));
        ChunkCtx ctx = ctx_of(k, c.nr);
        if (int rc = fetch(ctx, c.first, c.nr, c.b0, c.nb)) return rc;
        HIP_TRY(hipMemcpyAsync(sl.h + off_err(c.nr), sl.d[movi_index::kErr], c.nr, hipMemcpyDeviceToHost, sl.s));
        HIP_TRY(hipMemcpyAsync(sl.h + off_stats(c.nr, c.nb), sl.d_stats, sizeof(DevStats), hipMemcpyDeviceToHost, sl.s));
        fl[k].stage = 2;
        return MOVI_OK;
    };
    // chunks that have arrived on the host are harvested as soon as the loop comes by, not when their slot is needed again:
    // a harvest may have real work to hand on (the mask path's expansion runs on the host's worker threads beside the next walks)
    auto finish_arrived = [&]() -> int {
        for (int k = 0; k < S; k++) {
            if (fl[k].stage != 2) continue;
            const hipError_t q = hipStreamQuery(ix->pipe[k].s);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
            if (int rc = finish(k)) return rc;
        }
        return MOVI_OK;
    };
    // kPipeAhead chunks going up or being walked (their kernels share the GPU: the lanes in flight are the sum of
    // theirs) while one comes down; twice as many slots, so that a slot's previous chunk has long arrived on the host
    // when the slot is reused (with kPipeAhead + 1 slots every upload waited for the download issued just before it,
    // and the downloads did not queue back to back)
    int rc = MOVI_OK;
    const size_t n = chunks.size();
    size_t next_up = 0;
    // ALL THE READS GO UP AHEAD OF THE LOOP (calls of up to 2^31 bases, into the handle's call-wide staging): the copies queued behind one
    // another, an event behind each, and a chunk's walk waits for its event.  Enqueued chunk by chunk from the loop -- three ahead of the
    // walk being waited for -- the upload stream ran dry two or three times per call (0.1 ms each), and the upload is what bounds the
    // call.  (The chunks' offsets then must not travel by the copy engine: they would queue behind the whole call's reads.)
    if (n > 1 && total <= (1ull << 31) && !vector_down) {
        hipError_t e = hipSuccess;
        if (!ix->pipe_up) e = create_upload_stream(&ix->pipe_up);
        if (e == hipSuccess) e = grow(&ix->scratch[movi_index::kBases], &ix->scratch_cap[movi_index::kBases], total);
        while (e == hipSuccess && ix->pipe_ev.size() < n) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) ix->pipe_ev.push_back(ev);
        }
        if (e == hipSuccess) {
            all = static_cast<uint8_t *>(ix->scratch[movi_index::kBases]);
            for (size_t i = 0; i < n && e == hipSuccess; i++) {
                const Chunk &c = chunks[i];
                if (c.nb) e = hipMemcpyAsync(all + (c.b0 - chunks[0].b0), h_bases + c.b0, c.nb, hipMemcpyHostToDevice, ix->pipe_up);
                if (e == hipSuccess) e = hipEventRecord(ix->pipe_ev[i], ix->pipe_up);
            }
            if (e != hipSuccess) {                            // copies may be in flight: nothing reads the caller's buffers when the call returns
                (void)hipStreamSynchronize(ix->pipe_up);
                return fail_hip(e, "uploading the reads");
            }
        } else {
            (void)hipGetLastError();                          // no room for the call-wide staging: chunk by chunk, as before
        }
    }
    for (size_t i = 0; i < n && rc == MOVI_OK; i++) {
        for (; next_up < n && next_up < i + (size_t)movi_index::kPipeAhead && rc == MOVI_OK; next_up++) rc = up(chunks[next_up], (int)(next_up % S), next_up);
        if (rc == MOVI_OK) rc = down((int)(i % S));
        if (rc == MOVI_OK) rc = finish_arrived();
    }
    for (size_t j = 0; j < (size_t)S && rc == MOVI_OK; j++) rc = finish((int)((n + j) % S));   // oldest first
    if (rc != MOVI_OK) {
        // whatever happened, nothing may still be reading or writing the caller's buffers when the call returns
        const std::string keep = g_err;
        if (ix->pipe_up) (void)hipStreamSynchronize(ix->pipe_up);
        for (int k = 0; k < S; k++)
            if (ix->pipe[k].s) (void)hipStreamSynchronize(ix->pipe[k].s);
        g_err = keep;
    }
    // (after the streams have drained: a stream's host function may hand work to the pool; also on errors: the workers write caller memory)
    for (int k = 0; k < S; k++) HostPool::get().wait(&grp[k]);
    if (stats) *stats = acc;
    if (rc == MOVI_OK && total / n_reads >= 2ull * (uint64_t)(ix->cfg.seg_len > 0 ? ix->cfg.seg_len : 1)) ix->seg_seen = acc.segments != 0;
    return rc;
}

// Queries with per-read results (count, bins) only have their upload to hide.  A read takes as long as it takes: with
// long reads the last chunk's walk starts when the last byte has arrived and lasts as long as the whole batch's would
// (100 k x 10 kbp, bins: 43 ms overlapped, 38 ms synchronous); with short reads the walks are short against the
// transfers and overlapping pays (1 M x 150 bp: count 20.6 -> 25.1, bins 21.8 -> 28.8 Gbases/s).
bool worth_overlapping_small_results(const uint64_t *h_offsets, uint64_t n_reads) {
    const uint64_t total = h_offsets[n_reads] - h_offsets[0];
    return total != 0 && total / n_reads <= 2048;
}

template <typename Launch, typename Fetch, typename Harvest>
int run_host(bool overlapped, movi_index *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
             uint8_t *h_read_err, movi_query_stats_t *stats, Launch launch, Fetch fetch, Harvest harvest,
             size_t small_bytes, size_t mask_word_bytes = 0, bool vector_down = false) {
    movi_query_stats_t local{};
    overlapped = overlapped && ix->host_overlap;
    int rc = overlapped ? run_pipelined(ix, h_bases, h_offsets, n_reads, h_read_err, &local, launch, fetch, harvest, small_bytes, mask_word_bytes, vector_down)
                        : run_chunked(ix, h_bases, h_offsets, n_reads, h_read_err, &local, launch, fetch, harvest, small_bytes, mask_word_bytes);
    if (stats) *stats = local;
    if (rc) return rc;
    if (local.errors)
        return fail(MOVI_ERR_INVARIANT, std::to_string(local.errors) +
                                            " read(s) hit a move-structure invariant violation (corrupt index?)");
    return MOVI_OK;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------- page-locked host memory

int movi_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(MOVI_ERR_ARG, "out is NULL");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return MOVI_OK;
}

int movi_host_free(void *p) {
    if (!p) return MOVI_OK;
    HIP_TRY(hipHostFree(p));
    return MOVI_OK;
}

int movi_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return fail(MOVI_ERR_ARG, "NULL or empty range");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return MOVI_OK;
}

int movi_host_unregister(void *p) {
    if (!p) return MOVI_OK;
    HIP_TRY(hipHostUnregister(p));
    return MOVI_OK;
}

// Page-locks a caller's pageable range for the duration of one big *_host call ("host_autopin"); releases it on scope exit.
struct AutoPin {
    void *p = nullptr;
    bool pin(void *q, size_t bytes) {                        // true: the range is page-locked now (by us or already)
        if (!q || !bytes) return false;
        // page-locked already only if BOTH ends are (a caller -- or another thread's call on an adjacent slice of the same
        // buffer -- may have registered part of the span); a partly registered span cannot be registered again: synchronous path
        const bool first = is_pinned(q), last = is_pinned(static_cast<char *>(q) + bytes - 1);
        if (first && last) return true;
        if (first || last) return false;
        if (hipHostRegister(q, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
        p = q;
        return true;
    }
    ~AutoPin() {
        if (!p) return;
        const std::string keep = g_err;
        (void)hipHostUnregister(p);
        (void)hipGetLastError();
        g_err = keep;
    }
};
static bool autopin_worthwhile(const movi_index *ix, const uint64_t *h_offsets, uint64_t n_reads) {
    // (the overlapped path cuts a call into pieces of >= 2^15 reads: it needs at least three of them to overlap anything)
    return ix->host_autopin && ix->host_overlap && h_offsets[n_reads] - h_offsets[0] >= (1ull << 27) && n_reads >= 3 * kPipeMinReads;
}

// h_mask_words != NULL: the reset masks themselves are the result (movi_pml_mask_host).  Otherwise, PML with "host_masks": the
// walk writes masks, only they cross PCIe (1/8 byte per base instead of 2) and the u16 vector is expanded into the caller's buffer
// by the host's worker threads (movi_expand_host.cpp) -- in the overlapped path beside the walks of the chunks that follow.
// A chunk's reset-mask words have arrived in its slot's page-locked block (host function on the chunk's stream: no HIP call in here).
struct DeliverArg {
    const uint32_t *words = nullptr;
    uint32_t *mask_dst = nullptr;     // movi_pml_mask_host: where the words belong in the caller's array
    size_t mask_bytes = 0;
    bool expand = false;              // movi_pml_host: the worker pool expands them into the caller's vector
    ExpandJob job;
    int threads = 0;
    HostPool::Group *group = nullptr;
};
static void deliver_on_stream(void *p) {
    DeliverArg *a = static_cast<DeliverArg *>(p);
    if (a->mask_dst) memcpy(a->mask_dst, a->words, a->mask_bytes);
    if (a->expand) expand_parallel(a->job, a->threads, a->group);
    delete a;
}

static int ml_host(bool zml, movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                  uint16_t *h_out_pml, uint8_t *h_read_err, movi_query_stats_t *stats, uint32_t *h_mask_words = nullptr) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MOVI_OK; }
    // h_out_pml == NULL: the walk runs, error bytes and counters come back, the vectors stay on the device and are dropped
    // (`movi query --no-output`: the reference computes and discards)
    if (!h_offsets || (h_offsets[n_reads] != h_offsets[0] && !h_bases)) return fail(MOVI_ERR_ARG, "NULL host buffer");
    if (int rc0 = check_offsets(h_offsets, n_reads)) return rc0;
    HIP_TRY(hipSetDevice(ix->device));
    const uint64_t o0 = h_offsets[0], span = h_offsets[n_reads] - o0;
    // The way down of the u16 vector, chunk by chunk: as it is (2 bytes per base over PCIe, no host work where the caller's vector is
    // page-locked) or as reset masks (1/8 byte per base) that the host's worker threads expand into the caller's vector -- the DMA engine
    // and the host's cores are two resources.  "host_masks": -1 = masks for calls of >= 2^22 bases (no page-locking of the vector; with the
    // words handed to the pool by a host function on the chunk's stream and sixteen chunks per call this way alone runs at 33 - 34 Gbases/s
    // on 1 M x 150 bp, the vector's at 22.7, the 1 B / base upload's floor is 36.7), 0 = never masks, 1 = masks for every call, 2 = both
    // ways side by side, `host_mask_share` percent of the bases as masks (31 - 34: no better than masks alone since the host function;
    // profiles/r06_host_path.txt).
    enum { kVector = 0, kMasks = 1, kMixed = 2 };
    int route = kVector;
    if (h_mask_words) route = kMasks;
    else if (!zml && h_out_pml) {
        if (ix->host_masks > 0) route = ix->host_masks >= 2 ? kMixed : kMasks;
        else if (ix->host_masks < 0 && span >= (1ull << 22)) route = kMasks;
    }
    // A chunk's way down: `masks` = reset masks + host expansion, else the vector by DMA; a mixed call deals its chunks by `host_mask_share`.
    // (Tried: walks that leave both on the device and a choice at download time by the pool's backlog -- 28.1 against 29.9 Gbases/s for
    // the plain deal; a call's tail cut into halving chunks -- 27.3 against 28.1: profiles/r06_host_path.txt.)
    struct Way { bool masks = false, delivered = false; uint32_t phase = 0; };
    uint64_t acc_all = 0, acc_mask = 0;                       // bases launched so far / of them by the mask route
    std::unordered_map<uint64_t, Way> way_of;                 // first read of a chunk -> its way (launch, fetch and harvest run on the calling thread)
    const int threads = ix->host_threads > 0 ? ix->host_threads : host_threads_default();
    struct { void *p; } d_out{};
    auto phase_of = [&](uint64_t b0) { return (uint32_t)((b0 - o0) & 31u); };
    auto launch = [&](ChunkCtx &c, const uint8_t *db, const uint64_t *dof, uint64_t nr, uint64_t nb, uint8_t *derr) -> int {
        Way w;
        w.phase = phase_of(c.b0);
        const bool through_masks = !zml && ix->pml_via_mask != 0 && pml_vector_via_masks(ix->dev, ix->cfg, nr, nb, true, false);
        if (route == kMasks) w.masks = true;
        else if (route == kMixed) {                           // (the first chunk comes down as it is: its DMA starts when its walk ends)
            w.masks = (acc_mask + nb) * 100 <= (uint64_t)ix->host_mask_share * (acc_all + nb);
            acc_all += nb;
            if (w.masks) acc_mask += nb;
        }
        way_of[c.first] = w;
        if (w.masks) {
            MaskArgs m;
            m.phase = w.phase;
            HIP_TRY(c.alloc(movi_index::kMask, (size_t)pml_mask_words(nr, nb, m.phase) * 4, &d_out.p));
            m.words = static_cast<uint32_t *>(d_out.p);
            return ml_device(false, ix, db, dof, nr, nb, nullptr, derr, nullptr, c.s, ClsArgs(), c.d_stats, c.seg_ws, c.ragged_hint,
                             c.seg_verdict, &m, &c.d[movi_index::kTmp], &c.cap[movi_index::kTmp]);
        }
        HIP_TRY(c.alloc(movi_index::kOut, nb * 2, &d_out.p));
        // (the vector itself comes down: on the device it is written through reset masks where movi_pml_device's policy says so)
        if (through_masks || (!zml && ix->pml_via_mask > 0)) {
            void *mw = nullptr;
            HIP_TRY(c.alloc(movi_index::kMask, (size_t)pml_mask_words(nr, nb, 0) * 4, &mw));
            MaskArgs m;
            m.words = static_cast<uint32_t *>(mw);
            m.expand_out = static_cast<uint16_t *>(d_out.p);
            return ml_device(false, ix, db, dof, nr, nb, nullptr, derr, nullptr, c.s, ClsArgs(), c.d_stats, c.seg_ws, c.ragged_hint,
                             c.seg_verdict, &m);
        }
        return ml_device(zml, ix, db, dof, nr, nb, static_cast<uint16_t *>(d_out.p), derr, nullptr, c.s, ClsArgs(), c.d_stats,
                         c.seg_ws, c.ragged_hint, c.seg_verdict);
    };
    // what a chunk's masks are to the host: the words land in `words`; they are the result, or the vector is expanded from them
    auto deliver = [&](const uint32_t *words, uint64_t first, uint64_t nr, uint32_t ph, HostPool::Group *g) {
        const uint64_t b0 = h_offsets[first], nb = h_offsets[first + nr] - b0;
        if (h_mask_words) memcpy(h_mask_words + ((b0 - o0) >> 5) + first, words, (size_t)(((nb + ph) >> 5) + nr) * 4);
        if (h_out_pml) {
            ExpandJob j;
            j.words = words; j.offs = h_offsets; j.o0 = b0; j.phase = ph; j.ibase = first; j.i0 = first; j.i1 = first + nr;
            j.out = h_out_pml;
            expand_parallel(j, threads, g);
        }
    };
    // (the results are found through the chunk's own staging: with chunks in flight, launch() of the next chunk has
    // run before fetch() of this one)
    std::vector<uint32_t> sync_words;                        // synchronous path: a chunk's words on their way through the host
    auto fetch = [&](ChunkCtx &c, uint64_t first, uint64_t nr, uint64_t b0, uint64_t nb) -> int {
        const Way w = way_of[first];
        if (w.masks) {
            const size_t nw = (size_t)(((nb + w.phase) >> 5) + nr);
            if (c.async) {
                // into the slot's page-locked block, and on to the worker pool by a host function on the chunk's stream -- the moment the
                // words have arrived, not when the calling thread's loop next comes by (with 8 chunks per call the pool ran at 2/3 duty)
                HIP_TRY(c.down_small(nullptr, 0, c.d[movi_index::kMask], nw * 4));
                DeliverArg *a = new DeliverArg;
                a->words = reinterpret_cast<const uint32_t *>(c.h_small);
                a->mask_dst = h_mask_words ? h_mask_words + ((b0 - o0) >> 5) + first : nullptr;
                a->mask_bytes = nw * 4;
                a->expand = h_out_pml != nullptr;
                a->job.words = a->words; a->job.offs = h_offsets; a->job.o0 = b0; a->job.phase = w.phase; a->job.ibase = first;
                a->job.i0 = first; a->job.i1 = first + nr; a->job.out = h_out_pml;
                a->threads = threads;
                a->group = c.grp;
                const hipError_t eh = hipLaunchHostFunc(c.s, deliver_on_stream, a);
                if (eh != hipSuccess) { delete a; (void)hipGetLastError(); return MOVI_OK; }   // harvest() does it instead
                way_of[first].delivered = true;
                return MOVI_OK;
            }
            if (h_mask_words && !h_out_pml) {                 // straight to where they belong
                HIP_TRY(hipMemcpy(h_mask_words + ((b0 - o0) >> 5) + first, c.d[movi_index::kMask], nw * 4, hipMemcpyDeviceToHost));
                return MOVI_OK;
            }
            sync_words.resize(nw);
            HIP_TRY(hipMemcpy(sync_words.data(), c.d[movi_index::kMask], nw * 4, hipMemcpyDeviceToHost));
            deliver(sync_words.data(), first, nr, w.phase, nullptr);
            return MOVI_OK;
        }
        if (h_out_pml) HIP_TRY(c.down(h_out_pml + b0, c.d[movi_index::kOut], nb * 2));
        return MOVI_OK;
    };
    auto harvest = [&](const uint8_t *h_small, uint64_t first, uint64_t nr, HostPool::Group *g) {
        const Way w = way_of[first];
        if (!w.masks || w.delivered) return;
        deliver(reinterpret_cast<const uint32_t *>(h_small), first, nr, w.phase, g);
    };
    // masks: only the reads have to be page-locked for the overlapped path (the vector is written by host threads)
    bool overlapped = span != 0 && is_pinned(h_bases) && (route == kMasks || !h_out_pml || is_pinned(h_out_pml));
    // A big call on PAGEABLE buffers (what a std::vector-holding caller passes: INTEGRATION.md's stub): page-lock them for the
    // duration of the call and take the overlapped path.  Registering touched memory runs at hundreds of GB/s (DESIGN.md section
    // 5), so on >= 2^27 bases it is paid back several times over (the synchronous path: 12.4 Gbases/s PCIe-inclusive).
    // "host_autopin" 0 turns it off; anything that fails here falls back to the synchronous path.
    AutoPin pin_bases, pin_out;
    if (!overlapped && autopin_worthwhile(ix, h_offsets, n_reads))
        overlapped = pin_bases.pin(const_cast<uint8_t *>(h_bases) + h_offsets[0], span) &&
                     (route == kMasks || !h_out_pml || pin_out.pin(h_out_pml + h_offsets[0], span * 2));
    if (route == kMixed && !overlapped) route = kMasks;       // (the synchronous path: one way down for the whole call)
    // (vector_down: 2 bytes per base come down by DMA -- the way down bounds such a call, and with every read going up at once beside
    // it the downloads ran 8 % slower: those calls feed their uploads from the loop)
    return run_host(overlapped, ix, h_bases, h_offsets, n_reads, h_read_err, stats, launch, fetch, harvest, 0, route != kVector ? 4 : 0,
                    route != kMasks && h_out_pml != nullptr);
}

int movi_pml_mask_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                       uint32_t *h_mask_words, uint8_t *h_read_err, movi_query_stats_t *stats) {
    if (n_reads && !h_mask_words) return fail(MOVI_ERR_ARG, "NULL host buffer");
    return ml_host(false, ix, h_bases, h_offsets, n_reads, nullptr, h_read_err, stats, h_mask_words);
}

// Pure host code: no device is touched.
int movi_pml_expand_host(const uint32_t *h_mask_words, const uint64_t *h_offsets, uint64_t n_reads, uint16_t *h_out_pml,
                         int n_threads) {
    if (n_reads == 0) return MOVI_OK;
    if (!h_mask_words || !h_offsets || (h_offsets[n_reads] != h_offsets[0] && !h_out_pml)) return fail(MOVI_ERR_ARG, "NULL host buffer");
    if (int rc0 = check_offsets(h_offsets, n_reads)) return rc0;
    ExpandJob j;
    j.words = h_mask_words; j.offs = h_offsets; j.o0 = h_offsets[0]; j.phase = 0; j.ibase = 0; j.i0 = 0; j.i1 = n_reads;
    j.out = h_out_pml;
    expand_parallel(j, n_threads, nullptr);
    return MOVI_OK;
}

int movi_pml_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                  uint16_t *h_out_pml, uint8_t *h_read_err, movi_query_stats_t *stats) {
    return ml_host(false, ix, h_bases, h_offsets, n_reads, h_out_pml, h_read_err, stats);
}

// `movi query --logs`: PMLs plus, per base, the fast-forwards and scan rows MoveQuery::add_fastforward / add_scan
// collect (ClsArgs::log_ff / log_scan).  Synchronous path, first kernel.
int movi_pml_logs_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                       uint16_t *h_out_pml, uint16_t *h_fastforwards, uint16_t *h_scans, uint8_t *h_read_err,
                       movi_query_stats_t *stats) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MOVI_OK; }
    if (!h_offsets || !h_fastforwards || !h_scans || (h_offsets[n_reads] != h_offsets[0] && (!h_bases || !h_out_pml)))
        return fail(MOVI_ERR_ARG, "NULL host buffer");
    if (int rc0 = check_offsets(h_offsets, n_reads)) return rc0;
    HIP_TRY(hipSetDevice(ix->device));
    struct { void *p; } d_out{}, d_ff{}, d_sc{};
    auto launch = [&](ChunkCtx &c, const uint8_t *db, const uint64_t *dof, uint64_t nr, uint64_t nb, uint8_t *derr) -> int {
        HIP_TRY(c.alloc(movi_index::kOut, nb * 2, &d_out.p));
        HIP_TRY(c.alloc(movi_index::kA, nb * 2, &d_ff.p));
        HIP_TRY(c.alloc(movi_index::kS, nb * 2, &d_sc.p));
        HIP_TRY(hipMemsetAsync(d_ff.p, 0, nb * 2, c.s));               // a read of one base has no LF: its entry stays 0
        HIP_TRY(hipMemsetAsync(d_sc.p, 0, nb * 2, c.s));
        ClsArgs logs;
        logs.log_ff = static_cast<uint16_t *>(d_ff.p);
        logs.log_scan = static_cast<uint16_t *>(d_sc.p);
        return ml_device(false, ix, db, dof, nr, nb, static_cast<uint16_t *>(d_out.p), derr, nullptr, c.s, logs, c.d_stats);
    };
    auto fetch = [&](ChunkCtx &c, uint64_t, uint64_t, uint64_t b0, uint64_t nb) -> int {
        HIP_TRY(c.down(h_out_pml + b0, c.d[movi_index::kOut], nb * 2));
        HIP_TRY(c.down(h_fastforwards + b0, c.d[movi_index::kA], nb * 2));
        HIP_TRY(c.down(h_scans + b0, c.d[movi_index::kS], nb * 2));
        return MOVI_OK;
    };
    auto harvest = [](const uint8_t *, uint64_t, uint64_t, HostPool::Group *) {};
    return run_host(false, ix, h_bases, h_offsets, n_reads, h_read_err, stats, launch, fetch, harvest, 0);
}

int movi_zml_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                  uint16_t *h_out_zml, uint8_t *h_read_err, movi_query_stats_t *stats) {
    return ml_host(true, ix, h_bases, h_offsets, n_reads, h_out_zml, h_read_err, stats);
}

// ----------------------------------------------------------------- classification

int movi_classify_device(movi_index_t *ix, const uint16_t *d_pml, const uint64_t *d_offsets, uint64_t n_reads,
                         uint32_t bin_width, uint32_t max_value_thr, uint32_t *d_bins_above, uint32_t *d_bins_below,
                         uint64_t *d_sum_max, void *stream) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) return MOVI_OK;
    if (!d_pml || !d_offsets || !d_bins_above || !d_bins_below || !d_sum_max) return fail(MOVI_ERR_ARG, "NULL device buffer");
    if (bin_width == 0) return fail(MOVI_ERR_ARG, "bin_width must be > 0");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_classify(d_pml, d_offsets, n_reads, bin_width, max_value_thr, d_bins_above, d_bins_below, d_sum_max,
                            static_cast<hipStream_t>(stream)));
    return MOVI_OK;
}

constexpr uint64_t kClassifyTwoPassLen = 1024;          // mean read length from which vector + bins run as walk + pass (short reads: fused 71.5 against 63.5 on c2)
static int pml_classify_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                               uint64_t n_bases, uint32_t bin_width, uint32_t max_value_thr, uint16_t *d_out_pml,
                               uint32_t *d_bins_above, uint32_t *d_bins_below, uint64_t *d_sum_max, uint8_t *d_read_err,
                               const uint32_t *d_read_order, void *stream, DevStats *d_stats,
                               SegWorkspace *seg_ws = nullptr, int ragged_hint = -1, int *seg_verdict = nullptr) {
    if (bin_width == 0) return fail(MOVI_ERR_ARG, "bin_width must be > 0");
    if (n_reads && (!d_bins_above || !d_bins_below || !d_sum_max)) return fail(MOVI_ERR_ARG, "NULL device buffer");
    ClsArgs cls;
    cls.bin_width = bin_width;
    cls.thr = max_value_thr;
    cls.above = d_bins_above;
    cls.below = d_bins_below;
    cls.sum_max = d_sum_max;
    // When the PML vector is wanted anyway, the bins are cheaper as a second, streaming pass over the resident vectors
    // (classify_kernel: 2 B per base at HBM speed) than fused into the walk: the running bins cost the latency-bound walk
    // ~20 instructions per emission and eight registers -- 100 k x 10 kbp: fused 46.9, walk + pass 53.6 Gbases/s
    // (profiles/r04_classify.txt).  Bins WITHOUT the vector (--classify --filter, --no-output) stay fused: nothing is
    // written at all (60.7).  "classify_fused" 1 / 0 forces either; the pass needs the error bytes (failed reads report
    // no bins), so without d_read_err the fused kernel runs.
    const bool two_pass = d_out_pml != nullptr && d_read_err != nullptr && n_reads != 0 &&
                          (ix ? (ix->cfg.classify_fused == 0 || (ix->cfg.classify_fused < 0 && n_bases / n_reads >= kClassifyTwoPassLen)) : false);
    if (two_pass) {
        const int rc = ml_device(false, ix, d_bases, d_offsets, n_reads, n_bases, d_out_pml, d_read_err, d_read_order, stream, ClsArgs(),
                                 d_stats, seg_ws, ragged_hint, seg_verdict);
        if (rc != MOVI_OK) return rc;
        HIP_TRY(launch_classify(d_out_pml, d_offsets, n_reads, bin_width, max_value_thr, d_bins_above, d_bins_below, d_sum_max,
                                static_cast<hipStream_t>(stream), d_read_err, n_bases));
        return MOVI_OK;
    }
    // (a chunk in flight of the overlapped host path brings its own segment workspace, length hint and the call's probe
    // verdict: without them every chunk fell back to the handle's workspace -- shared by chunks on different streams --
    // and probed again)
    return ml_device(false, ix, d_bases, d_offsets, n_reads, n_bases, d_out_pml, d_read_err, d_read_order, stream, cls, d_stats,
                     seg_ws, ragged_hint, seg_verdict);
}

int movi_pml_classify_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                             uint64_t n_bases, uint32_t bin_width, uint32_t max_value_thr, uint16_t *d_out_pml,
                             uint32_t *d_bins_above, uint32_t *d_bins_below, uint64_t *d_sum_max, uint8_t *d_read_err,
                             const uint32_t *d_read_order, void *stream) {
    return pml_classify_device(ix, d_bases, d_offsets, n_reads, n_bases, bin_width, max_value_thr, d_out_pml, d_bins_above,
                               d_bins_below, d_sum_max, d_read_err, d_read_order, stream, nullptr);
}

int movi_pml_classify_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                           uint32_t bin_width, uint32_t max_value_thr, uint32_t *h_bins_above, uint32_t *h_bins_below,
                           uint64_t *h_sum_max, uint8_t *h_read_err, movi_query_stats_t *stats) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MOVI_OK; }
    if (!h_offsets || !h_bins_above || !h_bins_below || !h_sum_max || (h_offsets[n_reads] != h_offsets[0] && !h_bases))
        return fail(MOVI_ERR_ARG, "NULL host buffer");
    if (int rc0 = check_offsets(h_offsets, n_reads)) return rc0;
    if (bin_width == 0) return fail(MOVI_ERR_ARG, "bin_width must be > 0");
    HIP_TRY(hipSetDevice(ix->device));
    struct { void *p; } d_a{}, d_b{}, d_s{};
    auto launch = [&](ChunkCtx &c, const uint8_t *db, const uint64_t *dof, uint64_t nr, uint64_t nb, uint8_t *derr) -> int {
        HIP_TRY(c.alloc(movi_index::kA, nr * 4, &d_a.p));
        HIP_TRY(c.alloc(movi_index::kB, nr * 4, &d_b.p));
        HIP_TRY(c.alloc(movi_index::kS, nr * 8, &d_s.p));
        // bins reduced inside the PML kernel; no PML vector is written at all
        return pml_classify_device(ix, db, dof, nr, nb, bin_width, max_value_thr, nullptr, static_cast<uint32_t *>(d_a.p),
                                   static_cast<uint32_t *>(d_b.p), static_cast<uint64_t *>(d_s.p), derr, nullptr, c.s,
                                   c.d_stats, c.seg_ws, c.ragged_hint, c.seg_verdict);
    };
    // page-locked block of a chunk in flight: sum_max[nr] | above[nr] | below[nr]
    auto fetch = [&](ChunkCtx &c, uint64_t first, uint64_t nr, uint64_t, uint64_t) -> int {
        HIP_TRY(c.down_small(h_sum_max + first, 0, c.d[movi_index::kS], nr * 8));
        HIP_TRY(c.down_small(h_bins_above + first, nr * 8, c.d[movi_index::kA], nr * 4));
        HIP_TRY(c.down_small(h_bins_below + first, nr * 12, c.d[movi_index::kB], nr * 4));
        return MOVI_OK;
    };
    auto harvest = [&](const uint8_t *h, uint64_t first, uint64_t nr, HostPool::Group *) {
        memcpy(h_sum_max + first, h, nr * 8);
        memcpy(h_bins_above + first, h + nr * 8, nr * 4);
        memcpy(h_bins_below + first, h + nr * 12, nr * 4);
    };
    bool overlapped = worth_overlapping_small_results(h_offsets, n_reads) && is_pinned(h_bases);
    AutoPin pin_bases;
    if (!overlapped && worth_overlapping_small_results(h_offsets, n_reads) && autopin_worthwhile(ix, h_offsets, n_reads))
        overlapped = pin_bases.pin(const_cast<uint8_t *>(h_bases) + h_offsets[0], h_offsets[n_reads] - h_offsets[0]);
    return run_host(overlapped, ix, h_bases, h_offsets, n_reads, h_read_err, stats, launch, fetch, harvest, 16);
}

// -------------------------------------------------------------------------- count

static int ensure_ckpt(movi_index *ix, hipStream_t s) {
    if (ix->d_ckpt) return MOVI_OK;
    const uint64_t n_chunks = (ix->desc.r + (1ull << kPrefixShift) - 1) >> kPrefixShift;
    HIP_TRY(hipMalloc(&ix->d_ckpt, (n_chunks + 1) * sizeof(uint64_t)));
    hipError_t e = build_row_start_ckpt(ix->kmode, ix->d_rows, ix->desc.r, ix->d_ckpt, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);        // other streams (the overlapped host path) may use it next
    if (e != hipSuccess) {
        (void)hipFree(ix->d_ckpt);
        ix->d_ckpt = nullptr;
        return fail_hip(e, "building the row-start checkpoints");
    }
    ix->dev.row_start_ckpt = ix->d_ckpt;
    return MOVI_OK;
}

// The derived tables of the count query -- row-start checkpoints, interval table, and the look-ahead rows where the table's own
// statistic says the search will use them --: built by movi_index_prepare, or by the first count query on the handle.
static int ensure_count_tables(movi_index *ix, hipStream_t s, bool from_prepare = false) {
    int rc = ensure_ckpt(ix, s);
    if (rc) return rc;
    if (ix->ftab_auto > 0 && !ix->d_ftab && ftab_eligible(ix)) {       // the interval table (256 MB at K = 12)
        if (build_ftab_table(ix, (uint32_t)ix->ftab_auto, s) != MOVI_OK) { (void)hipGetLastError(); ix->ftab_auto = 0; }
    }
    // (round 5: the count query's default is the lane state machine on the PLAIN rows, launch_count; the look-ahead copy serves
    // count_kernel_v0 only, i.e. "count_variant" 0)
    if (ix->cfg.count_variant == 0 && ix->ahead_auto > 0 && !ix->d_rows2 && !ix->count_declined_ahead && ahead_eligible(ix) &&
        (from_prepare || !ix->prepared)) {
        // sampled first, so that a table that will not use the copy is not copied (16 B per row) to find out.  Only the table's
        // statistic declines for good; a device short of memory is asked again later (ahead_retry_in).
        sample_no_ff(ix, s);
        if (!ix->ahead_tallied) { /* the sample failed: the next call tries again */ }
        else if (ix->ahead_no_ff < kAheadCountRatio) ix->count_declined_ahead = true;
        else if (ix->ahead_retry_in > 0) ix->ahead_retry_in -= 1;
        else if (!ahead_wanted(ix)) ix->ahead_retry_in = 63;
        else if (build_ahead(ix, s, true) != MOVI_OK) { (void)hipGetLastError(); ix->ahead_auto = 0; }
        else if (ix->dev.rows2_count == 0u) {
            // (the copy's own tally -- every row, not a sample -- says the count query is better off on the plain rows: the copy
            // is not kept for a caller who may never ask for PMLs; the first PML query builds it again, 85 us per 14 M rows)
            ix->dev.rows2 = nullptr;
            ix->dev.rows2_tail = 0;
            ix->dev.hints = 0;
            (void)hipFree(ix->d_rows2);
            ix->d_rows2 = nullptr;
            ix->count_declined_ahead = true;
        }
    }
    return MOVI_OK;
}

static int count_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                        uint64_t n_bases, uint64_t *d_matched, uint64_t *d_count, uint8_t *d_read_err,
                        const uint32_t *d_read_order, void *stream, DevStats *d_stats) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) return MOVI_OK;
    if (!d_offsets || !d_matched || !d_count || (n_bases && !d_bases)) return fail(MOVI_ERR_ARG, "NULL device buffer");
    if (!d_stats) d_stats = ix->d_stats;
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = ensure_count_tables(ix, s);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(DevStats), s));
    if (n_reads > 0xFFFFFFFFull) return fail(MOVI_ERR_ARG, "more than 2^32 reads in one call");
    HIP_TRY(launch_count(ix->kmode, ix->dev, d_bases, d_offsets, n_reads, d_matched, d_count,
                         d_read_err, d_stats, d_read_order, ix->cfg, s, &ix->last_launch, n_bases));
    return MOVI_OK;
}

int movi_count_device(movi_index_t *ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                      uint64_t n_bases, uint64_t *d_matched, uint64_t *d_count, uint8_t *d_read_err,
                      const uint32_t *d_read_order, void *stream) {
    return count_device(ix, d_bases, d_offsets, n_reads, n_bases, d_matched, d_count, d_read_err, d_read_order, stream,
                        nullptr);
}

int movi_index_prepare(movi_index_t *ix, uint32_t what, void *stream, uint64_t *derived_bytes) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (what & ~(uint32_t)(MOVI_PREPARE_PML | MOVI_PREPARE_COUNT | MOVI_PREPARE_ZML)) return fail(MOVI_ERR_ARG, "unknown MOVI_PREPARE_* bit");
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    ix->ahead_retry_in = 0;                                           // an explicit call asks the device now
    if ((what & MOVI_PREPARE_PML) && mode_has_thresholds(ix->desc.mode)) {
        ensure_pml_tables(ix, s, true);
        // the walk kernels live in translation units of their own: their code objects are loaded here, not by the first walk
        (void)(ix->dev.idx32 ? preload_walk_u32() : preload_walk_u64());
        (void)(ix->dev.idx32 ? preload_walkseg_u32() : preload_walkseg_u64());
        (void)hipGetLastError();
    }
    if (what & MOVI_PREPARE_COUNT) {
        const int rc = ensure_count_tables(ix, s, true);
        if (rc) return rc;
    }
    ix->prepared = true;
    // (MOVI_PREPARE_ZML: the parse walks on the plain rows and derives nothing -- accepted so that callers need not know)
    HIP_TRY(hipStreamSynchronize(s));
    if (derived_bytes) {
        double v = 0.0;
        (void)movi_index_info(ix, "derived_bytes", &v);
        *derived_bytes = (uint64_t)v;
    }
    return MOVI_OK;
}

int movi_count_host(movi_index_t *ix, const uint8_t *h_bases, const uint64_t *h_offsets, uint64_t n_reads,
                    uint64_t *h_matched, uint64_t *h_count, uint8_t *h_read_err, movi_query_stats_t *stats) {
    if (!ix) return fail(MOVI_ERR_ARG, "index handle is NULL");
    if (n_reads == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MOVI_OK; }
    if (!h_offsets || !h_matched || !h_count || (h_offsets[n_reads] != h_offsets[0] && !h_bases))
        return fail(MOVI_ERR_ARG, "NULL host buffer");
    if (int rc0 = check_offsets(h_offsets, n_reads)) return rc0;
    HIP_TRY(hipSetDevice(ix->device));
    struct { void *p; } d_m{}, d_c{};
    auto launch = [&](ChunkCtx &c, const uint8_t *db, const uint64_t *dof, uint64_t nr, uint64_t nb, uint8_t *derr) -> int {
        HIP_TRY(c.alloc(movi_index::kA, nr * 8, &d_m.p));
        HIP_TRY(c.alloc(movi_index::kS, nr * 8, &d_c.p));
        return count_device(ix, db, dof, nr, nb, static_cast<uint64_t *>(d_m.p), static_cast<uint64_t *>(d_c.p), derr,
                            nullptr, c.s, c.d_stats);
    };
    // page-locked block of a chunk in flight: matched[nr] | count[nr]
    auto fetch = [&](ChunkCtx &c, uint64_t first, uint64_t nr, uint64_t, uint64_t) -> int {
        HIP_TRY(c.down_small(h_matched + first, 0, c.d[movi_index::kA], nr * 8));
        HIP_TRY(c.down_small(h_count + first, nr * 8, c.d[movi_index::kS], nr * 8));
        return MOVI_OK;
    };
    auto harvest = [&](const uint8_t *h, uint64_t first, uint64_t nr, HostPool::Group *) {
        memcpy(h_matched + first, h, nr * 8);
        memcpy(h_count + first, h + nr * 8, nr * 8);
    };
    bool overlapped = worth_overlapping_small_results(h_offsets, n_reads) && is_pinned(h_bases);
    AutoPin pin_bases;
    if (!overlapped && worth_overlapping_small_results(h_offsets, n_reads) && autopin_worthwhile(ix, h_offsets, n_reads))
        overlapped = pin_bases.pin(const_cast<uint8_t *>(h_bases) + h_offsets[0], h_offsets[n_reads] - h_offsets[0]);
    return run_host(overlapped, ix, h_bases, h_offsets, n_reads, h_read_err, stats, launch, fetch, harvest, 16);
}

}  // extern "C"
