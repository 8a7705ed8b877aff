// movi_kernels.hip -- gfx950 (CDNA4, wave64) kernels for the PML / count walk.
//
// What is computed (reference file:line under /root/reference):
//   per read, from the last base to the first, one step per base
//     LF_move        src/move_structure.cpp:59-87     idx = id(row); off += offset(row)
//     fast_forward   src/move_structure.cpp:524-545   while off >= n(idx): off -= n(idx); idx++
//     match / reposition_thresholds / reposition_up|down
//                    src/read_processor.cpp:188-238, src/move_structure_query.cpp:188-232,513-601
//     add_ml         include/move_query.hpp:26-38     u16 clamp, emitted last base first
//   count: update_interval src/move_structure_search.cpp:48-61, two LF_moves per base,
//     MoveInterval::count include/move_intervals.hpp:47-58 (O(1) here via row-start checkpoints).
//
// Shape of the work: integer pointer chasing.  One wavefront lane owns one read;
// every step is one dependent random 8-byte (mode 6) / 6-byte (mode 8) row gather
// plus a few sequential neighbour rows.  No MFMA: there is no contraction here.
#include "movi_device.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace movi {

// ------------------------------------------------------------------------- PML
// One lane per read; wave-uniform step loop, predicated per lane.
//   VARIANT 0: one byte load and one u16 store per step and lane.
//   VARIANT 1: packed I/O -- each lane fetches its read 8 bases at a time (one 8-byte load
//              per 8 steps) and emits PMLs 8 at a time (one 16-byte store per 8 steps), so the
//              per-step traffic to L2 is the row gather alone.
template <int MODE, int VARIANT, int CLS>
__global__ __launch_bounds__(256) void pml_kernel(DevIndex ix, const uint8_t *__restrict__ bases,
                                                     const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                     uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                     DevStats *stats, const uint32_t *__restrict__ order,
                                                     ClsArgs cls) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, repo_total = 0, failed = 0;
    const EndThr ethr = end_thresholds(ix);
    const bool valid = t < n_reads;
    // lane slot t works on read rid: the host may pass reads sorted by length so that the 64
    // lanes of a wave finish together (ragged batches)
    const uint64_t rid = (valid && order) ? order[t] : t;
    const uint64_t beg = valid ? offs[rid] : 0;
    const uint64_t len = valid ? offs[rid + 1] - beg : 0;
    const uint8_t *R = bases + beg;
    uint16_t *O = out + beg;
    // ReadProcessor::reset_process, src/read_processor.cpp:69-70
    uint64_t idx = ix.r - 1;
    uint2 row = load_row<MODE>(ix.rows, idx);
    uint32_t off = row_n<MODE>(row) - 1;
    uint32_t ml = 0;
    uint64_t rb = 0, rb_next = 0;                         // VARIANT 1: 8 bases, byte 7 = current step
    uint32_t have16 = 0;
    uint4 pk = make_uint4(0, 0, 0, 0), pk_old = pk;       // VARIANT 1: last 8 PMLs, oldest in the low bits
    ClsState cs;
    if (CLS) cs.init((uint32_t)len, cls.bin_width);
    const uint64_t packed_end = len & ~7ull;              // steps >= this are stored one by one
    for (uint64_t k = 0; wave_any(k < len && failed == 0u); ++k) {
        bool live = k < len && failed == 0u;
        if (VARIANT >= 1 && (k & 7) == 0) {
            // 16 bases per fetch when they exist (one L2 request per 16 steps), else 8, else bytes
            if ((k & 8) == 0 && live && k + 16 <= len) {
                uint64_t two[2];
                __builtin_memcpy(two, R + (len - 16 - k), 16);        // unaligned 16-byte load
                rb = two[1];
                rb_next = two[0];
                have16 = 1;
            } else if ((k & 8) != 0 && have16) {
                rb = rb_next;
                have16 = 0;
            } else if (live && k + 8 <= len) {
                __builtin_memcpy(&rb, R + (len - 8 - k), 8);          // unaligned 8-byte load
            } else if (live) {
                rb = 0;
                for (uint64_t i = 0; i < len - k; ++i) rb |= (uint64_t)R[len - 1 - k - i] << (8 * (7 - i));
            }
        }
        if (k != 0) {
            const uint32_t ff_before = ff_total;
            const uint32_t e = lf_step<MODE>(ix, live, idx, off, row, ff_total);
            if (VARIANT == 0 && cls.log_ff && live) {     // --logs: this LF's fast-forwards (entry k - 1; the last one twice)
                cls.log_ff[beg + k - 1] = (uint16_t)(ff_total - ff_before);
                if (k + 1 == len) cls.log_ff[beg + k] = (uint16_t)(ff_total - ff_before);
            }
            if (e) { failed = e; live = false; }
        }
        const uint32_t scan_before = scan_total;
        uint32_t a = 0xFFu;
        if (live) {
            if (VARIANT >= 1) a = s_code[(uint32_t)(rb >> (8 * (7 - (k & 7)))) & 0xFFu];
            else a = s_code[R[len - 1 - k]];
        }
        const uint32_t rc = row_c<MODE>(row);             // the '$' row decodes as c == 0
        // 0 = no scan, 1 = scanning down, 2 = scanning up
        uint32_t dir = 0;
        if (live) {
            if (a == 0xFFu) {
                ml = 0;                                   // check_alphabet failed
            } else if (rc == a) {
                ml += 1;
            } else {
                // reposition_thresholds, src/move_structure_query.cpp:513-601
                repo_total += 1;
                ml = 0;
                uint32_t down;
                if (idx == ix.end_bwt_idx) {
                    // end_bwt_idx_thresholds (without separators the '$' row matches 'A', so a is 1..3 here)
                    down = (off >= end_threshold(ix.sep, ethr, a)) ? 1u : 0u;
                } else if (ix.sep && rc == 0u) {
                    down = (off >= separator_threshold(ix, idx, a)) ? 1u : 0u;   // a row of the separator
                } else {
                    const uint32_t kk = thr_slot(ix.sep, a, rc);     // alphamap_3, utils.cpp:5-8
                    const uint32_t thr = row_thr<MODE>(row, kk > 2u ? 2u : kk) ? row_n<MODE>(row) : 0u;
                    down = (off >= thr) ? 1u : 0u;
                }
                dir = down ? 1u : 2u;
                // reposition_down / reposition_up return r (not found) at the table ends
                if (down && idx == ix.r - 1) { failed = kErrNoRunBelow; dir = 0; live = false; }
                if (!down && idx == 0) { failed = kErrNoRunAbove; dir = 0; live = false; }
            }
        }
        // reposition_down :211-232 / reposition_up :188-209 as one uniform loop, one row per trip
        uint32_t scanning = dir;
        while (wave_any(scanning != 0u)) {
            if (scanning) {
                uint64_t jj = (scanning == 1u) ? idx + 1 : idx - 1;
                if (scanning == 1u) { if (jj >= ix.r) jj = ix.r - 1; }
                else if (jj > idx) jj = 0;                          // wrapped below row 0
                const uint2 w = load_row<MODE>(ix.rows, jj);
                scan_total += 1;
                idx = (scanning == 1u) ? idx + 1 : idx - 1;
                row = w;
                const uint32_t c = row_c<MODE>(row);
                if (c == a) {
                    scanning = 0;
                } else if (scanning == 1u ? (idx >= ix.r - 1) : (idx == 0)) {
                    failed = scanning == 1u ? kErrNoRunBelow : kErrNoRunAbove;   // :582-598
                    scanning = 0;
                    live = false;
                }
            }
        }
        if (dir == 1u) off = 0;
        if (dir == 2u) off = row_n<MODE>(row) - 1;        // read_processor.cpp:223
        if (VARIANT == 0 && cls.log_scan && k < len) cls.log_scan[beg + k] = (uint16_t)(scan_total - scan_before);
        const uint32_t val = ml > 65535u ? 65535u : ml;   // MoveQuery::add_ml
        if (CLS && live) cs.add(val, (uint32_t)k, (uint32_t)len, cls.bin_width, cls.thr);
        if (CLS == 2) {
            // verdict bins only: the PML vector is never written
        } else if (VARIANT >= 1) {
            if (live && k >= packed_end) {
                O[k] = (uint16_t)val;
            } else if (live) {
                pk.x = (pk.x >> 16) | (pk.y << 16);
                pk.y = (pk.y >> 16) | (pk.z << 16);
                pk.z = (pk.z >> 16) | (pk.w << 16);
                pk.w = (pk.w >> 16) | (val << 16);
                // 16 PMLs leave together as two adjacent 16-byte stores (one 32-byte span per
                // 16 steps); an odd group of 8 before the tail goes out on its own
                if ((k & 15) == 7) {
                    if (k + 8 < packed_end) pk_old = pk;
                    else __builtin_memcpy(O + (k - 7), &pk, 16);
                } else if ((k & 15) == 15) {
                    __builtin_memcpy(O + (k - 15), &pk_old, 16);      // unaligned 16-byte stores
                    __builtin_memcpy(O + (k - 7), &pk, 16);
                }
            }
        } else if (live) {
            O[k] = (uint16_t)val;
        }
    }
    // a read that broke an invariant reports all-zero PMLs plus its error code (the
    // reference aborts the whole run there; the host turns the flag into exit code 1)
    if (failed && CLS != 2) {
        for (uint64_t k = 0; k < len; ++k) O[k] = 0;
    }
    if (CLS && valid) cs.store(cls, rid, failed != 0u);
    if (valid && err) err[rid] = (uint8_t)failed;
    // one atomic per wave per counter
    const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), rpw = wave_sum(repo_total),
                   erw = wave_sum(failed ? 1u : 0u);
    if ((threadIdx.x & 63) == 0 && stats) {
        if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
        if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
        if (rpw) atomicAdd(&stats->repositions, (unsigned long long)rpw);
        if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
    }
}


// Top-of-walk table (DevIndex::kmer): lane t walks the K-mer whose step-i base has code (t >> 2 i) & 3 (+ 1 on a
// separators index) from the state every read starts in -- exactly pml_kernel<MODE, 0>'s automaton -- and records where
// it stands after the LF towards base K.
template <int MODE>
__global__ __launch_bounds__(256) void kmer_table_kernel(DevIndex ix, uint32_t K, uint4 *__restrict__ table) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n = 1ull << (2 * K);
    const bool valid = t < n;
    const EndThr ethr = end_thresholds(ix);
    const uint64_t r1 = ix.r - 1;
    uint2 row = load_row<MODE>(ix.rows, r1);
    uint64_t idx = r1;
    uint32_t off = row_n<MODE>(row) - 1, ml = 0, ff = 0, scan = 0, repo = 0, failed = 0, mask = 0;
    for (uint32_t i = 0; i < K; ++i) {                    // K is wave-uniform
        const bool live = valid && failed == 0u;
        const uint32_t a = (uint32_t)((t >> (2 * i)) & 3u) + ix.sep;
        const uint32_t before = ml;
        const uint32_t e = walk_base<MODE>(ix, ethr, live, i != 0, a, idx, off, row, ml, ff, scan, repo);
        if (e) failed = e;
        mask |= (uint32_t)(live && e == 0u && ml == before + 1u) << i;
    }
    if (!valid) return;
    const uint64_t j = row_id<MODE>(row, idx, ix);        // LF_move of base K - 1: destination and offset, no fast-forward yet
    const uint32_t off2 = off + row_off<MODE>(row);
    const uint32_t ok = (uint32_t)(failed == 0u && j < ix.r && off2 < 4096u);
    uint4 e4;
    e4.x = (uint32_t)j;
    e4.y = (uint32_t)(j >> 32) | (off2 << 4) | (mask << 16) | (ok << 31);
    e4.z = ff;
    e4.w = scan;
    if (!ok) e4 = make_uint4(0, 0, 0, 0);
    table[t] = e4;
}

// The builders' tallies go to kTallySlots pairs of counters, a block to the pair of its index (the host adds them up): one
// pair for all took the look-ahead rows of a 1 B-row table 375 ms -- 31 M atomics on one address, ~12 ns each -- instead of
// the 6.8 ms its 16 GB of writes take (measured after the change: gpurun_out -> profiles/r04_c4_pair_shared_pmc.txt header).
__device__ __forceinline__ void tally_add(unsigned long long *tally, uint32_t a, uint32_t b) {
    unsigned long long *t = tally + 2u * (blockIdx.x & (kTallySlots - 1u));
    atomicAdd(t, (unsigned long long)a);
    atomicAdd(t + 1, (unsigned long long)b);
}
// Look-ahead rows (DevIndex::rows2): thread i copies row i into its line and writes the entry of its LF target next to it.
// tally (optional): [0] += the positions of row i that arrive at its LF target below the target's length (no fast-forward
// there), [1] += n(i): their ratio says how often a walk that follows the text can use an entry -- 0.83 on pangenome BWTs,
// 0.51 on uniformly random run sequences (tools/lf_chain_stats.py).
// What the walk would read at row j and where it goes from there (j2 = id(j)): one 8-byte half of an entry.
__device__ __forceinline__ uint2 ahead_half(uint2 rj, uint64_t j2) {
    return make_uint2((uint32_t)j2, row_n<6>(rj) | (row_off<6>(rj) << 11) | (row_c<6>(rj) << 22) | ((uint32_t)(j2 >> 32) << 25) | 0x80000000u);
}
// Round 5: HINTS = 1 (tables of fewer than 2^32 - 1 rows: DevIndex::hints has the layout) -- the row's copy and its entry also carry
// the row's three reposition hints, and ids are stored 32 bits wide (an id that is not a row as 0xFFFFFFFF).
template <int MODE, int HINTS>
__global__ __launch_bounds__(256) void ahead_rows_kernel(DevIndex ix, uint8_t *__restrict__ out, uint64_t tail, unsigned long long *tally) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < ix.r;
    uint2 row = in ? load_row<MODE>(ix.rows, i) : make_uint2(0u, 0u);
    const uint64_t j = in ? row_id<MODE>(row, i, ix) : ix.r;
    uint2 e = make_uint2(0u, 0u);
    uint32_t no_ff = 0;
    if (j < ix.r) {
        const uint2 rj = load_row<MODE>(ix.rows, j);
        const uint64_t j2 = row_id<MODE>(rj, j, ix);
        const uint32_t nj = row_n<MODE>(rj), ni = row_n<MODE>(row), oi = row_off<MODE>(row);
        no_ff = nj > oi ? (nj - oi < ni ? nj - oi : ni) : 0u;
        if (j2 < ix.r) e = ahead_half(rj, j2);
    }
    if (tally) {                                          // every lane of the wavefront is here
        const uint32_t a = wave_sum(no_ff), b = wave_sum(in ? row_n<MODE>(row) : 0u);
        if ((threadIdx.x & 63) == 0) tally_add(tally, a, b);
    }
    if (!in) return;
    if (HINTS) {
        // the window the walk sees row i in (pml_kernel_flatp: win_base) and the nearest run of each other base beyond its edges
        const uint64_t wb_last = ix.r - 4;
        const uint64_t wb = (i & ~3ull) < wb_last ? (i & ~3ull) : wb_last;
        const uint32_t c = row_c<MODE>(row);
        uint32_t h = 0;
        if (ix.sep == 0u && ix.sigma == 4u && i != ix.end_bwt_idx) {
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t a = k < c ? k : k + 1u;    // the base whose slot is k: alphamap_3[c][a] = a - (a > c), src/utils.cpp:5-8
                if (a > 3u) continue;
                uint32_t d = 0;
                if (row_thr<MODE>(row, k) == 0u) {        // threshold 0 <= offset: reposition_down
                    const uint64_t edge = wb + 3;
                    bool inside = false;
                    for (uint64_t t = i + 1; t <= edge && t < ix.r; ++t) inside = inside || row_c<MODE>(load_row<MODE>(ix.rows, t)) == a;
                    if (!inside)
                        for (uint64_t t = edge + 1; t <= edge + 7 && t < ix.r; ++t)
                            if (row_c<MODE>(load_row<MODE>(ix.rows, t)) == a) { d = (uint32_t)(t - edge); break; }
                } else {                                  // threshold n > offset: reposition_up
                    bool inside = false;
                    for (uint64_t t = wb; t < i; ++t) inside = inside || row_c<MODE>(load_row<MODE>(ix.rows, t)) == a;
                    if (!inside)
                        for (uint64_t s = 1; s <= 7 && s <= wb; ++s)
                            if (row_c<MODE>(load_row<MODE>(ix.rows, wb - s)) == a) { d = (uint32_t)s; break; }
                }
                h |= d << (3u * k);
            }
        }
        if (j >= ix.r) row.x = 0xFFFFFFFFu;               // (r < 2^32 - 1: still "not a row", move_structure.cpp:63-65)
        row.y = (row.y & 0x0FFFFFFFu) | (h << 28);
        e.y = (e.y & ~(0x3Fu << 25)) | (((h >> 4) & 0x3Fu) << 25);
    }
    uint8_t *line = out + (i >> 3) * 128u + (i & 7u) * 8u;
    __builtin_memcpy(line, &row, 8);
    __builtin_memcpy(line + 64, &e, 8);
    if (i + 4 >= ix.r) {                                  // the walk's last window: rows r-4 .. r-1 once more, in a line of their own
        uint8_t *tl = out + tail + (i + 4 - ix.r) * 8u;
        __builtin_memcpy(tl, &row, 8);
        __builtin_memcpy(tl + 64, &e, 8);
    }
}

// The tally alone, over every `stride`-th row: what share of the table's BWT positions reaches its LF target without a
// fast-forward -- the statistic that says whether look-ahead entries will be used (launch policy, movi_abi.hip) -- without
// building anything (two gathers per sampled row).
template <int MODE>
__global__ __launch_bounds__(256) void no_ff_share_kernel(DevIndex ix, uint64_t stride, unsigned long long *tally) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    uint32_t no_ff = 0, ni = 0;
    if (i < ix.r) {
        const uint2 row = load_row<MODE>(ix.rows, i);
        const uint64_t j = row_id<MODE>(row, i, ix);
        ni = row_n<MODE>(row);
        if (j < ix.r) {
            const uint32_t nj = row_n<MODE>(load_row<MODE>(ix.rows, j)), oi = row_off<MODE>(row);
            no_ff = nj > oi ? (nj - oi < ni ? nj - oi : ni) : 0u;
        }
    }
    const uint32_t a = wave_sum(no_ff), b = wave_sum(ni);
    if ((threadIdx.x & 63) == 0) tally_add(tally, a, b);
}
hipError_t tally_no_ff_share(int kmode, const DevIndex &ix, uint64_t stride, unsigned long long *d_tally, hipStream_t stream) {
    if (kmode != 6 || !d_tally || stride == 0) return hipErrorInvalidValue;
    const uint64_t n = (ix.r + stride - 1) / stride, blocks = (n + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(no_ff_share_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, stream, ix, stride, d_tally);
    return hipGetLastError();
}

uint64_t ahead_rows_bytes(uint64_t r) { return ((r + 7) / 8 + 1) * 128; }

bool ahead_rows_hinted(uint64_t r) { return r >= 8 && r < 0xFFFFFFFFull; }

hipError_t build_ahead_rows(int kmode, const DevIndex &ix, uint8_t *d_rows2, uint64_t *tail, hipStream_t stream, unsigned long long *d_tally) {
    if (!d_rows2 || !tail || ix.r < 8 || (ix.r >> 36) != 0 || kmode != 6) return hipErrorInvalidValue;
    *tail = ((ix.r + 7) / 8) * 128;
    hipError_t e = hipMemsetAsync(d_rows2, 0, ahead_rows_bytes(ix.r), stream);
    if (e != hipSuccess) return e;
    const uint64_t blocks = (ix.r + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (ahead_rows_hinted(ix.r)) hipLaunchKernelGGL((ahead_rows_kernel<6, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, ix, d_rows2, *tail, d_tally);
    else hipLaunchKernelGGL((ahead_rows_kernel<6, 0>), dim3((unsigned)blocks), dim3(256), 0, stream, ix, d_rows2, *tail, d_tally);
    return hipGetLastError();
}

// ---- deep rows (DevIndex::rows3, round 6): thread i writes row i of the copy -- the row, what the walk reads at j = id(i) and at
// j2 = id(j), and j3 = id(j2) (three dependent gathers) -- and its reposition hints for windows of three rows.
__global__ __launch_bounds__(256) void deep_rows_kernel(DevIndex ix, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t padded = ((ix.r + 2) / 3) * 3;
    if (i >= padded) return;
    uint32_t *W = out + (i / 3) * 16u;
    const uint32_t s = (uint32_t)(i % 3);
    uint32_t d0 = 0x0FFFFFFFu, d1 = (7u << 22) | (7u << 25) | (7u << 28), d2 = 0x0FFFFFFFu, d3 = 0, d4 = 0x0FFFFFFFu, x = 0;
    if (i < ix.r) {
        const uint2 row = load_row<6>(ix.rows, i);
        const uint64_t j = row_id<6>(row, i, ix);
        const uint32_t c = row_c<6>(row);
        d0 = (j < ix.r ? (uint32_t)j : 0x0FFFFFFFu) | (row_thr<6>(row, 0) << 28) | (row_thr<6>(row, 1) << 29) | (row_thr<6>(row, 2) << 30);
        d1 = row_n<6>(row) | (row_off<6>(row) << 11) | (c << 22) | (7u << 25) | (7u << 28);
        if (j < ix.r) {
            const uint2 rj = load_row<6>(ix.rows, j);
            const uint64_t j2 = row_id<6>(rj, j, ix);
            if (j2 < ix.r) {
                const uint2 rj2 = load_row<6>(ix.rows, j2);
                const uint64_t j3 = row_id<6>(rj2, j2, ix);
                d1 = (d1 & ~(7u << 25)) | (row_c<6>(rj) << 25);
                d2 = (uint32_t)j2;
                d3 = row_n<6>(rj) | (row_off<6>(rj) << 11);
                if (j3 < ix.r) {
                    const uint32_t n2 = row_n<6>(rj2), o2 = row_off<6>(rj2);
                    d1 = (d1 & ~(7u << 28)) | (row_c<6>(rj2) << 28);
                    d3 |= (n2 & 0x3FFu) << 22;
                    d4 = (uint32_t)j3 | ((n2 >> 10) << 28) | ((o2 & 7u) << 29);
                    x = o2 >> 3;
                }
            }
        }
        // hints: the nearest run of each other base beyond the edges of the window the walk sees row i in (ahead_rows_kernel's rule, three-row windows, reach 3)
        uint32_t h = 0;
        if (ix.sep == 0u && ix.sigma == 4u && i != ix.end_bwt_idx) {
            const uint64_t wb = (i / 3) * 3;
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t a = k < c ? k : k + 1u;    // the base whose slot is k: alphamap_3[c][a] = a - (a > c), src/utils.cpp:5-8
                if (a > 3u) continue;
                uint32_t d = 0;
                if (row_thr<6>(row, k) == 0u) {           // threshold 0 <= offset: reposition_down
                    const uint64_t edge = wb + 2;
                    bool inside = false;
                    for (uint64_t t = i + 1; t <= edge && t < ix.r; ++t) inside = inside || row_c<6>(load_row<6>(ix.rows, t)) == a;
                    if (!inside)
                        for (uint64_t t = edge + 1; t <= edge + 3 && t < ix.r; ++t)
                            if (row_c<6>(load_row<6>(ix.rows, t)) == a) { d = (uint32_t)(t - edge); break; }
                } else {                                  // threshold n > offset: reposition_up
                    bool inside = false;
                    for (uint64_t t = wb; t < i; ++t) inside = inside || row_c<6>(load_row<6>(ix.rows, t)) == a;
                    if (!inside)
                        for (uint64_t t = 1; t <= 3 && t <= wb; ++t)
                            if (row_c<6>(load_row<6>(ix.rows, wb - t)) == a) { d = (uint32_t)t; break; }
                }
                h |= d << (2u * k);
            }
        }
        d2 |= (h & 15u) << 28;
        x |= (h >> 4) << 8;
    }
    W[5u * s + 0u] = d0;
    W[5u * s + 1u] = d1;
    W[5u * s + 2u] = d2;
    W[5u * s + 3u] = d3;
    W[5u * s + 4u] = d4;
    atomicOr(&W[15], x << (10u * s));                     // (the table is zeroed first; three rows share the dword)
}

uint64_t deep_rows_bytes(uint64_t r) { return ((r + 2) / 3) * 64; }
bool deep_rows_eligible(uint64_t r) { return r >= 8 && r < 0x0FFFFFFFull; }

hipError_t build_deep_rows(int kmode, const DevIndex &ix, uint8_t *d_rows3, hipStream_t stream) {
    if (!d_rows3 || !deep_rows_eligible(ix.r) || kmode != 6) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_rows3, 0, deep_rows_bytes(ix.r), stream);
    if (e != hipSuccess) return e;
    const uint64_t blocks = (ix.r + 2 + 255) / 256;
    hipLaunchKernelGGL(deep_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ix, reinterpret_cast<uint32_t *>(d_rows3));
    return hipGetLastError();
}

hipError_t build_kmer_table(const DevIndex &ix, uint32_t K, uint4 *d_table, hipStream_t stream) {
    if (K < 1 || K > 12 || !d_table || ix.sigma - ix.sep != 4) return hipErrorInvalidValue;
    const uint64_t n = 1ull << (2 * K);
    hipLaunchKernelGGL(kmer_table_kernel<6>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ix, K, d_table);
    return hipGetLastError();
}

// The probe of the segmented path: does a walk started in the middle of a read fall into step quickly ON THIS BATCH?
// (Noisy long reads: within a few dozen bases.  Clean ones -- 0.1 % errors and below -- need a mismatch to meet at, i.e.
// hundreds to thousands of bases: cut into segments such a batch runs 1.5 - 2.3 x SLOWER than one lane per read.)
// Truth is not available before the walk, but two speculative walks are as good a witness: up to 1024 lanes each take
// a read, start walker B `lead` bases before the read's middle and walker A at the middle, and report whether the two
// are in the same state -- row, offset, match length -- within `reach` bases.
template <int MODE>
__global__ __launch_bounds__(256) void seg_probe_kernel(DevIndex ix, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs,
                                                       uint64_t n_reads, uint32_t lead, uint32_t reach, uint32_t *__restrict__ tally) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_lanes = (uint64_t)gridDim.x * blockDim.x;
    const EndThr ethr = end_thresholds(ix);
    // many reads: every stride-th one, probed at its middle; fewer reads than lanes: `per` probes spread along each read
    const uint64_t stride = n_reads > n_lanes ? n_reads / n_lanes : 1, per = n_reads < n_lanes ? n_lanes / n_reads : 1;
    const uint64_t rid = per > 1 ? t % n_reads : t * stride, slot = per > 1 ? t / n_reads : 0;
    bool valid = rid < n_reads && slot < per;
    const uint64_t beg = valid ? offs[rid] : 0, len = valid ? offs[rid + 1] - beg : 0;
    const uint64_t mid = len * (2 * slot + 1) / (2 * per);   // walker A starts just below this byte of the read
    valid = valid && mid >= reach && mid + lead <= len;
    const uint8_t *R = bases + beg + mid + lead;            // one past walker B's first base; walker A's is `lead` bases on
    const uint64_t r1 = ix.r - 1;
    const uint2 row0 = load_row<MODE>(ix.rows, r1);
    uint64_t ia = r1, ib = r1;
    uint32_t oa = row_n<MODE>(row0) - 1, ob = oa, ma = 0, mb = 0, ffx = 0, scx = 0, rpx = 0, failed = 0, met = 0;
    uint2 ra = row0, rb = row0;
    for (uint32_t k = 0; wave_any(valid && k < lead + reach && failed == 0u && met == 0u); ++k) {
        const bool live = valid && k < lead + reach && failed == 0u && met == 0u;
        uint32_t a = 0xFFu;
        if (live) a = s_code[*(R - 1 - (int64_t)k)];
        // (the two LF moves go out together once both walkers are under way: half the round trips)
        uint32_t e0 = 0;
        if (k > lead) e0 = lf_step2<MODE>(ix, live, ia, oa, ra, ib, ob, rb, ffx);
        else e0 = lf_step<MODE>(ix, live && k != 0, ib, ob, rb, ffx);
        const bool l2 = live && e0 == 0u;
        const uint32_t e = walk_base<MODE>(ix, ethr, l2, false, a, ib, ob, rb, mb, ffx, scx, rpx);
        const uint32_t e2 = walk_base<MODE>(ix, ethr, l2 && k >= lead, false, a, ia, oa, ra, ma, ffx, scx, rpx);
        if (e0 | e | e2) failed = 1;
        if (live && failed == 0u && k >= lead && ia == ib && oa == ob && ma == mb) met = 1;
        // half way through with fewer than half of the wavefront's probes in step: nine in ten will not make it -- stop
        // (a batch of clean reads otherwise walks every probe to the end: 1.2 ms instead of 0.4)
        if (k == lead + reach / 2 && 2 * __popcll(__ballot(met != 0u)) < __popcll(__ballot(valid))) break;
        if (k == lead + reach / 6 && 10 * __popcll(__ballot(met != 0u)) < __popcll(__ballot(valid))) break;   // (nor with < 10 % after a sixth)
        // ... and with 19 in 20 in step already the verdict of this wavefront is in (checked every 16 bases)
        if (k > lead && (k & 15u) == 15u && 20 * __popcll(__ballot(met != 0u)) >= 19 * __popcll(__ballot(valid))) break;
    }
    const uint32_t nv = wave_sum(valid ? 1u : 0u), nm = wave_sum(met);
    if ((threadIdx.x & 63) == 0) {
        if (nv) atomicAdd(&tally[0], nv);
        if (nm) atomicAdd(&tally[1], nm);
    }
}

// go = at least 8 probes, 9 in 10 of them in step within reach (or no probing asked for)
// The segment plan indexes its scratch (checkpoints: one per 32 bases; the verdict-only PML buffer) with the batch's own
// offsets, so it needs what the header states for the *_device entry points: offsets[0] == 0 and offsets[n_reads] <=
// n_bases.  The host entry points build their offsets that way; a device caller's are checked HERE, on the device,
// and a batch that breaks the contract is not cut (go = 0: every kernel of the plan idles, K3 walks whole reads).
__global__ void seg_decide_kernel(const uint32_t *__restrict__ tally, uint32_t probe, uint32_t *__restrict__ go,
                                  const uint64_t *__restrict__ offs, uint64_t n_reads, uint64_t n_bases) {
    const uint32_t fits = (uint32_t)(offs[0] == 0ull && offs[n_reads] <= n_bases);
    *go = fits & (probe ? (uint32_t)(tally[0] >= 8u && (uint64_t)tally[1] * 10ull >= (uint64_t)tally[0] * 9ull) : 1u);
}

// Segments of a read of `len` bases: n = len / seg_len of them (1 below 2 x seg_len), each T bases -- a multiple of
// 32, so that every checkpoint sits at an emission index k with k % 32 == 31 in the read's as in the segment's count --
// the last one whatever is left.
__device__ __forceinline__ void seg_shape(uint64_t len, uint32_t seg_len, uint64_t &n, uint64_t &T) {
    n = len >= 2ull * seg_len ? len / seg_len : 1;
    T = (((len + n - 1) / n) + 31) & ~31ull;
    if (T == 0) T = 32;
    n = len ? (len + T - 1) / T : 1;
}

__global__ __launch_bounds__(256) void seg_count_kernel(const uint64_t *__restrict__ offs, uint64_t n_reads, uint32_t seg_len,
                                                       uint64_t *__restrict__ n_of, const uint32_t *__restrict__ go) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_reads) return;
    uint64_t n = 0, T = 0;
    if (t < n_reads && *go != 0u) seg_shape(offs[t + 1] - offs[t], seg_len, n, T);   // go == 0: no segments at all
    n_of[t] = n;                                          // entry n_reads = 0: its exclusive sum is the total
}

__global__ __launch_bounds__(256) void seg_maxlen_kernel(const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                        uint32_t *__restrict__ max_len) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = t < n_reads ? (uint32_t)(offs[t + 1] - offs[t]) : 0u;   // (reads are shorter than 2^32)
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_len, m);
}

// Segment j of read t covers the emission indexes [j T, min(len, (j + 1) T)): the bases [len - k_end, len - k_begin) of
// the read, PMLs to offs[t] + k_begin; seg_rem = the bases from its first one to the end of the walk (the read's first base).
__global__ __launch_bounds__(256) void seg_fill_kernel(const uint64_t *__restrict__ offs, uint64_t n_reads, uint32_t seg_len,
                                                      const uint64_t *__restrict__ first, uint64_t *__restrict__ seg_in,
                                                      uint64_t *__restrict__ seg_out, uint32_t *__restrict__ seg_l,
                                                      uint32_t *__restrict__ seg_j, uint32_t *__restrict__ seg_rem) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads) return;
    const uint64_t beg = offs[t], len = offs[t + 1] - beg;
    uint64_t n = 0, T = 0;
    seg_shape(len, seg_len, n, T);
    const uint64_t s0 = first[t];
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t k0 = j * T, k1 = (k0 + T < len) ? k0 + T : len;
        seg_in[s0 + j] = beg + len - k1;
        seg_out[s0 + j] = beg + k0;
        seg_l[s0 + j] = (uint32_t)(k1 - k0);
        seg_j[s0 + j] = (uint32_t)(j > 0xFFFFFFFFull ? 0xFFFFFFFFull : j);
        seg_rem[s0 + j] = (uint32_t)(len - k0);
    }
}

// What a boundary lane found (K2).
struct SegJoin {
    uint32_t ff, scan, repo;     // the real walk's counters from this boundary to the meeting point, plus the speculative lane's after it
    uint32_t kend;               // last emission index (from this segment's first base) this lane walked
    uint32_t segs;               // segments it reached into beyond its own (0: met inside its own segment)
    uint32_t how;                // 0 unresolved (an invariant violation, or no meeting point within `max_over` bases), 1 met, 2 walked to the read's end
};

// K2: one lane per segment that is not the first of its read.  It takes up the walk where the segment before left it
// (that segment's final state is the read's real state there IF the chain of boundaries before it holds -- K3 checks
// that), walks on base by base -- the plain base-synchronous automaton of pml_kernel<MODE, 0> -- and stops at the first
// checkpoint (every 32 bases) where its row, offset and match length equal what the speculative lane of that stretch
// recorded: from there on the speculative PMLs are the real ones.  Usually that is a few dozen bases in; if not, it
// walks on into the following segments (up to max_over bases: those segments' own boundary lanes are then void).
// What it counted up to the meeting point plus what the speculative lane counted after it is this stretch's share of the
// read's fast-forwards / scans / repositions.  It runs twice: PASS 0 only looks for the meeting point and writes no PML
// -- a lane whose start state turns out not to be real would write rubbish over a stretch that belongs to another --;
// once K3 has followed the chains, PASS 1 walks the stretches of the lanes that count again (a few dozen bases each, as
// a rule) and writes their PMLs: those stretches are disjoint.
constexpr uint32_t kStitchLanes = 16;
template <int MODE, int PASS>
__global__ __launch_bounds__(256) void seg_stitch_kernel(DevIndex ix, const uint8_t *__restrict__ bases, SegArgs seg,
                                                        const uint32_t *__restrict__ seg_j, const uint32_t *__restrict__ seg_rem,
                                                        uint32_t max_over, const uint8_t *__restrict__ on_chain,
                                                        uint16_t *__restrict__ out, SegJoin *__restrict__ join) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();
    // kStitchLanes boundaries per wavefront (the other lanes idle): a wavefront walks until the LAST of its lanes has met its
    // checkpoint, and the meeting points are spread like lock-on + the wait for the next substitution -- the maximum over 16 lanes lies
    // a checkpoint or two nearer than the maximum over 64, and the launch has wavefront slots to spare (profiles/r06_few_long_reads.txt)
    const uint64_t s = (uint64_t)blockIdx.x * kStitchLanes + threadIdx.x;
    const bool mine = threadIdx.x < kStitchLanes && *seg.go != 0u && s < *seg.n_seg && seg_j[s] != 0 && (PASS == 0 || on_chain[s] == 2);
    const EndThr ethr = end_thresholds(ix);
    uint32_t ff_total = 0, scan_total = 0, repo_total = 0, failed = 0, how = 0;
    bool live0 = mine;
    if (PASS == 0 && mine && (seg.tot[s].flag != 0u || seg.tot[s - 1].flag != 0u)) live0 = false;   // a speculative walk broke an invariant: K3 decides
    const uint32_t T = live0 ? seg.seg_len[s] : 1u;      // (every segment of a read but its last has this length)
    const uint32_t rem = live0 ? seg_rem[s] : 0u;
    const uint32_t over = PASS == 0 ? (rem < max_over ? rem : max_over) : (live0 ? join[s].kend + 1u : 0u);
    const uint64_t len = over;
    const uint8_t *R = bases + (live0 ? seg.seg_in[s] + T : 0);          // one past this segment's first base
    const uint64_t obeg = live0 ? seg.seg_out[s] : 0;
    uint16_t *O = out + obeg;
    uint64_t idx = 0, rb8 = 0;
    uint32_t off = 0, ml = 0, kend = 0;
    SegJoin res{};
    if (live0) {
        const SegFin f = seg.fin[s - 1];
        idx = f.idx; off = f.off; ml = f.ml;
    }
    uint2 row = load_row<MODE>(ix.rows, idx);
    for (uint64_t k = 0; wave_any(k < len && failed == 0u && how == 0u); ++k) {
        bool live = k < len && failed == 0u && how == 0u;
        uint32_t a = 0xFFu;
        if (live && (k & 7ull) == 0ull) {                                // the next eight bases by one load (off the walk's dependent chain)
            if (k + 8 <= len) __builtin_memcpy(&rb8, R - 8 - (int64_t)k, 8);
            else {
                rb8 = 0;
                for (uint64_t i = 0; k + i < len; ++i) rb8 |= (uint64_t)*(R - 1 - (int64_t)(k + i)) << (8u * (7u - (uint32_t)i));
            }
        }
        if (live) a = s_code[(uint32_t)(rb8 >> (8u * (7u - (uint32_t)(k & 7ull)))) & 0xFFu];
        const uint32_t e = walk_base<MODE>(ix, ethr, live, true, a, idx, off, row, ml, ff_total, scan_total, repo_total);
        if (e) { failed = e; live = false; }
        if (PASS == 1) {
            if (live) O[k] = (uint16_t)(ml > 65535u ? 65535u : ml);
        } else if (live) {
            if (k < T) O[k] = (uint16_t)(ml > 65535u ? 65535u : ml);     // inside its own segment: no other lane writes there now
            kend = (uint32_t)k;
            if ((k & 31ull) == 31ull) {
                const SegCkpt c = seg.ckpt[(obeg + k) >> 5];
                if (c.idx == idx && c.off == off && c.ml == ml) {
                    const uint32_t ds = (uint32_t)(k / T);                // the segment the meeting point lies in
                    const SegTot spec = seg.tot[s + ds];
                    how = spec.flag == 0u ? 1u : 0u;
                    if (spec.flag != 0u) failed = spec.flag;             // (that speculative lane's records are not to be trusted)
                    res.ff = ff_total + (spec.ff - c.ff); res.scan = scan_total + (spec.scan - c.scan);
                    res.repo = repo_total + (spec.repo - c.repo); res.segs = ds;
                }
            }
            if (how == 0u && failed == 0u && k + 1 == rem) {             // the read's first base: nothing left to meet
                how = 2u;
                res.ff = ff_total; res.scan = scan_total; res.repo = repo_total; res.segs = (uint32_t)(k / T);
            }
        }
    }
    if (PASS == 0 && mine) {
        res.kend = kend;
        res.how = how;
        join[s] = res;
    }
}

// K3, first half: one lane per read follows the chain of its boundaries.  The first segment is real by construction;
// the lane of the boundary behind a real stretch started from the real state, so what it found holds: it met the
// speculative walk `segs` segments on (the boundaries in between are void) or walked to the read's end.  The lanes on
// the chain are marked for the writing pass of K2, their counters added up.  A chain that breaks -- an invariant violation anywhere in the
// read, no meeting point within reach -- puts the read on the list of pml_kernel_flatp<..., SEG = 2>, which walks it
// from end to end (and reports its error, if any).
__global__ __launch_bounds__(256) void seg_finalize_kernel(const uint32_t *__restrict__ go, const uint64_t *__restrict__ first,
                                                          uint64_t n_reads, const SegTot *__restrict__ tot,
                                                          SegJoin *__restrict__ join, const uint32_t *__restrict__ seg_len,
                                                          const uint32_t *__restrict__ seg_rem,
                                                          uint8_t *__restrict__ on_chain, uint8_t *__restrict__ read_fail,
                                                          uint8_t *__restrict__ err, DevStats *stats) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff = 0, scan = 0, repo = 0, nseg = 0, nbad = 0;
    if (t < n_reads && *go == 0u) {
        read_fail[t] = 1;                                 // the probe advised against segments: every read is walked by one lane
    } else if (t < n_reads) {
        const uint64_t s0 = first[t], s1 = first[t + 1];
        uint32_t bad = 0;
        for (uint64_t s = s0; s < s1; ++s) {
            bad |= (uint32_t)(tot[s].flag != 0u);
            on_chain[s] = 0;
        }
        ff = tot[s0].ff; scan = tot[s0].scan; repo = tot[s0].repo;
        uint64_t s = s0 + 1;
        while (!bad && s < s1) {
            const SegJoin j = join[s];
            if (j.how == 0u) { bad = 1; break; }
            ff += j.ff; scan += j.scan; repo += j.repo;
            on_chain[s] = 1;
            // The find pass wrote this lane's PMLs as far as its own segment goes; so did the lanes of the boundaries it
            // walked past, from states that were not the read's: it walks again (write pass) over everything they touched.
            const uint64_t T = seg_len[s], last_void = j.how == 2u ? s1 - 1 : s + (uint64_t)j.segs;
            if (last_void > s || (uint64_t)j.kend >= T) {
                uint64_t extent = j.kend;
                for (uint64_t q = s + 1; q <= last_void; ++q) {
                    const uint64_t tq = seg_len[q], kq = join[q].kend < tq ? join[q].kend : tq - 1;
                    const uint64_t reach = (q - s) * T + kq;
                    extent = reach > extent ? reach : extent;
                }
                const uint64_t rem = seg_rem[s];
                join[s].kend = (uint32_t)(extent < rem ? extent : rem - 1);
                on_chain[s] = 2;
            }
            if (j.how == 2u) break;
            s += (uint64_t)j.segs + 1;
        }
        if (bad)
            for (uint64_t q = s0 + 1; q < s1; ++q) on_chain[q] = 0;
        read_fail[t] = (uint8_t)bad;
        if (bad) { ff = 0; scan = 0; repo = 0; }
        else if (err) err[t] = 0;
        nseg = (uint32_t)(s1 - s0);
        nbad = bad;
    }
    const uint32_t ffw = wave_sum(ff), scw = wave_sum(scan), rpw = wave_sum(repo), sgw = wave_sum(nseg), bdw = wave_sum(nbad);
    if ((threadIdx.x & 63) == 0 && stats) {
        if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
        if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
        if (rpw) atomicAdd(&stats->repositions, (unsigned long long)rpw);
        if (sgw) atomicAdd(&stats->segments, (unsigned long long)sgw);
        if (bdw) atomicAdd(&stats->rewalked, (unsigned long long)bdw);
    }
}

namespace {
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr int kSegOverrun = 4;   // a boundary lane looks for its meeting point over at most this many seg_len of bases
// The segment length of one call: cfg.seg_len, or shorter (down to 512) when the batch is so small that even then the
// segments would not fill the GPU: `waves` wavefronts of segments per CU are aimed at.  The ZML parse is latency-bound
// and wants many (24: 25 k x 10 kbp 13.4 -> 17.2 Gbases/s, 5 k 1.0 uncut -> 6.9, 1 %-error reads 7.7 uncut -> 15.1); the
// PML walk pays more per boundary than it gains from lanes beyond ~8 per CU (with 24: 5 k x 10 kbp 7.5 -> 18.2, but
// 200 x 1 Mbp 27.5 -> 20.5 and 1 %-error reads 25.3 -> 20.4).  A full batch (100 k x 10 kbp) keeps cfg.seg_len.
uint32_t call_seg_len(const LaunchCfg &cfg, uint64_t n_bases, uint64_t waves) {
    const uint64_t want = (n_bases / ((uint64_t)cfg.num_cus * 64ull * waves)) & ~31ull;
    const uint64_t lo = cfg.seg_len < 512 ? (uint64_t)cfg.seg_len : 512ull;
    return (uint32_t)(want < lo ? lo : (want > (uint64_t)cfg.seg_len ? (uint64_t)cfg.seg_len : want));
}
}

// The segmented PML path: plan (count, scan, fill), K1, K2, K3.  Everything on `stream`, nothing read back: the grids are
// sized for the most segments the batch could have (n_reads + n_bases / seg_len) and surplus lanes leave at once.
// Is there anything to gain?  One lane per read already fills the GPU when there are enough reads of about the same
// length (100 k x 10 kbp: 40.5 Gbases/s either way); segments pay when lanes are scarce -- fewer than 4 wavefronts of reads
// per CU: 60 k x 10 kbp 28.6 -> 34.3, 25 k 12.1 -> 31.4, 200 x 1 Mbp 0.11 -> 27 Gbases/s -- or when the batch is ragged
// (its longest read holds a lane long after the others are done; log-normal lengths around 10 kbp: 12.2 -> 32.3).
// *declined = true: nothing was launched, the caller goes on with one lane per read.  `ragged_hint`: 1 / 0 when the
// caller knows the lengths (the *_host entry points), -1 when only the device does: then a big batch costs one
// reduction kernel and a 4-byte read-back -- this call waits for `stream` there.
static hipError_t launch_pml_segmented(const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                       uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                                       const LaunchCfg &cfg, hipStream_t stream, SegWorkspace *ws, bool big_batch_cap,
                                       int ragged_hint, bool *declined, const ClsArgs &bins, int *verdict, LaunchInfo *info) {
    // `verdict` (optional, in / out): a caller that cuts one batch into several launches (the overlapped host path) lets
    // the first one probe and hands its verdict to the others -- 1: cut without probing (no read-back, the launch stays
    // asynchronous), 0: do not cut; -1 on entry: not decided yet.
    // cfg.seg_probe == 2: the CALLER decides (cfg.seg_verdict) -- no probe, no length reduction, no read-back: the launch
    // stays asynchronous (stream capture, callers that pipeline several streams).
    int forced = -1;
    if (verdict && *verdict >= 0) forced = *verdict;
    if (cfg.seg_probe == 2) forced = cfg.seg_verdict ? 1 : 0;
    if (forced == 0) { *declined = true; return hipSuccess; }
    const bool probe = cfg.seg_probe == 1 && forced != 1;
    // a verdict-only call (bins, no PML vector) keeps its PMLs in the workspace, sized from n_bases: for offsets that only
    // the device has seen (ragged_hint < 0) the verdict of seg_decide_kernel -- which checks them -- is always read back
    const bool ws_pml = bins.bin_width && !d_out;
    if (ws_pml && ragged_hint < 0 && cfg.seg_probe == 2) { *declined = true; return hipSuccess; }
    const bool read_back = probe || (ws_pml && ragged_hint < 0);
    // Classification bins (bins.bin_width != 0): not fused into this walk -- a bin spans segments -- but reduced from the
    // resident PML vector afterwards (classify_kernel: 2 B per base, streaming); without a caller's vector (d_out == NULL:
    // verdicts only) the PMLs go to the workspace.
    *declined = false;
    const uint32_t S = call_seg_len(cfg, n_bases, 8);
    if (cfg.seg_probe == 1 && n_reads >= (uint64_t)cfg.num_cus * 64ull * 4ull) {
        if (ragged_hint == 0) { *declined = true; return hipSuccess; }
        if (ragged_hint < 0) {
            if (ws->cap < 64) {
                if (ws->buf) (void)hipFree(ws->buf);
                ws->buf = nullptr;
                ws->cap = 0;
                hipError_t ea = hipMalloc(&ws->buf, 4096);
                if (ea != hipSuccess) return ea;
                ws->cap = 4096;
            }
            uint32_t *d_max = static_cast<uint32_t *>(ws->buf), h_max = 0;
            hipError_t ea = hipMemsetAsync(d_max, 0, 4, stream);
            if (ea != hipSuccess) return ea;
            hipLaunchKernelGGL(seg_maxlen_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, d_offsets, n_reads, d_max);
            ea = hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, stream);
            if (ea == hipSuccess) ea = hipStreamSynchronize(stream);
            if (ea != hipSuccess) return ea;
            if ((uint64_t)h_max * 2ull <= (n_bases / n_reads) * 3ull) { *declined = true; return hipSuccess; }   // longest read <= 1.5 x the mean
        }
    }
    const uint64_t max_seg = n_reads + n_bases / S + 1;
    if (max_seg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const uint64_t n_ck = (n_bases >> 5) + 2;
    size_t temp_bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                                    (int)(n_reads + 1), stream);
    if (e != hipSuccess) return e;
    // carve the workspace
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes ? bytes : 8); return o; };
    const size_t o_nof = take((n_reads + 1) * 8), o_first = take((n_reads + 1) * 8), o_temp = take(temp_bytes),
                 o_in = take(max_seg * 8), o_out = take(max_seg * 8), o_len = take(max_seg * 4), o_j = take(max_seg * 4),
                 o_rem = take(max_seg * 4), o_fin = take(max_seg * sizeof(SegFin)), o_tot = take(max_seg * sizeof(SegTot)),
                 o_join = take(max_seg * sizeof(SegJoin)), o_chain = take(max_seg), o_fail = take(n_reads),
                 o_ck = take(n_ck * sizeof(SegCkpt)), o_go = take(32),
                 o_pml = take((bins.bin_width && !d_out) ? n_bases * 2 : 0), o_err = take((bins.bin_width && !d_err) ? n_reads : 0);
    if (ws->cap < off) {
        if (ws->buf) (void)hipFree(ws->buf);
        ws->buf = nullptr;
        ws->cap = 0;
        const size_t want = off + (off >> 3);
        e = hipMalloc(&ws->buf, want);
        if (e != hipSuccess) return e;
        ws->cap = want;
    }
    uint8_t *B = static_cast<uint8_t *>(ws->buf);
    if (bins.bin_width && !d_out) d_out = reinterpret_cast<uint16_t *>(B + o_pml);
    if (bins.bin_width && !d_err) d_err = B + o_err;
    uint64_t *n_of = reinterpret_cast<uint64_t *>(B + o_nof), *first = reinterpret_cast<uint64_t *>(B + o_first);
    uint64_t *seg_in = reinterpret_cast<uint64_t *>(B + o_in), *seg_out = reinterpret_cast<uint64_t *>(B + o_out);
    uint32_t *seg_l = reinterpret_cast<uint32_t *>(B + o_len), *seg_j = reinterpret_cast<uint32_t *>(B + o_j),
             *seg_rem = reinterpret_cast<uint32_t *>(B + o_rem);
    SegJoin *join = reinterpret_cast<SegJoin *>(B + o_join);
    uint8_t *on_chain = B + o_chain, *read_fail = B + o_fail;
    SegArgs seg;
    seg.seg_in = seg_in; seg.seg_out = seg_out; seg.seg_len = seg_l; seg.n_seg = first + n_reads;
    seg.ckpt = reinterpret_cast<SegCkpt *>(B + o_ck);
    seg.fin = reinterpret_cast<SegFin *>(B + o_fin);
    seg.tot = reinterpret_cast<SegTot *>(B + o_tot);
    seg.read_fail = read_fail;
    uint32_t *go = reinterpret_cast<uint32_t *>(B + o_go);               // go | probes | probes in step
    seg.go = go;
    const unsigned bt256 = 256;
    // the probe: is this a batch whose walks fall into step quickly?
    e = hipMemsetAsync(go, 0, 32, stream);
    if (e != hipSuccess) return e;
    if (probe)
        hipLaunchKernelGGL(seg_probe_kernel<6>, dim3(16), dim3(64), 0, stream, ix, d_bases, d_offsets, n_reads, 32u, 384u, go + 1);
    hipLaunchKernelGGL(seg_decide_kernel, dim3(1), dim3(1), 0, stream, go + 1, (uint32_t)probe, go, d_offsets, n_reads, n_bases);
    if (read_back) {
        // The verdict is read back (this call waits for the probe): a batch it advises against then takes exactly the
        // one-lane-per-read path -- fused bins included -- instead of a dozen kernels that find out one by one.
        uint32_t h_go = 0;
        e = hipMemcpyAsync(&h_go, go, 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (verdict && probe) *verdict = h_go ? 1 : 0;
        if (!h_go) { *declined = true; return hipSuccess; }
    }
    hipLaunchKernelGGL(seg_count_kernel, dim3((unsigned)((n_reads + 1 + bt256 - 1) / bt256)), dim3(bt256), 0, stream, d_offsets,
                       n_reads, S, n_of, go);
    e = hipcub::DeviceScan::ExclusiveSum(B + o_temp, temp_bytes, n_of, first, (int)(n_reads + 1), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, d_offsets,
                       n_reads, S, first, seg_in, seg_out, seg_l, seg_j, seg_rem);
    // K1 and K3b: the window-parallel lane state machine in blocks of one wavefront, capped like any big batch
    const int bt = 64;
    const bool seg_deep = ix.rows3 != nullptr && ix.idx32 != 0u && cfg.deep > 0 && cfg.stage_reads != 0;   // K1 / K3 on the deep rows only on request ("deep" 1): segments are long reads (launch_pml's rule)
    auto lds_for = [&](uint64_t lanes) -> size_t {
        int wpc = cfg.waves_per_cu;
        if (wpc < 0) wpc = 0;
        if (cfg.waves_per_cu == 0 && big_batch_cap && lanes > (uint64_t)cfg.num_cus * 64u * 18u)
            wpc = ((ix.rows2 != nullptr || seg_deep) && cfg.stage_reads != 0) ? kCapWavesAhead : kCapWaves;
        if (wpc > 0 && wpc < 32) return ((163840u / (unsigned)wpc) & ~1023u) - 1024u;
        return 0;
    };
    const ClsArgs cls;
    const uint32_t *d_order = nullptr;
    // (segments and re-walked reads stage their bases through LDS and walk on the look-ahead rows like any other launch:
    // launch_pml's policy -- the cap's padding, or what the launch's wavefronts per CU leave of the CU's LDS)
    DevIndex ixl = ix;
    ixl.inwin = cfg.inwin ? 1u : 0u;
    ixl.hint_w = seg_deep ? (cfg.hints != 0 ? 2u : 0u) : ((cfg.hints != 0 && ix.hints != 0u && ix.rows2 != nullptr) ? 3u : 0u);
    // (pair-shared gathers on tables beyond the TLBs' reach: launch_pml's rule)
    const bool seg_pair = !seg_deep && (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && ix.r * (ix.rows2 != nullptr ? 16ull : 8ull) >= kPairLoadBytes));
    size_t dyn_lds = 0;
    bool seg_ring = false;
    auto stage_for = [&](uint64_t lanes) {
        dyn_lds = lds_for(lanes);
        if (cfg.stage_reads != 0 && dyn_lds == 0) {
            const uint64_t wn = ((lanes + bt - 1) / bt + (uint64_t)cfg.num_cus - 1) / (uint64_t)cfg.num_cus;
            if (wn <= 18) dyn_lds = std::min<size_t>(21504 + (cfg.out_ring != 0 ? kOutRingBytes : 0u), ((163840u / (unsigned)(wn + std::max<uint64_t>(2, wn / 4))) & ~1023u) - 1024u);
        }
        // (segments are long reads: their PMLs leave through the ring in LDS where the block has room for it -- launch_pml)
        const size_t ring_b = (cfg.out_ring != 0 && cfg.stage_reads != 0 && dyn_lds >= kOutRingBytes + 96u * 64u) ? kOutRingBytes : 0;
        const uint32_t cap = (uint32_t)std::min<size_t>(1024, ((dyn_lds - ring_b) / 64) & ~(size_t)15);
        ixl.stage_lds = (cfg.stage_reads != 0 && cap >= 96) ? cap : 0u;
        seg_ring = ring_b != 0 && ixl.stage_lds != 0u;
    };
    auto launch_seg = [&](int segv, uint64_t lanes, LaunchInfo *li) -> hipError_t {
        stage_for(lanes);
        WalkLaunch L;
        L.grid = dim3((unsigned)((lanes + bt - 1) / bt)); L.block = dim3(bt); L.dyn_lds = dyn_lds; L.stream = stream;
        L.ix = ixl; L.bases = d_bases; L.offs = d_offsets; L.n = lanes; L.out = d_out; L.err = d_err; L.stats = d_stats;
        L.order = d_order; L.cls = cls; L.seg = seg;
        L.cls_mode = 0; L.sep = ix.sep ? 1 : 0; L.stg = ixl.stage_lds != 0u ? 1 : 0;
        L.ahd = (L.stg && seg_deep) ? 2 : ((L.stg && ix.rows2 != nullptr) ? 1 : 0); L.psh = (L.stg && seg_pair) ? 1 : 0; L.ring = seg_ring ? 1 : 0;
        return ix.idx32 ? launch_walkseg_u32(segv, L, li) : launch_walkseg_u64(segv, L, li);
    };
    e = launch_seg(1, max_seg, info);
    if (e != hipSuccess) return e;
    if (info) {                                           // the dominant kernel: K1
        info->variant = 14; info->block_threads = 64; info->segmented = 1; info->idx64 = ix.idx32 ? 0 : 1;
        info->waves_per_cu = 0; info->staged = (int)ixl.stage_lds; info->ahead = ixl.stage_lds == 0u ? 0 : (seg_deep ? 2 : (ix.rows2 != nullptr ? 1 : 0));
    }
    // (blocks of one wavefront: a boundary lane that has to walk far holds up only the 63 beside it)
    const uint32_t max_over = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)cfg.seg_len * (uint64_t)kSegOverrun);
    hipLaunchKernelGGL((seg_stitch_kernel<6, 0>), dim3((unsigned)((max_seg + kStitchLanes - 1) / kStitchLanes)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    hipLaunchKernelGGL(seg_finalize_kernel, dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, go, first, n_reads,
                       seg.tot, join, seg_l, seg_rem, on_chain, read_fail, d_err, d_stats);
    hipLaunchKernelGGL((seg_stitch_kernel<6, 1>), dim3((unsigned)((max_seg + kStitchLanes - 1) / kStitchLanes)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    e = launch_seg(2, n_reads, nullptr);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && bins.bin_width)
        e = launch_classify(d_out, d_offsets, n_reads, bins.bin_width, bins.thr, bins.above, bins.below, bins.sum_max, stream, d_err);
    return e;
}

__global__ __launch_bounds__(256) void copy_words_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}
hipError_t launch_copy_words(uint64_t *d_dst, const uint64_t *src, uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_dst, src, n);
    return hipGetLastError();
}

// ---- launch log (diagnostic; movi_launch_log): the distinct walk kernels launched since it was switched on / last read
static std::atomic<bool> g_launch_log_on{false};
static std::mutex g_launch_log_m;
static std::set<std::string> g_launch_log;
void note_walk_launch(const char *kernel_name) {
    if (!g_launch_log_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> g(g_launch_log_m);
    g_launch_log.insert(kernel_name);
}
size_t take_launch_log(char *buf, size_t cap) {
    g_launch_log_on.store(true, std::memory_order_relaxed);
    std::lock_guard<std::mutex> g(g_launch_log_m);
    std::string all;
    for (const auto &k : g_launch_log) { all += k; all += '\n'; }
    g_launch_log.clear();
    if (buf && cap) {
        const size_t n = all.size() < cap - 1 ? all.size() : cap - 1;
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size();
}

// The launch policy of one PML call, apart from the segment plan's own decisions (launch_pml_segmented): which kernel, which block,
// how much dynamic LDS and what it holds.  launch_pml acts on it; pml_mask_needs_tmp asks it whether the reset-mask output can
// come straight from the walk.
namespace {
struct PmlPlan {
    int v = 14, bt = 64, wpc = 0;
    bool seg_eligible = false;       // a batch of long reads: the segment plan gets the first say (it may decline)
    size_t dyn_lds = 0;
    uint32_t stage_lds = 0;
    bool use_ring = false, use_ahead = false, use_pair = false;
    bool use_deep = false;           // the walk runs on the deep rows (DevIndex::rows3: three bases per gather)
};
PmlPlan plan_pml(const DevIndex &ix, const LaunchCfg &cfg, uint64_t n_reads, uint64_t n_bases, int cm, bool logging, bool have_seg_ws,
                 bool ordered, bool want_mask) {
    PmlPlan P;
    // Variants: 0 first correct kernel (serves --logs), 1 base-synchronous packed I/O (tables of fewer than 8 rows, batches of
    // fewer than 16 bases; A/B), 14 = the lane state machine over row windows (pml_kernel_flatp, movi_walk.hpp; the default).
    // (2-13 were experiments -- branchy / row-at-a-time state machines, 2/4-row neighbour windows, the unpipelined window
    // kernel, hop-by-hop advances, lane refill -- measured slower or no faster and removed; numbers in DESIGN.md section 3.)
    // Auto selection (measured on MI355X, profiles/r02_*): the state machine in blocks of ONE wavefront, and -- when there are
    // more reads than ~18 waves per CU -- at most kCapWaves wavefronts resident per CU.  Why a cap: between two
    // iterations of a lane its cache lines (the row window's neighbours, its read, its output) must survive in the
    // 4 MiB L2 of its XCD; with all 32 wave slots of a CU walking, 8 MiB of lines are in flight per XCD and neighbour
    // rows are refetched from the fabric.  1 M x 150 bp, Gbases/s, uncapped / capped at 8-10 waves per CU /
    // variant 1 (base-synchronous: its neighbour loads follow the gather at once, so it wants all the occupancy it can
    // get): pangenome 14 M rows 43.2 / 48.2 / 46.4; random tables of 10 M rows 40.0 / 47.5 / 43.6, 60 M 35.7 / 40.8 /
    // 36.3, 250 M (2 GB) 29.1 / 32.8 / 29.2, 500 M 27.9 / 30.3 / 27.7, 1 B (8 GB) 27.4 / 27.7 / 28.3.
    int v = cfg.pml_variant;
    // (batches of up to ~18 waves per CU run in ONE round, uncapped: with the cap, 224 k reads = 13.7 waves per CU run as
    // a full round of 9 and a half-empty one -- 38.3 against 39.2 Gbases/s; 300 k reads: 38.2 against 41.2; from 400 k
    // reads on the cap wins: 43.9 against 41.7.  profiles/r02_occupancy_cap_sweeps.txt)
    const bool big_batch = n_reads > (uint64_t)cfg.num_cus * 64u * 18u;
    if (v < 0) v = 14;
    if (v == 14 && (ix.r < 8 || n_bases < 16)) v = 1;                        // the clamped window needs >= 4 rows (r >= 8: two windows), the
                                                                             // 16-base fetches >= 16 bytes of bases
    if (cm != 0 && v == 0) v = 1;                                            // the first kernel carries no fused bins
    if (logging) v = 0;                                                      // per-base logs: the first kernel keeps them
    P.v = v;
    // Batches of long reads: segment-parallel (plain PML through the default kernel only).  One lane per read leaves the
    // GPU short of walks -- 100 k reads are 6 wavefronts per CU, and a single 1 Mbp read holds its lane for 2 s --;
    // cut into segments the same batch fills it like a batch of short reads.
    P.seg_eligible = have_seg_ws && !logging && cfg.seg_len >= 32 && !ordered && v == 14 && cfg.block_threads <= 64 &&
                     n_bases / n_reads >= 2ull * (uint64_t)cfg.seg_len && n_reads + n_bases / (uint64_t)cfg.seg_len < 0x7FFFFFF0ull;
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : 64;           // one wavefront per block: finest dispatch grain
    P.bt = bt;
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    int wpc = cfg.waves_per_cu;
    if (wpc < 0) wpc = 0;
    const bool stage_ok = cfg.stage_reads != 0 && bt == 64 && v == 14;                  // the staged kernels: one-wavefront blocks of the default walk
    // ... on the deep rows where the handle holds them and the batch is one of short reads: three bases per gather pay where reads follow the
    // text (c2: fabric lines per base 0.584 -> 0.477, 80.3 -> 89.6 Gbases/s with reset masks out); 10 kbp reads with 8 % substitutions spend
    // their iterations on repositions, which three-row windows serve worse than four-row ones (59.3 -> 44.4): profiles/r06_deep_rows.txt
    const bool deep_ok = stage_ok && ix.rows3 != nullptr && ix.idx32 != 0u &&
                         (cfg.deep > 0 || (cfg.deep < 0 && n_bases / n_reads < kDeepReadLen));
    const bool ahead_ok = stage_ok && (ix.rows2 != nullptr || deep_ok);                 // ... or on the look-ahead rows
    if (cfg.waves_per_cu == 0 && v == 14 && big_batch)
        wpc = deep_ok ? kCapWavesDeep : (ahead_ok ? kCapWavesAhead : kCapWaves);     // the auto policy above
    P.wpc = wpc;
    // Occupancy cap: enforced by the dispatcher through the block's LDS allocation (160 KiB per CU); blocks beyond
    // the cap queue and start as resident ones retire.
    size_t dyn_lds = 0;
    if (wpc > 0) {
        int bpc = wpc / (bt / 64);
        if (bpc < 1) bpc = 1;
        if (bpc < 32) dyn_lds = ((163840u / (unsigned)bpc) & ~1023u) - 1024u;
    }
    // Reads staged through LDS (pml_kernel_flatp<..., STG = 1>): the block's dynamic LDS holds the next stage_lds bases of each
    // of its 64 reads.  A capped launch has that LDS anyway (the padding: 21 KiB = 336 bases per lane at the default cap of 7
    // wavefronts per CU, 16 KiB = 256 at 9); an uncapped one (a batch of at most ~18 wavefronts per CU, one round) gets what
    // its wavefronts per CU leave of the 160 KiB, so that the round stays one round.  Long reads roll through the same
    // stretch (stage_from in the kernel).  cfg.stage_reads: 1 = whenever it fits (default), 0 = never.
    // PMLs out through a ring in LDS (the kernel has the numbers): launches of long reads -- few wavefronts, each
    // one's own instruction stream most of an iteration -- where the block's LDS holds the ring beside 96 staged bases.
    // cfg.out_ring: -1 = this policy, 0 / 1 = never / wherever it fits (A/B).  (Reset masks out: no PML leaves, no ring.)
    const bool ring_wanted = stage_ok && !want_mask && (cfg.out_ring > 0 || (cfg.out_ring < 0 && n_bases / n_reads >= kOutRingReadLen));
    if (stage_ok && wpc == 0) {
        const uint64_t wn = (blocks + (uint64_t)cfg.num_cus - 1) / (uint64_t)cfg.num_cus;      // wavefronts per CU of this launch
        // (room for a quarter more: the dispatcher does not deal the blocks out evenly, and a CU that may hold no more than the
        // average leaves its surplus queued -- 150 k reads, 9.2 wavefronts per CU: 41.2 Gbases/s with room for 10, 45.8 for 12)
        const uint64_t room = wn + std::max<uint64_t>(2, wn / 4);
        if (wn <= 18) dyn_lds = std::min<size_t>(21504 + (ring_wanted ? kOutRingBytes : 0u), ((163840u / (unsigned)room) & ~1023u) - 1024u);
    }
    const size_t ring_b = (ring_wanted && dyn_lds >= kOutRingBytes + 96u * 64u) ? kOutRingBytes : 0;
    const uint32_t stage_cap = (uint32_t)std::min<size_t>(1024, ((dyn_lds - ring_b) / 64) & ~(size_t)15);
    P.dyn_lds = dyn_lds;
    P.stage_lds = (stage_ok && stage_cap >= 96) ? stage_cap : 0u;
    P.use_ring = ring_b != 0 && P.stage_lds != 0u;
    P.use_deep = deep_ok && P.stage_lds != 0u;
    P.use_ahead = ahead_ok && P.stage_lds != 0u;
    // pair-shared gathers (pml_kernel_flatp<..., PSH = 1>): the staged default walk on the plain or the look-ahead rows
    // Where: on tables beyond the reach of the per-CU TLBs (~2 GB), where a lane's two (four) 16-byte loads are as many
    // translation requests and the L2 TLB's request rate bounds the walk -- real BWT of 226 M rows on the look-ahead rows (3.6 GB
    // copy) 39.4 -> 50.8 Gbases/s, the random 1 B-row table 32.6 -> 34.7 on its plain rows and 21.4 -> 44.2 on the look-ahead
    // copy (16 GB); below that the exchange costs about what the merged accesses give (random 25 / 50 / 100 M rows +4 / +5 / -2 %,
    // real 113 M rows +1.5 %, c2 -2.5 %, c3 -9 %: profiles/r04_pair_shared_gathers.txt).  "pair_loads" 1 / 0 forces it.
    const uint64_t walked_bytes = ix.r * (P.use_ahead ? 16ull : 8ull);
    P.use_pair = (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && walked_bytes >= kPairLoadBytes)) && P.stage_lds != 0u && v == 14 && !P.use_deep;
    return P;
}
}  // namespace

// movi_pml_device's own choice ("pml_via_mask" -1): the vector through reset masks wherever the default walk writes the masks itself and
// the register packer would otherwise write the vector -- batches of short reads (mean length below kOutRingReadLen; long reads keep the
// ring in LDS, which is as good there: c3 16.76 against 16.66 ms).  Measured, vector out, packer -> fused masks: c2 on the deep rows 78.4
// -> 86.7 Gbases/s, the random 10 M-row table 62.4 -> 64.8, the 1 B-row table 42.3 -> 46.0 (profiles/r06_mask_path.txt).
bool pml_vector_via_masks(const DevIndex &ix, const LaunchCfg &cfg, uint64_t n_reads, uint64_t n_bases, bool have_seg_ws, bool ordered) {
    if (n_reads == 0) return false;
    const PmlPlan P = plan_pml(ix, cfg, n_reads, n_bases, 0, false, have_seg_ws, ordered, true);
    return !P.seg_eligible && P.v == 14 && P.stage_lds != 0u && n_bases / n_reads < kOutRingReadLen;
}

bool pml_mask_needs_tmp(const DevIndex &ix, const LaunchCfg &cfg, uint64_t n_reads, uint64_t n_bases, bool have_seg_ws) {
    if (n_reads == 0) return false;
    const PmlPlan P = plan_pml(ix, cfg, n_reads, n_bases, 0, false, have_seg_ws, false, true);
    return P.seg_eligible || P.v != 14 || P.stage_lds == 0u;
}

hipError_t launch_pml(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                      const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream, const ClsArgs &cls,
                      SegWorkspace *seg_ws, int ragged_hint, int *seg_verdict, LaunchInfo *info, const MaskArgs &mask) {
    if (n_reads == 0) return hipSuccess;
    // 0 = PML vector only, 1 = vector + classification bins, 2 = bins only
    bool want_mask = mask.words != nullptr;                // reset masks out instead of the vector (MaskArgs)
    const bool logging0 = cls.log_ff != nullptr || cls.log_scan != nullptr;
    if (want_mask && mask.expand_out) {
        // the caller wants the VECTOR and lends scratch for masks: through masks only where the walk writes them itself
        const PmlPlan P0 = plan_pml(ix, cfg, n_reads, n_bases, 0, logging0, seg_ws != nullptr, d_order != nullptr, true);
        if (P0.seg_eligible || P0.v != 14 || P0.stage_lds == 0u || cls.bin_width != 0) {
            want_mask = false;
            d_out = mask.expand_out;
        }
    }
    if (want_mask) d_out = mask.tmp_pml;                   // (the paths without a mask output of their own write here first)
    const int cm = cls.bin_width == 0 ? 0 : (d_out ? 1 : 2);
    if (cm == 0 && !d_out && !want_mask) return hipErrorInvalidValue;
    if (cm != 0 && (!cls.above || !cls.below || !cls.sum_max || want_mask)) return hipErrorInvalidValue;
    // One resident row layout: blocked- and sampled-thresholds tables are expanded to regular-thresholds rows at upload
    // (expand_blocked_kernel / expand_sampled_kernel), so the query kernels exist for MODE 6 only.
    if (mode != 6) return hipErrorInvalidValue;
    const bool logging = cls.log_ff != nullptr || cls.log_scan != nullptr;
    if (logging && (cm != 0 || want_mask)) return hipErrorInvalidValue;
    const PmlPlan P = plan_pml(ix, cfg, n_reads, n_bases, cm, logging, seg_ws != nullptr, d_order != nullptr, want_mask);
    const int v = P.v;
    if (P.seg_eligible && (!want_mask || d_out)) {
        bool declined = false;
        const hipError_t es = launch_pml_segmented(ix, d_bases, d_offsets, n_reads, n_bases, d_out, d_err, d_stats, cfg, stream,
                                                   seg_ws, cfg.pml_variant < 0 || cfg.pml_variant == 14, ragged_hint, &declined, cls,
                                                   seg_verdict, info);
        if (es != hipSuccess) return es;
        if (!declined) return want_mask ? launch_pml_to_mask(d_out, d_offsets, n_reads, n_bases, mask.phase, mask.words, stream) : hipSuccess;
    }
    const int bt = P.bt;
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block((unsigned)bt);
    size_t dyn_lds = P.dyn_lds;
    DevIndex ixl = ix;
    ixl.stage_lds = P.stage_lds;
    ixl.inwin = cfg.inwin ? 1u : 0u;
    ixl.hint_w = P.use_deep ? (cfg.hints != 0 ? 2u : 0u) : ((cfg.hints != 0 && ix.hints != 0u && ix.rows2 != nullptr) ? 3u : 0u);
    ixl.mask_phase = mask.phase & 31u;
    const bool mask_native = want_mask && v == 14 && P.stage_lds != 0u;
    // the vector from the mask walk itself: one-wavefront blocks whose 64 reads are one contiguous, 16-byte aligned stretch of it
    const bool fused_expand = mask_native && mask.expand_out != nullptr && bt == 64 && d_order == nullptr &&
                              (reinterpret_cast<uintptr_t>(mask.expand_out) & 15u) == 0 && cfg.fused_expand != 0;
    ixl.expand_out = fused_expand ? mask.expand_out : nullptr;
    if (want_mask && !mask_native && !d_out) return hipErrorInvalidValue;     // (pml_mask_needs_tmp told the caller)
    hipError_t e = hipSuccess;
    if (v == 14) {
        WalkLaunch L;
        L.grid = grid; L.block = block; L.dyn_lds = dyn_lds; L.stream = stream;
        L.ix = ixl; L.bases = d_bases; L.offs = d_offsets; L.n = n_reads; L.out = mask_native ? reinterpret_cast<uint16_t *>(mask.words) : d_out;
        L.err = d_err; L.stats = d_stats;
        L.order = d_order; L.cls = cls;
        L.cls_mode = cm; L.sep = ix.sep ? 1 : 0; L.stg = ixl.stage_lds != 0u ? 1 : 0;
        L.ahd = P.use_deep ? 2 : (P.use_ahead ? 1 : 0); L.psh = P.use_pair ? 1 : 0; L.ring = mask_native ? 2 : (P.use_ring ? 1 : 0);
        e = ix.idx32 ? launch_walk_u32(L, info) : launch_walk_u64(L, info);
    } else {
        // the base-synchronous kernels (every one takes at most 64 KiB of dynamic LDS: the cap's padding)
        if (dyn_lds > 65536) dyn_lds = 65536 - 1024;
#define MOVI_LAUNCH_PML(V, C) hipLaunchKernelGGL((pml_kernel<6, V, C>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads, d_out, d_err, d_stats, d_order, cls)
        if (v == 0) MOVI_LAUNCH_PML(0, 0);
        else if (cm == 0) MOVI_LAUNCH_PML(1, 0);
        else if (cm == 1) MOVI_LAUNCH_PML(1, 1);
        else MOVI_LAUNCH_PML(1, 2);
#undef MOVI_LAUNCH_PML
        if (info) snprintf(info->kernel, sizeof(info->kernel), "pml_kernel<6, %d, %d>", v, v == 0 ? 0 : cm);
        e = hipGetLastError();
    }
    if (info) {
        info->variant = v;
        info->block_threads = bt; info->waves_per_cu = P.wpc; info->segmented = 0; info->idx64 = ix.idx32 ? 0 : 1;
        info->staged = (int)ixl.stage_lds;
        info->ahead = P.use_deep ? 2 : (P.use_ahead ? 1 : 0);
    }
    if (e == hipSuccess && want_mask && !mask_native) e = launch_pml_to_mask(d_out, d_offsets, n_reads, n_bases, mask.phase, mask.words, stream);
    if (e == hipSuccess && mask_native && mask.expand_out != nullptr && !fused_expand)
        e = launch_pml_expand(mask.words, d_offsets, n_reads, n_bases, mask.phase, mask.expand_out, stream);
    return e;
}

// update_interval, src/move_structure_search.cpp:48-61 (get_char: the '$' row never equals a base): move the interval's
// start down to the first row of character b and its end up to the last one.  If [rs, re] holds such a row both searches
// find one and start <= end; if it holds none the interval is empty, and that is all the callers use (the reference lets
// the start run past the end instead).  So the two searches are independent, each bounded by the OTHER end's original row,
// and each takes the 4-row window around its next row per trip (window base clamped to r - 4: never outside the table)
// instead of one row: the trips of this loop -- max over the wave's lanes -- were most of a ZML step on divergent reads.
// Row / window / look-ahead entry of the table the count query walks on: AH = 0 the plain rows, AH = 1 the look-ahead copy
// (DevIndex::rows2: 8 rows + their 8 entries per 128-byte line, the last window in a line of its own).
// (The copy's reposition hints -- DevIndex::hints: y[31:28] of a row, y[30:25] of an entry -- are the PML walk's business: masked
// off here, which leaves the 32-bit ids of that form as they are.)
template <int MODE, int AH>
__device__ __forceinline__ uint2 tab_row(const DevIndex &ix, uint64_t i) {
    if (!AH) return load_row<MODE>(ix.rows, i);
    uint2 v;
    __builtin_memcpy(&v, ix.rows2 + (i >> 3) * 128u + (i & 7u) * 8u, 8);
    v.y &= ix.hints ? 0x0FFFFFFFu : 0xFFFFFFFFu;
    return v;
}
__device__ __forceinline__ uint2 tab_entry(const DevIndex &ix, uint64_t i) {
    uint2 v;
    __builtin_memcpy(&v, ix.rows2 + (i >> 3) * 128u + 64u + (i & 7u) * 8u, 8);
    v.y &= ix.hints ? ~(0x3Fu << 25) : 0xFFFFFFFFu;
    return v;
}
template <int MODE, int AH>
__device__ __forceinline__ void tab_window(const DevIndex &ix, uint64_t wb, uint2 (&w)[4]) {   // wb: aligned, or r - 4 (the last window)
    if (!AH) { load_window<MODE>(ix.rows, wb, w); return; }
    load_window<MODE>(ix.rows2 + (wb < ix.r - 4 ? (wb >> 3) * 128u + (wb & 4u) * 8u : ix.rows2_tail), 0, w);
    const uint32_t ym = ix.hints ? 0x0FFFFFFFu : 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t].y &= ym;
}

template <int MODE, int AH = 0>
__device__ __forceinline__ void shrink_interval(const DevIndex &ix, bool act, uint32_t b, uint64_t &rs, uint32_t &os,
                                                uint2 &rws, uint64_t &re, uint32_t &oe, uint2 &rwe,
                                                uint32_t &scan_total) {
    uint32_t gs = 0, ge = 0, dead = 0;
    if (act) {
        gs = (rs == ix.end_bwt_idx || row_c<MODE>(rws) != b) ? 1u : 0u;
        ge = (re == ix.end_bwt_idx || row_c<MODE>(rwe) != b) ? 1u : 0u;
    }
    const uint64_t lo = rs, hi = re, wb_last = ix.r - 4;
    while (wave_any((gs | ge) != 0u)) {
        if (gs) {
            if (rs >= hi) { dead = 1; gs = 0; ge = 0; }                  // no row of b in [lo, hi]
            else {
                uint64_t wb = (rs + 1) & ~3ull;
                if (wb > wb_last) wb = wb_last;
                uint2 w[4];
                tab_window<MODE, AH>(ix, wb, w);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (gs && wb + (uint64_t)t == rs + 1) {
                        rs += 1;
                        scan_total += 1;
                        if (rs != ix.end_bwt_idx && row_c<MODE>(w[t]) == b) { rws = w[t]; gs = 0; }
                        else if (rs >= hi) { dead = 1; gs = 0; ge = 0; }
                    }
                }
                os = 0;
            }
        }
        if (ge) {
            if (re <= lo) { dead = 1; gs = 0; ge = 0; }
            else {
                uint64_t wb = (re - 1) & ~3ull;
                if (wb > wb_last) wb = wb_last;
                uint2 w[4];
                tab_window<MODE, AH>(ix, wb, w);
#pragma unroll
                for (int t = 3; t >= 0; --t) {
                    if (ge && wb + (uint64_t)t + 1 == re) {
                        re -= 1;
                        scan_total += 1;
                        if (re != ix.end_bwt_idx && row_c<MODE>(w[t]) == b) { rwe = w[t]; oe = row_n<MODE>(w[t]) - 1; ge = 0; }
                        else if (re <= lo) { dead = 1; gs = 0; ge = 0; }
                    }
                }
            }
        }
    }
    if (dead) { rs = 1; re = 0; os = 0; oe = 0; }                        // empty, whatever the rows were
}

// The same, one row per end and trip (tables too small for a window).
template <int MODE>
__device__ __forceinline__ void shrink_interval_rows(const DevIndex &ix, bool act, uint32_t b, uint64_t &rs, uint32_t &os,
                                                     uint2 &rws, uint64_t &re, uint32_t &oe, uint2 &rwe,
                                                     uint32_t &scan_total) {
    uint32_t gs = 0, ge = 0;
    if (act) {
        gs = (rs == ix.end_bwt_idx || row_c<MODE>(rws) != b) ? 1u : 0u;
        ge = (re == ix.end_bwt_idx || row_c<MODE>(rwe) != b) ? 1u : 0u;
    }
    while (wave_any((gs | ge) != 0u)) {
        uint2 ws = rws, we = rwe;
        if (gs && rs + 1 < ix.r) ws = load_row<MODE>(ix.rows, rs + 1);
        if (ge && re > 0) we = load_row<MODE>(ix.rows, re - 1);
        if (gs) {
            rs += 1; os = 0; scan_total += 1;
            if (rs >= ix.r || rs > re) { gs = 0; ge = 0; }
            else { rws = ws; gs = (rs == ix.end_bwt_idx || row_c<MODE>(rws) != b) ? 1u : 0u; }
        }
        if (ge) {
            if (re == 0) { ge = 0; gs = 0; rs = 1; }          // nothing above row 0: empty
            else {
                re -= 1; scan_total += 1;
                rwe = we;
                oe = row_n<MODE>(rwe) - 1;
                if (re < rs) { ge = 0; gs = 0; }
                else ge = (re == ix.end_bwt_idx || row_c<MODE>(rwe) != b) ? 1u : 0u;
            }
        }
    }
}

// ----------------------------------------------------------------------- count

// BWT position of (row k, offset 0) from the 32-row checkpoints.
template <int MODE>
__device__ __forceinline__ uint64_t row_start(const DevIndex &ix, uint64_t k) {
    uint64_t j = (k >> kPrefixShift) << kPrefixShift;
    uint64_t p = ix.row_start_ckpt[k >> kPrefixShift];
    for (; j < k; ++j) p += row_n<MODE>(load_row<MODE>(ix.rows, j));
    return p;
}

// Interval table of the count query (DevIndex::ftab): lane t runs the backward search of the K-mer whose i-th consumed base
// (i = 0: the read's last base) has code (t >> 2 i) & 3 (+ 1 on a separators index) -- exactly count_kernel_v0's steps -- and
// records the interval after K bases, if it is still non-empty.
template <int MODE>
__global__ __launch_bounds__(256) void ftab_kernel(DevIndex ix, uint32_t K, uint4 *__restrict__ table) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n = 1ull << (2 * K);
    const bool valid = t < n;
    uint32_t ff_total = 0, scan_total = 0, failed = 0;
    const uint32_t a = (uint32_t)(t & 3u) + ix.sep;
    uint64_t rs = ix.first_runs[a + 1], re = ix.last_runs[a + 1];
    uint32_t os = (uint32_t)ix.first_offsets[a + 1], oe = (uint32_t)ix.last_offsets[a + 1];
    uint32_t run = (valid && ((rs < re) || (rs == re && os <= oe))) ? 1u : 0u;
    uint2 rws = make_uint2(0, 0), rwe = make_uint2(0, 0);
    if (run) {
        rws = load_row<MODE>(ix.rows, rs);
        rwe = load_row<MODE>(ix.rows, re);
    }
    for (uint32_t i = 1; i < K; ++i) {                    // K is wave-uniform
        const bool act = run != 0u;
        const uint32_t b = (uint32_t)((t >> (2 * i)) & 3u) + ix.sep;
        if (ix.r >= 8) shrink_interval<MODE>(ix, act && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
        else shrink_interval_rows<MODE>(ix, act && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
        bool nonempty = act && ((rs < re) || (rs == re && os <= oe));
        if (act && !nonempty) run = 0;
        const uint32_t e12 = lf_step2<MODE>(ix, nonempty, rs, os, rws, re, oe, rwe, ff_total);
        if (e12) { failed = e12; run = 0; nonempty = false; }
        if (nonempty && !((rs < re) || (rs == re && os <= oe))) run = 0;
    }
    if (!valid) return;
    const uint32_t ok = (uint32_t)(run != 0u && failed == 0u && ff_total < (1u << 15) && scan_total < (1u << 16) && os < 4096u && oe < 4096u);
    uint4 e4 = make_uint4(0, 0, 0, 0);
    if (ok) {
        e4.x = (uint32_t)rs;
        e4.y = (uint32_t)re;
        e4.z = (uint32_t)(rs >> 32) | ((uint32_t)(re >> 32) << 4) | (os << 8) | (oe << 20);
        e4.w = ff_total | (scan_total << 15) | (1u << 31);
    }
    table[t] = e4;
}

hipError_t build_ftab(int mode, const DevIndex &ix, uint32_t K, uint4 *d_table, hipStream_t stream) {
    if (K < 1 || K > 12 || !d_table || ix.sigma - ix.sep != 4) return hipErrorInvalidValue;
    const uint64_t n = 1ull << (2 * K);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (mode == 6) hipLaunchKernelGGL(ftab_kernel<6>, grid, block, 0, stream, ix, K, d_table);
    else if (mode == 3) hipLaunchKernelGGL(ftab_kernel<3>, grid, block, 0, stream, ix, K, d_table);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// The two ends of the interval arrive at their LF targets (ta, tb) with offsets (offa, offb): both row gathers -- on the
// look-ahead copy with the rows' entries -- and the shared fast-forward loop of lf_step2.  An end that fast-forwards loses
// its entry (it belongs to the row the gather was aimed at).
template <int MODE>
__device__ __forceinline__ uint32_t arrive2_ahead(const DevIndex &ix, bool live, uint64_t ta, uint64_t tb, uint64_t &ia, uint32_t &offa,
                                                  uint2 &rowa, uint2 &enta, uint64_t &ib, uint32_t &offb, uint2 &rowb, uint2 &entb,
                                                  uint32_t &ff_total) {
    uint32_t na = 0, nb = 0, ffa = 0, ffb = 0, ga = 0, gb = 0;
    uint64_t ja = ia, jb = ib;
    if (live) {
        ja = ta; jb = tb;
        rowa = tab_row<MODE, 1>(ix, ja);
        rowb = tab_row<MODE, 1>(ix, jb);
        enta = tab_entry(ix, ja);
        entb = tab_entry(ix, jb);
        na = row_n<MODE>(rowa);
        nb = row_n<MODE>(rowb);
        ga = (ja < ix.r - 1 && offa >= na) ? 1u : 0u;
        gb = (jb < ix.r - 1 && offb >= nb) ? 1u : 0u;
        if (ga) enta = make_uint2(0u, 0u);
        if (gb) entb = make_uint2(0u, 0u);
    }
    while (wave_any((ga | gb) != 0u)) {                 // fast_forward :524-545, both walkers
        uint2 wa = rowa, wb = rowb;
        if (ga) wa = tab_row<MODE, 1>(ix, ja + 1);
        if (gb) wb = tab_row<MODE, 1>(ix, jb + 1);
        if (ga) {
            offa -= na; ja += 1; ffa += 1; rowa = wa; na = row_n<MODE>(rowa);
            ga = (ja < ix.r - 1 && offa >= na && ffa < 65535u) ? 1u : 0u;
        }
        if (gb) {
            offb -= nb; jb += 1; ffb += 1; rowb = wb; nb = row_n<MODE>(rowb);
            gb = (jb < ix.r - 1 && offb >= nb && ffb < 65535u) ? 1u : 0u;
        }
    }
    ff_total += ffa + ffb;
    ia = ja; ib = jb;
    return (ffa >= 65535u || ffb >= 65535u) ? kErrFastForward : kErrNone;   // move_structure.cpp:72-75
}

// AH = 1 (look-ahead rows, DevIndex::rows2; MODE 6): the search walks on the table's second copy and every end carries the
// look-ahead entry of its row.  After the interval has been shrunk to the rows of base b, the NEXT base b2 is looked at:
// if both ends' LF targets hold b2 (neither is the '$' row) and both arrive there without a fast-forward -- all of it in
// the entries -- the step after this one needs no shrink and no rows: update_interval + two LF moves twice
// (src/move_structure_search.cpp:311-333 run for b and for b2), two bases for one pair of gathers.  The interval after b --
// what the reference reports if the one after b2 came out empty -- is known from the entries too.
template <int MODE, int AH = 0>
__global__ __launch_bounds__(256) void count_kernel_v0(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint64_t *__restrict__ matched,
                                                       uint64_t *__restrict__ count, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, failed = 0;
    const bool valid = t < n_reads;
    const uint64_t rid = (valid && order) ? order[t] : t;
    const uint64_t beg = valid ? offs[rid] : 0;
    const int64_t len = valid ? (int64_t)(offs[rid + 1] - beg) : 0;
    const uint8_t *R = bases + beg;
    int64_t pos = len - 1;
    // interval [rs:os, re:oe] and the previous one (MoveInterval, include/move_intervals.hpp:10-76)
    uint64_t rs = 0, re = 0, prs = 0, pre = 0;
    uint32_t os = 0, oe = 0, pos_ = 0, poe = 0;
    uint32_t run = 0;                 // 1 while the backward search of this lane continues
    uint32_t have = 0;                // 1 when an interval exists (last base legal)
    if (len > 0) {
        const uint32_t a = s_code[R[pos]];
        if (a != 0xFFu) {             // else: move_structure_search.cpp:344-347 -> "0/len 0"
            // initialize_backward_search :284-291
            rs = ix.first_runs[a + 1]; re = ix.last_runs[a + 1];
            os = (uint32_t)ix.first_offsets[a + 1]; oe = (uint32_t)ix.last_offsets[a + 1];
            have = 1;
            run = ((rs < re) || (rs == re && os <= oe)) ? 1u : 0u;
        }
    }
    // ---- interval table (DevIndex::ftab): the first K bases of the search by one lookup, when all K are legal and the
    // K-mer occurs; anything else starts the ordinary way
    if (ix.ftab_k != 0u) {                                // wave-uniform
        const uint32_t K = ix.ftab_k;
        uint32_t kidx = 0, bad = (uint32_t)(len < (int64_t)K);
        for (uint32_t i = 0; i < K; ++i) {
            const uint32_t cc = (bad ? 0xFFu : (uint32_t)s_code[R[len - 1 - (int64_t)i]]) - ix.sep;
            bad |= (uint32_t)(cc > 3u);
            kidx |= (cc & 3u) << (2u * i);
        }
        uint4 e4 = make_uint4(0, 0, 0, 0);
        if (!bad) e4 = ix.ftab[kidx];
        if (e4.w >> 31) {
            rs = (uint64_t)e4.x | ((uint64_t)(e4.z & 15u) << 32);
            re = (uint64_t)e4.y | ((uint64_t)((e4.z >> 4) & 15u) << 32);
            os = (e4.z >> 8) & 0xFFFu;
            oe = e4.z >> 20;
            ff_total = e4.w & 0x7FFFu;
            scan_total = (e4.w >> 15) & 0xFFFFu;
            pos = len - (int64_t)K;
            have = 1;
            run = 1;
        }
    }
    prs = rs; pre = re; pos_ = os; poe = oe;
    uint32_t empty = have && !run;
    // rows[rs], rows[re]: loaded for the first step, afterwards carried over from the LF moves
    // (lf_step2 leaves the rows of the new ends in rws / rwe), one dependent trip less per base
    uint2 rws = make_uint2(0, 0), rwe = make_uint2(0, 0), ens = make_uint2(0, 0), ene = make_uint2(0, 0);
    if (run) {
        rws = tab_row<MODE, AH>(ix, rs);
        rwe = tab_row<MODE, AH>(ix, re);
        if (AH) { ens = tab_entry(ix, rs); ene = tab_entry(ix, re); }
    }
    while (wave_any(run != 0u && pos > 0)) {             // backward_search :176
        const bool act = run != 0u && pos > 0;
        uint32_t b = 0xFFu, b2 = 0xFFu;
        if (act) {
            prs = rs; pre = re; pos_ = os; poe = oe;
            const uint32_t byte2 = (AH && pos > 1) ? (uint32_t)R[pos - 2] : 0u;   // (AH) the base after this one, fetched with it
            b = s_code[R[pos - 1]];
            if (AH && pos > 1) b2 = s_code[byte2];
            if (b == 0xFFu) { empty = 1; run = 0; }       // backward_search_step :321-324
        }
        const bool legal = act && b != 0xFFu;
        const uint64_t rs_in = rs, re_in = re;
        // update_interval, src/move_structure_search.cpp:48-61 (get_char: '$' never equals a base)
        // Both ends shrink in ONE wave-uniform loop, one row per end and trip.  When the interval
        // holds no row of character b the two ends cross instead of rs running all the way past re as
        // in the reference; either way the interval is empty and the previous one is reported.
        if (AH || ix.r >= 8) {                            // (the look-ahead copy exists for tables of 8 rows and more only)
            shrink_interval<MODE, AH>(ix, legal && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
        } else {
            shrink_interval_rows<MODE>(ix, legal && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
        }
        bool nonempty = legal && ((rs < re) || (rs == re && os <= oe));
        if (legal && !nonempty) { empty = 1; run = 0; }
        if (!AH) {
            // backward_search_step :326-330: two LF moves
            const uint32_t e12 = lf_step2<MODE>(ix, nonempty, rs, os, rws, re, oe, rwe, ff_total);
            if (e12) { failed = e12; run = 0; nonempty = false; }
            if (nonempty) {                               // backward_search :179-182
                if ((rs < re) || (rs == re && os <= oe)) pos -= 1;
                else { empty = 1; run = 0; }
            }
        } else {
            if (rs != rs_in) ens = make_uint2(0u, 0u);    // an end that moved: its entry belonged to the row it left
            if (re != re_in) ene = make_uint2(0u, 0u);    // (the shrink loads rows only)
            // backward_search_step :326-330: the two LF moves of base b ...
            uint64_t ta = rs, tb = re;
            uint32_t two = 0;
            if (nonempty) {
                ta = row_id<MODE>(rws, rs, ix);
                tb = row_id<MODE>(rwe, re, ix);
                if (ta >= ix.r || tb >= ix.r) {           // move_structure.cpp:63-65
                    failed = kErrIdRange; run = 0; nonempty = false;
                } else {
                    os += row_off<MODE>(rws);
                    oe += row_off<MODE>(rwe);
                    // ... and, from the entries, base b2's whole step: both targets hold b2, no fast-forward at either
                    const uint32_t n1s = ens.y & 0x7FFu, n1e = ene.y & 0x7FFu, c1s = (ens.y >> 22) & 7u, c1e = (ene.y >> 22) & 7u;
                    two = (ens.y >> 31) & (ene.y >> 31) & (uint32_t)(b2 != 0xFFu) & (uint32_t)(c1s == b2) & (uint32_t)(c1e == b2) &
                          (uint32_t)(os < n1s) & (uint32_t)(oe < n1e) & (uint32_t)(ta != ix.end_bwt_idx) & (uint32_t)(tb != ix.end_bwt_idx) &
                          (uint32_t)((ta < tb) || (ta == tb && os <= oe));
                    if (two) {
                        prs = ta; pre = tb; pos_ = os; poe = oe;           // the interval after b: reported if b2's comes out empty
                        os += (ens.y >> 11) & 0x7FFu;
                        oe += (ene.y >> 11) & 0x7FFu;
                        ta = (uint64_t)ens.x | ((uint64_t)((ens.y >> 25) & 15u) << 32);
                        tb = (uint64_t)ene.x | ((uint64_t)((ene.y >> 25) & 15u) << 32);
                    }
                }
            }
            const uint32_t e12 = arrive2_ahead<MODE>(ix, nonempty, ta, tb, rs, os, rws, ens, re, oe, rwe, ene, ff_total);
            if (e12) { failed = e12; run = 0; nonempty = false; }
            if (nonempty) {                               // backward_search :179-182, once or twice
                if ((rs < re) || (rs == re && os <= oe)) pos -= 1 + (int64_t)two;
                else { empty = 1; run = 0; pos -= (int64_t)two; }
            }
        }
    }
    if (valid) {
        uint64_t m_out = 0, c_out = 0;
        if (have && !failed) {
            if (empty) { rs = prs; re = pre; os = pos_; oe = poe; }
            m_out = (uint64_t)(len - pos);
            // MoveInterval::count, include/move_intervals.hpp:47-58, via the row-start checkpoints
            if (rs == re) c_out = (uint64_t)oe - os + 1;
            else c_out = (row_start<MODE>(ix, re) + oe) - (row_start<MODE>(ix, rs) + os) + 1;
        }
        matched[rid] = m_out;
        count[rid] = c_out;
        if (err) err[rid] = (uint8_t)failed;
    }
    const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), erw = wave_sum(failed ? 1u : 0u);
    if ((threadIdx.x & 63) == 0 && stats) {
        if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
        if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
        if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
    }
}

// the count query as a lane state machine (zml_kernel_flat<..., CNT = 1>, with the ZML kernels below)
static hipError_t launch_count_flat(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                    uint64_t *d_matched, uint64_t *d_count, uint8_t *d_err, DevStats *d_stats, const uint32_t *d_order,
                                    const LaunchCfg &cfg, hipStream_t stream, LaunchInfo *info, bool pair);

hipError_t launch_count(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                        uint64_t n_reads, uint64_t *d_matched, uint64_t *d_count, uint8_t *d_err,
                        DevStats *d_stats, const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream,
                        LaunchInfo *info, uint64_t n_bases) {
    if (n_reads == 0) return hipSuccess;
    // Blocks of one wavefront; on a cache-resident table (up to the 256 MiB of the Infinity Cache) and a batch of more than
    // ~24 wavefronts of reads per CU at most kCountCapWaves wavefronts resident per CU -- the same L2-retention effect as in
    // launch_pml: the search's neighbour rows (interval shrink, fast-forwards) must survive between a lane's steps.
    // profiles/r03_count_ftab.txt: pangenome 61.2 -> 67.5 Gbases/s (cap 15 - 16; 14: 66.2, 17: 64.2, 20: 62.5), random 80 MB
    // table 52.4 -> 54.6; HBM-resident tables lose with any cap (1.6 GB: 47.3 uncapped, 43.2 at 16) or are indifferent (8 GB).
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : 64;
    int wpc = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : 0;
    const bool ahead = mode == 6 && ix.rows2 != nullptr && ix.rows2_count != 0u;   // the search walks on the look-ahead rows where they pay
    // Round 5 -- the search as a LANE STATE MACHINE over row windows (zml_kernel_flat<..., CNT = 1>; by pairs of lanes on plain rows of
    // 2 GB and more) wherever it can run: tables of 8 rows and more, batches of 16 bases and more, one-wavefront blocks.  Gbases/s of
    // read bases, count_kernel_v0 -> the state machine (profiles/r05_c5_count_pmc.txt): the 1 B-row blocked-thresholds table of BASELINE
    // config 5 37.2 -> 56.4 - 56.9 (without the pairs 32.3), random 200 M rows 46.6 -> 68.0, the c2 pangenome 71.4 (on its look-ahead
    // rows) -> 84.0 (on the plain rows).  cfg.count_variant: -1 = this policy, 0 = count_kernel_v0 (A/B; tiny tables and batches), 1 = the
    // state machine or nothing.
    const bool flat_ok = (mode == 6 || mode == 3) && ix.r >= 8 && n_bases >= 16 && bt == 64;
    const bool big = ix.r * 8ull >= kPairLoadBytes;
    if (flat_ok && cfg.count_variant != 0)
        return launch_count_flat(mode, ix, d_bases, d_offsets, n_reads, d_matched, d_count, d_err, d_stats, d_order, cfg, stream, info,
                                 cfg.pair_loads > 0 || (cfg.pair_loads < 0 && big));
    if (cfg.count_variant > 0) return hipErrorInvalidValue;
    if (cfg.waves_per_cu == 0 && ix.r * (ahead ? 16ull : 8ull) <= (256ull << 20) &&   // (the bytes of the table the search walks on)
        n_reads > (uint64_t)cfg.num_cus * 64ull * 24ull) wpc = kCountCapWaves;
    size_t dyn_lds = 0;
    if (wpc > 0) {
        int bpc = wpc / (bt / 64);
        if (bpc < 1) bpc = 1;
        if (bpc < 32) dyn_lds = std::min<size_t>(65536 - 1024, ((163840u / (unsigned)bpc) & ~1023u) - 1024u);
    }
    if (info) {
        snprintf(info->kernel, sizeof(info->kernel), "count_kernel_v0<%d, %d>", mode, ahead ? 1 : 0);
        info->variant = 0; info->block_threads = bt; info->waves_per_cu = wpc; info->segmented = 0; info->idx64 = 1; info->staged = 0;
        info->ahead = ahead ? 1 : 0;
    }
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block((unsigned)bt);
    // resident layouts: 6 = regular-thresholds rows, 3 = regular rows (threshold-less types: 12-bit lengths)
    if (mode == 6 && ahead)
        hipLaunchKernelGGL((count_kernel_v0<6, 1>), grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads,
                           d_matched, d_count, d_err, d_stats, d_order);
    else if (mode == 6)
        hipLaunchKernelGGL(count_kernel_v0<6>, grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads,
                           d_matched, d_count, d_err, d_stats, d_order);
    else if (mode == 3)
        hipLaunchKernelGGL(count_kernel_v0<3>, grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads,
                           d_matched, d_count, d_err, d_stats, d_order);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------- ZML
// MoveStructure::query_zml, src/move_structure_query.cpp:690-785 (Ziv-Merhav cross parse): a
// greedy backward search that restarts at the next base whenever the match cannot be extended.
// The reference emits, for position pos, match_len BEFORE trying to extend to pos-1; restated
// per consumed base (emission step k handles base len-1-k):
//   phrase open  : extend the interval with the base (update_interval + 2 LF); still non-empty
//                  -> ml += 1, else the phrase ends, ml = 0 and this base opens the next phrase
//   no phrase    : ml = 0; a legal base opens a phrase (initialize_backward_search :284-291)
//   emit min(ml, 65535)
// which produces the same vector (value at a base = bases of its phrase to its right).  Every
// step consumes exactly one base, so the packed I/O of PML variant 1 carries over: 16 bases per
// fetch, values leave as paired 16-byte stores.  The walkers are the count query's.
// One base of query_zml (src/move_structure_query.cpp:690-785) for the lanes with `live`: a base either extends the open
// phrase (update_interval + two LF moves, ml += 1) or ends it (ml = 0) and opens the next one from the first / last run
// tables.  State: the interval [rs:os, re:oe] with the rows of its two ends, `open`, ml.  Wave-uniform loops inside:
// every lane of the wavefront must make the call.  Returns a kErr* code.
template <int MODE>
__device__ __forceinline__ uint32_t zml_base(const DevIndex &ix, bool live, uint32_t b, uint32_t &open, uint64_t &rs, uint32_t &os,
                                             uint2 &rws, uint64_t &re, uint32_t &oe, uint2 &rwe, uint32_t &ml, uint32_t &ff_total,
                                             uint32_t &scan_total) {
    uint32_t failed = 0;
    // backward_search_step, src/move_structure_search.cpp:311-333, for lanes with an open phrase
    const bool ext = live && open != 0u && b != 0xFFu;
    if (ix.r >= 8) {                     // update_interval :48-61, as in count_kernel_v0
        shrink_interval<MODE>(ix, ext && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
    } else {
        shrink_interval_rows<MODE>(ix, ext && rs <= re, b, rs, os, rws, re, oe, rwe, scan_total);
    }
    bool nonempty = ext && ((rs < re) || (rs == re && os <= oe));
    const uint32_t e12 = lf_step2<MODE>(ix, nonempty, rs, os, rws, re, oe, rwe, ff_total);
    if (e12) { failed = e12; nonempty = false; }
    if (nonempty && !((rs < re) || (rs == re && os <= oe))) nonempty = false;   // query_zml :717
    if (live && failed == 0u) {
        if (nonempty) {
            ml += 1;                                  // :718-720
        } else {
            ml = 0;                                   // :750-760, or no phrase yet (:696-704)
            open = 0;
            if (b != 0xFFu) {                         // this base opens the next phrase
                rs = ix.first_runs[b + 1]; re = ix.last_runs[b + 1];
                os = (uint32_t)ix.first_offsets[b + 1]; oe = (uint32_t)ix.last_offsets[b + 1];
                open = ((rs < re) || (rs == re && os <= oe)) ? 1u : 0u;
                if (open) {                           // the LF moves carry the rows over from here on
                    rws = load_row<MODE>(ix.rows, rs);
                    rwe = load_row<MODE>(ix.rows, re);
                }
            }
        }
    }
    return failed;
}

// SEG (segment-parallel long reads, as for PML): 0 = a lane parses a read; 1 = a lane parses one SEGMENT of a read from
// the state every read starts in, leaves a checkpoint of its state every 32 bases and its final state, and reports
// through its segment's record instead of err[] / the global counters; 2 = whole reads again, only those in read_fail.
template <int MODE, int SEG = 0>
__global__ __launch_bounds__(256) void zml_kernel(DevIndex ix, const uint8_t *__restrict__ bases,
                                                  const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                  uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                  DevStats *stats, const uint32_t *__restrict__ order, ZSegArgs seg) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, failed = 0;
    const bool valid = SEG == 1 ? t < *seg.n_seg : (t < n_reads && (SEG != 2 || seg.read_fail[t] != 0));
    const uint64_t rid = (valid && order && SEG == 0) ? order[t] : t;
    const uint64_t beg = valid ? (SEG == 1 ? seg.seg_in[rid] : offs[rid]) : 0;
    const uint64_t len = valid ? (SEG == 1 ? (uint64_t)seg.seg_len[rid] : offs[rid + 1] - beg) : 0;
    const uint64_t obeg = (SEG == 1 && valid) ? seg.seg_out[rid] : beg;
    const uint8_t *R = bases + beg;
    uint16_t *O = out + obeg;
    uint64_t rs = 0, re = 0;                              // MoveInterval [rs:os, re:oe]
    uint32_t os = 0, oe = 0;
    uint32_t open = 0;                                    // 1 while a phrase (non-empty interval) exists
    uint2 rws = make_uint2(0, 0), rwe = make_uint2(0, 0); // rows[rs], rows[re] while a phrase is open
    uint32_t ml = 0;
    uint64_t rb = 0, rb_next = 0;
    uint32_t have16 = 0;
    uint4 pk = make_uint4(0, 0, 0, 0), pk_old = pk;
    const uint64_t packed_end = len & ~7ull;
    for (uint64_t k = 0; wave_any(k < len && failed == 0u); ++k) {
        const bool live = k < len && failed == 0u;
        if ((k & 7) == 0) {                               // base window, as in pml_kernel VARIANT 1
            if ((k & 8) == 0 && live && k + 16 <= len) {
                uint64_t two[2];
                __builtin_memcpy(two, R + (len - 16 - k), 16);
                rb = two[1];
                rb_next = two[0];
                have16 = 1;
            } else if ((k & 8) != 0 && have16) {
                rb = rb_next;
                have16 = 0;
            } else if (live && k + 8 <= len) {
                __builtin_memcpy(&rb, R + (len - 8 - k), 8);
            } else if (live) {
                rb = 0;
                for (uint64_t i = 0; i < len - k; ++i) rb |= (uint64_t)R[len - 1 - k - i] << (8 * (7 - i));
            }
        }
        uint32_t b = 0xFFu;
        if (live) b = s_code[(uint32_t)(rb >> (8 * (7 - (k & 7)))) & 0xFFu];
        {
            const uint32_t ez = zml_base<MODE>(ix, live, b, open, rs, os, rws, re, oe, rwe, ml, ff_total, scan_total);
            if (ez) failed = ez;
        }
        if (SEG == 1 && live && failed == 0u) {
            if ((k & 31ull) == 31ull) {
                ZSegCkpt ck;
                ck.rs = rs; ck.re = re; ck.os = os; ck.oe = oe; ck.ml = ml; ck.open = open; ck.ff = ff_total; ck.scan = scan_total;
                seg.ckpt[(obeg + k) >> 5] = ck;
            }
            if (k + 1 == len) {
                ZSegFin fn;
                fn.rs = rs; fn.re = re; fn.os = os; fn.oe = oe; fn.ml = ml; fn.open = open;
                seg.fin[rid] = fn;
            }
        }
        const uint32_t val = ml > 65535u ? 65535u : ml;   // MoveQuery::add_ml
        if (live && k >= packed_end) {
            O[k] = (uint16_t)val;
        } else if (live) {
            pk.x = (pk.x >> 16) | (pk.y << 16);
            pk.y = (pk.y >> 16) | (pk.z << 16);
            pk.z = (pk.z >> 16) | (pk.w << 16);
            pk.w = (pk.w >> 16) | (val << 16);
            if ((k & 15) == 7) {
                if (k + 8 < packed_end) pk_old = pk;
                else __builtin_memcpy(O + (k - 7), &pk, 16);
            } else if ((k & 15) == 15) {
                __builtin_memcpy(O + (k - 15), &pk_old, 16);
                __builtin_memcpy(O + (k - 7), &pk, 16);
            }
        }
    }
    if (SEG == 1) {
        if (valid) {
            SegTot tt;
            tt.ff = ff_total; tt.scan = scan_total; tt.repo = 0; tt.flag = failed;
            seg.tot[rid] = tt;
        }
        return;
    }
    if (failed) {
        for (uint64_t k = 0; k < len; ++k) O[k] = 0;
    }
    if (valid && err) err[rid] = (uint8_t)failed;
    const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), erw = wave_sum(failed ? 1u : 0u);
    if ((threadIdx.x & 63) == 0 && stats) {
        if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
        if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
        if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
    }
}

// K2 of the segmented ZML parse: one lane per segment boundary continues the parse of the segment before until its state
// -- interval, open flag, match length -- equals a checkpoint of the speculative parse (usually at the first phrase both
// open at the same base).  PASS 0 finds the meeting point, PASS 1 writes the stretches of the lanes on a chain
// (seg_stitch_kernel has the reasoning).
template <int MODE, int PASS>
__global__ __launch_bounds__(256) void zml_stitch_kernel(DevIndex ix, const uint8_t *__restrict__ bases, ZSegArgs seg,
                                                        const uint32_t *__restrict__ seg_j, const uint32_t *__restrict__ seg_rem,
                                                        uint32_t max_over, const uint8_t *__restrict__ on_chain,
                                                        uint16_t *__restrict__ out, SegJoin *__restrict__ join) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = s < *seg.n_seg && seg_j[s] != 0 && (PASS == 0 || on_chain[s] == 2);
    uint32_t ff_total = 0, scan_total = 0, failed = 0, how = 0;
    bool live0 = mine;
    if (PASS == 0 && mine && (seg.tot[s].flag != 0u || seg.tot[s - 1].flag != 0u)) live0 = false;
    const uint32_t T = live0 ? seg.seg_len[s] : 1u;
    const uint32_t rem = live0 ? seg_rem[s] : 0u;
    const uint64_t len = PASS == 0 ? (rem < max_over ? rem : max_over) : (live0 ? join[s].kend + 1u : 0u);
    const uint8_t *R = bases + (live0 ? seg.seg_in[s] + T : 0);          // one past this segment's first base
    const uint64_t obeg = live0 ? seg.seg_out[s] : 0;
    uint16_t *O = out + obeg;
    uint64_t rs = 0, re = 0;
    uint32_t os = 0, oe = 0, ml = 0, open = 0, kend = 0;
    uint2 rws = make_uint2(0, 0), rwe = make_uint2(0, 0);
    SegJoin res{};
    if (live0) {
        const ZSegFin f = seg.fin[s - 1];
        rs = f.rs; re = f.re; os = f.os; oe = f.oe; ml = f.ml; open = f.open;
        if (open) {
            rws = load_row<MODE>(ix.rows, rs);
            rwe = load_row<MODE>(ix.rows, re);
        }
    }
    for (uint64_t k = 0; wave_any(k < len && failed == 0u && how == 0u); ++k) {
        const bool live = k < len && failed == 0u && how == 0u;
        uint32_t b = 0xFFu;
        if (live) b = s_code[*(R - 1 - (int64_t)k)];
        const uint32_t ez = zml_base<MODE>(ix, live, b, open, rs, os, rws, re, oe, rwe, ml, ff_total, scan_total);
        if (ez) failed = ez;
        if (PASS == 1) {
            if (live && failed == 0u) O[k] = (uint16_t)(ml > 65535u ? 65535u : ml);
        } else if (live && failed == 0u) {
            if (k < T) O[k] = (uint16_t)(ml > 65535u ? 65535u : ml);
            kend = (uint32_t)k;
            if ((k & 31ull) == 31ull) {
                const ZSegCkpt c = seg.ckpt[(obeg + k) >> 5];
                // (an interval is only compared while a phrase is open: a closed one holds stale rows)
                if (c.ml == ml && c.open == open && (open == 0u || (c.rs == rs && c.re == re && c.os == os && c.oe == oe))) {
                    const uint32_t ds = (uint32_t)(k / T);
                    const SegTot spec = seg.tot[s + ds];
                    how = spec.flag == 0u ? 1u : 0u;
                    if (spec.flag != 0u) failed = spec.flag;
                    res.ff = ff_total + (spec.ff - c.ff); res.scan = scan_total + (spec.scan - c.scan); res.repo = 0; res.segs = ds;
                }
            }
            if (how == 0u && failed == 0u && k + 1 == rem) {
                how = 2u;
                res.ff = ff_total; res.scan = scan_total; res.repo = 0; res.segs = (uint32_t)(k / T);
            }
        }
    }
    if (PASS == 0 && mine) {
        res.kend = kend;
        res.how = how;
        join[s] = res;
    }
}

// The probe of the segmented ZML parse: two speculative parses per sampled position, `lead` bases apart (seg_probe_kernel).
template <int MODE>
__global__ __launch_bounds__(256) void zml_probe_kernel(DevIndex ix, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs,
                                                       uint64_t n_reads, uint32_t lead, uint32_t reach, uint32_t *__restrict__ tally) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, n_lanes = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t stride = n_reads > n_lanes ? n_reads / n_lanes : 1, per = n_reads < n_lanes ? n_lanes / n_reads : 1;
    const uint64_t rid = per > 1 ? t % n_reads : t * stride, slot = per > 1 ? t / n_reads : 0;
    bool valid = rid < n_reads && slot < per;
    const uint64_t beg = valid ? offs[rid] : 0, len = valid ? offs[rid + 1] - beg : 0;
    const uint64_t mid = len * (2 * slot + 1) / (2 * per);
    valid = valid && mid >= reach && mid + lead <= len;
    const uint8_t *R = bases + beg + mid + lead;
    uint64_t ars = 0, are = 0, brs = 0, bre = 0;
    uint32_t aos = 0, aoe = 0, bos = 0, boe = 0, aml = 0, bml = 0, aop = 0, bop = 0, ffx = 0, scx = 0, failed = 0, met = 0;
    uint2 arws = make_uint2(0, 0), arwe = arws, brws = arws, brwe = arws;
    for (uint32_t k = 0; wave_any(valid && k < lead + reach && failed == 0u && met == 0u); ++k) {
        const bool live = valid && k < lead + reach && failed == 0u && met == 0u;
        uint32_t b = 0xFFu;
        if (live) b = s_code[*(R - 1 - (int64_t)k)];
        const uint32_t e1 = zml_base<MODE>(ix, live, b, bop, brs, bos, brws, bre, boe, brwe, bml, ffx, scx);
        const uint32_t e2 = zml_base<MODE>(ix, live && k >= lead, b, aop, ars, aos, arws, are, aoe, arwe, aml, ffx, scx);
        if (e1 | e2) failed = 1;
        if (live && failed == 0u && k >= lead && aml == bml && aop == bop &&
            (aop == 0u || (ars == brs && are == bre && aos == bos && aoe == boe)))
            met = 1;
        if (k == lead + reach / 2 && 2 * __popcll(__ballot(met != 0u)) < __popcll(__ballot(valid))) break;
        if (k == lead + reach / 6 && 10 * __popcll(__ballot(met != 0u)) < __popcll(__ballot(valid))) break;
        if (k > lead && (k & 15u) == 15u && 20 * __popcll(__ballot(met != 0u)) >= 19 * __popcll(__ballot(valid))) break;
    }
    const uint32_t nv = wave_sum(valid ? 1u : 0u), nm = wave_sum(met);
    if ((threadIdx.x & 63) == 0) {
        if (nv) atomicAdd(&tally[0], nv);
        if (nm) atomicAdd(&tally[1], nm);
    }
}

// ZML as a LANE STATE MACHINE ("zml_kernel_flat", "zml_variant" 1; the default for tables up to 3 GB: see launch_zml
// for the numbers): the base-synchronous kernel above costs a wave, per base, max-over-lanes(shrink trips) + 1 +
// max-over-lanes(fast-forward trips) dependent round trips.  Here every lane runs the per-base micro-steps
//   M0 start base k: phrase open and base legal -> which ends must scan (update_interval)      else FAIL
//   M1 scans done:   interval still non-empty   -> the two LF jumps; both ends fast-forward      else FAIL
//   M2 jumps done:   interval still non-empty   -> ml += 1, emit, next base (M0 in the same iteration)   else FAIL
//   FAIL             ml = 0, the base opens the next phrase (rows of its interval's ends are fetched), emit 0, next base
// on its own clock.  Each iteration a lane consumes ONE 4-row window per interval end -- the window around the next row
// that end needs, whatever the reason (scan step, LF target, fast-forward neighbour, first rows of a new phrase) --
// walks as far as that window reaches, then runs the micro-steps as far as they go without new rows (typically
// M2 -> M0 -> M1: one iteration per matched base), and issues the two fetches of the next iteration before its
// bookkeeping (emission, base decode) -- the software pipelining of pml_kernel_flatp, and its read-chunk handling.
// The end-by-end order inside an iteration (start end first, whole window; then the end end) is the order of
// shrink_interval's trips, so answers AND scan / fast-forward counts equal the base-synchronous kernel's.
// SEG = 1: a lane parses one SEGMENT of a read (K1 of launch_zml_segmented), as zml_kernel<MODE, 1>.
// PSH = 1 (round 4; plain rows, whole reads): the two windows of an iteration by pairs of lanes, as pml_kernel_flatp<..., PSH = 1>.
// CNT = 1 (round 5): the COUNT query (query_backward_search, src/move_structure_search.cpp:340-352) on the same machine.  A
// backward search is the parse's first phrase: the search ends where the phrase would (the interval comes out empty, or the
// base is illegal) and reports the last non-empty interval and how many bases it matched, instead of opening the next phrase;
// nothing is emitted per base.  The first K bases come from the interval table (DevIndex::ftab) like count_kernel_v0's.  Why: on
// tables beyond the TLBs' reach the base-synchronous count_kernel_v0 pays max-over-the-wavefront dependent round trips per base
// (interval shrink, two LF gathers, a fast-forward loop that takes one row per trip: 0.91 fast-forwards per LF on the random
// 1 B-row table); here every lane fetches two row windows per iteration -- by pairs of lanes (PSH) -- and resolves the
// fast-forwards and scans inside them in closed form.  `matched` / `count`: per read, as count_kernel_v0 writes them.
template <int MODE, typename IdxT, int SEG = 0, int AH = 0, int PSH = 0, int CNT = 0>
__global__ __launch_bounds__(256) void zml_kernel_flat(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order, ZSegArgs seg,
                                                       uint64_t *__restrict__ matched, uint64_t *__restrict__ count) {
    static_assert(AH == 0 || (MODE == 6 && SEG == 0), "look-ahead rows: regular-thresholds rows, whole reads");
    static_assert(PSH == 0 || (AH == 0 && SEG == 0 && (MODE == 6 || MODE == 3)), "pair-shared gathers: plain 8-byte rows, whole reads");
    static_assert(CNT == 0 || SEG == 0, "the count query: whole reads");
    enum : uint32_t { phStart = 0, phScan = 1, phLF = 2, phInit = 3, phDone = 4 };
    enum : uint32_t { pNone = 0, pScan = 1, pFF = 2 };       // what an interval end is waiting for
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, failed = 0;
    const bool valid = SEG == 1 ? t < *seg.n_seg : t < n_reads;
    const uint64_t rid = (valid && order && SEG == 0) ? order[t] : t;
    const uint64_t beg = valid ? (SEG == 1 ? seg.seg_in[rid] : offs[rid]) : 0;
    const uint32_t len = valid ? (SEG == 1 ? seg.seg_len[rid] : (uint32_t)(offs[rid + 1] - beg)) : 0;
    const uint64_t obeg = (SEG == 1 && valid) ? seg.seg_out[rid] : beg;
    uint16_t *O = CNT ? nullptr : out + obeg;
    const uint32_t packed_end = len & ~7u;
    const IdxT r1 = (IdxT)(ix.r - 1), end_row = (IdxT)ix.end_bwt_idx, wb_last = (IdxT)(ix.r - 4);
    // K1: the state right after the base just emitted (the machine may already be moving on to the next base when the
    // emission is booked)
    IdxT cap_rs = 0, cap_re = 0;
    uint32_t cap_os = 0, cap_oe = 0, cap_ml = 0, cap_open = 0, cap_ff = 0, cap_scan = 0;

    auto load_pair_at = [&](uint64_t e, uint64_t &c0, uint64_t &c1) {      // see pml_kernel_flatp
        uint64_t two[2];
        __builtin_memcpy(two, bases + (e >= 16 ? e - 16 : 0), 16);
        c0 = two[1];
        c1 = two[0];
    };
    auto fix_pair = [&](uint64_t e, uint64_t &c0, uint64_t &c1) {
        if (e < 16) {
            const uint32_t sh = 8u * (uint32_t)(16 - e);
            if (sh >= 64) { c0 = c1 << (sh - 64); c1 = 0; }
            else { c0 = (c0 << sh) | (c1 >> (64 - sh)); c1 <<= sh; }
        }
    };
    auto win_base = [&](IdxT nd) -> IdxT {
        const IdxT wb = nd & ~(IdxT)3;
        return wb < wb_last ? wb : wb_last;
    };

    // interval [rs:os, re:oe] (MoveInterval), the rows of its ends, and what each end waits for
    IdxT rs = 0, re = 0, lo = 0, hi = 0;
    uint32_t os = 0, oe = 0, open = 0, dead = 0;
    uint2 rws = make_uint2(0, 0), rwe = make_uint2(0, 0);
    uint2 ens = make_uint2(0, 0), ene = make_uint2(0, 0);   // AH: the look-ahead entries of the rows the two ends stand on (0 = none)
    uint32_t ps = pNone, pe = pNone, ffs = 0, ffe = 0;
    uint32_t ph = len > 0 ? phStart : phDone;
    uint32_t k = 0, ml = 0;
    uint64_t rb = 0, rb2 = 0, nx0 = 0, nx1 = 0;
    if (len > 0) {
        load_pair_at(beg + len, rb, rb2);
        fix_pair(beg + len, rb, rb2);
    }
    if (len > 16 && (SEG != 0 || ix.stage_lds == 0u)) load_pair_at(beg + len - 16, nx0, nx1);   // (staged reads: the bases come from LDS)
    // b = code of the base of step k, bn = of step k + 1, bn2 = of step k + 2: M2 -> M0 chains two bases inside one
    // iteration (three on the look-ahead rows), so the next bases are decoded ahead (rb always holds the 8-group of step
    // k + 2, the furthest one decoded)
    uint32_t b = s_code[(uint32_t)(rb >> 56) & 0xFFu];
    uint32_t bn = len > 1 ? s_code[(uint32_t)(rb >> 48) & 0xFFu] : 0xFFu;
    uint32_t bn2 = len > 2 ? s_code[(uint32_t)(rb >> 40) & 0xFFu] : 0xFFu;
    // READS STAGED THROUGH LDS (round 6; ix.stage_lds = bases per lane in the block's dynamic LDS, set by the launchers for blocks of one
    // wavefront, 0 = none): every lane copies the next stretch of its read into LDS -- 16 bytes per load from the read's end backwards,
    // so a wavefront's contiguous reads arrive as whole cache lines, each fetched once -- and the bases of steps k, k + 1, k + 2 come
    // from there (pml_kernel_flatp's staging: same layout, slot s of lane l at byte (s / 4) * 256 + 4 l + s % 4).  Without it a lane
    // fetches its read 16 bases at a time and every fetch is a line from the fabric: 0.0625 lines per base.  Longer reads roll.
    extern __shared__ __align__(16) uint8_t z_stage[];
    const uint32_t zcap = (SEG == 0) ? ix.stage_lds : 0u;
    uint32_t zbase = 0;
    auto zstage_from = [&](uint32_t k0, bool on) {        // every lane of the wavefront makes the call; lanes with `on` stage
        uint32_t *S = reinterpret_cast<uint32_t *>(z_stage);
        const uint32_t sl = threadIdx.x & 63u;
        const uint32_t left = (on && len > k0) ? len - k0 : 0u;
        const uint32_t cnt = left < zcap ? left : zcap;
        for (uint32_t g = 0; wave_any(16u * g < cnt); g += 2u) {
            uint64_t c0[2], c1[2];
#pragma unroll
            for (uint32_t u = 0; u < 2u; ++u) {
                const uint64_t e = 16u * (g + u) < cnt ? beg + len - k0 - 16u * (g + u) : 16u;
                load_pair_at(e, c0[u], c1[u]);
            }
#pragma unroll
            for (uint32_t u = 0; u < 2u; ++u) {
                if (16u * (g + u) < cnt) {
                    const uint64_t e = beg + len - k0 - 16u * (g + u);
                    fix_pair(e, c0[u], c1[u]);
                    const uint64_t r0 = __builtin_bswap64(c0[u]), r1x = __builtin_bswap64(c1[u]);
                    S[(4u * (g + u) + 0u) * 64u + sl] = (uint32_t)r0;
                    S[(4u * (g + u) + 1u) * 64u + sl] = (uint32_t)(r0 >> 32);
                    S[(4u * (g + u) + 2u) * 64u + sl] = (uint32_t)r1x;
                    S[(4u * (g + u) + 3u) * 64u + sl] = (uint32_t)(r1x >> 32);
                }
            }
        }
        if (on) zbase = k0;
    };
    auto zcode = [&](uint32_t j) -> uint32_t {            // code of the base of step j (inside the staged stretch: the rolls see to it), 0xFF beyond the read
        const uint32_t q0 = j - zbase, q = q0 < zcap ? q0 : zcap - 1u;
        const uint32_t c = s_code[z_stage[(q >> 2) * 256u + (threadIdx.x & 63u) * 4u + (q & 3u)]];
        return j < len ? c : 0xFFu;
    };
    if (zcap) zstage_from(0u, len > 0);
    uint4 pk = make_uint4(0, 0, 0, 0), pk_old = pk;
    // CNT: the last non-empty interval (what the search reports: backward_search :176-199) and whether the search has begun
    IdxT prs = 0, pre = 0;
    uint32_t pos_ = 0, poe = 0, have = 0;
    if (CNT && ix.ftab_k != 0u) {                             // the first K bases by one lookup (count_kernel_v0 has the table's story)
        const uint32_t K = ix.ftab_k;
        uint32_t kidx = 0, bad = (uint32_t)(len < K);
        for (uint32_t i = 0; i < K; ++i) {
            const uint64_t src = i < 8u ? rb : rb2;
            const uint32_t cc = (bad ? 0xFFu : (uint32_t)s_code[(uint32_t)(src >> (8u * (7u - (i & 7u)))) & 0xFFu]) - ix.sep;
            bad |= (uint32_t)(cc > 3u);
            kidx |= (cc & 3u) << (2u * i);
        }
        uint4 e4 = make_uint4(0, 0, 0, 0);
        if (!bad) e4 = ix.ftab[kidx];
        if (e4.w >> 31) {
            rs = (IdxT)((uint64_t)e4.x | ((uint64_t)(e4.z & 15u) << 32));
            re = (IdxT)((uint64_t)e4.y | ((uint64_t)((e4.z >> 4) & 15u) << 32));
            os = (e4.z >> 8) & 0xFFFu;
            oe = e4.z >> 20;
            ff_total = e4.w & 0x7FFFu;
            scan_total = (e4.w >> 15) & 0xFFFFu;
            prs = rs; pre = re; pos_ = os; poe = oe; have = 1;
            k = K;
            open = 1;
            if (k == len) ph = phDone;
            else { ps = pFF; pe = pFF; ffs = 65535u; ffe = 65535u; ph = phInit; }   // the rows of the two ends: no fast-forward
            // the decoder's state for step K: b, bn, bn2 = steps K .. K + 2 (K <= 12: all inside the first 16 bases), rb = the
            // 8-group of step K + 2
            auto code_at = [&](uint32_t j) -> uint32_t {
                const uint64_t src = j < 8u ? rb : rb2;
                return j < len ? (uint32_t)s_code[(uint32_t)(src >> (8u * (7u - (j & 7u)))) & 0xFFu] : 0xFFu;
            };
            b = code_at(K); bn = code_at(K + 1u); bn2 = code_at(K + 2u);
            if (K + 2u >= 8u) rb = rb2;
        }
    }
    uint2 ws[4], we[4], es[4], ee[4];                         // AH: es / ee = the look-ahead entries of the windows' rows
    // a window of the table the parse walks on: the plain rows, or (AH) the look-ahead copy with its rows' entries
    uint4 raw_s[2], raw_e[2];                                 // PSH: what this lane loaded for its pair, assembled at the loop's top
    const uint32_t odd_lane = threadIdx.x & 1u;
    auto fetch_win = [&](IdxT wb, uint2 (&wr)[4], uint2 (&en)[4], uint4 (&raw)[2]) {
        if (PSH) {                                           // each lane one 16-byte half of the even lane's window and of the odd lane's
            const uint64_t at = (uint64_t)wb * 8u;
            const uint64_t pat = (uint64_t)pair_swap((uint32_t)at) | ((uint64_t)pair_swap((uint32_t)(at >> 32)) << 32);
            __builtin_memcpy(&raw[0], ix.rows + (odd_lane ? pat : at) + 16u * odd_lane, 16);
            __builtin_memcpy(&raw[1], ix.rows + (odd_lane ? at : pat) + 16u * odd_lane, 16);
        } else if (AH) {
            const uint64_t at = wb < wb_last ? (uint64_t)(wb >> 3) * 128u + (uint64_t)((uint32_t)wb & 4u) * 8u : ix.rows2_tail;
            load_window<MODE>(ix.rows2 + at, 0, wr);
            load_window<MODE>(ix.rows2 + at + 64u, 0, en);
        } else {
            load_window<MODE>(ix.rows, (uint64_t)wb, wr);
        }
    };
    // (AH) the row j as the walk needs it -- id, length, offset, character -- from the entry of a row whose LF target it is
    auto row_of_entry = [&](uint2 e) -> uint2 {
        return make_uint2(e.x, (e.y & 0x7FFu) | (((e.y >> 22) & 7u) << 13) | (((e.y >> 11) & 0x7FFu) << 16) | (((e.y >> 25) & 15u) << 28));
    };
    fetch_win(0, ws, es, raw_s);
    fetch_win(0, we, ee, raw_e);
    IdxT wbs = 0, wbe = 0;                                   // bases of the two windows in flight

    uint32_t lane_steps = 0, wave_steps = 0;
    while (wave_any(ph != phDone)) {
        const bool act = ph != phDone;
        lane_steps += (uint32_t)act;
        wave_steps += 1;
        if (PSH) {                                           // the halves the pair loaded for each other change hands
            pair_assemble(odd_lane, raw_s[0], raw_s[1], ws);
            pair_assemble(odd_lane, raw_e[0], raw_e[1], we);
        }
        if (AH) {                                            // the copy's reposition hints (DevIndex::hints) are the walk's business: masked off
            const uint32_t ym = ix.hints ? 0x0FFFFFFFu : 0xFFFFFFFFu, em = ix.hints ? ~(0x3Fu << 25) : 0xFFFFFFFFu;
#pragma unroll
            for (int t = 0; t < 4; ++t) { ws[t].y &= ym; we[t].y &= ym; es[t].y &= em; ee[t].y &= em; }
        }
        // ---- 1. walk each end as far as its window reaches (start end first: shrink_interval's order), in closed form as
        // pml_kernel_flatp's window_advance: a fast-forward passes row i iff the offset covers the running sum of the
        // lengths up to i; a scan passes the leading run of rows that are not its target (a 4-bit mask + count zeros).
        {
            const uint32_t cb0 = row_c<MODE>(ws[0]) == b, cb1 = row_c<MODE>(ws[1]) == b, cb2 = row_c<MODE>(ws[2]) == b,
                           cb3 = row_c<MODE>(ws[3]) == b;
            const uint32_t eq = (uint32_t)(end_row - wbs);                 // window index of the '$' row: never a target
            const uint32_t hitm = (cb0 | (cb1 << 1) | (cb2 << 2) | (cb3 << 3)) & ~(eq < 4u ? (1u << eq) : 0u);
            if (ps == pScan) {                               // update_interval: start moves down to the next row of b
                if (rs >= hi) { dead = 1; ps = pNone; pe = pNone; }
                else {
                    const uint32_t qf = (uint32_t)((IdxT)(rs + 1) - wbs);  // first candidate; candidates are qf .. min(3, hi - wbs)
                    if (qf < 4u) {
                        const uint32_t last = (uint32_t)(hi - wbs) < 3u ? (uint32_t)(hi - wbs) : 3u;
                        const uint32_t cand = (hitm >> qf) << qf & ((2u << last) - 1u);
                        const uint32_t h = cand ? (uint32_t)__builtin_ctz(cand) : last;
                        scan_total += h - qf + 1;
                        rs = wbs + (IdxT)h;
                        os = 0;
                        if (cand) { rws = win_sel(ws, h); if (AH) ens = win_sel(es, h); ps = pNone; }
                        else if (rs >= hi) { dead = 1; ps = pNone; pe = pNone; }
                    }
                }
            } else if (ps == pFF) {                          // fast_forward of the start walker (also: first row of a phrase)
                const uint32_t q0 = (uint32_t)(rs - wbs);
                if (q0 < 4u) {
                    const uint32_t n0 = row_n<MODE>(ws[0]), n1 = row_n<MODE>(ws[1]), n2 = row_n<MODE>(ws[2]), n3 = row_n<MODE>(ws[3]);
                    const uint32_t m0 = q0 == 0u, m1 = q0 <= 1u, m2 = q0 <= 2u;
                    const uint32_t t1 = m0 ? n0 : 0u, t2 = t1 + (m1 ? n1 : 0u), t3 = t2 + (m2 ? n2 : 0u), t4 = t3 + n3;
                    const uint32_t lastw = (uint32_t)(wbs + 3 == r1);
                    uint32_t cnt = (m0 & (uint32_t)(os >= t1)) + (m1 & (uint32_t)(os >= t2)) + (m2 & (uint32_t)(os >= t3)) +
                                   ((uint32_t)(os >= t4) & (lastw ^ 1u));
                    const uint32_t room = 65535u - (ffs < 65535u ? ffs : 65535u);
                    cnt = cnt < room ? cnt : room;
                    const uint32_t q = q0 + cnt;             // where the walker stands now
                    os -= (q == 0u ? 0u : (q == 1u ? t1 : (q == 2u ? t2 : (q == 3u ? t3 : t4))));
                    ffs += cnt;
                    rs += (IdxT)cnt;
                    if (q < 4u) { rws = win_sel(ws, q); if (AH) ens = win_sel(es, q); ps = pNone; }
                }
            }
        }
        {
            const uint32_t cb0 = row_c<MODE>(we[0]) == b, cb1 = row_c<MODE>(we[1]) == b, cb2 = row_c<MODE>(we[2]) == b,
                           cb3 = row_c<MODE>(we[3]) == b;
            const uint32_t eq = (uint32_t)(end_row - wbe);
            const uint32_t hitm = (cb0 | (cb1 << 1) | (cb2 << 2) | (cb3 << 3)) & ~(eq < 4u ? (1u << eq) : 0u);
            if (pe == pScan) {                               // the end moves up to the previous row of b
                if (re <= lo) { dead = 1; ps = pNone; pe = pNone; }
                else {
                    const uint32_t qf = (uint32_t)((IdxT)(re - 1) - wbe);  // first candidate; candidates are qf down to max(0, lo - wbe)
                    if (qf < 4u) {
                        const uint32_t first = lo > wbe ? (uint32_t)(lo - wbe) : 0u;
                        const uint32_t cand = (hitm & ((2u << qf) - 1u)) >> first << first;
                        const uint32_t h = cand ? 31u - (uint32_t)__builtin_clz(cand) : first;
                        scan_total += qf - h + 1;
                        re = wbe + (IdxT)h;
                        if (cand) { rwe = win_sel(we, h); if (AH) ene = win_sel(ee, h); oe = row_n<MODE>(rwe) - 1; pe = pNone; }
                        else if (re <= lo) { dead = 1; ps = pNone; pe = pNone; }
                    }
                }
            } else if (pe == pFF) {
                const uint32_t q0 = (uint32_t)(re - wbe);
                if (q0 < 4u) {
                    const uint32_t n0 = row_n<MODE>(we[0]), n1 = row_n<MODE>(we[1]), n2 = row_n<MODE>(we[2]), n3 = row_n<MODE>(we[3]);
                    const uint32_t m0 = q0 == 0u, m1 = q0 <= 1u, m2 = q0 <= 2u;
                    const uint32_t t1 = m0 ? n0 : 0u, t2 = t1 + (m1 ? n1 : 0u), t3 = t2 + (m2 ? n2 : 0u), t4 = t3 + n3;
                    const uint32_t lastw = (uint32_t)(wbe + 3 == r1);
                    uint32_t cnt = (m0 & (uint32_t)(oe >= t1)) + (m1 & (uint32_t)(oe >= t2)) + (m2 & (uint32_t)(oe >= t3)) +
                                   ((uint32_t)(oe >= t4) & (lastw ^ 1u));
                    const uint32_t room = 65535u - (ffe < 65535u ? ffe : 65535u);
                    cnt = cnt < room ? cnt : room;
                    const uint32_t q = q0 + cnt;
                    oe -= (q == 0u ? 0u : (q == 1u ? t1 : (q == 2u ? t2 : (q == 3u ? t3 : t4))));
                    ffe += cnt;
                    re += (IdxT)cnt;
                    if (q < 4u) { rwe = win_sel(we, q); if (AH) ene = win_sel(ee, q); pe = pNone; }
                }
            }
        }
        // ---- 2. micro-steps, as far as they go without new rows.  Up to two emissions per iteration: A (the base whose rows
        // have arrived, or a phrase's end) and, on the look-ahead rows, B (a base both of whose LF moves are known from the ends'
        // entries to land without a fast-forward: nothing to fetch for it).
        uint32_t fail = 0, n_emit = 0, ekA = 0, valA = 0, ekB = 0, valB = 0, errc = kErrNone, la_taken = 0;
        auto book = [&]() {                                  // MoveQuery::add_ml of step k (u16 clamp), then on to the next base
            const uint32_t val = ml > 65535u ? 65535u : ml;
            if (n_emit == 0u) { ekA = k; valA = val; } else { ekB = k; valB = val; }
            n_emit += 1;
            if (SEG == 1) { cap_rs = rs; cap_re = re; cap_os = os; cap_oe = oe; cap_ml = ml; cap_open = open; cap_ff = ff_total; cap_scan = scan_total; }
            k += 1;
            b = bn; bn = bn2;
        };
        auto micro = [&](bool first) {
            const bool ready = act && ph != phDone && ps == pNone && pe == pNone;
            fail = 0;
            if (first && ready && ph == phLF) {              // M2: both jumps and their fast-forwards are done
                ff_total += ffs + ffe;
                if (ffs >= 65535u || ffe >= 65535u) errc = kErrFastForward;   // move_structure.cpp:72-75
                else if ((rs < re) || (rs == re && os <= oe)) {  // query_zml :717-720
                    ml += 1;
                    book();
                    if (CNT) { prs = rs; pre = re; pos_ = os; poe = oe; }
                    ph = k == len ? phDone : phStart;
                } else fail = 1;
            } else if (first && ready && ph == phInit) {     // the rows of a new phrase's ends have arrived
                ph = phStart;
            }
            if (ph == phStart && n_emit == 0u && act && b == 0xFFu) fail = 1;       // illegal base: no phrase
            if (ph == phStart && n_emit == 0u && act && open == 0u) fail = 1;       // no phrase to extend
            if (ph == phStart && act && fail == 0u && open != 0u && b != 0xFFu) {   // M0: update_interval begins
                ps = (rs == end_row || row_c<MODE>(rws) != b) ? pScan : pNone;
                pe = (re == end_row || row_c<MODE>(rwe) != b) ? pScan : pNone;
                lo = rs; hi = re; dead = 0;
                ph = phScan;
            }
            if (ph == phScan && act && ps == pNone && pe == pNone && errc == kErrNone) {   // M1: scans done (or none needed)
                if (dead == 0u && ((rs < re) || (rs == re && os <= oe))) {
                    const uint64_t ja = row_id<MODE>(rws, (uint64_t)rs, ix), jb = row_id<MODE>(rwe, (uint64_t)re, ix);
                    if (ja >= ix.r || jb >= ix.r) errc = kErrIdRange;       // move_structure.cpp:63-65
                    else {
                        os += row_off<MODE>(rws);
                        oe += row_off<MODE>(rwe);
                        rs = (IdxT)ja; re = (IdxT)jb;
                        // (AH) both targets are reached below their lengths -- known from the entries -- and the interval they
                        // span is not empty: the step is complete without their rows (query_zml :717-720: ml + 1), and what
                        // the next base needs of them -- character, offset, their own targets -- is in the entries too
                        const uint32_t la = AH ? (uint32_t)first & (ens.y >> 31) & (ene.y >> 31) & (uint32_t)(os < (ens.y & 0x7FFu)) &
                                                 (uint32_t)(oe < (ene.y & 0x7FFu)) & (uint32_t)((rs < re) || (rs == re && os <= oe))
                                               : 0u;
                        if (la) {
                            ml += 1;
                            book();
                            if (CNT) { prs = rs; pre = re; pos_ = os; poe = oe; }
                            rws = row_of_entry(ens); rwe = row_of_entry(ene);
                            ens = make_uint2(0u, 0u); ene = make_uint2(0u, 0u);
                            ph = k == len ? phDone : phStart;
                            la_taken = 1;
                        } else {
                            ps = pFF; pe = pFF; ffs = 0; ffe = 0;
                            ph = phLF;
                        }
                    }
                } else if (n_emit == 0u) {
                    fail = 1;                                // (with an emission already made in this iteration: decided again
                }                                            // in the next one -- ph stays phScan)
            }
            if (fail && CNT && k != 0u) {                    // the search ends here (backward_search :176-199): the last non-empty
                ph = phDone;                                 // interval is reported, this base is not matched
                ps = pNone; pe = pNone;
            } else if (fail) {                               // :750-760 / :696-704: the base opens the next phrase
                ml = 0;
                open = 0;
                ph = phStart;
                if (b != 0xFFu) {                            // initialize_backward_search :284-291
                    rs = (IdxT)ix.first_runs[b + 1]; re = (IdxT)ix.last_runs[b + 1];
                    os = (uint32_t)ix.first_offsets[b + 1]; oe = (uint32_t)ix.last_offsets[b + 1];
                    open = ((rs < re) || (rs == re && os <= oe)) ? 1u : 0u;
                    if (open) { ps = pFF; pe = pFF; ffs = 65535u; ffe = 65535u; ph = phInit; }   // rows only: no fast-forward
                    if (CNT) { prs = rs; pre = re; pos_ = os; poe = oe; have = 1; }
                }
                if (CNT && (b == 0xFFu || open == 0u)) {     // query_backward_search :344-347: an illegal last base matches nothing ("0/len");
                    if (b != 0xFFu) k = 1;                   // a base that does not occur: itself, with its (empty) interval
                    ph = phDone;
                } else {
                    book();
                    if (k == len) ph = phDone;
                }
            }
        };
        micro(true);
        if (AH) {
            if (wave_any(la_taken != 0u)) {                  // the base after a look-ahead step: its shrink begins, or its LF moves leave
                if (la_taken && errc == kErrNone) micro(false);
            }
        }
        if (errc) { failed = errc; ph = phDone; ps = pNone; pe = pNone; }
        // ---- 3. the next two windows leave now; everything below runs under their latency
        {
            const IdxT ns = ps == pScan ? (IdxT)(rs + 1) : rs, ne = pe == pScan ? (IdxT)(re - (re > 0 ? 1 : 0)) : re;
            wbs = ps != pNone ? win_base(ns) : (IdxT)0;
            wbe = pe != pNone ? win_base(ne) : (IdxT)0;
            fetch_win(wbs, ws, es, raw_s);
            fetch_win(wbe, we, ee, raw_e);
        }
        // ---- 4. bookkeeping: the emissions and the next bases
        uint32_t want_nx = 0;
        uint64_t nx_e = 0;
        auto emit_at = [&](uint32_t ek, uint32_t val) {
            if (SEG == 1) {
                if ((ek & 31u) == 31u) {
                    ZSegCkpt ck;
                    ck.rs = (uint64_t)cap_rs; ck.re = (uint64_t)cap_re; ck.os = cap_os; ck.oe = cap_oe; ck.ml = cap_ml; ck.open = cap_open;
                    ck.ff = cap_ff; ck.scan = cap_scan;
                    seg.ckpt[(obeg + ek) >> 5] = ck;
                }
                if (ek + 1 == len) {
                    ZSegFin fn;
                    fn.rs = (uint64_t)cap_rs; fn.re = (uint64_t)cap_re; fn.os = cap_os; fn.oe = cap_oe; fn.ml = cap_ml; fn.open = cap_open;
                    seg.fin[rid] = fn;
                }
            }
            if (ek >= packed_end) {
                O[ek] = (uint16_t)val;
            } else {
                pk.x = (pk.x >> 16) | (pk.y << 16);
                pk.y = (pk.y >> 16) | (pk.z << 16);
                pk.z = (pk.z >> 16) | (pk.w << 16);
                pk.w = (pk.w >> 16) | (val << 16);
                if ((ek & 15) == 7) {
                    if (ek + 8 < packed_end) pk_old = pk;
                    else __builtin_memcpy(O + (ek - 7), &pk, 16);
                } else if ((ek & 15) == 15) {
                    __builtin_memcpy(O + (ek - 15), &pk_old, 16);
                    __builtin_memcpy(O + (ek - 7), &pk, 16);
                }
            }
        };
        // one more base decoded ahead per emission: step j = (step of the emission) + 3 moves into the last place of (b, bn, bn2)
        auto decode_ahead = [&](uint32_t j, uint32_t &dst) {
            dst = 0xFFu;
            if (j < len) {
                if ((j & 15) == 8) {
                    rb = rb2;
                } else if ((j & 15) == 0) {
                    rb = nx0;
                    rb2 = nx1;
                    fix_pair(beg + len - j, rb, rb2);
                    if (j + 16 < len) { want_nx = 1; nx_e = beg + len - j - 16; }
                }
                dst = s_code[(uint32_t)(rb >> (8 * (7 - (j & 7)))) & 0xFFu];
            }
        };
        if (n_emit) {
            if (!CNT) emit_at(ekA, valA);
            if (n_emit == 2u && !CNT) emit_at(ekB, valB);
        }
        if (zcap) {
            // the bases of steps k .. k + 2 from the staged stretch; a lane about to leave it makes the whole wavefront stage again
            const uint32_t far = k + 2u < len ? k + 2u : len - 1u;                     // the furthest step decoded ahead that is still a base of the read
            const uint32_t out_of = (uint32_t)(ph != phDone) & (uint32_t)(k < len) & (uint32_t)(far - zbase >= zcap);
            if (wave_any(out_of != 0u)) zstage_from(k, ph != phDone);
            if (n_emit) { b = zcode(k); bn = zcode(k + 1u); bn2 = zcode(k + 2u); }
        } else if (n_emit) {
            if (n_emit == 2u) {
                decode_ahead(ekA + 3u, bn);                   // (b, bn, bn2) were shifted twice: two places to fill
                decode_ahead(ekB + 3u, bn2);
            } else {
                decode_ahead(ekA + 3u, bn2);
            }
        }
        if (want_nx) load_pair_at(nx_e, nx0, nx1);
    }
    if (SEG == 1) {
        if (valid) {
            SegTot tt;
            tt.ff = ff_total; tt.scan = scan_total; tt.repo = 0; tt.flag = failed;
            seg.tot[rid] = tt;
        }
    } else {
        if (CNT) {
            if (valid) {
                uint64_t m_out = 0, c_out = 0;
                if (have && !failed) {
                    m_out = (uint64_t)k;
                    // MoveInterval::count, include/move_intervals.hpp:47-58, via the row-start checkpoints
                    if (prs == pre) c_out = (uint64_t)poe - pos_ + 1;
                    else c_out = (row_start<MODE>(ix, (uint64_t)pre) + poe) - (row_start<MODE>(ix, (uint64_t)prs) + pos_) + 1;
                }
                matched[rid] = m_out;
                count[rid] = c_out;
            }
        } else if (failed) {
            for (uint32_t i = 0; i < len; ++i) O[i] = 0;
        }
        if (valid && err) err[rid] = (uint8_t)failed;
        const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), erw = wave_sum(failed ? 1u : 0u);
        if ((threadIdx.x & 63) == 0 && stats) {
            if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
            if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
            if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
        }
    }
    const uint32_t lsw = wave_sum(lane_steps);
    if ((threadIdx.x & 63) == 0 && stats) {
        atomicAdd(&stats->lane_steps, (unsigned long long)lsw);
        atomicAdd(&stats->wave_steps, (unsigned long long)wave_steps);
    }
}

// The segmented ZML parse (the PML one, launch_pml_segmented, has the reasoning): plan, probe (verdict read back), K1 =
// zml_kernel<MODE, 1> over the segments, K2 = zml_stitch_kernel find + write, K3 = seg_finalize_kernel + zml_kernel<MODE, 2>.
// Unlike the PML walk the ZML parse is latency-bound even on a full batch of long reads (100 k x 10 kbp: 19 Gbases/s), so
// there is something to gain up to 8 wavefronts of reads per CU.
template <int MODE>
static hipError_t launch_zml_segmented(const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                       uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats, const LaunchCfg &cfg,
                                       hipStream_t stream, SegWorkspace *ws, int ragged_hint, bool *declined, int *verdict) {
    *declined = false;
    int forced = -1;                                       // as launch_pml_segmented: seg_probe == 2 = the caller's verdict
    if (verdict && *verdict >= 0) forced = *verdict;
    if (cfg.seg_probe == 2) forced = cfg.seg_verdict ? 1 : 0;
    if (forced == 0) { *declined = true; return hipSuccess; }
    const bool probe = cfg.seg_probe == 1 && forced != 1;
    const uint32_t S = call_seg_len(cfg, n_bases, 24);
    hipError_t e = hipSuccess;
    if (cfg.seg_probe == 1 && n_reads >= (uint64_t)cfg.num_cus * 64ull * 8ull) {
        if (ragged_hint == 0) { *declined = true; return hipSuccess; }
        if (ragged_hint < 0) {
            if (ws->cap < 64) {
                if (ws->buf) (void)hipFree(ws->buf);
                ws->buf = nullptr;
                ws->cap = 0;
                e = hipMalloc(&ws->buf, 4096);
                if (e != hipSuccess) return e;
                ws->cap = 4096;
            }
            uint32_t *d_max = static_cast<uint32_t *>(ws->buf), h_max = 0;
            e = hipMemsetAsync(d_max, 0, 4, stream);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(seg_maxlen_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, stream, d_offsets, n_reads, d_max);
            e = hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            if ((uint64_t)h_max * 2ull <= (n_bases / n_reads) * 3ull) { *declined = true; return hipSuccess; }
        }
    }
    const uint64_t max_seg = n_reads + n_bases / S + 1;
    if (max_seg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const uint64_t n_ck = (n_bases >> 5) + 2;
    size_t temp_bytes = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int)(n_reads + 1), stream);
    if (e != hipSuccess) return e;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes ? bytes : 8); return o; };
    const size_t o_nof = take((n_reads + 1) * 8), o_first = take((n_reads + 1) * 8), o_temp = take(temp_bytes),
                 o_in = take(max_seg * 8), o_out = take(max_seg * 8), o_len = take(max_seg * 4), o_j = take(max_seg * 4),
                 o_rem = take(max_seg * 4), o_fin = take(max_seg * sizeof(ZSegFin)), o_tot = take(max_seg * sizeof(SegTot)),
                 o_join = take(max_seg * sizeof(SegJoin)), o_chain = take(max_seg), o_fail = take(n_reads),
                 o_ck = take(n_ck * sizeof(ZSegCkpt)), o_go = take(32);
    if (ws->cap < off) {
        if (ws->buf) (void)hipFree(ws->buf);
        ws->buf = nullptr;
        ws->cap = 0;
        const size_t want = off + (off >> 3);
        e = hipMalloc(&ws->buf, want);
        if (e != hipSuccess) return e;
        ws->cap = want;
    }
    uint8_t *B = static_cast<uint8_t *>(ws->buf);
    uint64_t *n_of = reinterpret_cast<uint64_t *>(B + o_nof), *first = reinterpret_cast<uint64_t *>(B + o_first);
    uint64_t *seg_in = reinterpret_cast<uint64_t *>(B + o_in), *seg_out = reinterpret_cast<uint64_t *>(B + o_out);
    uint32_t *seg_l = reinterpret_cast<uint32_t *>(B + o_len), *seg_j = reinterpret_cast<uint32_t *>(B + o_j),
             *seg_rem = reinterpret_cast<uint32_t *>(B + o_rem);
    SegJoin *join = reinterpret_cast<SegJoin *>(B + o_join);
    uint8_t *on_chain = B + o_chain, *read_fail = B + o_fail;
    ZSegArgs seg;
    seg.seg_in = seg_in; seg.seg_out = seg_out; seg.seg_len = seg_l; seg.n_seg = first + n_reads;
    seg.ckpt = reinterpret_cast<ZSegCkpt *>(B + o_ck);
    seg.fin = reinterpret_cast<ZSegFin *>(B + o_fin);
    seg.tot = reinterpret_cast<SegTot *>(B + o_tot);
    seg.read_fail = read_fail;
    uint32_t *go = reinterpret_cast<uint32_t *>(B + o_go);               // go | probes | probes in step
    const unsigned bt256 = 256;
    e = hipMemsetAsync(go, 0, 32, stream);
    if (e != hipSuccess) return e;
    if (probe)
        hipLaunchKernelGGL(zml_probe_kernel<MODE>, dim3(16), dim3(64), 0, stream, ix, d_bases, d_offsets, n_reads, 32u, 384u, go + 1);
    hipLaunchKernelGGL(seg_decide_kernel, dim3(1), dim3(1), 0, stream, go + 1, (uint32_t)probe, go, d_offsets, n_reads, n_bases);
    if (probe) {
        uint32_t h_go = 0;
        e = hipMemcpyAsync(&h_go, go, 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (verdict) *verdict = h_go ? 1 : 0;
        if (!h_go) { *declined = true; return hipSuccess; }
    }
    // (go == 0 -- a probe nobody read back cannot say so here, but offsets that break the contract can: seg_decide_kernel --
    // leaves the plan without segments: K1 / K2 idle, seg_finalize_kernel hands every read to K3)
    hipLaunchKernelGGL(seg_count_kernel, dim3((unsigned)((n_reads + 1 + bt256 - 1) / bt256)), dim3(bt256), 0, stream, d_offsets,
                       n_reads, S, n_of, go);
    e = hipcub::DeviceScan::ExclusiveSum(B + o_temp, temp_bytes, n_of, first, (int)(n_reads + 1), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, d_offsets,
                       n_reads, S, first, seg_in, seg_out, seg_l, seg_j, seg_rem);
    const uint32_t *d_order = nullptr;
    // K1: the lane state machine where the plain query would use it (tables up to 3 GB), else the base-synchronous kernel
    {   // (movi_launch_log: the plan's K1 and K3 kernels by name, like the walk kernel's)
        char nm[96];
        if (ix.r <= (3ull << 30) / 8 && ix.r >= 8 && n_bases >= 16) snprintf(nm, sizeof(nm), "zml_kernel_flat<%d, %s, 1, 0, 0, 0>", MODE, ix.idx32 ? "unsigned int" : "unsigned long");
        else snprintf(nm, sizeof(nm), "zml_kernel<%d, 1>", MODE);
        note_walk_launch(nm);
        snprintf(nm, sizeof(nm), "zml_kernel<%d, 2>", MODE);
        note_walk_launch(nm);
    }
    if (ix.r <= (3ull << 30) / 8 && ix.r >= 8 && n_bases >= 16) {
        if (ix.idx32)
            hipLaunchKernelGGL((zml_kernel_flat<MODE, uint32_t, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix,
                               d_bases, d_offsets, max_seg, d_out, d_err, d_stats, d_order, seg, nullptr, nullptr);
        else
            hipLaunchKernelGGL((zml_kernel_flat<MODE, uint64_t, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix,
                               d_bases, d_offsets, max_seg, d_out, d_err, d_stats, d_order, seg, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL((zml_kernel<MODE, 1>), dim3((unsigned)((max_seg + bt256 - 1) / bt256)), dim3(bt256), 0, stream, ix, d_bases,
                           d_offsets, max_seg, d_out, d_err, d_stats, d_order, seg);
    }
    const uint32_t max_over = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)cfg.seg_len * (uint64_t)kSegOverrun);
    hipLaunchKernelGGL((zml_stitch_kernel<MODE, 0>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    hipLaunchKernelGGL(seg_finalize_kernel, dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, go, first, n_reads,
                       seg.tot, join, seg_l, seg_rem, on_chain, read_fail, d_err, d_stats);
    hipLaunchKernelGGL((zml_stitch_kernel<MODE, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    hipLaunchKernelGGL((zml_kernel<MODE, 2>), dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, ix, d_bases,
                       d_offsets, n_reads, d_out, d_err, d_stats, d_order, seg);
    return hipGetLastError();
}

static hipError_t launch_count_flat(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                                    uint64_t *d_matched, uint64_t *d_count, uint8_t *d_err, DevStats *d_stats, const uint32_t *d_order,
                                    const LaunchCfg &cfg, hipStream_t stream, LaunchInfo *info, bool pair) {
    const uint64_t blocks = (n_reads + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(64);
    const int wpc = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : 0;     // "waves_per_cu": occupancy cap by LDS padding (<= 64 KiB here), as launch_zml
    size_t dyn_lds = 0;
    if (wpc > 0) {
        const int bpc = wpc < 3 ? 3 : wpc;
        if (bpc < 32) dyn_lds = ((163840u / (unsigned)bpc) & ~1023u) - 1024u;
    }
    // reads staged through LDS (round 6; blocks of one wavefront): the cap's padding, or kZmlStageBytes of their own (16 wavefronts per CU)
    DevIndex ixl = ix;
    if (cfg.stage_reads != 0) {
        if (dyn_lds < kZmlStageBytes) dyn_lds = kZmlStageBytes;
        ixl.stage_lds = (uint32_t)std::min<size_t>(1024, (dyn_lds / 64) & ~(size_t)15);
    }
    // ("zml_ahead" 1, round 6: the search on the look-ahead rows where the handle holds them -- a base both of whose LF moves land without a
    // fast-forward is complete without the target rows; measured: profiles/r06_zml_count.txt)
    const bool ahead = cfg.zml_ahead != 0 && mode == 6 && ix.rows2 != nullptr && !pair;
    {
        char nm[96];
        snprintf(nm, sizeof(nm), "zml_kernel_flat<%d, %s, 0, %d, %d, 1>", mode, ix.idx32 ? "unsigned int" : "unsigned long", ahead ? 1 : 0, pair ? 1 : 0);
        note_walk_launch(nm);
    }
    if (info) {
        snprintf(info->kernel, sizeof(info->kernel), "zml_kernel_flat<%d, %s, 0, %d, %d, 1>", mode, ix.idx32 ? "unsigned int" : "unsigned long", ahead ? 1 : 0, pair ? 1 : 0);
        info->variant = 1; info->block_threads = 64; info->waves_per_cu = wpc; info->segmented = 0; info->idx64 = ix.idx32 ? 0 : 1; info->staged = (int)ixl.stage_lds;
        info->ahead = ahead ? 1 : 0;
    }
    if (ahead) {
        if (ix.idx32)
            hipLaunchKernelGGL((zml_kernel_flat<6, uint32_t, 0, 1, 0, 1>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads,
                               (uint16_t *)nullptr, d_err, d_stats, d_order, ZSegArgs(), d_matched, d_count);
        else
            hipLaunchKernelGGL((zml_kernel_flat<6, uint64_t, 0, 1, 0, 1>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads,
                               (uint16_t *)nullptr, d_err, d_stats, d_order, ZSegArgs(), d_matched, d_count);
        return hipGetLastError();
    }
#define MOVI_LAUNCH_CNT(M, T, P)                                                                                          \
    hipLaunchKernelGGL((zml_kernel_flat<M, T, 0, 0, P, 1>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads, \
                       (uint16_t *)nullptr, d_err, d_stats, d_order, ZSegArgs(), d_matched, d_count)
#define MOVI_LAUNCH_CNT_M(M)                                                                                              \
    do {                                                                                                                  \
        if (ix.idx32) { if (pair) MOVI_LAUNCH_CNT(M, uint32_t, 1); else MOVI_LAUNCH_CNT(M, uint32_t, 0); }                \
        else { if (pair) MOVI_LAUNCH_CNT(M, uint64_t, 1); else MOVI_LAUNCH_CNT(M, uint64_t, 0); }                         \
    } while (0)
    if (mode == 6) MOVI_LAUNCH_CNT_M(6);
    else if (mode == 3) MOVI_LAUNCH_CNT_M(3);
    else return hipErrorInvalidValue;
#undef MOVI_LAUNCH_CNT_M
#undef MOVI_LAUNCH_CNT
    return hipGetLastError();
}

hipError_t launch_zml(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                      const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream, SegWorkspace *seg_ws,
                      int ragged_hint, int *seg_verdict, LaunchInfo *info) {
    if (n_reads == 0) return hipSuccess;
    if (seg_ws && cfg.seg_len >= 32 && !d_order && cfg.zml_variant < 0 && cfg.block_threads == 0 && cfg.waves_per_cu <= 0 &&
        n_bases / n_reads >= 2ull * (uint64_t)cfg.seg_len && n_reads + n_bases / (uint64_t)cfg.seg_len < 0x7FFFFFF0ull &&
        (mode == 6 || mode == 3)) {
        bool declined = false;
        const hipError_t es = mode == 6 ? launch_zml_segmented<6>(ix, d_bases, d_offsets, n_reads, n_bases, d_out, d_err, d_stats, cfg,
                                                                  stream, seg_ws, ragged_hint, &declined, seg_verdict)
                                        : launch_zml_segmented<3>(ix, d_bases, d_offsets, n_reads, n_bases, d_out, d_err, d_stats, cfg,
                                                                  stream, seg_ws, ragged_hint, &declined, seg_verdict);
        if (es == hipSuccess && !declined && info) {
            const bool sm = ix.r <= (3ull << 30) / 8 && ix.r >= 8 && n_bases >= 16;
            if (sm) snprintf(info->kernel, sizeof(info->kernel), "zml_kernel_flat<%d, %s, 1, 0, 0, 0>", mode, ix.idx32 ? "unsigned int" : "unsigned long");
            else snprintf(info->kernel, sizeof(info->kernel), "zml_kernel<%d, 1>", mode);
            info->variant = sm ? 1 : 0; info->block_threads = sm ? 64 : 256; info->waves_per_cu = 0; info->segmented = 1; info->staged = 0; info->ahead = 0;
            info->idx64 = ix.idx32 ? 0 : 1;
        }
        if (es != hipSuccess || !declined) return es;
    }
    // 0 = base-synchronous kernel, 1 = lane state machine.  Measured (profiles/r02_zml_state_machine.txt), Gbases/s,
    // kernel 0 / 1: 100 k x 10 kbp 12.1 / 19.1 (pangenome), 12.2 / 18.2 (random 10 M rows); 1 M x 150 bp 36.8 / 39.2 and
    // 34.1 / 36.8; random tables of 120 M rows 24.6 / 28.3, 250 M (2 GB) 23.4 / 26.6, 500 M (4 GB) 21.3 / 16.6, 1 B (8 GB)
    // 17.0 / 13.8.  The state machine fetches two 4-row windows (four 16-byte loads, i.e. ~8 TLB lookups) per iteration:
    // beyond the ~1.7 GB reach of a CU's TLB that costs more than the base-synchronous kernel's dependent trips, so auto
    // picks it for tables up to 3 GB.  (Its first form walked the windows with eight sequential hops and was no faster
    // than kernel 0 anywhere: 12.4 on the long reads, 32.6 on the short ones; the closed-form window walk made it.)
    // Round 4: with the two windows fetched by PAIRS of lanes (zml_kernel_flat<..., PSH = 1>: half the translation requests) the
    // state machine serves the tables beyond 3 GB too: 1 B rows 16.8 (kernel 0) / 13.9 (kernel 1) -> 25.0 Gbases/s; below 2 GB the
    // exchange costs more than it gives (c2: 38.2 -> 36.6), so there the lanes keep their own loads (profiles/r04_zml_ahead.txt;
    // "pair_loads" 0: the old policy, 1: pairs everywhere).
    // (`ahead` and `pair` are settled BEFORE the kernel is picked: a caller who asks for the look-ahead rows forgoes the pairs, and
    // beyond 3 GB the unpaired state machine is the slowest of the three -- 13.9 against kernel 0's 16.8 Gbases/s at 1 B rows)
    int v = cfg.zml_variant;
    const bool big = ix.r * 8ull >= kPairLoadBytes;
    const bool want_ahead = cfg.zml_ahead != 0 && mode == 6 && ix.rows2 != nullptr;
    const bool can_pair = !want_ahead && (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && big));
    if (v < 0) v = (ix.r <= (3ull << 30) / 8 || can_pair) ? 1 : 0;
    if (v == 1 && (ix.r < 8 || n_bases < 16)) v = 0;     // the clamped windows need >= 4 rows, the 16-base fetches 16 bytes
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : (v == 1 ? 64 : 256);
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block((unsigned)bt);
    // The state machine on the look-ahead rows (round 4; "zml_ahead" 1, where the copy exists): a base both of whose LF moves
    // land without a fast-forward is complete without the target rows.  Lane iterations per base on c2 1.45 -> 0.98 -- and 37.3
    // instead of 38.2 Gbases/s (eight 16-byte loads per iteration instead of four, SIMT 0.72 -> 0.64; random 10 M-row table 35.9 ->
    // 35.0: profiles/r04_zml_ahead.txt), so it is an option, not the default.
    const bool ahead = want_ahead && v == 1;
    const bool pair = v == 1 && can_pair;
    {
        char nm[96];
        if (v == 1) snprintf(nm, sizeof(nm), "zml_kernel_flat<%d, %s, 0, %d, %d, 0>", mode, ix.idx32 ? "unsigned int" : "unsigned long", ahead ? 1 : 0, pair ? 1 : 0);
        else snprintf(nm, sizeof(nm), "zml_kernel<%d, 0>", mode);
        note_walk_launch(nm);
    }
    if (info) {
        if (v == 1) snprintf(info->kernel, sizeof(info->kernel), "zml_kernel_flat<%d, %s, 0, %d, %d, 0>", mode, ix.idx32 ? "unsigned int" : "unsigned long",
                             ahead ? 1 : 0, pair ? 1 : 0);
        else snprintf(info->kernel, sizeof(info->kernel), "zml_kernel<%d, 0>", mode);
        info->variant = v; info->block_threads = bt; info->waves_per_cu = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : 0; info->staged = 0; info->ahead = ahead ? 1 : 0;
        info->segmented = 0; info->idx64 = ix.idx32 ? 0 : 1;
    }
    size_t dyn_lds = 0;                                  // occupancy cap by LDS padding, as in launch_pml (<= 64 KiB here)
    if (cfg.waves_per_cu > 0) {
        int bpc = cfg.waves_per_cu / (bt / 64);
        if (bpc < 3) bpc = 3;
        if (bpc < 32) dyn_lds = ((163840u / (unsigned)bpc) & ~1023u) - 1024u;
    }
    DevIndex ixl = ix;                                   // reads staged through LDS (round 6): the state machine in blocks of one wavefront
    if (v == 1 && bt == 64 && cfg.stage_reads != 0) {
        if (dyn_lds < kZmlStageBytes) dyn_lds = kZmlStageBytes;
        ixl.stage_lds = (uint32_t)std::min<size_t>(1024, (dyn_lds / 64) & ~(size_t)15);
    }
#define MOVI_LAUNCH_ZML(M)                                                                                     \
    do {                                                                                                       \
        if (v == 0)                                                                                            \
            hipLaunchKernelGGL(zml_kernel<M>, grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads,   \
                               d_out, d_err, d_stats, d_order, ZSegArgs());                                    \
        else if (pair && ix.idx32)                                                                             \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint32_t, 0, 0, 1>), grid, block, dyn_lds, stream, ixl,     \
                               d_bases, d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr); \
        else if (pair)                                                                                         \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint64_t, 0, 0, 1>), grid, block, dyn_lds, stream, ixl,     \
                               d_bases, d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr); \
        else if (ix.idx32)                                                                                     \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint32_t>), grid, block, dyn_lds, stream, ixl, d_bases,      \
                               d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr); \
        else                                                                                                   \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint64_t>), grid, block, dyn_lds, stream, ixl, d_bases,      \
                               d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr); \
    } while (0)
    // resident layouts: 6 = regular-thresholds rows, 3 = regular rows (threshold-less types: 12-bit lengths)
    if (mode == 6 && ahead) {
        if (ix.idx32)
            hipLaunchKernelGGL((zml_kernel_flat<6, uint32_t, 0, 1>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads, d_out,
                               d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr);
        else
            hipLaunchKernelGGL((zml_kernel_flat<6, uint64_t, 0, 1>), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads, d_out,
                               d_err, d_stats, d_order, ZSegArgs(), nullptr, nullptr);
    } else if (mode == 6) MOVI_LAUNCH_ZML(6);
    else if (mode == 3) MOVI_LAUNCH_ZML(3);
    else return hipErrorInvalidValue;
#undef MOVI_LAUNCH_ZML
    return hipGetLastError();
}

// ------------------------------------------------------------ classification bins
// Classifier::classify, src/classifier.cpp:99-143: the PML vector (emission order) is cut into
// bins of bin_width, the LAST bin absorbing a remainder shorter than bin_width; per read the
// number of bins whose maximum is >= thr / < thr and the sum of the bin maxima.  One lane per
// read, PMLs fetched 8 at a time (16-byte loads; the vector was just written, L2/MALL-warm).
__global__ __launch_bounds__(256) void classify_kernel(const uint16_t *__restrict__ pml,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint32_t bin_width, uint32_t thr,
                                                       uint32_t *__restrict__ above, uint32_t *__restrict__ below,
                                                       uint64_t *__restrict__ sum_max, const uint8_t *__restrict__ err) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads) return;
    if (err && err[t]) {                                  // a read that broke an invariant reports no bins (ClsState::store)
        above[t] = 0; below[t] = 0; sum_max[t] = 0;
        return;
    }
    const uint64_t beg = offs[t], n = offs[t + 1] - beg;
    const uint16_t *P = pml + beg;
    uint64_t nb = n / bin_width;                          // bins: [iW, (i+1)W) for i < nb-1, last = [(nb-1)W, n)
    if (nb == 0) nb = 1;
    uint32_t a = 0, b = 0, cur = 0;
    uint64_t sum = 0, bin = 0, next_cut = (nb > 1) ? bin_width : n;
    uint64_t k = 0;
    while (k < n) {
        uint16_t v[8];
        uint32_t m = 8;
        if (k + 8 <= n) {
            __builtin_memcpy(v, P + k, 16);
        } else {
            m = (uint32_t)(n - k);
            for (uint32_t i = 0; i < m; ++i) v[i] = P[k + i];
        }
        for (uint32_t i = 0; i < m; ++i) {
            cur = v[i] > cur ? v[i] : cur;
            if (k + i + 1 == next_cut) {
                if (cur >= thr) a += 1; else b += 1;
                sum += cur;
                cur = 0;
                bin += 1;
                next_cut = (bin + 1 < nb) ? next_cut + bin_width : n;
            }
        }
        k += m;
    }
    above[t] = a;
    below[t] = b;
    sum_max[t] = sum;
}

// The same reduction with a WAVEFRONT per read (long reads): lane l takes bins l, l + 64, ... -- a bin is bin_width contiguous
// values, so a wavefront streams 64 x bin_width x 2 contiguous bytes per round, every byte of every line used -- and the
// per-lane tallies are summed across the wavefront.  (One lane per read walks 16 bytes at a time through its own 20 KB:
// 100 k x 10 kbp in 1.35 ms = 1.5 TB/s; this form: see profiles/r04_classify.txt.)
__global__ __launch_bounds__(64) void classify_wave_kernel(const uint16_t *__restrict__ pml, const uint64_t *__restrict__ offs,
                                                           uint64_t n_reads, uint32_t bin_width, uint32_t thr,
                                                           uint32_t *__restrict__ above, uint32_t *__restrict__ below,
                                                           uint64_t *__restrict__ sum_max, const uint8_t *__restrict__ err) {
    const uint64_t t = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (t >= n_reads) return;
    if (err && err[t]) {
        if (lane == 0) { above[t] = 0; below[t] = 0; sum_max[t] = 0; }
        return;
    }
    const uint64_t beg = offs[t], n = offs[t + 1] - beg;
    const uint16_t *P = pml + beg;
    uint64_t nb = n / bin_width;                          // bins: [iW, (i+1)W) for i < nb-1, last = [(nb-1)W, n)
    if (nb == 0) nb = 1;
    uint32_t a = 0, b = 0, s_lo = 0, s_hi = 0;            // (the sum of a lane's bin maxima: < 2^16 x bins, carried in two halves)
    if (n > 0) {
        for (uint64_t bin = lane; bin < nb; bin += 64) {
            const uint64_t s = bin * bin_width, e = (bin + 1 < nb) ? s + bin_width : n;
            uint32_t cur = 0;
            uint64_t k = s;
            auto max8 = [](uint4 v) -> uint32_t {
                const uint32_t m0 = max(v.x & 0xFFFFu, v.x >> 16), m1 = max(v.y & 0xFFFFu, v.y >> 16),
                               m2 = max(v.z & 0xFFFFu, v.z >> 16), m3 = max(v.w & 0xFFFFu, v.w >> 16);
                return max(max(m0, m1), max(m2, m3));
            };
            for (; k + 64 <= e; k += 64) {                    // eight 16-byte loads in flight per lane
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) __builtin_memcpy(&v[u], P + k + 8 * u, 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) cur = max(cur, max8(v[u]));
            }
            for (; k + 8 <= e; k += 8) {
                uint4 v;
                __builtin_memcpy(&v, P + k, 16);
                cur = max(cur, max8(v));
            }
            for (; k < e; ++k) cur = max(cur, (uint32_t)P[k]);
            a += cur >= thr ? 1u : 0u;
            b += cur >= thr ? 0u : 1u;
            const uint32_t lo = s_lo + cur;
            s_hi += lo < s_lo ? 1u : 0u;
            s_lo = lo;
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    // 64-bit sum across the wavefront from two 32-bit halves (each lane's low half < 2^32; sum the halves as 64-bit values)
    uint64_t tot = (uint64_t)s_lo | ((uint64_t)s_hi << 32);
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        const uint32_t olo = __shfl_xor((uint32_t)tot, sh, 64), ohi = __shfl_xor((uint32_t)(tot >> 32), sh, 64);
        tot += (uint64_t)olo | ((uint64_t)ohi << 32);
    }
    if (lane == 0) { above[t] = a; below[t] = b; sum_max[t] = tot; }
}

hipError_t launch_classify(const uint16_t *d_pml, const uint64_t *d_offsets, uint64_t n_reads, uint32_t bin_width,
                           uint32_t thr, uint32_t *d_above, uint32_t *d_below, uint64_t *d_sum, hipStream_t stream,
                           const uint8_t *d_err, uint64_t n_bases) {
    if (n_reads == 0) return hipSuccess;
    // long reads (a caller that knows the batch's size says so): a wavefront per read
    if (n_bases / n_reads >= 1024 && n_reads <= 0x7FFFFFFFull && bin_width > 0) {
        hipLaunchKernelGGL(classify_wave_kernel, dim3((unsigned)n_reads), dim3(64), 0, stream, d_pml, d_offsets, n_reads, bin_width,
                           thr, d_above, d_below, d_sum, d_err);
        return hipGetLastError();
    }
    const unsigned bt = 256;
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(classify_kernel, dim3((unsigned)blocks), dim3(bt), 0, stream, d_pml, d_offsets, n_reads, bin_width,
                       thr, d_above, d_below, d_sum, d_err);
    return hipGetLastError();
}

// ------------------------------------------------------------ reset masks <-> PML vectors (round 6)
// PML[k] = reset(k) ? 0 : PML[k - 1] + 1 (process_char either increments match_len or zeroes it, src/read_processor.cpp:193-215;
// MoveQuery::add_ml clamps to u16, include/move_query.hpp:26-38): the vector is the run length since the last reset.  The walk can
// write the reset bits alone (pml_kernel_flatp<..., RING = 2>; layout: MaskArgs, movi_kernels.hpp); these two streaming kernels
// turn masks into vectors and vectors (of the paths that have no mask output of their own) into masks.

// One 32-bit mask word -> 32 PMLs (16 packed pairs), `run` = match_len before the word's first base.
__device__ __forceinline__ void expand_word(uint32_t m, uint32_t &run, uint4 (&g)[4]) {
    uint32_t v[16];
#pragma unroll
    for (int b = 0; b < 32; b += 2) {
        run = ((m >> b) & 1u) ? 0u : run + 1u;
        const uint32_t lo = run > 65535u ? 65535u : run;
        run = ((m >> (b + 1)) & 1u) ? 0u : run + 1u;
        const uint32_t hi = run > 65535u ? 65535u : run;
        v[b / 2] = lo | (hi << 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
// the first cnt (1 .. 32) of a word's 32 PMLs to O: whole groups of 8 as 16-byte stores, the rest one by one
__device__ __forceinline__ void store_word(uint16_t *O, const uint4 (&g)[4], uint32_t cnt) {
    if (cnt >= 8u) __builtin_memcpy(O, &g[0], 16);
    if (cnt >= 16u) __builtin_memcpy(O + 8, &g[1], 16);
    if (cnt >= 24u) __builtin_memcpy(O + 16, &g[2], 16);
    if (cnt >= 32u) { __builtin_memcpy(O + 24, &g[3], 16); return; }
    const uint32_t q = cnt >> 3;
    const uint4 t = q == 0u ? g[0] : (q == 1u ? g[1] : (q == 2u ? g[2] : g[3]));
    const uint32_t x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i)
        if (8u * q + i < cnt) O[8u * q + i] = (uint16_t)(x[i >> 1] >> (16u * (i & 1u)));
}

// SHORT reads (mean length below kExpandTileLen): ONE WAVEFRONT takes 64 consecutive reads -- their vectors are one contiguous stretch
// of the output -- and every lane expands ITS OWN read, sequentially (match_len carried in a register: no search for the read, no
// look-back over words), eight bases per step, into a tile of the stretch in LDS; the wavefront then copies the tile out as aligned
// 16-byte stores, a contiguous kilobyte per instruction.  Long stretches slide the tile along (a lane resumes where it stopped).
// ~0.09 vector instructions per element, where one thread per group of eight (pml_expand_group_kernel below: a search for the read and
// a look-back per group) spends 0.33: profiles/r06_mask_path.txt has the four versions measured (0.81 / 0.68 / 1.65 TB/s / this one).
constexpr uint32_t kExpandTileElems = 8192;               // 16 KB of LDS per wavefront
constexpr uint64_t kExpandTileLen = 512;                  // launch_pml_expand: batches whose mean read length is below this
__global__ __launch_bounds__(64) void pml_expand_tile_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ offs,
                                                             uint64_t n_reads, uint32_t phase, uint16_t *__restrict__ out) {
    __shared__ __align__(16) uint16_t tile[kExpandTileElems];
    const uint32_t lane = threadIdx.x;
    const uint64_t R0 = (uint64_t)blockIdx.x * 64u;
    const uint64_t R1 = R0 + 64u < n_reads ? R0 + 64u : n_reads;
    const uint64_t rid = R0 + lane;
    const bool have = rid < n_reads;
    const uint64_t beg = have ? offs[rid] : 0, end = have ? offs[rid + 1] : 0;
    const uint32_t len = (uint32_t)(end - beg);
    const uint32_t *M = words + ((beg + phase) >> 5) + rid;
    const uint64_t B0 = offs[R0], B1 = offs[R1];           // uniform: the stretch
    uint32_t k = 0, run = 0;
    uint32_t wcur = 0, w0 = 0, w1 = 0;
    if (len) { w0 = M[0]; w1 = M[1]; }                     // (the array has a spare word at its end: M[w + 1] is always readable)
    auto one = [&](uint32_t o) {                           // one base: step k of this lane's read to tile[o]
        if ((k >> 5) != wcur) { wcur = k >> 5; w0 = w1; w1 = M[wcur + 1]; }
        run = ((w0 >> (k & 31u)) & 1u) ? 0u : run + 1u;
        tile[o] = (uint16_t)(run > 65535u ? 65535u : run);
        k += 1;
    };
    for (uint64_t A = B0 & ~7ull; A < B1; A += kExpandTileElems) {
        const uint64_t hi = A + kExpandTileElems < B1 ? A + kExpandTileElems : B1;
        if (k < len && beg + k < hi) {                      // this lane's bases inside the tile (beg + k >= A: the tiles advance in order)
            uint32_t o = (uint32_t)(beg + k - A);
            const uint32_t stop = (uint32_t)((end < hi ? end : hi) - A);
            while (o < stop && (o & 7u)) one(o++);
            while (o + 8u <= stop) {                        // eight bases: their bits from the word pair, one 16-byte LDS store
                if ((k >> 5) != wcur) { wcur = k >> 5; w0 = w1; w1 = M[wcur + 1]; }
                const uint32_t bits = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (k & 31u)) & 0xFFu;
                uint32_t v[8];
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e) {
                    run = ((bits >> e) & 1u) ? 0u : run + 1u;
                    v[e] = run > 65535u ? 65535u : run;
                }
                *reinterpret_cast<uint4 *>(&tile[o]) = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
                o += 8u;
                k += 8u;
            }
            while (o < stop) one(o++);
        }
        __syncthreads();
        const uint32_t nelem = (uint32_t)(hi - A);
        for (uint32_t g = lane; g * 8u < nelem; g += 64u) {
            const uint64_t p = A + (uint64_t)g * 8u;
            const uint4 q = *reinterpret_cast<const uint4 *>(&tile[g * 8u]);
            if (p >= B0 && p + 8u <= hi) {
                *reinterpret_cast<uint4 *>(out + p) = q;    // (out is 16-byte aligned: the launcher's check; p is a multiple of 8)
            } else {                                        // the stretch's first / last group: shared with the neighbouring wavefront's
                const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (uint32_t e = 0; e < 8u; ++e)
                    if (p + e >= B0 && p + e < hi) out[p + e] = (uint16_t)(x[e >> 1] >> (16u * (e & 1u)));
            }
        }
        __syncthreads();
    }
}

// Reads of a few hundred to two thousand bases: a block of 256 threads takes kExpandReads consecutive reads and its threads share
// their stretch of the output by groups of EIGHT CONSECUTIVE ELEMENTS aligned to 16 bytes: one 16-byte store per group.  The reads'
// offsets sit in LDS (the read a group starts in: a binary search over 65 entries); match_len before a group's first position comes
// from the bits below it in its mask word, or from the words before (as many as the run of matches is long: reads of 2048 bases and
// more take the wavefront-per-read kernel below).  Groups that straddle the stretch's ends are written element by element.
constexpr uint32_t kExpandReads = 64;
__global__ __launch_bounds__(256) void pml_expand_group_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ offs,
                                                               uint64_t n_reads, uint32_t phase, uint16_t *__restrict__ out) {
    __shared__ uint64_t s_off[kExpandReads + 1];
    const uint64_t R0 = (uint64_t)blockIdx.x * kExpandReads;
    const uint32_t nr = (uint32_t)(n_reads - R0 < kExpandReads ? n_reads - R0 : kExpandReads);
    for (uint32_t t = threadIdx.x; t <= nr; t += blockDim.x) s_off[t] = offs[R0 + t];
    __syncthreads();
    const uint64_t B0 = s_off[0], B1 = s_off[nr];
    for (uint64_t g = (B0 >> 3) + threadIdx.x; g * 8u < B1; g += blockDim.x) {
        const uint64_t p0 = g * 8u > B0 ? g * 8u : B0;                                  // this block's first element of the group ...
        const uint64_t p1 = g * 8u + 8u < B1 ? g * 8u + 8u : B1;                        // ... and one past its last
        // the read that holds p0: the largest i with s_off[i] <= p0 (empty reads are skipped by construction)
        uint32_t lo = 0, hi = nr - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (s_off[mid] <= p0) lo = mid; else hi = mid - 1u;
        }
        uint32_t i = lo;
        uint64_t beg = s_off[i], end = s_off[i + 1];
        const uint32_t *M = words + ((beg + phase) >> 5) + (R0 + i);
        uint32_t k = (uint32_t)(p0 - beg);
        uint32_t w = k >> 5, cur = M[w];
        uint32_t run;                                                                   // match_len before step k of read i
        {
            const uint32_t b = k & 31u, below = cur & ((1u << b) - 1u);
            if (below) {
                run = b - (32u - (uint32_t)__builtin_clz(below));
            } else {
                run = b;
                uint32_t ww = w;
                while (ww > 0) {
                    const uint32_t m = M[--ww];
                    if (m) { run += (uint32_t)__builtin_clz(m); break; }
                    run += 32u;
                }
            }
        }
        uint32_t v[8];
        const uint32_t e0 = (uint32_t)(p0 - g * 8u), e1 = (uint32_t)(p1 - g * 8u);
#pragma unroll
        for (uint32_t e = 0; e < 8u; ++e) {
            v[e] = 0;
            if (e >= e0 && e < e1) {
                while (g * 8u + e >= end) {                                             // the next read (empty ones have no positions)
                    i += 1;
                    beg = end;
                    end = s_off[i + 1];
                    M = words + ((beg + phase) >> 5) + (R0 + i);
                    k = 0; run = 0; w = 0; cur = M[0];
                }
                if ((k >> 5) != w) { w = k >> 5; cur = M[w]; }
                run = ((cur >> (k & 31u)) & 1u) ? 0u : run + 1u;
                v[e] = run > 65535u ? 65535u : run;
                k += 1;
            }
        }
        uint16_t *O = out + g * 8u;
        if (e0 == 0u && e1 == 8u) {
            *reinterpret_cast<uint4 *>(O) = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));   // (out is 16-byte aligned: the launcher's check)
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e)
                if (e >= e0 && e < e1) O[e] = (uint16_t)v[e];
        }
    }
}

// A wavefront per read (long reads): lane l takes word 64 i + l of round i -- the wavefront streams 4 KB of contiguous PMLs per
// round -- and match_len at the start of every word comes from a scan over (has a reset, bases after the last one) across the
// wavefront plus the carry of the rounds before.
__global__ __launch_bounds__(64) void pml_expand_wave_kernel(const uint32_t *__restrict__ words, const uint64_t *__restrict__ offs,
                                                             uint64_t n_reads, uint32_t phase, uint16_t *__restrict__ out) {
    const uint64_t t = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (t >= n_reads) return;
    const uint64_t beg = offs[t];
    const uint32_t len = (uint32_t)(offs[t + 1] - beg);
    const uint32_t *M = words + ((beg + phase) >> 5) + t;
    uint16_t *O = out + beg;
    uint32_t carry = 0;
    for (uint64_t w0 = 0; w0 * 32u < len; w0 += 64u) {
        const uint64_t k = (w0 + lane) * 32u;
        const bool live = k < len;
        const uint32_t m = live ? M[w0 + lane] : 0u;
        const uint32_t cnt = live ? (len - k < 32u ? (uint32_t)(len - k) : 32u) : 0u;
        // (h, tl): the word holds a reset / bases after its last reset (all of them if it holds none); bits beyond cnt are 0
        uint32_t h = m != 0u, tl = m ? cnt - 32u + (uint32_t)__builtin_clz(m) : cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ph = __shfl_up(h, d, 64), pt = __shfl_up(tl, d, 64);
            if ((int)lane >= d) { tl = h ? tl : pt + tl; h |= ph; }
        }
        const uint32_t ph = __shfl_up(h, 1, 64), pt = __shfl_up(tl, 1, 64);
        uint32_t run = lane == 0u ? carry : (ph ? pt : carry + pt);
        const uint32_t h63 = __shfl(h, 63, 64), t63 = __shfl(tl, 63, 64);
        carry = h63 ? t63 : carry + t63;
        if (live) {
            uint4 g[4];
            expand_word(m, run, g);
            store_word(O + k, g, cnt);
        }
    }
}

hipError_t launch_pml_expand(const uint32_t *d_words, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t phase,
                             uint16_t *d_out, hipStream_t stream) {
    if (n_reads == 0) return hipSuccess;
    if (n_bases / n_reads >= 2048 && n_reads <= 0x7FFFFFFFull) {
        hipLaunchKernelGGL(pml_expand_wave_kernel, dim3((unsigned)n_reads), dim3(64), 0, stream, d_words, d_offsets, n_reads, phase & 31u, d_out);
        return hipGetLastError();
    }
    const uint64_t blocks = (n_reads + kExpandReads - 1) / kExpandReads;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) {  // the group kernel stores 16 aligned bytes at a time: an odd vector takes the other kernel
        hipLaunchKernelGGL(pml_expand_wave_kernel, dim3((unsigned)n_reads), dim3(64), 0, stream, d_words, d_offsets, n_reads, phase & 31u, d_out);
        return hipGetLastError();
    }
    if (n_bases / n_reads < kExpandTileLen) {               // short reads: a wavefront per 64 reads, lanes on their own reads, out through an LDS tile
        const uint64_t wblocks = (n_reads + 63) / 64;
        hipLaunchKernelGGL(pml_expand_tile_kernel, dim3((unsigned)wblocks), dim3(64), 0, stream, d_words, d_offsets, n_reads, phase & 31u, d_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pml_expand_group_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_words, d_offsets, n_reads, phase & 31u, d_out);
    return hipGetLastError();
}

// PML vector -> masks: bit = (PML == 0).  WAVE = 0: one lane per read; 1: a wavefront per read, lane l on words l, l + 64, ...
template <int WAVE>
__global__ __launch_bounds__(256) void pml_to_mask_kernel(const uint16_t *__restrict__ pml, const uint64_t *__restrict__ offs,
                                                          uint64_t n_reads, uint32_t phase, uint32_t *__restrict__ words) {
    const uint64_t t = WAVE ? (uint64_t)blockIdx.x : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_reads) return;
    const uint64_t beg = offs[t];
    const uint32_t len = (uint32_t)(offs[t + 1] - beg);
    uint32_t *M = words + ((beg + phase) >> 5) + t;
    const uint16_t *P = pml + beg;
    for (uint64_t w = WAVE ? threadIdx.x : 0u; w * 32u < len; w += WAVE ? 64u : 1u) {
        const uint64_t k = w * 32u;
        const uint32_t cnt = len - k < 32u ? (uint32_t)(len - k) : 32u;
        uint32_t m = 0;
        if (cnt == 32u) {
            uint4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_memcpy(&g[q], P + k + 8 * q, 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m |= (uint32_t)((x[i] & 0xFFFFu) == 0u) << (8 * q + 2 * i);
                    m |= (uint32_t)((x[i] >> 16) == 0u) << (8 * q + 2 * i + 1);
                }
            }
        } else {
            for (uint32_t b = 0; b < cnt; ++b) m |= (uint32_t)(P[k + b] == 0) << b;
        }
        M[w] = m;
    }
}

hipError_t launch_pml_to_mask(const uint16_t *d_pml, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, uint32_t phase,
                              uint32_t *d_words, hipStream_t stream) {
    if (n_reads == 0) return hipSuccess;
    if (n_bases / n_reads >= 2048 && n_reads <= 0x7FFFFFFFull) {
        hipLaunchKernelGGL(pml_to_mask_kernel<1>, dim3((unsigned)n_reads), dim3(64), 0, stream, d_pml, d_offsets, n_reads, phase & 31u, d_words);
        return hipGetLastError();
    }
    const uint64_t blocks = (n_reads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pml_to_mask_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, d_pml, d_offsets, n_reads, phase & 31u, d_words);
    return hipGetLastError();
}

uint64_t pml_mask_words(uint64_t n_reads, uint64_t n_bases, uint32_t phase) { return ((n_bases + (phase & 31u)) >> 5) + n_reads + 1; }

// ------------------------------------------------- row-start checkpoints (setup)

template <int MODE>
__global__ __launch_bounds__(256) void chunk_sum_kernel(const uint8_t *__restrict__ rows, uint64_t r,
                                                        uint64_t n_chunks, uint64_t *__restrict__ sums) {
    // one wave per 64 chunks would coalesce better; this runs once per index load.
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_chunks) return;
    uint64_t lo = j << kPrefixShift, hi = lo + (1ull << kPrefixShift);
    if (hi > r) hi = r;
    uint64_t s = 0;
    for (uint64_t k = lo; k < hi; ++k) s += row_n<MODE>(load_row<MODE>(rows, k));
    sums[j] = s;
}

// Mode 7 upload: the file's 3-byte rows -> one aligned dword per row (top byte zero).
__global__ __launch_bounds__(256) void widen_rows_kernel(const uint8_t *__restrict__ packed, uint64_t r,
                                                         uint32_t *__restrict__ wide) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r) return;
    const uint8_t *p = packed + i * 3;
    wide[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

hipError_t widen_rows(const uint8_t *d_packed, uint64_t r, uint32_t *d_wide, hipStream_t stream) {
    const unsigned bt = 256;
    const uint64_t blocks = (r + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(widen_rows_kernel, dim3((unsigned)blocks), dim3(bt), 0, stream, d_packed, r, d_wide);
    return hipGetLastError();
}

// Sampled-thresholds -> regular-thresholds rows, once per index: the sampled format trades time for space (no id in
// the row; MoveStructure::get_id scans to the next checkpoint and walks the destination rows back, two dependent
// misses and ~1300 VALU instructions per LF here), a trade that makes no sense next to 288 GB of HBM.  So the ids are
// recovered ONCE, on the GPU, by that very get_id (tally_id above), and written out in the 8-byte layout of mode 6
// (n <= 511, 9-bit offsets, the same three threshold bits and character: everything fits); queries then run the mode-6
// kernels, state machines included, on identical rows -- identical answers, fast-forward and scan counts included.
// TM = 7: sampled-thresholds rows; TM = 5: sampled rows (no thresholds, 10-bit lengths: count / ZML queries only).
template <int TM>
__global__ __launch_bounds__(256) void expand_sampled_kernel(DevIndex ix, uint2 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < ix.r;
    const uint2 row = live ? load_row<TM>(ix.rows, i) : make_uint2(0u, 0u);
    const uint64_t id = tally_id<TM>(ix, live, i, row);                 // r on the reference's throws: LF_move rejects it
    if (live) {
        const uint32_t n = row_n<TM>(row), off = row_off<TM>(row), c = row_c<TM>(row);
        const uint32_t n16 = n | (row_thr<TM>(row, 1) << 11) | (row_thr<TM>(row, 2) << 12) | (c << 13);
        const uint32_t off16 = off | (row_thr<TM>(row, 0) << 11) | ((uint32_t)((id >> 32) & 0xFu) << 12);
        out[i] = make_uint2((uint32_t)id, n16 | (off16 << 16));
    }
}

// Blocked(-thresholds) -> regular(-thresholds) rows, once per index: get_id = blocked id + the (character, block) check
// point + first_runs[c + 1] (src/move_structure.cpp:91-102) evaluated for every row, written in the 8-byte layout.
// SM = 8: blocked-thresholds -> the regular-thresholds layout; SM = 2: blocked -> the regular layout (12-bit fields, no
// threshold bits; count / ZML queries only).
template <int SM>
__global__ __launch_bounds__(256) void expand_blocked_kernel(DevIndex ix, uint2 *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ix.r) return;
    const uint2 row = load_row<SM>(ix.rows, i);
    const uint64_t id = row_id<SM>(row, i, ix);
    const uint32_t n16 = row_n<SM>(row) | (row_thr<SM>(row, 1) << 11) | (row_thr<SM>(row, 2) << 12) | (row_c<SM>(row) << 13);
    const uint32_t off16 = row_off<SM>(row) | (row_thr<SM>(row, 0) << 11) | ((uint32_t)((id >> 32) & 0xFu) << 12);
    out[i] = make_uint2((uint32_t)id, n16 | (off16 << 16));
}

hipError_t expand_blocked_rows(int mode, const DevIndex &ix, void *d_rows6, hipStream_t stream) {
    const unsigned bt = 256;
    const uint64_t blocks = (ix.r + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (mode == 8)
        hipLaunchKernelGGL(expand_blocked_kernel<8>, dim3((unsigned)blocks), dim3(bt), 0, stream, ix, static_cast<uint2 *>(d_rows6));
    else
        hipLaunchKernelGGL(expand_blocked_kernel<2>, dim3((unsigned)blocks), dim3(bt), 0, stream, ix, static_cast<uint2 *>(d_rows6));
    return hipGetLastError();
}

hipError_t expand_sampled_rows(int mode, const DevIndex &ix, void *d_rows6, hipStream_t stream) {
    const unsigned bt = 256;
    const uint64_t blocks = (ix.r + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (mode == 7)
        hipLaunchKernelGGL(expand_sampled_kernel<7>, dim3((unsigned)blocks), dim3(bt), 0, stream, ix, static_cast<uint2 *>(d_rows6));
    else
        hipLaunchKernelGGL(expand_sampled_kernel<5>, dim3((unsigned)blocks), dim3(bt), 0, stream, ix, static_cast<uint2 *>(d_rows6));
    return hipGetLastError();
}

hipError_t build_row_start_ckpt(int mode, const uint8_t *d_rows, uint64_t r, uint64_t *d_ckpt,
                                hipStream_t stream) {
    // d_ckpt has n_chunks + 1 entries; entry j = sum of n over rows [0, 32 j).
    const uint64_t n_chunks = (r + (1ull << kPrefixShift) - 1) >> kPrefixShift;
    uint64_t *d_sums = nullptr;
    hipError_t e = hipMalloc(&d_sums, (n_chunks + 1) * sizeof(uint64_t));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_sums, 0, (n_chunks + 1) * sizeof(uint64_t), stream);
    if (e != hipSuccess) { (void)hipFree(d_sums); return e; }
    const unsigned bt = 256;
    const unsigned blocks = (unsigned)((n_chunks + bt - 1) / bt);
    if (mode == 6) hipLaunchKernelGGL(chunk_sum_kernel<6>, dim3(blocks), dim3(bt), 0, stream, d_rows, r, n_chunks, d_sums);
    else if (mode == 3) hipLaunchKernelGGL(chunk_sum_kernel<3>, dim3(blocks), dim3(bt), 0, stream, d_rows, r, n_chunks, d_sums);
    else { (void)hipFree(d_sums); return hipErrorInvalidValue; }
    e = hipGetLastError();
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    if (e == hipSuccess)
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_sums, d_ckpt, (int)(n_chunks + 1), stream);
    if (e == hipSuccess) e = hipMalloc(&d_temp, temp_bytes ? temp_bytes : 8);
    if (e == hipSuccess)
        e = hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_sums, d_ckpt, (int)(n_chunks + 1), stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (d_temp) (void)hipFree(d_temp);
    (void)hipFree(d_sums);
    return e;
}

}  // namespace movi