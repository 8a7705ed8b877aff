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
#include "movi_kernels.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>

namespace movi {


// ------------------------------------------------------------------ row decode
// A row is carried in registers as two dwords.
//   mode 6 (8 B, include/move_row.hpp:131-142; masks move_row_configs.hpp:34-51):
//     x = id[31:0]            y = n16 | offset16 << 16
//     n16:  [10:0] n, [11] thr1, [12] thr2, [15:13] c
//     off16:[10:0] offset, [11] thr0, [15:12] id[35:32]
//   mode 8 (6 B, move_row.hpp:128-142; masks move_row_configs.hpp:76-104):
//     x = id16 | n16 << 16    y = offset16
//     n16:  [9:0] n, [15:10] id[21:16]
//     off16:[9:0] offset, [12:10] c, [13] thr0, [14] thr1, [15] thr2
//   mode 7 (3 B, sampled-thresholds; move_row.hpp:122-127, masks move_row_configs.hpp:120-136): no id in the row
//     x = n8 | offset8 << 8 | cbyte << 16     cbyte: [0] offset bit 8, [1] n bit 8, [4:2] c, [5] thr0, [6] thr1, [7] thr2
//     (widened to one dword per row at upload, then expanded to mode-6 rows: expand_sampled_kernel)
template <int MODE>
__device__ __forceinline__ uint2 load_row(const uint8_t *rows, uint64_t i) {
    if (MODE == 6 || MODE == 3) {
        return *reinterpret_cast<const uint2 *>(rows + i * 8);
    } else if (MODE == 8 || MODE == 2) {
        // 6-byte rows: ONE unaligned 8-byte load of the bytes [6i-2, 6i+6) (for row 0: [0, 8)), shifted into
        // place -- never reads outside the table -- instead of three 2-byte loads
        const uint32_t lead = i ? 2u : 0u;
        unsigned long long v;
        __builtin_memcpy(&v, rows + i * 6 - lead, 8);
        v >>= 8u * lead;
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32) & 0xFFFFu);
    } else {
        // 3-byte rows, widened to one aligned dword per row when the index is uploaded (widen_rows_kernel); only
        // expand_sampled_kernel reads them: queries run on the mode-6 rows it writes
        return make_uint2(*reinterpret_cast<const uint32_t *>(rows + i * 4), 0u);
    }
}
template <int MODE> __device__ __forceinline__ uint32_t row_n(uint2 w) {
    if (MODE == 5) return (w.x & 0xFFu) | (((w.x >> 18) & 3u) << 8);      // sampled, no thresholds: configs :107-118
    if (MODE == 7) return (w.x & 0xFFu) | (((w.x >> 17) & 1u) << 8);
    if (MODE == 3) return w.y & 0xFFFu;                                    // regular, no thresholds: 12 bits (configs :21-32)
    return MODE == 6 ? (w.y & 0x7FFu) : ((w.x >> 16) & 0x3FFu);           // modes 8 and 2: 10 bits
}
template <int MODE> __device__ __forceinline__ uint32_t row_off(uint2 w) {
    if (MODE == 5) return ((w.x >> 8) & 0xFFu) | (((w.x >> 16) & 3u) << 8);
    if (MODE == 7) return ((w.x >> 8) & 0xFFu) | (((w.x >> 16) & 1u) << 8);
    if (MODE == 3) return (w.y >> 16) & 0xFFFu;
    return MODE == 6 ? ((w.y >> 16) & 0x7FFu) : (w.y & 0x3FFu);
}
template <int MODE> __device__ __forceinline__ uint32_t row_c(uint2 w) {
    if (MODE == 5) return (w.x >> 20) & 15u;
    if (MODE == 7) return (w.x >> 18) & 7u;
    return (MODE == 6 || MODE == 3) ? ((w.y >> 13) & 7u) : ((w.y >> 10) & 7u);
}
// threshold bit k in {0,1,2} (MoveRow::get_threshold, move_row.hpp:304-347)
template <int MODE> __device__ __forceinline__ uint32_t row_thr(uint2 w, uint32_t k) {
    if (MODE == 5 || MODE == 3 || MODE == 2) return 0u;   // no thresholds in these index types
    if (MODE == 6) {
        // k=0 -> off16 bit 11 (y bit 27); k=1 -> n16 bit 11; k=2 -> n16 bit 12
        uint32_t sh = (k == 0) ? 27u : (10u + k);
        return (w.y >> sh) & 1u;
    } else if (MODE == 8) {
        return (w.y >> (13u + k)) & 1u;
    } else {
        return (w.x >> (21u + k)) & 1u;
    }
}
// MoveStructure::get_id, src/move_structure.cpp:91-102
template <int MODE>
__device__ __forceinline__ uint64_t row_id(uint2 w, uint64_t idx, const DevIndex &ix) {
    static_assert(MODE == 6 || MODE == 8 || MODE == 3 || MODE == 2, "the sampled modes have no id in the row: tally_id()");
    if (MODE == 6 || MODE == 3) {
        return (uint64_t)w.x | ((uint64_t)(w.y >> 28) << 32);
    } else {
        uint64_t bid = (uint64_t)(w.x & 0xFFFFu) | ((uint64_t)(w.x >> 26) << 16);
        if (MODE == 2) bid |= (uint64_t)((w.y >> 14) & 3u) << 22;          // two more id bits in `offset` (move_row.hpp:274-280)
        if (idx == ix.end_bwt_idx) return bid;
        uint32_t c = row_c<MODE>(w);
        const uint64_t blk = ix.block_shift != 0xFFFFFFFFu ? (idx >> ix.block_shift) : idx / ix.block_size;
        const uint64_t slot = (uint64_t)c * ix.n_blocks + blk;
        const uint32_t base = ix.id_blocks[slot];                           // check point of (character, block)
        return bid + (uint64_t)base + ix.first_runs[c + 1];
    }
}

// ---- reposition_thresholds, src/move_structure_query.cpp:513-601: which threshold applies.
// Read base code a and row code c are alphamap values: 0..3, or 1..4 on a separators index (code 0 = '%').
// Slot of a DNA row: alphamap_3[c - sep][a - sep] (src/utils.cpp:5-8) = (a - sep) - (a > c) for a != c.
__device__ __forceinline__ uint32_t thr_slot(uint32_t sep, uint32_t a, uint32_t c) {
    return (a - sep - (uint32_t)(a > c)) & 3u;
}
// end_bwt_idx_thresholds[a - sep] (:534-535).  The four values are clamped to 32 bits once per kernel (offsets
// are < 2^11, so `off >= t` is unchanged) and picked with selects: written as a ladder over the kernel-argument
// array, hipcc turned the pick into an indexed LOAD from the kernarg segment plus `s_waitcnt vmcnt(0)` -- one
// more memory round trip in every iteration of the latency-bound state machine (c3: 39.6 -> 32.2 Gbases/s).
struct EndThr { uint32_t e0, e1, e2, e3; };
__device__ __forceinline__ EndThr end_thresholds(const DevIndex &ix) {
    auto clamp = [](uint64_t v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; };
    return EndThr{clamp(ix.end_thr[0]), clamp(ix.end_thr[1]), clamp(ix.end_thr[2]), clamp(ix.end_thr[3])};
}
__device__ __forceinline__ uint32_t end_threshold(uint32_t sep, const EndThr &e, uint32_t a) {
    const uint32_t k = a - sep;
    const uint32_t lo = (k & 1u) ? e.e1 : e.e0, hi = (k & 1u) ? e.e3 : e.e2;
    return (k & 2u) ? hi : lo;
}
// separators_thresholds[separators_thresholds_map[idx]].values[a - 1] (:540-541) for a row of the separator;
// a missing key reads entry 0 of an empty-initialised map in the reference: 0 here.  Rare path: binary search.
__device__ __forceinline__ uint32_t separator_threshold(const DevIndex &ix, uint64_t idx, uint32_t a) {
    uint32_t lo = 0, hi = ix.n_sep;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ix.sep_rows[mid] < idx) lo = mid + 1; else hi = mid;
    }
    if (lo >= ix.n_sep || ix.sep_rows[lo] != idx) return 0u;
    const uint2 v = ix.sep_vals[lo];
    const uint32_t k = a - 1u;
    const uint32_t w = (k & 2u) ? v.y : v.x;
    return (k & 1u) ? (w >> 16) : (w & 0xFFFFu);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// Per-read error codes (the reference throws in each of these cases).
enum : uint32_t {
    kErrNone = 0,
    kErrIdRange = 1,       // LF destination >= r              (move_structure.cpp:63-65)
    kErrFastForward = 2,   // >= 65535 fast-forward steps      (move_structure.cpp:72-75)
    kErrNoRunBelow = 3,    // reposition_down found no run     (move_structure_query.cpp:582-586)
    kErrNoRunAbove = 4,    // reposition_up found no run       (move_structure_query.cpp:594-598)
};

// Control-flow note (ROCm 7.2 / gfx950): every data-dependent loop below is written
// as a WAVE-UNIFORM loop (`while (__any(pred))`) with a predicated body and all
// loop-carried state in integer VGPRs.  A divergent `while` whose result is consumed
// as a boolean after the loop (`found = (c == a)`) was miscompiled by hipcc: the exit
// compare of the LAST iteration (vcc) was reused for lanes that had left the loop
// earlier.  Uniform loops are also the cheaper form on a 64-wide wavefront.
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// MoveStructure::get_id for the sampled ("tally") mode 7, src/move_structure.cpp:104-283, forward branch (the
// reference fixes forward_direciton = true, :146): the row holds no id; every tally_cp rows the id of the latest
// run of each character is kept.  The id of row idx = the id stored at the next checkpoint for idx's character,
// walked back over the destination rows by the BWT positions of that character between idx and the stored run.
// Wave-uniform loops, predicated per lane (see the control-flow note above).  Returns r on the reference's throws.
// IdxT = uint32_t when the table has fewer than 2^32 rows.  Rows are read four at a time as the aligned 16-byte
// group that holds them (the widened table has 16 bytes of slack, so the last group may be read whole).
template <int TM, typename IdxT>                          // TM: 7 = sampled-thresholds rows, 5 = sampled rows
__device__ __forceinline__ uint64_t tally_id_t(const DevIndex &ix, bool live, uint64_t idx64, uint2 row) {
    const IdxT idx = (IdxT)idx64, r = (IdxT)ix.r, end_row = (IdxT)ix.end_bwt_idx;
    const uint32_t ci = row_c<TM>(row);
    const IdxT cp = (IdxT)ix.tally_cp;
    IdxT id = live ? 0 : idx;                            // lanes that take no step keep their row (callers store the result)
    uint32_t walk = 0;                                   // 1 while the lane still scans / walks
    uint32_t bad = 0;                                    // one of the reference's throws (or an id >= r): returns r
    IdxT next_cp = idx;
    uint32_t rows_until = 0;                             // <= tally_cp rows of <= 511 positions
    uint32_t last_n = 0, last_off = 0, last_is_idx = 1;
    if (live && idx != end_row) {                        // '$' goes to row 0 (:106-108)
        const IdxT ta = idx / cp;
        const uint64_t *tl = ix.tally + (uint64_t)ci * ix.tally_len;
        uint64_t raw;
        if (idx == r - 1) raw = tl[ix.tally_len - 1];    // :114-117
        else if (ta * cp == idx) raw = tl[ta];           // :121-124
        else {
            next_cp = (ta + 1) * cp;
            if (next_cp >= r) next_cp = r - 1;           // :137-139
            raw = tl[ta + 1];
            walk = 1;
        }
        id = (IdxT)raw;
        if (raw >= ix.r) { bad = 1; walk = 0; }          // LF_move throws on it (move_structure.cpp:63-65)
    }
    // rows of idx's character in [idx, next_cp) (:168-174) -- row idx itself is one of them -- and the row at
    // next_cp, group by group
    uint32_t scan = walk;
    IdxT g = idx & ~(IdxT)3;
    uint32_t wn = 0;                                     // the row at next_cp
    while (wave_any(scan != 0u)) {
        if (scan) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ix.rows + (uint64_t)g * 4);
            const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint2 w = make_uint2(x[t], 0u);
                const IdxT it = g + (IdxT)t;
                if (it >= idx && it < next_cp && it != end_row && row_c<TM>(w) == ci) {
                    rows_until += row_n<TM>(w);
                    last_n = row_n<TM>(w);
                    last_off = row_off<TM>(w);
                    last_is_idx = (it == idx) ? 1u : 0u;
                }
                wn = (it == next_cp) ? x[t] : wn;
            }
            g += 4;
            scan = (g <= next_cp) ? 1u : 0u;
        }
    }
    // the stored id is idx's own (:178-180), or the walk starts at row id with `offset` positions to spare (:186-209)
    uint32_t back = 0, offset = 0;
    if (walk) {
        const uint2 wnr = make_uint2(wn, 0u);
        const uint32_t same = (next_cp != end_row && row_c<TM>(wnr) == ci) ? 1u : 0u;
        if (!(last_is_idx && !same)) {
            offset = row_off<TM>(wnr);
            if (!same) { rows_until -= last_n; offset = last_off; }       // :194-197
            back = 1;
        }
    }
    // :200-219: row id first (offset >= n(id) throws; offset >= rows_until: id it is; else rows_until -= offset + 1
    // and on to id - 1), then `while (rows_until) { rows_until >= n(id) ? (rows_until -= n(id), id--) : rows_until = 0 }`
    uint32_t first = 1;
    while (wave_any(back != 0u)) {
        if (back) {
            const IdxT gb = id & ~(IdxT)3;
            const uint4 v = *reinterpret_cast<const uint4 *>(ix.rows + (uint64_t)gb * 4);
            const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                const uint32_t nrow = row_n<TM>(make_uint2(x[t], 0u));
                if (back && (gb + (IdxT)t) == id) {
                    uint32_t step_down = 0;
                    if (first) {
                        first = 0;
                        if (offset >= nrow) { bad = 1; back = 0; }
                        else if (offset >= rows_until) back = 0;
                        else { rows_until -= offset + 1; step_down = 1; }
                    } else if (rows_until == 0) {
                        back = 0;
                    } else if (rows_until >= nrow) {
                        rows_until -= nrow;
                        step_down = 1;
                    } else {
                        rows_until = 0;
                        back = 0;
                    }
                    if (step_down) { if (id == 0) { bad = 1; back = 0; } else id -= 1; }
                }
            }
        }
    }
    return bad ? ix.r : (uint64_t)id;
}
template <int TM>
__device__ __forceinline__ uint64_t tally_id(const DevIndex &ix, bool live, uint64_t idx, uint2 row) {
    if (ix.idx32) return tally_id_t<TM, uint32_t>(ix, live, idx, row);           // wave-uniform choice
    return tally_id_t<TM, uint64_t>(ix, live, idx, row);
}

// LF_move + fast_forward.  On entry `row` is rows[idx]; on exit it is the row of
// the new idx.  `live` lanes take the step; returns a kErr* code (0 = ok) per lane.
template <int MODE>
__device__ __forceinline__ uint32_t lf_step(const DevIndex &ix, bool live, uint64_t &idx, uint32_t &off,
                                            uint2 &row, uint32_t &ff_total) {
    uint32_t errc = kErrNone;
    uint64_t j = idx;
    uint32_t n = 0, ff = 0;
    uint32_t going = 0;
    if (live) {
        j = row_id<MODE>(row, idx, ix);
        if (j >= ix.r) {                                // move_structure.cpp:63-65
            errc = kErrIdRange;
            j = idx;
        } else {
            off += row_off<MODE>(row);
            row = load_row<MODE>(ix.rows, j);           // THE dependent random gather
            n = row_n<MODE>(row);
            going = (j < ix.r - 1 && off >= n) ? 1u : 0u;
        }
    }
    // fast_forward :524-545: the next row sits in the line the gather just brought in (L2 hit)
    while (wave_any(going != 0u)) {
        if (going) {
            const uint64_t jj = j + 1;
            const uint2 w = load_row<MODE>(ix.rows, jj < ix.r ? jj : ix.r - 1);
            off -= n;
            j += 1;
            ff += 1;
            row = w;
            n = row_n<MODE>(row);
            going = (j < ix.r - 1 && off >= n && ff < 65535u) ? 1u : 0u;
        }
    }
    if (ff >= 65535u) errc = kErrFastForward;           // move_structure.cpp:72-75
    ff_total += ff;
    idx = j;
    return errc;
}

// Two independent LF_moves (the two ends of a backward-search interval) advanced together:
// both gathers are issued before either result is needed and the two fast-forwards share one
// wave-uniform loop, so an interval step costs the trips of ONE walker.
template <int MODE>
__device__ __forceinline__ uint32_t lf_step2(const DevIndex &ix, bool live, uint64_t &ia, uint32_t &offa, uint2 &rowa,
                                             uint64_t &ib, uint32_t &offb, uint2 &rowb, uint32_t &ff_total) {
    uint32_t errc = kErrNone;
    uint64_t ja = ia, jb = ib;
    uint32_t na = 0, nb = 0, ffa = 0, ffb = 0, ga = 0, gb = 0;
    if (live) {
        ja = row_id<MODE>(rowa, ia, ix);
        jb = row_id<MODE>(rowb, ib, ix);
        if (ja >= ix.r || jb >= ix.r) {                 // move_structure.cpp:63-65
            errc = kErrIdRange;
            ja = ia; jb = ib;
        } else {
            offa += row_off<MODE>(rowa);
            offb += row_off<MODE>(rowb);
            rowa = load_row<MODE>(ix.rows, ja);
            rowb = load_row<MODE>(ix.rows, jb);
            na = row_n<MODE>(rowa);
            nb = row_n<MODE>(rowb);
            ga = (ja < ix.r - 1 && offa >= na) ? 1u : 0u;
            gb = (jb < ix.r - 1 && offb >= nb) ? 1u : 0u;
        }
    }
    while (wave_any((ga | gb) != 0u)) {                 // fast_forward :524-545, both walkers
        uint2 wa = rowa, wb = rowb;
        if (ga) wa = load_row<MODE>(ix.rows, ja + 1);
        if (gb) wb = load_row<MODE>(ix.rows, jb + 1);
        if (ga) {
            offa -= na; ja += 1; ffa += 1; rowa = wa; na = row_n<MODE>(rowa);
            ga = (ja < ix.r - 1 && offa >= na && ffa < 65535u) ? 1u : 0u;
        }
        if (gb) {
            offb -= nb; jb += 1; ffb += 1; rowb = wb; nb = row_n<MODE>(rowb);
            gb = (jb < ix.r - 1 && offb >= nb && ffb < 65535u) ? 1u : 0u;
        }
    }
    if (ffa >= 65535u || ffb >= 65535u) errc = kErrFastForward;   // move_structure.cpp:72-75
    ff_total += ffa + ffb;
    ia = ja; ib = jb;
    return errc;
}

// Classifier::classify (src/classifier.cpp:99-143) as a running reduction over the values a lane emits:
// bins of bin_width in emission order, the last bin absorbing a remainder shorter than bin_width.
// CLS template parameter of the PML kernels: 0 = PML vector only, 1 = vector + bins, 2 = bins only.
struct ClsState {
    uint32_t cur = 0, above = 0, below = 0, bin = 0, nb = 1, next_cut = 0;
    uint64_t sum = 0;
    __device__ __forceinline__ void init(uint32_t len, uint32_t w) {
        nb = w ? len / w : 0;
        if (nb == 0) nb = 1;
        next_cut = nb > 1 ? w : len;
    }
    __device__ __forceinline__ void add(uint32_t val, uint32_t k, uint32_t len, uint32_t w, uint32_t thr) {
        cur = val > cur ? val : cur;
        if (k + 1 == next_cut) {
            above += cur >= thr ? 1u : 0u;
            below += cur >= thr ? 0u : 1u;
            sum += cur;
            cur = 0;
            bin += 1;
            next_cut = (bin + 1 < nb) ? next_cut + w : len;
        }
    }
    __device__ __forceinline__ void store(const ClsArgs &c, uint64_t rid, bool failed) const {
        c.above[rid] = failed ? 0u : above;
        c.below[rid] = failed ? 0u : below;
        c.sum_max[rid] = failed ? 0ull : sum;
    }
};

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

// VARIANT 7 ("flat lane state machine"): the SIMT-friendly form of the walk for batches with few
// reads.  In variants 0/1 all 64 lanes advance base by base, so every step costs the wave
// 1 + max_lanes(fast-forwards) + max_lanes(scan rows) dependent memory round trips.  Here each lane
// runs its own little automaton (states: fast-forwarding to / resolving a base, scanning down,
// scanning up, done) and every iteration of the wave-uniform loop issues exactly ONE row load per
// lane for whatever that lane needs next; a lane that needs three extra rows falls three
// iterations behind its neighbours instead of stalling them.  The first version of this kernel used
// ordinary nested branches (variant 2, removed: 11-12 % slower -- latency-bound at 1-2 waves per
// SIMD it spent ~37 % of its wave cycles issuing ~270 instructions per iteration, many of them
// exec-mask bookkeeping and phi copies); this one is straight-line predicated code: every state
// update is a select, the only branches guard memory side effects, and the base code of step k
// is looked up in LDS when k advances -- the lookup then overlaps the next row gather instead of
// sitting between the row's arrival and the compare.
// IdxT = uint32_t when the table has fewer than 2^32 rows (half the index arithmetic).
template <int MODE, typename IdxT, int CLS>
__global__ __launch_bounds__(256) void pml_kernel_flat(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order,
                                                       ClsArgs cls) {
    enum : uint32_t { sFF = 0, sDown = 1, sUp = 2, sDone = 3 };
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, repo_total = 0, failed = 0;
    const EndThr ethr = end_thresholds(ix);
    const bool valid = t < n_reads;
    const uint64_t rid = (valid && order) ? order[t] : t;
    const uint64_t beg = valid ? offs[rid] : 0;
    const uint32_t len = valid ? (uint32_t)(offs[rid + 1] - beg) : 0;   // reads are shorter than 2^32 (checked on the host)
    const uint8_t *R = bases + beg;
    uint16_t *O = out + beg;
    const uint32_t packed_end = len & ~7u;
    const IdxT r1 = (IdxT)(ix.r - 1), end_row = (IdxT)ix.end_bwt_idx;

    auto load_chunk = [&](uint32_t kk) -> uint64_t {
        uint64_t v = 0;
        if (beg + len >= (uint64_t)kk + 8) {
            __builtin_memcpy(&v, R + len - kk - 8, 8);    // may start before R: still inside `bases`
        } else {
            for (uint32_t i = 0; i < len - kk; ++i) v |= (uint64_t)R[len - 1 - kk - i] << (8 * (7 - i));
        }
        return v;
    };

    uint32_t st = len > 0 ? sFF : sDone;
    IdxT need = r1;                                       // ReadProcessor::reset_process :69-70
    uint32_t k = 0;
    uint32_t ml = 0, ff_run = 0;
    uint32_t off = row_n<MODE>(load_row<MODE>(ix.rows, r1)) - 1;
    uint64_t rb = st != sDone ? load_chunk(0) : 0;
    uint32_t a = s_code[(uint32_t)(rb >> 56) & 0xFFu];    // code of the base of step k (k = 0)
    uint4 pk = make_uint4(0, 0, 0, 0);
    ClsState cs;
    if (CLS) cs.init(len, cls.bin_width);

    while (wave_any(st != sDone)) {
        uint2 row = make_uint2(0, 0);
        if (st != sDone) row = load_row<MODE>(ix.rows, need);
        const uint32_t n = row_n<MODE>(row), c = row_c<MODE>(row);
        // predicates as 0/1 integers combined with & | (no short-circuit control flow)
        const uint32_t isFF = st == sFF, isDown = st == sDown, isUp = st == sUp;
        // fast_forward, move_structure.cpp:524-545
        const uint32_t ffm = isFF & (uint32_t)(need < r1) & (uint32_t)(off >= n);
        const uint32_t ff_over = ffm & (uint32_t)(ff_run + 1 >= 65535u);  // :72-75
        const uint32_t resolved = isFF & (ffm ^ 1u);
        // the base of step k against the row (read_processor.cpp:188-238)
        const uint32_t illegal = a == 0xFFu, match = c == a;
        const uint32_t mism = resolved & (illegal ^ 1u) & (match ^ 1u);
        // reposition_thresholds, src/move_structure_query.cpp:513-601
        const uint32_t kk = thr_slot(ix.sep, a, c);                       // alphamap_3[c][a]
        uint32_t thr = row_thr<MODE>(row, kk > 2u ? 2u : kk) ? n : 0u;
        if (ix.sep) {                                                     // a row of the separator: side table
            if (mism & (uint32_t)(c == 0u) & (uint32_t)(need != end_row)) thr = separator_threshold(ix, (uint64_t)need, a);
        }
        const uint32_t down = (uint32_t)(off >= ((need == end_row) ? end_threshold(ix.sep, ethr, a) : thr));
        const uint32_t at_last = need >= r1, at_first = need == 0;
        const uint32_t repo_edge = mism & (down ? at_last : at_first);
        // reposition_down :211-232 / reposition_up :188-209, one row per iteration
        const uint32_t scanning = isDown | isUp;
        const uint32_t hit = scanning & match;
        const uint32_t scan_edge = scanning & (hit ^ 1u) & (isDown ? at_last : at_first);
        const uint32_t emit = (resolved & (illegal | match)) | hit;
        const uint32_t errc = ff_over ? kErrFastForward
                              : (repo_edge ? (down ? kErrNoRunBelow : kErrNoRunAbove)
                                 : (scan_edge ? (isDown ? kErrNoRunBelow : kErrNoRunAbove) : kErrNone));
        // ---- state update, all selects
        ml = resolved ? (match ? ml + 1 : 0u) : ml;
        ff_total += resolved ? ff_run : 0u;
        ff_run += ffm;
        repo_total += mism;
        scan_total += scanning;
        off = ffm ? off - n : (hit ? (isDown ? 0u : n - 1) : off);        // read_processor.cpp:223
        const uint32_t step_fwd = ffm | (mism & down) | (scanning & (hit ^ 1u) & isDown);
        const uint32_t step_back = (mism & (down ^ 1u)) | (scanning & (hit ^ 1u) & isUp);
        IdxT need_next = need + step_fwd - step_back;
        uint32_t st_next = mism ? (down ? sDown : sUp) : st;
        if (emit) {
            const uint32_t val = ml > 65535u ? 65535u : ml;               // MoveQuery::add_ml
            if (CLS) cs.add(val, k, len, cls.bin_width, cls.thr);
            if (CLS == 2) {
                // verdict bins only
            } else if (k >= packed_end) {
                O[k] = (uint16_t)val;
            } else {
                pk.x = (pk.x >> 16) | (pk.y << 16);
                pk.y = (pk.y >> 16) | (pk.z << 16);
                pk.z = (pk.z >> 16) | (pk.w << 16);
                pk.w = (pk.w >> 16) | (val << 16);
                if ((k & 7) == 7) __builtin_memcpy(O + (k - 7), &pk, 16);
            }
            k += 1;
            if (k == len) {
                st_next = sDone;
            } else {
                const uint64_t j = row_id<MODE>(row, need, ix);      // LF_move, move_structure.cpp:59-67
                if (j >= ix.r) {
                    failed = kErrIdRange;
                    st_next = sDone;
                } else {
                    off += row_off<MODE>(row);
                    need_next = (IdxT)j;
                    ff_run = 0;
                    st_next = sFF;
                    if ((k & 7) == 0) rb = load_chunk(k);
                    a = s_code[(uint32_t)(rb >> (8 * (7 - (k & 7)))) & 0xFFu];
                }
            }
        }
        if (errc) { failed = errc; st_next = sDone; }
        need = need_next;
        st = st_next;
    }
    if (failed && CLS != 2) {
        for (uint32_t i = 0; i < len; ++i) O[i] = 0;
    }
    if (CLS && valid) cs.store(cls, rid, failed != 0u);
    if (valid && err) err[rid] = (uint8_t)failed;
    const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), rpw = wave_sum(repo_total),
                   erw = wave_sum(failed ? 1u : 0u);
    if ((threadIdx.x & 63) == 0 && stats) {
        if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
        if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
        if (rpw) atomicAdd(&stats->repositions, (unsigned long long)rpw);
        if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
    }
}

template <int MODE>
__device__ __forceinline__ void load_window(const uint8_t *rows, uint64_t wbase, uint2 (&w)[4]) {
    static_assert(MODE == 6 || MODE == 3, "queries run on 8-byte regular(-thresholds) rows only (the other types are expanded at upload)");
    uint4 p0, p1;
    __builtin_memcpy(&p0, rows + wbase * 8, 16);
    __builtin_memcpy(&p1, rows + wbase * 8 + 16, 16);
    w[0] = make_uint2(p0.x, p0.y); w[1] = make_uint2(p0.z, p0.w);
    w[2] = make_uint2(p1.x, p1.y); w[3] = make_uint2(p1.z, p1.w);
}
__device__ __forceinline__ uint2 win_sel(const uint2 (&w)[4], uint32_t q) {
    const uint2 lo = (q & 1u) ? w[1] : w[0];
    const uint2 hi = (q & 1u) ? w[3] : w[2];
    return (q & 2u) ? hi : lo;
}

// VARIANT 10 ("flat state machine + row window, software-pipelined"; with HA < 0 = variant 14, the default everywhere):
// variant 7 with
//   * the 4-row WINDOW around the row it needs fetched instead of the row (32 B / 24 B of the same cache line,
//     the same single L2 request) and up to HA cheap fast-forward / scan hops taken inside it before the full
//     automaton step, so most neighbour rows cost no memory round trip (+6 % on the pangenome, +16-19 % on
//     random tables over variant 7);
//   * the next window's address computed from the row alone (all selects) and its load issued BEFORE the
//     step's bookkeeping (PML packing and stores, bins, counters, base decode), which then runs under the
//     gather's latency; the load is unpredicated and branch-free (the table's last window is pulled back to
//     rows [r-4, r); finished lanes re-read window 0) -- with a predicated two-path fetch hipcc parked a
//     `s_waitcnt vmcnt(0)` right behind the load and the overlap was gone;
//   * read chunks double-buffered, 16 bases per fetch: the 16 bases after the current ones are fetched when a
//     16-group is entered, so the chunk load (always an L2 miss: its line was evicted long ago) overlaps sixteen row
//     gathers and costs 1/16 instead of 1/8 line per base; PMLs leave as paired 16-byte stores.
// Measured (100 k x 10 kbp, Gbases/s, pangenome / random table): unpipelined window kernel (variant 8, removed)
// 36.4 / 33.6; chunk double-buffering alone 35.1 / 32.4; pipelined HA = 1 / 2 / 3: 40.2 / 39.6 / 38.9 (pangenome),
// 36.3 / 36.6 / 36.2 (random) -> HA = 2 shipped.  Hops after the step (HC > 0) measured slower and are gone.
// REFILL = 1 ("lane refill", variant 13; staged kernels only): the same automaton as a PERSISTENT grid of
// num_cus x waves_per_cu wavefronts (any grid works: blocks that start late find the counter further on) whose lanes take a
// new read when they have finished one.  Without it a wavefront runs
// until its slowest lane is done, and reads differ a lot: a substitution costs a read about a dozen repositions (the walk
// needs ~15 bases to fall back into step with the text), so on 1 M x 150 bp with 1 % substitutions only 68 % of the lane
// iterations do work on the look-ahead rows (tools/iter_model.c predicts the figure).
//   * Reads come from ONE global ticket counter (DevStats::ticket) in chunks of 16 consecutive reads -- an atomic per chunk --
//     into a POOL of upcoming reads per wavefront: 64 slots across the lanes (read number, where its bases start, its
//     length), consumed in ring order by whichever lanes are idle at a switch (ds_bpermute).  A chunk takes two switches to
//     arrive -- its ticket is drawn at one, its offsets are requested at the next, they join the pool at the one after -- so
//     nothing about the pool waits on memory by itself.  (The first version dealt the reads to the wavefronts statically and
//     let every lane hold its next read: the slowest wavefront ended 9 - 22 % behind the mean, and busy lanes sat on
//     reservations idle lanes could have used: profiles/r04_lane_refill.txt.)
//   * Refills come in BATCHES: a switch stages the new read's bases into the lane's LDS stretch and takes its first K
//     bases from the top-of-walk table -- memory round trips in which the whole wavefront stands still -- so idle lanes
//     wait until DevIndex::refill_batch of them (or every lane that still has work) can switch together.
//   * Results of a read (error byte, bins, zero-fill on failure) are written when its lane switches (or at the end).
// `order` is not supported (longest-first ordering is what refill replaces).
// SEG (segment-parallel long reads, movi_kernels.hpp): 0 = a lane walks a read; 1 = a lane walks one SEGMENT of a read
// from the state every read starts in (K1: its "read" is the segment -- bases at seg_in, PMLs to seg_out --, it leaves a
// checkpoint of its state and counters every 32 bases and its final state, reports an invariant violation in its
// segment's flag instead of err[] / zero-filling, and adds nothing to the global fast-forward / scan counters: which
// part of its work belongs to the read's real walk is only known after K2); 2 = whole reads again, but only those in
// seg.read_fail (K3).
// AHD (look-ahead rows, DevIndex::rows2; staged kernels only): the window comes from the table's second copy, together
// with the look-ahead entries of its four rows (the other half of the same 128-byte line).  When the step's emitted base is followed by a base that
// matches at the LF target j = id(row) without a fast-forward -- known from the entry: c(j), n(j) against the offset --
// the walk emits that PML as well and goes straight on to id(j): two bases for one gather.  Everything else (a
// mismatch, a fast-forward at j, the read's end, an invalid entry) takes the one-base step it always took.
// (Fetching only the entry of the row the window was fetched FOR -- 8 bytes instead of 32 -- misses the steps that end on a
// neighbour after a fast-forward or scan: 68.5 against 74.4 Gbases/s on c2, 54.1 against 62.8 on the random table.)
// (Round 4 built the same with entries that look TWO rows ahead -- "chain rows", 16 bytes per row, up to three bases per gather:
// bit-exact, lane iterations per base 0.68 -> 0.56 on c2 as tools/iter_model.c predicts, and 10 % SLOWER there, 38 % slower on a
// 113 M-row real BWT: twice the bytes, six loads and 18 % more instructions per iteration.  Measured with PMC
// (profiles/r04_chain_rows.txt) and removed again; the code is in the history: commit b8d3f4e.)
// PSH = 1 (round 4: "pair-shared gathers"; staged kernels, plain and look-ahead rows): the two lanes of a pair (2i, 2i + 1) fetch
// their windows TOGETHER -- one load instruction brings the even lane's window (each lane one 16-byte half), the next the odd
// lane's, and one exchange across the pair (DPP quad_perm) hands every lane the half it is missing.  Same loads per lane,
// same bytes -- but the two lanes' requests for adjacent bytes of a page are ONE address translation and ONE 32-byte access,
// where a lane's two 16-byte loads are two of each: tools/tlb_bench (profiles/r04_pair_shared_gather_microbench.txt), 32-byte
// window per chain step: 8 GB table 27.3 -> 49.4 G/s (the 8-byte gather rate), 2 GB 44.3 -> 54.8, 134 MB 52.0 -> 59.5.
__device__ __forceinline__ uint32_t pair_swap(uint32_t v) {      // the other lane's value: lanes 2i <-> 2i + 1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1, 0, 3, 2]
}
// What lanes 2i / 2i+1 loaded as (r1: their half of the EVEN lane's 32 bytes, r2: of the ODD lane's) -> this lane's own 32 bytes
__device__ __forceinline__ void pair_assemble(uint32_t odd, const uint4 &r1, const uint4 &r2, uint2 (&w)[4]) {
    // even: first half = own r1, second = the odd lane's r1;  odd: first half = the even lane's r2, second = own r2.
    // (Every lane makes every exchange -- a DPP read of a lane that sits out a branch returns nothing --, then selects.)
    const uint32_t give_x = odd ? r1.x : r2.x, give_y = odd ? r1.y : r2.y, give_z = odd ? r1.z : r2.z, give_w = odd ? r1.w : r2.w;
    const uint32_t got_x = pair_swap(give_x), got_y = pair_swap(give_y), got_z = pair_swap(give_z), got_w = pair_swap(give_w);
    w[0] = odd ? make_uint2(got_x, got_y) : make_uint2(r1.x, r1.y);
    w[1] = odd ? make_uint2(got_z, got_w) : make_uint2(r1.z, r1.w);
    w[2] = odd ? make_uint2(r2.x, r2.y) : make_uint2(got_x, got_y);
    w[3] = odd ? make_uint2(r2.z, r2.w) : make_uint2(got_z, got_w);
}
template <int MODE, typename IdxT, int HA, int CLS, int SEP, int REFILL, int SEG = 0, int STG = 0, int AHD = 0, int PSH = 0, int RING = 0>
__global__ __launch_bounds__(256) void pml_kernel_flatp(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order,
                                                       ClsArgs cls, SegArgs seg) {
    static_assert(SEG == 0 || (CLS == 0 && REFILL == 0), "segments: plain PML, no refill");
    static_assert(REFILL == 0 || (STG == 1 && HA < 0), "lane refill: staged reads, window-parallel advance");
    static_assert(AHD == 0 || (STG == 1 && HA < 0), "look-ahead rows: staged reads, window-parallel advance");
    static_assert(AHD == 0 || AHD == 1, "plain rows or look-ahead rows");
    static_assert(PSH == 0 || (STG == 1 && REFILL == 0), "pair-shared gathers: staged kernels without refill");
    static_assert(RING == 0 || (STG == 1 && REFILL == 0), "PMLs out through the LDS ring: staged kernels without refill");
    enum : uint32_t { sFF = 0, sDown = 1, sUp = 2, sDone = 3 };
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();

    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ff_total = 0, scan_total = 0, repo_total = 0, failed = 0, err_total = 0;
    const EndThr ethr = end_thresholds(ix);
    const IdxT r1 = (IdxT)(ix.r - 1), end_row = (IdxT)ix.end_bwt_idx;
    const uint2 row_r1 = load_row<MODE>(ix.rows, r1);       // ReadProcessor::reset_process :69-70: every read starts here
    const uint32_t off0 = row_n<MODE>(row_r1) - 1;

    // ---- the lane's current read
    const bool valid = !REFILL && (SEG == 1 ? (*seg.go != 0u && t < *seg.n_seg) : (t < n_reads && (SEG != 2 || seg.read_fail[t] != 0)));
    uint64_t rid = (valid && order && SEG == 0) ? order[t] : t;
    uint64_t beg = valid ? (SEG == 1 ? seg.seg_in[rid] : offs[rid]) : 0;
    uint32_t len = valid ? (SEG == 1 ? seg.seg_len[rid] : (uint32_t)(offs[rid + 1] - beg)) : 0;   // reads are shorter than 2^32 (checked on the host)
    uint64_t obeg = (SEG == 1 && valid) ? seg.seg_out[rid] : beg;   // where the read's (segment's) PMLs go
    constexpr bool ring = RING != 0;                      // PMLs leave through the ring in LDS (below) instead of the register packer
    uint32_t packed_end = len & (ring ? ~15u : ~7u);      // PMLs of steps >= this are stored one by one

    // The 16 bases of steps kk .. kk+15 of the read (b, l) are the bytes [b + l - kk - 16, b + l - kk) of `bases`, last
    // step first: ONE unconditional 16-byte load -- c0 = steps kk .. kk+7 (step kk in the top byte), c1 = steps kk+8 ..
    // kk+15.  Bytes that belong to steps >= l are never looked at, so a read's last, partial group needs no special
    // case; it merely reaches back into the previous read.  Only a read that starts in the first 16 bytes of the
    // batch can reach back past the buffer: its address is clamped to 0 and fix_pair() shifts the bytes into place
    // WHEN THEY ARE USED.  (Every read-chunk fetch is a 128-byte line from the fabric -- its line is evicted long
    // before the lane comes back -- so 8-base fetches cost 0.125 lines per base, 11 % of all line fetches on c3.)
    // No branch, no select and no zero-fill may touch c0 / c1 at the load: the prefetched groups are consumed 16 steps later,
    // and anything that reads or overwrites the registers of a load in flight makes hipcc park an `s_waitcnt vmcnt(0)`
    // behind it -- i.e. behind the row gather issued just before -- which un-pipelines the iteration (the byte-wise
    // tail variants of the first version did exactly that once per read and lane: every third iteration of a wave).
    // The launcher guarantees >= 16 bytes of bases in the batch.
    // (e = b + l - kk: one past the byte of step kk)
    auto load_pair_at = [&](uint64_t e, uint64_t &c0, uint64_t &c1) {
        uint64_t two[2];
        __builtin_memcpy(two, bases + (e >= 16 ? e - 16 : 0), 16);
        c0 = two[1];
        c1 = two[0];
    };
    auto fix_pair = [&](uint64_t e, uint64_t &c0, uint64_t &c1) {
        if (e < 16) {                                     // the 128-bit value (c0:c1) << 8 * (16 - e); e >= 1
            const uint32_t sh = 8u * (uint32_t)(16 - e);  // 8 .. 120
            if (sh >= 64) { c0 = c1 << (sh - 64); c1 = 0; }
            else { c0 = (c0 << sh) | (c1 >> (64 - sh)); c1 <<= sh; }
        }
    };
    // The 4-row window that holds row nd: aligned, except that the table's last window is pulled back to
    // rows [r-4, r) so that the fetch never leaves the table and needs no special case (r >= 4, checked at
    // launch).  Unpredicated: finished lanes re-read window 0 (a cache hit) instead of branching around the load.
    constexpr uint32_t WN = 4u;                           // rows per window
    const IdxT wb_last = (IdxT)(ix.r - WN);
    auto win_base = [&](IdxT nd) -> IdxT {
        const IdxT wb = nd & ~(IdxT)(WN - 1u);
        return wb < wb_last ? wb : wb_last;
    };
    uint2 ahw[4];                                         // AHD: the look-ahead entries of the window's four rows ...
    uint4 raw[4];                                         // PSH: what this lane loaded for its pair (rows: 0, 1; entries: 2, 3), assembled at the loop's top
    const uint32_t odd_lane = threadIdx.x & 1u;
    auto fetch = [&](IdxT nd, bool act, uint2 (&w)[4]) {
        if (PSH) {
            // byte offset of this lane's window in the table it walks on (AHD: the look-ahead copy, entries 64 bytes further on)
            uint64_t at;
            if (AHD) {
                const IdxT wb = nd & ~(IdxT)3;
                at = wb < wb_last ? (uint64_t)(wb >> 3) * 128u + (uint64_t)((uint32_t)wb & 4u) * 8u : ix.rows2_tail;
            } else {
                at = (uint64_t)win_base(nd) * 8u;
            }
            if (!act) at = 0;
            const uint64_t pat = (uint64_t)pair_swap((uint32_t)at) | ((uint64_t)pair_swap((uint32_t)(at >> 32)) << 32);
            const uint8_t *tab = AHD ? ix.rows2 : ix.rows;
            const uint8_t *pe = tab + (odd_lane ? pat : at) + 16u * odd_lane;    // this lane's half of the even lane's window
            const uint8_t *po = tab + (odd_lane ? at : pat) + 16u * odd_lane;    // ... and of the odd lane's
            __builtin_memcpy(&raw[0], pe, 16);
            if (AHD) __builtin_memcpy(&raw[2], pe + 64u, 16);
            __builtin_memcpy(&raw[1], po, 16);
            if (AHD) __builtin_memcpy(&raw[3], po + 64u, 16);
            return;
        }
        if (AHD) {                                        // line = 8 rows + their 8 entries; the last window has a line of its own
            const IdxT wb = nd & ~(IdxT)3;
            const bool body = wb < wb_last;
            uint64_t at = body ? (uint64_t)(wb >> 3) * 128u + (uint64_t)((uint32_t)wb & 4u) * 8u : ix.rows2_tail;
            if (!act) at = 0;
            load_window<MODE>(ix.rows2 + at, 0, w);
            load_window<MODE>(ix.rows2 + at + 64u, 0, ahw);
        } else {
            load_window<MODE>(ix.rows, (uint64_t)(act ? win_base(nd) : (IdxT)0), w);
        }
    };
    // end of a read: what the reference's exception / output paths do with it
    ClsState cs;
    auto finish_read = [&]() {
        if (SEG == 1) {                                   // K1: the segment's counters and how its walk ended
            SegTot tt;
            tt.ff = ff_total; tt.scan = scan_total; tt.repo = repo_total; tt.flag = failed;
            seg.tot[rid] = tt;
            return;
        }
        if (failed && CLS != 2) {
            for (uint32_t i = 0; i < len; ++i) out[obeg + i] = 0;
        }
        if (CLS) cs.store(cls, rid, failed != 0u);
        if (err) err[rid] = (uint8_t)failed;
        err_total += failed ? 1u : 0u;
    };

    uint32_t st = len > 0 ? sFF : sDone;
    IdxT need = r1;
    uint32_t k = 0;
    uint32_t ml = 0, ff_run = 0;
    uint32_t off = off0;
    uint64_t rb = 0, rb2 = 0, nx0 = 0, nx1 = 0;           // current 8 bases, the 8 after them, and the next 16 (in flight)
    if (st != sDone) {
        load_pair_at(beg + len, rb, rb2);
        fix_pair(beg + len, rb, rb2);
    }
    if (!STG && len > 16) load_pair_at(beg + len - 16, nx0, nx1);
    uint32_t a = s_code[(uint32_t)(rb >> 56) & 0xFFu];    // code of the base of step k (k = 0)
    uint4 pk = make_uint4(0, 0, 0, 0), pk_old = pk;
    if (CLS) cs.init(len, cls.bin_width);
    // ---- reads staged through LDS (STG; ix.stage_lds = bases per lane, a multiple of 16, >= 128: the block's dynamic LDS --
    // the occupancy cap's padding, or what the launcher adds for it): every lane copies the next ix.stage_lds bases of its
    // read into LDS -- 16 bytes per load from the read's end backwards, so the 64 x 150 contiguous bytes of a wavefront of
    // short reads come in as whole cache lines, each fetched ONCE (the lines stay in the CU's L1 over these back-to-back
    // loads) -- and takes every base from there.  The other way to the bases, 16 at a time from global memory (STG = 0,
    // below), re-fetches a read's cache line for every 16 bases: its line is evicted long before the lane comes back
    // (0.0625 lines per base, 6 % of all line fetches of a big batch).  Longer reads ROLL: when any lane of the wavefront
    // is about to leave its staged stretch, every lane stages again from where it stands (stage_from in the loop) -- one
    // extra round trip per >= stage_lds / 2 iterations.  Layout: slot s of lane l at byte (s / 4) * 256 + 4 l + s % 4 --
    // lanes in step read consecutive banks; slot s holds the base of step kbase + s.
    //
    // PMLs out (RING = 1: launches of long reads, launch_pml): behind the staged bases the same dynamic LDS holds a ring of 32
    // PMLs per lane (kOutRingBytes) -- entry e of lane l at byte (e / 8) * 1024 + 16 l + 2 (e % 8): a lane's 8 consecutive PMLs are
    // 16 contiguous bytes, 64 lanes' 16 bytes a conflict-free kilobyte.  An emission is one ds_write_b16; when a lane's k crosses a
    // multiple of 16 the finished group leaves as two adjacent 16-byte stores (two ds_read_b128) -- at most three emissions per
    // iteration, so the ring's other half is always free.  The register packer it stands in for (four v_perm per PML, a saved
    // copy of the first 8 of each 16, three nested divergent branches) is 81 of the loop's ~400 VALU instructions, the ring 35.
    // Where it pays: 100 k x 10 kbp reads -- 6 wavefronts per CU, where a wavefront's own instruction stream is most of an
    // iteration -- 54.7 -> 57.5 Gbases/s; big batches of short reads are bound by the fabric's line rate and lose 1 % (c2 75.1 ->
    // 74.5, the 113 M-row table 53.2 -> 52.6) and the LDS the ring takes (profiles/r04_valu.txt).  The stores must leave where the
    // packer's did, right behind the gather: at the iteration's end c3 drops to 45.3, at the top of the next to 55.1 (c4 -12 %).
    extern __shared__ __align__(16) uint8_t s_stage[];
    uint8_t *const s_ring = s_stage + ix.stage_lds * 64u;
    const uint32_t ring_lane = (threadIdx.x & 63u) * 16u;
    auto ring_put = [&](uint32_t kk, uint32_t val) {
        *reinterpret_cast<uint16_t *>(s_ring + ((kk >> 3) & 3u) * 1024u + ring_lane + (kk & 7u) * 2u) = (uint16_t)val;
    };
    auto ring_flush = [&](uint32_t k0) {                  // the group of 16 that step k0 lies in: complete, and all of it below packed_end
        const uint32_t g = (k0 >> 4) & 1u;
        const uint4 lo = *reinterpret_cast<const uint4 *>(s_ring + (2u * g) * 1024u + ring_lane);
        const uint4 hi = *reinterpret_cast<const uint4 *>(s_ring + (2u * g + 1u) * 1024u + ring_lane);
        uint16_t *dst = out + obeg + (k0 & ~15u);
        __builtin_memcpy(dst, &lo, 16);                   // unaligned 16-byte stores
        __builtin_memcpy(dst + 8, &hi, 16);
    };
    uint32_t kbase = 0;
    const uint32_t stage_cap = ix.stage_lds;
    // (the loads of kStageUnroll groups leave together -- unconditional, lanes without the group re-read the batch's first
    // bytes -- before the first of them is waited for.  Two at a time: c2 74.6 -> 75.0, c3 54.8 -> 54.9 Gbases/s; four or
    // eight in flight cost c3 11 % (48.9: profiles/r04_stage_unroll.txt) although the loop then makes a quarter of the trips)
#ifndef MOVI_STAGE_UNROLL
#define MOVI_STAGE_UNROLL 2
#endif
    constexpr uint32_t kStageUnroll = MOVI_STAGE_UNROLL;
    auto stage_from = [&](uint32_t k0, bool on) {         // every lane of the wavefront makes the call; lanes with `on` stage
        uint32_t *S = reinterpret_cast<uint32_t *>(s_stage);
        const uint32_t sl = threadIdx.x & 63u;
        const uint32_t left = (on && len > k0) ? len - k0 : 0u;
        const uint32_t cnt = left < stage_cap ? left : stage_cap;
        for (uint32_t g = 0; wave_any(16u * g < cnt); g += kStageUnroll) {
            uint64_t c0[kStageUnroll], c1[kStageUnroll];
#pragma unroll
            for (uint32_t u = 0; u < kStageUnroll; ++u) {
                const uint64_t e = 16u * (g + u) < cnt ? beg + len - k0 - 16u * (g + u) : 16u;
                load_pair_at(e, c0[u], c1[u]);
            }
#pragma unroll
            for (uint32_t u = 0; u < kStageUnroll; ++u) {
                if (16u * (g + u) < cnt) {
                    const uint64_t e = beg + len - k0 - 16u * (g + u);
                    fix_pair(e, c0[u], c1[u]);
                    const uint64_t r0 = __builtin_bswap64(c0[u]), r1 = __builtin_bswap64(c1[u]);   // step k0 + 16 (g + u) in the low byte
                    S[(4u * (g + u) + 0u) * 64u + sl] = (uint32_t)r0;
                    S[(4u * (g + u) + 1u) * 64u + sl] = (uint32_t)(r0 >> 32);
                    S[(4u * (g + u) + 2u) * 64u + sl] = (uint32_t)r1;
                    S[(4u * (g + u) + 3u) * 64u + sl] = (uint32_t)(r1 >> 32);
                }
            }
        }
        if (on) kbase = k0;
    };
    auto staged_code = [&](uint32_t slot) -> uint32_t {   // code of the base in `slot` (clamped into the staged stretch)
        const uint32_t q = slot < stage_cap ? slot : stage_cap - 1u;
        return s_code[s_stage[(q >> 2) * 256u + (threadIdx.x & 63u) * 4u + (q & 3u)]];
    };
    if (STG) stage_from(0u, st != sDone);
    // ---- top of the walk (DevIndex::kmer): the first K bases of the read (segment) by ONE table lookup.  Reads with an
    // illegal base among them, reads of K bases or fewer and K-mers whose walk throws take the ordinary walk.
    // cand: lanes whose K-mer `kidx` is to be looked up (k == 0 there); returns the lanes that took the entry.
    auto top_of_walk = [&](uint32_t cand, uint32_t kidx) -> uint32_t {
        const uint32_t K = ix.kmer_k;
        uint4 e4 = make_uint4(0, 0, 0, 0);
        if (cand) e4 = ix.kmer[kidx];
        const uint32_t use = cand & (e4.y >> 31);
        if (use) {
            const uint32_t mask = (e4.y >> 16) & 0xFFFu;
            uint16_t *O = out + obeg;
            uint32_t run = 0;
            for (uint32_t i = 0; i < K; ++i) {            // the K PMLs, through the same packing as the loop's emissions
                run = ((mask >> i) & 1u) ? run + 1u : 0u;
                if (CLS) cs.add(run, k, len, cls.bin_width, cls.thr);
                if (CLS == 2) {
                } else if (k >= packed_end) {
                    O[k] = (uint16_t)run;
                } else if (STG && ring) {
                    ring_put(k, run);                     // (K <= 12: no group of 16 is completed here)
                } else {
                    pk.x = (pk.x >> 16) | (pk.y << 16);
                    pk.y = (pk.y >> 16) | (pk.z << 16);
                    pk.z = (pk.z >> 16) | (pk.w << 16);
                    pk.w = (pk.w >> 16) | (run << 16);
                    if ((k & 15) == 7) {
                        if (k + 8 < packed_end) pk_old = pk;
                        else __builtin_memcpy(O + (k - 7), &pk, 16);
                    }
                }
                k += 1;
            }
            ml = run;
            need = (IdxT)((uint64_t)e4.x | ((uint64_t)(e4.y & 15u) << 32));
            off = (e4.y >> 4) & 0xFFFu;
            ff_total += e4.z;
            scan_total += e4.w;
            repo_total += K - (uint32_t)__popc(mask);
        }
        return use;
    };
    if (!REFILL && ix.kmer_k != 0u) {                     // wave-uniform
        const uint32_t K = ix.kmer_k;
        uint32_t kidx = 0, bad = 0;
        for (uint32_t i = 0; i < K; ++i) {
            const uint64_t src = i < 8u ? rb : rb2;
            const uint32_t cc = (uint32_t)s_code[(uint32_t)(src >> (8u * (7u - (i & 7u)))) & 0xFFu] - (uint32_t)SEP;
            bad |= (uint32_t)(cc > 3u);
            kidx |= (cc & 3u) << (2u * i);
        }
        if (top_of_walk((uint32_t)(st != sDone) & (uint32_t)(len > K) & (bad ^ 1u), kidx)) {
            if (K >= 8u) rb = rb2;
            a = s_code[(uint32_t)(rb >> (8u * (7u - (K & 7u)))) & 0xFFu];
        }
    }
    // AHD: the code of the base after the current one (beyond the read's end: never looked at)
    uint32_t a1 = 0xFFu;
    if (AHD) a1 = staged_code(k + 1);
    uint2 w[4];
    fetch(need, st != sDone, w);

    // ---- lane refill (REFILL): a POOL of upcoming reads per wavefront, fed in chunks of 16 consecutive reads from one global
    // ticket counter (DevStats::ticket, zeroed with the counters): pool slot `lane` holds a read's number, where its bases
    // start and its length; slots are consumed in ring order by whichever lanes are idle when the wavefront switches.  A chunk
    // takes two switches to arrive -- the ticket is drawn (one atomic) at one switch, the chunk's offsets are requested at the
    // next and merged into the pool at the one after -- so nothing about the pool ever waits on memory by itself.
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t cur_valid = 0;
    uint64_t P_beg = 0, N_beg = 0;                        // pool slot / chunk in flight: byte offset of the read's bases ...
    uint32_t P_len = 0, P_rid = 0, N_end = 0;             // ... its length and number (in flight: low half of the next offset)
    uint32_t p_head = 0, p_count = 0;                     // ring of 64 slots (wave-uniform)
    uint32_t ld_on = 0, tk_on = 0, no_more = 0;           // a chunk's offsets in flight / a ticket in flight / the counter ran past n_reads
    uint64_t ld_base = 0;
    unsigned long long T = 0;                             // lane 0: the ticket drawn
    auto draw_ticket = [&]() {
        if (lane == 0u) T = atomicAdd(&stats->ticket, 16ull);
        tk_on = 1;
    };
    if (REFILL) {                                         // the first four chunks at once
        if (lane == 0u) T = atomicAdd(&stats->ticket, 64ull);
        const uint64_t base = __builtin_amdgcn_readfirstlane((uint32_t)T) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(T >> 32)) << 32);
        if (base < n_reads) {
            const uint64_t left = n_reads - base;
            p_count = left < 64u ? (uint32_t)left : 64u;
            const uint64_t rr = base + lane < n_reads ? base + lane : n_reads - 1;
            P_beg = offs[rr];
            P_len = (uint32_t)offs[rr + 1] - (uint32_t)P_beg;
            P_rid = (uint32_t)rr;
            if (p_count == 64u) draw_ticket(); else no_more = 1;
        } else {
            no_more = 1;
        }
    }

    uint32_t lane_steps = 0, wave_steps = 0;
    while (wave_any(st != sDone) || (REFILL && (p_count != 0u || ld_on != 0u || tk_on != 0u))) {
        if (REFILL) {
            // ---- idle lanes take the pool's next reads -- in batches, when ix.refill_batch lanes (or every lane that still has
            // work) wait for one: the switch stages the new reads' bases and looks their first K bases up in the top-of-walk
            // table, memory round trips during which the whole wavefront stands still.
            const uint32_t idle = (uint32_t)(st == sDone);
            const uint64_t idm = __ballot(idle != 0u);
            const uint32_t n_idle = (uint32_t)__popcll(idm);
            if (idm == ~0ull || (n_idle >= ix.refill_batch && (p_count != 0u || ld_on != 0u || tk_on != 0u))) {
                // (a) the chunk whose offsets were requested at the last switch joins the pool
                if (ld_on) {
                    const uint64_t left = n_reads - ld_base;
                    const uint32_t cnt = left < 16u ? (uint32_t)left : 16u;
                    const uint32_t tail = (p_head + p_count) & 63u;           // a multiple of 16: every chunk but the last is full
                    if ((lane & 48u) == tail) {
                        P_beg = N_beg;
                        P_len = N_end - (uint32_t)N_beg;
                        P_rid = (uint32_t)ld_base + (lane & 15u);
                    }
                    p_count += cnt;
                    ld_on = 0;
                    if (cnt < 16u) no_more = 1;
                }
                // (b) the ticket drawn at the last switch: its chunk's offsets are requested now
                if (tk_on) {
                    const uint64_t base = __builtin_amdgcn_readfirstlane((uint32_t)T) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(T >> 32)) << 32);
                    tk_on = 0;
                    if (base < n_reads) {
                        ld_base = base;
                        const uint64_t rr = base + (lane & 15u) < n_reads ? base + (lane & 15u) : n_reads - 1;
                        N_beg = offs[rr];
                        N_end = *reinterpret_cast<const uint32_t *>(offs + rr + 1);
                        ld_on = 1;
                    } else {
                        no_more = 1;
                    }
                }
                // (c) room for another chunk: draw its ticket
                if (!no_more && !tk_on && p_count + (ld_on ? 16u : 0u) <= 16u) draw_ticket();
                // (d) the switch: the first min(idle lanes, pool) idle lanes take the pool's next reads, in ring order
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idm, 0u));
                const uint32_t take = n_idle < p_count ? n_idle : p_count;
                const uint32_t src = (p_head + rank) & 63u;
                const uint32_t g_lo = __shfl((uint32_t)P_beg, (int)src, 64), g_hi = __shfl((uint32_t)(P_beg >> 32), (int)src, 64),
                               g_len = __shfl(P_len, (int)src, 64), g_rid = __shfl(P_rid, (int)src, 64);
                const uint32_t sw = idle & (uint32_t)(rank < take);
                p_head = (p_head + take) & 63u;
                p_count -= take;
                if (idle && cur_valid) {                  // the read that ended: error byte, bins, zero-fill on failure
                    finish_read();
                    cur_valid = 0;
                }
                if (sw) {
                    rid = g_rid; beg = (uint64_t)g_lo | ((uint64_t)g_hi << 32); obeg = beg; len = g_len; packed_end = len & (ring ? ~15u : ~7u);
                    k = 0; ml = 0; ff_run = 0; off = off0; need = r1; failed = 0; cur_valid = 1;
                    if (CLS) { cs = ClsState(); cs.init(len, cls.bin_width); }
                    st = len > 0 ? sFF : sDone;
                }
                const uint32_t fresh = sw & (uint32_t)(st != sDone);
                stage_from(0u, fresh != 0u);
                if (ix.kmer_k != 0u) {
                    const uint32_t K = ix.kmer_k;
                    uint32_t kidx = 0, bad = 0;
                    for (uint32_t i = 0; i < K; ++i) {
                        const uint32_t cc = staged_code(i) - (uint32_t)SEP;
                        bad |= (uint32_t)(cc > 3u);
                        kidx |= (cc & 3u) << (2u * i);
                    }
                    top_of_walk(fresh & (uint32_t)(len > K) & (bad ^ 1u), kidx);
                }
                if (fresh) {
                    a = staged_code(k);
                    if (AHD) a1 = staged_code(k + 1);
                    fetch(need, true, w);
                }
            }
        }
        const bool act = st < sDone;
        lane_steps += (uint32_t)act;
        wave_steps += 1;
#if defined(MOVI_PAD_PRE) && MOVI_PAD_PRE > 0
        {   // experiment (profiles/r04_valu.txt): MOVI_PAD_PRE dependent VALU instructions between the window's arrival and the next gather
            uint32_t pad = wave_steps;
#pragma unroll
            for (int i = 0; i < MOVI_PAD_PRE; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad));
        }
#endif
        if (PSH) {                                        // the halves the pair loaded for each other change hands
            pair_assemble(odd_lane, raw[0], raw[1], w);
            if (AHD) pair_assemble(odd_lane, raw[2], raw[3], ahw);
        }
        const IdxT wbase = win_base(need);
        // cheap hop: a fast-forward or scan step that only moves on (everything that resolves a base,
        // starts a scan, ends one or fails is left to the full step below)
        auto hop = [&]() {
            const uint32_t q = (uint32_t)(need - wbase);
            const uint32_t inwin = (uint32_t)(q < WN) & (uint32_t)(st < sDone);
            const uint2 hr = win_sel(w, q);
            const uint32_t hn = row_n<MODE>(hr), hc = row_c<MODE>(hr);
            const uint32_t ffh = inwin & (uint32_t)(st == sFF) & (uint32_t)(need < r1) & (uint32_t)(off >= hn) &
                                 (uint32_t)(ff_run + 1 < 65535u);
            const uint32_t nomatch = hc != a;
            const uint32_t dnh = inwin & (uint32_t)(st == sDown) & nomatch & (uint32_t)(need < r1);
            const uint32_t uph = inwin & (uint32_t)(st == sUp) & nomatch & (uint32_t)(need != 0);
            off = ffh ? off - hn : off;
            ff_run += ffh;
            scan_total += dnh | uph;
            need = need + (IdxT)(ffh + dnh) - (IdxT)uph;
        };
        // HA < 0 ("window-parallel", variant 14): everything the hops could do inside this window, in closed form instead
        // of one dependent select-compare-update round per hop.  A fast-forward passes row i iff off >= the running sum of
        // the lengths up to and including i (monotone, so the number of rows passed is a sum of four compares); a scan
        // passes the leading run of non-matching rows from its position (a 4-bit mask and a count-trailing / leading-ones).
        // Same state afterwards as four hop() calls -- identical answers and counts -- at a third of the dependency depth.
        // bit i of nm = row i of the window does not hold the base of step k
        const uint32_t nm = (uint32_t)(row_c<MODE>(w[0]) != a) | ((uint32_t)(row_c<MODE>(w[1]) != a) << 1) |
                            ((uint32_t)(row_c<MODE>(w[2]) != a) << 2) | ((uint32_t)(row_c<MODE>(w[3]) != a) << 3);
        const uint32_t last_win = (uint32_t)(wbase + 3 == r1);             // the table ends inside (at the end of) this window
        const uint32_t first_win = (uint32_t)(wbase == 0);
        auto window_advance = [&]() {
            const uint32_t q0 = (uint32_t)(need - wbase);
            const uint32_t inwin = (uint32_t)(q0 < 4u) & (uint32_t)(st < sDone);
            const uint32_t n0 = row_n<MODE>(w[0]), n1 = row_n<MODE>(w[1]), n2 = row_n<MODE>(w[2]), n3 = row_n<MODE>(w[3]);
            // ---- fast-forward: rows q0 .. 3 (need < r1 can only fail at index 3 of the last window)
            const uint32_t m0 = q0 == 0u, m1 = q0 <= 1u, m2 = q0 <= 2u;   // row i takes part (i >= q0); row 3 always does
            const uint32_t t1 = m0 ? n0 : 0u, t2 = t1 + (m1 ? n1 : 0u), t3 = t2 + (m2 ? n2 : 0u), t4 = t3 + n3;
            const uint32_t isff = inwin & (uint32_t)(st == sFF);
            const uint32_t p0 = isff & m0 & (uint32_t)(off >= t1), p1 = isff & m1 & (uint32_t)(off >= t2),
                           p2 = isff & m2 & (uint32_t)(off >= t3), p3 = isff & (uint32_t)(off >= t4) & (last_win ^ 1u);
            const uint32_t cf = p0 + p1 + p2 + p3;
            off -= (p3 ? t4 : (p2 ? t3 : (p1 ? t2 : (p0 ? t1 : 0u))));
            ff_run += cf;
            // ---- scans
            // down: leading run of 1s from bit q0 upwards; row r-1 is never passed (need < r1)
            const uint32_t dmask = (nm & (last_win ? 7u : 15u)) >> (q0 & 3u);
            const uint32_t cd = (inwin & (uint32_t)(st == sDown)) ? (uint32_t)__builtin_ctz(~dmask | 16u) : 0u;
            // up: leading run of 1s from bit q0 downwards; row 0 is never passed (need != 0)
            const uint32_t umask = ((nm & (first_win ? 14u : 15u)) << (3u - (q0 & 3u))) & 15u;
            const uint32_t cu = (inwin & (uint32_t)(st == sUp)) ? (uint32_t)__builtin_clz(((~umask) & 15u) << 28 | 0x08000000u) : 0u;
            scan_total += cd + cu;
            need = need + (IdxT)(cf + cd) - (IdxT)cu;
        };
        if (HA >= 0) {
#pragma unroll
            for (int h = 0; h < (HA >= 0 ? HA : 0); ++h) hop();
        } else if (wave_any(st == sFF && ff_run >= 65520u)) {
            for (int h = 0; h < 4; ++h) hop();                       // near the reference's fast-forward limit: step by step
        } else {
            window_advance();
        }
        const uint32_t qn = (uint32_t)(need - wbase);
        const uint32_t inwin = (uint32_t)(qn < WN) & (uint32_t)act;
        const uint2 row = win_sel(w, qn);
        const uint32_t n = row_n<MODE>(row), c = row_c<MODE>(row);
        const uint32_t isFF = (uint32_t)(st == sFF) & inwin, isDown = (uint32_t)(st == sDown) & inwin,
                       isUp = (uint32_t)(st == sUp) & inwin;
        // fast_forward, move_structure.cpp:524-545
        const uint32_t ffm = isFF & (uint32_t)(need < r1) & (uint32_t)(off >= n);
        const uint32_t ff_over = ffm & (uint32_t)(ff_run + 1 >= 65535u);  // :72-75
        const uint32_t resolved = isFF & (ffm ^ 1u);
        // the base of step k against the row (read_processor.cpp:188-238)
        const uint32_t illegal = a == 0xFFu, match = c == a;
        const uint32_t mism = resolved & (illegal ^ 1u) & (match ^ 1u);
        // reposition_thresholds, src/move_structure_query.cpp:513-601
        // (SEP is a template parameter here: the separator branch and its selects sit on the critical path
        // between the window's arrival and the next gather, and cost 4 % on c3 as a run-time flag)
        const uint32_t kk = thr_slot(SEP, a, c);                          // alphamap_3[c][a]
        uint32_t thr = row_thr<MODE>(row, kk > 2u ? 2u : kk) ? n : 0u;
        if (SEP) {                                                        // a row of the separator: side table
            if (mism & (uint32_t)(c == 0u) & (uint32_t)(need != end_row)) thr = separator_threshold(ix, (uint64_t)need, a);
        }
        const uint32_t down = (uint32_t)(off >= ((need == end_row) ? end_threshold(SEP, ethr, a) : thr));
        const uint32_t at_last = need >= r1, at_first = need == 0;
        const uint32_t repo_edge = mism & (down ? at_last : at_first);
        // reposition_down :211-232 / reposition_up :188-209.  A run of the base among the window's OTHER rows is found in
        // this very iteration (the nearest one in the scan's direction: what the row-by-row scan stops at) -- a reposition
        // whose target shares the window costs no round trip of its own (tools/iter_model.c: half of all repositions; lane
        // iterations per base -5 % on 150 bp reads with 1 % substitutions, -15 % on 10 kbp reads with 8 %).  Anything
        // further away is scanned for one window per iteration, as before.
        const uint32_t has = (nm ^ 15u) & (down ? (14u << (qn & 3u)) & 15u : (1u << (qn & 3u)) - 1u);   // rows that hold the base, beyond row qn
        const uint32_t found = mism & (uint32_t)(has != 0u) & ix.inwin;
        const uint32_t qf = found ? (down ? (uint32_t)__builtin_ctz(has | 16u) : 31u - (uint32_t)__builtin_clz(has | 1u)) : qn;
        const uint32_t far = mism & (found ^ 1u);                          // the scan leaves the window
        const uint2 rowf = win_sel(w, qf);                                 // the row the base is resolved at, if it is resolved now
        const uint32_t nf = row_n<MODE>(rowf), rofff = row_off<MODE>(rowf);
        const IdxT needf = (IdxT)(wbase + qf);
        const uint32_t scanning = isDown | isUp;
        const uint32_t hit = scanning & match;
        const uint32_t landed = hit | found;                               // a scan ended at this row: offset 0 / n - 1 (read_processor.cpp:223)
        const uint32_t landed_down = hit ? isDown : down;
        const uint32_t scan_edge = scanning & (hit ^ 1u) & (isDown ? at_last : at_first);
        const uint32_t emit = (resolved & (illegal | match)) | landed;
        // LF_move of the emitted base, move_structure.cpp:59-67 (emit and the error cases are exclusive)
        const uint32_t lf = emit & (uint32_t)(k + 1 != len);
        uint64_t j = 0;
        if (MODE == 6 || lf) j = row_id<MODE>(rowf, needf, ix);
        const uint32_t lf_bad = lf & (uint32_t)(j >= ix.r);
        const uint32_t step_fwd = ffm | (far & down) | (scanning & (hit ^ 1u) & isDown);
        const uint32_t step_back = (far & (down ^ 1u)) | (scanning & (hit ^ 1u) & isUp);
        IdxT need_next = lf ? (IdxT)j : (IdxT)(need + step_fwd - step_back);
        uint32_t st_next = (emit & (lf ^ 1u)) ? sDone : (lf ? sFF : (far ? (down ? sDown : sUp) : st));
        // AHD: the base after this one, resolved at the LF target from the look-ahead entry (read_processor.cpp:188-238 with
        // match and no fast-forward: ml + 1, then LF_move again) -- the target row itself is never fetched
        uint32_t dbl = 0, lf2 = 0, off1 = 0;
        IdxT j2 = 0;
        if (AHD) {
            const uint2 ah = win_sel(ahw, qf);            // the entry of the row the base was resolved at
            const uint32_t n1 = ah.y & 0x7FFu, c1 = (ah.y >> 22) & 7u;
            const uint32_t off_e = (landed ? (landed_down ? 0u : nf - 1u) : off) + rofff;
            dbl = lf & (ah.y >> 31) & (uint32_t)(a1 == c1) & (uint32_t)(off_e < n1);
            lf2 = dbl & (uint32_t)(k + 2 != len);
            off1 = (ah.y >> 11) & 0x7FFu;
            j2 = (IdxT)((uint64_t)ah.x | ((uint64_t)((ah.y >> 25) & 15u) << 32));
            need_next = dbl ? (lf2 ? j2 : need) : need_next;
            st_next = dbl ? (lf2 ? sFF : sDone) : st_next;
        }
        // The reference's throws: practically never, so which one it was is sorted out off the common path (as one
        // select ladder over need_next / st_next it cost ~45 instructions between a window's arrival and the next
        // gather's issue in every iteration).  An error freezes the lane where it is: no out-of-table window is fetched.
        uint32_t errc = kErrNone;
        if (wave_any((ff_over | repo_edge | scan_edge | lf_bad) != 0u)) {
            errc = ff_over ? kErrFastForward
                           : (repo_edge ? (down ? kErrNoRunBelow : kErrNoRunAbove)
                              : (scan_edge ? (isDown ? kErrNoRunBelow : kErrNoRunAbove)
                                 : (lf_bad ? kErrIdRange : kErrNone)));
            if (errc) { need_next = need; st_next = sDone; }
        }
        // ---- the next gather leaves now; everything below runs under its latency
        // (`row` is not touched below, so the new window can land in the old one's registers)
        fetch(need_next, st_next != sDone, w);
#if defined(MOVI_PAD_POST) && MOVI_PAD_POST > 0
        {   // ... and MOVI_PAD_POST of them under the gather's latency
            uint32_t pad = wave_steps;
#pragma unroll
            for (int i = 0; i < MOVI_PAD_POST; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad));
        }
#endif
        // ---- bookkeeping, all selects
        uint32_t want_nx = 0;                             // this lane asks for the 16 bases that end at byte nx_e
        uint64_t nx_e = 0;
        ml = resolved ? (match ? ml + 1 : 0u) : ml;
        ff_total += resolved ? ff_run : 0u;
        ff_run = lf ? 0u : ff_run + ffm;
        repo_total += mism;
        scan_total += scanning + (found ? (down ? qf - qn : qn - qf) : 0u);
        off = ffm ? off - n : (landed ? (landed_down ? 0u : nf - 1) : off);   // read_processor.cpp:223
        const uint32_t off_pre = off;                                     // (before the LF to the next base: what K1 records)
        off += lf ? rofff : 0u;
        if (emit) {
            uint16_t *O = out + obeg;
            // MoveQuery::add_ml for the base of step k (u16 clamp), through the bins and the 16-byte packer: 16 PMLs leave
            // together as two adjacent 16-byte stores; an odd group of 8 before the tail on its own
            auto emit_pml = [&](uint32_t mlv) {
                const uint32_t val = mlv > 65535u ? 65535u : mlv;
                if (CLS) cs.add(val, k, len, cls.bin_width, cls.thr);
                if (CLS == 2) {
                    // verdict bins only
                } else if (k >= packed_end) {
                    O[k] = (uint16_t)val;
                } else if (STG && ring) {
                    ring_put(k, val);
                } else {
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
                k += 1;
            };
            // K1's records (SEG == 1): the state a one-base walk has after the base of step k -- at row `at`, before its LF
            auto seg_record = [&](uint64_t at, uint32_t off_at) {
                if ((k & 31u) == 31u) {
                    SegCkpt ck;
                    ck.idx = at; ck.off = off_at; ck.ml = ml;
                    ck.ff = ff_total; ck.scan = scan_total; ck.repo = repo_total; ck.pad_ = 0;
                    seg.ckpt[(obeg + k) >> 5] = ck;
                }
                if (k + 1 == len) {
                    SegFin fn;
                    fn.idx = at; fn.off = off_at; fn.ml = ml;
                    seg.fin[rid] = fn;
                }
            };
            const uint32_t k_in = k;
            if (SEG == 1) seg_record((uint64_t)needf, off_pre);
            emit_pml(ml);
            if (AHD && dbl) {                             // the second base of a multi-base step: matched, no fast-forward
                ml += 1;
                if (SEG == 1) seg_record(j, off);
                off += lf2 ? off1 : 0u;
                emit_pml(ml);
            }
            if (STG && CLS != 2 && ring && ((k ^ k_in) & 16u) != 0u) ring_flush(k_in);   // a group of 16 PMLs is complete
            if (STG) {
                // (the next base's code: after the state update below, where a lane about to leave its staged stretch is seen)
            } else if (lf) {
                {
                    if ((k & 15) == 8) {
                        rb = rb2;
                    } else if ((k & 15) == 0) {
                        rb = nx0;
                        rb2 = nx1;
                        fix_pair(beg + len - k, rb, rb2);
                        if (k + 16 < len) { want_nx = 1; nx_e = beg + len - k - 16; }
                    }
                    a = s_code[(uint32_t)(rb >> (8 * (7 - (k & 7)))) & 0xFFu];
                }
            }
        }
        if (errc) failed = errc;
        need = need_next;
        st = st_next;
        if (STG) {
            // a lane whose next bases lie beyond its staged stretch: the whole wavefront stages again, each lane from its own step
            const uint32_t ahead_of = k - kbase;          // < 2^31: k >= kbase always
            const uint32_t out_of = (uint32_t)(st != sDone) &
                                    ((uint32_t)(ahead_of >= stage_cap) | ((uint32_t)(ahead_of + 1u >= stage_cap) & (uint32_t)(k + 1 < len)));
            if (wave_any(out_of != 0u)) stage_from(k, st != sDone);
            a = staged_code(k - kbase);
            if (AHD) a1 = staged_code(k + 1 - kbase);
        }
        // ONE load site per prefetch register set and iteration, behind every read of those registers: a second site (or
        // a temporary that the register allocator parks in them where they are dead) costs an `s_waitcnt` on a load
        // in flight, i.e. on the row gather issued above
        if (want_nx) load_pair_at(nx_e, nx0, nx1);
    }
    if (REFILL ? cur_valid != 0u : valid) finish_read();
    if (SEG != 1) {
        const uint32_t ffw = wave_sum(ff_total), scw = wave_sum(scan_total), rpw = wave_sum(repo_total),
                       erw = wave_sum(err_total);
        if ((threadIdx.x & 63) == 0 && stats) {
            if (ffw) atomicAdd(&stats->fast_forwards, (unsigned long long)ffw);
            if (scw) atomicAdd(&stats->scans, (unsigned long long)scw);
            if (rpw) atomicAdd(&stats->repositions, (unsigned long long)rpw);
            if (erw) atomicAdd(&stats->errors, (unsigned long long)erw);
        }
    }
    const uint32_t lsw = wave_sum(lane_steps);
    if ((threadIdx.x & 63) == 0 && stats) {
        atomicAdd(&stats->lane_steps, (unsigned long long)lsw);
        atomicAdd(&stats->wave_steps, (unsigned long long)wave_steps);
    }
}

// ------------------------------------------------------------- segment-parallel long reads (movi_kernels.hpp)

// One base of the plain base-synchronous automaton (pml_kernel<MODE, 0>) for the lanes with `live`: the LF from the
// base before (unless this is the walk's first base), then match / illegal / reposition_thresholds + scan against base
// code `a`.  Wave-uniform loops inside: every lane of the wavefront must make the call.  Returns a kErr* code.
template <int MODE>
__device__ __forceinline__ uint32_t walk_base(const DevIndex &ix, const EndThr &ethr, bool live, bool lf, uint32_t a, uint64_t &idx,
                                              uint32_t &off, uint2 &row, uint32_t &ml, uint32_t &ff_total, uint32_t &scan_total,
                                              uint32_t &repo_total) {
    uint32_t failed = lf_step<MODE>(ix, live && lf, idx, off, row, ff_total);
    if (failed) live = false;
    const uint32_t rc = row_c<MODE>(row);
    uint32_t dir = 0;
    if (live) {
        if (a == 0xFFu) {
            ml = 0;
        } else if (rc == a) {
            ml += 1;
        } else {                                          // reposition_thresholds, as in pml_kernel
            repo_total += 1;
            ml = 0;
            uint32_t down;
            if (idx == ix.end_bwt_idx) {
                down = (off >= end_threshold(ix.sep, ethr, a)) ? 1u : 0u;
            } else if (ix.sep && rc == 0u) {
                down = (off >= separator_threshold(ix, idx, a)) ? 1u : 0u;
            } else {
                const uint32_t kk = thr_slot(ix.sep, a, rc);
                const uint32_t thr = row_thr<MODE>(row, kk > 2u ? 2u : kk) ? row_n<MODE>(row) : 0u;
                down = (off >= thr) ? 1u : 0u;
            }
            dir = down ? 1u : 2u;
            if (down && idx == ix.r - 1) { failed = kErrNoRunBelow; dir = 0; }
            if (!down && idx == 0) { failed = kErrNoRunAbove; dir = 0; }
        }
    }
    uint32_t scanning = dir;
    while (wave_any(scanning != 0u)) {
        if (scanning) {
            uint64_t jj = (scanning == 1u) ? idx + 1 : idx - 1;
            if (scanning == 1u) { if (jj >= ix.r) jj = ix.r - 1; }
            else if (jj > idx) jj = 0;
            const uint2 w = load_row<MODE>(ix.rows, jj);
            scan_total += 1;
            idx = (scanning == 1u) ? idx + 1 : idx - 1;
            row = w;
            const uint32_t c = row_c<MODE>(row);
            if (c == a) {
                scanning = 0;
            } else if (scanning == 1u ? (idx >= ix.r - 1) : (idx == 0)) {
                failed = scanning == 1u ? kErrNoRunBelow : kErrNoRunAbove;
                scanning = 0;
            }
        }
    }
    if (failed == 0u && dir == 1u) off = 0;
    if (failed == 0u && dir == 2u) off = row_n<MODE>(row) - 1;
    return failed;
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
template <int MODE>
__global__ __launch_bounds__(256) void ahead_rows_kernel(DevIndex ix, uint8_t *__restrict__ out, uint64_t tail, unsigned long long *tally) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < ix.r;
    const uint2 row = in ? load_row<MODE>(ix.rows, i) : make_uint2(0u, 0u);
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

hipError_t build_ahead_rows(int kmode, const DevIndex &ix, uint8_t *d_rows2, uint64_t *tail, hipStream_t stream, unsigned long long *d_tally) {
    if (!d_rows2 || !tail || ix.r < 8 || (ix.r >> 36) != 0 || kmode != 6) return hipErrorInvalidValue;
    *tail = ((ix.r + 7) / 8) * 128;
    hipError_t e = hipMemsetAsync(d_rows2, 0, ahead_rows_bytes(ix.r), stream);
    if (e != hipSuccess) return e;
    const uint64_t blocks = (ix.r + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ahead_rows_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, stream, ix, d_rows2, *tail, d_tally);
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
template <int MODE, int PASS>
__global__ __launch_bounds__(256) void seg_stitch_kernel(DevIndex ix, const uint8_t *__restrict__ bases, SegArgs seg,
                                                        const uint32_t *__restrict__ seg_j, const uint32_t *__restrict__ seg_rem,
                                                        uint32_t max_over, const uint8_t *__restrict__ on_chain,
                                                        uint16_t *__restrict__ out, SegJoin *__restrict__ join) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = ix.code_of[i];
    __syncthreads();
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = *seg.go != 0u && s < *seg.n_seg && seg_j[s] != 0 && (PASS == 0 || on_chain[s] == 2);
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
    uint64_t idx = 0;
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
        if (live) a = s_code[*(R - 1 - (int64_t)k)];
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
    auto lds_for = [&](uint64_t lanes) -> size_t {
        int wpc = cfg.waves_per_cu;
        if (wpc < 0) wpc = 0;
        if (cfg.waves_per_cu == 0 && big_batch_cap && lanes > (uint64_t)cfg.num_cus * 64u * 18u)
            wpc = (ix.rows2 != nullptr && cfg.stage_reads != 0) ? kCapWavesAhead : kCapWaves;
        if (wpc > 0 && wpc < 32) return ((163840u / (unsigned)wpc) & ~1023u) - 1024u;
        return 0;
    };
    const ClsArgs cls;
    const uint32_t *d_order = nullptr;
    // (segments and re-walked reads stage their bases through LDS and walk on the look-ahead rows like any other launch:
    // launch_pml's policy -- the cap's padding, or what the launch's wavefronts per CU leave of the CU's LDS)
    DevIndex ixl = ix;
    ixl.inwin = cfg.inwin ? 1u : 0u;
    // (pair-shared gathers on tables beyond the TLBs' reach: launch_pml's rule)
    const bool seg_pair = (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && ix.r * (ix.rows2 != nullptr ? 16ull : 8ull) >= kPairLoadBytes));
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
#define MOVI_LAUNCH_SEG(SEGV, LANES, ...)                                                                             \
    do {                                                                                                              \
        if (dyn_lds > 65536) {                                                                                        \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&__VA_ARGS__),                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);                        \
            if (e != hipSuccess) return e;                                                                            \
        }                                                                                                             \
        hipLaunchKernelGGL((__VA_ARGS__), dim3((unsigned)(((LANES) + bt - 1) / bt)), dim3(bt), dyn_lds, stream, ixl,  \
                           d_bases, d_offsets, (uint64_t)(LANES), d_out, d_err, d_stats, d_order, cls, seg);          \
    } while (0)
#define MOVI_LAUNCH_SEG_S(SEGV, LANES, T, S)                                                                          \
    do {                                                                                                              \
        if (seg_ring && seg_pair && ix.rows2 != nullptr) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 1, 1, 1>); \
        else if (seg_ring && seg_pair) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 0, 1, 1>);   \
        else if (seg_ring && ix.rows2 != nullptr) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 1, 0, 1>); \
        else if (seg_ring) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 0, 0, 1>);               \
        else if (seg_pair && ixl.stage_lds != 0u && ix.rows2 != nullptr) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 1, 1>); \
        else if (seg_pair && ixl.stage_lds != 0u) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 0, 1>); \
        else if (ixl.stage_lds != 0u && ix.rows2 != nullptr) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 1>); \
        else if (ixl.stage_lds != 0u) MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV, 1, 0>);  \
        else MOVI_LAUNCH_SEG(SEGV, LANES, pml_kernel_flatp<6, T, -1, 0, S, 0, SEGV>);                                 \
    } while (0)
#define MOVI_LAUNCH_SEG_T(SEGV, LANES)                                                                                \
    do {                                                                                                              \
        stage_for(LANES);                                                                                             \
        if (ix.idx32) {                                                                                               \
            if (ix.sep) MOVI_LAUNCH_SEG_S(SEGV, LANES, uint32_t, 1); else MOVI_LAUNCH_SEG_S(SEGV, LANES, uint32_t, 0); \
        } else {                                                                                                      \
            if (ix.sep) MOVI_LAUNCH_SEG_S(SEGV, LANES, uint64_t, 1); else MOVI_LAUNCH_SEG_S(SEGV, LANES, uint64_t, 0); \
        }                                                                                                             \
    } while (0)
    MOVI_LAUNCH_SEG_T(1, max_seg);
    if (info) {                                           // the dominant kernel: K1
        const int stg = ixl.stage_lds != 0u ? 1 : 0, ahd = (stg && ix.rows2 != nullptr) ? 1 : 0;
        snprintf(info->kernel, sizeof(info->kernel), "pml_kernel_flatp<6, %s, -1, 0, %d, 0, 1, %d, %d%s>",
                 ix.idx32 ? "unsigned int" : "unsigned long", ix.sep ? 1 : 0, stg, ahd,
                 seg_ring ? (seg_pair ? ", 1, 1" : ", 0, 1") : ((seg_pair && stg) ? ", 1" : ""));
        info->variant = 14; info->block_threads = 64; info->segmented = 1; info->idx64 = ix.idx32 ? 0 : 1;
        info->waves_per_cu = 0; info->staged = (int)ixl.stage_lds; info->ahead = ahd;
    }
    // (blocks of one wavefront: a boundary lane that has to walk far holds up only the 63 beside it)
    const uint32_t max_over = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)cfg.seg_len * (uint64_t)kSegOverrun);
    hipLaunchKernelGGL((seg_stitch_kernel<6, 0>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    hipLaunchKernelGGL(seg_finalize_kernel, dim3((unsigned)((n_reads + bt256 - 1) / bt256)), dim3(bt256), 0, stream, go, first, n_reads,
                       seg.tot, join, seg_l, seg_rem, on_chain, read_fail, d_err, d_stats);
    hipLaunchKernelGGL((seg_stitch_kernel<6, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix, d_bases, seg,
                       seg_j, seg_rem, max_over, on_chain, d_out, join);
    MOVI_LAUNCH_SEG_T(2, n_reads);
#undef MOVI_LAUNCH_SEG_T
#undef MOVI_LAUNCH_SEG_S
#undef MOVI_LAUNCH_SEG
    e = hipGetLastError();
    if (e == hipSuccess && bins.bin_width)
        e = launch_classify(d_out, d_offsets, n_reads, bins.bin_width, bins.thr, bins.above, bins.below, bins.sum_max, stream, d_err);
    return e;
}

hipError_t launch_pml(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                      uint64_t n_reads, uint64_t n_bases, uint16_t *d_out, uint8_t *d_err, DevStats *d_stats,
                      const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream, const ClsArgs &cls,
                      SegWorkspace *seg_ws, int ragged_hint, int *seg_verdict, LaunchInfo *info) {
    if (n_reads == 0) return hipSuccess;
    // 0 = PML vector only, 1 = vector + classification bins, 2 = bins only
    const int cm = cls.bin_width == 0 ? 0 : (d_out ? 1 : 2);
    if (cm == 0 && !d_out) return hipErrorInvalidValue;
    if (cm != 0 && (!cls.above || !cls.below || !cls.sum_max)) return hipErrorInvalidValue;
    // One resident row layout: blocked- and sampled-thresholds tables are expanded to regular-thresholds rows at upload
    // (expand_blocked_kernel / expand_sampled_kernel), so the query kernels exist for MODE 6 only.
    if (mode != 6) return hipErrorInvalidValue;
    // Variants: 0 first correct kernel, 1 base-synchronous packed I/O, 7 flat lane state machine,
    // 10 = 7 + row window + software pipelining, 13 = 10 as a persistent grid with lane refill, 14 = 10 with every in-window
    // fast-forward / scan step resolved in closed form (window_advance; the default).  (2-6, 8, 9, 11, 12 were
    // experiments -- branchy state machine, 2/4-row neighbour windows, the unpipelined window kernel, other hop counts --
    // measured slower and removed; numbers in DESIGN.md section 3.)
    // Auto selection (measured on MI355X, profiles/r02_*): variant 10 in blocks of ONE wavefront, and -- when there are
    // more reads than ~18 waves per CU -- at most kCapWaves wavefronts resident per CU.  Why a cap: between two
    // iterations of a lane its cache lines (the row window's neighbours, its read, its output) must survive in the
    // 4 MiB L2 of its XCD; with all 32 wave slots of a CU walking, 8 MiB of lines are in flight per XCD and neighbour
    // rows are refetched from the fabric.  1 M x 150 bp, Gbases/s, variant 10 uncapped / capped at 8-10 waves per CU /
    // variant 1 (base-synchronous: its neighbour loads follow the gather at once, so it wants all the occupancy it can
    // get): pangenome 14 M rows 43.2 / 48.2 / 46.4; random tables of 10 M rows 40.0 / 47.5 / 43.6, 60 M 35.7 / 40.8 /
    // 36.3, 250 M (2 GB) 29.1 / 32.8 / 29.2, 500 M 27.9 / 30.3 / 27.7, 1 B (8 GB) 27.4 / 27.7 / 28.3.
    // Variant 13 (lane refill) lifts the share of busy lanes from 80 % to 92 % but runs every rare-per-lane block
    // (chunk fetch, PML stores, refill) in every iteration because its lanes are never in step: 46.1 on the pangenome
    // (variant 10 capped: 47.7), 43.5 on the random table; on log-normal read lengths 36.5 against 35.8.  Selectable,
    // not the default.
    int v = cfg.pml_variant;
    const bool logging = cls.log_ff != nullptr || cls.log_scan != nullptr;
    // (batches of up to ~18 waves per CU run in ONE round, uncapped: with the cap, 224 k reads = 13.7 waves per CU run as
    // a full round of 9 and a half-empty one -- 38.3 against 39.2 Gbases/s; 300 k reads: 38.2 against 41.2; from 400 k
    // reads on the cap wins: 43.9 against 41.7.  profiles/r02_occupancy_cap_sweeps.txt)
    const bool big_batch = n_reads > (uint64_t)cfg.num_cus * 64u * 18u;
    if (v < 0) v = 14;
    // variant 14 = variant 10 with the window-parallel advance instead of two sequential hops: +2.5 % on long reads,
    // +5.5 % on the 8 GB table, neutral on the fabric-bound big batches (profiles/r02_window_parallel.txt) -> the default
    bool wp = v == 14;
    if (wp) v = 10;
    const int wpc_refill = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : kCapWaves;
    uint64_t refill_blocks = cfg.refill_blocks > 0 ? (uint64_t)cfg.refill_blocks
                                                   : (uint64_t)cfg.num_cus * (uint64_t)wpc_refill;   // in wavefronts (blocks of 64)
    if (v == 13 && (d_order || n_reads <= refill_blocks * 64u)) { v = 10; wp = true; }   // nothing to refill: the default walk
    if ((v == 10 || v == 13) && (ix.r < 8 || n_bases < 16)) v = 7;           // the clamped window needs >= 4 rows, the
                                                                             // 16-base fetches >= 16 bytes of bases
    if (cm != 0 && (v == 0 || v == 7)) v = (v == 0 || ix.r < 8 || n_bases < 16) ? 1 : 10;   // the A/B kernels carry no fused bins
    if (logging) {                                                           // per-base logs: the first kernel keeps them
        if (cm != 0) return hipErrorInvalidValue;
        v = 0;
    }
    // Batches of long reads: segment-parallel (plain PML through the default kernel only).  One lane per read leaves the
    // GPU short of walks -- 100 k reads are 6 wavefronts per CU, and a single 1 Mbp read holds its lane for 2 s --;
    // cut into segments the same batch fills it like a batch of short reads.
    if (seg_ws && !logging && cfg.seg_len >= 32 && !d_order && wp && v == 10 && cfg.block_threads <= 64 &&
        n_bases / n_reads >= 2ull * (uint64_t)cfg.seg_len && n_reads + n_bases / (uint64_t)cfg.seg_len < 0x7FFFFFF0ull) {
        bool declined = false;
        const hipError_t es = launch_pml_segmented(ix, d_bases, d_offsets, n_reads, n_bases, d_out, d_err, d_stats, cfg, stream,
                                                   seg_ws, cfg.pml_variant < 0 || cfg.pml_variant == 14, ragged_hint, &declined, cls,
                                                   seg_verdict, info);
        if (es != hipSuccess || !declined) return es;
    }
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : 64;           // one wavefront per block: finest dispatch grain
    uint64_t blocks = (n_reads + bt - 1) / bt;
    int wpc = cfg.waves_per_cu;
    if (wpc < 0) wpc = 0;
    if (v == 13 && (cfg.stage_reads == 0 || bt != 64)) { v = 10; wp = true; }         // lane refill: staged one-wavefront blocks only
    const bool stage_ok = cfg.stage_reads != 0 && bt == 64 && ((v == 10 && wp) || v == 13);   // the staged kernels: one-wavefront blocks of the default walk
    const bool ahead_ok = stage_ok && ix.rows2 != nullptr;                              // ... on the look-ahead rows where they exist
    if (cfg.waves_per_cu == 0 && (cfg.pml_variant < 0 || cfg.pml_variant == 14) && v == 10 && big_batch)
        wpc = ahead_ok ? kCapWavesAhead : kCapWaves;                                 // the auto policy above
    if (v == 13) {
        wpc = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : (ahead_ok ? kCapWavesAhead : kCapWaves);
        if (cfg.refill_blocks == 0) refill_blocks = (uint64_t)cfg.num_cus * (uint64_t)wpc;
        const uint64_t resident = (refill_blocks * 64u + bt - 1) / bt;
        if (blocks > resident) blocks = resident;
    }
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block((unsigned)bt);
    // Occupancy cap: enforced by the dispatcher through the block's LDS allocation (160 KiB per CU); blocks beyond
    // the cap queue and start as resident ones retire.  For the persistent grid of variant 13 the same padding makes
    // the dispatcher spread the blocks evenly: exactly wpc wavefronts on every CU.
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
    DevIndex ixl = ix;
    // PMLs out through a ring in LDS (ix.out_ring; the kernel has the numbers): launches of long reads -- few wavefronts, each
    // one's own instruction stream most of an iteration -- where the block's LDS holds the ring beside 96 staged bases.
    // cfg.out_ring: -1 = this policy, 0 / 1 = never / wherever it fits (A/B).
    const bool ring_wanted = stage_ok && v == 10 && (cfg.out_ring > 0 || (cfg.out_ring < 0 && n_bases / n_reads >= kOutRingReadLen));
    if (stage_ok && wpc == 0) {
        const uint64_t wn = (blocks + (uint64_t)cfg.num_cus - 1) / (uint64_t)cfg.num_cus;      // wavefronts per CU of this launch
        // (room for a quarter more: the dispatcher does not deal the blocks out evenly, and a CU that may hold no more than the
        // average leaves its surplus queued -- 150 k reads, 9.2 wavefronts per CU: 41.2 Gbases/s with room for 10, 45.8 for 12)
        const uint64_t room = wn + std::max<uint64_t>(2, wn / 4);
        if (wn <= 18) dyn_lds = std::min<size_t>(21504 + (ring_wanted ? kOutRingBytes : 0u), ((163840u / (unsigned)room) & ~1023u) - 1024u);
    }
    const size_t ring_b = (ring_wanted && dyn_lds >= kOutRingBytes + 96u * 64u) ? kOutRingBytes : 0;
    const uint32_t stage_cap = (uint32_t)std::min<size_t>(1024, ((dyn_lds - ring_b) / 64) & ~(size_t)15);
    ixl.stage_lds = (stage_ok && stage_cap >= 96) ? stage_cap : 0u;
    const bool use_ring = ring_b != 0 && ixl.stage_lds != 0u;
    ixl.refill_batch = cfg.refill_batch > 0 ? (uint32_t)cfg.refill_batch : 16u;
    ixl.inwin = cfg.inwin ? 1u : 0u;
    if (v == 13 && ixl.stage_lds == 0u) return hipErrorInvalidValue;                  // (cannot happen: the refill launch is capped)
    const bool use_ahead = ahead_ok && ixl.stage_lds != 0u;
    // pair-shared gathers (pml_kernel_flatp<..., PSH = 1>): the staged default walk on the plain or the look-ahead rows
    // Where: on tables beyond the reach of the per-CU TLBs (~2 GB), where a lane's two (four) 16-byte loads are as many
    // translation requests and the L2 TLB's request rate bounds the walk -- real BWT of 226 M rows on the look-ahead rows (3.6 GB
    // copy) 39.4 -> 50.8 Gbases/s, the random 1 B-row table 32.6 -> 34.7 on its plain rows and 21.4 -> 44.2 on the look-ahead
    // copy (16 GB); below that the exchange costs about what the merged accesses give (random 25 / 50 / 100 M rows +4 / +5 / -2 %,
    // real 113 M rows +1.5 %, c2 -2.5 %, c3 -9 %: profiles/r04_pair_shared_gathers.txt).  "pair_loads" 1 / 0 forces it.
    const uint64_t walked_bytes = ix.r * (use_ahead ? 16ull : 8ull);
    const bool use_pair = (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && walked_bytes >= kPairLoadBytes)) && ixl.stage_lds != 0u && v == 10;
    const SegArgs no_seg;
    // every kernel that is handed more than 64 KiB of dynamic LDS must opt in first
#define MOVI_SEG_0
#define MOVI_SEG_1 , no_seg
#define MOVI_LAUNCH_K(...) MOVI_LAUNCH_KX(0, __VA_ARGS__)
#define MOVI_LAUNCH_KX(X, ...)                                                                              \
    do {                                                                                                    \
        if (dyn_lds > 65536) {                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&__VA_ARGS__),               \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);  \
            if (ea != hipSuccess) return ea;                                                                \
        }                                                                                                   \
        if (v == 13 && cfg.refill_blocks == 0) {                                                            \
            /* the persistent grid must be resident as a whole: reads are dealt to its waves statically */   \
            int mb = 0;                                                                                     \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&mb, __VA_ARGS__, bt, dyn_lds) == hipSuccess && \
                mb > 0 && (uint64_t)grid.x > (uint64_t)mb * (uint64_t)cfg.num_cus)                          \
                grid.x = (unsigned)((uint64_t)mb * (uint64_t)cfg.num_cus);                                  \
        }                                                                                                   \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, dyn_lds, stream, ixl, d_bases, d_offsets, n_reads,   \
                           d_out, d_err, d_stats, d_order, cls MOVI_SEG_##X);                               \
    } while (0)
#define MOVI_LAUNCH_PML(M, V, C) MOVI_LAUNCH_K(pml_kernel<M, V, C>)
#ifndef MOVI_HA
#define MOVI_HA 2
#endif
#define MOVI_LAUNCH_FLATP_H(M, H, C, S, R)                                                                  \
    do {                                                                                                    \
        if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, H, C, S, R>);                         \
        else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, H, C, S, R>);                                  \
    } while (0)
#define MOVI_LAUNCH_FLATP_G(M, C, S, A, P)                                                                  \
    do {                                                                                                    \
        if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, -1, C, S, 0, 0, 1, A, P, 1>);         \
        else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, -1, C, S, 0, 0, 1, A, P, 1>);                  \
    } while (0)
#define MOVI_LAUNCH_FLATP_STG(M, C, S, R)                                                                   \
    do {                                                                                                    \
        if (use_ring && R == 0) {                                                                           \
            if (use_pair && use_ahead) MOVI_LAUNCH_FLATP_G(M, C, S, 1, 1);                                  \
            else if (use_pair) MOVI_LAUNCH_FLATP_G(M, C, S, 0, 1);                                          \
            else if (use_ahead) MOVI_LAUNCH_FLATP_G(M, C, S, 1, 0);                                         \
            else MOVI_LAUNCH_FLATP_G(M, C, S, 0, 0);                                                        \
        } else if (use_pair && R == 0 && use_ahead) {                                                       \
            if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, -1, C, S, 0, 0, 1, 1, 1>);        \
            else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, -1, C, S, 0, 0, 1, 1, 1>);                 \
        } else if (use_pair && R == 0) {                                                                    \
            if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, -1, C, S, 0, 0, 1, 0, 1>);        \
            else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, -1, C, S, 0, 0, 1, 0, 1>);                 \
        } else if (use_ahead) {                                                                             \
            if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, -1, C, S, R, 0, 1, 1>);           \
            else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, -1, C, S, R, 0, 1, 1>);                    \
        } else {                                                                                            \
            if (ix.idx32) MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint32_t, -1, C, S, R, 0, 1>);              \
            else MOVI_LAUNCH_KX(1, pml_kernel_flatp<M, uint64_t, -1, C, S, R, 0, 1>);                       \
        }                                                                                                   \
    } while (0)
#define MOVI_LAUNCH_FLATP_R(M, C, S, R)                                                                     \
    do {                                                                                                    \
        if (R == 1) MOVI_LAUNCH_FLATP_STG(M, C, S, 1);                                                      \
        else if (wp && ixl.stage_lds) MOVI_LAUNCH_FLATP_STG(M, C, S, 0);                                    \
        else if (wp) MOVI_LAUNCH_FLATP_H(M, -1, C, S, 0); else MOVI_LAUNCH_FLATP_H(M, MOVI_HA, C, S, 0);    \
    } while (0)
#define MOVI_LAUNCH_FLATP_S(M, C, S)                                                                        \
    do {                                                                                                    \
        if (v == 13) MOVI_LAUNCH_FLATP_R(M, C, S, 1); else MOVI_LAUNCH_FLATP_R(M, C, S, 0);                 \
    } while (0)
#define MOVI_LAUNCH_FLATP(M, C)                                                                             \
    do {                                                                                                    \
        if (ix.sep) MOVI_LAUNCH_FLATP_S(M, C, 1); else MOVI_LAUNCH_FLATP_S(M, C, 0);                        \
    } while (0)
#define MOVI_LAUNCH_FLAT(M)                                                                                 \
    do {                                                                                                    \
        if (ix.idx32) MOVI_LAUNCH_K(pml_kernel_flat<M, uint32_t, 0>);                                       \
        else MOVI_LAUNCH_K(pml_kernel_flat<M, uint64_t, 0>);                                                \
    } while (0)
#define MOVI_BY_CLS(LAUNCH, ...)                                                                            \
    do {                                                                                                    \
        if (cm == 0) LAUNCH(__VA_ARGS__, 0); else if (cm == 1) LAUNCH(__VA_ARGS__, 1); else LAUNCH(__VA_ARGS__, 2); \
    } while (0)
    if (v == 0) MOVI_LAUNCH_PML(6, 0, 0); else if (v == 1) MOVI_BY_CLS(MOVI_LAUNCH_PML, 6, 1);
    else if (v == 7) MOVI_LAUNCH_FLAT(6); else MOVI_BY_CLS(MOVI_LAUNCH_FLATP, 6);
    if (info) {
        const char *it = ix.idx32 ? "unsigned int" : "unsigned long";
        if (v == 0 || v == 1) snprintf(info->kernel, sizeof(info->kernel), "pml_kernel<6, %d, %d>", v, v == 0 ? 0 : cm);
        else if (v == 7) snprintf(info->kernel, sizeof(info->kernel), "pml_kernel_flat<6, %s, 0>", it);
        else snprintf(info->kernel, sizeof(info->kernel), "pml_kernel_flatp<6, %s, %d, %d, %d, %d, 0, %d, %d%s>", it, (wp || v == 13) ? -1 : MOVI_HA, cm,
                      ix.sep ? 1 : 0, v == 13 ? 1 : 0, ixl.stage_lds ? 1 : 0, use_ahead ? 1 : 0,
                      use_ring ? (use_pair ? ", 1, 1" : ", 0, 1") : (use_pair ? ", 1" : ""));
        info->variant = (v == 10 && wp) ? 14 : v;
        info->block_threads = bt; info->waves_per_cu = wpc; info->segmented = 0; info->idx64 = ix.idx32 ? 0 : 1;
        info->staged = (int)ixl.stage_lds;
        info->ahead = use_ahead ? 1 : 0;
    }
#undef MOVI_LAUNCH_PML
#undef MOVI_LAUNCH_K
#undef MOVI_LAUNCH_KX
#undef MOVI_SEG_0
#undef MOVI_SEG_1
#undef MOVI_LAUNCH_FLAT
#undef MOVI_LAUNCH_FLATP
#undef MOVI_LAUNCH_FLATP_S
#undef MOVI_LAUNCH_FLATP_R
#undef MOVI_LAUNCH_FLATP_STG
#undef MOVI_LAUNCH_FLATP_G
#undef MOVI_LAUNCH_FLATP_H
#undef MOVI_BY_CLS
    return hipGetLastError();
}

// update_interval, src/move_structure_search.cpp:48-61 (get_char: the '$' row never equals a base): move the interval's
// start down to the first row of character b and its end up to the last one.  If [rs, re] holds such a row both searches
// find one and start <= end; if it holds none the interval is empty, and that is all the callers use (the reference lets
// the start run past the end instead).  So the two searches are independent, each bounded by the OTHER end's original row,
// and each takes the 4-row window around its next row per trip (window base clamped to r - 4: never outside the table)
// instead of one row: the trips of this loop -- max over the wave's lanes -- were most of a ZML step on divergent reads.
// Row / window / look-ahead entry of the table the count query walks on: AH = 0 the plain rows, AH = 1 the look-ahead copy
// (DevIndex::rows2: 8 rows + their 8 entries per 128-byte line, the last window in a line of its own).
template <int MODE, int AH>
__device__ __forceinline__ uint2 tab_row(const DevIndex &ix, uint64_t i) {
    if (!AH) return load_row<MODE>(ix.rows, i);
    uint2 v;
    __builtin_memcpy(&v, ix.rows2 + (i >> 3) * 128u + (i & 7u) * 8u, 8);
    return v;
}
__device__ __forceinline__ uint2 tab_entry(const DevIndex &ix, uint64_t i) {
    uint2 v;
    __builtin_memcpy(&v, ix.rows2 + (i >> 3) * 128u + 64u + (i & 7u) * 8u, 8);
    return v;
}
template <int MODE, int AH>
__device__ __forceinline__ void tab_window(const DevIndex &ix, uint64_t wb, uint2 (&w)[4]) {   // wb: aligned, or r - 4 (the last window)
    if (!AH) load_window<MODE>(ix.rows, wb, w);
    else load_window<MODE>(ix.rows2 + (wb < ix.r - 4 ? (wb >> 3) * 128u + (wb & 4u) * 8u : ix.rows2_tail), 0, w);
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

hipError_t launch_count(int mode, const DevIndex &ix, const uint8_t *d_bases, const uint64_t *d_offsets,
                        uint64_t n_reads, uint64_t *d_matched, uint64_t *d_count, uint8_t *d_err,
                        DevStats *d_stats, const uint32_t *d_order, const LaunchCfg &cfg, hipStream_t stream,
                        LaunchInfo *info) {
    if (n_reads == 0) return hipSuccess;
    // Blocks of one wavefront; on a cache-resident table (up to the 256 MiB of the Infinity Cache) and a batch of more than
    // ~24 wavefronts of reads per CU at most kCountCapWaves wavefronts resident per CU -- the same L2-retention effect as in
    // launch_pml: the search's neighbour rows (interval shrink, fast-forwards) must survive between a lane's steps.
    // profiles/r03_count_ftab.txt: pangenome 61.2 -> 67.5 Gbases/s (cap 15 - 16; 14: 66.2, 17: 64.2, 20: 62.5), random 80 MB
    // table 52.4 -> 54.6; HBM-resident tables lose with any cap (1.6 GB: 47.3 uncapped, 43.2 at 16) or are indifferent (8 GB).
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : 64;
    int wpc = cfg.waves_per_cu > 0 ? cfg.waves_per_cu : 0;
    const bool ahead = mode == 6 && ix.rows2 != nullptr && ix.rows2_count != 0u;   // the search walks on the look-ahead rows where they pay
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
template <int MODE, typename IdxT, int SEG = 0, int AH = 0, int PSH = 0>
__global__ __launch_bounds__(256) void zml_kernel_flat(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order, ZSegArgs seg) {
    static_assert(AH == 0 || (MODE == 6 && SEG == 0), "look-ahead rows: regular-thresholds rows, whole reads");
    static_assert(PSH == 0 || (AH == 0 && SEG == 0 && (MODE == 6 || MODE == 3)), "pair-shared gathers: plain 8-byte rows, whole reads");
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
    uint16_t *O = out + obeg;
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
    if (len > 16) load_pair_at(beg + len - 16, nx0, nx1);
    // b = code of the base of step k, bn = of step k + 1, bn2 = of step k + 2: M2 -> M0 chains two bases inside one
    // iteration (three on the look-ahead rows), so the next bases are decoded ahead (rb always holds the 8-group of step
    // k + 2, the furthest one decoded)
    uint32_t b = s_code[(uint32_t)(rb >> 56) & 0xFFu];
    uint32_t bn = len > 1 ? s_code[(uint32_t)(rb >> 48) & 0xFFu] : 0xFFu;
    uint32_t bn2 = len > 2 ? s_code[(uint32_t)(rb >> 40) & 0xFFu] : 0xFFu;
    uint4 pk = make_uint4(0, 0, 0, 0), pk_old = pk;
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
            if (fail) {                                      // :750-760 / :696-704: the base opens the next phrase
                ml = 0;
                open = 0;
                ph = phStart;
                if (b != 0xFFu) {                            // initialize_backward_search :284-291
                    rs = (IdxT)ix.first_runs[b + 1]; re = (IdxT)ix.last_runs[b + 1];
                    os = (uint32_t)ix.first_offsets[b + 1]; oe = (uint32_t)ix.last_offsets[b + 1];
                    open = ((rs < re) || (rs == re && os <= oe)) ? 1u : 0u;
                    if (open) { ps = pFF; pe = pFF; ffs = 65535u; ffe = 65535u; ph = phInit; }   // rows only: no fast-forward
                }
                book();
                if (k == len) ph = phDone;
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
            emit_at(ekA, valA);
            if (n_emit == 2u) {
                emit_at(ekB, valB);
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
        if (failed) {
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
    if (ix.r <= (3ull << 30) / 8 && ix.r >= 8 && n_bases >= 16) {
        if (ix.idx32)
            hipLaunchKernelGGL((zml_kernel_flat<MODE, uint32_t, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix,
                               d_bases, d_offsets, max_seg, d_out, d_err, d_stats, d_order, seg);
        else
            hipLaunchKernelGGL((zml_kernel_flat<MODE, uint64_t, 1>), dim3((unsigned)((max_seg + 63) / 64)), dim3(64), 0, stream, ix,
                               d_bases, d_offsets, max_seg, d_out, d_err, d_stats, d_order, seg);
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
            if (sm) snprintf(info->kernel, sizeof(info->kernel), "zml_kernel_flat<%d, %s, 1>", mode, ix.idx32 ? "unsigned int" : "unsigned long");
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
    int v = cfg.zml_variant;
    const bool big = ix.r * 8ull >= kPairLoadBytes;
    if (v < 0) v = (ix.r <= (3ull << 30) / 8 || (big && cfg.pair_loads != 0)) ? 1 : 0;
    if (v == 1 && (ix.r < 8 || n_bases < 16)) v = 0;     // the clamped windows need >= 4 rows, the 16-base fetches 16 bytes
    const int bt = cfg.block_threads > 0 ? cfg.block_threads : (v == 1 ? 64 : 256);
    const uint64_t blocks = (n_reads + bt - 1) / bt;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks), block((unsigned)bt);
    // The state machine on the look-ahead rows (round 4; "zml_ahead" 1, where the copy exists): a base both of whose LF moves
    // land without a fast-forward is complete without the target rows.  Lane iterations per base on c2 1.45 -> 0.98 -- and 37.3
    // instead of 38.2 Gbases/s (eight 16-byte loads per iteration instead of four, SIMT 0.72 -> 0.64; random 10 M-row table 35.9 ->
    // 35.0: profiles/r04_zml_ahead.txt), so it is an option, not the default.
    const bool ahead = cfg.zml_ahead != 0 && v == 1 && mode == 6 && ix.rows2 != nullptr;
    const bool pair = v == 1 && !ahead && (cfg.pair_loads > 0 || (cfg.pair_loads < 0 && big));
    if (info) {
        if (v == 1) snprintf(info->kernel, sizeof(info->kernel), "zml_kernel_flat<%d, %s, 0%s>", mode, ix.idx32 ? "unsigned int" : "unsigned long",
                             ahead ? ", 1" : (pair ? ", 0, 1" : ""));
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
#define MOVI_LAUNCH_ZML(M)                                                                                     \
    do {                                                                                                       \
        if (v == 0)                                                                                            \
            hipLaunchKernelGGL(zml_kernel<M>, grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads,   \
                               d_out, d_err, d_stats, d_order, ZSegArgs());                                    \
        else if (pair && ix.idx32)                                                                             \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint32_t, 0, 0, 1>), grid, block, dyn_lds, stream, ix,      \
                               d_bases, d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs());       \
        else if (pair)                                                                                         \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint64_t, 0, 0, 1>), grid, block, dyn_lds, stream, ix,      \
                               d_bases, d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs());       \
        else if (ix.idx32)                                                                                     \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint32_t>), grid, block, dyn_lds, stream, ix, d_bases,      \
                               d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs());                \
        else                                                                                                   \
            hipLaunchKernelGGL((zml_kernel_flat<M, uint64_t>), grid, block, dyn_lds, stream, ix, d_bases,      \
                               d_offsets, n_reads, d_out, d_err, d_stats, d_order, ZSegArgs());                \
    } while (0)
    // resident layouts: 6 = regular-thresholds rows, 3 = regular rows (threshold-less types: 12-bit lengths)
    if (mode == 6 && ahead) {
        if (ix.idx32)
            hipLaunchKernelGGL((zml_kernel_flat<6, uint32_t, 0, 1>), grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads, d_out,
                               d_err, d_stats, d_order, ZSegArgs());
        else
            hipLaunchKernelGGL((zml_kernel_flat<6, uint64_t, 0, 1>), grid, block, dyn_lds, stream, ix, d_bases, d_offsets, n_reads, d_out,
                               d_err, d_stats, d_order, ZSegArgs());
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
