// movi_device.hpp -- device-side helpers shared by every kernel translation unit: row decode, LF_move / fast_forward,
// the reposition rules, the classification bins.  (Split out of movi_kernels.hip in round 5 so that the walk kernel's
// instantiations compile in parallel translation units; the reference citations are those of the functions below.)
#pragma once
#include "movi_kernels.hpp"

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

// pair-shared gathers (movi_walk.hpp, zml_kernel_flat): lanes 2i / 2i + 1 fetch their windows together
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

}  // namespace movi
