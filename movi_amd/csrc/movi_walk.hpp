// movi_walk.hpp -- the PML walk: pml_kernel_flatp, the lane state machine every PML query runs on, and the dispatch
// from a launch's run-time choices to its instantiation.  Included by the movi_walk_*.hip translation units only.
#pragma once
#include "movi_device.hpp"

namespace movi {

// The walk ("flat state machine + row window, software-pipelined, window-parallel advance": what rounds 2 - 4 called
// variant 14): every lane runs the reference's per-base automaton (LF_move -> fast_forward -> match / reposition scan) for its
// own read, one row-window gather per iteration, with
//   * the 4-row WINDOW around the row it needs fetched instead of the row (32 B of one cache line, one L2 request) and every
//     fast-forward / scan step that stays inside it resolved in closed form (window_advance);
//   * the next window's address computed from the row alone (all selects) and its load issued BEFORE the step's
//     bookkeeping (PML packing and stores, bins, counters, base decode), which then runs under the gather's latency; the
//     load is unpredicated and branch-free (the table's last window is pulled back to rows [r-4, r); finished lanes re-read
//     window 0) -- with a predicated two-path fetch hipcc parked a `s_waitcnt vmcnt(0)` right behind the load and the
//     overlap was gone;
//   * STG = 0 only: read chunks double-buffered, 16 bases per fetch (the chunk load is always an L2 miss: its line was
//     evicted long ago), PMLs out as paired 16-byte stores.
// Removed in round 5 (measured, never a default; the code is in the history at commit ceb31ea): the hop-by-hop advance
// (HA >= 0: up to HA dependent in-window hops instead of the closed form; -2.5 % on long reads, -5.5 % on the 8 GB table,
// profiles/r02_window_parallel.txt), the row-at-a-time state machine without a window (pml_kernel_flat, "variant 7"), and
// LANE REFILL (REFILL = 1, "variant 13": a persistent grid whose idle lanes take the next reads of a per-wavefront pool fed
// from one global ticket counter; SIMT efficiency 0.72 -> 0.87 on c2 and no faster in rounds 2 and 4 -- fuller wavefronts
// issue more gathers per iteration and the fabric serves them no faster: profiles/r04_lane_refill.txt).
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
template <int MODE, typename IdxT, int CLS, int SEP, int SEG, int STG, int AHD, int PSH, int RING>
__global__ __launch_bounds__(256) void pml_kernel_flatp(DevIndex ix, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offs, uint64_t n_reads,
                                                       uint16_t *__restrict__ out, uint8_t *__restrict__ err,
                                                       DevStats *stats, const uint32_t *__restrict__ order,
                                                       ClsArgs cls, SegArgs seg) {
    static_assert(SEG == 0 || CLS == 0, "segments: plain PML");
    static_assert(AHD == 0 || STG == 1, "look-ahead rows: staged reads");
    static_assert(AHD == 0 || AHD == 1 || AHD == 2, "plain rows, look-ahead rows or deep rows");
    static_assert(AHD != 2 || (PSH == 0 && sizeof(IdxT) == 4), "deep rows: tables of fewer than 2^28 rows, no pair-shared gathers");
    static_assert(PSH == 0 || STG == 1, "pair-shared gathers: staged kernels");
    static_assert(RING == 0 || STG == 1, "PMLs out through the LDS ring / as reset masks: staged kernels");
    static_assert(RING != 2 || (CLS == 0 && SEG == 0), "reset masks: plain PML of whole reads");
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
    const bool valid = (SEG == 1 ? (*seg.go != 0u && t < *seg.n_seg) : (t < n_reads && (SEG != 2 || seg.read_fail[t] != 0)));
    const uint64_t rid = (valid && order && SEG == 0) ? order[t] : t;
    const uint64_t beg = valid ? (SEG == 1 ? seg.seg_in[rid] : offs[rid]) : 0;
    const uint32_t len = valid ? (SEG == 1 ? seg.seg_len[rid] : (uint32_t)(offs[rid + 1] - beg)) : 0;   // reads are shorter than 2^32 (checked on the host)
    const uint64_t obeg = (SEG == 1 && valid) ? seg.seg_out[rid] : beg;   // where the read's (segment's) PMLs go
    constexpr bool ring = RING == 1;                      // PMLs leave through the ring in LDS (below) instead of the register packer
    // RING == 2 (round 6): RESET MASKS out instead of PMLs.  PML[k] = reset(k) ? 0 : PML[k - 1] + 1 (MoveQuery::add_ml of a match_len
    // that process_char either increments or zeroes, src/read_processor.cpp:193-215, include/move_query.hpp:26-38), so the vector is
    // a function of one bit per base: 1 = match_len was reset at this base (mismatch or illegal character).  `out` is then an array
    // of 32-bit words: bit k % 32 of word ((beg + ix.mask_phase) >> 5) + rid + k / 32 belongs to step k of read rid -- every read
    // starts a word of its own without a prefix sum over the reads (floor((o + l) / 32) + 1 >= floor(o / 32) + ceil(l / 32)).  An
    // emission is one v_lshl_or_b32, a finished word one 4-byte store; pml_expand_kernel / movi_pml_expand_host turn the words back
    // into the u16 vector (u16 clamp included).  1/16 of the bytes of the vector.
    constexpr bool msk = RING == 2;
    uint32_t *const M = msk ? reinterpret_cast<uint32_t *>(out) + (((beg + ix.mask_phase) >> 5) + rid) : nullptr;
    uint32_t mk = 0;
    const uint32_t packed_end = len & (ring ? ~15u : ~7u);      // PMLs of steps >= this are stored one by one

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
    uint4 dq[4];                                          // AHD == 2: the 64-byte window of the deep rows (three rows of 5 dwords + their 10 extra bits each in dword 15)
    const uint32_t odd_lane = threadIdx.x & 1u;
    auto fetch = [&](IdxT nd, bool act, uint2 (&w)[4]) {
        if (AHD == 2) {                                   // window q = rows 3q .. 3q + 2 at byte 64 q (DevIndex::rows3); the last window is padded
            const uint32_t q3 = __umulhi((uint32_t)nd, 0xAAAAAAABu) >> 1;
            const uint8_t *at = ix.rows3 + (act ? (uint64_t)q3 * 64u : 0u);
            __builtin_memcpy(&dq[0], at, 16);
            __builtin_memcpy(&dq[1], at + 16, 16);
            __builtin_memcpy(&dq[2], at + 32, 16);
            __builtin_memcpy(&dq[3], at + 48, 16);
            return;
        }
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
        if (msk) {                                        // a failed read reports all-zero PMLs: every base a reset
            if (failed) {
                for (uint32_t wd = 0; wd * 32u < len; ++wd) M[wd] = (len - wd * 32u >= 32u) ? 0xFFFFFFFFu : ((1u << (len & 31u)) - 1u);
            } else if (len & 31u) {
                M[len >> 5] = mk;                         // the read's last, partial word
            }
        } else if (failed && CLS != 2) {
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
        if (msk) {
            if (use) {                                    // K <= 12 bases: the match mask's complement is the word so far
                const uint32_t mask = (e4.y >> 16) & 0xFFFu;
                mk = ~mask & ((1u << K) - 1u);
                k = K;
                ml = (uint32_t)__builtin_clz(~(mask << (32u - K)));      // the run of matches that ends at base K - 1
                need = (IdxT)((uint64_t)e4.x | ((uint64_t)(e4.y & 15u) << 32));
                off = (e4.y >> 4) & 0xFFFu;
                ff_total += e4.z;
                scan_total += e4.w;
                repo_total += K - (uint32_t)__popc(mask);
            }
        } else if (use) {
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
    if (ix.kmer_k != 0u) {                     // wave-uniform
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
    uint32_t a1 = 0xFFu, a2 = 0xFFu;
    if (AHD) a1 = staged_code(k + 1);
    if (AHD == 2) a2 = staged_code(k + 2);
    uint2 w[4];
    fetch(need, st != sDone, w);

    uint32_t lane_steps = 0, wave_steps = 0;
    while (wave_any(st != sDone)) {
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
        // What the analysis of this iteration's window hands to the bookkeeping below (the two layouts -- four-row windows of 8-byte rows
        // with or without look-ahead entries; three-row windows of deep rows -- analyse their windows in their own code and meet here).
        struct StepOut {
            uint32_t resolved, match, ffm, lf, mism, scanning, found, down, qf, qn, jump, jdist, landed, landed_down, nf, n, rofff, emit;
            uint32_t dbl, lf2, off1, tpl, lf3, off2, errc, st_next;
            uint64_t j;
            IdxT needf, need_next, j2, j3;
        } R;
        R.tpl = 0; R.lf3 = 0; R.off2 = 0; R.j3 = 0;
        if (AHD == 2) {
            // ---- DEEP ROWS (round 6; DevIndex::rows3): the window is the aligned group of THREE rows that holds `need` (64 bytes: half a
            // cache line), and every row carries what the walk reads at its LF target j AND at j's target j2 -- so that up to THREE
            // bases are resolved per gather: base k here; base k + 1 at j if it is c(j) and the offset arrives below n(j); base k + 2 at
            // j2 likewise; then the gather goes to j3 = id(j2).  Same automaton, same answers as the four-row code in the other branch.
            const uint32_t need32 = (uint32_t)need;
            const uint32_t wbase = (__umulhi(need32, 0xAAAAAAABu) >> 1) * 3u;
            const uint32_t r1u = (uint32_t)r1;
            const uint32_t d1a = dq[0].y, d1b = dq[1].z, d1c = dq[2].w;                   // dword 1 of rows 0, 1, 2: n | off << 11 | c << 22 | ...
            const uint32_t n0 = d1a & 0x7FFu, n1w = d1b & 0x7FFu, n2w = d1c & 0x7FFu;
            const uint32_t nm = (uint32_t)(((d1a >> 22) & 7u) != a) | ((uint32_t)(((d1b >> 22) & 7u) != a) << 1) | ((uint32_t)(((d1c >> 22) & 7u) != a) << 2);
            const uint32_t lim = r1u - wbase;                                              // rows of the window before row r - 1 (>= 3 everywhere but in the last window)
            const uint32_t first_win = (uint32_t)(wbase == 0u);
            uint32_t nd = need32;
            {   // window_advance for three rows, in closed form (near the reference's fast-forward limit the count is clamped: `room`)
                const uint32_t q0 = nd - wbase;
                const uint32_t inw = (uint32_t)(q0 < 3u) & (uint32_t)(st < sDone);
                const uint32_t m0 = q0 == 0u, m1 = q0 <= 1u;
                const uint32_t t1 = m0 ? n0 : 0u, t2 = t1 + (m1 ? n1w : 0u), t3 = t2 + n2w;
                const uint32_t isff = inw & (uint32_t)(st == sFF);
                const uint32_t room = 65534u - (ff_run < 65534u ? ff_run : 65534u);       // fast-forwards left before the reference throws (:72-75): the throw itself is the full step's
                uint32_t p0 = isff & m0 & (uint32_t)(off >= t1) & (uint32_t)(0u < lim), p1 = isff & m1 & (uint32_t)(off >= t2) & (uint32_t)(1u < lim),
                         p2 = isff & (uint32_t)(off >= t3) & (uint32_t)(2u < lim);
                uint32_t cf = p0 + p1 + p2;
                if (cf > room) {                                                           // (practically never) stop where the limit is: rows are passed in order
                    cf = room;
                    const uint32_t first = m0 ? 0u : (m1 ? 1u : 2u);                       // first row that takes part
                    p0 = p0 & (uint32_t)(first + cf > 0u);
                    p1 = p1 & (uint32_t)(first + cf > 1u);
                    p2 = p2 & (uint32_t)(first + cf > 2u);
                }
                off -= (p2 ? t3 : (p1 ? t2 : (p0 ? t1 : 0u)));
                ff_run += cf;
                const uint32_t dmask = (nm & (lim >= 3u ? 7u : ((1u << lim) - 1u))) >> q0;    // row r - 1 is never passed
                const uint32_t cd = (inw & (uint32_t)(st == sDown)) ? (uint32_t)__builtin_ctz(~dmask | 8u) : 0u;
                const uint32_t umask = ((nm & (first_win ? 6u : 7u)) << (2u - (q0 < 3u ? q0 : 2u))) & 7u;   // row 0 is never passed
                const uint32_t cu = (inw & (uint32_t)(st == sUp)) ? (uint32_t)__builtin_clz(((~umask) & 7u) << 29 | 0x10000000u) : 0u;
                scan_total += cd + cu;
                nd = nd + cf + cd - cu;
            }
            need = (IdxT)nd;
            const uint32_t qn = nd - wbase;
            const uint32_t inwin = (uint32_t)(qn < 3u) & (uint32_t)act;
            auto sel3 = [](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t q) { return q == 0u ? x0 : (q == 1u ? x1 : x2); };
            const uint32_t rD0 = sel3(dq[0].x, dq[1].y, dq[2].z, qn), rD1 = sel3(d1a, d1b, d1c, qn);
            const uint32_t n = rD1 & 0x7FFu, c = (rD1 >> 22) & 7u;
            const uint32_t isFF = (uint32_t)(st == sFF) & inwin, isDown = (uint32_t)(st == sDown) & inwin, isUp = (uint32_t)(st == sUp) & inwin;
            const uint32_t ffm = isFF & (uint32_t)(nd < r1u) & (uint32_t)(off >= n);       // fast_forward, move_structure.cpp:524-545
            const uint32_t ff_over = ffm & (uint32_t)(ff_run + 1 >= 65535u);               // :72-75
            const uint32_t resolved = isFF & (ffm ^ 1u);
            const uint32_t illegal = a == 0xFFu, match = c == a;
            const uint32_t mism = resolved & (illegal ^ 1u) & (match ^ 1u);
            const uint32_t kk = thr_slot(SEP, a, c);                                       // alphamap_3[c][a]
            const uint32_t kc = kk > 2u ? 2u : kk;
            uint32_t thr = ((rD0 >> (28u + kc)) & 1u) ? n : 0u;
            if (SEP) {
                if (mism & (uint32_t)(c == 0u) & (uint32_t)(need != end_row)) thr = separator_threshold(ix, (uint64_t)need, a);
            }
            const uint32_t down = (uint32_t)(off >= ((need == end_row) ? end_threshold(SEP, ethr, a) : thr));
            const uint32_t at_last = nd >= r1u, at_first = nd == 0u;
            const uint32_t repo_edge = mism & (down ? at_last : at_first);
            const uint32_t has = (nm ^ 7u) & (down ? (6u << qn) & 7u : (1u << qn) - 1u);   // rows of the window that hold the base, beyond row qn
            const uint32_t found = mism & (uint32_t)(has != 0u) & ix.inwin;
            const uint32_t qf = found ? (down ? (uint32_t)__builtin_ctz(has | 8u) : 31u - (uint32_t)__builtin_clz(has | 1u)) : qn;
            const uint32_t far = mism & (found ^ 1u);
            // the row the base is resolved at (qf): its five dwords and ten extra bits
            const uint32_t f0 = sel3(dq[0].x, dq[1].y, dq[2].z, qf), f1 = sel3(d1a, d1b, d1c, qf), f2 = sel3(dq[0].z, dq[1].w, dq[3].x, qf),
                           f3 = sel3(dq[0].w, dq[2].x, dq[3].y, qf), f4 = sel3(dq[1].x, dq[2].y, dq[3].z, qf);
            const uint32_t fx = (dq[3].w >> (10u * (qf < 3u ? qf : 0u))) & 0x3FFu;
            const uint32_t nf = f1 & 0x7FFu, rofff = (f1 >> 11) & 0x7FFu;
            const uint32_t scanning = isDown | isUp;
            const uint32_t hit = scanning & match;
            const uint32_t landed = hit | found;
            const uint32_t landed_down = hit ? isDown : down;
            const uint32_t scan_edge = scanning & (hit ^ 1u) & (isDown ? at_last : at_first);
            const uint32_t emit = (resolved & (illegal | match)) | landed;
            const uint32_t lf = emit & (uint32_t)(k + 1 != len);
            const uint32_t jj = f0 & 0x0FFFFFFFu;                                           // id(row): 28 bits; 0x0FFFFFFF = not a row (r < 2^28)
            const uint32_t lf_bad = lf & (uint32_t)(jj >= (uint32_t)ix.r);
            // reposition hints: two bits per threshold slot (rows beyond the window's edge: 1 .. 3)
            const uint32_t h6 = (f2 >> 28) | ((fx >> 8) << 4);
            const uint32_t hd = __builtin_amdgcn_ubfe(h6, kc + kc, ix.hint_w);            // (hint_w: 2 here, or 0 = switched off)
            const uint32_t jump = far & (uint32_t)(hd != 0u);
            const uint32_t jtgt = down ? wbase + 2u + hd : wbase - hd;
            const uint32_t jdist = down ? jtgt - nd : nd - jtgt;
            const uint32_t step_fwd = ffm | (far & down) | (scanning & (hit ^ 1u) & isDown);
            const uint32_t step_back = (far & (down ^ 1u)) | (scanning & (hit ^ 1u) & isUp);
            uint32_t need_next = lf ? jj : (jump ? jtgt : nd + step_fwd - step_back);
            uint32_t st_next = (emit & (lf ^ 1u)) ? sDone : (lf ? sFF : (far ? (down ? sDown : sUp) : st));
            // the two bases after this one, at j and at j2 (read_processor.cpp:188-238 with match and no fast-forward: ml + 1, LF_move again)
            const uint32_t e1n = f3 & 0x7FFu, e1off = (f3 >> 11) & 0x7FFu, e1c = (f1 >> 25) & 7u, jj2 = f2 & 0x0FFFFFFFu;
            const uint32_t e2n = (f3 >> 22) | (((f4 >> 28) & 1u) << 10), e2off = ((f4 >> 29) & 7u) | ((fx & 0xFFu) << 3), e2c = (f1 >> 28) & 7u,
                           jj3 = f4 & 0x0FFFFFFFu;
            const uint32_t off_e = (landed ? (landed_down ? 0u : nf - 1u) : off) + rofff;
            const uint32_t dbl = lf & (uint32_t)(a1 == e1c) & (uint32_t)(off_e < e1n);    // (an invalid entry has c = 7: no base code equals it)
            const uint32_t lf2 = dbl & (uint32_t)(k + 2 != len);
            const uint32_t off_e2 = off_e + e1off;
            const uint32_t tpl = lf2 & (uint32_t)(a2 == e2c) & (uint32_t)(off_e2 < e2n);
            const uint32_t lf3 = tpl & (uint32_t)(k + 3 != len);
            need_next = tpl ? (lf3 ? jj3 : nd) : (dbl ? (lf2 ? jj2 : nd) : need_next);
            st_next = tpl ? (lf3 ? sFF : sDone) : (dbl ? (lf2 ? sFF : sDone) : st_next);
            uint32_t errc = kErrNone;
            if (wave_any((ff_over | repo_edge | scan_edge | lf_bad) != 0u)) {
                errc = ff_over ? kErrFastForward
                               : (repo_edge ? (down ? kErrNoRunBelow : kErrNoRunAbove)
                                  : (scan_edge ? (isDown ? kErrNoRunBelow : kErrNoRunAbove)
                                     : (lf_bad ? kErrIdRange : kErrNone)));
                if (errc) { need_next = nd; st_next = sDone; }
            }
            R.resolved = resolved; R.match = match; R.ffm = ffm; R.lf = lf; R.mism = mism; R.scanning = scanning; R.found = found; R.down = down;
            R.qf = qf; R.qn = qn; R.jump = jump; R.jdist = jdist; R.landed = landed; R.landed_down = landed_down; R.nf = nf; R.n = n; R.rofff = rofff;
            R.emit = emit; R.dbl = dbl; R.lf2 = lf2; R.off1 = e1off; R.tpl = tpl; R.lf3 = lf3; R.off2 = e2off; R.errc = errc; R.st_next = st_next;
            R.j = jj; R.needf = (IdxT)(wbase + qf); R.need_next = (IdxT)need_next; R.j2 = (IdxT)jj2; R.j3 = (IdxT)jj3;
        } else {
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
            // "window-parallel" advance: everything the hops could do inside this window, in closed form instead
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
            if (wave_any(st == sFF && ff_run >= 65520u)) {
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
            // (ids in the look-ahead copy of a table of fewer than 2^32 - 1 rows are 32 bits wide: DevIndex::hints)
            constexpr bool id32 = AHD != 0 && sizeof(IdxT) == 4;
            uint64_t j = 0;
            if (id32) j = (uint64_t)rowf.x;
            else if (MODE == 6 || lf) j = (AHD && ix.hints) ? (uint64_t)rowf.x : row_id<MODE>(rowf, needf, ix);
            const uint32_t lf_bad = lf & (uint32_t)(j >= ix.r);
            // AHD: the entry of the row the base is resolved at -- or, on a mismatch that is not resolved here, of the row it was seen at
            uint2 ah = make_uint2(0u, 0u);
            if (AHD) ah = win_sel(ahw, qf);
            // Reposition hints (DevIndex::hints): a mismatch whose scan leaves the window knows, for scans of up to 7 rows beyond the
            // window's edge, WHERE the scan ends -- the next iteration gathers that row's window (in the scanning state: the row
            // matches, the scan lands there) instead of the neighbouring window, and the ones after it.  tools/iter_model.c: lane
            // iterations per base 1.123 -> 0.987 on 10 kbp reads with 8 % substitutions, 0.649 -> 0.626 on 150 bp reads with 1 %.
            uint32_t jump = 0, jdist = 0;
            IdxT jtgt = 0;
            if (AHD) {
                const uint32_t hbits = (rowf.y >> 28) | ((ah.y >> 21) & 0x3F0u);
                const uint32_t kc = kk > 2u ? 2u : kk;
                const uint32_t hd = __builtin_amdgcn_ubfe(hbits, kc + 2u * kc, ix.hint_w);          // (hint_w: 3, or 0 = no hints / switched off)
                jump = far & (uint32_t)(hd != 0u);
                jtgt = down ? (IdxT)(wbase + 3u + hd) : (IdxT)(wbase - hd);
                jdist = down ? (uint32_t)(jtgt - need) : (uint32_t)(need - jtgt);
            }
            const uint32_t step_fwd = ffm | (far & down) | (scanning & (hit ^ 1u) & isDown);
            const uint32_t step_back = (far & (down ^ 1u)) | (scanning & (hit ^ 1u) & isUp);
            IdxT need_next = lf ? (IdxT)j : (jump ? jtgt : (IdxT)(need + step_fwd - step_back));
            uint32_t st_next = (emit & (lf ^ 1u)) ? sDone : (lf ? sFF : (far ? (down ? sDown : sUp) : st));
            // AHD: the base after this one, resolved at the LF target from the look-ahead entry (read_processor.cpp:188-238 with
            // match and no fast-forward: ml + 1, then LF_move again) -- the target row itself is never fetched
            uint32_t dbl = 0, lf2 = 0, off1 = 0;
            IdxT j2 = 0;
            if (AHD) {
                const uint32_t n1 = ah.y & 0x7FFu, c1 = (ah.y >> 22) & 7u;
                const uint32_t off_e = (landed ? (landed_down ? 0u : nf - 1u) : off) + rofff;
                dbl = lf & (ah.y >> 31) & (uint32_t)(a1 == c1) & (uint32_t)(off_e < n1);
                lf2 = dbl & (uint32_t)(k + 2 != len);
                off1 = (ah.y >> 11) & 0x7FFu;
                j2 = (id32 || ix.hints) ? (IdxT)ah.x : (IdxT)((uint64_t)ah.x | ((uint64_t)((ah.y >> 25) & 15u) << 32));
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
            R.resolved = resolved; R.match = match; R.ffm = ffm; R.lf = lf; R.mism = mism; R.scanning = scanning; R.found = found; R.down = down;
            R.qf = qf; R.qn = qn; R.jump = jump; R.jdist = jdist; R.landed = landed; R.landed_down = landed_down; R.nf = nf; R.n = n; R.rofff = rofff;
            R.emit = emit; R.dbl = dbl; R.lf2 = lf2; R.off1 = off1; R.errc = errc; R.st_next = st_next;
            R.j = j; R.needf = needf; R.need_next = need_next; R.j2 = j2;
        }
        const uint32_t resolved = R.resolved, match = R.match, ffm = R.ffm, lf = R.lf, mism = R.mism, scanning = R.scanning, found = R.found,
                       down = R.down, qf = R.qf, qn = R.qn, jump = R.jump, jdist = R.jdist, landed = R.landed, landed_down = R.landed_down,
                       nf = R.nf, n = R.n, rofff = R.rofff, emit = R.emit, dbl = R.dbl, lf2 = R.lf2, off1 = R.off1, tpl = R.tpl, lf3 = R.lf3,
                       off2 = R.off2, errc = R.errc, st_next = R.st_next;
        const uint64_t j = R.j;
        const IdxT needf = R.needf, need_next = R.need_next, j2 = R.j2, j3 = R.j3;
        (void)j3; (void)tpl; (void)lf3; (void)off2;
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
        scan_total += scanning + (found ? (down ? qf - qn : qn - qf) : 0u) + (jump ? jdist - 1u : 0u);   // (a jump's last row is counted where it lands)
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
                } else if (msk) {
                    mk |= (uint32_t)(mlv == 0u) << (k & 31u);
                    if ((k & 31u) == 31u) { M[k >> 5] = mk; mk = 0u; }
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
            if (AHD == 2 && tpl) {                        // ... and the third (deep rows): at j2
                ml += 1;
                if (SEG == 1) seg_record((uint64_t)j2, off);
                off += lf3 ? off2 : 0u;
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
                                    ((uint32_t)(ahead_of >= stage_cap) | ((uint32_t)(ahead_of + 1u >= stage_cap) & (uint32_t)(k + 1 < len)) |
                                     (AHD == 2 ? ((uint32_t)(ahead_of + 2u >= stage_cap) & (uint32_t)(k + 2 < len)) : 0u));
            if (wave_any(out_of != 0u)) stage_from(k, st != sDone);
            a = staged_code(k - kbase);
            if (AHD) a1 = staged_code(k + 1 - kbase);
            if (AHD == 2) a2 = staged_code(k + 2 - kbase);
        }
        // ONE load site per prefetch register set and iteration, behind every read of those registers: a second site (or
        // a temporary that the register allocator parks in them where they are dead) costs an `s_waitcnt` on a load
        // in flight, i.e. on the row gather issued above
        if (want_nx) load_pair_at(nx_e, nx0, nx1);
    }
    if (valid) finish_read();
    if (msk && ix.expand_out != nullptr) {
        // ---- the wavefront's walks are over: its reads' reset masks -> their u16 PML vectors (round 6).  The 64 reads of a one-wavefront
        // block (reads in order) are ONE contiguous stretch of the vector: every lane expands its own read -- match_len carried in a register,
        // eight bases per step -- into a tile of the stretch in the LDS its bases were staged in, and the wavefront copies the tile out as
        // aligned 16-byte stores, a contiguous kilobyte per instruction (pml_expand_tile_kernel is the same code as a kernel of its own).
        // The other wavefronts of the CU are still waiting on their gathers: the expansion costs the launch next to nothing.
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");            // this lane's own word stores before its loads of them
        __syncthreads();
        uint16_t *tile = reinterpret_cast<uint16_t *>(s_stage);
        const uint32_t tile_elems = ix.stage_lds * 32u;                   // (stage_lds bytes per lane x 64 lanes / 2; a multiple of 8)
        const uint32_t ln = threadIdx.x & 63u;
        const unsigned long long vb = __ballot(valid);
        if (vb != 0ull) {
            const uint64_t B0 = __shfl(beg, 0, 64);                        // lane 0 is valid whenever any lane is
            const uint64_t B1 = __shfl(beg + len, 63 - __builtin_clzll(vb), 64);
            const uint32_t *Mx = reinterpret_cast<const uint32_t *>(out) + (((beg + ix.mask_phase) >> 5) + rid);
            uint16_t *vec = ix.expand_out;
            uint32_t kx = 0, run = 0, wcur = 0, w0 = 0, w1 = 0;
            const uint64_t endx = beg + len;
            if (valid && len) { w0 = Mx[0]; w1 = Mx[1]; }                 // (the array has a spare word at its end)
            auto one = [&](uint32_t o) {
                if ((kx >> 5) != wcur) { wcur = kx >> 5; w0 = w1; w1 = Mx[wcur + 1]; }
                run = ((w0 >> (kx & 31u)) & 1u) ? 0u : run + 1u;
                tile[o] = (uint16_t)(run > 65535u ? 65535u : run);
                kx += 1;
            };
            for (uint64_t A = B0 & ~7ull; A < B1; A += tile_elems) {
                const uint64_t hi = A + tile_elems < B1 ? A + tile_elems : B1;
                if (valid && kx < len && beg + kx < hi) {
                    uint32_t o = (uint32_t)(beg + kx - A);
                    const uint32_t stop = (uint32_t)((endx < hi ? endx : hi) - A);
                    while (o < stop && (o & 7u)) one(o++);
                    while (o + 8u <= stop) {
                        if ((kx >> 5) != wcur) { wcur = kx >> 5; w0 = w1; w1 = Mx[wcur + 1]; }
                        const uint32_t bits = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (kx & 31u)) & 0xFFu;
                        uint32_t v[8];
#pragma unroll
                        for (uint32_t e = 0; e < 8u; ++e) {
                            run = ((bits >> e) & 1u) ? 0u : run + 1u;
                            v[e] = run > 65535u ? 65535u : run;
                        }
                        *reinterpret_cast<uint4 *>(&tile[o]) = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
                        o += 8u;
                        kx += 8u;
                    }
                    while (o < stop) one(o++);
                }
                __syncthreads();
                const uint32_t nelem = (uint32_t)(hi - A);
                for (uint32_t g = ln; g * 8u < nelem; g += 64u) {
                    const uint64_t p = A + (uint64_t)g * 8u;
                    const uint4 q = *reinterpret_cast<const uint4 *>(&tile[g * 8u]);
                    if (p >= B0 && p + 8u <= hi) {
                        *reinterpret_cast<uint4 *>(vec + p) = q;
                    } else {
                        const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (uint32_t e = 0; e < 8u; ++e)
                            if (p + e >= B0 && p + e < hi) vec[p + e] = (uint16_t)(x[e >> 1] >> (16u * (e & 1u)));
                    }
                }
                __syncthreads();
            }
        }
    }
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

// ---- from a launch's run-time choices (WalkLaunch, movi_kernels.hpp) to its instantiation.  Every translation unit that
// includes this header instantiates the kernels of ONE (IdxT, SEG class): movi_walk_u32.hip / _u64.hip (whole reads),
// movi_walkseg_u32.hip / _u64.hip (segments and re-walked reads), so that they compile side by side.
template <typename IdxT, int SEG, int CLS, int SEP, int STG, int AHD, int PSH, int RING>
static hipError_t walk_go(const WalkLaunch &L, LaunchInfo *info) {
    auto kern = pml_kernel_flatp<6, IdxT, CLS, SEP, SEG, STG, AHD, PSH, RING>;
    if (L.dyn_lds > 65536) {                               // every kernel that is handed more than 64 KiB of dynamic LDS must opt in first
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.dyn_lds);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(kern, L.grid, L.block, L.dyn_lds, L.stream, L.ix, L.bases, L.offs, L.n, L.out, L.err, L.stats, L.order, L.cls, L.seg);
    char name[96];                                         // the name as rocprofv3 prints it: every template argument, none dropped
    snprintf(name, sizeof(name), "pml_kernel_flatp<6, %s, %d, %d, %d, %d, %d, %d, %d>", sizeof(IdxT) == 4 ? "unsigned int" : "unsigned long",
             CLS, SEP, SEG, STG, AHD, PSH, RING);
    if (info) snprintf(info->kernel, sizeof(info->kernel), "%s", name);
    note_walk_launch(name);
    return hipGetLastError();
}
template <typename IdxT, int SEG, int CLS, int SEP>
static hipError_t walk_pick(const WalkLaunch &L, LaunchInfo *info) {
    if (!L.stg) return walk_go<IdxT, SEG, CLS, SEP, 0, 0, 0, 0>(L, info);
    if (L.ahd == 2) {                                      // deep rows: 32-bit row indexes, no pair-shared gathers (launch_pml sees to both)
        if constexpr (sizeof(IdxT) == 4) {
            if (L.ring == 2) {
                if constexpr (SEG == 0 && CLS == 0) return walk_go<IdxT, 0, 0, SEP, 1, 2, 0, 2>(L, info);
                else return hipErrorInvalidValue;
            }
            return L.ring ? walk_go<IdxT, SEG, CLS, SEP, 1, 2, 0, 1>(L, info) : walk_go<IdxT, SEG, CLS, SEP, 1, 2, 0, 0>(L, info);
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (L.ring == 2) {                                     // reset masks out (plain PML of whole reads: launch_pml sees to it)
        if constexpr (SEG == 0 && CLS == 0) {
            switch ((L.ahd ? 2 : 0) | (L.psh ? 1 : 0)) {
            case 0: return walk_go<IdxT, 0, 0, SEP, 1, 0, 0, 2>(L, info);
            case 1: return walk_go<IdxT, 0, 0, SEP, 1, 0, 1, 2>(L, info);
            case 2: return walk_go<IdxT, 0, 0, SEP, 1, 1, 0, 2>(L, info);
            default: return walk_go<IdxT, 0, 0, SEP, 1, 1, 1, 2>(L, info);
            }
        } else {
            return hipErrorInvalidValue;
        }
    }
    switch ((L.ahd ? 4 : 0) | (L.psh ? 2 : 0) | (L.ring ? 1 : 0)) {
    case 0: return walk_go<IdxT, SEG, CLS, SEP, 1, 0, 0, 0>(L, info);
    case 1: return walk_go<IdxT, SEG, CLS, SEP, 1, 0, 0, 1>(L, info);
    case 2: return walk_go<IdxT, SEG, CLS, SEP, 1, 0, 1, 0>(L, info);
    case 3: return walk_go<IdxT, SEG, CLS, SEP, 1, 0, 1, 1>(L, info);
    case 4: return walk_go<IdxT, SEG, CLS, SEP, 1, 1, 0, 0>(L, info);
    case 5: return walk_go<IdxT, SEG, CLS, SEP, 1, 1, 0, 1>(L, info);
    case 6: return walk_go<IdxT, SEG, CLS, SEP, 1, 1, 1, 0>(L, info);
    default: return walk_go<IdxT, SEG, CLS, SEP, 1, 1, 1, 1>(L, info);
    }
}
template <typename IdxT, int SEG>
static hipError_t walk_dispatch(const WalkLaunch &L, LaunchInfo *info) {
    if (SEG != 0 || L.cls_mode == 0) return L.sep ? walk_pick<IdxT, SEG, 0, 1>(L, info) : walk_pick<IdxT, SEG, 0, 0>(L, info);
    if (SEG == 0 && L.cls_mode == 1) return L.sep ? walk_pick<IdxT, 0, 1, 1>(L, info) : walk_pick<IdxT, 0, 1, 0>(L, info);
    if (SEG == 0) return L.sep ? walk_pick<IdxT, 0, 2, 1>(L, info) : walk_pick<IdxT, 0, 2, 0>(L, info);
    return hipErrorInvalidValue;
}

}  // namespace movi
