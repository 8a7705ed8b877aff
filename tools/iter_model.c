// iter_model.c -- CPU model of the PML kernel's ITERATION count per base under different table layouts (design
// tooling, not product code and not the oracle): which layout cuts gathers per base on a given index + read set,
// before any of it is built in HIP.  It walks reads exactly like the kernel's automaton walks them
// (LF -> fast-forward -> match / reposition scan; /root/reference src/move_structure.cpp:59-87,524-545,
// src/move_structure_query.cpp:188-232,513-601) and charges one lane iteration per row window fetched:
//   * arrival at the LF target: the window (W aligned rows) that holds it; fast-forwards inside it are free,
//     every further window costs an iteration;
//   * mismatch: the step that sees it emits nothing; the scan then costs one iteration per window visited
//     (with the scan side-array, option A: ONE iteration whatever the distance, up to `sa_reach` rows);
//   * in-window repositions (inwin): a reposition whose target is a row of the window already held costs nothing;
//   * own = 1: only the gather's own row carries an entry (a step that ends on a neighbour cannot ride);
//   * look-ahead chains, depth S: after a base is resolved at row i, up to S following bases ride along while
//     they match c(j_s) and arrive without a fast-forward (entry of row i holds the chain j_1 .. j_S).
// Output: lane iterations per base, the share of iterations by kind, SIMT efficiency for waves of 64 consecutive
// reads (wave iterations = max over lanes), and the distributions the designs depend on.
//
// usage: iter_model <index.movi (mode 6)> <reads.bin> <read_len> [n_reads] [top_k]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t *rows;
static uint64_t r, end_row;
static uint64_t end_thr[4];

static inline uint32_t rn(uint64_t i) { uint16_t v; memcpy(&v, rows + 8 * i + 4, 2); return v & 0x7FF; }
static inline uint32_t rc(uint64_t i) { uint16_t v; memcpy(&v, rows + 8 * i + 4, 2); return v >> 13; }
static inline uint32_t roff(uint64_t i) { uint16_t v; memcpy(&v, rows + 8 * i + 6, 2); return v & 0x7FF; }
static inline uint64_t rid(uint64_t i) {
    uint32_t lo; uint16_t v; memcpy(&lo, rows + 8 * i, 4); memcpy(&v, rows + 8 * i + 6, 2);
    return (uint64_t)lo | ((uint64_t)(v >> 12) << 32);
}
static inline uint32_t rthr(uint64_t i, uint32_t k) {
    uint16_t nn, oo; memcpy(&nn, rows + 8 * i + 4, 2); memcpy(&oo, rows + 8 * i + 6, 2);
    return k == 0 ? (oo >> 11) & 1 : (nn >> (10 + k)) & 1;
}
static int code_of(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; }

typedef struct { int S, W, side, side_reach, inwin, own, rpl; } Design;   // rpl: rows per 128-byte line of the walked layout (round 6: fabric lines per base)
typedef struct {
    uint64_t iters, bases, it_arrive, it_ffwin, it_mism, it_scanwin, chained, waves_iters, lanes_iters;
    uint64_t ff_rows, scan_rows, repos;
    uint64_t lines;                                  // window fetches that touch another 128-byte line than the lane's previous fetch
} Tally;

// One read under one design; returns its lane iterations.
static uint32_t walk(const uint8_t *rd, uint32_t len, uint32_t top_k, const Design *d, Tally *t) {
    uint64_t idx = r - 1;
    uint32_t off = rn(idx) - 1, it = 0;
    const uint64_t W = (uint64_t)d->W, RPL = (uint64_t)(d->rpl ? d->rpl : 8);
    uint64_t last_line = ~0ull;
#define FETCH(row) do { const uint64_t ln_ = (uint64_t)(row) / RPL; if (ln_ != last_line) { t->lines++; last_line = ln_; } } while (0)
    int chain_left = 0;                              // bases that may still ride on the current entry
    for (uint32_t k = 0; k < len; k++) {
        const int a = code_of(rd[len - 1 - k]);
        int rode = 0, arr_ff = 0;
        if (k != 0) {                                // LF_move + fast_forward
            const uint64_t j = rid(idx);
            off += roff(idx);
            uint64_t jj = j;
            uint32_t ff = 0;
            while (jj < r - 1 && off >= rn(jj)) { off -= rn(jj); jj++; ff++; }
            t->ff_rows += ff; arr_ff = ff != 0;
            // does this base ride on the look-ahead chain? (match at j, no fast-forward)
            if (chain_left > 0 && ff == 0 && a >= 0 && (int)rc(j) == a && k >= top_k) { rode = 1; chain_left--; }
            else chain_left = 0;
            if (!rode && k >= top_k) {
                it++; t->it_arrive++;
                const uint64_t w0 = j / W, w1 = jj / W;
                it += (uint32_t)(w1 - w0); t->it_ffwin += w1 - w0;
                for (uint64_t w = w0; w <= w1; w++) FETCH(w * W);
            }
            idx = jj;
        } else if (top_k == 0) { it++; t->it_arrive++; FETCH(idx); }
        if (rode) { t->chained++; continue; }        // matched by construction
        if (a < 0) { chain_left = (k >= top_k && !(d->own && arr_ff)) ? d->S : 0; continue; }
        if ((int)rc(idx) == a) { chain_left = (k >= top_k && !(d->own && arr_ff)) ? d->S : 0; continue; }
        // reposition
        t->repos++;
        uint32_t down;
        if (idx == end_row) down = off >= end_thr[a];
        else {
            const uint32_t c = rc(idx);
            const uint32_t slot = ((uint32_t)a - (uint32_t)((uint32_t)a > c)) & 3u;
            down = off >= (rthr(idx, slot > 2 ? 2 : slot) ? rn(idx) : 0u);
        }
        const uint64_t from = idx;
        uint32_t sc = 0;
        if (down) { do { idx++; sc++; } while (idx < r - 1 && (int)rc(idx) != a); off = 0; }
        else { do { idx--; sc++; } while (idx > 0 && (int)rc(idx) != a); off = rn(idx) - 1; }
        t->scan_rows += sc;
        if (k >= top_k) {
            if (d->inwin && from / W == idx / W) {
                // the target sits in the window already held: resolved in the arrival iteration itself
            } else {
            it++; t->it_mism++;                      // the scan's first window (the mismatch step itself emitted nothing: it was the arrival iteration)
            // side = 2: the hint counts the rows BEYOND the window's edge (1 .. side_reach), so that its reach does not depend on where in the window the row sits
            const uint64_t edge = down ? (from / W) * W + (W - 1) : (from / W) * W;
            const uint64_t beyond = down ? idx - edge : edge - idx;
            if (d->side == 2 ? beyond <= (uint64_t)d->side_reach : (d->side && sc <= (uint32_t)d->side_reach)) {
                // side array told the distance: the one iteration above fetched the target's window directly
                FETCH((idx / W) * W);
            } else {
                const uint64_t first = down ? from + 1 : from - 1;
                const uint64_t wa = first / W, wb = idx / W;
                const uint64_t extra = wa > wb ? wa - wb : wb - wa;
                it += (uint32_t)extra; t->it_scanwin += extra;
                if (wa <= wb) { for (uint64_t w = wa; w <= wb; w++) FETCH(w * W); }
                else { for (uint64_t w = wa + 1; w-- > wb;) FETCH(w * W); }
            }
            }
        }
        chain_left = (k >= top_k && !d->own) ? d->S : 0;
    }
    return it;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: iter_model index.movi reads.bin read_len [n_reads] [top_k]\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *img = malloc(sz);
    if (fread(img, 1, sz, f) != (size_t)sz) return 1;
    fclose(f);
    if (img[7] != 6) { fprintf(stderr, "mode 6 only\n"); return 1; }
    memcpy(&r, img + 24, 8); memcpy(&end_row, img + 40, 8); memcpy(end_thr, img + 48, 32);
    uint64_t sigma; memcpy(&sigma, img + 48 + 96 + 8 + 2048, 8);
    rows = img + 48 + 96 + 8 + 2048 + 8 + sigma + 3;
    const uint32_t L = (uint32_t)atoi(argv[3]);
    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    fseek(f, 0, SEEK_END); long rsz = ftell(f); fseek(f, 0, SEEK_SET);
    uint64_t n_reads = (uint64_t)rsz / L;
    if (argc >= 5 && (uint64_t)atoll(argv[4]) < n_reads) n_reads = (uint64_t)atoll(argv[4]);
    const uint32_t top_k = argc >= 6 ? (uint32_t)atoi(argv[5]) : 12;
    uint8_t *reads = malloc(n_reads * L);
    if (fread(reads, 1, n_reads * L, f) != n_reads * L) return 1;
    fclose(f);
    const Design designs[] = {          // S, W, side, side_reach, inwin, own
        {0, 4, 0, 0, 0, 0}, {1, 4, 0, 0, 0, 0}, {1, 4, 0, 0, 0, 1}, {1, 4, 0, 0, 1, 0}, {2, 4, 0, 0, 0, 0}, {2, 4, 0, 0, 1, 0}, {3, 4, 0, 0, 1, 0},
        {7, 4, 0, 0, 1, 0}, {1, 4, 1, 64, 1, 0}, {2, 4, 1, 64, 1, 0}, {1, 2, 0, 0, 1, 0}, {1, 8, 0, 0, 1, 0}, {2, 8, 0, 0, 1, 0},
        {1, 4, 1, 7, 1, 0}, {1, 4, 1, 15, 1, 0}, {1, 4, 1, 31, 1, 0}, {1, 4, 1, 63, 1, 0}, {1, 4, 2, 7, 1, 0}, {1, 4, 2, 15, 1, 0},  // round 5: reposition hints of 3 .. 6 bits in the rows' spare bits
        // round 6: entries TWO rows deep packed to 21.33 bytes per row (6 rows per 128-byte line, windows of 3 rows = half a line), with / without hints;
        // beside them what ships (S = 1, W = 4, 8 rows per line) and round 4's chain rows (S = 2, W = 4, 4 rows per line) with hints
        {1, 4, 2, 7, 1, 0, 8}, {2, 4, 2, 7, 1, 0, 4}, {2, 3, 0, 0, 1, 0, 6}, {2, 3, 2, 7, 1, 0, 6}, {2, 3, 2, 3, 1, 0, 6}, {2, 6, 2, 7, 1, 0, 6}, {3, 3, 2, 7, 1, 0, 6},
    };
    printf("r = %llu rows, %llu reads x %u, top-of-walk K = %u\n", (unsigned long long)r, (unsigned long long)n_reads, L, top_k);
    printf("%-40s %9s %9s | %7s %7s %7s %7s | %7s | %6s %6s %6s | %7s\n", "design", "iter/base", "SIMT", "arrive", "ff-win", "mismat", "scanwin",
           "chained", "ff/b", "scan/b", "repo/b", "lines/b");
    for (size_t di = 0; di < sizeof designs / sizeof designs[0]; di++) {
        Tally t; memset(&t, 0, sizeof t);
        uint32_t wave_max = 0;
        FILE *df = NULL;
        if (getenv("ITER_DUMP")) { char fn[256]; snprintf(fn, sizeof fn, "%s.%zu", getenv("ITER_DUMP"), di); df = fopen(fn, "wb"); }
        for (uint64_t i = 0; i < n_reads; i++) {
            const uint32_t it = walk(reads + i * L, L, top_k, &designs[di], &t);
            t.iters += it; t.bases += L;
            if (df) fwrite(&it, 4, 1, df);
            if (it > wave_max) wave_max = it;
            if ((i & 63) == 63 || i + 1 == n_reads) { t.waves_iters += (uint64_t)wave_max * 64; wave_max = 0; }
        }
        if (df) fclose(df);
        char name[96];
        snprintf(name, sizeof name, "S=%d W=%d side=%d(%d) inwin=%d own=%d rpl=%d", designs[di].S, designs[di].W, designs[di].side, designs[di].side_reach, designs[di].inwin, designs[di].own, designs[di].rpl ? designs[di].rpl : 8);
        const double b = (double)t.bases;
        printf("%-40s %9.4f %9.3f | %7.4f %7.4f %7.4f %7.4f | %7.4f | %6.3f %6.3f %6.4f | %7.4f\n", name, t.iters / b, (double)t.iters / t.waves_iters,
               t.it_arrive / b, t.it_ffwin / b, t.it_mism / b, t.it_scanwin / b, t.chained / b, t.ff_rows / b, t.scan_rows / b, t.repos / b, t.lines / b);
    }
    return 0;
}
