/*
 * synth.c -- deterministic synthetic move-structure indexes and reads (bench / test tooling).
 *
 * SURVEY.md section 8(d), recipe (ii) "direct move-table synthesis", O(r): draw a
 * run-length BWT directly (run characters, run lengths, one terminator run), derive
 * the LF mapping of every run head (`C[c] + rank_c`, the formula of
 * src/move_structure.cpp:515-523 in the reference), threshold bits that always point
 * at an existing run, the base intervals, blocked ids for mode 8, and write the
 * bytes of a v2 `index.movi` (layout of src/move_structure_io.cpp:435-469).  No text
 * is ever materialised.
 *
 * A uniformly random run sequence has no suffix-order correlation between
 * neighbouring runs: it is the cache-worst and reposition-heaviest case (about 0.9
 * fast-forwards and 0.25-0.4 scan rows per base against ~0.2 / ~0.24 on a real BWT),
 * so throughput measured on it is conservative.
 *
 * Reads are produced by walking LF from random (row, offset) starts and recording
 * the run characters, i.e. they are exact substrings of the (virtual) indexed text;
 * substitutions and `N`s are then applied.  Every read has its own counter-based
 * random stream, so the output does not depend on the thread count.
 *
 * This file feeds the measured path: it does not use anything from oracle/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int mode;
    uint64_t r, n, end_bwt_idx;
    uint64_t end_thr[4], counts[4];
    uint64_t first_runs[5], first_offsets[5], last_runs[5], last_offsets[5];
    uint16_t *lens;      /* r */
    uint8_t *code;       /* r */
    uint64_t *dest;      /* r: destination row of the run head */
    uint16_t *doff;      /* r: offset of the head inside the destination row */
    uint8_t *thr;        /* r: bit k = threshold bit k */
    /* mode 8 */
    uint32_t *blocked;   /* r */
    uint32_t *id_blocks; /* [4][n_blocks] */
    uint64_t n_blocks, block_size;
} synth_t;

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static const int alphamap_3[4][4] = {{3, 0, 1, 2}, {0, 3, 1, 2}, {0, 1, 3, 2}, {0, 1, 2, 3}};

void synth_free(synth_t *s) {
    if (!s) return;
    free(s->lens); free(s->code); free(s->dest); free(s->doff); free(s->thr);
    free(s->blocked); free(s->id_blocks); free(s);
}

static int blocked_ids(synth_t *s) {
    uint64_t block_size = 1ull << 20, max_allowed = (1ull << 22) - 1;   /* move_row_configs.hpp:102-103 */
    const uint64_t r = s->r;
    s->blocked = (uint32_t *)malloc(r * sizeof(uint32_t));
    if (!s->blocked) return -1;
    for (;;) {
        uint64_t nb = (r + block_size - 1) / block_size;
        uint32_t *ib = (uint32_t *)calloc(4 * nb, sizeof(uint32_t));
        if (!ib) return -1;
        uint64_t last[4] = {0, 0, 0, 0};
        int ok = 1;
        for (uint64_t i = 0; i < r && ok; i++) {
            if (i % block_size == 0)
                for (int a = 0; a < 4; a++) ib[a * nb + i / block_size] = (uint32_t)last[a];
            if (i == s->end_bwt_idx) { s->blocked[i] = 0; continue; }
            int c = s->code[i];
            uint64_t adj = s->dest[i] - s->first_runs[c + 1];
            uint64_t b = adj - ib[c * nb + i / block_size];
            if (b > max_allowed) { ok = 0; break; }
            s->blocked[i] = (uint32_t)b;
            last[c] = adj;
        }
        if (ok) { s->id_blocks = ib; s->n_blocks = nb; s->block_size = block_size; return 0; }
        free(ib);
        block_size /= 2;
        max_allowed = ((max_allowed + 1) / 2) - 1;
        if (block_size == 0) return -1;
    }
}

synth_t *synth_create(uint64_t r, int mode, uint64_t seed, double mean_run) {
    if (r < 16 || (mode != 6 && mode != 8) || mean_run < 1.0) return NULL;
    synth_t *s = (synth_t *)calloc(1, sizeof(synth_t));
    if (!s) return NULL;
    s->mode = mode; s->r = r;
    const uint32_t maxrun = mode == 6 ? 2047 : 1023;
    s->lens = (uint16_t *)malloc(r * 2);
    s->code = (uint8_t *)malloc(r);
    s->dest = (uint64_t *)malloc(r * 8);
    s->doff = (uint16_t *)malloc(r * 2);
    s->thr = (uint8_t *)calloc(r, 1);
    uint64_t *all_p = (uint64_t *)malloc((r + 1) * 8);
    if (!s->lens || !s->code || !s->dest || !s->doff || !s->thr || !all_p) { free(all_p); synth_free(s); return NULL; }
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 12345;
    s->end_bwt_idx = 1 + splitmix64(&st) % (r - 2);
    const double lg = log(1.0 - 1.0 / mean_run);
    int c = (int)(splitmix64(&st) & 3);
    uint64_t pos = 0;
    for (uint64_t i = 0; i < r; i++) {
        c = (c + 1 + (int)(splitmix64(&st) % 3)) & 3;            /* consecutive runs differ */
        uint32_t len = mean_run <= 1.0 ? 1 : 1 + (uint32_t)(log(1.0 - u01(&st)) / lg);
        if (len > maxrun) len = maxrun;
        if (i == s->end_bwt_idx) { len = 1; s->code[i] = 0; }      /* terminator row stores c == 0 */
        else { s->code[i] = (uint8_t)c; s->counts[c] += len; }
        s->lens[i] = (uint16_t)len;
        all_p[i] = pos;
        pos += len;
    }
    all_p[r] = pos;
    s->n = pos;
    /* LF of run heads; destinations are monotone per character -> four merge pointers */
    uint64_t C[4], rank[4] = {0, 0, 0, 0}, ptr[4] = {0, 0, 0, 0};
    C[0] = 1;
    for (int a = 1; a < 4; a++) C[a] = C[a - 1] + s->counts[a - 1];
    for (uint64_t i = 0; i < r; i++) {
        if (i == s->end_bwt_idx) { s->dest[i] = 0; s->doff[i] = 0; continue; }
        int a = s->code[i];
        uint64_t lf = C[a] + rank[a];
        rank[a] += s->lens[i];
        uint64_t p = ptr[a];
        while (all_p[p + 1] <= lf) p++;
        ptr[a] = p;
        s->dest[i] = p;
        s->doff[i] = (uint16_t)(lf - all_p[p]);
    }
    /* base intervals (what find_base_interval_data, move_structure_build.cpp:694-731, stores) */
    uint64_t cc = 1;
    for (int a = 0; a < 4; a++) {
        uint64_t lr = s->last_runs[a], lo = s->last_offsets[a];
        if (lo + 1 >= s->lens[lr]) { s->first_runs[a + 1] = lr + 1; s->first_offsets[a + 1] = 0; }
        else { s->first_runs[a + 1] = lr; s->first_offsets[a + 1] = lo + 1; }
        cc += s->counts[a];
        uint64_t lo_i = 0, hi_i = r;                              /* rows starting before cc */
        while (lo_i < hi_i) { uint64_t mid = (lo_i + hi_i) / 2; if (all_p[mid] < cc) lo_i = mid + 1; else hi_i = mid; }
        s->last_runs[a + 1] = lo_i - 1;
        s->last_offsets[a + 1] = cc - all_p[lo_i - 1] - 1;
    }
    free(all_p);
    /* threshold bits: in the gap between two consecutive j-rows the rows above a random
     * split go up (bit 1), the others down (bit 0); before the first j-row: down; after the
     * last: up.  The terminator row scans as an 'A' row (its c field is 0), as in the reference. */
    for (int j = 0; j < 4; j++) {
        uint64_t T = 0;                                          /* rows k < T go up; 0 = no j-row above: down */
        uint64_t i = 0;
        while (i < r) {
            /* find the next j-row at or after i */
            uint64_t nx = i;
            while (nx < r && s->code[nx] != j) nx++;
            if (nx == r) T = r;                                  /* no j-row below: up */
            for (uint64_t k = i; k < nx; k++) {
                int bit = k < T;
                if (k == s->end_bwt_idx) { if (j != 0) s->end_thr[j] = (uint64_t)bit; }
                else s->thr[k] |= (uint8_t)(bit << alphamap_3[s->code[k]][j]);
            }
            if (nx == r) break;
            /* draw the split of the gap that starts after this j-row */
            uint64_t nn = nx + 1;
            while (nn < r && s->code[nn] != j) nn++;
            if (nn < r) T = nx + 1 + (uint64_t)(u01(&st) * (double)(nn - nx));
            i = nx + 1;
        }
    }
    if (mode == 8 && blocked_ids(s) != 0) { synth_free(s); return NULL; }
    return s;
}

uint64_t synth_r(const synth_t *s) { return s->r; }
uint64_t synth_n(const synth_t *s) { return s->n; }
uint64_t synth_end_bwt_idx(const synth_t *s) { return s->end_bwt_idx; }

uint64_t synth_image_size(const synth_t *s) {
    uint64_t sz = 48 + 96 + 8 + 2048 + 8 + 4 + 3 + s->r * (s->mode == 6 ? 8 : 6) + 24 + 8 + 32 + 8 + 160;
    if (s->mode == 8) sz += 8 + 4 * s->n_blocks * 4 + 8;
    return sz;
}

static uint8_t *put64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); return p + 8; }

/* v2 index.movi: header (include/utils.hpp:32-61) | basic data | rows | empty overflow
 * tables | counts | base intervals | id blocks (mode 8). */
void synth_write_image(const synth_t *s, uint8_t *out) {
    uint8_t *p = out;
    memset(p, 0, 48);
    uint32_t magic = 0x4D4F5649u;
    memcpy(p, &magic, 4);
    p[4] = 2; p[5] = 0; p[6] = 0; p[7] = (uint8_t)s->mode; p[8] = 0;
    memcpy(p + 16, &s->n, 8); memcpy(p + 24, &s->r, 8); memcpy(p + 32, &s->r, 8); memcpy(p + 40, &s->end_bwt_idx, 8);
    p += 48;
    for (int i = 0; i < 4; i++) p = put64(p, s->end_thr[i]);
    for (int i = 0; i < 8; i++) p = put64(p, 0);
    p = put64(p, 256);
    for (int ch = 0; ch < 256; ch++) {
        uint64_t v = 256;
        if (ch == 'A') v = 0; else if (ch == 'C') v = 1; else if (ch == 'G') v = 2; else if (ch == 'T') v = 3;
        p = put64(p, v);
    }
    p = put64(p, 4);
    memcpy(p, "ACGT", 4); p += 4;
    p[0] = p[1] = p[2] = 0; p += 3;
    if (s->mode == 6) {
        for (uint64_t i = 0; i < s->r; i++) {
            uint64_t d = s->dest[i];
            uint32_t t = s->thr[i];
            uint16_t w[4];
            w[0] = (uint16_t)(d & 0xFFFF);
            w[1] = (uint16_t)((d >> 16) & 0xFFFF);
            w[2] = (uint16_t)(s->lens[i] | (((t >> 1) & 1) << 11) | (((t >> 2) & 1) << 12) | ((uint32_t)s->code[i] << 13));
            w[3] = (uint16_t)(s->doff[i] | ((t & 1) << 11) | ((uint32_t)(d >> 32) << 12));
            memcpy(p, w, 8); p += 8;
        }
    } else {
        for (uint64_t i = 0; i < s->r; i++) {
            uint32_t b = s->blocked[i], t = s->thr[i];
            uint16_t w[3];
            w[0] = (uint16_t)(b & 0xFFFF);
            w[1] = (uint16_t)(s->lens[i] | ((b >> 16) << 10));
            w[2] = (uint16_t)(s->doff[i] | ((uint32_t)s->code[i] << 10) | ((t & 1) << 13) | (((t >> 1) & 1) << 14) | (((t >> 2) & 1) << 15));
            memcpy(p, w, 6); p += 6;
        }
    }
    for (int i = 0; i < 3; i++) p = put64(p, 0);
    p = put64(p, 4);
    for (int i = 0; i < 4; i++) p = put64(p, s->counts[i]);
    p = put64(p, 5);
    for (int i = 0; i < 5; i++) p = put64(p, s->last_runs[i]);
    for (int i = 0; i < 5; i++) p = put64(p, s->last_offsets[i]);
    for (int i = 0; i < 5; i++) p = put64(p, s->first_runs[i]);
    for (int i = 0; i < 5; i++) p = put64(p, s->first_offsets[i]);
    if (s->mode == 8) {
        p = put64(p, s->n_blocks);
        memcpy(p, s->id_blocks, 4 * s->n_blocks * 4); p += 4 * s->n_blocks * 4;
        p = put64(p, s->block_size);
    }
}

/* Reads: offs has n_reads+1 entries into out (bytes). */
void synth_reads(const synth_t *s, uint64_t n_reads, const uint64_t *offs, uint64_t seed,
                 double sub_rate, double n_rate, uint8_t *out, int threads) {
    static const char alpha[4] = {'A', 'C', 'G', 'T'};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const uint64_t r1 = s->r - 1;
    #pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t st = (seed + 1) * 0xD1342543DE82EF95ull + i * 0x9E3779B97F4A7C15ull;
        uint64_t idx = splitmix64(&st) % s->r;
        uint64_t off = (uint64_t)(u01(&st) * (double)s->lens[idx]);
        const uint64_t len = offs[i + 1] - offs[i];
        uint8_t *R = out + offs[i];
        for (uint64_t k = 0; k < len; k++) {
            R[len - 1 - k] = (uint8_t)alpha[s->code[idx] & 3];       /* the walk goes last base -> first */
            uint64_t j = s->dest[idx];
            off += s->doff[idx];
            while (j < r1 && off >= s->lens[j]) { off -= s->lens[j]; j++; }
            idx = j;
        }
        for (uint64_t k = 0; k < len; k++) {
            double u = u01(&st);
            if (u < n_rate) R[k] = 'N';
            else if (u < n_rate + sub_rate) R[k] = (uint8_t)alpha[splitmix64(&st) & 3];
        }
    }
}
