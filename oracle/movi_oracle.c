/*
 * movi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar CPU restatement of the reference's PML / count query path
 * (mohsenzakeri/Movi, modes 6 "regular-thresholds" and 8 "blocked-thresholds").
 * It exists only so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check (or time) the HIP path against an independent
 * statement of the same algorithm.  Nothing under movi_amd/ may import, link or
 * call it.
 *
 * Parity pin: this restatement is checked in tests/test_oracle_golden.py against
 * the golden vectors the reference's own tests hold for this path
 * (tests_data/sample.fastq.pmls.sorted on an index of tests_data/ref.fasta,
 * tests/test_pml.cpp:89-105; index-size known answers 948119 B / 711733 B,
 * tests/test_build.cpp:37,53; MoveQuery u16 clamp tests/test_basics.cpp:304-315).
 * The reference itself is NOT buildable in this image (needs sdsl-lite and
 * hclust-cpp headers, fetched by its CMake from the network), so there is no
 * oracle/_ref binary; see DESIGN.md.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * follows.  Rows are decoded straight from the on-disk bytes of index.movi.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK            0
#define ORACLE_ERR_FORMAT   -1   /* not a v2 index.movi of mode 6/8            */
#define ORACLE_ERR_INVARIANT -2  /* a "this should not happen" throw in the ref */

typedef struct {
    uint8_t  mode;                 /* header.type: 6 or 8 (include/utils.hpp:37) */
    uint64_t row_bytes;            /* 8 or 6 (include/move_row.hpp:112-117)       */
    uint64_t length, r, original_r, end_bwt_idx;
    uint64_t end_thr[4];           /* end_bwt_idx_thresholds (move_structure.hpp:340) */
    uint64_t alphamap[256];
    uint64_t alphabet_size;
    uint8_t  alphabet[16];
    uint8_t *rows;                 /* r * row_bytes, owned copy                   */
    uint64_t counts_size;
    uint64_t counts[16];
    uint64_t k;                    /* alphabet_size + 1                           */
    uint64_t last_runs[16], last_offsets[16], first_runs[16], first_offsets[16];
    uint64_t nblocks;              /* mode 8 only                                 */
    uint32_t *id_blocks;           /* [alphabet_size][nblocks]                    */
    uint64_t block_size;
    /* sampled ("tally") mode 7: ids kept only every tally_checkpoints rows, move_structure.hpp:361-363 */
    uint32_t tally_checkpoints;
    uint64_t tally_len;
    uint8_t *tally_ids;            /* [alphabet_size][tally_len] x 5-byte MoveTally (move_row.hpp:13-40) */
    /* separators (movi build --separators): alphabet = '%' + ACGT; explicit thresholds of the
     * separator rows, move_structure.hpp:344-346, file layout move_structure_io.cpp:399-433 */
    int      sep;                  /* MoveStructure::use_separator, move_structure.cpp:547-552 */
    uint64_t n_sep;                /* separators_thresholds_map entries, sorted by row */
    uint64_t *sep_rows;
    uint16_t (*sep_vals)[4];
} oracle_index;

#define ORACLE_SEPARATOR '%'       /* include/commons.hpp:63 */

/* alphamap_3: src/utils.cpp:5-8 */
static const uint32_t alphamap_3[4][4] = {{3, 0, 1, 2},
                                          {0, 3, 1, 2},
                                          {0, 1, 3, 2},
                                          {0, 1, 2, 3}};

/* ---------------------------------------------------------------- index file */

static int rd(const uint8_t *buf, size_t n, size_t *pos, void *dst, size_t len) {
    if (*pos + len > n) return -1;
    memcpy(dst, buf + *pos, len);
    *pos += len;
    return 0;
}

/* MoveStructure::deserialize, src/move_structure_io.cpp:471-511:
 * header (:66-109, struct include/utils.hpp:32-61) | basic data (:145-184) |
 * main table (:361-397) | overflow tables (:219-254) | counts + base intervals
 * (:269-287) | id blocks for blocked modes (:305-324). */
oracle_index *oracle_open(const uint8_t *buf, size_t n) {
    oracle_index *ix = (oracle_index *)calloc(1, sizeof(oracle_index));
    size_t p = 0;
    uint8_t hdr[48];
    if (!ix || rd(buf, n, &p, hdr, 48)) goto bad;
    uint32_t magic; memcpy(&magic, hdr, 4);
    if (magic != 0x4D4F5649u) goto bad;                 /* MOVI_MAGIC utils.hpp:29 */
    ix->mode = hdr[7];
    if (ix->mode != 6 && ix->mode != 8 && ix->mode != 7 && ix->mode != 5 && ix->mode != 3 && ix->mode != 2) goto bad;
    ix->row_bytes = (ix->mode == 6 || ix->mode == 3) ? 8 : ((ix->mode == 8 || ix->mode == 2) ? 6 : 3);   /* MoveRow::row_size, move_row.hpp:104-120 */
    memcpy(&ix->length, hdr + 16, 8);
    memcpy(&ix->r, hdr + 24, 8);
    memcpy(&ix->original_r, hdr + 32, 8);
    memcpy(&ix->end_bwt_idx, hdr + 40, 8);
    uint64_t skip[8];
    if (rd(buf, n, &p, ix->end_thr, 32)) goto bad;      /* end_bwt_idx_thresholds */
    if (rd(buf, n, &p, skip, 64)) goto bad;             /* next_down, next_up     */
    uint64_t amap_size;
    if (rd(buf, n, &p, &amap_size, 8) || amap_size != 256) goto bad;
    if (rd(buf, n, &p, ix->alphamap, 256 * 8)) goto bad;
    if (rd(buf, n, &p, &ix->alphabet_size, 8) || ix->alphabet_size > 8) goto bad;
    if (rd(buf, n, &p, ix->alphabet, ix->alphabet_size)) goto bad;
    uint8_t flags[3];                                   /* u16 nt_splitting, bool constant (move_structure.hpp:322-323) */
    if (rd(buf, n, &p, flags, 3)) goto bad;
    ix->rows = (uint8_t *)malloc(ix->r * ix->row_bytes + 16);
    if (!ix->rows || rd(buf, n, &p, ix->rows, ix->r * ix->row_bytes)) goto bad;
    if (ix->mode == 7 || ix->mode == 5) {               /* read_tally_table, io.cpp:338-349 */
        if (rd(buf, n, &p, &ix->tally_checkpoints, 4) || ix->tally_checkpoints == 0) goto bad;
        if (rd(buf, n, &p, &ix->tally_len, 8) || ix->tally_len > (n - p) / 5) goto bad;
        size_t bytes = ix->alphabet_size * ix->tally_len * 5;
        ix->tally_ids = (uint8_t *)malloc(bytes + 8);
        if (!ix->tally_ids || rd(buf, n, &p, ix->tally_ids, bytes)) goto bad;
    }
    for (int t = 0; t < 3; t++) {                       /* overflow tables: empty  */
        uint64_t sz;
        if (rd(buf, n, &p, &sz, 8)) goto bad;
        uint64_t per = (t == 2) ? (ix->alphabet_size - 1) * 8 : 8;
        if (p + sz * per > n) goto bad;
        p += sz * per;
    }
    if (rd(buf, n, &p, &ix->counts_size, 8) || ix->counts_size > 16) goto bad;
    if (rd(buf, n, &p, ix->counts, ix->counts_size * 8)) goto bad;
    if (rd(buf, n, &p, &ix->k, 8) || ix->k > 16) goto bad;
    if (rd(buf, n, &p, ix->last_runs, ix->k * 8)) goto bad;
    if (rd(buf, n, &p, ix->last_offsets, ix->k * 8)) goto bad;
    if (rd(buf, n, &p, ix->first_runs, ix->k * 8)) goto bad;
    if (rd(buf, n, &p, ix->first_offsets, ix->k * 8)) goto bad;
    if (ix->mode == 8 || ix->mode == 2) {
        if (rd(buf, n, &p, &ix->nblocks, 8)) goto bad;
        if (ix->nblocks) {
            size_t bytes = ix->alphabet_size * ix->nblocks * 4;
            ix->id_blocks = (uint32_t *)malloc(bytes);
            if (!ix->id_blocks || rd(buf, n, &p, ix->id_blocks, bytes)) goto bad;
        }
        ix->block_size = ix->mode == 8 ? 1048576 : 4194304;   /* BLOCK_SIZE move_row_configs.hpp:102 / :73 */
        if (p + 8 <= n) rd(buf, n, &p, &ix->block_size, 8);   /* io.cpp:321-323 */
    }
    ix->sep = ix->alphabet_size == 5 && ix->alphabet[0] == ORACLE_SEPARATOR;
    if (ix->sep && ix->mode != 5 && ix->mode != 3 && ix->mode != 2) {   /* read_separators_thresholds, io.cpp:415-433 (USE_THRESHOLDS only) */
        uint64_t nt, nm;
        if (rd(buf, n, &p, &nt, 8) || nt > (n - p) / 8) goto bad;
        uint16_t (*vals)[4] = (uint16_t (*)[4])malloc((nt ? nt : 1) * 8);
        if (!vals || rd(buf, n, &p, vals, nt * 8)) { free(vals); goto bad; }
        if (rd(buf, n, &p, &nm, 8) || nm > (n - p) / 16) { free(vals); goto bad; }
        ix->sep_rows = (uint64_t *)malloc((nm ? nm : 1) * 8);
        ix->sep_vals = (uint16_t (*)[4])malloc((nm ? nm : 1) * 8);
        for (uint64_t e = 0; e < nm; e++) {             /* the map, kept sorted by row (insertion sort: few entries) */
            uint64_t kv[2];
            if (rd(buf, n, &p, kv, 16) || kv[1] >= nt) { free(vals); goto bad; }
            uint64_t at = e;
            while (at > 0 && ix->sep_rows[at - 1] > kv[0]) {
                ix->sep_rows[at] = ix->sep_rows[at - 1];
                memcpy(ix->sep_vals[at], ix->sep_vals[at - 1], 8);
                at--;
            }
            ix->sep_rows[at] = kv[0];
            memcpy(ix->sep_vals[at], vals[kv[1]], 8);
        }
        ix->n_sep = nm;
        free(vals);
    }
    return ix;
bad:
    if (ix) { free(ix->rows); free(ix->id_blocks); free(ix->tally_ids); free(ix->sep_rows); free(ix->sep_vals); free(ix); }
    return NULL;
}

void oracle_close(oracle_index *ix) {
    if (!ix) return;
    free(ix->rows); free(ix->id_blocks); free(ix->tally_ids); free(ix->sep_rows); free(ix->sep_vals); free(ix);
}

uint64_t oracle_r(const oracle_index *ix) { return ix->r; }
uint64_t oracle_length(const oracle_index *ix) { return ix->length; }
uint64_t oracle_end_bwt_idx(const oracle_index *ix) { return ix->end_bwt_idx; }
int      oracle_mode(const oracle_index *ix) { return ix->mode; }

/* ------------------------------------------------------------------ row decode */

static inline void row16(const oracle_index *ix, uint64_t i, uint16_t w[4]) {
    memcpy(w, ix->rows + i * ix->row_bytes, ix->row_bytes);
}

/* MoveRow::get_n, include/move_row.hpp:245-248 (mode 6, 11 bits) / :287-290 (mode 8, 10 bits) */
static inline uint64_t get_n(const oracle_index *ix, uint64_t i) {
    if (ix->mode == 7) {                                /* tally rows, move_row.hpp:209-212: n | (c bit 1) << 8 */
        const uint8_t *b = ix->rows + i * 3;
        return (uint64_t)b[0] | ((uint64_t)((b[2] >> 1) & 1) << 8);
    }
    if (ix->mode == 5) {                                /* sampled, no thresholds: SHIFT_N 2, 2 bits (configs :107-118) */
        const uint8_t *b = ix->rows + i * 3;
        return (uint64_t)b[0] | ((uint64_t)((b[2] >> 2) & 3) << 8);
    }
    uint16_t w[4]; row16(ix, i, w);
    if (ix->mode == 3) return w[2] & 0xFFF;             /* regular, no thresholds: LENGTH_BITS 12 (configs :21-32) */
    return ix->mode == 6 ? (w[2] & 0x7FF) : (w[1] & 0x3FF);             /* modes 8 and 2: 10 bits */
}
/* MoveRow::get_offset, move_row.hpp:250-253 / :292-295 */
static inline uint64_t get_offset(const oracle_index *ix, uint64_t i) {
    if (ix->mode == 7) {                                /* move_row.hpp:214-217: offset | (c bit 0) << 8 */
        const uint8_t *b = ix->rows + i * 3;
        return (uint64_t)b[1] | ((uint64_t)(b[2] & 1) << 8);
    }
    if (ix->mode == 5) {                                /* SHIFT_OFFSET 0, 2 bits */
        const uint8_t *b = ix->rows + i * 3;
        return (uint64_t)b[1] | ((uint64_t)(b[2] & 3) << 8);
    }
    uint16_t w[4]; row16(ix, i, w);
    if (ix->mode == 3) return w[3] & 0xFFF;
    return ix->mode == 6 ? (w[3] & 0x7FF) : (w[2] & 0x3FF);
}
/* MoveRow::get_c, move_row.hpp:255-257 (n >> 13) / :297-299 ((offset >> 10) & 7) */
static inline uint32_t get_c(const oracle_index *ix, uint64_t i) {
    if (ix->mode == 7) return (ix->rows[i * 3 + 2] >> 2) & 7;          /* move_row.hpp:219-221, SHIFT_C 2 */
    if (ix->mode == 5) return (ix->rows[i * 3 + 2] >> 4) & 15;         /* SHIFT_C 4, 4 bits */
    uint16_t w[4]; row16(ix, i, w);
    return (ix->mode == 6 || ix->mode == 3) ? (uint32_t)(w[2] >> 13) : (uint32_t)((w[2] >> 10) & 7);
}
/* MoveRow::get_threshold, move_row.hpp:304-317 (mode 6) / :319-332 (mode 8) */
static inline uint32_t get_threshold_bit(const oracle_index *ix, uint64_t i, uint32_t k) {
    if (ix->mode == 7) return (ix->rows[i * 3 + 2] >> (5 + k)) & 1;    /* move_row.hpp:334-347 */
    uint16_t w[4]; row16(ix, i, w);
    if (ix->mode == 6) {
        switch (k) {
            case 0: return (w[3] >> 11) & 1;
            case 1: return (w[2] >> 11) & 1;
            default: return (w[2] >> 12) & 1;
        }
    }
    return (w[2] >> (13 + k)) & 1;
}
/* MoveStructure::get_char, src/move_structure.cpp:288-293 */
static inline int get_char(const oracle_index *ix, uint64_t i) {
    if (i == ix->end_bwt_idx) return '$';
    return ix->alphabet[get_c(ix, i)];
}
/* MoveTally::get, include/move_row.hpp:28-38 */
static inline uint64_t tally_get(const oracle_index *ix, uint32_t c, uint64_t k) {
    const uint8_t *b = ix->tally_ids + ((uint64_t)c * ix->tally_len + k) * 5;
    uint32_t right; memcpy(&right, b, 4);
    return (uint64_t)right | ((uint64_t)b[4] << 32);
}
/* MoveStructure::get_id for the sampled modes, src/move_structure.cpp:104-283 (the forward branch: the
 * reference fixes forward_direciton = true, :146).  The id of row idx is recovered from the id stored at the
 * next checkpoint for idx's character: count the BWT positions of that character between idx and that
 * stored run, then walk the destination rows backwards by that many positions. */
static uint64_t get_id_tally(const oracle_index *ix, uint64_t idx) {
    if (idx == ix->end_bwt_idx) return 0;                              /* :106-108 */
    uint32_t ci = get_c(ix, idx);
    uint64_t cp = ix->tally_checkpoints, ta = idx / cp;
    if (idx == ix->r - 1) return tally_get(ix, ci, ix->tally_len - 1); /* :114-117 */
    if (idx % cp == 0) return tally_get(ix, ci, ta);                   /* :121-124 */
    uint64_t tb = ta + 1, next_cp = tb * cp;
    if (next_cp >= ix->r) next_cp = ix->r - 1;                         /* :137-139 */
    uint64_t id = tally_get(ix, ci, tb), rows_until = 0, last_id = ix->r;
    for (uint64_t i = idx; i < next_cp; i++)                           /* :168-174 */
        if (get_char(ix, i) == get_char(ix, idx)) { rows_until += get_n(ix, i); last_id = i; }
    if (last_id == idx && get_char(ix, idx) != get_char(ix, next_cp)) return id;   /* :178-180 */
    if (last_id == ix->r) return ix->r;                                /* :181-183 throws */
    uint64_t offset = get_offset(ix, next_cp);
    if (get_char(ix, idx) != get_char(ix, next_cp)) {                  /* :194-197 */
        rows_until -= get_n(ix, last_id);
        offset = get_offset(ix, last_id);
    }
    if (id >= ix->r || offset >= get_n(ix, id)) return ix->r;          /* :200-203 throws */
    if (offset >= rows_until) return id;                               /* :204-209 */
    rows_until -= offset + 1;
    id -= 1;
    while (rows_until != 0) {                                          /* :211-219 */
        if (rows_until >= get_n(ix, id)) { rows_until -= get_n(ix, id); id -= 1; }
        else rows_until = 0;
    }
    return id;
}
/* MoveStructure::get_id, src/move_structure.cpp:91-102 with MoveRow::get_id
 * move_row.hpp:232-243 (mode 6: id32 | (offset>>12)<<32) / :267-285 (mode 8: id16 | (n>>10)<<16) */
static inline uint64_t get_id(const oracle_index *ix, uint64_t i) {
    if (ix->mode == 7 || ix->mode == 5) return get_id_tally(ix, i);
    uint16_t w[4]; row16(ix, i, w);
    if (ix->mode == 6 || ix->mode == 3) {
        uint64_t id = (uint64_t)w[0] | ((uint64_t)w[1] << 16);
        return id | ((uint64_t)(w[3] >> 12) << 32);
    }
    uint64_t bid = (uint64_t)w[0] | ((uint64_t)(w[1] >> 10) << 16);
    if (ix->mode == 2) bid |= (uint64_t)(w[2] >> 14) << 22;            /* move_row.hpp:274-280: two more id bits in `offset` */
    if (i == ix->end_bwt_idx) return bid;
    uint32_t c = (w[2] >> 10) & 7;
    return bid + (uint64_t)ix->id_blocks[c * ix->nblocks + i / ix->block_size] + ix->first_runs[c + 1];
}
/* MoveStructure::get_thresholds, src/move_structure.cpp:305-309 */
static inline uint64_t get_thresholds(const oracle_index *ix, uint64_t i, uint32_t k) {
    return get_threshold_bit(ix, i, k) == 0 ? 0 : get_n(ix, i);
}
/* MoveStructure::check_alphabet, src/move_structure.cpp:383-397 with
 * ignore_illegal_chars == 0.  Bytes >= 128 index out of the
 * 256-entry alphamap in the reference (char sign extension); treated as illegal. */
static inline int check_alphabet(const oracle_index *ix, uint8_t c) {
    if (c >= 128) return 0;
    if (ix->sep && c == ORACLE_SEPARATOR) return 0;                    /* :384-388 */
    return ix->alphamap[c] != 256;
}

/* ------------------------------------------------------------------- LF + ff */

/* MoveStructure::fast_forward, src/move_structure.cpp:524-545 */
static inline uint64_t fast_forward(const oracle_index *ix, uint64_t *offset, uint64_t idx) {
    uint64_t idx_ = idx;
    while (idx < ix->r - 1 && *offset >= get_n(ix, idx)) {
        *offset -= get_n(ix, idx);
        idx += 1;
    }
    return idx - idx_;
}

/* MoveStructure::LF_move, src/move_structure.cpp:59-87. Returns ff count or <0. */
static inline int64_t LF_move(const oracle_index *ix, uint64_t *offset, uint64_t *i) {
    uint64_t idx = get_id(ix, *i);
    if (idx >= ix->r) return ORACLE_ERR_INVARIANT;                     /* :63-65 */
    *offset = get_offset(ix, *i) + *offset;
    uint64_t ff = 0;
    if (idx < ix->r - 1 && *offset >= get_n(ix, idx)) {
        ff = fast_forward(ix, offset, idx);
        idx += ff;
        if (ff >= 65535) return ORACLE_ERR_INVARIANT;                  /* :72-75 */
    }
    *i = idx;
    return (int64_t)ff;
}

/* ------------------------------------------------------------- repositioning */

/* MoveStructure::reposition_up, src/move_structure_query.cpp:188-209 */
static uint64_t reposition_up(const oracle_index *ix, uint64_t idx, uint8_t c, uint64_t *scan) {
    if (idx == 0) return ix->r;
    uint8_t row_c = ix->alphabet[get_c(ix, idx)];
    while (idx > 0 && row_c != c) {
        *scan += 1;
        idx -= 1;
        row_c = ix->alphabet[get_c(ix, idx)];
    }
    return row_c == c ? idx : ix->r;
}
/* MoveStructure::reposition_down, src/move_structure_query.cpp:211-232 */
static uint64_t reposition_down(const oracle_index *ix, uint64_t idx, uint8_t c, uint64_t *scan) {
    if (idx == ix->r - 1) return ix->r;
    uint8_t row_c = ix->alphabet[get_c(ix, idx)];
    while (idx < ix->r - 1 && row_c != c) {
        *scan += 1;
        idx += 1;
        row_c = ix->alphabet[get_c(ix, idx)];
    }
    return row_c == c ? idx : ix->r;
}

/* separators_thresholds[separators_thresholds_map[idx]] (move_structure_query.cpp:541); an unordered_map
 * operator[] on a missing key would insert entry 0 -- a row that is a separator always has its key. */
static const uint16_t *sep_lookup(const oracle_index *ix, uint64_t idx) {
    uint64_t lo = 0, hi = ix->n_sep;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (ix->sep_rows[mid] < idx) lo = mid + 1; else hi = mid;
    }
    return (lo < ix->n_sep && ix->sep_rows[lo] == idx) ? ix->sep_vals[lo] : NULL;
}

/* MoveStructure::reposition_thresholds, src/move_structure_query.cpp:513-601.
 * Returns 1 = up, 0 = down, <0 = invariant broken. */
static int reposition_thresholds(const oracle_index *ix, uint64_t *idx, uint64_t offset,
                                 uint8_t r_char, uint64_t *scan) {
    uint64_t saved_idx = *idx;
    uint64_t alphabet_index = ix->alphamap[r_char];
    if (ix->sep) {                                                      /* :518-523 */
        if (alphabet_index == 0) return ORACLE_ERR_INVARIANT;
        alphabet_index -= 1;
    }
    *scan = 0;
    uint8_t rlbwt_char = ix->alphabet[get_c(ix, *idx)];
    uint64_t threshold_value;
    if (*idx == ix->end_bwt_idx) {
        threshold_value = ix->end_thr[alphabet_index];                 /* :534-535 */
    } else if (ix->sep && rlbwt_char == ORACLE_SEPARATOR) {             /* :540-541 */
        const uint16_t *v = sep_lookup(ix, *idx);
        if (!v) return ORACLE_ERR_INVARIANT;
        threshold_value = v[alphabet_index];
    } else {
        alphabet_index = alphamap_3[ix->alphamap[rlbwt_char] - (ix->sep ? 1 : 0)][alphabet_index];   /* :545-556 */
        if (alphabet_index == 3) return ORACLE_ERR_INVARIANT;          /* :559-561 */
        threshold_value = get_thresholds(ix, *idx, (uint32_t)alphabet_index);
    }
    if (offset >= threshold_value) {                                    /* :575 */
        *idx = reposition_down(ix, saved_idx, r_char, scan);
        if (*idx >= ix->r || r_char != ix->alphabet[get_c(ix, *idx)]) return ORACLE_ERR_INVARIANT;
        return 0;
    } else {
        *idx = reposition_up(ix, saved_idx, r_char, scan);
        if (*idx >= ix->r || r_char != ix->alphabet[get_c(ix, *idx)]) return ORACLE_ERR_INVARIANT;
        return 1;
    }
}

/* ------------------------------------------------------------------------ PML */

typedef struct {            /* Strand, include/read_processor.hpp:9-53 (PML fields) */
    const uint8_t *R;
    int64_t  len;
    int64_t  pos_on_r;
    uint64_t idx, offset, match_len;
    uint16_t *out;          /* MoveQuery::matching_lens: emission order = last base first */
    uint64_t emitted;
} strand_t;

/* ReadProcessor::reset_process, src/read_processor.cpp:65-96 (start row :69-70,
 * identical to query_pml src/move_structure_query.cpp:237-238) */
static inline void strand_reset(const oracle_index *ix, strand_t *s, const uint8_t *R,
                                int64_t len, uint16_t *out) {
    s->R = R; s->len = len; s->pos_on_r = len - 1;
    s->idx = ix->r - 1;
    s->offset = get_n(ix, s->idx) - 1;
    s->match_len = 0; s->out = out; s->emitted = 0;
}

/* ReadProcessor::process_char, src/read_processor.cpp:99-256 (core :100-104,
 * :188-238); same per-base step as MoveStructure::query_pml
 * src/move_structure_query.cpp:266-361 with the LF taken before the step. */
static inline int process_char(const oracle_index *ix, strand_t *s, uint64_t *ff_tot, uint64_t *scan_tot) {
    if (s->pos_on_r < s->len - 1) {
        int64_t ff = LF_move(ix, &s->offset, &s->idx);
        if (ff < 0) return (int)ff;
        *ff_tot += (uint64_t)ff;
    }
    uint8_t row_c = ix->alphabet[get_c(ix, s->idx)];    /* the '$' row decodes as c==0 */
    uint8_t ch = s->R[s->pos_on_r];
    uint64_t scan = 0;
    if (!check_alphabet(ix, ch)) {
        s->match_len = 0;
    } else if (row_c == ch) {
        s->match_len += 1;
    } else {
        int up = reposition_thresholds(ix, &s->idx, s->offset, ch, &scan);
        if (up < 0) return up;
        s->match_len = 0;
        s->offset = up ? get_n(ix, s->idx) - 1 : 0;     /* read_processor.cpp:223 */
    }
    *scan_tot += scan;
    /* MoveQuery::add_ml, include/move_query.hpp:26-38: u16 clamp */
    s->out[s->emitted++] = s->match_len > 65535 ? 65535 : (uint16_t)s->match_len;
    s->pos_on_r -= 1;
    return ORACLE_OK;
}

/* One read, the --no-prefetch answer (MoveStructure::query_pml).  out gets len
 * values in emission order (last base first), as written to the BPF file. */
int oracle_pml(const oracle_index *ix, const uint8_t *R, int64_t len, uint16_t *out,
               uint64_t *ff_tot, uint64_t *scan_tot) {
    strand_t s;
    uint64_t ff = 0, sc = 0;
    /* an index without thresholds repositions RANDOMLY in the reference (reposition_randomly,
     * src/move_structure_query.cpp:603-): its PMLs are not reproducible, so there is nothing to restate */
    if (ix->mode == 5 || ix->mode == 3 || ix->mode == 2) return ORACLE_ERR_FORMAT;
    if (len <= 0) { if (ff_tot) *ff_tot = 0; if (scan_tot) *scan_tot = 0; return ORACLE_OK; }
    strand_reset(ix, &s, R, len, out);
    while (s.pos_on_r > -1) {
        int rc = process_char(ix, &s, &ff, &sc);
        if (rc < 0) return rc;
    }
    if (ff_tot) *ff_tot = ff;
    if (scan_tot) *scan_tot = sc;
    return ORACLE_OK;
}

/* `movi query --logs`, prefetch mode: what MoveQuery::add_fastforward / add_scan collect for one read
 * (ReadProcessor::process_char src/read_processor.cpp:99-121: after the LF into a base, the LF's fast-forward count and the
 * scan count of the base BEFORE it; ReadProcessor::write_mls :586-596: once more when the read is written out -- the last
 * LF's count again, and the last base's scans).  So, in emission order: scans[k] = rows base k's reposition scanned,
 * fastforwards[k] = fast-forwards of the LF from base k to base k + 1, the last entry repeating the one before it
 * (0 for a read of one base: the reference leaves the strand's previous value there).  Values are truncated to u16 like
 * MoveQuery's vectors (include/move_query.hpp:84-85). */
int oracle_pml_logs(const oracle_index *ix, const uint8_t *R, int64_t len, uint16_t *out, uint16_t *ff_out, uint16_t *scan_out) {
    strand_t s;
    uint64_t ff = 0, sc = 0;
    if (ix->mode == 5 || ix->mode == 3 || ix->mode == 2) return ORACLE_ERR_FORMAT;
    if (len <= 0) return ORACLE_OK;
    strand_reset(ix, &s, R, len, out);
    int64_t k = 0;
    ff_out[0] = 0;
    while (s.pos_on_r > -1) {
        const uint64_t ff0 = ff, sc0 = sc;
        int rc = process_char(ix, &s, &ff, &sc);
        if (rc < 0) return rc;
        if (k > 0) ff_out[k - 1] = (uint16_t)(ff - ff0);
        if (k > 0 && k == len - 1) ff_out[k] = (uint16_t)(ff - ff0);
        scan_out[k] = (uint16_t)(sc - sc0);
        k++;
    }
    return ORACLE_OK;
}

/* oracle_pml that also records the walker's position (row, offset) after every base: for studies of
 * how quickly walks started at different places of a read fall into step (tools/sync_study.py). */
int oracle_pml_trace(const oracle_index *ix, const uint8_t *R, int64_t len, uint16_t *out,
                     uint64_t *idx_out, uint32_t *off_out) {
    strand_t s;
    uint64_t ff = 0, sc = 0;
    if (ix->mode == 5 || ix->mode == 3 || ix->mode == 2) return ORACLE_ERR_FORMAT;
    if (len <= 0) return ORACLE_OK;
    strand_reset(ix, &s, R, len, out);
    while (s.pos_on_r > -1) {
        int rc = process_char(ix, &s, &ff, &sc);
        if (rc < 0) return rc;
        idx_out[s.emitted - 1] = s.idx;
        off_out[s.emitted - 1] = (uint32_t)s.offset;
    }
    return ORACLE_OK;
}

/* A batch, scheduled like ReadProcessor::process_latency_hiding
 * (src/read_processor.cpp:641-730): `strands` reads in flight per thread, one
 * base each per round, software prefetch of the next row (:719-722); OpenMP team
 * over groups of reads like src/movi.cpp:274-301.  offs has n_reads+1 entries
 * into seqs (bytes) and into out (u16 elements).  Used as the CPU baseline. */
int oracle_pml_batch(const oracle_index *ix, const uint8_t *seqs, const uint64_t *offs,
                     uint64_t n_reads, uint16_t *out, int threads, int strands,
                     uint64_t *ff_tot, uint64_t *scan_tot) {
    if (ix->mode == 5 || ix->mode == 3 || ix->mode == 2) return ORACLE_ERR_FORMAT;        /* no thresholds: see oracle_pml */
    if (strands < 1) strands = 1;
    if (strands > 64) strands = 64;
    uint64_t n_groups = (n_reads + (uint64_t)strands - 1) / (uint64_t)strands;
    uint64_t ff_all = 0, sc_all = 0;
    int err = ORACLE_OK;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    #pragma omp parallel for schedule(dynamic, 4) reduction(+:ff_all, sc_all)
    for (uint64_t g = 0; g < n_groups; g++) {
        strand_t st[64];
        int live[64];
        uint64_t first = g * (uint64_t)strands;
        int ns = 0, alive = 0;
        for (int k = 0; k < strands && first + k < n_reads; k++, ns++) {
            uint64_t rd_i = first + k;
            int64_t len = (int64_t)(offs[rd_i + 1] - offs[rd_i]);
            strand_reset(ix, &st[k], seqs + offs[rd_i], len, out + offs[rd_i]);
            live[k] = len > 0;
            alive += live[k];
        }
        uint64_t ff = 0, sc = 0;
        while (alive) {
            for (int k = 0; k < ns; k++) {
                if (!live[k]) continue;
                int rc = process_char(ix, &st[k], &ff, &sc);
                if (rc < 0) {
                    #pragma omp atomic write
                    err = rc;
                    live[k] = 0; alive--;
                    continue;
                }
                if (st[k].pos_on_r <= -1) { live[k] = 0; alive--; }
                else __builtin_prefetch(ix->rows + get_id(ix, st[k].idx) * ix->row_bytes, 0, 1);
            }
        }
        ff_all += ff; sc_all += sc;
    }
    if (ff_tot) *ff_tot = ff_all;
    if (scan_tot) *scan_tot = sc_all;
    return err;
}

/* ---------------------------------------------------------------------- count */

typedef struct { uint64_t rs, os, re, oe; } interval_t;   /* MoveInterval, include/move_intervals.hpp:10-76 */

/* MoveInterval::is_empty, move_intervals.hpp:43-45 */
static inline int iv_empty(const interval_t *v) {
    return !((v->rs < v->re) || (v->rs == v->re && v->os <= v->oe));
}
/* MoveInterval::count, move_intervals.hpp:47-58 */
static uint64_t iv_count(const oracle_index *ix, const interval_t *v) {
    uint64_t row_count;
    if (v->rs == v->re) {
        row_count = v->oe - v->os + 1;
    } else {
        row_count = (get_n(ix, v->rs) - v->os) + (v->oe + 1);
        for (uint64_t k = v->rs + 1; k < v->re; k++) row_count += get_n(ix, k);
    }
    return row_count;
}

/* MoveStructure::update_interval, src/move_structure_search.cpp:48-61 */
static void update_interval(const oracle_index *ix, interval_t *v, uint8_t next_char) {
    while (v->rs <= v->re && get_char(ix, v->rs) != next_char) {
        v->rs += 1;
        v->os = 0;
        if (v->rs >= ix->r) break;
    }
    while (v->re >= v->rs && get_char(ix, v->re) != next_char) {
        v->re -= 1;
        v->oe = get_n(ix, v->re) - 1;
        if (v->re == 0) break;
    }
}

/* MoveStructure::query_backward_search, src/move_structure_search.cpp:340-352, with
 * initialize_backward_search :261-293 (ftab_k == 0), backward_search :169-201 and
 * backward_search_step :311-333.  Output as src/utils.cpp:248-256 prints it:
 * matched = len - pos_on_r, count = rows in the last non-empty interval. */
int oracle_count(const oracle_index *ix, const uint8_t *R, int64_t len,
                 uint64_t *matched, uint64_t *count) {
    if (len <= 0) { *matched = 0; *count = 0; return ORACLE_OK; }
    int64_t pos_on_r = len - 1;
    if (!check_alphabet(ix, R[pos_on_r])) {             /* :344-347 */
        pos_on_r += 1;
        *matched = (uint64_t)(len - pos_on_r);
        *count = 0;
        return ORACLE_OK;
    }
    uint64_t ci = ix->alphamap[R[pos_on_r]] + 1;         /* :284-291 */
    interval_t iv = { ix->first_runs[ci], ix->first_offsets[ci], ix->last_runs[ci], ix->last_offsets[ci] };
    interval_t prev = iv;
    while (pos_on_r > 0 && !iv_empty(&iv)) {            /* :176 */
        prev = iv;
        if (!check_alphabet(ix, R[pos_on_r - 1])) {     /* :321-324 */
            iv.rs = 1; iv.os = 0; iv.re = 0; iv.oe = 0; /* make_empty, move_intervals.hpp:36-41 */
        } else {
            update_interval(ix, &iv, R[pos_on_r - 1]);
            if (!iv_empty(&iv)) {
                if (LF_move(ix, &iv.os, &iv.rs) < 0) return ORACLE_ERR_INVARIANT;
                if (LF_move(ix, &iv.oe, &iv.re) < 0) return ORACLE_ERR_INVARIANT;
            }
        }
        if (!iv_empty(&iv)) pos_on_r -= 1;
    }
    const interval_t *res = iv_empty(&iv) ? &prev : &iv;
    *matched = (uint64_t)(len - pos_on_r);
    *count = iv_count(ix, res);
    return ORACLE_OK;
}

int oracle_count_batch(const oracle_index *ix, const uint8_t *seqs, const uint64_t *offs,
                       uint64_t n_reads, uint64_t *matched, uint64_t *count, int threads) {
    int err = ORACLE_OK;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t i = 0; i < n_reads; i++) {
        int rc = oracle_count(ix, seqs + offs[i], (int64_t)(offs[i + 1] - offs[i]), &matched[i], &count[i]);
        if (rc < 0) {
            #pragma omp atomic write
            err = rc;
        }
    }
    return err;
}

/* ------------------------------------------------------------------ ZML
 * MoveStructure::query_zml, src/move_structure_query.cpp:690-785 (the Ziv-Merhav cross parse):
 * greedy backward search from the last base; while the match extends, position pos gets the
 * number of bases already matched to its right (match_len before the step); when it cannot be
 * extended the phrase ends, pos gets match_len, and a new search starts at pos-1.  One u16 per
 * base (MoveQuery::add_ml clamp), emitted last base first like the PMLs.  Multi-classify
 * (doc_pats) is not restated.
 * NOT pinned to a reference output (the reference's tests hold no ZML vector): pinned in
 * tests/ against a brute-force substring search over the fixture text instead. */
int oracle_zml(const oracle_index *ix, const uint8_t *R, int64_t len, uint16_t *out) {
    int64_t pos_on_r = len - 1, k = 0;
    uint64_t match_len = 0;
    /* :696-704 (the reference tests the character before the bound; same outputs) */
    while (pos_on_r >= 0 && !check_alphabet(ix, R[pos_on_r])) { out[k++] = 0; pos_on_r -= 1; }
    if (pos_on_r < 0) return ORACLE_OK;
    uint64_t ci = ix->alphamap[R[pos_on_r]] + 1;         /* initialize_backward_search :284-291 */
    interval_t iv = { ix->first_runs[ci], ix->first_offsets[ci], ix->last_runs[ci], ix->last_offsets[ci] };
    while (pos_on_r > 0) {                               /* :714 */
        /* backward_search_step(query_seq, pos_on_r, interval), move_structure_search.cpp:311-333 */
        if (!check_alphabet(ix, R[pos_on_r - 1])) {
            iv.rs = 1; iv.os = 0; iv.re = 0; iv.oe = 0;
        } else {
            update_interval(ix, &iv, R[pos_on_r - 1]);
            if (!iv_empty(&iv)) {
                if (LF_move(ix, &iv.os, &iv.rs) < 0) return ORACLE_ERR_INVARIANT;
                if (LF_move(ix, &iv.oe, &iv.re) < 0) return ORACLE_ERR_INVARIANT;
            }
        }
        if (!iv_empty(&iv)) {                            /* :717-720 */
            out[k++] = (uint16_t)(match_len < 65535 ? match_len : 65535);
            pos_on_r -= 1;
            match_len += 1;
        } else {                                         /* :750-760 */
            out[k++] = (uint16_t)(match_len < 65535 ? match_len : 65535);
            pos_on_r -= 1;
            match_len = 0;
            while (!check_alphabet(ix, R[pos_on_r]) && pos_on_r > 0) { out[k++] = 0; pos_on_r -= 1; }
            if (check_alphabet(ix, R[pos_on_r])) {
                ci = ix->alphamap[R[pos_on_r]] + 1;
                iv.rs = ix->first_runs[ci]; iv.os = ix->first_offsets[ci];
                iv.re = ix->last_runs[ci];  iv.oe = ix->last_offsets[ci];
            }
        }
    }
    if (iv_empty(&iv)) match_len = 0;                    /* :762-765 */
    out[k++] = (uint16_t)(match_len < 65535 ? match_len : 65535);
    return ORACLE_OK;
}

int oracle_zml_batch(const oracle_index *ix, const uint8_t *seqs, const uint64_t *offs,
                     uint64_t n_reads, uint16_t *out, int threads) {
    int err = ORACLE_OK;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t i = 0; i < n_reads; i++) {
        int rc = oracle_zml(ix, seqs + offs[i], (int64_t)(offs[i + 1] - offs[i]), out + offs[i]);
        if (rc < 0) {
            #pragma omp atomic write
            err = rc;
        }
    }
    return err;
}

/* Single LF step exposed for generator / property tests. */
/* MoveStructure::get_id of every row (ids[i] = r where the reference throws).  For the sampled modes this is the
 * checkpoint scan + walk-back, INCLUDING what it does when r is a multiple of tally_checkpoints: the builder stores the
 * final ids at entry r / cp + 1 (src/move_structure_build.cpp:677-682) but rows of the last span look at entry
 * r / cp (src/move_structure.cpp:127, :151), which nothing ever wrote -- such LFs throw or go astray in the reference. */
void oracle_get_ids(const oracle_index *ix, uint64_t *ids) {
    for (uint64_t i = 0; i < ix->r; i++) {
        uint64_t id = get_id(ix, i);
        ids[i] = id >= ix->r ? ix->r : id;
    }
}

int oracle_lf(const oracle_index *ix, uint64_t *idx, uint64_t *offset) {
    int64_t ff = LF_move(ix, offset, idx);
    return ff < 0 ? (int)ff : ORACLE_OK;
}
