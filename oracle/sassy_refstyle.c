/*
 * sassy_refstyle.c -- CPU ORACLE, part 2 (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A C restatement of the *shape* of the reference's v1 scan: text tiled in 64-column words,
 * LANES far-apart chunks of one text advanced together as one SIMD vector, bounded rows with the
 * reference's re-check cadence, per-lane minima scan, lane-overlap pruning.  It exists for three
 * things only:
 *   (1) the "cpu_baseline" leg of bench.py (kind = "port": the reference is Rust and cannot be
 *       built in this image, so its algorithm is timed through this port; g++/gcc -O3 -mavx2
 *       -mbmi2 turns the 4x u64 GCC vectors into AVX2 registers),
 *   (2) differential tests against the naive definition in sassy_oracle.c,
 *   (3) quantifying the reference's lane-seam artefact: with LANES text
 *       chunks each starting with decreasing=true, low-complexity texts can yield reports that a
 *       single left-to-right pass would not give.
 *
 * Follows (reference file:line, relative to /root/reference):
 *   src/search.rs:1008-1070 (search_prep), :1074-1199 (search_internal), :941-975 (min_in_lane,
 *   check_lanes), :1202-1240 (prune_lane_overlaps), :1244-1271 (reset_rows,
 *   should_terminate_early), :1286-1369 (find_minima_with_overhang, alpha=None),
 *   src/bitpacking.rs:63-85 (compute_block_simd), src/minima.rs:5-92 (prefix_min),
 *   src/profiles/dna.rs:26-40 and iupac.rs:68-128 (encode_ref), src/search.rs:192-210
 *   (update_and_encode incl. 'X' padding).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__BMI2__) || defined(__AVX2__)
#include <immintrin.h>
#endif

#ifndef RS_LANES
#define RS_LANES 4
#endif
typedef uint64_t vlane __attribute__((vector_size(8 * RS_LANES)));

#define ORC_ASCII 0
#define ORC_DNA 1
#define ORC_IUPAC 2
#define CHECK_AT_LEAST_ROWS 8 /* src/search.rs:361 */

extern uint8_t orc_iupac_code(uint8_t c);

/* ------------------------------------------------------------ prefix_min */
/* src/minima.rs: minimum over t in [0,64] of sum_{b<t} (p_b - m_b). Only .0 (the min) is used
 * by the caller (src/search.rs:947). */
#ifdef __BMI2__
static int8_t packed_min[256], packed_end[256];
#else
static int8_t nib_min[256], nib_end[256];
#endif
static int tables_ready = 0;
static void init_tables(void) {
    if (tables_ready) return;
#ifdef __BMI2__
    for (int i = 0; i < 256; i++) { /* minima.rs:5-27: bit=1 -> -1, bit=0 -> +1 */
        int mn = 0, cur = 0;
        for (int j = 0; j < 8; j++) {
            cur += ((i >> j) & 1) ? -1 : 1;
            if (cur < mn) mn = cur;
        }
        packed_min[i] = (int8_t)mn;
        packed_end[i] = (int8_t)cur;
    }
#else
    for (int i = 0; i < 256; i++) { /* minima.rs:32-57 */
        int mn = 0, cur = 0, pos = i & 15, neg = i >> 4;
        for (int j = 0; j < 4; j++) {
            cur += ((pos >> j) & 1) - ((neg >> j) & 1);
            if (cur < mn) mn = cur;
        }
        nib_min[i] = (int8_t)mn;
        nib_end[i] = (int8_t)cur;
    }
#endif
    tables_ready = 1;
}
static inline int prefix_min(uint64_t p, uint64_t m) {
#ifdef __BMI2__
    uint64_t delta = p | m; /* minima.rs:62-77 */
    uint64_t deltas = _pext_u64(m, delta);
    int mn = 0, cur = 0;
    for (int i = 0; i < 8; i++) {
        uint8_t byte = (uint8_t)(deltas >> (i * 8));
        int a = cur + packed_min[byte];
        if (a < mn) mn = a;
        cur += packed_end[byte];
    }
    return mn;
#else
    int mn = 0, cur = 0; /* minima.rs:81-92 */
    for (int i = 0; i < 16; i++) {
        uint8_t byte = (uint8_t)(((p >> (i * 4)) & 15) | ((m >> (i * 4)) << 4));
        int a = cur + nib_min[byte];
        if (a < mn) mn = a;
        cur += nib_end[byte];
    }
    return mn;
#endif
}

/* --------------------------------------------------------------- profile */
typedef struct {
    int profile;
    int nslots;
    uint8_t slot_char[256]; /* pattern letter (or code) owning each slot */
    uint16_t *row_slot;     /* per pattern row: slot index */
} rs_profile;

/* encode_pattern: dna.rs:19-23 (slot = 2-bit code), iupac.rs:18-36 (A,C,T,G then extra letters
 * in first-seen order, upper-cased), ascii.rs:18-29 (distinct bytes in first-seen order). */
static int build_profile(rs_profile *pr, int profile, const uint8_t *pat, size_t m) {
    pr->profile = profile;
    pr->row_slot = (uint16_t *)malloc((m ? m : 1) * sizeof(uint16_t));
    if (profile == ORC_DNA) {
        pr->nslots = 4;
        for (size_t j = 0; j < m; j++) pr->row_slot[j] = (pat[j] >> 1) & 3;
        return 0;
    }
    if (profile == ORC_IUPAC) {
        static const char base4[4] = {'A', 'C', 'T', 'G'};
        pr->nslots = 4;
        for (int i = 0; i < 4; i++) pr->slot_char[i] = (uint8_t)base4[i];
    } else {
        pr->nslots = 0;
    }
    for (size_t j = 0; j < m; j++) {
        uint8_t c = profile == ORC_IUPAC ? (uint8_t)(pat[j] & ~0x20) : pat[j];
        int s = -1;
        for (int q = 0; q < pr->nslots; q++)
            if (pr->slot_char[q] == c) { s = q; break; }
        if (s < 0) {
            if (pr->nslots >= 256) return -1;
            s = pr->nslots;
            pr->slot_char[pr->nslots++] = c;
        }
        pr->row_slot[j] = (uint16_t)s;
    }
    return 0;
}

/* encode_ref: one u64 mask per slot for 64 text bytes.  The reference builds these with two
 * 32-byte SIMD compares + movemask per slot; the AVX2 path below does the same so that the timed
 * CPU baseline is not handicapped by a scalar profile builder. */
#ifdef __AVX2__
static inline uint64_t mm2(__m256i lo, __m256i hi) {
    return (uint64_t)(uint32_t)_mm256_movemask_epi8(lo) |
           ((uint64_t)(uint32_t)_mm256_movemask_epi8(hi) << 32);
}
static __m256i iupac_lo_tab, iupac_hi_tab;
static int iupac_simd_ready = 0;
static void iupac_simd_init(void) {
    if (iupac_simd_ready) return;
    uint8_t lo[32], hi[32];
    for (int i = 0; i < 16; i++) {
        lo[i] = lo[i + 16] = orc_iupac_code((uint8_t)i) & 0x0F;
        hi[i] = hi[i + 16] = orc_iupac_code((uint8_t)(i + 16)) & 0x0F;
    }
    iupac_lo_tab = _mm256_loadu_si256((const __m256i *)lo);
    iupac_hi_tab = _mm256_loadu_si256((const __m256i *)hi);
    iupac_simd_ready = 1;
}
static inline __m256i iupac_nibbles(__m256i c) {
    __m256i idx5 = _mm256_and_si256(c, _mm256_set1_epi8(0x1F));
    __m256i low4 = _mm256_and_si256(c, _mm256_set1_epi8(0x0F));
    __m256i is_hi = _mm256_cmpgt_epi8(idx5, _mm256_set1_epi8(15));
    __m256i a = _mm256_shuffle_epi8(iupac_lo_tab, low4);
    __m256i b = _mm256_shuffle_epi8(iupac_hi_tab, low4);
    return _mm256_blendv_epi8(a, b, is_hi);
}
#endif
static void encode_block(const rs_profile *pr, const uint8_t b[64], uint64_t *out) {
#ifdef __AVX2__
    __m256i c0 = _mm256_loadu_si256((const __m256i *)b);
    __m256i c1 = _mm256_loadu_si256((const __m256i *)(b + 32));
    if (pr->profile == ORC_DNA) {
        __m256i three = _mm256_set1_epi8(3);
        __m256i b0 = _mm256_and_si256(_mm256_srli_epi16(c0, 1), three);
        __m256i b1 = _mm256_and_si256(_mm256_srli_epi16(c1, 1), three);
        for (int s = 0; s < 4; s++) {
            __m256i code = _mm256_set1_epi8((char)s);
            out[s] = mm2(_mm256_cmpeq_epi8(b0, code), _mm256_cmpeq_epi8(b1, code));
        }
    } else if (pr->profile == ORC_IUPAC) {
        iupac_simd_init();
        __m256i n0 = iupac_nibbles(c0), n1 = iupac_nibbles(c1);
        __m256i z = _mm256_setzero_si256();
        for (int s = 0; s < pr->nslots; s++) {
            __m256i code = _mm256_set1_epi8((char)(orc_iupac_code(pr->slot_char[s]) & 0x0F));
            __m256i e0 = _mm256_cmpeq_epi8(_mm256_and_si256(n0, code), z);
            __m256i e1 = _mm256_cmpeq_epi8(_mm256_and_si256(n1, code), z);
            out[s] = ~mm2(e0, e1);
        }
    } else {
        for (int s = 0; s < pr->nslots; s++) {
            __m256i code = _mm256_set1_epi8((char)pr->slot_char[s]);
            out[s] = mm2(_mm256_cmpeq_epi8(c0, code), _mm256_cmpeq_epi8(c1, code));
        }
    }
#else
    for (int s = 0; s < pr->nslots; s++) out[s] = 0;
    if (pr->profile == ORC_DNA) {
        for (int i = 0; i < 64; i++) out[(b[i] >> 1) & 3] |= 1ULL << i;
    } else if (pr->profile == ORC_IUPAC) {
        for (int i = 0; i < 64; i++) {
            uint8_t nib = orc_iupac_code(b[i]) & 0x0F;
            for (int s = 0; s < pr->nslots; s++)
                if (nib & orc_iupac_code(pr->slot_char[s])) out[s] |= 1ULL << i;
        }
    } else {
        for (int i = 0; i < 64; i++)
            for (int s = 0; s < pr->nslots; s++)
                if (b[i] == pr->slot_char[s]) out[s] |= 1ULL << i;
    }
#endif
}

/* ------------------------------------------------------------ scan state */
typedef struct {
    uint64_t pos;
    int32_t cost;
} rs_end;
typedef struct {
    rs_end *v;
    size_t n, cap;
} rs_endvec;
static void ev_push(rs_endvec *e, uint64_t pos, int32_t cost) {
    if (e->n == e->cap) {
        e->cap = e->cap ? 2 * e->cap : 16;
        e->v = (rs_end *)realloc(e->v, e->cap * sizeof(rs_end));
    }
    e->v[e->n].pos = pos;
    e->v[e->n].cost = cost;
    e->n++;
}

typedef struct {
    int decreasing;
    size_t chunk_offset, lane_end;
    rs_endvec ends;
    uint64_t *masks; /* nslots */
} rs_lane;

/* src/search.rs:1286-1369 with alpha = None. */
static void find_minima(rs_lane *ln, uint64_t p, uint64_t m, int32_t cur_cost, int32_t k,
                        size_t text_len, size_t base_pos, int all_minima) {
    size_t max_pos = text_len;
    int32_t cost = cur_cost, prev_cost = cur_cost;
    size_t prev_pos = base_pos;
    if (base_pos >= max_pos) return;
    if (all_minima && cost <= k && prev_pos == 0) ev_push(&ln->ends, prev_pos, cost);
    for (int bit = 1; bit <= 64; bit++) {
        size_t pos = base_pos + (size_t)bit;
        if (pos > max_pos) break;
        cost += (int32_t)((p >> (bit - 1)) & 1);
        cost -= (int32_t)((m >> (bit - 1)) & 1);
        if (all_minima) {
            if (cost <= k) ev_push(&ln->ends, pos, cost);
        } else {
            if (ln->decreasing && cost > prev_cost && prev_cost <= k)
                ev_push(&ln->ends, prev_pos, prev_cost);
            ln->decreasing = (cost < prev_cost) || (ln->decreasing && cost == prev_cost);
        }
        prev_cost = cost;
        prev_pos = pos;
    }
    if (!all_minima && prev_pos == max_pos && ln->decreasing && prev_cost <= k)
        ev_push(&ln->ends, prev_pos, prev_cost);
}

/*
 * The reference-shaped scan.  Writes the (end_pos, cost) reports, lane after lane, into
 * *out_pos / *out_cost (malloc'ed, caller frees) and returns their number.
 * stats (optional, 2 x u64): [0] = word-rows computed (one per compute_block_simd call),
 * [1] = blocks visited.
 */
size_t rs_scan(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n, int32_t k,
               int all_minima, uint64_t **out_pos, int32_t **out_cost, uint64_t *stats) {
    init_tables();
    *out_pos = NULL;
    *out_cost = NULL;
    rs_profile pr;
    if (build_profile(&pr, profile, pat, m) != 0) return 0;

    /* search_prep, single text + single pattern: src/search.rs:1018-1056 */
    size_t overlap = (m + (size_t)k + 63) / 64;
    size_t nblocks = (n + 63) / 64;
    size_t rest = nblocks > overlap ? nblocks - overlap : 0;
    size_t bpc = (rest + RS_LANES - 1) / RS_LANES;
    rs_lane lanes[RS_LANES];
    memset(lanes, 0, sizeof lanes);
    for (int l = 0; l < RS_LANES; l++) {
        lanes[l].chunk_offset = (size_t)l * bpc;
        lanes[l].lane_end = ((size_t)l + 1) * bpc * 64;
        lanes[l].decreasing = 1;
        lanes[l].masks = (uint64_t *)calloc((size_t)pr.nslots ? (size_t)pr.nslots : 1, 8);
    }
    vlane *hp = (vlane *)aligned_alloc(64, (m ? m : 1) * sizeof(vlane));
    vlane *hm = (vlane *)aligned_alloc(64, (m ? m : 1) * sizeof(vlane));
    vlane one, zero;
    for (int l = 0; l < RS_LANES; l++) { one[l] = 1; zero[l] = 0; }
    for (size_t j = 0; j < m; j++) { hp[j] = one; hm[j] = zero; }

    size_t prev_max_j = 0, prev_end_last_below = 0;
    uint64_t rows_done = 0, blocks_done = 0;

    for (size_t i = 0; i < bpc + overlap; i++) {
        vlane vp = zero, vm = zero;
        for (int l = 0; l < RS_LANES; l++) { /* update_and_encode: src/search.rs:192-210 */
            size_t start = lanes[l].chunk_offset * 64 + 64 * i;
            lanes[l].lane_end = start + 64;
            uint8_t slice[64];
            if (start + 64 <= n) {
                memcpy(slice, text + start, 64);
            } else {
                memset(slice, 'X', 64);
                if (start <= n) memcpy(slice, text + start, n - start);
            }
            encode_block(&pr, slice, lanes[l].masks);
        }
        blocks_done++;
        vlane dist_to_start = zero, dist_to_end = zero;
        size_t cur_end_last_below = 0;
        int skipped = 0, terminate = 0;

        for (size_t j = 0; j < m; j++) {
            dist_to_start += hp[j];
            dist_to_start -= hm[j];
            vlane eq;
            for (int l = 0; l < RS_LANES; l++) eq[l] = lanes[l].masks[pr.row_slot[j]];
            /* compute_block_simd: src/bitpacking.rs:63-85 */
            vlane vx = eq | vm;
            vlane eq2 = eq | hm[j];
            vlane hx = (((eq2 & vp) + vp) ^ vp) | eq2;
            vlane hpv = vm | ~(hx | vp);
            vlane hmv = vp & hx;
            vlane hpw = hpv >> 63, hmw = hmv >> 63;
            hpv = (hpv << 1) | hp[j];
            hmv = (hmv << 1) | hm[j];
            hp[j] = hpw;
            hm[j] = hmw;
            vp = hmv | ~(vx | hpv);
            vm = hpv & vx;
            rows_done++;

            dist_to_end += hp[j];
            dist_to_end -= hm[j];
            int any_below = 0;
            for (int l = 0; l < RS_LANES; l++)
                if (dist_to_end[l] < (uint64_t)k + 1) any_below = 1;
            if (any_below) cur_end_last_below = j;

            if (j > prev_end_last_below) {
                /* check_lanes: src/search.rs:952-975 */
                int found = 0;
                for (int l = 0; l < RS_LANES; l++) {
                    int32_t mn = prefix_min(vp[l], vm[l]) + (int32_t)dist_to_start[l];
                    if (mn <= k) {
                        size_t rows_needed = (size_t)(k - mn);
                        prev_end_last_below =
                            j + (rows_needed > CHECK_AT_LEAST_ROWS ? rows_needed : CHECK_AT_LEAST_ROWS);
                        found = 1;
                        break;
                    }
                }
                if (found) continue;
                for (size_t j2 = j + 1; j2 <= prev_max_j && j2 < m; j2++) { hp[j2] = one; hm[j2] = zero; }
                prev_end_last_below =
                    cur_end_last_below > CHECK_AT_LEAST_ROWS ? cur_end_last_below : CHECK_AT_LEAST_ROWS;
                prev_max_j = j;
                /* should_terminate_early: src/search.rs:1253-1271 */
                if (i >= bpc) {
                    size_t d = 64 * (i - bpc);
                    d = d > j ? d - j : 0;
                    if (d > (size_t)k) terminate = 1;
                }
                skipped = 1;
                break;
            }
        }
        if (terminate) break;
        if (skipped) continue;

        for (int l = 0; l < RS_LANES; l++) {
            int32_t cost = (int32_t)dist_to_start[l];
            if (prefix_min(vp[l], vm[l]) + cost <= k) {
                size_t base_pos = lanes[l].chunk_offset * 64 + 64 * i;
                find_minima(&lanes[l], vp[l], vm[l], cost, k, n, base_pos, all_minima);
            }
        }
        prev_end_last_below =
            cur_end_last_below > CHECK_AT_LEAST_ROWS ? cur_end_last_below : CHECK_AT_LEAST_ROWS;
        prev_max_j = m ? m - 1 : 0;
    }

    /* prune_lane_overlaps: src/search.rs:1202-1240 */
    size_t total = 0;
    for (int l = 0; l < RS_LANES; l++) {
        rs_endvec *e = &lanes[l].ends;
        size_t w = 0;
        for (size_t q = 0; q < e->n; q++) {
            uint64_t pos = e->v[q].pos;
            int keep;
            if (l == 0) {
                keep = pos < lanes[0].lane_end;
            } else {
                int too_early = pos < lanes[l - 1].lane_end;
                int too_late = pos >= lanes[l].lane_end && l != RS_LANES - 1;
                keep = !too_early && !too_late;
            }
            if (keep) e->v[w++] = e->v[q];
        }
        e->n = w;
        total += w;
    }
    uint64_t *op = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    int32_t *oc = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));
    size_t w = 0;
    for (int l = 0; l < RS_LANES; l++) {
        for (size_t q = 0; q < lanes[l].ends.n; q++) {
            op[w] = lanes[l].ends.v[q].pos;
            oc[w] = lanes[l].ends.v[q].cost;
            w++;
        }
        free(lanes[l].ends.v);
        free(lanes[l].masks);
    }
    free(hp); free(hm); free(pr.row_slot);
    if (stats) { stats[0] = rows_done; stats[1] = blocks_done; }
    *out_pos = op;
    *out_cost = oc;
    return total;
}


/* ------------------------------------------------------------ many host threads (bench.py's cpu_baseline)
 * The analogue of the reference CLI's record-parallel threads (bin/grep.rs:476-503: N threads, each with
 * its own Searcher): the text is cut into `threads` shards, every persistent worker scans its shard plus
 * `overlap` bytes to its left with rs_scan and keeps the reports it owns (end position inside the shard;
 * shard 0 also owns position 0).  The workers are created once, meet at a barrier, and then repeat the
 * scan pass after pass until `min_seconds` of wall time have gone by (at least one pass); the clock
 * runs between the first barrier and the last one, so thread creation is not timed.  Returns the number
 * of owned reports of the last pass (merged in shard order into *out_pos / *out_cost, malloc'ed) and
 * writes the passes done and the seconds they took. */
#include <pthread.h>
#include <time.h>

typedef struct {
    int profile; const uint8_t *pat; size_t m; const uint8_t *text; size_t n; int32_t k;
    size_t a, b, s;              /* owns (a, b] (+ 0 if a == 0), scans [s, b) */
    uint64_t *pos; int32_t *cost; size_t cnt;
    double busy;                 /* seconds this worker spent scanning */
    pthread_barrier_t *bar; volatile int *stop; int id;
    double min_seconds; double *elapsed; int *passes;
} rs_mt_worker;

static double rs_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *rs_mt_main(void *arg) {
    rs_mt_worker *w = (rs_mt_worker *)arg;
    double t0 = 0;
    pthread_barrier_wait(w->bar);
    if (w->id == 0) t0 = rs_now();
    for (;;) {
        uint64_t *op = NULL; int32_t *oc = NULL;
        const double b0 = rs_now();
        size_t c = rs_scan(w->profile, w->pat, w->m, w->text + w->s, w->b - w->s, w->k, 0, &op, &oc, NULL);
        size_t keep = 0;
        for (size_t i = 0; i < c; i++) {
            const uint64_t g = op[i] + w->s;
            if ((g > w->a || w->a == 0) && g <= w->b) { op[keep] = g; oc[keep] = oc[i]; keep++; }
        }
        w->busy += rs_now() - b0;
        free(w->pos); free(w->cost);
        w->pos = op; w->cost = oc; w->cnt = keep;
        pthread_barrier_wait(w->bar);
        if (w->id == 0) {
            *w->passes += 1;
            *w->elapsed = rs_now() - t0;
            if (*w->elapsed >= w->min_seconds) *w->stop = 1;
        }
        pthread_barrier_wait(w->bar);
        if (*w->stop) break;
    }
    return NULL;
}

size_t rs_scan_mt(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n, int32_t k,
                  int threads, double min_seconds, uint64_t **out_pos, int32_t **out_cost,
                  int *passes_out, double *seconds_out, int *shards_out, double *busy_out) {
    init_tables();
    *out_pos = NULL; *out_cost = NULL;
    if (threads < 1) threads = 1;
    const size_t ov = ((m + (size_t)k + 1 + 63) / 64) * 64;
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    per = (per + 63) / 64 * 64;
    if (per == 0) per = 64;
    rs_mt_worker *ws = (rs_mt_worker *)calloc((size_t)threads, sizeof *ws);
    int T = 0;
    for (int t = 0; t < threads; t++) {
        const size_t a = (size_t)t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        ws[T].a = a; ws[T].b = b; ws[T].s = a > ov ? a - ov : 0;
        T++;
    }
    if (T == 0) { free(ws); *passes_out = 0; *seconds_out = 0; *shards_out = 0; return 0; }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)T);
    volatile int stop = 0;
    int passes = 0;
    double elapsed = 0;
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof *th);
    for (int t = 0; t < T; t++) {
        rs_mt_worker *w = &ws[t];
        w->profile = profile; w->pat = pat; w->m = m; w->text = text; w->n = n; w->k = k;
        w->bar = &bar; w->stop = &stop; w->id = t; w->min_seconds = min_seconds;
        w->elapsed = &elapsed; w->passes = &passes;
        pthread_create(&th[t], NULL, rs_mt_main, w);
    }
    size_t total = 0;
    double busy = 0;
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); total += ws[t].cnt; busy += ws[t].busy; }
    pthread_barrier_destroy(&bar);
    uint64_t *op = (uint64_t *)malloc((total ? total : 1) * sizeof *op);
    int32_t *oc = (int32_t *)malloc((total ? total : 1) * sizeof *oc);
    size_t w = 0;
    for (int t = 0; t < T; t++) {
        for (size_t i = 0; i < ws[t].cnt; i++) { op[w] = ws[t].pos[i]; oc[w] = ws[t].cost[i]; w++; }
        free(ws[t].pos); free(ws[t].cost);
    }
    free(th); free(ws);
    *out_pos = op; *out_cost = oc;
    *passes_out = passes; *seconds_out = elapsed; *shards_out = T;
    if (busy_out) *busy_out = busy;
    return total;
}

int rs_lanes(void) { return RS_LANES; }
void rs_free(void *p) { free(p); }

/* Test hook: profile masks of one 64-byte block for the profile built from `pat`
 * (pins encode_ref against dna.rs:173-233 and iupac.rs:378-472). Returns the slot count. */
int rs_encode_block_test(int profile, const uint8_t *pat, size_t m, const uint8_t *block64,
                         uint64_t *out) {
    rs_profile pr;
    if (build_profile(&pr, profile, pat, m) != 0) return -1;
    encode_block(&pr, block64, out);
    free(pr.row_slot);
    return pr.nslots;
}
