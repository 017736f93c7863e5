/*
 * sassy_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the *definition* of what sassy's search path computes,
 * written from the behavioural description of the reference (RagnarGrootKoerkamp/sassy).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * The product library (libsassy_hip.so) never links or loads it.
 *
 * Pinning: checked against the reference's own known-answer tests committed as
 * tests/golden/kats.json (see tests/test_oracle_kats.py). The reference itself is Rust
 * and cannot be built here (no cargo/rustc), so there is no oracle/_ref.
 *
 * What is restated (reference file:line, relative to /root/reference):
 *   - cost model: unit-cost edit distance, free start in the text, D[j][0] = j
 *       src/bitpacking.rs:8-28, src/search.rs:1060-1061 (vertical deltas +1), :1101 (horizontal 0)
 *   - profiles: Dna   src/profiles/dna.rs:19-23,48-50,100-102,121-133
 *               Iupac src/profiles/iupac.rs:18-36,136-138,156-204,235-344
 *               Ascii src/profiles/ascii.rs:18-59 (case sensitive, as used by src/c.rs:64)
 *   - which end positions are reported (rightmost position of every local-minimum plateau
 *     with cost <= k; or every position with cost <= k for search_all)
 *       src/search.rs:1286-1369
 *   - traceback on the window text[end-(m+k) .. end) with preference '=', 'X', 'D', 'I'
 *       src/search.rs:1477-1478, src/trace.rs:57-104 (fill), :273-406 (get_trace)
 *   - reverse-complement strand handling   src/search.rs:813-878
 *   - pre-encoded multi-pattern search ("v2"): equal as a sorted set to one forward search per
 *     pattern, plus one forward search of rc(pattern) tagged Rc when the searcher is rc
 *       src/pattern_tiling/search.rs:690-848 (the reference's own differential test states this)
 *
 * Also here (SURVEY 8f rows the library implements): overhang (alpha, max_overhang; orc_search_overhang,
 * src/search.rs:347-356, 1274-1308, 1695-1748, src/trace.rs:36-47); max_n_frac (src/n_filter.rs:8-60) is restated
 * in oracle/__init__.py on top of these searches.
 * Deliberately NOT restated here: SIMD lanes, bounded rows, chunking (see sassy_refstyle.c for
 * the reference-shaped algorithm).
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#define ORC_ASCII 0
#define ORC_DNA 1
#define ORC_IUPAC 2

/* ---------------------------------------------------------------- profiles */

/* src/profiles/iupac.rs:281-317 -- 5-bit letter index -> set of bases (A=1,C=2,T=4,G=8);
 * everything that is not an IUPAC letter is 255; X is the empty set. */
static uint8_t iupac_code_tab[32];
static int iupac_ready = 0;
static void iupac_init(void) {
    if (iupac_ready) return;
    const uint8_t A = 1, C = 2, T = 4, G = 8;
    for (int i = 0; i < 32; i++) iupac_code_tab[i] = 255;
    iupac_code_tab['A' & 31] = A;
    iupac_code_tab['C' & 31] = C;
    iupac_code_tab['T' & 31] = T;
    iupac_code_tab['U' & 31] = T;
    iupac_code_tab['G' & 31] = G;
    iupac_code_tab['N' & 31] = A | C | T | G;
    iupac_code_tab['R' & 31] = A | G;
    iupac_code_tab['Y' & 31] = C | T;
    iupac_code_tab['S' & 31] = G | C;
    iupac_code_tab['W' & 31] = A | T;
    iupac_code_tab['K' & 31] = G | T;
    iupac_code_tab['M' & 31] = A | C;
    iupac_code_tab['B' & 31] = C | G | T;
    iupac_code_tab['D' & 31] = A | G | T;
    iupac_code_tab['H' & 31] = A | C | T;
    iupac_code_tab['V' & 31] = A | C | G;
    iupac_code_tab['X' & 31] = 0;
    iupac_ready = 1;
}
uint8_t orc_iupac_code(uint8_t c) {
    iupac_init();
    return iupac_code_tab[c & 31];
}

/* Equality used by the *scan* (the bit-parallel profile).
 *  Dna:   2-bit code (c>>1)&3 compared for any byte        dna.rs:19-23,26-40
 *  Iupac: low nibbles of the two codes intersect           iupac.rs:68-128 (mask = nib & code(base))
 *  Ascii: byte equality                                    ascii.rs:75-90 */
static inline int scan_eq(int profile, uint8_t p, uint8_t t) {
    switch (profile) {
    case ORC_DNA: return ((p >> 1) & 3) == ((t >> 1) & 3);
    case ORC_IUPAC: return ((iupac_code_tab[p & 31] & iupac_code_tab[t & 31]) & 0x0F) != 0;
    default: return p == t;
    }
}
/* Equality used by the *traceback* (Profile::is_match).
 *  dna.rs:48-50, iupac.rs:136-138, ascii.rs:44-51 */
static inline int trace_is_match(int profile, uint8_t p, uint8_t t) {
    switch (profile) {
    case ORC_DNA: return (p | 0x20) == (t | 0x20);
    case ORC_IUPAC: return (iupac_code_tab[p & 31] & iupac_code_tab[t & 31]) > 0;
    default: return p == t;
    }
}

/* Pattern validity. Iupac: letters whose code != 255 (iupac.rs:156-204). Dna and Ascii accept
 * anything in encode_pattern (dna.rs:19-23, ascii.rs:18-29). */
int orc_valid_pattern(int profile, const uint8_t *p, size_t m) {
    iupac_init();
    if (profile != ORC_IUPAC) return 1;
    for (size_t i = 0; i < m; i++) {
        uint8_t c = p[i] & (uint8_t)~0x20;
        if (c <= '@' || c >= 'Z' || iupac_code_tab[c & 31] == 255) return 0;
    }
    return 1;
}

/* Complement tables: dna.rs:121-133 (upper case ACGT only), iupac.rs:235-278 (both cases). */
static uint8_t comp_of(int profile, uint8_t c) {
    if (profile == ORC_DNA) {
        switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        default: return c;
        }
    }
    static const char from[] = "ACTGRYSWKMBDHVNX";
    static const char to[] = "TGACYRSWMKVHDBNX";
    for (int i = 0; from[i]; i++) {
        if (c == (uint8_t)from[i]) return (uint8_t)to[i];
        if (c == (uint8_t)(from[i] | 0x20)) return (uint8_t)(to[i] | 0x20);
    }
    return c;
}
void orc_complement(int profile, const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = comp_of(profile, in[i]);
}
void orc_reverse_complement(int profile, const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = comp_of(profile, in[n - 1 - i]);
}

/* ------------------------------------------------- last row of the DP matrix */

/* C[i] = D[m][i], i in [0,n].  D[0][i] = 0, D[j][0] = j,
 * D[j][i] = min(D[j-1][i-1] + !eq, D[j][i-1] + 1, D[j-1][i] + 1).      */
void orc_last_row(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n,
                  int32_t *C) {
    iupac_init();
    int32_t *col = (int32_t *)malloc((m + 1) * sizeof(int32_t));
    for (size_t j = 0; j <= m; j++) col[j] = (int32_t)j;
    C[0] = (int32_t)m;
    for (size_t i = 1; i <= n; i++) {
        uint8_t t = text[i - 1];
        int32_t diag = col[0]; /* D[j-1][i-1] */
        col[0] = 0;
        for (size_t j = 1; j <= m; j++) {
            int32_t up = col[j - 1];  /* D[j-1][i]  */
            int32_t left = col[j];    /* D[j][i-1]  */
            int32_t v = diag + (scan_eq(profile, pat[j - 1], t) ? 0 : 1);
            if (left + 1 < v) v = left + 1;
            if (up + 1 < v) v = up + 1;
            diag = left;
            col[j] = v;
        }
        C[i] = col[m];
    }
    free(col);
}

/* Which (end_pos, cost) are reported -- src/search.rs:1310-1368 on the un-chunked text.
 * all_minima=0: rightmost position of each local-minimum plateau with cost <= k, plus the text
 * end if still decreasing; all_minima=1: every position with cost <= k (position 0 included).
 * Returns the number of reports written (at most cap). */
size_t orc_find_ends(const int32_t *C, size_t n, int32_t k, int all_minima, uint64_t *pos,
                     int32_t *cost, size_t cap) {
    size_t cnt = 0;
    if (n == 0) return 0; /* base_pos >= max_pos returns early: search.rs:1314-1316 */
    if (all_minima) {
        for (size_t i = 0; i <= n; i++)
            if (C[i] <= k) {
                if (cnt < cap) { pos[cnt] = i; cost[cnt] = C[i]; }
                cnt++;
            }
        return cnt;
    }
    int dec = 1;
    int32_t prev = C[0];
    for (size_t i = 1; i <= n; i++) {
        if (dec && C[i] > prev && prev <= k) {
            if (cnt < cap) { pos[cnt] = i - 1; cost[cnt] = prev; }
            cnt++;
        }
        dec = (C[i] < prev) || (dec && C[i] == prev);
        prev = C[i];
    }
    if (dec && prev <= k) {
        if (cnt < cap) { pos[cnt] = n; cost[cnt] = prev; }
        cnt++;
    }
    return cnt;
}

/* ------------------------------------------------------------- traceback */

/* Trace one end position.  Window w = text[o .. min(e,n)), o = max(0, e-(m+k))
 * (src/search.rs:1477-1478); local matrix L[j][0] = j, L[0][i] = 0 (src/trace.rs:80-103);
 * walk from (m, e-o) preferring '=', then 'X', 'D', 'I' (src/trace.rs:337-365).
 * ops receives the CIGAR characters in pattern direction (already reversed), one per column
 * of the alignment; returns its length, or -1 if no ancestor is found (the reference panics).
 */
long orc_trace(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n, size_t e,
               int32_t k, uint64_t *text_start, int32_t *cost_out, char *ops, size_t ops_cap) {
    iupac_init();
    size_t fill = m + (size_t)k;
    size_t o = e > fill ? e - fill : 0;
    size_t wend = e < n ? e : n;
    size_t wl = wend - o; /* window length; e <= n when there is no overhang */
    size_t W = wl + 1;
    int32_t *L = (int32_t *)malloc((m + 1) * W * sizeof(int32_t));
#define LL(j, i) L[(j) * W + (i)]
    for (size_t i = 0; i <= wl; i++) LL(0, i) = 0;
    for (size_t j = 1; j <= m; j++) {
        LL(j, 0) = (int32_t)j;
        for (size_t i = 1; i <= wl; i++) {
            int32_t v = LL(j - 1, i - 1) + (scan_eq(profile, pat[j - 1], text[o + i - 1]) ? 0 : 1);
            if (LL(j, i - 1) + 1 < v) v = LL(j, i - 1) + 1;
            if (LL(j - 1, i) + 1 < v) v = LL(j - 1, i) + 1;
            LL(j, i) = v;
        }
    }
    size_t j = m, i = e - o;
    if (i > wl) i = wl;
    int32_t g = LL(j, i);
    *cost_out = g;
    size_t nops = 0;
    long rc = 0;
    while (j > 0) {
        if (nops >= ops_cap) { rc = -2; break; }
        if (i > 0 && LL(j - 1, i - 1) == g && trace_is_match(profile, pat[j - 1], text[o + i - 1])) {
            ops[nops++] = '=';
            j--; i--;
            continue;
        }
        g -= 1;
        if (i > 0 && LL(j - 1, i - 1) == g) { ops[nops++] = 'X'; j--; i--; continue; }
        if (i > 0 && LL(j, i - 1) == g) { ops[nops++] = 'D'; i--; continue; }
        if (LL(j - 1, i) == g) { ops[nops++] = 'I'; j--; continue; }
        rc = -1; /* reference: panic "Trace failed! No ancestor found" (trace.rs:384-387) */
        break;
    }
#undef LL
    free(L);
    if (rc < 0) return rc;
    if (g != 0) return -3; /* reference asserts g == 0 (trace.rs:390) */
    for (size_t a = 0, b = nops; a + 1 < b; a++, b--) { /* reverse: trace ran end -> start */
        char tmp = ops[a]; ops[a] = ops[b - 1]; ops[b - 1] = tmp;
    }
    *text_start = o + i;
    return (long)nops;
}

/* --------------------------------------------------------------- full search */

typedef struct {
    uint64_t pattern_idx;
    uint64_t text_start, text_end, pattern_start, pattern_end;
    int32_t cost;
    uint8_t strand; /* 0 = Fwd, 1 = Rc */
    uint64_t cigar_off; /* offset into the ops pool */
    uint32_t cigar_len; /* number of op characters (not run-length encoded) */
} orc_match;

typedef struct {
    orc_match *m;
    size_t n, cap;
    char *ops;
    size_t ops_n, ops_cap;
    int failed;
} orc_result;

static void res_push(orc_result *r, orc_match mm, const char *ops, size_t nops) {
    if (r->n == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 64;
        r->m = (orc_match *)realloc(r->m, r->cap * sizeof(orc_match));
    }
    while (r->ops_n + nops + 1 > r->ops_cap) {
        r->ops_cap = r->ops_cap ? 2 * r->ops_cap : 4096;
        r->ops = (char *)realloc(r->ops, r->ops_cap);
    }
    mm.cigar_off = r->ops_n;
    mm.cigar_len = (uint32_t)nops;
    memcpy(r->ops + r->ops_n, ops, nops);
    r->ops_n += nops;
    r->m[r->n++] = mm;
}

/* One strand: scan, report ends, trace each (src/search.rs:884-937, :1372-1517). */
static void one_strand(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n,
                       int32_t k, int all_minima, orc_result *r) {
    int32_t *C = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    orc_last_row(profile, pat, m, text, n, C);
    size_t cap = n + 2;
    uint64_t *pos = (uint64_t *)malloc(cap * sizeof(uint64_t));
    int32_t *cost = (int32_t *)malloc(cap * sizeof(int32_t));
    size_t cnt = orc_find_ends(C, n, k, all_minima, pos, cost, cap);
    char *ops = (char *)malloc(2 * (m + (size_t)k) + 8);
    for (size_t q = 0; q < cnt; q++) {
        orc_match mm;
        memset(&mm, 0, sizeof mm);
        int32_t c2 = 0;
        uint64_t ts = 0;
        long nops = orc_trace(profile, pat, m, text, n, pos[q], k, &ts, &c2, ops, 2 * (m + (size_t)k) + 8);
        if (nops < 0 || c2 > cost[q] || c2 > k) { r->failed = 1; continue; }
        mm.text_start = ts;
        mm.text_end = pos[q] < n ? pos[q] : n;
        mm.pattern_start = 0;
        mm.pattern_end = m;
        mm.cost = c2;
        mm.strand = 0;
        res_push(r, mm, ops, (size_t)nops);
    }
    free(ops); free(pos); free(cost); free(C);
}

/* Searcher::search / search_all (src/search.rs:510-525, :685-700, :787-881).
 * Output order: all Fwd matches by increasing end, then all Rc matches by increasing end in the
 * reversed text.  Rc: complement(pattern) against reverse(text), then
 * text_start = n - rc_end, text_end = n - rc_start (search.rs:859-877). */
orc_result *orc_search(int profile, int rc, int all_minima, const uint8_t *pat, size_t m,
                       const uint8_t *text, size_t n, int32_t k) {
    orc_result *r = (orc_result *)calloc(1, sizeof(orc_result));
    iupac_init();
    one_strand(profile, pat, m, text, n, k, all_minima, r);
    if (rc) {
        size_t first = r->n;
        uint8_t *cp = (uint8_t *)malloc(m ? m : 1);
        uint8_t *rt = (uint8_t *)malloc(n ? n : 1);
        orc_complement(profile, pat, m, cp);
        for (size_t i = 0; i < n; i++) rt[i] = text[n - 1 - i];
        one_strand(profile, cp, m, rt, n, k, all_minima, r);
        for (size_t q = first; q < r->n; q++) {
            uint64_t s = r->m[q].text_start, e = r->m[q].text_end;
            r->m[q].strand = 1;
            r->m[q].text_start = n - e;
            r->m[q].text_end = n - s;
        }
        free(cp); free(rt);
    }
    return r;
}

/* Searcher::search_encoded_patterns semantics (src/search.rs:415-423,
 * src/pattern_tiling/general.rs:335-404): every pattern (all of one length) searched forward;
 * if rc, also rc(pattern) searched forward and tagged strand = Rc with the same pattern_idx
 * (tqueries.rs:74-80, trace.rs:444-449).  The reference's order is an implementation artefact;
 * callers sort by (pattern_idx, text_start, text_end, cost, strand, cigar) before comparing
 * (pattern_tiling/search.rs:748-757).  `pats` holds npat patterns of length m back to back. */
orc_result *orc_search_encoded(int profile, int rc, int all_minima, const uint8_t *pats,
                               size_t npat, size_t m, const uint8_t *text, size_t n, int32_t k) {
    orc_result *r = (orc_result *)calloc(1, sizeof(orc_result));
    iupac_init();
    uint8_t *rcp = (uint8_t *)malloc(m ? m : 1);
    for (size_t p = 0; p < npat; p++) {
        size_t first = r->n;
        one_strand(profile, pats + p * m, m, text, n, k, all_minima, r);
        for (size_t q = first; q < r->n; q++) r->m[q].pattern_idx = p;
        if (rc) {
            first = r->n;
            orc_reverse_complement(ORC_IUPAC, pats + p * m, m, rcp); /* tqueries.rs:2,77 */
            one_strand(profile, rcp, m, text, n, k, all_minima, r);
            for (size_t q = first; q < r->n; q++) { r->m[q].pattern_idx = p; r->m[q].strand = 1; }
        }
    }
    free(rcp);
    return r;
}

/* ------------------------------------------------------------------ overhang
 * Searcher::new_*_with_overhang(alpha) / with_max_overhang (Iupac only, src/search.rs:373-440):
 *  - left edge: the vertical delta of pattern row i at text position 0 is
 *    floor((i+1) alpha) - floor(i alpha) for i < max_overhang, else 1 (search.rs:1695-1748), i.e.
 *    D[j][0] = floor(min(j, mo) alpha) + max(0, j - mo) (trace.rs:36-47);
 *  - right edge: the text is virtually extended by steps = min(m, ceil((k + alpha) / alpha), mo)
 *    'N' columns (search.rs:347-356, :1024-1026); an end position pos > n costs an extra
 *    floor(alpha (pos - n)) (search.rs:1274-1282) and positions up to n + steps are considered
 *    (search.rs:1298-1308);
 *  - traceback (trace.rs:273-406): window text[e-(m+k) .. min(e, n)) whose local matrix ALWAYS has
 *    the overhang left column (also when the window starts inside the text) and 'N' beyond the
 *    text; an end past the text first steps back diagonally (pattern_end = m - overshoot); a walk
 *    that reaches column 0 stops there (pattern_start = remaining rows).
 * All float arithmetic is f32, as in the reference. */
static size_t ov_steps(size_t m, int32_t k, float alpha, long mo) {
    size_t s = m;
    if (alpha > 0.0f) {
        float q = ceilf(((float)k + alpha) / alpha);
        if (q < (float)s) s = (size_t)q;
    }
    if (mo >= 0 && (size_t)mo < s) s = (size_t)mo;
    return s;
}
static int32_t ov_left(size_t j, float alpha, long mo) {
    size_t a = j, extra = 0;
    if (mo >= 0 && (size_t)mo < j) { a = (size_t)mo; extra = j - (size_t)mo; }
    return (int32_t)floorf((float)a * alpha) + (int32_t)extra;
}
static int32_t ov_cost(size_t overshoot, float alpha) {
    return overshoot ? (int32_t)floorf(alpha * (float)overshoot) : 0;
}

/* Last DP row over the text extended by `steps` 'N' columns: C[i], i = 0 .. n + steps. */
static void last_row_ov(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n,
                        size_t steps, float alpha, long mo, int32_t *C) {
    int32_t *col = (int32_t *)malloc((m + 1) * sizeof(int32_t));
    for (size_t j = 0; j <= m; j++) col[j] = ov_left(j, alpha, mo);
    C[0] = col[m];
    for (size_t i = 1; i <= n + steps; i++) {
        const uint8_t t = i <= n ? text[i - 1] : (uint8_t)'N';
        int32_t diag = col[0];
        col[0] = 0;
        for (size_t j = 1; j <= m; j++) {
            int32_t left = col[j], up = col[j - 1];
            int32_t v = diag + (scan_eq(profile, pat[j - 1], t) ? 0 : 1);
            if (left + 1 < v) v = left + 1;
            if (up + 1 < v) v = up + 1;
            diag = left;
            col[j] = v;
        }
        C[i] = col[m];
    }
    free(col);
}

/* find_minima_with_overhang (search.rs:1286-1369) on total costs, positions 0 .. n + steps. */
static size_t find_ends_ov(const int32_t *C, size_t n, size_t steps, int32_t k, int all_minima, float alpha,
                           uint64_t *pos, int32_t *cost, size_t cap) {
    size_t cnt = 0, max_pos = n + steps;
    if (max_pos == 0) return 0;
#define TOT(i) (C[i] + ov_cost((i) > n ? (i) - n : 0, alpha))
    if (all_minima) {
        for (size_t i = 0; i <= max_pos; i++)
            if (TOT(i) <= k) {
                if (cnt < cap) { pos[cnt] = i; cost[cnt] = TOT(i); }
                cnt++;
            }
        return cnt;
    }
    int dec = 1;
    int32_t prev = TOT(0);
    for (size_t i = 1; i <= max_pos; i++) {
        int32_t t = TOT(i);
        if (dec && t > prev && prev <= k) {
            if (cnt < cap) { pos[cnt] = i - 1; cost[cnt] = prev; }
            cnt++;
        }
        dec = (t < prev) || (dec && t == prev);
        prev = t;
    }
    if (dec && prev <= k) {
        if (cnt < cap) { pos[cnt] = max_pos; cost[cnt] = prev; }
        cnt++;
    }
#undef TOT
    return cnt;
}

/* get_trace with overhang (trace.rs:273-406); fills the match fields, returns the number of ops
 * (pattern direction) or < 0 where the reference would panic. */
static long trace_ov(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n, size_t e,
                     int32_t k, float alpha, long mo, orc_match *mm, char *ops, size_t ops_cap) {
    size_t fill = m + (size_t)k;
    size_t o = e > fill ? e - fill : 0;
    size_t wend = e < n ? e : n;
    size_t wl = wend > o ? wend - o : 0;  /* text chars in the window */
    size_t iend = e - o;                   /* column of the end cell; > wl past the text end */
    size_t W = iend + 1;
    int32_t *L = (int32_t *)malloc((m + 1) * W * sizeof(int32_t));
#define LL(j, i) L[(j) * W + (i)]
    for (size_t i = 0; i <= iend; i++) LL(0, i) = 0;
    for (size_t j = 1; j <= m; j++) {
        LL(j, 0) = ov_left(j, alpha, mo);
        for (size_t i = 1; i <= iend; i++) {
            const uint8_t t = i <= wl ? text[o + i - 1] : (uint8_t)'N';
            int32_t v = LL(j - 1, i - 1) + (scan_eq(profile, pat[j - 1], t) ? 0 : 1);
            if (LL(j, i - 1) + 1 < v) v = LL(j, i - 1) + 1;
            if (LL(j - 1, i) + 1 < v) v = LL(j - 1, i) + 1;
            LL(j, i) = v;
        }
    }
    size_t j = m, i = iend;
    int32_t g = LL(j, i);
    int32_t total = g;
    size_t pattern_start = 0, pattern_end = m;
    long rc = 0;
    if (i > wl) {
        size_t over = i - wl;
        if (over > m) { free(L); return -4; }
        pattern_end -= over;
        total += ov_cost(over, alpha);
        i -= over;
        j -= over;
    }
    size_t nops = 0;
    for (;;) {
        if (j == 0) break;
        if (i == 0) {  /* overshoot at the start */
            pattern_start = j;
            g -= ov_left(j, alpha, mo);
            break;
        }
        if (nops >= ops_cap) { rc = -2; break; }
        if (LL(j - 1, i - 1) == g && trace_is_match(profile, pat[j - 1], text[o + i - 1])) {
            ops[nops++] = '='; j--; i--; continue;
        }
        g -= 1;
        if (LL(j - 1, i - 1) == g) { ops[nops++] = 'X'; j--; i--; continue; }
        if (LL(j, i - 1) == g) { ops[nops++] = 'D'; i--; continue; }
        if (LL(j - 1, i) == g) { ops[nops++] = 'I'; j--; continue; }
        rc = -1;
        break;
    }
#undef LL
    free(L);
    if (rc < 0) return rc;
    if (g != 0) return -3;
    for (size_t a = 0, b = nops; a + 1 < b; a++, b--) {
        char tmp = ops[a]; ops[a] = ops[b - 1]; ops[b - 1] = tmp;
    }
    mm->text_start = o + i;
    mm->text_end = o + wl;
    mm->pattern_start = pattern_start;
    mm->pattern_end = pattern_end;
    mm->cost = total;
    return (long)nops;
}

static void one_strand_ov(int profile, const uint8_t *pat, size_t m, const uint8_t *text, size_t n,
                          int32_t k, int all_minima, float alpha, long mo, orc_result *r) {
    if (n == 0) return; /* this build reports nothing for an empty text (overhang) */
    size_t steps = ov_steps(m, k, alpha, mo);
    int32_t *C = (int32_t *)malloc((n + steps + 1) * sizeof(int32_t));
    last_row_ov(profile, pat, m, text, n, steps, alpha, mo, C);
    size_t cap = n + steps + 2;
    uint64_t *pos = (uint64_t *)malloc(cap * sizeof(uint64_t));
    int32_t *cost = (int32_t *)malloc(cap * sizeof(int32_t));
    size_t cnt = find_ends_ov(C, n, steps, k, all_minima, alpha, pos, cost, cap);
    char *ops = (char *)malloc(2 * (m + (size_t)k) + 8);
    for (size_t q = 0; q < cnt; q++) {
        orc_match mm;
        memset(&mm, 0, sizeof mm);
        long nops = trace_ov(profile, pat, m, text, n, pos[q], k, alpha, mo, &mm, ops, 2 * (m + (size_t)k) + 8);
        if (nops < 0 || mm.cost > cost[q] || mm.cost > k) { r->failed = 1; continue; }
        mm.strand = 0;
        res_push(r, mm, ops, (size_t)nops);
    }
    free(ops); free(pos); free(cost); free(C);
}

/* Searcher::<Iupac>::new_{fwd,rc}_with_overhang(alpha).with_max_overhang(mo).search / search_all;
 * mo < 0 = no max_overhang. */
orc_result *orc_search_overhang(int profile, int rc, int all_minima, const uint8_t *pat, size_t m,
                                const uint8_t *text, size_t n, int32_t k, float alpha, long mo) {
    orc_result *r = (orc_result *)calloc(1, sizeof(orc_result));
    iupac_init();
    one_strand_ov(profile, pat, m, text, n, k, all_minima, alpha, mo, r);
    if (rc) {
        size_t first = r->n;
        uint8_t *cp = (uint8_t *)malloc(m ? m : 1);
        uint8_t *rt = (uint8_t *)malloc(n ? n : 1);
        orc_complement(profile, pat, m, cp);
        for (size_t i = 0; i < n; i++) rt[i] = text[n - 1 - i];
        one_strand_ov(profile, cp, m, rt, n, k, all_minima, alpha, mo, r);
        for (size_t q = first; q < r->n; q++) {
            uint64_t s = r->m[q].text_start, e = r->m[q].text_end;
            r->m[q].strand = 1;
            r->m[q].text_start = n - e;
            r->m[q].text_end = n - s;
        }
        free(cp); free(rt);
    }
    return r;
}

size_t orc_result_len(const orc_result *r) { return r->n; }
int orc_result_failed(const orc_result *r) { return r->failed; }
const orc_match *orc_result_matches(const orc_result *r) { return r->m; }
const char *orc_result_ops(const orc_result *r) { return r->ops; }
void orc_result_free(orc_result *r) {
    if (!r) return;
    free(r->m); free(r->ops); free(r);
}

/* --------------------------------------------------- synthetic text generator */

/* text byte i = "ACGT"[(h(seed, i>>5) >> (2*(i&31))) & 3],
 * h = splitmix64 finaliser of (seed * 0x9E3779B97F4A7C15 + (i>>5)).  The device generator in
 * the product library computes the same function; tests compare the two byte for byte. */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
uint64_t orc_hash(uint64_t seed, uint64_t idx) {
    return splitmix64(seed * 0x9E3779B97F4A7C15ULL + idx);
}
void orc_generate_dna(uint64_t seed, uint64_t first, uint64_t n, uint8_t *out) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t q = 0; q < n; q++) {
        uint64_t i = first + q;
        uint64_t h = orc_hash(seed, i >> 5);
        out[q] = (uint8_t)acgt[(h >> (2 * (i & 31))) & 3];
    }
}

/* ----------------------------------------------------------- planted matches */

/* random ACGT text has ~0 natural matches at m=32,k=3, so the benchmark plants one
 * mutated copy of the pattern every `stride` bytes (at offset stride/2 inside each stride).
 * Plant q carries e = q mod (k+1) edits drawn from the same counter-based hash:
 *   r = orc_hash(seed ^ "plant", 64*q + t); type = r % 3 (0 sub, 1 ins, 2 del);
 *   pos = (r >> 8) % len; base = (r >> 40) & 3.
 * The product library restates this on the host side of its device generator
 * (sassy_amd/csrc/c_abi.hip: sassy_hip_generate_dna / sassy_hip_plant); tests compare the two byte for byte. */
#define ORC_PLANT_SALT 0x706c616e74ULL
size_t orc_make_plant(uint64_t seed, uint64_t q, const uint8_t *pat, size_t m, int edits,
                      uint8_t *out) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    size_t len = m;
    memcpy(out, pat, m);
    for (int t = 0; t < edits; t++) {
        uint64_t r = orc_hash(seed ^ ORC_PLANT_SALT, 64 * q + (uint64_t)t);
        int type = (int)(r % 3);
        size_t pos = (size_t)((r >> 8) % len);
        int b = (int)((r >> 40) & 3);
        if (type == 0) {
            int idx = 0;
            for (int a = 0; a < 4; a++)
                if (out[pos] == (uint8_t)acgt[a]) idx = a;
            out[pos] = (uint8_t)acgt[(idx + 1 + (b % 3)) & 3];
        } else if (type == 1) {
            memmove(out + pos + 1, out + pos, len - pos);
            out[pos] = (uint8_t)acgt[b];
            len++;
        } else if (len > 1) {
            memmove(out + pos, out + pos + 1, len - pos - 1);
            len--;
        }
    }
    return len;
}

/* Overwrite the window [first, first+n) of a text of total length total_n with its plants. */
size_t orc_plant_window(uint64_t seed, uint64_t total_n, uint64_t first, uint64_t n, uint8_t *buf,
                        const uint8_t *pat, size_t m, int k, uint64_t stride) {
    uint8_t *tmp = (uint8_t *)malloc(m + (size_t)k + 1);
    size_t planted = 0;
    for (uint64_t q = 0;; q++) {
        uint64_t p = q * stride + stride / 2;
        if (p + m + (uint64_t)k > total_n) break;
        if (p >= first + n) break;
        size_t len = orc_make_plant(seed, q, pat, m, (int)(q % (uint64_t)(k + 1)), tmp);
        if (p + len <= first) continue;
        for (size_t i = 0; i < len; i++) {
            uint64_t g = p + i;
            if (g >= first && g < first + n) buf[g - first] = tmp[i];
        }
        planted++;
    }
    free(tmp);
    return planted;
}
