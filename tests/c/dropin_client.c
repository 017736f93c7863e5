/* A C client of the drop-in boundary (include/sassy.h), compiled and linked by
 * tests/test_gpu_parity.py::test_compiled_c_client_links_and_runs:
 *
 *     gcc tests/c/dropin_client.c -Iinclude -Lsassy_amd/lib -lsassy_hip -lpthread -lm
 *
 * It uses nothing but the four symbols the reference's header declares (c/sassy.h:38-63) in the call
 * order of the reference's own example (c/example.c:14-29): sassy_searcher -> search ->
 * sassy_matches_free -> sassy_searcher_free.  Two host threads run at the same time, each with a
 * searcher of its own ("one searcher per thread", src/c.rs / SURVEY 8b); every thread repeats its
 * searches a few times and prints its matches once, one line per match:
 *
 *     <thread> <search> <text_start> <text_end> <pattern_start> <pattern_end> <cost> <strand>
 *
 * The Python side runs the oracle on the same inputs and compares. */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sassy.h"

#define TEXT_LEN 200000u

struct job {
    int id;
    const char *alphabet;
    bool rc;
    const char *patterns[3];
    size_t k[3];
    unsigned char *text;
    char *out;      /* printed lines */
    size_t out_len;
    int failed;
};

/* a small deterministic text: xorshift letters, with near-copies of the patterns planted */
static void fill_text(unsigned char *t, size_t n, unsigned seed, const char *const *pats, int npat) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        t[i] = (unsigned char)acgt[(x >> 9) & 3];
    }
    for (int p = 0; p < npat; p++) {
        const size_t m = strlen(pats[p]);
        for (size_t at = 1000 + 977u * (size_t)p; at + m + 8 < n; at += 40000) {
            memcpy(t + at, pats[p], m);
            t[at + m / 2] = (unsigned char)(t[at + m / 2] == 'A' ? 'C' : 'A'); /* one substitution */
        }
    }
    /* an exact copy that straddles the end of the text is NOT planted: ends are the oracle's business */
}

static void *run(void *arg) {
    struct job *j = (struct job *)arg;
    sassy_SearcherType *s = sassy_searcher(j->alphabet, j->rc, NAN);
    size_t cap = 1 << 16;
    j->out = (char *)malloc(cap);
    j->out_len = 0;
    for (int rep = 0; rep < 4; rep++) {
        for (int q = 0; q < 3; q++) {
            sassy_Match *ms = NULL;
            const size_t n = search(s, (const uint8_t *)j->patterns[q], strlen(j->patterns[q]), j->text, TEXT_LEN,
                                    j->k[q], &ms);
            if (ms == NULL) j->failed = 1; /* never null, also for zero matches */
            if (rep == 3) {
                for (size_t i = 0; i < n; i++) {
                    if (j->out_len + 160 > cap) { cap *= 2; j->out = (char *)realloc(j->out, cap); }
                    j->out_len += (size_t)snprintf(j->out + j->out_len, cap - j->out_len, "%d %d %zu %zu %zu %zu %d %c\n",
                                                   j->id, q, ms[i].text_start, ms[i].text_end, ms[i].pattern_start,
                                                   ms[i].pattern_end, (int)ms[i].cost, ms[i].strand == 0 ? '+' : '-');
                }
            }
            sassy_matches_free(ms, n);
        }
    }
    sassy_searcher_free(s);
    return NULL;
}

int main(void) {
    static const char *p0[3] = {"ACGGTCAGGTTACGATCGGATCAGTTAGCAAT", "TTGACCAGTA", "GATTACAGATTACA"};
    static const char *p1[3] = {"ACGNTCAGGTYACGATCGRATCAGTTAGCWAT", "CCATGGCATGCCATGG", "AAAAAAAAAAAAAAAAAAAA"};
    struct job jobs[2] = {
        {0, "dna", true, {p0[0], p0[1], p0[2]}, {3, 1, 2}, NULL, NULL, 0, 0},
        {1, "iupac", false, {p1[0], p1[1], p1[2]}, {3, 2, 3}, NULL, NULL, 0, 0},
    };
    pthread_t th[2];
    for (int t = 0; t < 2; t++) {
        jobs[t].text = (unsigned char *)malloc(TEXT_LEN);
        fill_text(jobs[t].text, TEXT_LEN, 7u + (unsigned)t, t == 0 ? p0 : p1, 3);
    }
    /* the text is part of the output so that the checker searches exactly these bytes */
    for (int t = 0; t < 2; t++) {
        printf("text %d ", t);
        fwrite(jobs[t].text, 1, TEXT_LEN, stdout);
        printf("\n");
    }
    for (int t = 0; t < 2; t++) pthread_create(&th[t], NULL, run, &jobs[t]);
    for (int t = 0; t < 2; t++) pthread_join(th[t], NULL);
    int failed = 0;
    for (int t = 0; t < 2; t++) {
        fwrite(jobs[t].out, 1, jobs[t].out_len, stdout);
        failed |= jobs[t].failed;
        free(jobs[t].out);
        free(jobs[t].text);
    }
    printf("done %d\n", failed);
    return failed;
}
