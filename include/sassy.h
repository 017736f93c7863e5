/*
 * sassy.h -- the drop-in C-ABI of the MI355X-native search path.
 *
 * Same symbols, argument meaning, struct layout and ownership rules as the reference's
 * generated header (reference: c/sassy.h:9-63, implemented by src/c.rs:52-131), so a caller of
 * the reference library links against libsassy_hip.so unchanged.
 * Every entry point below cites the reference interface it replaces.
 *
 * Errors: the reference panics (process abort) on null pointers, unknown alphabet, invalid IUPAC
 * pattern, or overhang with a non-IUPAC alphabet (src/c.rs:57,66,76,99; src/profiles/iupac.rs:19-24;
 * src/search.rs:373-383).  These entry points do the same: message on stderr, then abort().
 * There is no CPU fallback: without a usable HIP device `search` aborts with a message.
 */
#ifndef SASSY_H
#define SASSY_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque searcher (reference: c/sassy.h:9, src/c.rs:10-14). */
typedef struct sassy_SearcherType sassy_SearcherType;

/* One match; repr(C) layout of src/c.rs:16-27 / c/sassy.h:11-21: 40 bytes, align 8.
 * strand: 0 = Fwd, 1 = Rc.  For Rc matches text_start/text_end index the FORWARD text. */
typedef struct sassy_Match {
  uintptr_t text_start;
  uintptr_t text_end;
  uintptr_t pattern_start;
  uintptr_t pattern_end;
  int32_t cost;
  uint8_t strand;
} sassy_Match;

/* Replaces `sassy_searcher` (c/sassy.h:38, src/c.rs:52-70).
 * alphabet: "ascii" | "dna" | "iupac" (case-insensitive).  rc: also search the reverse
 * complement strand.  alpha: overhang cost per overhanging pattern character, 0 <= alpha <= 1,
 * NAN disables; overhang is defined for "iupac" only -- any other alphabet with a non-NAN alpha
 * aborts with a message, like the reference's Searcher::new panics (src/search.rs:373-383).
 * "ascii" with rc = true constructs (as in the reference) and aborts at the first search: the
 * reference's Ascii profile has no complement. */
struct sassy_SearcherType *sassy_searcher(const char *alphabet, bool rc, float alpha);

/* Replaces `sassy_searcher_free` (c/sassy.h:43, src/c.rs:74-81). */
void sassy_searcher_free(struct sassy_SearcherType *ptr);

/* Replaces `search` (c/sassy.h:52-58, src/c.rs:89-122) = Searcher::<P>::search
 * (src/search.rs:510-525): one match per rightmost local-minimum end position with cost <= k,
 * forward matches by increasing end, then reverse-complement matches.
 * `text` is a host pointer; it is copied to the device, scanned by the HIP kernels, and the
 * matches are written to a malloc'ed array stored in *out_matches.  Returns the match count.
 * Free with sassy_matches_free(ptr, len) using the returned count. */
uintptr_t search(struct sassy_SearcherType *searcher, const uint8_t *pattern,
                 uintptr_t pattern_len, const uint8_t *text, uintptr_t text_len, uintptr_t k,
                 struct sassy_Match **out_matches);

/* Replaces `sassy_matches_free` (c/sassy.h:63, src/c.rs:126-131). ptr must be non-null. */
void sassy_matches_free(struct sassy_Match *ptr, uintptr_t len);

#ifdef __cplusplus
}
#endif
#endif /* SASSY_H */
