// tiled_step.h -- the Myers column step with the bits along the pattern (one pattern per lane), shared by the
// pattern-tiled scan (tiled_kernel.hip) and the seeded search's verification (seed_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sassy_hip {

// One text character for the pattern of a lane (src/pattern_tiling/search.rs:148-175): `eq` = the rows of the
// pattern that match the character; the last row's cost is tracked in `cost`.  The fresh state (vp = ones, vn = 0,
// cost = m) is a fixed point of a character that matches no row (eq = 0).
template <typename Word>
struct TiledState {
  Word vp, vn;
  int cost;
};
template <typename Word>
__device__ __forceinline__ void tiled_step(TiledState<Word>& S, const Word eq, const uint32_t top_shift) {
  const Word sum = (eq & S.vp) + S.vp;
  const Word xh = (sum ^ S.vp) | eq;
  const Word mh = S.vp & xh;
  const Word ph = S.vn | ~(xh | S.vp);
  // (the top row sits in the upper half of a 64-bit word: WORDS = 2 is used for m > 32 only)
  const uint32_t pht = sizeof(Word) == 8 ? (uint32_t)((unsigned long long)ph >> 32) : (uint32_t)ph;
  const uint32_t mht = sizeof(Word) == 8 ? (uint32_t)((unsigned long long)mh >> 32) : (uint32_t)mh;
  S.cost += (int)__builtin_amdgcn_ubfe(pht, top_shift, 1u) + (int)__builtin_amdgcn_sbfe((int)mht, top_shift, 1u);
  const Word phs = ph << 1;  // the row above the pattern is free: 0 shifted in
  S.vp = (mh << 1) | ~(eq | S.vn | phs);
  S.vn = phs & (eq | S.vn);
}

}  // namespace sassy_hip
