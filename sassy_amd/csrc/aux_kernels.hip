// aux_kernels.hip -- small device helpers around the scan kernel: synthetic text generator,
// plant scatter, text reversal for the reverse-complement strand.
#include <hip/hip_runtime.h>

#include "common.h"

namespace sassy_hip {

// ---------------------------------------------------------------- synthetic text
// byte i = "ACGT"[(h(seed, i>>5) >> (2*(i&31))) & 3], h = splitmix64(seed*0x9E3779B97F4A7C15 + (i>>5)).
// The CPU twin used by the tests is oracle/sassy_oracle.c:orc_generate_dna.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// One thread per 32-character hash block (32 output bytes, written as two 16-byte stores when the
// destination is aligned and fully inside the buffer).
__global__ __launch_bounds__(256) void generate_dna_kernel(uint8_t* out, uint64_t n, uint64_t seed,
                                                           uint64_t first) {
  const uint64_t nb = (first + n + 31) / 32 - first / 32;  // hash blocks touched
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nb; t += stride) {
    const uint64_t hb = first / 32 + t;
    const uint64_t h = splitmix64(seed * 0x9E3779B97F4A7C15ull + hb);
    // 32 chars; char c of the block: "ACGT"[(h >> 2c) & 3].  'A'=0x41 'C'=0x43 'G'=0x47 'T'=0x54
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t code = (uint32_t)(h >> (2 * (4 * q + b))) & 3u;
        const uint32_t ch = (0x54474341u >> (8 * code)) & 0xFFu;
        v |= ch << (8 * b);
      }
      w[q] = v;
    }
    const uint64_t g0 = hb * 32;  // global index of the block's first char
    if (g0 >= first && g0 + 32 <= first + n && (((uintptr_t)(out + (g0 - first))) & 15) == 0) {
      uint4* dst = reinterpret_cast<uint4*>(out + (g0 - first));
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
      for (int c = 0; c < 32; ++c) {
        const uint64_t g = g0 + c;
        if (g >= first && g < first + n) out[g - first] = (uint8_t)((w[c >> 2] >> (8 * (c & 3))) & 0xFFu);
      }
    }
  }
}

// A repeat-rich synthetic text for measuring the path on something less kind than i.i.d. letters (what a
// real genome does to a prefilter): the text is cut into 4 KiB regions whose kind is drawn from the region
// index --
//    6 %  microsatellite: a unit of 1..6 letters repeated through the region, 2 % of the letters substituted;
//   10 %  interspersed repeat: a copy of one of 4 family consensus sequences (4 KiB each, copy starts at a random
//         phase), 8 % of the letters substituted;
//    2 %  (only with `with_n`) a run of 'N' over the whole region; 1 % soft-masked (lower case) random letters;
//   rest  i.i.d. ACGT as generate_dna_kernel gives them.
// Every byte is a pure function of (seed, global index): any slice can be regenerated anywhere.
__device__ __forceinline__ uint8_t genome_like_byte(uint64_t seed, uint64_t g, bool with_n) {
  const uint64_t region = g >> 12;
  const uint32_t off = (uint32_t)(g & 4095u);
  const uint64_t h = splitmix64(seed * 0x9E3779B97F4A7C15ull + 0x5eed0000ull + region);
  const uint32_t kind = (uint32_t)(h % 100u);
  const uint64_t hb = splitmix64(seed * 0x9E3779B97F4A7C15ull + (g >> 5));
  const uint32_t rnd = (uint32_t)(hb >> (2 * (g & 31u))) & 3u;           // the i.i.d. letter of this position
  const uint64_t hm = splitmix64((seed ^ 0x6d757461ull) * 0x9E3779B97F4A7C15ull + g);  // mutation draw
  uint32_t code = rnd;
  bool lower = false;
  if (kind < 6) {
    const uint32_t period = 1u + (uint32_t)((h >> 8) % 6u);
    const uint32_t unit = (uint32_t)(h >> 16);                           // 2 bits per unit letter
    code = (unit >> (2 * (off % period))) & 3u;
    if (hm % 100u < 2u) code = (code + 1u + (uint32_t)((hm >> 8) % 3u)) & 3u;
  } else if (kind < 16) {
    const uint64_t fam = (h >> 8) & 3u;
    const uint32_t phase = (uint32_t)((h >> 12) & 4095u);
    const uint32_t cp = (off + phase) & 4095u;                           // position inside the family consensus
    const uint64_t hc = splitmix64((0xfa000000ull + fam) * 0x9E3779B97F4A7C15ull + (cp >> 5));
    code = (uint32_t)(hc >> (2 * (cp & 31u))) & 3u;
    if (hm % 100u < 8u) code = (code + 1u + (uint32_t)((hm >> 8) % 3u)) & 3u;
  } else if (kind < 18 && with_n) {
    return (uint8_t)'N';
  } else if (kind == 18) {
    lower = true;
  }
  const uint8_t ch = (uint8_t)((0x54474341u >> (8 * code)) & 0xFFu);
  return lower ? (uint8_t)(ch | 0x20u) : ch;
}
__global__ __launch_bounds__(256) void generate_genome_like_kernel(uint8_t* out, uint64_t n, uint64_t seed, uint64_t first,
                                                                   int with_n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    uint32_t v = 0;
    for (int b = 0; b < 4; ++b) v |= (uint32_t)genome_like_byte(seed, first + i + b, with_n != 0) << (8 * b);
    if (i + 4 <= n && ((uintptr_t)(out + i) & 3) == 0) *reinterpret_cast<uint32_t*>(out + i) = v;
    else
      for (int b = 0; b < 4 && i + b < n; ++b) out[i + b] = (uint8_t)(v >> (8 * b));
  }
}

// text[pos[i] - first] = val[i] for the planted bytes that fall into [first, first + n)
__global__ void scatter_bytes_kernel(uint8_t* text, uint64_t n, uint64_t first, const uint64_t* pos,
                                     const uint8_t* val, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t p = pos[i];
  if (p >= first && p < first + n) text[p - first] = val[i];
}

// out[i] = in[n-1-i]   (the reference searches complement(pattern) against the reversed text for
// the Rc strand, reference: src/search.rs:813-858).  HBM bound (n bytes in, n bytes out): every
// thread writes one aligned 16-byte vector; its 16 source bytes [n-o-16, n-o) are misaligned by
// n mod 16, so it reads the two aligned vectors that hold them (misaligned 16-byte loads cost a
// third of the stream rate, tools/ubench/unaligned_read.hip) and one v_perm_b32 per output dword
// does the byte shift and the reversal at once.  `in` and `out` are 16-byte aligned.
__global__ __launch_bounds__(256) void reverse_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t nv = (n + 15) / 16;
  const uint32_t r = (uint32_t)(n & 15u);      // the source pieces start r bytes into an aligned vector
  const uint32_t rb = r & 3u, rd = r >> 2;
  // v_perm_b32(hi, lo, sel): byte k of the result = byte sel[k] of (hi:lo); reversed bytes rb+3 .. rb
  const uint32_t sel = ((rb + 3u)) | ((rb + 2u) << 8) | ((rb + 1u) << 16) | (rb << 24);
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += stride) {
    const uint64_t o0 = t * 16;
    if (t >= 1 && o0 + 16 <= n) {
      const uint64_t a = n - o0 - 16 - r;      // aligned; [a, a + 32) lies inside the text for t >= 1
      const uint4 lo = *reinterpret_cast<const uint4*>(in + a);
      uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, 0u, 0u, 0u, 0u};
      if (r != 0) {
        const uint4 hi = *reinterpret_cast<const uint4*>(in + a + 16);
        w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
      }
      // source dword j (bytes r + 4j .. r + 4j + 3 of the 32) -> output dword 3 - j, bytes reversed;
      // rd is uniform, so the dword pick is a scalar switch with static register indices
      uint32_t d[4];
#define SASSY_REV_CASE(RD)                                                                   \
  case RD:                                                                                   \
    d[3] = __builtin_amdgcn_perm(w[RD + 1], w[RD], sel);                                     \
    d[2] = __builtin_amdgcn_perm(w[RD + 2], w[RD + 1], sel);                                 \
    d[1] = __builtin_amdgcn_perm(w[RD + 3], w[RD + 2], sel);                                 \
    d[0] = __builtin_amdgcn_perm(w[RD + 4], w[RD + 3], sel);                                 \
    break;
      switch (rd) {
        SASSY_REV_CASE(0)
        SASSY_REV_CASE(1)
        SASSY_REV_CASE(2)
        default:
          d[3] = __builtin_amdgcn_perm(w[4], w[3], sel);
          d[2] = __builtin_amdgcn_perm(w[5], w[4], sel);
          d[1] = __builtin_amdgcn_perm(w[6], w[5], sel);
          d[0] = __builtin_amdgcn_perm(w[7], w[6], sel);  // rb = 3 at most: bytes 3..6 of (w7:w6)
          break;
      }
#undef SASSY_REV_CASE
      *reinterpret_cast<uint4*>(out + o0) = make_uint4(d[0], d[1], d[2], d[3]);
    } else {  // the first vector (its second source vector would lie past the text) and the ragged last one
      for (int q = 0; q < 16 && o0 + q < n; ++q) out[o0 + q] = in[n - 1 - (o0 + q)];
    }
  }
}

// ---------------------------------------------------------------- K0b: hit bitmap -> chunk list
// A hit in block h (an exact piece occurrence ends there) means: cells <= k are possible in
// blocks [h, h+L]; the DP needs `wb` warm-up blocks in front.  A' = union of [h-wb, h+L] over all
// hits.  Every maximal run of A' becomes one chunk (long runs are cut at absolute multiples of
// `maxlen` blocks); a chunk whose left neighbour block is outside A' is flagged kDescClearBefore
// (its left edge holds no cell <= k, so the report rule starts exactly with decreasing = true).
struct BuildParams {
  const unsigned long long* hit;
  uint64_t n_words;
  uint64_t n_blocks;
  uint64_t first_owned;
  uint32_t wb, L, maxlen;
  ChunkDesc* desc;
  uint32_t* desc_count;
  uint32_t desc_cap;
  unsigned long long* hit_count;  // statistics: number of hit blocks (one atomic per workgroup)
};

// (long patterns: the reach of a hit -- L blocks to the right, wb to the left -- exceeds a bitmap word; rare, and then
// the chunk builder's time does not matter: bit by bit over the words in reach)
__device__ __noinline__ unsigned long long dilated_word_far(const unsigned long long* hit, uint64_t n_words, uint32_t L, uint32_t wb, long long w) {
  unsigned long long a = 0;
  const long long lo = w * 64 - (long long)L, hi = w * 64 + 63 + (long long)wb;  // marked blocks that reach word w
  for (long long s = (lo < 0 ? 0 : lo) >> 6; s <= (hi >> 6) && (uint64_t)s < n_words; ++s) {
    unsigned long long h = hit[s];
    while (h) {
      const long long b = s * 64 + (__ffsll((long long)h) - 1);
      h &= h - 1;
      // block b makes [b - wb, b + L] part of A'
      long long x0 = b - (long long)wb - w * 64, x1 = b + (long long)L - w * 64;
      if (x1 < 0 || x0 > 63) continue;
      if (x0 < 0) x0 = 0;
      if (x1 > 63) x1 = 63;
      a |= (x1 - x0 == 63) ? ~0ull : (((1ull << (x1 - x0 + 1)) - 1ull) << x0);
    }
  }
  return a;
}
__device__ __forceinline__ unsigned long long dilated_word(const BuildParams& P, long long w) {
  if (w < 0 || (uint64_t)w >= P.n_words) return 0ull;
  unsigned long long a;
  if (P.L >= 64 || P.wb >= 64) {
    a = dilated_word_far(P.hit, P.n_words, P.L, P.wb, w);
  } else {
    const unsigned long long cur = P.hit[w];
    const unsigned long long prev = w > 0 ? P.hit[w - 1] : 0ull;
    const unsigned long long next = (uint64_t)(w + 1) < P.n_words ? P.hit[w + 1] : 0ull;
    a = cur;
    for (uint32_t d = 1; d <= P.L; ++d) a |= (cur << d) | (prev >> (64 - d));    // h -> h + d
    for (uint32_t d = 1; d <= P.wb; ++d) a |= (cur >> d) | (next << (64 - d));   // h -> h - d (warm-up)
  }
  // blocks past the end of the buffer do not exist
  const uint64_t base = (uint64_t)w * 64;
  if (base + 64 > P.n_blocks) a &= (P.n_blocks > base) ? (~0ull >> (64 - (P.n_blocks - base))) : 0ull;
  return a;
}
__device__ __forceinline__ bool dilated_bit(const BuildParams& P, uint64_t blk) {
  return (dilated_word(P, (long long)(blk >> 6)) >> (blk & 63)) & 1ull;
}

// 1024-thread workgroups, one bitmap word (64 blocks) per thread.  The descriptor slots are claimed with ONE
// atomic per workgroup.  What bounds the kernel (23 us for the 5.9 MB bitmap of a 3 GB text) is not the
// number of those same-address atomics -- 8 words per thread, an eighth of the workgroups and atomics,
// measured 24 us -- but the latency chain inside a workgroup (bitmap loads, scan, atomic round trip,
// descriptor stores) at two resident workgroups per CU.
//
// Inside its slot range a workgroup files its chunks BY LENGTH (block visits of the lane that will walk the
// chunk: owned blocks + warm-up; classes 1 .. 7 and "8 or more"): a wave of the list kernel takes 64
// consecutive descriptors and runs as long as its longest chunk, so mixing a 6-block chunk among 2-block
// chunks makes 63 lanes idle for two thirds of the wave's life (config 4: average 2.7 block visits per chunk,
// 5 per wave).  The order of the descriptors is otherwise free (the list kernels take any lane <-> chunk
// assignment; the host sorts the chunk table when it needs the seam chain).
constexpr int kBuildWaves = 16;
constexpr int kLenClasses = 8;

// end (one past the last block) of the chunk that starts at bit i of word w
__device__ __forceinline__ uint64_t chunk_end(const BuildParams& P, uint64_t w, unsigned long long Aw, int i) {
  uint64_t e = w * 64 + (uint64_t)i + 1;
  unsigned long long rest = (i < 63) ? (Aw >> (i + 1)) : 0ull;      // A' bits of the blocks behind the start ..
  int avail = 63 - i;                                               // .. still inside this word
  uint64_t ww = w;
  for (;;) {
    // consecutive A' blocks at the bottom of `rest` (bits past `avail` are zero or ignored)
    const unsigned long long inv = ~rest;
    int run = inv ? (__ffsll((long long)inv) - 1) : 64;
    if (run > avail) run = avail;
    e += (uint64_t)run;
    if (run < avail) break;                       // hit a block outside A'
    ++ww;
    if (ww * 64 >= P.n_blocks) break;             // end of the buffer
    if (((ww * 64) & (uint64_t)(P.maxlen - 1)) == 0) break;  // cut at a multiple of maxlen
    rest = dilated_word(P, (long long)ww);
    avail = 64;
  }
  if (e > P.n_blocks) e = P.n_blocks;
  // cuts at absolute multiples of maxlen (a power of two): the chunk ends at the first one behind its start
  const uint64_t lo = w * 64 + (uint64_t)i;
  const uint64_t cut = (lo / P.maxlen + 1) * (uint64_t)P.maxlen;
  return e < cut ? e : cut;
}

__global__ __launch_bounds__(1024) void build_chunks_kernel(const BuildParams P) {
  __shared__ unsigned long long wave_cnt[kBuildWaves][2];  // per wave: chunks per length class, 16 bits each
  __shared__ uint32_t wave_hits[kBuildWaves];
  __shared__ uint32_t class_base[kLenClasses];             // first slot of each class inside the group's range
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long A = 0, starts = 0;
  bool prev_top = false;
  uint32_t my_hits = 0;
  unsigned long long cnt0 = 0, cnt1 = 0;  // this thread's chunks per class (classes 0-3 | 4-7), 16-bit fields
  if (w < P.n_words) {
    my_hits = (uint32_t)__popcll(P.hit[w]);
    A = dilated_word(P, (long long)w);
    if (A) {
      prev_top = w > 0 ? ((dilated_word(P, (long long)w - 1) >> 63) != 0) : false;
      // a chunk starts at every absolute multiple of maxlen (a power of two >= 8) that lies in A'
      unsigned long long align;
      if (P.maxlen >= 64) align = ((w * 64) & (uint64_t)(P.maxlen - 1)) == 0 ? 1ull : 0ull;
      else {
        align = 1ull;
        for (uint32_t sh = P.maxlen; sh < 64; sh <<= 1) align |= align << sh;  // bit 0 repeated every maxlen bits
      }
      unsigned long long own = ~0ull;  // only blocks >= first_owned are owned (the halo is warm-up)
      if (w * 64 < P.first_owned) own = (P.first_owned - w * 64 >= 64) ? 0ull : (~0ull << (P.first_owned - w * 64));
      unsigned long long first_bit = 0;  // the first owned block starts a chunk if it is in A'
      if (P.first_owned >= w * 64 && P.first_owned < w * 64 + 64) first_bit = 1ull << (P.first_owned - w * 64);
      starts = A & own & (~((A << 1) | (prev_top ? 1ull : 0ull)) | align | first_bit);
    }
  }
  // length class of a chunk: the block visits of its lane, 1 .. 7 -> class 0 .. 6, longer -> class 7
  auto visits_class = [&](uint64_t lo, uint64_t e, bool left_in) -> uint32_t {
    uint64_t v = e - lo;
    if (left_in) v += lo > P.wb ? P.wb : lo;  // a continuation chunk walks its warm-up blocks first
    return (uint32_t)(v >= kLenClasses ? kLenClasses - 1 : v - 1);
  };
  uint64_t e_first = 0, e_second = 0;  // ends of the thread's first two chunks (most threads have at most one):
  uint32_t seen = 0;                   // the second pass below does not walk their runs again
  for (unsigned long long st = starts; st; ++seen) {
    const int i = __ffsll((long long)st) - 1;
    st &= st - 1;
    const uint64_t lo = w * 64 + (uint64_t)i;
    const bool left_in = i > 0 ? ((A >> (i - 1)) & 1ull) : prev_top;
    const uint64_t e = chunk_end(P, w, A, i);
    if (seen == 0) e_first = e;
    if (seen == 1) e_second = e;
    const uint32_t c = visits_class(lo, e, left_in);
    if (c < 4) cnt0 += 1ull << (16 * c); else cnt1 += 1ull << (16 * (c - 4));
  }
  // exclusive scan of the packed per-class counts over the workgroup (one atomic per workgroup)
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  unsigned long long inc0 = cnt0, inc1 = cnt1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long u0 = __shfl_up(inc0, d), u1 = __shfl_up(inc1, d);
    if (lane >= (uint32_t)d) { inc0 += u0; inc1 += u1; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) my_hits += __shfl_xor(my_hits, d);
  if (lane == 63) { wave_cnt[wv][0] = inc0; wave_cnt[wv][1] = inc1; wave_hits[wv] = my_hits; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot[kLenClasses] = {0, 0, 0, 0, 0, 0, 0, 0}, hits = 0, total = 0;
    for (int v = 0; v < kBuildWaves; ++v) {
      for (int c = 0; c < kLenClasses; ++c) tot[c] += (uint32_t)((wave_cnt[v][c >> 2] >> (16 * (c & 3))) & 0xFFFFu);
      hits += wave_hits[v];
    }
    for (int c = 0; c < kLenClasses; ++c) total += tot[c];
    uint32_t base = total ? atomicAdd(P.desc_count, total) : 0u;
    // the longest chunks first: their waves start first and do not trail behind the kernel's end
    for (int c = kLenClasses - 1; c >= 0; --c) { class_base[c] = base; base += tot[c]; }
    if (hits) atomicAdd(P.hit_count, (unsigned long long)hits);
  }
  __syncthreads();
  // this thread's first slot in every class: class base + the waves before it + the lanes before it
  uint32_t next[kLenClasses];
  {
    const unsigned long long ex0 = inc0 - cnt0, ex1 = inc1 - cnt1;
#pragma unroll
    for (int c = 0; c < kLenClasses; ++c) {
      uint32_t x = class_base[c] + (uint32_t)(((c < 4 ? ex0 : ex1) >> (16 * (c & 3))) & 0xFFFFu);
      for (uint32_t v = 0; v < wv; ++v) x += (uint32_t)((wave_cnt[v][c >> 2] >> (16 * (c & 3))) & 0xFFFFu);
      next[c] = x;
    }
  }
  seen = 0;
  for (unsigned long long st = starts; st; ++seen) {
    const int i = __ffsll((long long)st) - 1;
    st &= st - 1;
    const uint64_t lo = w * 64 + (uint64_t)i;
    const bool left_in = i > 0 ? ((A >> (i - 1)) & 1ull) : prev_top;
    const uint64_t e = seen == 0 ? e_first : seen == 1 ? e_second : chunk_end(P, w, A, i);
    const uint32_t c = visits_class(lo, e, left_in);
    uint32_t idx = 0;
#pragma unroll
    for (int q = 0; q < kLenClasses; ++q)  // (register array: select, do not index)
      if ((uint32_t)q == c) { idx = next[q]; next[q] = idx + 1; }
    if (idx < P.desc_cap) {
      ChunkDesc d;
      d.own_lo = (uint32_t)lo;
      d.own_hi = (uint32_t)e;
      d.flags = left_in ? 0u : kDescClearBefore;
      d.pad_ = 0;
      P.desc[idx] = d;
    }
  }
}

// ------------------------------------------------------------------ the counting filter's own chunk list
// filter_count_kernel<.., DIRECT> leaves its chunk descriptors in one region of kRegionSlots slots per wave and the
// number of each region's entries in region_count (count_filter.hip).  This packs them into the dense list the list
// kernels read: a workgroup per 1024 regions -- the entries in front of its regions it counts itself (a few thousand
// numbers), a scan inside the workgroup gives every region its place.  A region that holds more runs than slots has
// lost some: the fuse word tells the host, which runs that search with the bitmap and the chunk builder instead.
__global__ __launch_bounds__(1024) void compact_chunks_kernel(const ChunkDesc* __restrict__ regions, const uint32_t* __restrict__ region_count,
                                                              uint32_t n_regions, ChunkDesc* __restrict__ desc, uint32_t* desc_count,
                                                              uint32_t desc_cap, uint32_t* fuse_word) {
  __shared__ uint32_t part[16];
  __shared__ uint32_t wave_tot[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t r0 = blockIdx.x * 1024u;
  // entries of the regions in front of this workgroup's
  uint32_t before = 0;
  bool over = false;
  for (uint32_t r = tid; r < r0; r += 1024u) {
    const uint32_t c = region_count[r];
    before += c < kRegionSlots ? c : kRegionSlots;
  }
  const uint32_t r = r0 + tid;
  uint32_t c = r < n_regions ? region_count[r] : 0u;
  if (c > kRegionSlots) { over = true; c = kRegionSlots; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
  uint32_t inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= (uint32_t)d) inc += t;
  }
  if (lane == 63) { part[wv] = before; wave_tot[wv] = inc; }
  __syncthreads();
  uint32_t off = 0;
  for (uint32_t v = 0; v < 16; ++v) off += part[v] + (v < wv ? wave_tot[v] : 0u);
  off += inc - c;
  const ChunkDesc* src = regions + (size_t)r * kRegionSlots;
  for (uint32_t i = 0; i < c; ++i)
    if (off + i < desc_cap) desc[off + i] = src[i];
  if (over) atomicOr(fuse_word, kFuseOverflow);
  // the last region's thread knows the total (the count keeps counting past the capacity, as the builder's does)
  if (r + 1 == n_regions) *desc_count = off + c;
}

// ------------------------------------------------------------------ report ranking
// The scan kernels append reports with an atomic counter, i.e. in arbitrary order; the result
// order is by end position (reference: matches of one strand come out by increasing end,
// src/search.rs:1180-1199).  Reports are few (thousands): every report counts the reports that
// precede it.  The count^2 comparisons are spread over the chip as (256 reports) x (256 others)
// tiles, one workgroup per tile, partial ranks added with one atomic per report and tile; a
// second pass moves every report to its rank.  Above kRankLimit the host sorts instead.
__global__ __launch_bounds__(256) void rank_count_kernel(const Candidate* __restrict__ cand,
                                                         const uint32_t* __restrict__ count_p, uint32_t cap,
                                                         uint32_t* __restrict__ rank) {
  __shared__ uint64_t tile[256];
  uint32_t count = *count_p;
  if (count > cap) count = cap;
  if (count > kRankLimit) return;
  const uint32_t nb = (count + 255) / 256;
  const uint32_t npairs = nb * nb;
  for (uint32_t p = blockIdx.x; p < npairs; p += gridDim.x) {
    const uint32_t cb = p / nb, sb = p - cb * nb;
    const uint32_t c = cb * 256 + threadIdx.x;
    const uint64_t mine = c < count ? cand[c].pos : ~0ull;
    const uint32_t x0 = sb * 256;
    __syncthreads();
    tile[threadIdx.x] = x0 + threadIdx.x < count ? cand[x0 + threadIdx.x].pos : ~0ull;
    __syncthreads();
    uint32_t r = 0;
    // every lane reads the same LDS word (broadcast); the ~0 padding never precedes a report
#pragma unroll 8
    for (uint32_t x = 0; x < 256; ++x) {
      const uint64_t q = tile[x];
      r += (q < mine || (q == mine && x0 + x < c)) ? 1u : 0u;
    }
    if (c < count && r) atomicAdd(&rank[c], r);
  }
}

// Index of the text that holds (or is followed by the separator that holds) buffer position pos:
// the largest t with start[t] <= pos (0 if pos lies before the first text).
__device__ __forceinline__ uint32_t text_of(const TextTable& T, uint64_t pos) {
  uint32_t lo = 0, hi = T.n;  // invariant: start[lo] <= pos < start[hi]
  while (lo + 1 < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (T.start[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

// Also the hand-over to the host: the first host_cap reports in result order and the 64-byte
// control block (counts, counters -- final once this kernel runs) are written straight into the
// caller's pinned, device-mapped staging buffer, so that no copy has to be queued behind the kernels.
// Multi-text buffers: the report learns its text here.  A report that ends inside the separator
// behind text t can only be the right end of a plateau that started at or before the text's end
// (costs never decrease across characters that match nothing, and after k of them the cost
// exceeds k): it stands for the end-of-text report of text t (reference: src/search.rs:1352-1366)
// and is moved there; in search_all mode such positions do not exist in the single-text search and
// are dropped.
__global__ __launch_bounds__(256) void rank_scatter_kernel(const Candidate* __restrict__ cand,
                                                           const uint32_t* __restrict__ count_p, uint32_t cap,
                                                           const uint32_t* __restrict__ rank,
                                                           Candidate* __restrict__ sorted,
                                                           Candidate* __restrict__ host_sorted, uint32_t host_cap,
                                                           uint4* __restrict__ host_ctl, const TextTable texts) {
  uint32_t count = *count_p;
  if (blockIdx.x == 0 && threadIdx.x < 4 && host_ctl)
    host_ctl[threadIdx.x] = reinterpret_cast<const uint4*>(count_p)[threadIdx.x];
  if (count > cap) count = cap;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= count) return;
  const uint32_t r = count > kRankLimit ? c : rank[c];
  Candidate v = cand[c];
  if (texts.n && !texts.per_text) {
    const uint32_t t = text_of(texts, v.pos);
    const uint64_t te = texts.start[t] + texts.len[t];
    v.flags = (v.flags & 0xFFu) | (t << kCandTextShift);
    if (v.pos > te) {
      if (texts.all_minima) v.flags |= kCandDrop;
      else v.pos = te;
    }
  }
  sorted[r] = v;
  if (r < host_cap) host_sorted[r] = v;
}

// ------------------------------------------------------------------ per-text reversal
// Block-aligned multi-text buffer (per-text mode): every text reversed inside its own slot, the
// padding behind it kept in place.  blk2text maps each 64-byte block to the text it belongs to.
// A thread writes 16 bytes (the slots are whole 64-byte blocks: the 16 belong to one text): where they all lie inside the
// text they are 16 consecutive source bytes, byte-swapped (a byte per thread was 0.84 ms for 330 MB of reads).
__global__ __launch_bounds__(256) void reverse_texts_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            uint64_t n, const uint32_t* __restrict__ blk2text,
                                                            const uint64_t* __restrict__ start,
                                                            const uint64_t* __restrict__ len, uint32_t pad) {
  const uint64_t n16 = n / 16;  // (n is a multiple of 64)
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = g * 16;
    const uint32_t t = blk2text[i >> 6];
    const uint64_t s0 = start[t], l = len[t], off = i - s0;
    uint4 out;
    if (off + 16 <= l) {
      uint32_t w[4];
      __builtin_memcpy(w, src + s0 + l - 16 - off, 16);  // (unaligned: four dword loads)
      out = make_uint4(__builtin_bswap32(w[3]), __builtin_bswap32(w[2]), __builtin_bswap32(w[1]), __builtin_bswap32(w[0]));
    } else {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t c = off + j < l ? src[s0 + l - 1 - (off + j)] : (pad & 0xFFu);
        w[j >> 2] |= c << (8 * (j & 3));
      }
      out = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(dst + i) = out;
  }
}

// ------------------------------------------------------------------ "is the text plain ACGT?"
// Sets *flag when some byte is not one of A C G T a c g t.  On such text the Iupac profile computes
// exactly what the Dna profile computes (same Eq relation, same is_match), and the Dna kernels are
// cheaper (two bit planes instead of a five-plane table lookup), so multi-pattern searches test
// the text once and then run the Dna kernels.  Same SWAR test as filter_table_kernel: the byte must
// equal the letter its 2-bit code stands for.
// allow_x: 'X' bytes (the separators of a multi-text buffer) pass too.
__global__ __launch_bounds__(256) void acgt_check_kernel(const uint4* __restrict__ text16, uint64_t n16,
                                                         const uint8_t* __restrict__ tail, uint32_t n_tail,
                                                         uint32_t* __restrict__ flag, int allow_x) {
  uint32_t bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = text16[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t sel = (w[d] >> 1) & 0x03030303u;
      const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel);  // 'A' 'C' 'T' 'G' by code
      uint32_t diff = (w[d] & 0xDFDFDFDFu) ^ expect;
      if (allow_x) {  // bytes equal to 'X': zero-byte test on w ^ 'XXXX', widened to byte masks
        const uint32_t z = w[d] ^ 0x58585858u;
        const uint32_t is_x = (~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u) >> 7;  // (exact per byte: no borrows)
        diff &= ~(is_x * 0xFFu);
      }
      bad |= diff;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < n_tail) {
    const uint32_t c = tail[threadIdx.x] & 0xDFu;
    bad |= (c != 'A' && c != 'C' && c != 'G' && c != 'T' && !(allow_x && tail[threadIdx.x] == 'X')) ? 1u : 0u;
  }
  if (bad) *flag = 1u;
}

// ------------------------------------------------------------------ N counting (max_n_frac)
// count[i] = number of 'N' / 'n' bytes in text[range[2i] .. range[2i+1]) -- the input of the
// reference's N-fraction filters (src/n_filter.rs:8-60) when the text lives on the device.
__global__ __launch_bounds__(256) void count_n_kernel(const uint8_t* __restrict__ text,
                                                      const uint64_t* __restrict__ range, uint32_t n,
                                                      uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t c = 0;
  for (uint64_t x = range[2 * i]; x < range[2 * i + 1]; ++x) c += ((text[x] | 0x20u) == 'n') ? 1u : 0u;
  count[i] = c;
}

// ------------------------------------------------------------------ launchers
hipError_t launch_reverse_texts(const uint8_t* d_src, uint8_t* d_dst, uint64_t n, const uint32_t* d_blk2text,
                                const uint64_t* d_start, const uint64_t* d_len, uint32_t pad, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(reverse_texts_kernel, dim3(8192), dim3(256), 0, stream, d_src, d_dst, n, d_blk2text, d_start, d_len, pad);
  return hipGetLastError();
}

hipError_t launch_acgt_check(const uint8_t* d_text, uint64_t n, uint32_t* d_flag, hipStream_t stream, int allow_x) {
  const uint64_t n16 = n / 16;
  hipLaunchKernelGGL(acgt_check_kernel, dim3(4096), dim3(256), 0, stream, reinterpret_cast<const uint4*>(d_text), n16,
                     d_text + n16 * 16, (uint32_t)(n - n16 * 16), d_flag, allow_x);
  return hipGetLastError();
}

hipError_t launch_count_n(const uint8_t* d_text, const uint64_t* d_range, uint32_t n, uint32_t* d_count,
                          hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(count_n_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_text, d_range, n, d_count);
  return hipGetLastError();
}

// d_rank: cap zeroed counters (zeroed by the caller together with its control block).
// h_sorted / h_ctl: device-mapped pinned host memory (first host_cap sorted reports, 64-byte
// control block); d_count points at the control block.
hipError_t launch_rank(const Candidate* d_cand, const uint32_t* d_count, uint32_t cap, uint32_t* d_rank,
                       Candidate* d_sorted, Candidate* h_sorted, uint32_t host_cap, void* h_ctl,
                       const TextTable& texts, hipStream_t stream) {
  if (cap == 0) return hipSuccess;
  // the count lives on the device: fixed grid, surplus workgroups exit at once
  hipLaunchKernelGGL(rank_count_kernel, dim3(1024), dim3(256), 0, stream, d_cand, d_count, cap, d_rank);
  hipLaunchKernelGGL(rank_scatter_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, d_cand, d_count, cap,
                     d_rank, d_sorted, h_sorted, host_cap, reinterpret_cast<uint4*>(h_ctl), texts);
  return hipGetLastError();
}

hipError_t launch_generate_dna(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first,
                               hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint64_t nb = (first + n + 31) / 32 - first / 32;
  uint64_t blocks = (nb + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(generate_dna_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_text, n, seed, first);
  return hipGetLastError();
}

hipError_t launch_generate_genome_like(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first, int with_n,
                                       hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(generate_genome_like_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_text, n, seed, first, with_n);
  return hipGetLastError();
}

hipError_t launch_scatter_bytes(uint8_t* d_text, uint64_t n, uint64_t first, const uint64_t* d_pos,
                                const uint8_t* d_val, uint64_t count, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_bytes_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, stream,
                     d_text, n, first, d_pos, d_val, count);
  return hipGetLastError();
}

hipError_t launch_reverse(const uint8_t* d_in, uint8_t* d_out, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = ((n + 15) / 16 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(reverse_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_in, d_out, n);
  return hipGetLastError();
}

hipError_t launch_compact_chunks(const ChunkDesc* d_regions, const uint32_t* d_region_count, uint32_t n_regions, ChunkDesc* d_desc,
                                 uint32_t* d_desc_count, uint32_t desc_cap, uint32_t* d_fuse_word, hipStream_t stream) {
  if (n_regions == 0) return hipSuccess;
  hipLaunchKernelGGL(compact_chunks_kernel, dim3((n_regions + 1023u) / 1024u), dim3(1024), 0, stream, d_regions, d_region_count, n_regions,
                     d_desc, d_desc_count, desc_cap, d_fuse_word);
  return hipGetLastError();
}

hipError_t launch_build_chunks(const unsigned long long* d_hit, uint64_t n_words, uint64_t n_blocks,
                               uint64_t first_owned, uint32_t wb, uint32_t L, uint32_t maxlen,
                               ChunkDesc* d_desc, uint32_t* d_desc_count, uint32_t desc_cap,
                               unsigned long long* d_hit_count, hipStream_t stream) {
  if (n_words == 0) return hipSuccess;
  BuildParams P;
  P.hit = d_hit; P.n_words = n_words; P.n_blocks = n_blocks; P.first_owned = first_owned;
  P.wb = wb; P.L = L; P.maxlen = maxlen; P.desc = d_desc; P.desc_count = d_desc_count; P.desc_cap = desc_cap;
  P.hit_count = d_hit_count;
  hipLaunchKernelGGL(build_chunks_kernel, dim3((uint32_t)((n_words + 1023) / 1024)), dim3(1024), 0, stream, P);
  return hipGetLastError();
}

}  // namespace sassy_hip
