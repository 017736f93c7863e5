// K0, q-gram counting variant (gfx950): the prefilter for patterns the bit-plane filter does not take
// (Iupac patterns, more than 8 pieces) and for long patterns in general.
//
// q-gram lemma (Jokinen & Ukkonen 1991): an occurrence of a pattern of m rows with at most k edits
// keeps at least  t = (m - Q + 1) - k Q  of the pattern's Q-grams intact -- every edit destroys at
// most Q of them -- and the intact ones end at t different text positions inside the occurrence.
// With the reference's cost model (unit cost substitution / insertion / deletion, a row matches a
// text letter if the profile says so: src/profiles/*.rs) "intact" means: Q consecutive text letters
// that the Q pattern rows accept.  So a match can only END in text block b if the blocks
// b-W+1 .. b (W = ceil((m + k - Q) / 64) + 1, the blocks an occurrence that ends in b can touch
// with a Q-gram end) hold at least t positions where some pattern Q-gram ends.  The pigeonhole
// filter of the other kernels is the case t >= 1 with Q = m / (k+1); a smaller Q with a large t is
// far more selective (m = 32, k = 3, Q = 6: t = 9 against 0.8 expected chance hits per window), which
// leaves next to nothing for the chunk DP behind it.
//
// The host builds one byte per (Q+R-1)-gram: how many of the R Q-grams it ends with occur in the
// pattern (ambiguous pattern letters expanded).  Every lane walks its consecutive blocks, keeps the
// 2-bit codes of the last text letters in a rolling register, looks up one byte per R positions
// (LDS, shared by the workgroup), keeps the counts of its last W blocks in a small LDS ring, and
// marks block b and b+1 (the report rule looks one column ahead) when the window sum reaches t.
// The 2-bit code (c >> 1) & 3 is exact for A C G T U in either case and is the Dna profile's own
// definition of a text byte (src/profiles/dna.rs:19-40); under the Iupac profile a block that holds
// any other byte counts as t hits (its letters may match more than their code says).
#include <hip/hip_runtime.h>

#include "common.h"

namespace sassy_hip {
namespace {

__device__ __forceinline__ uint64_t chunk_first_block(uint64_t first_owned, uint32_t bpl, uint32_t back, uint64_t chunk) {
  const uint64_t start = first_owned + chunk * (uint64_t)bpl;
  return start > back ? start - back : 0;
}

// 16 text bytes that straddle or lie past the end of the buffer (cold path); bytes past the end
// read as 'X' (reference: src/search.rs:202-207)
__device__ __noinline__ uint4 tail16(const uint8_t* text, uint64_t off, uint64_t text_len) {
  uint32_t w[4] = {0x58585858u, 0x58585858u, 0x58585858u, 0x58585858u};
  if (off < text_len) {
    const uint32_t valid = (uint32_t)min((uint64_t)16, text_len - off);
#pragma unroll 1
    for (uint32_t q = 0; q < valid; ++q) {
      const uint32_t sh = 8u * (q & 3u);
      w[q >> 2] = (w[q >> 2] & ~(0xFFu << sh)) | ((uint32_t)text[off + q] << sh);
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// WPG: waves per workgroup.  The q-gram table is per workgroup: sixteen waves around one copy leave room for sixteen
// waves per CU (four around each of three copies: twelve).
template <int Q, int R, int SB, int WPG>
__global__ __launch_bounds__(64 * WPG) void filter_count_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kTableBytes = 1u << (2 * (Q + R - 1));
  constexpr uint32_t kRowBytes = 64u * SB;
  constexpr uint32_t kSlots = 4u * SB;
  constexpr uint32_t kOwnersPerInstr = 64u / kSlots;
  constexpr int kStageInstr = 4 * SB;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* table = smem;
  unsigned char* tile = smem + kTableBytes + (size_t)wave * P.lds_per_wave;
  unsigned char* ring = tile + 4096u * SB + lane;  // [slot][lane] bytes
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.qgram_table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (uint32_t x = threadIdx.x; x < kTableBytes / 16; x += blockDim.x) dst[x] = src[x];
  }
  const uint32_t W = P.count_window;
  for (uint32_t s = 0; s < W; ++s) ring[s * 64] = 0;
  __syncthreads();

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * WPG + wave) * kWave;
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;
  const uint32_t bpl = P.bpl;
  const uint64_t first_owned = P.first_owned_block;
  // W blocks in front of the owned ones: one to fill the rolling code, W - 1 to fill the window
  // (+ one if that makes the first block even: a staged pair is then one aligned 128-byte line)
  const uint32_t back = W + (uint32_t)((first_owned + W) & 1u);
  const uint64_t own_lo = first_owned + chunk * (uint64_t)bpl;
  uint64_t own_hi = own_lo + bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_first_block(first_owned, bpl, back, chunk);

  const uint64_t wave_blk0 = chunk_first_block(first_owned, bpl, back, wave_chunk0);
  const uint8_t* text_base = P.text + wave_blk0 * 64;
  // staging as in the other streaming kernels: every global_load_dwordx4 fetches whole 16-byte
  // pieces of kOwnersPerInstr lanes' rows, swizzled so that the owners' ds_read_b128 are conflict free
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ (SB == 2 ? ((owner >> 1) & 7u) : ((owner >> 2) & 3u));
    soff[i] = (uint32_t)((chunk_first_block(first_owned, bpl, back, wave_chunk0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  const uint64_t wave_last = chunk_first_block(first_owned, bpl, back, wave_chunk0 + 63) + P.n_iter + 2;
  const bool interior = wave_last * 64 <= P.text_len;
  const uint32_t fsw = SB == 2 ? ((lane >> 1) & 7u) : ((lane >> 2) & 3u);
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ (fsw & 3u)) << 4);
  const bool check_text = P.profile == PROFILE_IUPAC;  // wave-uniform
  const uint32_t thresh = P.count_thresh;

  uint32_t h2 = 0;       // twice the 2-bit codes of the last text letters, newest lowest
  uint32_t sum = 0;      // hits in the last W blocks
  uint32_t pos = 0;      // ring slot of the oldest block (wave-uniform)
  uint32_t forced = 0;   // blocks (this one included) whose window still holds a block with such a byte
  uint4 nxt[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    nxt[i] = make_uint4(0u, 0u, 0u, 0u);
    if (interior) nxt[i] = stream_load16<SASSY_NT_COUNT>(text_base + soff[i]);
  }

  for (uint32_t it = 0; it < P.n_iter; ++it) {
    const uint32_t sub = SB == 2 ? (it & 1u) : 0u;
    if (sub == 0) {
      if (interior) {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = nxt[i];
        if (it + SB < P.n_iter) {
#pragma unroll
          for (int i = 0; i < kStageInstr; ++i)
            nxt[i] = stream_load16<SASSY_NT_COUNT>(text_base + (uint64_t)(it + SB) * 64 + soff[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
          uint4 v;
          if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
          else v = tail16(P.text, off, P.text_len);
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      }
    }
    const uint32_t hs = SB == 2 ? (((sub << 2) ^ (fsw & 4u)) << 4) : 0u;
    uint32_t x[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
      x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      // four letters -> 8 bits, first letter highest: the byte values are 2 * code, so the
      // weights 64 16 4 1 give twice the packed code
      // (h2 = twice the rolling code: the dot product accumulates onto the shifted register, and the
      // factor two comes off in the bit-field extract of the lookup index)
      h2 = __builtin_amdgcn_udot4(x[d] & 0x06060606u, 0x01041040u, h2 << 8, false);
#pragma unroll
      for (int j = 0; j < 4 / R; ++j) {
        const uint32_t shift = 2u * (4u - (uint32_t)(j + 1) * R);
        cnt += table[__builtin_amdgcn_ubfe(h2, shift + 1u, 2 * (Q + R - 1))];
      }
    }
    uint32_t bad = 0;
    if (check_text) {
      // bytes other than A C G T U (either case): compare with the letter their code stands for
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const uint32_t sel = (x[d] >> 1) & 0x03030303u;
        const uint32_t e1 = __builtin_amdgcn_perm(0u, 0x47544341u, sel);   // 'A' 'C' 'T' 'G' by code
        // what may differ from that letter: the case bit, and for code 2 bit 0 ('U' = 'T' + 1)
        const uint32_t ok1 = __builtin_amdgcn_perm(0u, 0xDFDEDFDFu, sel);
        bad |= (x[d] ^ e1) & ok1;
      }
    }
    // a q-gram that holds the bad byte ends in its block or (Q <= 64) the next one: W + 1 windows
    if (bad != 0) forced = W + 1;
    sum += cnt - (uint32_t)ring[pos * 64];
    ring[pos * 64] = (unsigned char)cnt;   // <= 64
    pos = pos + 1 == W ? 0u : pos + 1;
    const uint64_t b = blk0 + it;
    const bool evaluate = has_chunk && b >= own_lo && b < own_hi;
    const bool reach = sum >= thresh || forced != 0;
    if (forced != 0) --forced;
    if (evaluate && reach) {
      atomicOr(&P.hit_bitmap[b >> 6], 1ull << (b & 63));
      if (b + 1 < P.n_blocks) atomicOr(&P.hit_bitmap[(b + 1) >> 6], 1ull << ((b + 1) & 63));
      if (P.count_rc) {
        // The table also holds the Rc strand's q-grams (reversed).  An Rc match that ends at column
        // c of the REVERSED text covers forward positions [n - c, n - c + m + k); all its q-gram
        // ends lie in this window when b is the block of forward position n - c + m + k - 1 (or
        // the last block, for matches that start within m + k of the reversed text's start).
        const int64_t n = (int64_t)P.text_len;
        const int64_t top = n + (int64_t)P.m + (int64_t)P.k - 1;
        int64_t c_lo = top - (int64_t)(b * 64) - 63;       // reversed end columns whose window ends in b
        int64_t c_hi = top - (int64_t)(b * 64) + 1;         // (+ 1: the look-ahead column)
        if (b + 1 == P.n_blocks) c_lo = 1;
        if (c_lo < 1) c_lo = 1;
        if (c_hi >= 1) {
          uint64_t blo = (uint64_t)(c_lo - 1) >> 6, bhi = (uint64_t)(c_hi - 1) >> 6;
          if (bhi >= P.n_blocks) bhi = P.n_blocks - 1;
          for (uint64_t x = blo; x <= bhi; ++x) atomicOr(&P.hit_bitmap_rc[x >> 6], 1ull << (x & 63));
        }
      }
    }
  }
}

template <int Q, int R, int SB, int WPG>
hipError_t launch_qr(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = ((size_t)1 << (2 * (Q + R - 1))) + (size_t)WPG * P.lds_per_wave;
  static DeviceOnce attr_set;  // LDS beyond the 64 KiB default needs an explicit opt-in
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_count_kernel<Q, R, SB, WPG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.done();
  }
  hipLaunchKernelGGL((filter_count_kernel<Q, R, SB, WPG>), dim3(grid), dim3(64 * WPG), smem, stream, P);
  return hipGetLastError();
}
template <int Q, int R>
hipError_t launch_sb(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  if (P.waves_per_group == 16) return P.stage_blocks == 2 ? launch_qr<Q, R, 2, 16>(P, grid, stream) : launch_qr<Q, R, 1, 16>(P, grid, stream);
  return P.stage_blocks == 2 ? launch_qr<Q, R, 2, 4>(P, grid, stream) : launch_qr<Q, R, 1, 4>(P, grid, stream);
}

}  // namespace

// (Q, R) variants: 16 KiB tables (7,1) (6,2), 4 KiB (5,2) (6,1), 64 KiB (7,2)
hipError_t launch_filter_count(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const uint32_t key = P.piece_len * 10u + P.count_r;
  switch (key) {
    case 52: return launch_sb<5, 2>(P, grid, stream);
    case 61: return launch_sb<6, 1>(P, grid, stream);
    case 62: return launch_sb<6, 2>(P, grid, stream);
    case 71: return launch_sb<7, 1>(P, grid, stream);
    case 72: return launch_sb<7, 2>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sassy_hip
