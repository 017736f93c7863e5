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

// DIRECT: the lanes file the runs of candidate blocks they find as chunk descriptors themselves (no hit bitmap with its
// 6 MB memset per 3 GB, no chunk builder over it: 27 + 5 us of a lone config-3 search).  Without a single atomic: a
// returning atomic on the list's counter from inside the stream cost the kernel 55 us (3 000 of them: the wave waits for
// the atomic behind its own prefetch in the memory pipeline).  Every WAVE owns kRegionSlots slots of the list (region =
// its index in the launch): the lanes that file in one step rank themselves by ballot, the stores need no reply, and the
// wave leaves its count in region_count when it is done (every wave does: nothing to clear).  compact_chunks_kernel
// (aux_kernels.hip) then packs the regions into the dense list the list kernels read; a wave with more runs than slots
// leaves its true count and the packer raises the fuse word: the host runs the classic chain for that search.
// A chunk = [lo, hi), a run of candidate blocks, never with kDescClearBefore: the list kernel warms up on the wb blocks in
// front of it (the chunk builder makes them part of the chunk instead -- the same blocks, the same start).
// WPG: waves per workgroup.  The q-gram table is per workgroup: sixteen waves around one copy leave room for sixteen
// waves per CU (four around each of three copies: twelve).
template <int Q, int R, int SB, int WPG, bool DIRECT = false>
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
  // DIRECT: the run of candidate blocks the lane is collecting -- run_lo = its first block (kNoRun: none); run_st bit 0 /
  // bit 1 = the window sum of the last block / the block in front of it reached the threshold
  constexpr uint32_t kNoRun = 0xFFFFFFFFu;
  uint32_t run_lo = kNoRun, run_st = 0;
  // (DIRECT) the wave's slots in the descriptor list (scalars: the wave index comes from the thread index, which the compiler
  // does not know to be wave-uniform)
  const uint32_t region = blockIdx.x * WPG + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
  uint32_t region_n = 0;                                       // ... and how many it has used (wave-uniform)
  uint4 nxt[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    nxt[i] = make_uint4(0u, 0u, 0u, 0u);
    if (interior) nxt[i] = stream_load16<SASSY_NT_COUNT>(text_base + soff[i]);
  }

  // DIRECT: the run logic for block bp, whose window sum is bit 0 of run_st (bit 1: the block in front of it).  Run at the
  // top of the NEXT iteration, where the block's sixteen text words are no longer in registers (behind the look-ups the
  // descriptor store pushed the sixteen-wave kernel over its 128 registers: a scratch segment), and only when some lane
  // has anything to decide -- nearly every block of nearly every wave: one vote, and on with the stream.
  // (block indices as 32-bit words here: the host asks for this variant below 2^31 blocks; a lane without blocks files
  // nothing: an empty range)
  const uint32_t blk0_32 = (uint32_t)blk0;
  const uint32_t f_hi = has_chunk ? (uint32_t)min(own_hi + 1, P.n_blocks) : 0u;
  const uint32_t f_lo = has_chunk ? (uint32_t)own_lo + (chunk == 0 ? 0u : 1u) : 1u;
  auto direct_step = [&](uint32_t bp) {  // bp: the block's iteration
    if (__any(run_st != 0u || run_lo != kNoRun)) {
      // Block bp is a candidate by its own window sum or by that of the block in front (the report rule looks one column
      // ahead).  A lane files the blocks (own_lo, own_hi] -- its last block's successor, the next lane's first block,
      // included (it looks at that block's window too: one iteration more), its own first block left to the lane in front,
      // whose window sums decide about it.  (The launch's first lane also files its first block: nothing lies in front.)
      const uint32_t b32 = blk0_32 + bp;
      const bool own = b32 >= f_lo && b32 < f_hi;
      const bool cand = own && (run_st & 3u) != 0u && b32 >= (uint32_t)P.dp_first_owned;
      if (cand && run_lo == kNoRun) run_lo = b32;
      uint32_t file_hi = 0;
      if (own && run_lo != kNoRun) {
        // the run ends in front of this block, or with the lane's last block, or is cut (a long run leaves in pieces)
        if (!cand) file_hi = b32;
        else if (b32 + 1u == f_hi || b32 + 1u - run_lo >= P.count_maxlen) file_hi = b32 + 1u;
      }
      // (wave-uniform here: the lanes that file rank themselves by ballot -- no counter in memory)
      const unsigned long long filing = __ballot(file_hi != 0u);
      if (file_hi != 0u) {
        const uint32_t idx = region_n + (uint32_t)__popcll(filing & ((1ull << lane) - 1ull));
        if (idx < kRegionSlots) {
          ChunkDesc d;
          d.own_lo = run_lo;
          d.own_hi = file_hi;
          d.flags = 0u;
          d.pad_ = 0;
          const_cast<ChunkDesc*>(P.desc)[(size_t)region * kRegionSlots + idx] = d;
        }
        run_lo = kNoRun;
      }
      region_n += (uint32_t)__popcll(filing);
      run_st = (run_st << 1) & 2u;
    }
  };
  for (uint32_t it = 0; it < P.n_iter; ++it) {
    if constexpr (DIRECT) {
      if (it) direct_step(it - 1u);
    }
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
          // (the wave at the buffer's end: the offset through an opaque copy -- hoisted out of the loop these eight 64-bit
          // addresses cost the sixteen-wave kernel more registers than it has)
          uint32_t so = soff[i];
          asm volatile("" : "+v"(so));
          const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + so;
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
    if constexpr (DIRECT) {
      run_st |= reach ? 1u : 0u;  // (filed at the top of the next iteration: direct_step)
    } else if (evaluate && reach) {
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
  if constexpr (DIRECT) {
    direct_step(P.n_iter - 1u);
    if (lane == 0) const_cast<uint32_t*>(P.region_count)[region] = region_n;
  }
}

template <int Q, int R, int SB, int WPG, bool DIRECT = false>
hipError_t launch_qr(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = ((size_t)1 << (2 * (Q + R - 1))) + (size_t)WPG * P.lds_per_wave;
  static DeviceOnce attr_set;  // LDS beyond the 64 KiB default needs an explicit opt-in
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_count_kernel<Q, R, SB, WPG, DIRECT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.done();
  }
  hipLaunchKernelGGL((filter_count_kernel<Q, R, SB, WPG, DIRECT>), dim3(grid), dim3(64 * WPG), smem, stream, P);
  return hipGetLastError();
}
template <int Q, int R>
hipError_t launch_sb(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  if (P.count_direct)  // (whole lines, sixteen or four waves per workgroup; one strand)
    return P.waves_per_group == 16 ? launch_qr<Q, R, 2, 16, true>(P, grid, stream) : launch_qr<Q, R, 2, 4, true>(P, grid, stream);
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
