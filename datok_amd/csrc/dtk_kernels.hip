// dtk_kernels.hip -- gfx950 (MI355X, wave64) kernels of the batch tokenizer.
//
// Pipeline per batch (all on one HIP stream, no host round trip):
//   1. symbolise : bytes -> uint16 symbol stream (UTF-8 decode with Go's
//                  DecodeRune rules + sigma lookup), fully parallel, HBM bound.
//   2. walk      : the FSA transition walk of matrix.go:348-698 /
//                  datok.go:781-1135, one document per lane, writing one event
//                  byte per cursor position (no output allocation problem).
//   3. compact   : wave-per-document ballot / popcount / prefix-sum pass that
//                  turns event bytes into the offset arrays NewTokenWriter
//                  (token_writer.go:36-175) would have collected.  Pass 1
//                  counts, a scan sizes the CSR rows, pass 2 writes.
//
// Integer table lookups only: no MFMA.  The walk is a per-lane dependent-load
// chain (latency bound); 1 and 3 are streaming passes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dtk_internal.h"

#define WAVE 64
#ifndef DTK_WARM_TAG
#define DTK_WARM_TAG 64u
#endif  // how far behind a warm-up start an opening angle bracket is looked for
// knock-out builds for cost measurements (scripts/ko.sh): results are wrong, only timings mean something
#ifndef DTK_KO
#define DTK_KO 0
#endif

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int highest(unsigned long long m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint32_t popc(unsigned long long m) { return (uint32_t)__popcll(m); }

// exclusive prefix sum over the 64 lanes; total = sum of all lanes.  DPP row shifts inside the
// rows of 16 lanes, then the two row broadcasts (lane 15 -> next row, lane 31 -> upper half):
// six adds, no LDS traffic (a __shfl_up ladder is six ds_bpermute round trips).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total) {
  uint32_t x = v;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
  total = (uint32_t)__builtin_amdgcn_readlane((int)x, WAVE - 1);
  return x - v;
}

// ---------------------------------------------------------------- symbolise
//
// One wave per 4 KiB of input (staged in LDS), 512 bytes (8 per lane) per iteration.
//   light: every byte < 0x80 is a complete rune: its entry comes from a 128-entry
//          table in LDS and the lane's 8 entries leave as one 16-byte store.  Positions
//          holding a byte >= 0x80 (a few percent of European text) are appended to a
//          queue in LDS.
//   heavy: once per KiB, lane i takes the i-th queued position, decodes it with Go's
//          DecodeRune rules (matrix.go:392), decides whether that byte really
//          starts a rune (look-back of up to 3 bytes), looks the rune up in the sigma
//          map (runes < 256: a table; the others: binary search, both in LDS) and
//          overwrites that one entry.  Documents never share a rune: look-back
//          and look-ahead stop at the document boundary (reader EOF,
//          matrix.go:394-399).

// width Go's DecodeRune reports at a position (b0 first byte, `avail` bytes left in the
// document).  Integer predicates on purpose: bool && chains become branches.
__device__ __forceinline__ uint32_t go_width(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t avail) {
  const uint32_t c1 = (b1 & 0xC0u) == 0x80u, c2 = (b2 & 0xC0u) == 0x80u, c3 = (b3 & 0xC0u) == 0x80u;
  const uint32_t two = (uint32_t)(b0 - 0xC2u < 0x1Eu) & (uint32_t)(avail >= 2u) & c1;  // C2..DF
  const uint32_t lo3 = b0 == 0xE0u ? 0xA0u : 0x80u, hi3 = b0 == 0xEDu ? 0x9Fu : 0xBFu;
  const uint32_t three = (uint32_t)((b0 & 0xF0u) == 0xE0u) & (uint32_t)(avail >= 3u) & (uint32_t)(b1 >= lo3) &
                         (uint32_t)(b1 <= hi3) & c2;
  const uint32_t lo4 = b0 == 0xF0u ? 0x90u : 0x80u, hi4 = b0 == 0xF4u ? 0x8Fu : 0xBFu;
  const uint32_t four = (uint32_t)(b0 - 0xF0u <= 4u) & (uint32_t)(avail >= 4u) & (uint32_t)(b1 >= lo4) &
                        (uint32_t)(b1 <= hi4) & c2 & c3;
  return 1u + two + 2u * three + 3u * four;  // mutually exclusive
}

__device__ __forceinline__ uint32_t doc_of(const uint64_t *__restrict__ doc_off, uint32_t lo, uint32_t hi,
                                           uint64_t g) {
  // largest d in [lo, hi) with doc_off[d] <= g   (invariant: doc_off[lo] <= g < doc_off[hi])
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (doc_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

#define SYM_BLOCK_BYTES DTK_SYM_BLOCK_BYTES
#define SYM_TILE 512u
#define SYM_SIG_LDS 64u  // runes >= 256 of the sigma kept in LDS (40 in the shipped models)
#define SYM_HALF 1024u  // bytes per wave (its queue of bytes >= 0x80: 2 B of LDS per byte)
#define SYM_THREADS (WAVE * (SYM_BLOCK_BYTES / SYM_HALF))  // 256: four waves per 4 KiB block
#define SYM_DOFF (SYM_BLOCK_BYTES / 16u)  // document offsets of a block kept in LDS (documents of 16 bytes on average and longer)

// SYM8: the stream holds one code per byte (DtkSigmaDev's code table) instead of the 16-bit entries; lut / lat then
// hold codes too.
template <bool ALIGNED4, bool SYM8>
__global__ __launch_bounds__(SYM_THREADS) void k_symbolize(const uint8_t *__restrict__ text,
                                                    const uint64_t *__restrict__ doc_off,
                                                    uint32_t n_docs, uint64_t total, DtkSigmaDev sig,
                                                    void *__restrict__ sym_,
                                                    const uint32_t *__restrict__ blk_doc,
                                                    unsigned long long *__restrict__ n_invalid,
                                                    uint32_t *__restrict__ rs_bits,
                                                    uint32_t *__restrict__ ev_bits, uint32_t bit_words,
                                                    uint4 *__restrict__ acc, uint32_t acc16,
                                                    unsigned long long epoch) {
  // The run's accumulator block (totals, per-document counts, status and check words; dtk_batch_run) starts from
  // zero: the first blocks clear it here instead of a launch of its own in front (7 us of a batch's 230).  All but
  // totals[6], which blocks of this very launch write: the number of the last run that saw an invalid byte.
  if (acc) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t i = blockIdx.x * SYM_THREADS + threadIdx.x; i < acc16; i += gridDim.x * SYM_THREADS) {
      if (i == 3u) reinterpret_cast<unsigned long long *>(acc)[7] = 0ull;  // totals[7]; totals[6] stays
      else acc[i] = z;
    }
  }
  uint16_t *__restrict__ sym = static_cast<uint16_t *>(sym_);
  uint8_t *__restrict__ sym8 = static_cast<uint8_t *>(sym_);
  __shared__ uint32_t s_rs[SYM_BLOCK_BYTES / 32];  // bit i: byte i of the block starts a rune
  __shared__ uint16_t lut[128];       // symbol | class | width 1 for the runes < 128 (index = byte)
  __shared__ uint16_t lat[256];       // symbol | class for runes < 256 (heavy path, Latin-1)
  __shared__ uint32_t s_runes[SYM_SIG_LDS];   // sigma map (runes >= 256), if it fits
  __shared__ uint16_t s_syms[SYM_SIG_LDS];  // (SYM8: their codes)
  __shared__ uint32_t s_txt[SYM_BLOCK_BYTES / 4 + 4];  // the block's bytes, one dword of halo either side
  __shared__ uint16_t s_qs[SYM_BLOCK_BYTES / SYM_HALF][SYM_HALF];  // per wave: positions (offset in the block) of the bytes >= 0x80 of its quarter
  __shared__ uint64_t s_doff[SYM_DOFF];  // the offsets of the documents of this block (if they are that few)
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1u), half = tid >> 6;  // one wave per quarter (1 KiB) of the block
  uint16_t *s_q = s_qs[half];
  const bool sig_lds = sig.n_runes <= SYM_SIG_LDS;
  for (uint32_t i = tid; i < SYM_BLOCK_BYTES / 32; i += SYM_THREADS) s_rs[i] = 0;
  for (uint32_t i = tid; i < 256u; i += SYM_THREADS) {
    // matrix.go:421-426: runes < 256 go through sigmaASCII; rune 4 is EOT
    if (SYM8) {
      lat[i] = sig.code_lt256[i];  // (lut is not used: the code of a byte < 128 is the byte)
      if (sig_lds && i < sig.n_runes) { s_runes[i] = sig.runes[i]; s_syms[i] = sig.code_runes[i]; }
    } else {
      const uint32_t e = (sig.ascii[i] & DTK_SYM_MASK) | (i == DTK_EOT ? (1u << DTK_SYM_CLS_SHIFT) : 0u);
      lat[i] = (uint16_t)e;
      if (i < 128u) lut[i] = (uint16_t)(e | (1u << DTK_SYM_W_SHIFT));
      if (sig_lds && i < sig.n_runes) { s_runes[i] = sig.runes[i]; s_syms[i] = sig.syms[i]; }
    }
  }
  const uint64_t block_start = (uint64_t)blockIdx.x * SYM_BLOCK_BYTES;
  const uint32_t n_here = (uint32_t)min((uint64_t)SYM_BLOCK_BYTES, total - block_start);
  // documents that can own bytes of this block: host-computed (document of each block's
  // first byte), so no lane walks the offset table from scratch
  const uint32_t d_lo = blk_doc[blockIdx.x];
  const uint32_t d_hi = min(blk_doc[blockIdx.x + 1], n_docs - 1);
  // their offsets, doc_off[d_lo .. d_hi + 1], in LDS: a block of tiny documents otherwise searches the table in memory
  // once per byte >= 0x80 (six dependent loads each; 64-byte documents: 203 us of symbolising per 32 MiB)
  const uint32_t n_off = d_hi - d_lo + 2u;
  if (n_off <= SYM_DOFF && tid < n_off) s_doff[tid] = doc_off[d_lo + tid];
  {
    // all loads of the block are issued before anything waits for one of them
    auto load4 = [&](uint64_t g) -> uint32_t {  // bytes g..g+3, zero outside [0, total)
      if (ALIGNED4) return g < total ? *reinterpret_cast<const uint32_t *>(text + g) : 0u;
      uint32_t x = 0;
      for (int k = 0; k < 4; k++)
        if (g + k < total) x |= (uint32_t)text[g + k] << (8 * k);
      return x;
    };
    uint32_t v[SYM_BLOCK_BYTES / 4 / SYM_THREADS];
#pragma unroll
    for (uint32_t r = 0; r < SYM_BLOCK_BYTES / 4 / SYM_THREADS; r++) v[r] = load4(block_start + (r * SYM_THREADS + tid) * 4u);
    uint32_t halo = 0;
    if (tid == 0 && block_start >= 4) halo = load4(block_start - 4);
    if (tid == 1) halo = load4(block_start + SYM_BLOCK_BYTES);
#pragma unroll
    for (uint32_t r = 0; r < SYM_BLOCK_BYTES / 4 / SYM_THREADS; r++) s_txt[1 + r * SYM_THREADS + tid] = v[r];
    if (tid == 0) s_txt[0] = halo;
    if (tid == 1) s_txt[1 + SYM_BLOCK_BYTES / 4] = halo;
  }
  __syncthreads();
  const uint8_t *__restrict__ sb = reinterpret_cast<const uint8_t *>(s_txt) + 4;  // sb[i] = text[block_start + i]

  const uint64_t lo_start = doc_off[d_lo], lo_end = doc_off[d_lo + 1];  // the block's first document
  if (ev_bits) {
    // The walk's event bitmaps start from zero: every block clears the words of the cursor positions of its bytes
    // (bit = byte + document index, dtk_internal.h; neighbours overlap by a word or two), the last block the rest --
    // a few stores per lane here instead of a 10 MB clear kernel in front.
    const uint64_t ga = block_start + d_lo, gb = block_start + n_here + d_hi + 1u;
    const uint32_t wa = (uint32_t)(ga >> 5);
    uint32_t wb = (uint32_t)((gb + 31u) >> 5);
    if (wb > bit_words || blockIdx.x == gridDim.x - 1) wb = bit_words;
    for (uint32_t w = wa + tid; w < wb; w += SYM_THREADS)
#pragma unroll
      for (uint32_t k = 0; k < EVB_KINDS; k++) ev_bits[(size_t)k * bit_words + w] = 0u;
  }
  // (the light path's vector stores: 8 entries = 16 bytes, or 8 codes = 8 bytes)
  const bool sym16 = ((reinterpret_cast<uintptr_t>(sym_) + (SYM8 ? 1ull : 2ull) * block_start) & (SYM8 ? 7u : 15u)) == 0;

  // (the four quarters used to be one wave's four rounds: a chain of load -> light -> wait for the stores -> heavy,
  //  four times over, with four waves per SIMD to hide it; now the rounds are four waves)
  if (half * SYM_HALF < n_here) {
    // ---- light: every byte < 0x80 is a complete rune: its entry goes straight to memory
    //      (16-byte stores, 8 bytes of input per lane); the positions of the other bytes are queued
    //      (slots from a wave-uniform counter and four ballots: an LDS atomicAdd with per-lane
    //      values compiles to a loop over the active lanes)
    uint32_t qn = 0;
    auto light = [&](auto full_tag, uint32_t it) {
      constexpr bool FULL = decltype(full_tag)::value;  // a whole block, entries 16-byte aligned
      const uint32_t i0 = half * SYM_HALF + it * SYM_TILE + lane * 8u;  // my 8 bytes (offset in the block)
      const uint32_t w0 = s_txt[1 + (i0 >> 2)], w1 = s_txt[2 + (i0 >> 2)];
      uint32_t left = 8u;
      if (!FULL) left = i0 < n_here ? (n_here - i0 >= 8u ? 8u : n_here - i0) : 0u;
      if (SYM8) {
        // the code of a byte < 128 is the byte itself (upload()): the light path is a copy
        if (FULL || (left == 8u && sym16)) {
          *reinterpret_cast<uint2 *>(sym8 + block_start + i0) = make_uint2(w0, w1);
        } else {
          for (uint32_t j = 0; j < left; j++) sym8[block_start + i0 + j] = (uint8_t)((j < 4u ? w0 >> (8u * j) : w1 >> (8u * j - 32u)));
        }
      } else {
        const uint32_t e0 = lut[w0 & 0x7Fu], e1 = lut[(w0 >> 8) & 0x7Fu];
        const uint32_t e2 = lut[(w0 >> 16) & 0x7Fu], e3 = lut[(w0 >> 24) & 0x7Fu];
        const uint32_t e4 = lut[w1 & 0x7Fu], e5 = lut[(w1 >> 8) & 0x7Fu];
        const uint32_t e6 = lut[(w1 >> 16) & 0x7Fu], e7 = lut[(w1 >> 24) & 0x7Fu];
        if (FULL || (left == 8u && sym16)) {
          *reinterpret_cast<uint4 *>(sym + block_start + i0) =
              make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
        } else {
          const uint32_t o[8] = {e0, e1, e2, e3, e4, e5, e6, e7};
          for (uint32_t j = 0; j < left; j++) sym[block_start + i0 + j] = (uint16_t)o[j];
        }
      }
      // one bit per byte: bytes < 0x80 start a rune (the others are decided one by one below)
      auto nib = [](uint32_t x) { return ((x >> 7) & 1u) | ((x >> 14) & 2u) | ((x >> 21) & 4u) | ((x >> 28) & 8u); };
      const uint32_t valid = FULL ? 0xFFu : ((1u << left) - 1u);
      const uint32_t rare = (nib(w0 & 0x80808080u) | (nib(w1 & 0x80808080u) << 4)) & valid;
      const uint32_t asc = ~rare & valid;
      if (asc) atomicOr(&s_rs[i0 >> 5], asc << (i0 & 31u));
      // queue slots: exclusive prefix of the lanes' counts (0..8) from four ballots
      const uint32_t cnt = (uint32_t)__popc(rare);
      const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u),
                               b3 = __ballot(cnt & 8u);
      if ((b0 | b1 | b2 | b3) == 0ull) return;  // wave-uniform
      const unsigned long long lt = lanemask_lt();
      uint32_t slot = qn + popc(b0 & lt) + 2u * popc(b1 & lt) + 4u * popc(b2 & lt) + 8u * popc(b3 & lt);
      qn += popc(b0) + 2u * popc(b1) + 4u * popc(b2) + 8u * popc(b3);
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (rare & (1u << j)) s_q[slot++] = (uint16_t)(i0 + j);
    };
    const bool full_block = n_here == SYM_BLOCK_BYTES && sym16;  // wave-uniform
#pragma unroll 1
    for (uint32_t it = 0; it < SYM_HALF / SYM_TILE; it++) {
      if (half * SYM_HALF + it * SYM_TILE >= n_here) break;
      if (full_block) light(std::true_type{}, it); else light(std::false_type{}, it);
    }
    // (the queue is the wave's own: no block barrier, the wave's LDS operations complete in order)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nq = qn;  // wave-uniform
    // the heavy lanes overwrite single entries written above: those stores must have landed
    // (staging the block's entries in LDS instead costs more in occupancy than this wait: measured)
    if (nq) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): stores count in vmcnt on gfx950

    // ---- heavy: one queued position per lane
    for (uint32_t q0 = 0; q0 < nq; q0 += WAVE) {
      bool bad = false;
      if (q0 + lane < nq) {
        const uint32_t pos = s_q[q0 + lane];
        const uint64_t g = block_start + pos;
        uint64_t dstart = lo_start, dend = lo_end;
        if (g >= lo_end) {  // a later document of this block
          if (n_off <= SYM_DOFF) {
            uint32_t lo = 0, hi = n_off - 1u;  // largest i with s_doff[i] <= g  (s_doff[0] <= g < s_doff[n_off - 1])
            while (hi - lo > 1u) {
              const uint32_t mid = lo + ((hi - lo) >> 1);
              if (s_doff[mid] <= g) lo = mid; else hi = mid;
            }
            dstart = s_doff[lo]; dend = s_doff[lo + 1u];
          } else {
            const uint32_t d = doc_of(doc_off, d_lo, d_hi + 1, g);
            dstart = doc_off[d]; dend = doc_off[d + 1];
          }
        }
        const uint64_t l64 = dend - g, b64 = g - dstart;
        const uint32_t avail = l64 > 8 ? 8u : (uint32_t)l64, back = b64 > 3 ? 3u : (uint32_t)b64;
        // bytes g-3 .. g+3 (inside the document), from the staged block
        uint32_t bb[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const int o = k - 3;
          const bool in = o < 0 ? (uint32_t)(-o) <= back : (uint32_t)o < avail;
          bb[k] = in ? (uint32_t)sb[(int)pos + o] : 0u;
        }
        const uint32_t b0 = bb[3], b1 = bb[4], b2 = bb[5], b3 = bb[6];
        // One width decides everything.  A non-continuation byte starts a rune of go_width(b0 ..) bytes.  A
        // continuation byte starts one (U+FFFD, one byte) unless the nearest non-continuation byte within the previous
        // 3 of the same document begins a valid sequence that reaches it: the width of THAT sequence is the question.
        const uint32_t n1 = (uint32_t)((bb[2] & 0xC0u) != 0x80u), n2 = (uint32_t)((bb[1] & 0xC0u) != 0x80u),
                       n3 = (uint32_t)((bb[0] & 0xC0u) != 0x80u);
        const uint32_t in1 = (uint32_t)(back >= 1u), in2 = (uint32_t)(back >= 2u), in3 = (uint32_t)(back >= 3u);
        const uint32_t cont = (uint32_t)((b0 & 0xC0u) == 0x80u);
        const uint32_t k1 = cont & in1 & n1, k2 = cont & in2 & (n1 ^ 1u) & n2, k3 = cont & in3 & (n1 ^ 1u) & (n2 ^ 1u) & n3;
        const uint32_t k = k1 + 2u * k2 + 3u * k3;  // distance to that byte; 0: none (or b0 is no continuation byte)
        const uint32_t c0 = k3 ? bb[0] : (k2 ? bb[1] : (k1 ? bb[2] : b0)), c1 = k3 ? bb[1] : (k2 ? bb[2] : (k1 ? b0 : b1));
        const uint32_t c2 = k3 ? bb[2] : (k2 ? b0 : (k1 ? b1 : b2)), c3 = k3 ? b0 : (k2 ? b1 : (k1 ? b2 : b3));
        const uint32_t wseq = go_width(c0, c1, c2, c3, avail + k);
        const uint32_t start = k ? (uint32_t)(wseq <= k) : 1u;
        const uint32_t wd = cont ? 1u : wseq;  // a continuation byte that starts a rune is invalid on its own
        // rune value for the decoded width (U+FFFD for an invalid byte; b0 >= 0x80 here)
        const uint32_t r2 = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
        const uint32_t r3 = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
        const uint32_t r4 = ((b0 & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
        const uint32_t rune = wd == 1 ? 0xFFFDu : (wd == 2 ? r2 : (wd == 3 ? r3 : r4));
        uint32_t a_cls;  // (SYM8: the code of the rune in the width it has here)
        if (rune < 256u) {
          a_cls = lat[rune];  // (two bytes wide: 128..255)
        } else {  // matrix.go:427-435: a, ok = sigma[char]; !ok -> identity
          int l = 0, h = (int)sig.n_runes - 1;
          a_cls = (sig.identity & DTK_SYM_MASK) | (3u << DTK_SYM_CLS_SHIFT);
          if (SYM8)  // (selects: a per-lane index into the kernel argument would go through scratch memory)
            a_cls = wd == 1u ? sig.code_ident[1] : (wd == 2u ? sig.code_ident[2] : (wd == 3u ? sig.code_ident[3] : sig.code_ident[4]));
          while (l <= h) {
            const int m = (l + h) >> 1;
            const uint32_t r = sig_lds ? s_runes[m] : sig.runes[m];
            if (r == rune) {
              if (SYM8)  // (U+FFFD itself in the sigma: three bytes wide as a rune, one as an invalid byte)
                a_cls = wd == 1u ? (uint32_t)sig.code_fffd1 : (sig_lds ? (uint32_t)s_syms[m] : (uint32_t)sig.code_runes[m]);
              else
                a_cls = ((sig_lds ? (uint32_t)s_syms[m] : (uint32_t)sig.syms[m]) & DTK_SYM_MASK) |
                        (2u << DTK_SYM_CLS_SHIFT);
              break;
            }
            if (r < rune) l = m + 1; else h = m - 1;
          }
        }
        if (SYM8) sym8[g] = (uint8_t)(start ? a_cls : DTK_SYM_CONT);
        else sym[g] = (uint16_t)(a_cls | (start ? wd << DTK_SYM_W_SHIFT : 0u));
        if (start) atomicOr(&s_rs[pos >> 5], 1u << (pos & 31u));
        // a byte that decodes to U+FFFD with width 1 prints as three bytes (the renderer's slow path)
        bad = start && wd == 1u;
      }
      // The host only asks whether the run saw such a byte (the renderer's slow path): the word holds the number of
      // the last run that did, and a wave looks before it writes.  (It used to be a count: documents cut through
      // their runes -- 64-byte pieces of running text -- made 30 000 adds to this one address queue up, 140 us of
      // a 32 MiB batch.)
      if (__ballot(bad) != 0ull && lane == 0 &&
          __hip_atomic_load(n_invalid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
        atomicMax(n_invalid, epoch);
    }
  }
  // the block's rune-start bitmap (bit g of the array = input byte g): the compaction counts
  // runes with it instead of reading the symbol stream again
  __syncthreads();
  for (uint32_t i = tid; i < SYM_BLOCK_BYTES / 32; i += SYM_THREADS)
    if (i * 32u < n_here) rs_bits[(block_start >> 5) + i] = s_rs[i];
}

// --------------------------------------------------------------------- walk
//
// Transition policies.  A "state" is (t, aux): aux is unused for the matrix and
// holds the device base word of t for the double array.

template <typename CELL>
struct MatrixTrans {
  const CELL *tab;
  uint32_t stride, n_eps, start;
  static constexpr CELL FLAG = (CELL)((CELL)1 << (sizeof(CELL) * 8 - 1));
  static constexpr bool FUSED = false;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return start; }
  __device__ __forceinline__ uint32_t start_aux() const { return 0; }
  // matrix.go:442 `array[(epsilon-1)*stateCount+t0] != 0` after renumbering
  __device__ __forceinline__ bool has_eps(uint32_t t, uint32_t) const { return t <= n_eps; }
  // matrix.go:459-464; column 0 is zero so a == 0 fails without a branch
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t, uint32_t a, uint32_t &t,
                                       uint32_t &aux, bool &nontoken, uint32_t &st, uint32_t &via) const {
    via = 0;
    const CELL x = tab[(size_t)t0 * stride + a];
    t = (uint32_t)(x & (CELL)~FLAG);  // matrix.go:629 t &= ^FIRSTBIT
    nontoken = (x & FLAG) != 0;       // matrix.go:584
    aux = 0;
    (void)st;
    return t != 0;  // matrix.go:472
  }
};

// Matrix with fused cells (uint32): where state t has an epsilon arc to e and no arc on
// symbol a, but e has one, the cell (t, a) holds  1<<31 | e<<16 | cell(e, a).  The walk then
// does in one lookup what the reference does in three (matrix.go:472-497 fail + backtrack to
// the state remembered at this very rune, :563-576 epsilon step, :579-591 the rune from e).
// Only built when state ids fit 15 bits.
struct MatrixFusedTrans {
  const uint32_t *tab;
  uint32_t stride, n_eps, start;
  uint32_t ident_guard;  // the identity symbol if the model has arcs on `unknown`, else no symbol
  static constexpr bool FUSED = true;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return start; }
  __device__ __forceinline__ uint32_t start_aux() const { return 0; }
  __device__ __forceinline__ bool has_eps(uint32_t t, uint32_t) const { return t <= n_eps; }
  // via: 0, or the epsilon target e the fused cell goes through
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t, uint32_t a, uint32_t &t, uint32_t &aux,
                                       bool &nontoken, uint32_t &st, uint32_t &via) const {
    const uint32_t x = tab[(size_t)t0 * stride + a];
    t = x & 0x7FFFu;
    nontoken = (x & 0x8000u) != 0;
    via = x >> 31 ? (x >> 16) & 0x7FFFu : 0u;
    aux = 0;
    (void)st;
    return t != 0;
  }
};

// The same table walked by the lean loop (walk_fused): chosen by the launcher when no state has an
// arc on `unknown` (a separate type so that the kernels only carry one loop: fewer registers).
struct MatrixLeanTrans : MatrixFusedTrans {
  static constexpr bool LEAN = true;
};

struct DaTrans {
  const uint2 *arr;  // .x base (bit31 separate, bit30 has-epsilon cache), .y check
  uint32_t len, size, base1;
  static constexpr bool FUSED = false;
  static constexpr bool LEAN = false;
  __device__ __forceinline__ uint32_t start_state() const { return 1u; }  // datok.go:784
  __device__ __forceinline__ uint32_t start_aux() const { return base1; }
  // datok.go:876, precomputed per index at load
  __device__ __forceinline__ bool has_eps(uint32_t, uint32_t aux) const {
    return (aux & DTK_SECONDBIT) != 0;
  }
  // datok.go:889-901 and :1056-1058
  __device__ __forceinline__ bool step(uint32_t t0, uint32_t aux0, uint32_t a, uint32_t &t,
                                       uint32_t &aux, bool &nontoken, uint32_t &st, uint32_t &via) const {
    via = 0;
    const uint32_t idx = (aux0 & DTK_RESTBIT) + a;
    if (idx >= len) { st |= ST_BAD_MODEL; return false; }  // Go: index panic
    const uint2 ta = arr[idx];
    if (idx > size || (ta.y & DTK_RESTBIT) != t0) return false;
    nontoken = (ta.y & DTK_FIRSTBIT) != 0;  // datok.go:994 isNonToken
    if (ta.x & DTK_FIRSTBIT) {              // isSeparate: move to the representative
      t = ta.x & DTK_RESTBIT;
      if (t >= len) { st |= ST_BAD_MODEL; return false; }
      aux = arr[t].x;
    } else {
      t = idx;
      aux = ta.x;
    }
    return true;
  }
};

// What the walk reports: one bit per event in the bitmap of its kind (dtk_internal.h).  Token ends, token starts
// and epsilon SentenceEnds -- three bits per token and a bit -- are OR-ed into the wave's bitmaps in LDS, which the
// wave writes out as whole words when its lanes are done; the rare kinds (EOT calls) and positions outside the
// wave's range (a lane's last token may end far behind its chunk) go straight to memory.  A lane only reports inside
// its window
//   opening kinds (START, SEPS): lo <= pos < hi      closing kinds (END, TEOT, SEOT): lo < pos <= hi
// (whole document: lo = 0, hi = 0xFFFFFFFF); an event outside is dropped and remembered, the check pass then
// knows the lane left its window.
// Calls the bitmaps cannot order flag the document ST_IRREGULAR for the exact pass: a second epsilon SentenceEnd
// at one cursor, a Token call with an empty surface or a negative one (two starts or two ends on one bit), an
// EOT fired twice at one position (double array, datok.go:1019-1030 keeps its window).
// (an LDS pointer that stays one: through a plain pointer the compiler loses the address space and emits FLAT atomics)
typedef __attribute__((address_space(3))) uint32_t dtk_lds_u32;

struct EventSink {
  uint32_t *g;         // the batch's bitmaps
  uint32_t gw;         // words per kind
  uint32_t gb;         // bit of position 0 of the document
  uint32_t gbr;        // the same, counted from LDS word 0
  dtk_lds_u32 *lds;    // the wave's bitmaps in LDS (END, START, SEPS); lw == 0: none
  uint32_t lw;         // words per kind there
  uint32_t w0;         // global word of LDS word 0
  uint32_t *tailw;     // the document's tail word
  uint32_t lo, hi;     // window
  uint32_t last_s_p1;  // position of the last epsilon SentenceEnd, plus one (0: none)
  uint32_t last_eot_p; // position of the last EOT pair
  uint32_t st;
  uint32_t dropped;
  // what NewTokenWriter would have collected from this lane's calls
  // (token_writer.go:72-81, 104-109, 131-159): tokens, ints of the sentence list, texts
  uint32_t c_tok, c_sent, c_text;
  uint32_t c_sev;         // SentenceEnd calls (all of them, also where the reference would panic)
  uint32_t e_pos, e_tok;  // the last EOT TextEnd of this lane: position, Token calls before it
  __device__ __forceinline__ void init(const DtkWalkArgs &A, uint64_t off, uint32_t d, uint32_t wlo, uint32_t whi,
                                       uint32_t *lds_bits = nullptr, uint32_t lds_words = 0, uint32_t word0 = 0) {
    g = A.bits; gw = A.bit_words; gb = (uint32_t)DTK_EV_BIT(off, d); tailw = A.doc_tail ? A.doc_tail + d : nullptr;
    lds = (dtk_lds_u32 *)lds_bits; lw = lds_bits ? lds_words : 0u; w0 = word0; gbr = gb - (word0 << 5);
    lo = wlo; hi = whi;
    last_s_p1 = 0u; last_eot_p = 0xFFFFFFFFu; st = 0; dropped = 0;
    c_tok = c_sent = c_text = 0;
    c_sev = 0; e_pos = 0xFFFFFFFFu; e_tok = 0;
  }
  __device__ __forceinline__ void put(uint32_t kind, uint32_t pos) {
    if (DTK_KO & 8) return;
    const uint32_t G = gb + pos, m = 1u << (G & 31u), w = (G >> 5) - w0;
    // (one wave-uniform test keeps the common case free of exec-mask juggling: all lanes inside the LDS range)
    if (__builtin_amdgcn_ballot_w64(w >= lw) == 0ull) {
      __hip_atomic_fetch_or(&lds[kind * lw + w], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_or_b32
    } else if (w < lw) {
      __hip_atomic_fetch_or(&lds[kind * lw + w], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      atomicOr(&g[kind * gw + (G >> 5)], m);
    }
  }
  __device__ __forceinline__ bool in_closing(uint32_t p) const { return p > lo && p <= hi; }
  __device__ __forceinline__ bool in_opening(uint32_t p) const { return p >= lo && p < hi; }

  // Token(bufft, buffer[:buffc]) -- matrix.go:528,569,675
  // sent_first: no token since the last SentenceEnd / TextEnd (the writer's sentB)
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token(uint32_t /*bs*/, uint32_t tp, uint32_t p, bool sent_first) {
    if (!in_closing(p)) { dropped = 1; return; }
    c_tok++;
    c_sent += sent_first ? 1u : 0u;
    if (p <= tp) st |= ST_IRREGULAR;
    // The double array keeps its window over an EOT: a token may be flushed AFTER the EOT's SentenceEnd / TextEnd and
    // end BEFORE it (the walk backtracked behind the EOT; if the retry then reads the EOT as an ordinary rune,
    // matrix.go:555 / datok.go, nothing fires twice).  Calls out of position order: the exact pass.
    if (!IS_MATRIX && last_eot_p != 0xFFFFFFFFu && p < last_eot_p) st |= ST_IRREGULAR;
    put(EVB_END, p);
    put(EVB_START, tp);
  }
  // SentenceEnd? + TextEnd fired by an EOT rune -- matrix.go:593-600
  // has_tok: the current text has a token (else the reference panics in position modes)
  template <bool IS_MATRIX>
  __device__ __forceinline__ void eot(uint32_t /*bs*/, uint32_t p, bool with_sentence, bool has_tok) {
    // (The double array has no upper bound here: it keeps its window over an EOT, so the EOT is no sync point and a
    //  lane may fire one behind its stop position and then backtrack to a token end in front of it -- the construct
    //  of the exact pass.  The fire is the lane's: counted, its bit set; the successor's second fire finds the bit
    //  and flags the document.  Dropped, it made the lane fail its check in every repair round.)
    if (IS_MATRIX ? !in_closing(p) : p <= lo) { dropped = 1; return; }
    c_text++;
    c_sev += with_sentence ? 1u : 0u;
    e_pos = p; e_tok = c_tok;
    if (has_tok) c_sent += with_sentence ? 1u : 0u; else st |= ST_EMPTY_TEXT;
    last_eot_p = p;
    const uint32_t G = gb + p, m = 1u << (G & 31u);
    if (atomicOr(&g[EVB_TEOT * gw + (G >> 5)], m) & m) st |= ST_IRREGULAR;  // the same EOT fired before (by any lane)
    if (with_sentence) atomicOr(&g[EVB_SEOT * gw + (G >> 5)], m);
  }
  // SentenceEnd from an epsilon arc on an empty token -- matrix.go:574-575
  template <bool IS_MATRIX>
  __device__ __forceinline__ void sentence(uint32_t /*bs*/, uint32_t p, bool has_tok) {
    if (!in_opening(p)) { dropped = 1; return; }
    c_sev++;
    if (has_tok) c_sent++; else st |= ST_EMPTY_TEXT;
    if (p < last_s_p1) st |= ST_IRREGULAR;  // twice at one position, or behind a backtrack: not in position order
    last_s_p1 = p + 1u;
    put(EVB_SEPS, p);
  }
  // First pass (k_spec_both), events before the lane's stop position: the window is open-ended and the position lies
  // inside the wave's LDS bitmaps (their 64 chunks plus a bit per document boundary; DTK_LDS_BIT_WORDS) -- no window
  // test, no range test.  What a lane reports at or behind its stop position goes through the calls above.
  __device__ __forceinline__ void put_first(uint32_t kind, uint32_t pos) {
    if (DTK_KO & 8) return;
    // (k_spec_both only runs with LDS bitmaps: without them dtk_batch_run launches start records and walk apart;
    //  gbr = the document's bit base relative to the wave's first LDS word)
    const uint32_t G = gbr + pos;
    __hip_atomic_fetch_or(&lds[kind * lw + (G >> 5)], 1u << (G & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token_first(uint32_t tp, uint32_t p, bool sent_first) {
    c_tok++;
    c_sent += sent_first ? 1u : 0u;
    if (!IS_MATRIX && last_eot_p != 0xFFFFFFFFu && p < last_eot_p) st |= ST_IRREGULAR;  // (see token())
    put_first(EVB_END, p);
    put_first(EVB_START, tp);
  }
  __device__ __forceinline__ void sentence_first(uint32_t p, bool has_tok) {
    c_sev++;
    if (has_tok) c_sent++; else st |= ST_EMPTY_TEXT;
    if (p < last_s_p1) st |= ST_IRREGULAR;  // twice at one position, or behind a backtrack: not in position order
    last_s_p1 = p + 1u;
    put_first(EVB_SEPS, p);
  }
  // final SentenceEnd / TextEnd -- matrix.go:683-691
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail_(uint32_t p, bool sentence_end, bool text_end, bool has_tok) {
    const uint32_t bits = (sentence_end ? 0u : DTK_TAIL_S) | (text_end ? 0u : DTK_TAIL_E);
    if (!bits) return;
    if (!in_opening(p)) { dropped = 1; return; }
    if (!text_end) c_text++;
    c_sev += sentence_end ? 0u : 1u;
    if (has_tok) c_sent += sentence_end ? 0u : 1u; else st |= ST_EMPTY_TEXT;
    if (tailw && !(DTK_KO & 8)) *tailw = (p << 2) | bits;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail(uint32_t /*bs*/, uint32_t p, bool sentence_end, bool text_end, bool has_tok) {
    tail_<IS_MATRIX>(p, sentence_end, text_end, has_tok);
  }
  // calls that the position-indexed bitmaps cannot order
  __device__ __forceinline__ void out_of_order() { st |= ST_IRREGULAR; }
  __device__ __forceinline__ void flush() {}
};

// the wave's LDS bitmaps: cleared before the lanes walk, OR-ed into memory afterwards (all 64 lanes take part;
// one wave per block, LDS operations of a wave complete in order)
__device__ __forceinline__ void lds_bits_clear(uint32_t *lds_, uint32_t lw) {
  dtk_lds_u32 *lds = (dtk_lds_u32 *)lds_;
  for (uint32_t j = threadIdx.x; j < 3u * lw; j += WAVE) lds[j] = 0u;
  __syncthreads();
}
__device__ __forceinline__ void lds_bits_flush(const uint32_t *lds_, uint32_t lw, uint32_t *g, uint32_t gw, uint32_t w0) {
  const dtk_lds_u32 *lds = (const dtk_lds_u32 *)lds_;
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < lw; j += WAVE) {
    if (w0 + j >= gw) break;
#pragma unroll
    for (uint32_t k = 0; k < 3u; k++) {
      const uint32_t v = lds[k * lw + j];
      if (v) atomicOr(&g[k * gw + w0 + j], v);
    }
  }
}
// global word of LDS word 0 for the wave whose first lane is L0
__device__ __forceinline__ uint32_t lds_bits_word0(const DtkWalkArgs &A, const DtkSpecArgs &S, uint32_t L0) {
  const uint32_t d0 = S.lane_doc[L0];
  return (uint32_t)((DTK_EV_BIT(A.doc_off[d0], d0) + (uint64_t)(L0 - S.chunk_off[d0]) * S.chunk) >> 5);
}

template <typename TRANS>
__device__ __forceinline__ uint32_t guard_of(const TRANS &tr) {
  if constexpr (TRANS::FUSED) return tr.ident_guard; else return 0xFFFFFFFFu;
}

// What a lane is asked to do.
enum { MODE_DOC = 0,    // whole document from the initial state, all events
       MODE_START = 1,  // speculative warm-up: no events, stop at the first rewind at/after stop_pos
       MODE_CHUNK = 2   // walk from a recorded start, events inside the window, stop at the first
                        // rewind at/after stop_pos (or run the EOF tail)
};

// Runes in [from, to): only needed when a window may have outgrown the
// reference's 1024-rune buffer (matrix.go:365), i.e. when it spans > 1024 bytes.
struct DtkSymAt { DtkSym S; uint64_t off; };  // a document's stretch of the stream
__device__ __noinline__ uint32_t count_runes(const DtkSymAt &s, uint32_t from, uint32_t to) {
  uint32_t n = 0;
  for (uint32_t i = from; i < to; i++) n += dtk_sym_is_start(s.S, s.off + i) ? 1u : 0u;
  return n;
}

// window of the symbol stream in LDS (see walk_fused)
typedef uint2 __attribute__((may_alias)) dtk_u2a;
typedef uint16_t __attribute__((may_alias)) dtk_u16a;
typedef uint8_t __attribute__((may_alias)) dtk_u8a;
// (the general loop keeps entries in its rows: from a stream of codes they are translated on the way in -- 32 table
//  reads per refill, cached; the loop itself does not know the difference.  `at` = stream index of the row's entry 0,
//  a multiple of 8)
__device__ __forceinline__ void win_fill(dtk_u16a *row, const DtkSym &S, uint64_t at) {
  if (S.lut) {
    const uint2 *__restrict__ g = reinterpret_cast<const uint2 *>(static_cast<const uint8_t *>(S.base) + at);
    uint2 c[DTK_WIN / 8u];
#pragma unroll
    for (uint32_t i = 0; i < DTK_WIN / 8u; i++) c[i] = g[i];
    dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
    for (uint32_t i = 0; i < DTK_WIN / 8u; i++) {
      const uint32_t lo = c[i].x, hi = c[i].y;
      r[2 * i] = make_uint2((uint32_t)S.lut[lo & 255u] | ((uint32_t)S.lut[(lo >> 8) & 255u] << 16),
                            (uint32_t)S.lut[(lo >> 16) & 255u] | ((uint32_t)S.lut[lo >> 24] << 16));
      r[2 * i + 1] = make_uint2((uint32_t)S.lut[hi & 255u] | ((uint32_t)S.lut[(hi >> 8) & 255u] << 16),
                                (uint32_t)S.lut[(hi >> 16) & 255u] | ((uint32_t)S.lut[hi >> 24] << 16));
    }
    return;
  }
  const uint4 *__restrict__ g = reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(S.base) + at);
  uint4 v[DTK_WIN / 8u];
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN / 8u; i++) v[i] = g[i];  // all loads before the first LDS write
  dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN / 8u; i++) {
    r[2 * i] = make_uint2(v[i].x, v[i].y);
    r[2 * i + 1] = make_uint2(v[i].z, v[i].w);
  }
}

// the lean loop's row: DTK_WIN8 codes
__device__ __forceinline__ void win_fill8(dtk_u8a *row, const uint8_t *__restrict__ from) {
  const uint4 *__restrict__ g = reinterpret_cast<const uint4 *>(from);
  uint4 v[DTK_WIN8 / 16u];
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN8 / 16u; i++) v[i] = g[i];  // all loads before the first LDS write
  dtk_u2a *r = reinterpret_cast<dtk_u2a *>(row);
#pragma unroll
  for (uint32_t i = 0; i < DTK_WIN8 / 16u; i++) {
    r[2 * i] = make_uint2(v[i].x, v[i].y);
    r[2 * i + 1] = make_uint2(v[i].z, v[i].w);
  }
}

// The walk of matrix.go:348-698 / datok.go:781-1135 for one lane.
// Returns through `fin`: p == 0xFFFFFFFF means "ran to EOF" (MODE_START: no
// rewind found; otherwise: tail done).
//
// The reference's rune window is not materialised: p / tp / bs / hi are byte
// positions of buffer[buffc] / buffer[bufft] / buffer[0] / buffer[buffi], and the
// symbol stream (read through the lane's window in LDS) replaces the rune -> symbol lookups
// of matrix.go:421-435.
template <typename TRANS, bool IS_MATRIX, int MODE, typename SINK = EventSink>
__device__ __forceinline__ void walk_lane(const TRANS &tr, const DtkSym &sym,
                                          uint64_t off, uint32_t len, DtkLaneState init, uint32_t stop_pos,
                                          SINK &sink, uint32_t epsilon, uint32_t unknown,
                                          uint32_t identity, uint32_t cap, DtkLaneState &fin,
                                          uint32_t &st_out, uint32_t &steps_out, uint16_t *win_row) {
  // the lane's window of the symbol stream in LDS: entries (pos + o7) in [wbase, wbase + DTK_WIN)
  dtk_u16a *row = reinterpret_cast<dtk_u16a *>(win_row);
  const uint32_t o7 = (uint32_t)(off & 7u);
  const uint64_t aligned = off - o7;
  const DtkSymAt s{sym, off};

  uint32_t a = 0, t0 = 0, aux0 = 0;
  uint32_t t = init.t, aux = init.aux;  // matrix.go:351 `t := uint32(1)`
  const uint32_t t_start = tr.start_state(), aux_start = tr.start_aux();
  bool ok = (init.flags & LANE_F_OK) != 0;  // sticky `ok` of matrix.go:352 / datok.go:785
  uint32_t eps_t = 0, eps_aux = 0, eps_p = 0;  // epsilonState / epsilonOffset
  bool sentence_end = (init.flags & LANE_F_SENT) != 0, text_end = (init.flags & LANE_F_TEXT) != 0;
  uint32_t p = init.p;   // buffer[buffc]
  uint32_t tp = init.p;  // buffer[bufft]
  uint32_t bs = init.p;  // buffer[0]: position of the last rewind
  uint32_t hi = init.p;  // behind buffer[buffi-1]: read high-water mark
  uint32_t w = 1;        // width of the rune at p
  bool eot = false, newchar = true;
  uint32_t st = 0, my_steps = 0;
  fin.p = 0xFFFFFFFFu; fin.t = 0; fin.aux = 0; fin.flags = 0;
  bool stopped = false;
  // NewTokenWriter state that the counts need.  A lane starts right after a rewind:
  // p > 0 means a token was flushed or an EOT fired there, so "a token exists in the
  // document" is p > 0 and "in the current text" additionally needs !textEnd.
  bool any_tok = init.p > 0;               // some Token call happened (else sentB is still true)
  bool has_tok = init.p > 0 && !text_end;  // pos[] of the current text is not empty

  uint32_t wbase = (init.p + o7) & ~7u;
  win_fill(row, sym, aligned + wbase);

  // One table lookup per iteration (the reference's loop body, matrix.go:384-635),
  // written as predicates + selects so that the 64 lanes of a wave, which are all
  // in different phases of their tokens, share one short instruction stream.  Real
  // branches are kept for the symbol fetch, the two event stores and three rare
  // paths (EOF drain; hard fail; EOT / window limit / end of chunk).
  bool done = false;
  do {
    if (newchar && p >= len) {
      // reader at EOF: the drain of matrix.go:650-668 / datok.go:1085-1103
      const bool he = tr.has_eps(t, aux);          // goto PARSECHARM with a = epsilon
      const bool bt = !he && eps_t != 0;           // or pop the remembered epsilon state
      t0 = bt ? eps_t : t; aux0 = bt ? eps_aux : aux;
      p = bt ? eps_p : p;
      eps_t = bt ? 0u : eps_t;
      a = epsilon;
      newchar = false;
      done = !he && !bt;
    }
    // a lane that needs a rune outside its window: all lanes of the wave re-base theirs
    if (__builtin_amdgcn_ballot_w64(newchar && (p + o7 - wbase) >= DTK_WIN) != 0ull) {
      wbase = (p + o7) & ~7u;
      win_fill(row, sym, aligned + wbase);
    }
    if (newchar) {
      const uint32_t e = row[p + o7 - wbase];
      a = e & DTK_SYM_MASK;
      w = DTK_SYM_WIDTH(e);
      const uint32_t cls = (e >> DTK_SYM_CLS_SHIFT) & 3u;
      hi = max(hi, p + w);             // matrix.go:388-408: runes enter the window once
      eot = cls == 1u;                 // matrix.go:422
      ok = cls >= 2u ? cls == 2u : ok; // matrix.go:427: only runes >= 256 write `ok`
      t0 = t; aux0 = aux;              // matrix.go:437
      const bool he = tr.has_eps(t0, aux0);  // matrix.go:442-454
      eps_t = he ? t0 : eps_t; eps_aux = he ? aux0 : eps_aux;
      eps_p = he ? p : eps_p;
    }

    bool nontoken = false;
    uint32_t via = 0;
    const bool fresh = newchar;  // this lookup is the first one for the rune at p
    bool good = tr.step(t0, aux0, a, t, aux, nontoken, st, via);  // a finished lane looks up harmlessly
    const bool act = !done;
    my_steps += act ? 1u : 0u;
    const bool is_eps = a == epsilon;
    // A fused cell stands for: this rune has no arc here, the epsilon state remembered at this
    // very rune is t0 itself (matrix.go:442-454), take its epsilon arc to `via`, then the rune
    // from there.  Not taken where the reference would first retry with the unknown symbol
    // (matrix.go:478-485; only observable if the model has such arcs), nor where the rewind
    // would end this lane's chunk (the plain path then stops at the rewind).
    bool comp = false;
    if (TRANS::FUSED) {
      comp = act && good && via != 0 && fresh && !(!ok && a == guard_of(tr)) &&
             !(MODE != MODE_DOC && p >= stop_pos);
      good = good && (via == 0 || comp);
    }
    const bool succ = act && good && !comp, fail = act && !good;
    const bool retry_unknown = fail && !ok && a == identity;               // matrix.go:478-485
    const bool backtrack = fail && !retry_unknown && !is_eps && eps_t != 0;   // matrix.go:487-497
    bool hardfail = fail && !retry_unknown && !backtrack;                  // matrix.go:499-552
    const bool flush_eps = succ && is_eps && p > tp;                       // matrix.go:565-572
    const bool sent_eps = succ && is_eps && p <= tp;                       // matrix.go:573-576
    const bool advance = succ && !is_eps;                                  // matrix.go:579-591
    // matrix.go:593-605: after ANY successful step while `eot` is set.  It is set by the rune just read, cleared by a
    // retry (:555) and by the next rune -- so an epsilon step sees it only in the EOF drain behind a hard fail on a
    // trailing EOT (the hard-fail branch leaves it set, :499-552, and no rune follows to clear it).
    const bool eot_now = (succ || comp) && eot;

    if (hardfail || my_steps > cap) {  // rare
      if (hardfail) {  // drop what is buffered as a token, restart at state 1
        if (is_eps) { st |= ST_BAD_MODEL; done = true; hardfail = false; }  // stale-buffer case
        else if (p <= tp) { p += w; }                                        // matrix.go:515-516
        if (hardfail && p < tp) st |= ST_BAD_OFFSET;  // Token(bufft, buffer[:buffc]) with bufft > buffc
        t = t_start; aux = aux_start;                                        // matrix.go:548
      }
      if (my_steps > cap) { st |= ST_STEP_LIMIT; done = true; hardfail = false; }
    }
    const bool flush_c = comp && p > tp, sent_c = comp && p <= tp;  // the epsilon half of a fused cell
    const bool flush = flush_eps || hardfail || flush_c;
    if (MODE != MODE_START) {
      if (flush)  // matrix.go:528 / 569
        sink.template token<IS_MATRIX>(bs, tp, p, sentence_end || text_end || !any_tok);
      if (sent_eps || sent_c) sink.template sentence<IS_MATRIX>(bs, p, has_tok);  // matrix.go:575
    }
    any_tok = any_tok || flush;
    has_tok = has_tok || flush;
    // consume the rune (for a fused cell: from the epsilon target, right after its rewind, so
    // the rune is the first of the window)
    // (a fused cell's rune is the first of its token if the epsilon half flushed, or if the token was empty; after a
    //  backtrack to a slot BEHIND the token start -- bufft > buffc, the reference's own odd case -- it is neither)
    const bool skip = (advance && p == tp && nontoken) || (comp && nontoken && p >= tp);  // matrix.go:584-588
    const uint32_t p_old = p;
    p = (advance || comp) ? p + w : p;
    tp = skip ? p : (flush_c ? p_old : tp);  // (the epsilon half of a fused cell rewinds only if it flushed)
    // the EOT fires a SentenceEnd unless one is pending (after the epsilon half of a fused cell)
    const bool eot_sent = !((flush_c || flush_eps) ? false : ((sent_c || sent_eps) ? true : sentence_end));
    sentence_end = eot_now ? true : (flush ? false : ((sent_eps || sent_c) ? true : sentence_end));
    text_end = eot_now ? true : (flush ? false : text_end);
    // retries keep the rune, everything else fetches a new one
    t0 = backtrack ? eps_t : t0; aux0 = backtrack ? eps_aux : aux0;
    p = backtrack ? eps_p : p;
    a = backtrack ? epsilon : (retry_unknown ? unknown : a);
    eot = (retry_unknown || backtrack) ? false : eot;  // matrix.go:555: a retry forgets that the rune was EOT
    newchar = succ || hardfail || comp;
    eot = eot_now ? false : eot;  // matrix.go:594
    const bool rewind = flush || (IS_MATRIX && eot_now);  // matrix.go:601 vs datok.go:1019-1030
    eps_t = (backtrack || rewind || comp) ? 0u : eps_t;
    if (TRANS::FUSED) {
      // the epsilon target is the state the rune was read in: remembered if it has an epsilon arc
      const bool he2 = comp && !(IS_MATRIX && eot_now) && tr.has_eps(via, 0u);
      eps_t = he2 ? via : eps_t; eps_p = he2 ? p_old : eps_p;
    }
    // rare: EOT calls, the reference's 1024-rune window limit (checked where the window was
    // longest), end of this lane's chunk
    // a fused cell rewinds before its rune: that rewind is at p_old, never the end of the chunk
    const bool rewind_end = (flush && !comp) || (IS_MATRIX && eot_now);
    const bool long_win = hi - bs > DTK_WINDOW_BYTES;  // overflowed for certain: the lane stops (see walk_fused)
    // (a hard fail on the document's last rune, an EOT: `eot` stays set for the EOF drain, and a start record has no
    //  place for it -- the lane that read the rune runs the drain itself)
    const bool at_stop = rewind_end && MODE != MODE_DOC && p >= stop_pos && !(hardfail && eot && p >= len);
    if (eot_now || (rewind && hi - bs > DTK_WINDOW) || at_stop || long_win) {
      if (eot_now) {
        // (fired by an epsilon step -- the stale `eot` -- the TextEnd follows a Token that ends at the same position:
        //  rows in call order, the exact pass)
        if (is_eps && MODE != MODE_START) sink.out_of_order();
        // (the epsilon half of a fused cell has rewound the window to p_old before its rune was read)
        if (MODE != MODE_START) sink.template eot<IS_MATRIX>(flush_c ? p_old : bs, p, eot_sent, has_tok);
        has_tok = false;  // TextEnd: pos = pos[:0] (token_writer.go:158)
      }
      if (rewind) {
        if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
        if (at_stop) {
          fin.p = p; fin.t = t; fin.aux = aux;
          fin.flags = (sentence_end ? LANE_F_SENT : 0u) | (text_end ? LANE_F_TEXT : 0u) |
                      (ok ? LANE_F_OK : 0u);
          stopped = true;
          done = true;
        }
      }
      // (the tail below then closes the document at this position: every document keeps its TextEnd)
      if (long_win && !done) { st |= ST_WINDOW_OVERFLOW; done = true; }
    }
    tp = rewind_end ? p : tp;  // matrix.go:537-543 / 608-627
    bs = rewind_end ? p : (flush_c ? p_old : bs);
  } while (!done);

  if (!stopped && !(st & (ST_STEP_LIMIT | ST_BAD_MODEL))) {
    if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
    if (MODE != MODE_START) {
      if (p > tp) {  // matrix.go:671-678
        sink.template token<IS_MATRIX>(bs, tp, p, sentence_end || text_end || !any_tok);
        sentence_end = false; text_end = false;
        has_tok = true;
      }
      sink.template tail<IS_MATRIX>(bs, p, sentence_end, text_end, has_tok);  // matrix.go:683-691
    }
  }
  st_out = st;
  steps_out = my_steps;
}

// The same walk for the common case -- matrix with fused cells, no arc on the `unknown` symbol
// anywhere (so the sticky `ok` and the retry of matrix.go:478-485 have no observable effect) --
// written for a short instruction stream: one symbol prefetch and one cell load per iteration,
// lane state in plain integers, and one guarded block for everything that happens less than
// once per token (hard fail, EOT, end of input, end of the chunk, the window limit).
// Behaviour is identical to walk_lane<MatrixFusedTrans, true, MODE> for such models.
//
// Symbol stream: every lane keeps a window of DTK_WIN entries of its own stretch of the stream in
// LDS (its private row; rows are 72 B apart so that the 64 lanes start in different banks).  The
// window is filled with four 16-byte loads per lane and read with one ds_read_u16 per iteration;
// when any lane of the wave leaves its window all lanes re-base theirs (a wave-uniform branch,
// once per ~28 iterations).  Read straight from memory in 8-byte groups, the lanes' 64 stream
// lines and the table lines evict each other from the 32 KiB L1 and every group load goes to L2.
#ifdef DTK_PROBE
// (scripts/probe.py) cycles the waves spend waiting for the cell and the entry at the end of an iteration / in the
// loop / iterations / waves -- chunk walks [0..3], warm-up walks [4..7]
__device__ unsigned long long g_probe[8];
extern "C" int dtk_probe_read(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(g_probe));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); }
  return 0;
}
#endif

template <int MODE, bool FIRST = false, bool IS_MATRIX = true>
__device__ __forceinline__ void walk_fused(const MatrixFusedTrans &tr, const DtkSym &sym,
                                           uint64_t off, uint32_t len, DtkLaneState init, uint32_t stop_pos,
                                           EventSink &sink, uint32_t epsilon, uint32_t cap, DtkLaneState &fin,
                                           uint32_t &st_out, uint32_t &steps_out, uint16_t *win_row,
                                           const uint16_t *lut) {
  const DtkSymAt s{sym, off};
  const uint32_t *__restrict__ tab = tr.tab;
  const uint32_t stride = tr.stride, n_eps = tr.n_eps;
  uint32_t t = init.t;
  uint32_t p = init.p, tp = init.p, bs = init.p, hi = init.p;
  uint32_t eps_t = 0, eps_p = 0;
  // F: 1 sentenceEnd, 2 textEnd (matrix.go:360-363), 4 some Token call happened in this document,
  //    8 the current text has a token (what NewTokenWriter's sentB / pos need)
  uint32_t F = (init.flags & (LANE_F_SENT | LANE_F_TEXT)) | (init.p > 0 ? 4u : 0u);
  F |= (init.p > 0 && !(init.flags & LANE_F_TEXT)) ? 8u : 0u;
  static_assert(LANE_F_SENT == 1u && LANE_F_TEXT == 2u, "flag layout");
  uint32_t st = 0;
  uint32_t budget = cap;  // lookups left; the one that finds none left sets ST_STEP_LIMIT
  fin.p = 0xFFFFFFFFu; fin.t = 0; fin.aux = 0; fin.flags = 0;  // p stays "ran to EOF" unless the lane stops
  bool done = false;
  // the lane's window of the symbol stream: the CODE of position q at row[q - wb7], q - wb7 in [0, DTK_WIN8); its
  // entry is lut[code] (the model's code table, in LDS)
  dtk_u8a *row = reinterpret_cast<dtk_u8a *>(win_row);
  const uint32_t o7 = (uint32_t)(off & 15u);  // (windows start at multiples of 16 codes: 16-byte loads)
  const uint8_t *__restrict__ aligned = static_cast<const uint8_t *>(sym.base) + (off - o7);
#define DTK_ENTRY(q_) ((uint32_t)lut[row[(q_) - wb7]])
#define DTK_REFILL(q_) { wb7 = (((q_) + o7) & ~15u) - o7; win_fill8(row, aligned + (wb7 + o7)); }
  uint32_t wb7;
  DTK_REFILL(p)
  // The entry the next lookup is made with: the stream entry of the rune at p -- or, right after a backtrack,
  // the bare epsilon symbol: width 0, so that iteration consumes nothing and reads no rune (matrix.go:487-497).
  uint32_t e = DTK_ENTRY(p);

  // Reader at EOF before a rune is read (matrix.go:650-668): epsilon arcs are taken as long as the state has one
  // (here, on the spot: one lookup each); then the remembered epsilon state is popped -- the walk goes on from
  // there with an epsilon iteration -- or the walk is over.
#define DTK_EOF_DRAIN()                                                                                       \
  if (p >= len) {                                                                                             \
    bool first_ = true;                                                                                       \
    while (t <= n_eps && !done) {                                                                             \
      const uint32_t x_ = tab[__umul24(t, stride) + epsilon];                                                 \
      const bool ov_ = __builtin_usub_overflow(budget, 1u, &budget);                                          \
      if ((int32_t)x_ <= 0) { st |= ST_BAD_MODEL; done = true; break; }                                       \
      if (p > tp) { /* matrix.go:565-572 */                                                                   \
        if (MODE != MODE_START) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);                   \
        F = 12u;                                                                                              \
        if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;            \
        tp = p; bs = p; eps_t = 0;                                                                            \
        if (MODE != MODE_DOC && p >= stop_pos) {                                                              \
          fin.p = p; fin.t = x_ & 0x7FFFu; fin.aux = 0; fin.flags = init.flags & LANE_F_OK;                   \
          done = true;                                                                                        \
        }                                                                                                     \
      } else { /* matrix.go:573-576 */                                                                        \
        if (MODE != MODE_START) sink.template sentence<IS_MATRIX>(bs, p, (F & 8u) != 0);                           \
        F |= 1u;                                                                                              \
      }                                                                                                       \
      t = x_ & 0x7FFFu;                                                                                       \
      if (eot_stale && first_ && !done) { /* matrix.go:593-605 behind the first successful step, see the hard-fail block */ \
        /* (a TextEnd behind a Token that ends at the same position: rows in call order, the exact pass) */   \
        if (MODE != MODE_START) sink.out_of_order();                                                          \
        if (MODE != MODE_START) sink.template eot<IS_MATRIX>(bs, p, (F & 1u) == 0u, (F & 8u) != 0);           \
        F = (F & 4u) | 3u;                                                                                    \
        if (IS_MATRIX) {                                                                                      \
          eps_t = 0; tp = p; bs = p;                                                                          \
          if (MODE != MODE_DOC && p >= stop_pos) {                                                            \
            fin.p = p; fin.t = t; fin.aux = 0; fin.flags = (F & 3u) | (init.flags & LANE_F_OK);               \
            done = true;                                                                                      \
          }                                                                                                   \
        }                                                                                                     \
      }                                                                                                       \
      first_ = false;                                                                                         \
      if (ov_) { st |= ST_STEP_LIMIT; done = true; }                                                          \
    }                                                                                                         \
    if (!done) {                                                                                              \
      if (eps_t != 0) { t = eps_t; p = eps_p; eps_t = 0; e = epsilon; } else done = true;                     \
    }                                                                                                         \
  }
  {
    const bool eot_stale = false;
    DTK_EOF_DRAIN()
  }
  // The loop is rotated: the cell of the NEXT lookup is requested as soon as this one's cell says where the walk goes
  // (a dozen instructions behind its arrival), and everything else an iteration does -- events, token window, flags,
  // the epsilon slot -- runs while that request is under way.  In program order the whole iteration used to stand
  // between a cell's arrival and the next request, and a wave issues in order.  What the rare block decides (hard
  // fail, EOT, EOF drain) is not known yet when the request leaves: it asks again.
  //   x  : the cell (t, e)          en : the stream entry behind the rune at p (position p + width(e))
  uint32_t x = 0, en = 0;
#define DTK_TAB(t_, e_) (*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tab) +                  \
                                                            ((__umul24((t_), stride) + ((e_) & DTK_SYM_MASK)) << 2)))
  // the fused table is at most 2^15 states x 2^11 symbols x 4 B: a 32-bit byte offset from the
  // uniform base (one 24-bit multiply-add) instead of 64-bit address arithmetic
  if (!done) {
    x = DTK_TAB(t, e);
    const uint32_t pn0 = p + ((e >> DTK_SYM_W_SHIFT) & 7u);
    if (pn0 - wb7 >= DTK_WIN8) DTK_REFILL(pn0)
    en = DTK_ENTRY(pn0);
  }
#ifdef DTK_PROBE
  unsigned long long pr_wait = 0, pr_t0 = clock64(), pr_n = 0;
#endif
  while (!done) {
    // (the lookup cap as a budget counted down: the borrow of the subtraction is the test -- one instruction, not two)
    const bool over = __builtin_usub_overflow(budget, 1u, &budget);
    const uint32_t w = (e >> DTK_SYM_W_SHIFT) & 7u;     // bytes of the rune at p; 0: an epsilon iteration
    const uint32_t pn = p + w;
    // matrix.go:442-454.  (An epsilon iteration -- state and position of the slot it was popped from -- would put
    // the same slot back; it is dropped again below because every epsilon step drops it.)
    const bool he = t <= n_eps;
    eps_t = he ? t : eps_t; eps_p = he ? p : eps_p;
    const bool r = w == 0u;
    const uint32_t tgt = x & 0x7FFFu, via = (x >> 16) & 0x7FFFu;
    const bool comp = (int32_t)x < 0;                   // a fused cell: the epsilon arc of t, then the rune from there
    const bool plain = (int32_t)x > 0;
    const bool fail = x == 0u;
    const bool advance = comp || (plain && !r);         // the rune is consumed, matrix.go:579-591
    const bool backtrack = fail && !r && eps_t != 0;    // matrix.go:487-497
    // ---- where the walk goes, and the request for its cell
    const uint32_t t_n = backtrack ? eps_t : (fail ? t : tgt);
    const uint32_t p_n = backtrack ? eps_p : (advance ? pn : p);
    // right after a backtrack the bare epsilon symbol: width 0, that iteration consumes nothing (matrix.go:487-497)
    const uint32_t e_n = backtrack ? epsilon : en;
    const uint32_t x_n = DTK_TAB(t_n, e_n);
    // nontoken && (comp || (advance && p == tp)), matrix.go:584-588.  (As lane masks combined in scalar registers,
    //  and here, in the block of the comparisons: written with && / || or & / | further down the compiler builds
    //  the predicate from 0/1 integers in vector registers, seven instructions instead of two.)
    const unsigned long long m_comp = __builtin_amdgcn_ballot_w64((int32_t)x < 0),
                             m_adv = m_comp | (__builtin_amdgcn_ballot_w64((int32_t)x > 0) & ~__builtin_amdgcn_ballot_w64(w == 0u));
    // (a fused cell's rune is the first of its token unless the walk has backtracked to a slot BEHIND the token
    //  start -- bufft > buffc, the reference's own odd case: then its epsilon half neither flushes nor rewinds)
    const unsigned long long m_skip = __builtin_amdgcn_ballot_w64((x & 0x8000u) != 0u) &
                                      ((m_comp & __builtin_amdgcn_ballot_w64(p > tp)) | (m_adv & __builtin_amdgcn_ballot_w64(p == tp)));
    uint32_t code_n;  // (its entry is looked up at the end of the iteration: the code has arrived by then)
    {
      const uint32_t pn_n = p_n + ((e_n >> DTK_SYM_W_SHIFT) & 7u);
      uint32_t iw = pn_n - wb7;
      if (__builtin_amdgcn_ballot_w64(iw >= DTK_WIN8) != 0ull) {  // also a backtrack to before the window
        DTK_REFILL(pn_n)
        iw = pn_n - wb7;
      }
      code_n = row[iw];
    }
    // ---- this iteration's bookkeeping, under the request
    hi = max(hi, pn);                                   // matrix.go:388-408
    const bool epsE = comp || (plain && r);             // an epsilon arc is taken at p
    const bool flush = epsE && p > tp;                  // matrix.go:565-572
    const bool sentE = epsE && p <= tp;                 // matrix.go:573-576
    const bool hardfail = fail && !backtrack;
    // (first pass: an epsilon step at or behind the stop position is the rare block's -- the lane's last token, or a
    //  SentenceEnd on its way there; the positions before it need no test, see EventSink::put_first)
    const bool beyond = MODE != MODE_DOC && p >= stop_pos;
    const uint32_t tp_old = tp, F_old = F;
    if (MODE != MODE_START && !(DTK_KO & 1)) {
      if (FIRST) {
        if (flush && !beyond) sink.template token_first<IS_MATRIX>(tp, p, ((F ^ 4u) & 7u) != 0);
        if (sentE && !beyond) sink.sentence_first(p, (F & 8u) != 0);
      } else {
        if (flush) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
        if (sentE) sink.template sentence<IS_MATRIX>(bs, p, (F & 8u) != 0);
      }
    }
    const uint32_t win = (DTK_KO & 2) ? 0u : hi - bs;   // bytes the window holds (before this iteration's rewind)
    const uint32_t bs_old = bs, p_old = p;
    F = flush ? 12u : (F | (sentE ? 1u : 0u));
    bs = flush ? p_old : bs;

    tp = flush ? p_old : tp;  // (a fused cell with p <= tp takes its epsilon arc without a rewind)
    tp = __builtin_amdgcn_inverse_ballot_w64(m_skip) ? pn : tp;
    // the epsilon slot: dropped by a backtrack and by every epsilon step; a fused cell remembers the state it
    // read its rune in (the epsilon target, at p_old) if that state has an epsilon arc
    const bool he2 = comp && via <= n_eps;
    eps_t = he2 ? via : ((backtrack || epsE) ? 0u : eps_t);
    eps_p = he2 ? p_old : eps_p;
    const uint32_t e_cur = e, e_next = en;
    p = p_n; t = t_n; e = e_n; x = x_n; en = lut[code_n];
    // everything that happens less than once per token: hard fail, EOT, the first rewind at or behind the end of
    // the chunk (a fused cell's too: the lane then ends BEFORE the cell's rune), the window limit, the lookup cap,
    // the reader at EOF
    const bool eot_now = advance && ((e_cur >> DTK_SYM_CLS_SHIFT) & 3u) == 1u;  // matrix.go:593-605
    const bool at_stop = flush && beyond;
    // (one chain of bit operations: `||` makes the compiler branch between the tests)
    if (hardfail | eot_now | (beyond & (FIRST ? epsE : flush)) | (flush & (win > DTK_WINDOW)) | (win > DTK_WINDOW_BYTES) |
        over | ((p >= len) & !backtrack)) {
      if (FIRST && MODE != MODE_START && beyond && !(DTK_KO & 1)) {  // what the common path left to this block
        if (flush) sink.template token<IS_MATRIX>(bs_old, tp_old, p_old, ((F_old ^ 4u) & 7u) != 0);
        if (sentE) sink.template sentence<IS_MATRIX>(bs_old, p_old, (F_old & 8u) != 0);
      }
      if (flush && win > DTK_WINDOW && count_runes(s, bs_old, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
      // `eot` of the reference survives a hard fail (matrix.go:499-552 does not clear it; the next rune does): if that
      // rune was the document's last, the first successful epsilon step of the EOF drain -- right below: the hard fail
      // has dropped the epsilon slot, so the drain either takes that step or ends the walk -- fires the EOT's
      // SentenceEnd / TextEnd
      bool eot_stale = false;
      if (at_stop) {
        // the state right after the rewind at p_old: the target of the epsilon arc
        fin.p = p_old; fin.t = comp ? via : tgt; fin.aux = 0;
        fin.flags = init.flags & LANE_F_OK;
        done = true;
      } else {
        if (hardfail) {  // matrix.go:499-552: drop what is buffered as a token, restart at state 1
          if (r) { st |= ST_BAD_MODEL; done = true; }
          else {
            if (p <= tp) { p = pn; }  // matrix.go:515-516
            if (p < tp) st |= ST_BAD_OFFSET;  // Token(bufft, buffer[:buffc]) with bufft > buffc
            e = p == pn ? e_next : e_cur;     // the rune at p (read again if it was not consumed)
            eot_stale = p == pn && p >= len && ((e_cur >> DTK_SYM_CLS_SHIFT) & 3u) == 1u;
            if (MODE != MODE_START) sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
            F = 12u;
            if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
            t = tr.start; eps_t = 0;
            tp = p; bs = p;
            // (with a stale `eot` the lane goes on into the EOF drain itself: a start record has no place for it)
            if (MODE != MODE_DOC && p >= stop_pos && !eot_stale) {
              fin.p = p; fin.t = t; fin.aux = 0;
              fin.flags = (init.flags & LANE_F_OK);
              done = true;
            }
          }
        }
        if (eot_now) {
          if (MODE != MODE_START) sink.template eot<IS_MATRIX>(bs, p, (F & 1u) == 0u, (F & 8u) != 0);
          F = (F & 4u) | 3u;  // sentenceEnd, textEnd; TextEnd: pos = pos[:0] (token_writer.go:158)
          if (IS_MATRIX) {    // matrix.go:601 rewinds; the double array keeps window and epsilon slot (datok.go:1019-1030)
            eps_t = 0;
            if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
            tp = p; bs = p;
            if (MODE != MODE_DOC && p >= stop_pos && !done) {
              fin.p = p; fin.t = t; fin.aux = 0;
              fin.flags = (F & 3u) | (init.flags & LANE_F_OK);
              done = true;
            }
          }
        }
        if (over && !done) { st |= ST_STEP_LIMIT; done = true; }
        // More bytes buffered than 1024 runes can have: the reference's window has overflowed for certain
        // (matrix.go:365,406).  The lane stops there -- a blank-free blob of megabytes would otherwise be walked to
        // its end by every lane whose chunk lies inside it.  (The tail below then closes the document at this
        // position: every document keeps its TextEnd.)
        if (hi - bs > DTK_WINDOW_BYTES && !done) { st |= ST_WINDOW_OVERFLOW; done = true; }
      }
      if (!done && !backtrack) { DTK_EOF_DRAIN() }
      if (!done) {  // ask again: state, position or entry may have changed
        x = DTK_TAB(t, e);
        const uint32_t pn2 = p + ((e >> DTK_SYM_W_SHIFT) & 7u);
        if (pn2 - wb7 >= DTK_WIN8) DTK_REFILL(pn2)
        en = DTK_ENTRY(pn2);
      }
    }
#ifdef DTK_PROBE
    {  // the end of the iteration: what is left of the wait for the next cell and entry
      const unsigned long long a_ = clock64();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      pr_wait += clock64() - a_;
      pr_n++;
    }
#endif
  }
#undef DTK_TAB
#undef DTK_ENTRY
#undef DTK_REFILL
#ifdef DTK_PROBE
  {
    // (the wave's clock: every lane reads the same counter; the wave leaves the loop with its last lane)
    const unsigned long long tot_ = clock64() - pr_t0;
    unsigned long long wmax = pr_wait, nmax = pr_n;
    for (int o = 32; o; o >>= 1) {
      const unsigned long long a_ = __shfl_xor(wmax, o), b_ = __shfl_xor(nmax, o);
      wmax = a_ > wmax ? a_ : wmax; nmax = b_ > nmax ? b_ : nmax;
    }
    if (MODE == MODE_CHUNK && lane_id() == 0) {
      atomicAdd(&g_probe[0], wmax); atomicAdd(&g_probe[1], tot_); atomicAdd(&g_probe[2], nmax); atomicAdd(&g_probe[3], 1ull);
    }
    if (MODE == MODE_START && lane_id() == 0) {
      atomicAdd(&g_probe[4], wmax); atomicAdd(&g_probe[5], tot_); atomicAdd(&g_probe[6], nmax); atomicAdd(&g_probe[7], 1ull);
    }
  }
#endif
#undef DTK_EOF_DRAIN

  if (fin.p == 0xFFFFFFFFu && !(st & (ST_STEP_LIMIT | ST_BAD_MODEL))) {
    if (hi - bs > DTK_WINDOW && count_runes(s, bs, hi) > DTK_WINDOW) st |= ST_WINDOW_OVERFLOW;
    if (MODE != MODE_START) {
      if (p > tp) {  // matrix.go:671-678
        sink.template token<IS_MATRIX>(bs, tp, p, ((F ^ 4u) & 7u) != 0);
        F = (F & ~3u) | 8u;
      }
      sink.template tail<IS_MATRIX>(bs, p, (F & 1u) != 0, (F & 2u) != 0, (F & 8u) != 0);  // matrix.go:683-691
    }
  }
  st_out = st;
  steps_out = cap - budget;  // lookups (modulo 2^32: the budget wraps when it runs out)
}

// the lean walk: fused cells and no arc on `unknown` (MatrixLeanTrans, picked by the launcher)
template <typename TRANS, bool IS_MATRIX, int MODE, bool FIRST = false>
__device__ __forceinline__ void walk_any(const TRANS &tr, const DtkSym &sym, uint64_t off,
                                         uint32_t len, DtkLaneState init, uint32_t stop_pos, EventSink &sink,
                                         uint32_t epsilon, uint32_t unknown, uint32_t identity, uint32_t cap,
                                         DtkLaneState &fin, uint32_t &st_out, uint32_t &steps_out,
                                         uint16_t *win_row, const uint16_t *lut) {
  if constexpr (TRANS::LEAN)
    walk_fused<MODE, FIRST, IS_MATRIX>(tr, sym, off, len, init, stop_pos, sink, epsilon, cap, fin, st_out, steps_out, win_row, lut);
  else
    walk_lane<TRANS, IS_MATRIX, MODE>(tr, sym, off, len, init, stop_pos, sink, epsilon, unknown, identity, cap,
                                      fin, st_out, steps_out, win_row);
}

__device__ __forceinline__ uint32_t step_cap(uint32_t factor, uint32_t len) {
  unsigned long long c = (unsigned long long)factor * ((unsigned long long)len + 2ull);
  return c > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)c;
}

__device__ __forceinline__ void add_steps(unsigned long long *counter, uint32_t mine) {
  unsigned long long tot = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
  if (lane_id() == 0 && tot) atomicAdd(counter + (blockIdx.x & (DTK_STEP_STRIPES - 1u)) * 16u, tot);  // see DTK_STEP_STRIPES
}

// ---- the exact pass: one lane per irregular document ----
//
// NewTokenWriter (token_writer.go:36-175) with TOKEN_POS | SENTENCE_POS semantics, fed by the walk in the
// reference's own call order: what the compaction derives from position-indexed event bytes for every
// other document.  Writes the document's rows of the result arrays and, call by call, the list a closure
// replay needs (kind 0 Token: a = byte position of buffer[0], b = of buffer[offset], c = end;
// kind 1 SentenceEnd(a); kind 2 TextEnd(a) -- the int arguments as upstream: matrix.go:575,597,600,684,691
// pass buffc, datok.go:1015,1026,1119,1127 pass 0 and only :1023 buffc).
struct ExactSink {
  DtkSymAt s;          // the document's stretch of the symbol stream (rune starts)
  const uint8_t *txt;  // the document's bytes
  bool nl_rule, write;
  DtkCall *log;
  int32_t *rstart, *rend, *sent;
  uint32_t *bstart, *bend, *ttok, *tsent, *sbefore, *ts_end;
  uint32_t tok_n, sent_n, text_n;  // row lengths
  // token_writer.go:38-42
  int32_t posC, last_rend;
  bool init, sentB;
  uint32_t n_tok, n_sent, n_text, n_sev, text_tok0, n_calls;
  uint32_t st;
  __device__ __forceinline__ void start() {
    posC = 0; last_rend = 0; init = true; sentB = true;
    n_tok = n_sent = n_text = n_sev = text_tok0 = n_calls = 0; st = 0;
  }
  __device__ __forceinline__ int32_t runes(uint32_t from, uint32_t to) const {
    return to > from ? (int32_t)count_runes(s, from, to) : 0;
  }
  __device__ __forceinline__ void call(uint32_t kind, int32_t a, uint32_t b, uint32_t c) {
    if (write) log[n_calls] = DtkCall{kind, a, b, c};
    n_calls++;
  }
  __device__ __forceinline__ void push_sent(int32_t v) {
    if (write) { if (n_sent < sent_n) sent[n_sent] = v; else st |= ST_INTERNAL; }
    n_sent++;
  }
  __device__ __forceinline__ void sentence_end(int32_t arg) {  // token_writer.go:104-115
    call(1u, arg, 0u, 0u);
    n_sev++;
    if (n_tok == text_tok0) st |= ST_EMPTY_TEXT; else push_sent(last_rend);
    sentB = true;
  }
  __device__ __forceinline__ void text_end(int32_t arg) {  // token_writer.go:131-159
    call(2u, arg, 0u, 0u);
    if (n_tok == text_tok0) st |= ST_EMPTY_TEXT;
    if (write) {
      if (n_text < text_n) {
        ttok[n_text] = n_tok; tsent[n_text] = n_sent;
        if (ts_end) ts_end[n_text] = n_sev;
      } else st |= ST_INTERNAL;
    }
    n_text++;
    sentB = true; posC = 0; text_tok0 = n_tok;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void token(uint32_t bs, uint32_t tp, uint32_t p, bool) {  // token_writer.go:58-88
    call(0u, (int32_t)bs, tp, p);
    const int32_t off_r = runes(bs, tp), len_r = runes(bs, p);  // offset, len(buf)
    if (posC == 0 && nl_rule && p > bs && txt[bs] == '\n' && !init) posC--;  // :66-68
    init = false;
    posC += off_r;
    const int32_t rs = posC;
    if (sentB) { sentB = false; push_sent(rs); }
    posC += len_r - off_r;
    last_rend = posC;
    if (write) {
      if (n_tok < tok_n) {
        rstart[n_tok] = rs; rend[n_tok] = posC;
        bstart[n_tok] = tp < p ? tp : p; bend[n_tok] = p;  // an empty surface: the empty range at the end of the buffer
        if (sbefore) sbefore[n_tok] = n_sev;
      } else st |= ST_INTERNAL;
    }
    n_tok++;
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void eot(uint32_t bs, uint32_t p, bool with_sentence, bool) {
    const int32_t buffc = runes(bs, p);
    if (with_sentence) sentence_end(buffc);      // matrix.go:597 / datok.go:1023
    text_end(IS_MATRIX ? buffc : 0);             // matrix.go:600 / datok.go:1026
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void sentence(uint32_t bs, uint32_t p, bool) {
    sentence_end(IS_MATRIX ? runes(bs, p) : 0);  // matrix.go:575 / datok.go:1015
  }
  template <bool IS_MATRIX>
  __device__ __forceinline__ void tail(uint32_t bs, uint32_t p, bool sentence_end_, bool text_end_, bool) {
    const int32_t arg = IS_MATRIX ? runes(bs, p) : 0;
    if (!sentence_end_) sentence_end(arg);       // matrix.go:683-684 / datok.go:1118-1119
    if (!text_end_) text_end(arg);               // matrix.go:690-691 / datok.go:1126-1127
  }
  __device__ __forceinline__ void out_of_order() {}  // (call order is what this sink records)
};

// the lanes' windows of the symbol stream and, for the lean loop, the model's code table (one wave per block)
#define DTK_WINDOWS(TRANS, SYM)                                                                       \
  constexpr uint32_t WIN_ROW_ = TRANS::LEAN ? (DTK_WIN8 + 8u) / 2u : DTK_WIN_ROW;                     \
  __shared__ uint16_t s_win[WAVE * WIN_ROW_];                                                         \
  __shared__ uint16_t s_lut[TRANS::LEAN ? 256 : 1];                                                   \
  uint16_t *win_row = s_win + threadIdx.x * WIN_ROW_;                                                 \
  if constexpr (TRANS::LEAN) {                                                                        \
    for (uint32_t i_ = threadIdx.x; i_ < 256u; i_ += WAVE) s_lut[i_] = (SYM).lut[i_];                 \
    __syncthreads();                                                                                  \
  }

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_exact_doc(TRANS tr, DtkExactArgs X, uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  __shared__ uint16_t s_win[WAVE * DTK_WIN_ROW];
  uint16_t *win_row = s_win + threadIdx.x * DTK_WIN_ROW;
  const uint32_t i = blockIdx.x * WAVE + threadIdx.x;
  if (i >= X.n) return;
  const uint32_t d = X.docs[i];
  const uint64_t off = X.doc_off[d];
  const uint32_t len = (uint32_t)(X.doc_off[d + 1] - off);
  ExactSink sink;
  sink.s = DtkSymAt{X.sym, off}; sink.txt = X.text + off;
  sink.nl_rule = (X.flags & 16u) != 0; sink.write = X.pass != 0;
  sink.log = X.pass ? X.calls + X.call_off[i] : nullptr;
  const uint64_t t0 = X.tok_off[d], s0 = X.sent_off[d], x0 = X.text_off[d];
  sink.rstart = X.tok_rstart + t0; sink.rend = X.tok_rend + t0;
  sink.bstart = X.tok_bstart + t0; sink.bend = X.tok_bend + t0;
  sink.sbefore = X.tok_sbefore ? X.tok_sbefore + t0 : nullptr;
  sink.sent = X.sent + s0;
  sink.ttok = X.text_tok_end + x0; sink.tsent = X.text_sent_end + x0;
  sink.ts_end = X.text_s_end ? X.text_s_end + x0 : nullptr;
  sink.tok_n = (uint32_t)(X.tok_off[d + 1] - t0); sink.sent_n = (uint32_t)(X.sent_off[d + 1] - s0);
  sink.text_n = (uint32_t)(X.text_off[d + 1] - x0);
  sink.start();
  DtkLaneState init{0u, tr.start_state(), tr.start_aux(), 0u}, fin;
  uint32_t st = 0, steps = 0;
  walk_lane<TRANS, IS_MATRIX, MODE_DOC, ExactSink>(tr, X.sym, off, len, init, 0u, sink, epsilon, unknown, identity,
                                                   step_cap(X.step_factor, len), fin, st, steps, win_row);
  if (X.pass == 0) { X.n_calls[i] = sink.n_calls; return; }
  // the rows were sized by the first walk's counts: both walks make the same calls
  if (sink.n_tok != sink.tok_n || sink.n_sent != sink.sent_n || sink.n_text != sink.text_n) st |= ST_INTERNAL;
  X.status[d] = st | sink.st;
  if (X.doc_ns) X.doc_ns[d] = sink.n_sev;
}

// ---- one document per lane (no speculation) ----
template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_walk_doc(TRANS tr, DtkWalkArgs A, uint32_t epsilon,
                                                   uint32_t unknown, uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  const uint32_t d = blockIdx.x * WAVE + threadIdx.x;
  uint32_t steps = 0;
  if (d < A.n_docs) {
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    EventSink sink;
    sink.init(A, off, d, 0u, 0xFFFFFFFFu);  // (documents of any length: the bits go straight to memory)
    DtkLaneState init{0u, tr.start_state(), tr.start_aux(), 0u}, fin;
    uint32_t st;
    walk_any<TRANS, IS_MATRIX, MODE_DOC>(tr, A.sym, off, len, init, 0u, sink, epsilon, unknown,
                                          identity, step_cap(A.step_factor, len), fin, st, steps, win_row, s_lut);
    A.status[d] = st | sink.st;
    A.tok_cnt[d] = sink.c_tok; A.sent_cnt[d] = sink.c_sent; A.text_cnt[d] = sink.c_text;
  }
  add_steps(A.steps, steps);
}

// ---- speculative chunk lanes ----
//
// Lane (d, k) covers the rewinds ("sync points": the moments the reference
// rewinds its window, where the whole loop state is (position, state, three
// flags)) that fall into [k*C, (k+1)*C) of document d.
//   k_spec_start : lane k >= 1 walks from k*C - W in the start state and records
//                  the first sync point at or after k*C  (the automaton
//                  re-synchronises within a token or two).
//   k_spec_link  : per document, the first lane without a linked successor.
//   k_spec_walk  : each lane walks from its record to the next lane's record,
//                  storing events inside its window only.
//   k_spec_verify: per lane, verifies that it arrived exactly at its
//                  successor's record (position, state, flags).  A document that
//                  fails is repaired from the first bad lane on (host loop,
//                  normally never entered) -- the result is exact either way.

#ifdef DTK_PROBE
__shared__ unsigned long long s_probe_mark;
__device__ unsigned long long g_phase[8];  // cycles per wave: prologue+search, warm-up walk, chunk walk, epilogue; waves
extern "C" int dtk_phase_read(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
  return 0;
}
#endif

// the start record of lane k >= 0 of document d (k_spec_start, k_spec_both)
template <typename TRANS, bool IS_MATRIX>
__device__ __forceinline__ DtkLaneState start_record(const TRANS &tr, const DtkWalkArgs &A, const DtkSpecArgs &S,
                                                     uint32_t k, uint64_t off, uint32_t len, uint32_t epsilon,
                                                     uint32_t unknown, uint32_t identity, uint16_t *win_row,
                                                     const uint16_t *s_lut, uint32_t &steps) {
  DtkLaneState rec{0u, tr.start_state(), tr.start_aux(), 0u};
  if (k > 0) {
    const uint32_t kc = k * S.chunk;
    uint32_t sp = kc > S.warm ? kc - S.warm : 0u;
    if (sp > 0 && S.warm_ws && S.text) {
      // The walk re-synchronises at token boundaries, and blanks are boundaries in every
      // tokenizer of this kind: start behind the warm_ws-th run of blanks before the chunk
      // instead of a fixed distance (never further back than `warm`).  A wrong guess only
      // costs a repair round.
      const uint8_t *tx = S.text + off;
      uint32_t runs = 0, q = kc > S.warm_min ? kc - S.warm_min : 0u;
      if (q < sp) q = sp;
      bool in_ws = false;
      while (q > sp) {
        const uint8_t c = tx[q - 1u];
        const bool ws = c == ' ' || c == '\n' || c == '\t' || c == '\r';
        if (in_ws && !ws) { if (++runs == S.warm_ws) break; }
        in_ws = ws;
        q--;
      }
      sp = q;  // first byte of the run of blanks (the walk skips them), or the fixed start
    }
    if (sp > 0 && S.warm_extend && S.text) {
      // A start inside a long blank-free token (a URL, say) makes the warm-up invent token ends the
      // real walk does not have -- a repair round.  Blanks are token boundaries in every tokenizer of
      // this kind: move the start back to the previous blank (typically half a dozen bytes; at
      // most warm_extend).
      // (8 bytes per step: the dwords around them are read whole, a blank is found with byte-wise
      // zero tests on  bytes ^ pattern)
      const uint8_t *tx = S.text + off;
      // bytes tx[q-8 .. q-1] as one 64-bit word, tx[q-1] on top
      auto load8 = [&](uint32_t q) -> uint64_t {
        if (off + q >= 16u) {
          const uint64_t a0 = reinterpret_cast<uint64_t>(tx + q) - 8u;  // address of the first of the 8 bytes
          const uint32_t sh = (uint32_t)(a0 & 3u) * 8u;
          const uint32_t *wp = reinterpret_cast<const uint32_t *>(a0 & ~3ull);
          const uint64_t lo64 = (uint64_t)wp[0] | ((uint64_t)wp[1] << 32);
          return sh ? (lo64 >> sh) | ((uint64_t)wp[2] << (64u - sh)) : lo64;
        }
        uint64_t w = 0;  // the first bytes of the batch: one by one
        for (uint32_t k = 0; k < 8u; k++)
          if (q + k >= 8u) w |= (uint64_t)tx[q + k - 8u] << (8u * k);
        return w;
      };
      auto zb = [](uint64_t x) {
        return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
      };
      const uint32_t lim = sp > S.warm_extend ? sp - S.warm_extend : 0u;
      uint32_t q = sp;
      while (q > lim) {
        const uint64_t w = load8(q);
        uint64_t m = zb(w ^ 0x2020202020202020ull) | zb(w ^ 0x0A0A0A0A0A0A0A0Aull) |
                     zb(w ^ 0x0909090909090909ull) | zb(w ^ 0x0D0D0D0D0D0D0D0Dull);
        if (q < 8u) m &= ~0ull << ((8u - q) * 8u);  // bytes before the document do not count
        if (m) {  // start behind the last blank
          q = q - 8u + (7u - ((uint32_t)__clzll((long long)m) >> 3)) + 1u;
          break;
        }
        q = q >= 8u ? q - 8u : 0u;
      }
      if (q < lim) q = lim;
      sp = q;
      // Inside a markup tag (<a href="...">, <!-- a comment -->) blanks are no token boundaries: if the nearest
      // angle bracket behind the start is an opening one, the warm-up starts at it (text with a tag every few hundred
      // bytes otherwise costs a repair round in every batch).  At most DTK_WARM_TAG bytes back.
      // (all loads first: one after the other they cost a cache round trip each, 8 us per batch on plain text,
      //  where no bracket ends the search early)
      uint64_t wt[DTK_WARM_TAG / 8u];
#pragma unroll
      for (uint32_t i = 0; i < DTK_WARM_TAG / 8u; i++) wt[i] = sp > 8u * i ? load8(sp - 8u * i) : 0ull;
#pragma unroll
      for (uint32_t i = 0; i < DTK_WARM_TAG / 8u; i++) {
        const uint32_t qq = sp > 8u * i ? sp - 8u * i : 0u;  // wt[i] = bytes qq-8 .. qq-1
        uint64_t mo = zb(wt[i] ^ 0x3C3C3C3C3C3C3C3Cull), mc = zb(wt[i] ^ 0x3E3E3E3E3E3E3E3Eull);
        if (qq < 8u) { const uint64_t in = qq ? ~0ull << ((8u - qq) * 8u) : 0ull; mo &= in; mc &= in; }
        if (mo | mc) {
          if (mo > mc) sp = qq - 8u + (7u - ((uint32_t)__clzll((long long)mo) >> 3));  // the nearest one opens a tag
          break;
        }
      }
    }
#ifdef DTK_PROBE
    s_probe_mark = clock64();  // (the wave is in lockstep: every lane that comes here writes the same time)
#endif
    EventSink sink;  // (a warm-up reports nothing)
    sink.g = nullptr; sink.lds = (dtk_lds_u32 *)nullptr; sink.tailw = nullptr; sink.lw = 0; sink.lo = sink.hi = 0;
    uint32_t st;
    if (sp > 0) {
      while (sp < len && !dtk_sym_is_start(A.sym, off + sp)) sp++;
      DtkLaneState init{sp, tr.start_state(), tr.start_aux(), 0u};
      walk_any<TRANS, IS_MATRIX, MODE_START>(tr, A.sym, off, len, init, kc, sink, epsilon, unknown,
                                              identity, step_cap(A.step_factor, len), rec, st, steps, win_row, s_lut);
    } else {
      // sp == 0: the walk from the true initial state; its first sync point at/after kc
      walk_any<TRANS, IS_MATRIX, MODE_START>(tr, A.sym, off, len, rec, kc, sink, epsilon, unknown,
                                              identity, step_cap(A.step_factor, len), rec, st, steps, win_row, s_lut);
    }
  }
  return rec;
}

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_spec_start(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                     uint32_t epsilon, uint32_t unknown,
                                                     uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const uint32_t k = L - S.chunk_off[d];
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    S.lane_start[L] = start_record<TRANS, IS_MATRIX>(tr, A, S, k, off, len, epsilon, unknown, identity, win_row, s_lut, steps);
  }
  add_steps(A.steps, steps);
}

// First pass in one launch: the lane finds its start record (as k_spec_start) and walks on from it
// to its first sync point at or behind the end of its chunk -- which is the record its successor
// finds for itself if the speculation holds, and k_spec_verify checks exactly that.  Every lane
// with a record walks; what lanes behind a broken chain stored is cleared by the repair round.
extern __shared__ uint32_t s_dyn_bits[];  // the wave's event bitmaps (3 kinds x S.lds_words), if any

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_spec_both(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                    uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  uint32_t *lds_bits = S.lds_words ? s_dyn_bits : nullptr;
#ifdef DTK_PROBE
  const unsigned long long pt0 = clock64();
  unsigned long long pt1 = pt0, pt2 = pt0;
  s_probe_mark = pt0;
#endif
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  const uint32_t w0 = S.lds_words ? lds_bits_word0(A, S, blockIdx.x * WAVE) : 0u;
  if (lds_bits) lds_bits_clear(lds_bits, S.lds_words);
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const uint32_t k = L - S.chunk_off[d];
    const uint64_t off = A.doc_off[d];
    const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
    const DtkLaneState rec =
        start_record<TRANS, IS_MATRIX>(tr, A, S, k, off, len, epsilon, unknown, identity, win_row, s_lut, steps);
#ifdef DTK_PROBE
    pt1 = clock64();
#endif
    S.lane_start[L] = rec;
    DtkLaneState fin{0xFFFFFFFFu, 0u, 0u, LANE_F_IDLE};
    DtkLaneCount cnt{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    const uint32_t stop = (L + 1u < S.chunk_off[d + 1]) ? (k + 1u) * S.chunk : 0xFFFFFFFFu;
    if (rec.p != 0xFFFFFFFFu && rec.p >= stop) {
      // my first sync point lies behind my whole chunk (a token longer than a chunk): it is my
      // successor's record too, and I own nothing
      fin = rec;
      fin.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
    } else if (rec.p != 0xFFFFFFFFu) {
      EventSink sink;
      sink.init(A, off, d, rec.p, 0xFFFFFFFFu, lds_bits, S.lds_words, w0);
      uint32_t st = 0, steps2 = 0;
      walk_any<TRANS, IS_MATRIX, MODE_CHUNK, true>(tr, A.sym, off, len, rec, stop, sink, epsilon, unknown,
                                                    identity, step_cap(A.step_factor, len), fin, st, steps2, win_row, s_lut);
      steps += steps2;
      if (sink.dropped) fin.flags |= LANE_F_DROPPED;
      cnt.tok = sink.c_tok; cnt.sent = sink.c_sent; cnt.text = sink.c_text; cnt.status = st | sink.st;
      cnt.sev = sink.c_sev; cnt.e_pos = sink.e_pos; cnt.e_tok = sink.e_tok;
    }
#ifdef DTK_PROBE
    pt2 = clock64();
#endif
    S.lane_end[L] = fin;
    S.lane_cnt[L] = cnt;
  }
  if (lds_bits) lds_bits_flush(lds_bits, S.lds_words, A.bits, A.bit_words, w0);
  add_steps(A.steps, steps);
#ifdef DTK_PROBE
  {
    __syncthreads();
    const unsigned long long pt3 = clock64(), mark = s_probe_mark;
    if (threadIdx.x == 0) {
      atomicAdd(&g_phase[0], mark - pt0); atomicAdd(&g_phase[1], pt1 - mark); atomicAdd(&g_phase[2], pt2 - pt1);
      atomicAdd(&g_phase[3], pt3 - pt2); atomicAdd(&g_phase[4], 1ull);
    }
  }
#endif
}

// Every lane derives its window from the start records and first_bad[d] (k_spec_link), in the
// first pass and in repair rounds alike.
__device__ __forceinline__ DtkLanePlan plan_of(const DtkSpecArgs &S, uint32_t L, uint32_t d) {
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  const uint32_t k = L - L0, fb = ~S.first_bad[d];
  DtkLanePlan pl;
  pl.pad = 0;
  if (k < fb) {
    pl.stop = pl.wend = S.lane_start[L + 1].p;
    pl.mode = PLAN_CHAINED;
  } else if (k == fb) {
    // last enabled lane: stops at the first sync point behind its own chunk (or EOF)
    pl.stop = (L + 1 < L1) ? (k + 1u) * S.chunk : 0xFFFFFFFFu;
    pl.wend = 0xFFFFFFFFu;
    pl.mode = PLAN_LAST;
  } else {
    pl.stop = pl.wend = 0;
    pl.mode = PLAN_OFF;
  }
  return pl;
}

// One thread per lane: is my successor's record present and not before mine?
// first_bad[d] becomes the first chunk index without such a successor (stored
// bit-inverted so that a zero fill means "none yet"); the last lane of a document
// never has one.
__global__ __launch_bounds__(256) void k_spec_link(DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= S.n_lanes) return;
  const uint32_t d = S.lane_doc[L];
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  bool linked = false;
  if (L + 1 < L1) {
    const uint32_t np = S.lane_start[L + 1].p, mp = S.lane_start[L].p;
    linked = np != 0xFFFFFFFFu && mp != 0xFFFFFFFFu && np >= mp;
  }
  if (!linked) atomicMax(&S.first_bad[d], ~(L - L0));  // stored inverted: zero fill = none
}

template <typename TRANS, bool IS_MATRIX>
__global__ __launch_bounds__(WAVE) void k_spec_walk(TRANS tr, DtkWalkArgs A, DtkSpecArgs S,
                                                    uint32_t epsilon, uint32_t unknown,
                                                    uint32_t identity) {
  DTK_WINDOWS(TRANS, A.sym)
  if (S.go && *S.go == 0u) return;
  uint32_t *lds_bits = S.lds_words ? s_dyn_bits : nullptr;
  const uint32_t L = blockIdx.x * WAVE + threadIdx.x;
  const uint32_t w0 = S.lds_words ? lds_bits_word0(A, S, blockIdx.x * WAVE) : 0u;
  if (lds_bits) lds_bits_clear(lds_bits, S.lds_words);
  uint32_t steps = 0;
  if (L < S.n_lanes) {
    const uint32_t d = S.lane_doc[L];
    const bool redo = S.redo_from != nullptr;
    if (!redo || (S.redo_from[d] != 0xFFFFFFFFu && L >= S.redo_from[d])) {
      const DtkLanePlan pl = plan_of(S, L, d);
      DtkLaneState fin{0xFFFFFFFFu, 0u, 0u, LANE_F_IDLE};
      DtkLaneCount cnt{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
      const DtkLaneState init0 = pl.mode == PLAN_CHAINED ? S.lane_start[L] : DtkLaneState{0xFFFFFFFFu, 0u, 0u, 0u};
      if (pl.mode == PLAN_CHAINED && init0.p != 0xFFFFFFFFu && init0.p >= pl.stop) {
        // my record is my successor's too (a token longer than a chunk): nothing of it is mine
        fin = init0;
        fin.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
      } else if (pl.mode != PLAN_OFF) {
        const uint64_t off = A.doc_off[d];
        const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
        const DtkLaneState init = S.lane_start[L];
        EventSink sink;
        sink.init(A, off, d, init.p, pl.wend, lds_bits, S.lds_words, w0);
        uint32_t st = 0;
        walk_any<TRANS, IS_MATRIX, MODE_CHUNK>(tr, A.sym, off, len, init, pl.stop, sink, epsilon, unknown,
                                                identity, step_cap(A.step_factor, len), fin, st, steps, win_row, s_lut);
        if (sink.dropped) fin.flags |= LANE_F_DROPPED;
        cnt.tok = sink.c_tok; cnt.sent = sink.c_sent; cnt.text = sink.c_text; cnt.status = st | sink.st;
        cnt.sev = sink.c_sev; cnt.e_pos = sink.e_pos; cnt.e_tok = sink.e_tok;
      }
      S.lane_end[L] = fin;
      S.lane_cnt[L] = cnt;
    }
  }
  if (lds_bits) lds_bits_flush(lds_bits, S.lds_words, A.bits, A.bit_words, w0);
  add_steps(A.steps, steps);
}

// One thread per lane: did I arrive exactly at my successor's record (position,
// state, flags)?  Lanes that did add their counts / status to the document; the
// first lane that did not is recorded (bit-inverted, zero = none) in fail_lane[d].
//
// local_link (the first pass, where every lane with a record has walked to the first sync point behind its own chunk):
// the lane decides from its own and its successor's record what k_spec_link + first_bad[d] decide otherwise -- one
// launch and a round of atomics less.  A lane with a record either has a successor record at or behind its own (then it
// must have arrived exactly there) or it is the chain's last lane (then it must have reached EOF, and no later lane
// may have a record: the lane in front of such a record, which has none itself, finds the chain's last lane by
// walking back).  The first lane that fails is the same one in both formulations.
// (256 threads.  With 1024 -- 16 waves adding up before their atomics -- one long document verified faster still, but
//  with three batches in flight the blocks have to wait for four free wave slots on every SIMD of one CU, which the
//  other batches' walks, six waves per SIMD, rarely leave: 154 -> 143 GB/s.)
#define VERIFY_TB 256u
__global__ __launch_bounds__(VERIFY_TB) void k_spec_verify(DtkWalkArgs A, DtkSpecArgs S, uint32_t cmp_mask, uint32_t local_link) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = L < S.n_lanes;  // every lane stays for the wave reduction below
  uint32_t d = 0xFFFFFFFFu;
  DtkLaneCount c{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
  if (live) {
    d = S.lane_doc[L];
    if (S.redo_from && S.redo_from[d] == 0xFFFFFFFFu) { live = false; d = 0xFFFFFFFFu; }  // a repair round: not this document
  }
  if (live && local_link) {
    const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
    const bool has_next = L + 1u < L1;
    const DtkLaneState own = S.lane_start[L], en = S.lane_end[L];
    DtkLaneState nx{0xFFFFFFFFu, 0u, 0u, 0u};
    if (has_next) nx = S.lane_start[L + 1];
    uint32_t link_ok = 1u;
    if (own.p == 0xFFFFFFFFu) {
      // behind the chain: legitimate only if the chain ran to EOF and no later lane found a sync point
      if (nx.p != 0xFFFFFFFFu) {
        uint32_t x = L - 1u;  // (lane 0 of a document always has a record)
        while (x > L0 && S.lane_start[x].p == 0xFFFFFFFFu) x--;
        atomicMax(&S.fail_lane[d], ~x);
      }
    } else {
      bool good;
      if (nx.p != 0xFFFFFFFFu && nx.p >= own.p) {
        good = en.p == nx.p && en.t == nx.t && ((en.flags ^ nx.flags) & cmp_mask) == 0 && !(en.flags & LANE_F_DROPPED);
      } else {
        // the chain's last lane must reach EOF (a successor record before my own breaks the chain right here)
        good = nx.p == 0xFFFFFFFFu && en.p == 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE);
      }
      if (good) c = S.lane_cnt[L]; else atomicMax(&S.fail_lane[d], ~L);
      link_ok = good ? 1u : 0u;
    }
    S.lane_plan[L].pad = link_ok;
  } else if (live) {
    const uint32_t L0 = S.chunk_off[d];
    const uint32_t k = L - L0, fb = ~S.first_bad[d];
    uint32_t link_ok = 1u;  // did I arrive exactly at my successor's record (k_redo_spread reads it)
    if (k > fb) {
      // a lane behind the chain: legitimate only if the chain ran to EOF and I found no sync point
      if (S.lane_start[L].p != 0xFFFFFFFFu) atomicMax(&S.fail_lane[d], ~(L0 + fb));
    } else {
      const DtkLaneState en = S.lane_end[L];
      bool good;
      if (k < fb) {
        const DtkLaneState nx = S.lane_start[L + 1];
        good = en.p == nx.p && en.t == nx.t && ((en.flags ^ nx.flags) & cmp_mask) == 0 &&
               !(en.flags & LANE_F_DROPPED);
      } else {
        good = en.p == 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE);  // the chain's last lane must reach EOF
      }
      if (good) c = S.lane_cnt[L]; else atomicMax(&S.fail_lane[d], ~L);
      link_ok = good ? 1u : 0u;
    }
    S.lane_plan[L].pad = link_ok;
  }
  // The lanes of a document are consecutive: add up the counts of each run of equal
  // documents inside the wave, then one atomic per run instead of one per lane.
  uint32_t tok = c.tok, sent = c.sent, text = c.text, st = c.status;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const uint32_t dn = __shfl_down(d, o);
    const uint32_t t2 = __shfl_down(tok, o), s2 = __shfl_down(sent, o), x2 = __shfl_down(text, o),
                   st2 = __shfl_down(st, o);
    const bool same = (lane_id() + o < WAVE) && dn == d;
    tok += same ? t2 : 0u; sent += same ? s2 : 0u; text += same ? x2 : 0u; st |= same ? st2 : 0u;
  }
  const uint32_t dprev = __shfl_up(d, 1);
  const bool head = live && (lane_id() == 0 || dprev != d);
  // A block whose waves all lie inside one document adds up once more: the waves of a long document otherwise
  // queue their atomics at the same three addresses (one 64 MiB document: 4096 waves, 106 us of verification).
  __shared__ uint32_t s_d[VERIFY_TB / WAVE], s_v[VERIFY_TB / WAVE][4];
  const uint32_t wid = threadIdx.x >> 6;
  const bool whole = __ballot(head) == 1ull;  // one run of lanes, starting at lane 0
  if (lane_id() == 0) {
    s_d[wid] = whole ? d : 0xFFFFFFFFu;
    s_v[wid][0] = tok; s_v[wid][1] = sent; s_v[wid][2] = text; s_v[wid][3] = st;
  }
  __syncthreads();
  bool merged = s_d[0] != 0xFFFFFFFFu;
#pragma unroll
  for (uint32_t w = 1; w < VERIFY_TB / WAVE; w++) merged = merged && s_d[w] == s_d[0];
  if (merged) {
    if (threadIdx.x == 0) {
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (uint32_t w = 0; w < VERIFY_TB / WAVE; w++) { a0 += s_v[w][0]; a1 += s_v[w][1]; a2 += s_v[w][2]; a3 |= s_v[w][3]; }
      if (a0) atomicAdd((unsigned long long *)&A.tok_cnt[d], (unsigned long long)a0);
      if (a1) atomicAdd((unsigned long long *)&A.sent_cnt[d], (unsigned long long)a1);
      if (a2) atomicAdd((unsigned long long *)&A.text_cnt[d], (unsigned long long)a2);
      if (a3) atomicOr(&A.status[d], a3);
    }
  } else if (head) {
    if (tok) atomicAdd((unsigned long long *)&A.tok_cnt[d], (unsigned long long)tok);
    if (sent) atomicAdd((unsigned long long *)&A.sent_cnt[d], (unsigned long long)sent);
    if (text) atomicAdd((unsigned long long *)&A.text_cnt[d], (unsigned long long)text);
    if (st) atomicOr(&A.status[d], st);
  }
}

// One thread per document: nothing to do unless a lane failed; then the lane
// `bad` started from a true state (every lane before it checked out), so where it
// really ended is the true record of its successor: redo from `bad` on.
__device__ __forceinline__ void mark_redo(const DtkSpecArgs &S, uint32_t d, uint32_t bad, uint32_t *redo_out,
                                          uint32_t *n_bad) {
  const uint32_t L0 = S.chunk_off[d], L1 = S.chunk_off[d + 1];
  DtkLaneState en = S.lane_end[bad];
  en.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
  if (en.p != 0xFFFFFFFFu && bad + 1 < L1) S.lane_start[bad + 1] = en;
  // (ran to EOF: no later lane has a sync point -- k_redo_spread withdraws their records)
  // The round walks again from the last lane before `bad` that owns anything: it started from a true record as
  // well, and everything a lane behind the broken link can have reported lies behind that record -- the round
  // clears from there on without having to tell true reports from false ones.
  uint32_t r0 = bad;
  if (bad > L0) {
    r0 = bad - 1u;
    while (r0 > L0 && S.lane_end[r0].p == S.lane_start[r0].p) r0--;
  }
  redo_out[d] = r0;
  atomicAdd(n_bad, 1u);
}

__global__ __launch_bounds__(256) void k_spec_fix(DtkWalkArgs A, DtkSpecArgs S, uint32_t *redo_out,
                                                  uint32_t *n_bad) {
  if (S.go && *S.go == 0u) return;
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= A.n_docs) return;
  const uint32_t bad = ~S.fail_lane[d];
  if (bad == 0xFFFFFFFFu) { redo_out[d] = 0xFFFFFFFFu; return; }
  mark_redo(S, d, bad, redo_out, n_bad);
}

// ---- repair rounds (a document whose chain broke is redone from the last owning lane before its first bad lane on).
// All per lane / per document / per bitmap word, so that a long document repairs as fast
// as a batch of short ones:
//   k_redo_spread : the first bad lane started from a true state, so k_spec_fix made its end the
//                   record of its successor.  Further down the document, a lane whose predecessor
//                   arrived exactly but which itself missed its successor is in the same position
//                   with high probability: its end becomes its successor's record too (speculation
//                   again -- the next verification decides), so that one round repairs all isolated
//                   misses of a document, not just the first.
//   k_redo_reset  : per document, the counters and check words the round re-derives, and the tail word.
//   k_redo_clear  : event bits behind the record the round walks from.
// then k_spec_link, k_spec_walk (redone lanes only), k_spec_verify (repaired documents only), k_spec_fix.
__global__ __launch_bounds__(256) void k_redo_spread(DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= S.n_lanes) return;
  const uint32_t d = S.lane_doc[L];
  if (S.redo_from[d] == 0xFFFFFFFFu) return;
  const uint32_t bad = ~S.fail_lane[d];  // the first lane that missed (k_redo_reset clears the word afterwards)
  if (L <= bad) return;
  const uint32_t L1 = S.chunk_off[d + 1];
  if (S.lane_end[bad].p == 0xFFFFFFFFu) {  // the first bad lane ran to EOF: no later lane has a sync point
    S.lane_start[L].p = 0xFFFFFFFFu;
    return;
  }
  if (L + 1 >= L1) return;
  DtkLaneState en = S.lane_end[L];
  if (S.lane_plan[L].pad == 0u && S.lane_plan[L - 1].pad != 0u && L - 1 != bad &&
      en.p != 0xFFFFFFFFu && !(en.flags & LANE_F_IDLE)) {  // (a lane that overshot dropped events: still a true end)
    en.flags &= (LANE_F_SENT | LANE_F_TEXT | LANE_F_OK);
    S.lane_start[L + 1] = en;
  }
}

__global__ __launch_bounds__(256) void k_redo_reset(DtkWalkArgs A, DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= A.n_docs || S.redo_from[d] == 0xFFFFFFFFu) return;
  A.tok_cnt[d] = 0; A.sent_cnt[d] = 0; A.text_cnt[d] = 0; A.status[d] = 0;
  S.first_bad[d] = 0; S.fail_lane[d] = 0;
  A.doc_tail[d] = 0;
}

// One thread per bitmap word.  In a repaired document every bit behind the record the round walks from is
// cleared; at that very position only the opening kinds are (the closing kinds there were reported by the lane
// that stopped at it: a true report that nobody makes again).
__global__ __launch_bounds__(256) void k_redo_clear(DtkWalkArgs A, DtkSpecArgs S) {
  if (S.go && *S.go == 0u) return;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.bit_words) return;
  const uint64_t G0 = 32ull * j, G1 = G0 + 32ull;
  // the last document that starts at or before bit G0
  uint32_t lo = 0, hi = A.n_docs;
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (DTK_EV_BIT(A.doc_off[mid], mid) <= G0) lo = mid; else hi = mid;
  }
  uint32_t m_all = 0, m_open = 0;
  for (uint32_t d = lo; d < A.n_docs; d++) {
    const uint64_t key = DTK_EV_BIT(A.doc_off[d], d);
    if (key >= G1) break;
    const uint32_t r0 = S.redo_from[d];
    if (r0 == 0xFFFFFFFFu) continue;
    const uint32_t from = S.lane_start[r0].p;
    if (from == 0xFFFFFFFFu) continue;
    const uint64_t Gf = key + from, Ge = DTK_EV_BIT(A.doc_off[d + 1], d + 1);  // bits (Gf, Ge) and, opening kinds, Gf
    const uint64_t a = Gf + 1 > G0 ? Gf + 1 : G0, b = Ge < G1 ? Ge : G1;
    if (a < b) {
      const uint32_t n = (uint32_t)(b - a), sh = (uint32_t)(a - G0);
      const uint32_t m = (n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << sh;
      m_all |= m; m_open |= m;
    }
    if (Gf >= G0 && Gf < G1) m_open |= 1u << (uint32_t)(Gf - G0);
  }
  if (m_all) {
    A.bits[EVB_END * A.bit_words + j] &= ~m_all;
    A.bits[EVB_TEOT * A.bit_words + j] &= ~m_all;
    A.bits[EVB_SEOT * A.bit_words + j] &= ~m_all;
  }
  if (m_open) {
    A.bits[EVB_START * A.bit_words + j] &= ~m_open;
    A.bits[EVB_SEPS * A.bit_words + j] &= ~m_open;
  }
}

// ------------------------------------------------------------------ compact

#define CQ_CAP 512u  // ring capacity in queued positions (power of two; one light step adds at most 256)
#define CT_CAP 512u  // rows of a fast tile staged in LDS (a tile of 2048 positions holds ~300-450 tokens; more go out directly)

// Range of one segment of a long document: closing kinds in (p0, p1], opening kinds in
// [p0, p1) -- or [p0, p1] for the document's last segment.  false: nothing to do.
struct SegRange { uint32_t d, p0, p1; bool first, last; };
__device__ __forceinline__ bool seg_range(const DtkCompactArgs &A, uint32_t s, uint32_t len_of_d, SegRange &r) {
  const uint32_t La = A.seg_lane0[s], Lb = La + A.seg_nl[s];
  const uint32_t L0 = A.chunk_off[r.d], L1 = A.chunk_off[r.d + 1];
  r.first = La == L0;
  r.last = Lb >= L1;
  r.p0 = r.first ? 0u : A.lane_start[La].p;
  if (r.p0 == 0xFFFFFFFFu) return false;  // the lane chain reached the end of the document before
  r.p1 = r.last ? len_of_d : A.lane_start[Lb].p;
  if (r.p1 == 0xFFFFFFFFu) { r.p1 = len_of_d; r.last = true; }
  return true;
}

// 32 bits of a bitmap starting at bit `bit` (the arrays are padded by two words)
__device__ __forceinline__ uint32_t bits32(const uint32_t *__restrict__ b, uint32_t bit) {
  const uint32_t w = bit >> 5, sh = bit & 31u;
  const uint32_t lo = b[w], hi = b[w + 1];
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);  // (one 64-bit shift: no test of sh between the loads and their use)
}
__device__ __forceinline__ uint32_t lowmask(uint32_t n) { return n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u; }

// One wave per document (or per segment of a long one).
// Light: tiles of 2048 cursor positions, 32 per lane -- the lane's word of every bitmap (token ends, token starts,
// epsilon SentenceEnds, EOT calls; rune starts from k_symbolize's bitmap), rune counts by a wave scan.  A tile is
// queued in eight steps of 256 positions (4 per lane, the words fetched from their lanes by shuffles): every
// position that carries a call goes into a ring in LDS with its rune index and, for a token end, where the token
// started (the highest START bit below it: in the same word or the one before; longer tokens search backwards).
// Heavy: whenever 64 positions are queued (or at the end), lane i takes the i-th and everything NewTokenWriter
// tracks (token_writer.go:38-42: posC, pos, sentB, sent) is recovered with ballots, popcounts of the lanes
// below and a handful of shuffles; wave-uniform carries link the rounds.  Order of the calls at one position =
// bit order of the queued flags.
//
// Two kernels share the text below.  FULL = false: the documents without an EOT call -- tokens and epsilon SentenceEnds
// only, every tile takes the fast path; none of the queue, of the heavy rounds or of their carries is compiled in
// (about half the registers, a third of the code).  FULL = true: the documents with one.  Which is which follows from
// what the walk counted: a document has an EOT call iff it has more than one TextEnd or its only TextEnd is not the
// tail's.  The first kernel tells the host that the second is needed (any_eot); a batch object whose last run
// needed it launches it right away (dtk_host.cpp).
#ifdef DTK_PROBE
// cycles per wave of k_compact_plain: prologue, tile loads + rune scan, counts + latch, token loop, sentence loop +
// carries, tail; [6] waves, [7] tiles
__device__ unsigned long long g_cphase[8];
extern "C" int dtk_cphase_read(unsigned long long *out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cphase), sizeof(g_cphase));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_cphase), z, sizeof(z)); }
  return 0;
}
#define CPROBE(i) do { if (!FULL) { const unsigned long long n_ = clock64(); pr_c[i] += n_ - pr_last; pr_last = n_; } } while (0)
#else
#define CPROBE(i) do { } while (0)
#endif

template <bool FULL>
__device__ __forceinline__ void compact_unit(const DtkCompactArgs &A, uint32_t small_max, const uint32_t *big_docs) {
  __shared__ uint32_t qpos[FULL ? CQ_CAP : 1u], qrn[FULL ? CQ_CAP : 1u], qst[FULL ? CQ_CAP : 1u], qsr[FULL ? CQ_CAP : 1u];
  __shared__ uint8_t qfl[FULL ? CQ_CAP : 1u];
  __shared__ uint2 s_tok[CT_CAP];      // a fast tile's rows on their way out (see the token loop)
  __shared__ uint16_t s_sb[CT_CAP];
#ifdef DTK_PROBE
  unsigned long long pr_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pr_last = clock64();
#endif
  const bool seg_mode = A.seg_doc != nullptr;
  // one wave per segment, per document, or per document of the list of those that k_compact_small leaves to me
  // (a scalar: what is indexed with it below then comes through the scalar cache, in one batch of requests)
  const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(seg_mode ? A.seg_doc[blockIdx.x] : (big_docs ? big_docs[blockIdx.x] : blockIdx.x)));
  // Everything the wave needs to know about its document, requested before any of it is looked at: the tests
  // below used to stand between the loads, seven memory round trips in a row before the first tile.
  const uint32_t skip_now = A.skip_if ? *A.skip_if : 0u;
  const uint64_t off = A.doc_off[d], off_end = A.doc_off[d + 1];
  const uint32_t st_d = A.status[d], tail_d = A.doc_tail[d];
  const uint64_t tot0 = A.totals[0], tot1 = A.totals[1], tot2 = A.totals[2];
  const uint64_t tok_base = A.tok_off[d], sent_base = A.sent_off[d], text_base = A.text_off[d];
  const uint64_t tok_lim = A.tok_off[d + 1], sent_lim = A.sent_off[d + 1], text_lim = A.text_off[d + 1];
  if (skip_now != 0u) return;  // documents are still to be repaired: the host runs this pass afterwards
  const uint32_t len = (uint32_t)(off_end - off);
  if (len <= small_max && small_max != 0u) return;  // (segment mode: a small document's one segment)
  const uint32_t gb = (uint32_t)DTK_EV_BIT(off, d);
  const uint32_t *__restrict__ bE = A.bits + (size_t)EVB_END * A.bit_words;
  const uint32_t *__restrict__ bS = A.bits + (size_t)EVB_START * A.bit_words;
  const uint32_t *__restrict__ bP = A.bits + (size_t)EVB_SEPS * A.bit_words;
  const uint32_t *__restrict__ bT = A.bits + (size_t)EVB_TEOT * A.bit_words;
  const uint32_t *__restrict__ bU = A.bits + (size_t)EVB_SEOT * A.bit_words;
  const uint8_t *__restrict__ txt = A.text + off;
  const bool nl_rule = (A.flags & 16u) != 0;  // NEWLINE_AFTER_EOT
  const bool is_matrix = A.kind == DTK_KIND_MATRIX;
  const uint32_t lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  // a document whose calls are not in position order: its rows are written by the exact pass (k_exact_doc)
  if (st_d & ST_IRREGULAR) {
    if (lane == 0) atomicOr(A.any_irregular, 1u);
    return;
  }

  // rows were sized by the walk's counts + scan; skip everything if the output arrays
  // are too small (the host grows them and re-launches this pass)
  if (tot0 > A.tok_cap || tot1 > A.sent_cap || tot2 > A.text_cap) return;
  {
    const bool has_eot = text_lim - text_base != 1ull || !(tail_d & DTK_TAIL_E);
    if (has_eot != FULL) {
      // (looked at before it is written: a batch in which every document has an EOT would otherwise queue one
      //  atomic per document at this address)
      if (!FULL && lane == 0 && __hip_atomic_load(A.any_eot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        atomicOr(A.any_eot, 1u);
      return;
    }
  }

  // wave-uniform carries
  uint32_t cR = 0;           // runes started before the tile
  uint32_t cTE = 0;          // token ends before the heavy round
  uint32_t cNE = 0, cNSev = 0;  // TextEnd / SentenceEnd calls before the round
  uint32_t cNSent = 0;       // sentence ints pushed before the round
  uint32_t cSEatEnd = 0, cEatEnd = 0;  // calls seen when the last token ended
  uint32_t cLastEndR = 0, cLastEndByte = 0;
  int32_t cLastRend = 0;
  uint32_t cBase = 0;        // rune index that maps to offset 0 in the current text
  uint32_t cLastER = 0, cLastEByte = 0, cTokAtLastE = 0;
  bool cHaveE = false;
  uint32_t status = 0;
  uint32_t qhead = 0, qn = 0;  // ring: first entry, entries queued

  // the positions of this wave: the whole document, or one segment of a long one
  SegRange sr{d, 0u, len, true, true};
  if (seg_mode) {
    if (!seg_range(A, blockIdx.x, len, sr)) return;
    if (A.chunk_off[d + 1] - A.chunk_off[d] > DTK_SEG_LANES && A.doc_seq[d]) {  // sequential after all
      if (!sr.first) return;
      sr.p1 = len; sr.last = true;
    }
    if (!sr.first) {
      // Everything the sequential pass would carry into position p0 (a sync point of the walk: the
      // window was rewound there) follows from the totals of the lanes before it and from the last
      // EOT TextEnd before it (k_seg_scan).  (The matrix rewinds at an EOT, matrix.go:601; a double-array
      // document with an EOT does not get here.)
      const DtkSegIn in = A.seg_in[blockIdx.x];
      cR = in.runes; cTE = in.tok; cNE = in.text; cNSev = in.sev; cNSent = in.sent;
      cHaveE = in.e_pos != 0xFFFFFFFFu;
      if (cHaveE) {
        cLastER = in.e_runes; cTokAtLastE = in.e_tok;
        cLastEByte = (nl_rule && in.e_pos < len) ? txt[in.e_pos] : 0u;
      }
      // the text that is open at p0 has a token already: its first token fixed the rune base
      const uint32_t kf = cHaveE ? cTokAtLastE : 0u;
      if (cTE > kf)
        cBase = kf == 0u ? (cHaveE ? cLastER : 0u)  // it was the document's first token
                         : cLastER + ((nl_rule && cLastEByte == '\n') ? 1u : 0u);
      if (bits32(bE, gb + sr.p0) & 1u) {  // the rewind at p0 was a token flush: that token is "the last one"
        cLastEndR = cR;
        cLastEndByte = (nl_rule && sr.p0 < len) ? txt[sr.p0] : 0u;
        cSEatEnd = cNSev + cNE; cEatEnd = cNE;
        cLastRend = (int32_t)(cR - cBase);
      }  // else an EOT TextEnd: whatever ended before it is behind a call, the zeros above do
    }
  }

  const uint32_t n_pos = sr.p1 + 1u;  // cursor positions p0..p1
  CPROBE(0);

  // ---- heavy: the queued positions, 64 at a time (all of them if `drain`)
  auto heavy_rounds = [&](bool drain) {
      while (qn >= WAVE || (drain && qn > 0)) {
      const uint32_t take = qn < WAVE ? qn : WAVE;
      uint32_t P = 0, f = 0, R = 0, tb = 0, startP = 0, startR = 0;
      if (lane < take) {
        const uint32_t at = (qhead + lane) & (CQ_CAP - 1u);
        P = qpos[at]; f = qfl[at]; R = qrn[at];
        if (f & EV_TOK_END) { startP = qst[at]; startR = qsr[at]; }
        // byte behind a token / an EOT: only the NEWLINE_AFTER_EOT rule looks at it (token_writer.go:66-68)
        if (nl_rule && (f & (EV_TOK_END | EV_E_EOT)) && P < len) tb = txt[P];
      }
      qhead += take;
      qn -= take;
      const unsigned long long mEND = __ballot(f & EV_TOK_END);
      const unsigned long long mEEOT = __ballot(f & EV_E_EOT);
      const unsigned long long mEEOF = __ballot(f & EV_E_EOF);
      const unsigned long long mS1 = __ballot(f & EV_S_EOT);
      const unsigned long long mS2 = __ballot(f & EV_S_EPS);
      const unsigned long long mS4 = __ballot(f & EV_S_EOF);

      // Order of the calls at one position (bit order): S_EOT, E_EOT, TOK_END, S_EPS, S_EOF, E_EOF.
      const uint32_t te = cTE + popc(mEND & lt);   // tokens ended at lower positions
      const bool isEnd = (f & EV_TOK_END) != 0;
      const bool hasEEOT = (f & EV_E_EOT) != 0;
      const uint32_t s1 = (f & EV_S_EOT) ? 1u : 0u;
      const uint32_t sLate = popc((unsigned long long)(f & (EV_S_EPS | EV_S_EOF)));
      // TextEnd / SentenceEnd calls fired before this lane's TOK_END (own EOT pair included)
      const uint32_t eBeforeEnd = cNE + popc(mEEOT & lt) + popc(mEEOF & lt) + (hasEEOT ? 1u : 0u);
      const uint32_t sBeforeEnd = cNSev + popc(mS1 & lt) + popc(mS2 & lt) + popc(mS4 & lt) + s1;
      const uint32_t tokLate = te + (isEnd ? 1u : 0u);  // tokens ended before this lane's late calls

      // previous token end (strictly below this lane)
      const unsigned long long mPrevEnd = mEND & lt;
      const bool havePrev = mPrevEnd != 0ull;
      const int jp = havePrev ? highest(mPrevEnd) : 0;
      const uint32_t seAtPrev_t = __shfl(eBeforeEnd + sBeforeEnd, jp);
      const uint32_t eAtPrev_t = __shfl(eBeforeEnd, jp);
      const uint32_t RatPrev_t = __shfl(R, jp);
      const uint32_t byteAtPrev_t = nl_rule ? __shfl(tb, jp) : 0u;
      const uint32_t seAtPrev = havePrev ? seAtPrev_t : cSEatEnd;
      const uint32_t eAtPrev = havePrev ? eAtPrev_t : cEatEnd;
      const uint32_t RatPrev = havePrev ? RatPrev_t : cLastEndR;
      const uint32_t byteAtPrev = havePrev ? byteAtPrev_t : cLastEndByte;

      const uint32_t k = te;  // index of the token that ends here
      const bool text_first = isEnd && (k == 0 || eBeforeEnd > eAtPrev);
      const bool sent_first = isEnd && (k == 0 || (eBeforeEnd + sBeforeEnd) > seAtPrev);

      // last E_EOT strictly below this lane
      const unsigned long long mPrevE = mEEOT & lt;
      const bool haveE = mPrevE != 0ull;
      const int je = haveE ? highest(mPrevE) : 0;
      uint32_t RatE_t = 0, byteAtE_t = 0, tokAtE_t = 0;
      if (mEEOT) {  // wave-uniform: most rounds hold no EOT
        RatE_t = __shfl(R, je);
        byteAtE_t = __shfl(tb, je);
        tokAtE_t = __shfl(te, je);  // an E_EOT precedes a token end at its own position
      }
      const uint32_t RatE = haveE ? RatE_t : cLastER;
      const uint32_t byteAtE = haveE ? byteAtE_t : cLastEByte;
      const uint32_t tokAtPrevE = haveE ? tokAtE_t : cTokAtLastE;
      const bool anyE = haveE || cHaveE;

      // rune index that counts as offset 0 for the text this token opens
      // (token_writer.go:66-81: posC restarts at 0; the offset handed to Token is
      // counted from the start of the window, which the matrix rewinds to the rune
      // after EOT (matrix.go:601) and the double array only at token flushes).
      uint32_t base_mine;
      if (k == 0) {
        base_mine = (is_matrix && anyE) ? RatE : 0u;
      } else if (is_matrix) {
        base_mine = RatE + ((nl_rule && byteAtE == '\n') ? 1u : 0u);
      } else {
        base_mine = RatPrev + ((nl_rule && byteAtPrev == '\n') ? 1u : 0u);
      }
      const unsigned long long mTF = __ballot(text_first);
      const unsigned long long mPrevTF = mTF & lt;
      const int jt = mPrevTF ? highest(mPrevTF) : 0;
      const uint32_t baseFrom_t = __shfl(base_mine, jt);
      const uint32_t tbase = text_first ? base_mine : (mPrevTF ? baseFrom_t : cBase);
      const int32_t rend = (int32_t)(R - tbase);
      const int32_t rstart = rend - (int32_t)(R - startR);
      // NEWLINE_AFTER_EOT is modelled as one shift per text, but token_writer.go:66-68 fires whenever posC == 0 and the
      // buffer starts with a newline: a token that ends at offset 0 (it began at -1: a tokenizer that makes a token
      // of the newline itself) may make it fire again.  Left to the exact pass.
      if (nl_rule && isEnd && rend == 0) status |= ST_INTERNAL;

      // end offset of the last token below this lane / at or below it
      const int32_t rendPrev_t = __shfl(rend, jp);
      const int32_t rendBelow = havePrev ? rendPrev_t : cLastRend;
      const int32_t rendLate = isEnd ? rend : rendBelow;

      // SentenceEnd / TextEnd with no token in the current text (reference panics)
      const bool emptyEarly = te == tokAtPrevE;                       // for S_EOT, E_EOT
      const bool emptyLate = hasEEOT ? !isEnd : (tokLate == tokAtPrevE);  // for S_EPS.., E_EOF
      const uint32_t s1_valid = emptyEarly ? 0u : s1;
      const uint32_t sLate_valid = emptyLate ? 0u : sLate;
      if ((s1 && emptyEarly) || (hasEEOT && emptyEarly) || (sLate && emptyLate) ||
          ((f & EV_E_EOF) && emptyLate))
        status |= ST_EMPTY_TEXT;

      const uint32_t c = s1_valid + (sent_first ? 1u : 0u) + sLate_valid;
      uint32_t cTotal;
      const uint32_t excl = wave_excl_scan(c, cTotal);

      {
        if (isEnd && tok_base + k < tok_lim) {
          A.tok_bstart[tok_base + k] = startP;
          A.tok_bend[tok_base + k] = P;
          A.tok_rstart[tok_base + k] = rstart;
          A.tok_rend[tok_base + k] = rend;
          if (A.tok_sbefore) A.tok_sbefore[tok_base + k] = sBeforeEnd;  // SentenceEnd calls before this Token call
        }
        uint64_t si = sent_base + cNSent + excl;
        if (si + c <= sent_lim) {
          if (s1_valid) A.sent[si++] = rendBelow;         // token_writer.go:108
          if (sent_first) A.sent[si++] = rstart;          // token_writer.go:76-79
          for (uint32_t q = 0; q < sLate_valid; q++) A.sent[si++] = rendLate;
        } else if (c) {
          status |= ST_INTERNAL;
        }
        if (hasEEOT) {
          const uint64_t ti = text_base + eBeforeEnd - 1u;
          if (ti < text_lim) {
            A.text_tok_end[ti] = te;
            A.text_sent_end[ti] = cNSent + excl + s1_valid;
            if (A.text_s_end) A.text_s_end[ti] = sBeforeEnd;  // SentenceEnd calls before this TextEnd call
          } else status |= ST_INTERNAL;
        }
        if (f & EV_E_EOF) {
          const uint64_t ti = text_base + eBeforeEnd;
          if (ti < text_lim) {
            A.text_tok_end[ti] = tokLate;
            A.text_sent_end[ti] = cNSent + excl + c;
            if (A.text_s_end) A.text_s_end[ti] = sBeforeEnd + sLate;
          } else status |= ST_INTERNAL;
        }
        if (isEnd && tok_base + k >= tok_lim) status |= ST_INTERNAL;
      }

      // carries for the next round
      if (mEND) {
        const int jl = highest(mEND);
        cSEatEnd = __shfl(eBeforeEnd + sBeforeEnd, jl);
        cEatEnd = __shfl(eBeforeEnd, jl);
        cLastEndR = __shfl(R, jl);
        cLastEndByte = nl_rule ? __shfl(tb, jl) : 0u;
        cLastRend = __shfl(rend, jl);
        cBase = __shfl(tbase, jl);
      }
      if (mEEOT) {
        const int jl = highest(mEEOT);
        cLastER = __shfl(R, jl);
        cLastEByte = __shfl(tb, jl);
        cTokAtLastE = __shfl(te, jl);
        cHaveE = true;
      }
      cTE += popc(mEND);
      cNE += popc(mEEOT) + popc(mEEOF);
      cNSev += popc(mS1) + popc(mS2) + popc(mS4);
      cNSent += cTotal;
      }  // heavy rounds
  };

  // what the tile before left for the first lane: the START and rune-start words of the 32 positions before
  // the tile, and the rune index at their first position (a segment starts at a rewind: no token spans it)
  uint32_t pS_in = 0, pR_in = 0, pRb_in = cR;
  for (uint32_t T0 = sr.p0; T0 < n_pos; T0 += 32u * WAVE) {
    // ---- the lane's words: positions q0 .. q0 + 31
    const uint32_t q0 = T0 + 32u * lane;
    uint32_t wE = 0, wS = 0, wP = 0, wT = 0, wU = 0, wR = 0;
    {
      // twelve loads in one go: a lane whose 32 positions lie behind the document reads the tile's first word
      // instead and masks everything (a test around the loads makes them wait for each other)
      const bool in = q0 < n_pos;
      const uint32_t qc = in ? q0 : T0;
      const uint32_t valid = in ? lowmask(n_pos - q0) : 0u;
      const uint32_t validR = q0 < len ? lowmask(len - q0) : 0u;
      const uint32_t qr = q0 < len ? q0 : (T0 < len ? T0 : 0u);
      const uint32_t xE = bits32(bE, gb + qc), xS = bits32(bS, gb + qc), xP = bits32(bP, gb + qc);
      const uint32_t xT = bits32(bT, gb + qc), xU = bits32(bU, gb + qc);
      const uint32_t xR = bits32(A.rs_bits, (uint32_t)off + qr);  // rune starts: bit = input byte
      wE = xE & valid; wS = xS & valid; wP = xP & valid; wT = xT & valid; wU = xU & valid; wR = xR & validR;
    }
    if (q0 < n_pos) {
      if (seg_mode) {  // closing kinds in (p0, p1], opening kinds in [p0, p1) or, at the end, [p0, p1]
        if (q0 == sr.p0) { wE &= ~1u; wT &= ~1u; wU &= ~1u; }
        if (!sr.last && sr.p1 >= q0 && sr.p1 - q0 < 32u) { wS &= ~(1u << (sr.p1 - q0)); wP &= ~(1u << (sr.p1 - q0)); }
      }
    }
    uint32_t tileR;
    const uint32_t rB = cR + wave_excl_scan((uint32_t)__popc(wR), tileR);  // rune index at q0
    // the words of the 32 positions before mine
    uint32_t pS = __shfl_up(wS, 1), pR = __shfl_up(wR, 1), pRb = __shfl_up(rB, 1);
    if (lane == 0) { pS = pS_in; pR = pR_in; pRb = pRb_in; }
    const bool any_eot = __ballot((wT | wU) != 0u) != 0ull;  // wave-uniform: most tiles hold no EOT
    if (!FULL && any_eot) status |= ST_INTERNAL;  // (the counts said there is none: the exact pass decides)

    if (!FULL || !any_eot) {
      // ---- fast: no EOT call in the tile, so the only calls are Token (END) and the epsilon SentenceEnd (SEPS) and
      //      every lane can work through its own 32 positions: no queue.  What crosses lanes comes from three wave
      //      scans (tokens, SentenceEnd calls, sentence ints) and a carry chain over two ballots: "is a sentence
      //      start pending" is a latch -- set by a SentenceEnd, reset by a token end -- whose state before every
      //      position is the carry vector of  a + b  with generate = SEPS and kill = END & ~SEPS.
      const uint32_t t0 = cHaveE ? cTokAtLastE : 0u;  // Token calls before the current text
      uint32_t base = cBase;
      if (cTE == t0) {  // the text has no token yet: its first one (in this tile or later) fixes the rune base
        base = cTE == 0u ? ((is_matrix && cHaveE) ? cLastER : 0u)
                         : (is_matrix ? cLastER + ((nl_rule && cLastEByte == '\n') ? 1u : 0u)
                                      : cLastEndR + ((nl_rule && cLastEndByte == '\n') ? 1u : 0u));
      }
      CPROBE(1);
      const uint32_t nTok = (uint32_t)__popc(wE), nP = (uint32_t)__popc(wP);
      uint32_t tot2;
      const uint32_t ex2 = wave_excl_scan(nTok | (nP << 16), tot2);
      const uint32_t tokB = ex2 & 0xFFFFu, pB = ex2 >> 16;  // Token / SentenceEnd calls of the lanes below
      // the latch
      const uint32_t la = ~(wE & ~wP), lb = wP;
      const bool genW = (((uint64_t)la + lb) >> 32) != 0ull, propW = (wE | wP) == 0u;
      const unsigned long long GG = __ballot(genW), PP = __ballot(propW);
      const bool pend_in = cTE == 0u || (cNE + cNSev) > cSEatEnd;
      const unsigned long long cA = GG | PP, carries = (cA + GG + (pend_in ? 1ull : 0ull)) ^ cA ^ GG;
      const uint32_t cin = (uint32_t)(carries >> lane) & 1u;
      const uint32_t sfm = wE & ((uint32_t)((uint64_t)la + lb + cin) ^ la ^ lb);  // tokens that start a sentence
      // a SentenceEnd counts only if its text has a token (token_writer.go:108 panics otherwise)
      uint32_t vP = wP;
      if (cTE + tokB <= t0) vP = wE ? (wP & ~lowmask((uint32_t)__ffs((int)wE) - 1u)) : 0u;
      if (vP != wP) status |= ST_EMPTY_TEXT;
      uint32_t totS;
      const uint32_t sentB4 = wave_excl_scan((uint32_t)__popc(sfm) + (uint32_t)__popc(vP), totS);
      // rune offset of the end of the last token below my word (for SentenceEnds before my first token)
      const unsigned long long mTokLanes = __ballot(nTok != 0u);
      int32_t myLastRend = 0;
      uint32_t myLastR = 0, myLastBit = 0;
      if (nTok) {
        myLastBit = 31u - (uint32_t)__clz((int)wE);
        myLastR = rB + (uint32_t)__popc(wR & lowmask(myLastBit));
        myLastRend = (int32_t)(myLastR - base);
      }
      const unsigned long long below = mTokLanes & lt;
      const int32_t rendBelow_t = __shfl(myLastRend, below ? highest(below) : 0);
      const int32_t rendBelowW = below ? rendBelow_t : cLastRend;
      CPROBE(2);
      // tokens
      uint32_t me = wE, j = 0;
      while (me) {
        const uint32_t b = (uint32_t)__ffs((int)me) - 1u;
        me &= me - 1u;
        const uint32_t P = q0 + b, R = rB + (uint32_t)__popc(wR & lowmask(b));
        uint32_t sp, sR;
        bool far = false;
        const uint32_t m = wS & lowmask(b);
        if (m) {
          const uint32_t sb = 31u - (uint32_t)__clz((int)m);
          sp = q0 + sb; sR = rB + (uint32_t)__popc(wR & lowmask(sb));
        } else if (pS) {
          const uint32_t sb = 31u - (uint32_t)__clz((int)pS);
          sp = q0 - 32u + sb; sR = pRb + (uint32_t)__popc(pR & lowmask(sb));
        } else {  // a token of more than 32 bytes: search backwards (rare)
          uint32_t q = q0 >= sr.p0 + 32u ? q0 - 32u : sr.p0, w = 0;
          while (!w && q > sr.p0) {
            const uint32_t n = q - sr.p0 < 32u ? q - sr.p0 : 32u;
            q -= n;
            w = bits32(bS, gb + q) & lowmask(n);
          }
          sp = w ? q + 31u - (uint32_t)__clz((int)w) : sr.p0;
          sR = R;
          for (uint32_t z = sp; z < P; z += 32u) sR -= (uint32_t)__popc(bits32(A.rs_bits, (uint32_t)off + z) & lowmask(P - z));
          far = true;
        }
        // The rows go through LDS (four 16-bit fields relative to the tile, one 8-byte write) and leave as runs of
        // consecutive rows below: written from here, lane by lane, every 4-byte store of the wave lands in another
        // cache line -- 64 address cycles per store instruction, which is what this kernel's time was made of.
        if (nl_rule && R == base) status |= ST_INTERNAL;  // (a token that ends at offset 0: see the heavy rounds)
        const uint32_t li = tokB + j;  // row within the tile
        const uint64_t k = tok_base + cTE + li;
        const uint32_t sbef = pB + (uint32_t)__popc(wP & lowmask(b));
        if (li < CT_CAP) {
          const bool direct = far || k >= tok_lim;
          s_tok[li] = direct ? make_uint2(0xFFFFu, 0u)
                             : make_uint2((sp - T0 + 64u) | ((P - T0 + 64u) << 16), (sR - cR + 64u) | ((R - cR + 64u) << 16));
          if (A.tok_sbefore) s_sb[li] = (uint16_t)sbef;
        }
        if (k >= tok_lim) status |= ST_INTERNAL;
        else if (far || li >= CT_CAP) {
          A.tok_bstart[k] = sp; A.tok_bend[k] = P;
          A.tok_rstart[k] = (int32_t)(sR - base); A.tok_rend[k] = (int32_t)(R - base);
          if (A.tok_sbefore) A.tok_sbefore[k] = cNSev + sbef;
        }
        if (sfm & (1u << b)) {  // token_writer.go:76-79
          const uint64_t si = sent_base + cNSent + sentB4 + (uint32_t)__popc(sfm & lowmask(b)) + (uint32_t)__popc(vP & lowmask(b));
          if (si < sent_lim) A.sent[si] = (int32_t)(sR - base); else status |= ST_INTERNAL;
        }
        j++;
      }
      CPROBE(3);
      // SentenceEnds: the end offset of the last token at or below their position (token_writer.go:108)
      uint32_t mp = vP;
      while (mp) {
        const uint32_t b = (uint32_t)__ffs((int)mp) - 1u;
        mp &= mp - 1u;
        const uint32_t e = wE & lowmask(b + 1u);
        int32_t v = rendBelowW;
        if (e) v = (int32_t)(rB + (uint32_t)__popc(wR & lowmask(31u - (uint32_t)__clz((int)e))) - base);
        const uint64_t si = sent_base + cNSent + sentB4 + (uint32_t)__popc(sfm & lowmask(b + 1u)) + (uint32_t)__popc(vP & lowmask(b));
        if (si < sent_lim) A.sent[si] = v; else status |= ST_INTERNAL;
      }
      // the tile's rows: lane i writes rows i, i + 64, ... -- consecutive addresses across the wave
      {
        __syncthreads();
        const uint32_t tn = (tot2 & 0xFFFFu) < CT_CAP ? (tot2 & 0xFFFFu) : CT_CAP;
        const uint32_t pb = T0 - 64u, rb = cR - base - 64u;
        for (uint32_t i = lane; i < tn; i += WAVE) {
          const uint2 v = s_tok[i];
          const uint64_t k = tok_base + cTE + i;
          if ((v.x & 0xFFFFu) != 0xFFFFu) {
            A.tok_bstart[k] = (v.x & 0xFFFFu) + pb; A.tok_bend[k] = (v.x >> 16) + pb;
            A.tok_rstart[k] = (int32_t)((v.y & 0xFFFFu) + rb); A.tok_rend[k] = (int32_t)((v.y >> 16) + rb);
            if (A.tok_sbefore) A.tok_sbefore[k] = cNSev + s_sb[i];
          }
        }
        __syncthreads();
      }
      // carries
      if (mTokLanes) {
        const int jl = highest(mTokLanes);
        cLastEndR = __shfl(myLastR, jl);
        cLastRend = __shfl(myLastRend, jl);
        cSEatEnd = cNE + cNSev + __shfl(pB + (uint32_t)__popc(wP & lowmask(myLastBit)), jl);
        cEatEnd = cNE;
        const uint32_t lastP = __shfl(q0 + myLastBit, jl);
        cLastEndByte = (nl_rule && lastP < len) ? txt[lastP] : 0u;
        cBase = base;
      }
      cTE += tot2 & 0xFFFFu;
      cNSev += tot2 >> 16;
      cNSent += totS;
    } else if constexpr (FULL) {
      for (uint32_t step = 0; step < 8u; step++) {
        const uint32_t S0 = T0 + 256u * step;  // first position of the step (wave-uniform)
        if (S0 >= n_pos) break;
        // ---- light: my 4 positions P0 .. P0 + 3 live in the word of lane `src`, bits ns .. ns + 3
        const uint32_t src = 8u * step + (lane >> 3), ns = (lane & 7u) * 4u;
        const uint32_t P0 = S0 + 4u * lane;
        const uint32_t xE = __shfl(wE, src), xP = __shfl(wP, src), xS = __shfl(wS, src), xR = __shfl(wR, src);
        const uint32_t xRb = __shfl(rB, src), yS = __shfl(pS, src), yR = __shfl(pR, src), yRb = __shfl(pRb, src);
        const uint32_t nE = (xE >> ns) & 15u, nP = (xP >> ns) & 15u;
        const uint32_t nT = (__shfl(wT, src) >> ns) & 15u, nU = (__shfl(wU, src) >> ns) & 15u;
        const uint32_t evn = nE | nP | nT | nU;  // positions of mine that carry a call
        uint32_t tot;
        const uint32_t ex = wave_excl_scan((uint32_t)__popc(evn), tot);
        uint32_t slot = qhead + qn + ex;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          if (evn & (1u << j)) {
            const uint32_t b = ns + j, at = slot & (CQ_CAP - 1u);
            const uint32_t R = xRb + (uint32_t)__popc(xR & lowmask(b));
            qpos[at] = P0 + j;
            qrn[at] = R;
            qfl[at] = (uint8_t)(((nU >> j) & 1u) * EV_S_EOT | ((nT >> j) & 1u) * EV_E_EOT | ((nE >> j) & 1u) * EV_TOK_END |
                                ((nP >> j) & 1u) * EV_S_EPS);
            if (nE & (1u << j)) {
              // the token's first byte: the highest START bit below this position
              uint32_t sp, sr_;
              const uint32_t m = xS & lowmask(b);
              if (m) {
                const uint32_t sb = 31u - (uint32_t)__clz((int)m);
                sp = P0 + j - (b - sb);
                sr_ = xRb + (uint32_t)__popc(xR & lowmask(sb));
              } else if (yS) {
                const uint32_t sb = 31u - (uint32_t)__clz((int)yS);
                sp = P0 + j - b - 32u + sb;
                sr_ = yRb + (uint32_t)__popc(yR & lowmask(sb));
              } else {  // a token of more than 32 bytes: search backwards (rare)
                const uint32_t ws = P0 + j - b;  // first position of my word; the 32 before it hold no START
                uint32_t q = ws >= sr.p0 + 32u ? ws - 32u : sr.p0;
                uint32_t w = 0;
                while (!w && q > sr.p0) {
                  const uint32_t n = q - sr.p0 < 32u ? q - sr.p0 : 32u;
                  q -= n;
                  w = bits32(bS, gb + q) & lowmask(n);
                }
                sp = w ? q + 31u - (uint32_t)__clz((int)w) : sr.p0;
                sr_ = R;
                for (uint32_t z = sp; z < P0 + j; z += 32u)  // rune starts in [sp, position)
                  sr_ -= (uint32_t)__popc(bits32(A.rs_bits, (uint32_t)off + z) & lowmask(P0 + j - z));
              }
              qst[at] = sp; qsr[at] = sr_;
            }
            slot++;
          }
        }
        qn += tot;
        __syncthreads();
        heavy_rounds(false);
        __syncthreads();
      }
      heavy_rounds(true);  // nothing stays queued across a tile: the next one may take the fast path
      __syncthreads();
    }
    CPROBE(4);
#ifdef DTK_PROBE
    pr_c[7]++;
#endif
    // hand the last lane's words to the next tile's first lane
    pS_in = __shfl(wS, WAVE - 1); pR_in = __shfl(wR, WAVE - 1); pRb_in = __shfl(rB, WAVE - 1);
    cR += tileR;
  }
  if (sr.last) {
    // the final SentenceEnd / TextEnd of the document (matrix.go:683-691), behind everything: from the carries
    const uint32_t tw = tail_d;
    const bool empty = cTE == (cHaveE ? cTokAtLastE : 0u);  // the text has no token (token_writer.go:108,135 panic)
    if ((tw & 3u) && empty) status |= ST_EMPTY_TEXT;
    if (tw & DTK_TAIL_S) {
      if (!empty) {
        if (sent_base + cNSent < sent_lim) { if (lane == 0) A.sent[sent_base + cNSent] = cLastRend; } else status |= ST_INTERNAL;
        cNSent++;
      }
      cNSev++;
    }
    if (tw & DTK_TAIL_E) {
      if (text_base + cNE < text_lim) {
        if (lane == 0) {
          A.text_tok_end[text_base + cNE] = cTE;
          A.text_sent_end[text_base + cNE] = cNSent;
          if (A.text_s_end) A.text_s_end[text_base + cNE] = cNSev;
        }
      } else status |= ST_INTERNAL;
      cNE++;
    }
  }

  // the walk's counts sized the rows: they must agree with what was written here
  uint32_t sred = status;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sred |= __shfl_down(sred, o);
  sred = __shfl(sred, 0);
  if (lane == 0) {
    // The walk's counts sized the rows.  If the bitmaps do not add up to them, two calls fell on one bit: the
    // double array fired one EOT twice from two different lanes, say.  The exact pass decides: it walks the
    // document in call order and reports ST_INTERNAL itself if its calls do not fill the rows either.
    if (sr.last && (tok_base + cTE != tok_lim || sent_base + cNSent != sent_lim || text_base + cNE != text_lim))
      sred |= ST_INTERNAL;
    sred &= ST_INTERNAL;  // everything else was reported by the walk already
    if (sred) {
      atomicOr(&A.status[d], ST_IRREGULAR);
      atomicOr(A.any_irregular, 1u);
    }
    if (sr.last && A.doc_ns) A.doc_ns[d] = cNSev;  // SentenceEnd calls of this document (for rendering)
  }
#ifdef DTK_PROBE
  CPROBE(5);
  if (!FULL && lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(&g_cphase[i], pr_c[i]);
    atomicAdd(&g_cphase[6], 1ull); atomicAdd(&g_cphase[7], pr_c[7]);
  }
#endif
}

// (69 VGPRs.  Forced down to 64 for eight waves per SIMD the compiler spills four of them: slower, 22.6 -> 25.1 us per
//  16 MiB on a saturated chip)
__global__ __launch_bounds__(WAVE) void k_compact_plain(DtkCompactArgs A, uint32_t small_max, const uint32_t *big_docs) {
  compact_unit<false>(A, small_max, big_docs);
}
__global__ __launch_bounds__(WAVE) void k_compact_eot(DtkCompactArgs A, uint32_t small_max, const uint32_t *big_docs) {
  compact_unit<true>(A, small_max, big_docs);
}

// ---- small documents: one LANE per document.
// A wave per document spends most of its instructions on cross-lane bookkeeping; for a batch of many small documents
// (tens of thousands of tweets or sentences) that is two orders of magnitude more work than the documents hold.
// Here every lane is NewTokenWriter for its own document (token_writer.go:36-175; same capture semantics as the
// exact pass): it walks the set bits of its document's bitmap words in position order -- SEOT, TEOT, END, SEPS at
// one position, then the START bit, which belongs to the next token -- and writes its rows.  k_compact skips these
// documents (small_max).
__global__ __launch_bounds__(256) void k_compact_small(DtkCompactArgs A, uint32_t small_max) {
  if (A.skip_if && *A.skip_if != 0u) return;
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= A.n_docs) return;
  // (all of the document's facts requested before any is tested: the tests used to stand between the loads)
  const uint64_t off = A.doc_off[d], off_end = A.doc_off[d + 1];
  const uint32_t st_d = A.status[d];
  const uint64_t tot0 = A.totals[0], tot1 = A.totals[1], tot2 = A.totals[2];
  const uint64_t tok_base = A.tok_off[d], sent_base = A.sent_off[d], text_base = A.text_off[d];
  const uint64_t tok_lim = A.tok_off[d + 1], sent_lim = A.sent_off[d + 1], text_lim = A.text_off[d + 1];
  const uint32_t tw = A.doc_tail[d];         // matrix.go:683-691
  const uint32_t len = (uint32_t)(off_end - off);
  if (len > small_max) return;
  if (st_d & ST_IRREGULAR) { atomicOr(A.any_irregular, 1u); return; }
  if (tot0 > A.tok_cap || tot1 > A.sent_cap || tot2 > A.text_cap) return;
  const uint32_t tok_n = (uint32_t)(tok_lim - tok_base), sent_n = (uint32_t)(sent_lim - sent_base),
                 text_n = (uint32_t)(text_lim - text_base);
  const uint32_t gb = (uint32_t)DTK_EV_BIT(off, d);
  const uint32_t *__restrict__ bE = A.bits + (size_t)EVB_END * A.bit_words;
  const uint32_t *__restrict__ bS = A.bits + (size_t)EVB_START * A.bit_words;
  const uint32_t *__restrict__ bP = A.bits + (size_t)EVB_SEPS * A.bit_words;
  const uint32_t *__restrict__ bT = A.bits + (size_t)EVB_TEOT * A.bit_words;
  const uint32_t *__restrict__ bU = A.bits + (size_t)EVB_SEOT * A.bit_words;
  const uint8_t *__restrict__ txt = A.text + off;
  const bool nl_rule = (A.flags & 16u) != 0, is_matrix = A.kind == DTK_KIND_MATRIX;
  // token_writer.go:38-42, and the window: B = byte position of buffer[0], with its rune index
  int32_t posC = 0, last_rend = 0;
  bool init = true, sentB = true;
  uint32_t n_tok = 0, n_sent = 0, n_text = 0, n_sev = 0, text_tok0 = 0;
  uint32_t B = 0, RB = 0, cs = 0, Rcs = 0;  // window start; start of the token under way
  uint32_t st = 0, Rw = 0;                  // rune index at the first position of the word
  auto sentence_end = [&]() {               // token_writer.go:104-115
    n_sev++;
    if (n_tok != text_tok0) {
      if (n_sent < sent_n) A.sent[sent_base + n_sent] = last_rend; else st |= ST_INTERNAL;
      n_sent++;
    }
    sentB = true;
  };
  auto text_end = [&]() {                   // token_writer.go:131-159
    if (n_text < text_n) {
      A.text_tok_end[text_base + n_text] = n_tok; A.text_sent_end[text_base + n_text] = n_sent;
      if (A.text_s_end) A.text_s_end[text_base + n_text] = n_sev;
    } else st |= ST_INTERNAL;
    n_text++;
    sentB = true; posC = 0; text_tok0 = n_tok;
  };
  // the next 32 positions' words are requested before this word's calls are worked through
  uint32_t nE = bits32(bE, gb), nS = bits32(bS, gb), nP = bits32(bP, gb), nT = bits32(bT, gb), nU = bits32(bU, gb),
           nR = bits32(A.rs_bits, (uint32_t)off);
  for (uint32_t q0 = 0; q0 <= len; q0 += 32u) {
    const uint32_t valid = lowmask(len + 1u - q0);
    const uint32_t wE = nE & valid, wS = nS & valid, wP = nP & valid, wT = nT & valid, wU = nU & valid;
    const uint32_t wR = q0 < len ? nR & lowmask(len - q0) : 0u;
    {
      const uint32_t q1 = q0 + 32u <= len ? q0 + 32u : q0;  // (behind the document: this word again, unused)
      nE = bits32(bE, gb + q1); nS = bits32(bS, gb + q1); nP = bits32(bP, gb + q1);
      nT = bits32(bT, gb + q1); nU = bits32(bU, gb + q1);
      nR = bits32(A.rs_bits, (uint32_t)off + (q1 < len ? q1 : 0u));
    }
    // (the START bits are looked up, not walked through: a token's first byte is the highest START bit below its
    //  END bit -- in this word, or the last one of the words before: half the iterations of the divergent loop)
    uint32_t ev = wE | wP | wT;
    while (ev) {
      const uint32_t b = (uint32_t)__ffs((int)ev) - 1u, m = 1u << b;
      ev &= ev - 1u;
      const uint32_t p = q0 + b, R = Rw + (uint32_t)__popc(wR & lowmask(b));
      if (wT & m) {                          // matrix.go:593-605
        if (wU & m) sentence_end();
        text_end();
        if (is_matrix) { B = p; RB = R; }    // matrix.go:601 rewinds, datok.go:1019-1030 does not
      }
      if (wE & m) {                          // Token(offset, buf), token_writer.go:58-88
        const uint32_t ms = wS & lowmask(b);
        if (ms) {
          const uint32_t sb = 31u - (uint32_t)__clz((int)ms);
          cs = q0 + sb; Rcs = Rw + (uint32_t)__popc(wR & lowmask(sb));
        }
        if (posC == 0 && nl_rule && p > B && txt[B] == '\n' && !init) posC--;
        init = false;
        posC += (int32_t)(Rcs - RB);
        const int32_t rs = posC;
        if (sentB) {
          sentB = false;
          if (n_sent < sent_n) A.sent[sent_base + n_sent] = rs; else st |= ST_INTERNAL;
          n_sent++;
        }
        posC += (int32_t)(R - Rcs);
        last_rend = posC;
        if (n_tok < tok_n) {
          A.tok_bstart[tok_base + n_tok] = cs; A.tok_bend[tok_base + n_tok] = p;
          A.tok_rstart[tok_base + n_tok] = rs; A.tok_rend[tok_base + n_tok] = posC;
          if (A.tok_sbefore) A.tok_sbefore[tok_base + n_tok] = n_sev;
        } else st |= ST_INTERNAL;
        n_tok++;
        B = p; RB = R;
      }
      if (wP & m) sentence_end();            // matrix.go:574-575
    }
    if (wS) {  // the token that is under way at the end of this word started here
      const uint32_t sb = 31u - (uint32_t)__clz((int)wS);
      cs = q0 + sb; Rcs = Rw + (uint32_t)__popc(wR & lowmask(sb));
    }
    Rw += (uint32_t)__popc(wR);
  }
  if (tw & DTK_TAIL_S) sentence_end();
  if (tw & DTK_TAIL_E) text_end();
  if (st || n_tok != tok_n || n_sent != sent_n || n_text != text_n) {  // (see k_compact: the exact pass decides)
    atomicOr(&A.status[d], ST_IRREGULAR);
    atomicOr(A.any_irregular, 1u);
  }
  if (A.doc_ns) A.doc_ns[d] = n_sev;
}

// ---- long documents: what each segment adds (k_seg_sum), then per document an exclusive scan of
//      the segments (k_seg_scan) -> the carries k_compact starts a segment with

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// rune starts among the input bytes [g0, g1) (bit g of rs_bits = byte g), all lanes take part
__device__ __forceinline__ uint32_t runes_between(const uint32_t *__restrict__ bits, uint64_t g0, uint64_t g1) {
  uint32_t n = 0;
  if (g1 > g0) {
    const uint64_t w0 = g0 >> 5, wl = (g1 - 1) >> 5;  // first and last word touched
    for (uint64_t w = w0 + lane_id(); w <= wl; w += WAVE) {
      uint32_t x = bits[w];
      if (w == w0) x &= 0xFFFFFFFFu << (g0 & 31u);
      if (w == wl && (g1 & 31u)) x &= (1u << (g1 & 31u)) - 1u;
      n += (uint32_t)__popc(x);
    }
  }
  return wave_sum(n);
}

__global__ __launch_bounds__(WAVE) void k_seg_sum(DtkCompactArgs A) {
  const uint32_t s = blockIdx.x;
  if (s >= A.n_segs) return;
  if (A.skip_if && *A.skip_if != 0u) return;
  const uint32_t d = A.seg_doc[s];
  if (A.chunk_off[d + 1] - A.chunk_off[d] <= DTK_SEG_LANES) return;  // a single segment needs no carry
  const uint64_t off = A.doc_off[d];
  const uint32_t len = (uint32_t)(A.doc_off[d + 1] - off);
  const uint32_t lane = lane_id();
  DtkSegSum o{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
  SegRange sr{d, 0u, len, true, true};
  if (seg_range(A, s, len, sr)) {
    const uint32_t La = A.seg_lane0[s], nl = A.seg_nl[s];
    DtkLaneCount c{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    if (lane < nl) c = A.lane_cnt[La + lane];
    uint32_t tot;
    const uint32_t tok_before = wave_excl_scan(c.tok, tot);  // Token calls of the segment's earlier lanes
    o.tok = tot;
    o.sent = wave_sum(c.sent); o.text = wave_sum(c.text); o.sev = wave_sum(c.sev);
    o.runes = runes_between(A.rs_bits, off + sr.p0, off + sr.p1);
    const unsigned long long mE = __ballot(c.e_pos != 0xFFFFFFFFu);
    if (mE) {  // the last lane with an EOT TextEnd
      const int j = highest(mE);
      o.e_pos = __shfl(c.e_pos, j);
      o.e_tok = __shfl(tok_before + c.e_tok, j);
      o.e_runes = runes_between(A.rs_bits, off + sr.p0, off + o.e_pos);
    }
  }
  if (lane == 0) A.seg_sum[s] = o;
}

// one wave per document with more than one segment: exclusive scan of its segment sums
__global__ __launch_bounds__(WAVE) void k_seg_scan(DtkCompactArgs A, const uint32_t *doc_seg0) {
  const uint32_t d = blockIdx.x;
  if (d >= A.n_docs) return;
  if (A.skip_if && *A.skip_if != 0u) return;
  const uint32_t s0 = doc_seg0[d], s1 = doc_seg0[d + 1];
  if (s1 - s0 <= 1u) return;
  const uint32_t lane = lane_id();
  bool any_e = false;  // an EOT TextEnd somewhere in the document
  uint32_t bt = 0, bs = 0, bx = 0, bv = 0, br = 0;        // running totals before the current group of 64
  uint32_t ce_pos = 0xFFFFFFFFu, ce_tok = 0, ce_runes = 0;  // last EOT TextEnd so far (absolute)
  for (uint32_t g = s0; g < s1; g += WAVE) {
    const uint32_t s = g + lane;
    DtkSegSum v{0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    if (s < s1) v = A.seg_sum[s];
    uint32_t tt, ts, tx, tv, tr;
    const uint32_t et = wave_excl_scan(v.tok, tt), es = wave_excl_scan(v.sent, ts), ex = wave_excl_scan(v.text, tx);
    const uint32_t ev = wave_excl_scan(v.sev, tv), er = wave_excl_scan(v.runes, tr);
    // last EOT TextEnd before my segment: the nearest lower lane of this group that has one, else the carry
    const unsigned long long mE = __ballot(v.e_pos != 0xFFFFFFFFu);
    const unsigned long long below = mE & lanemask_lt();
    const int j = below ? highest(below) : 0;
    const uint32_t jp = __shfl(v.e_pos, j), jt = __shfl(bt + et + v.e_tok, j), jr = __shfl(br + er + v.e_runes, j);
    if (s < s1) {
      DtkSegIn in;
      in.tok = bt + et; in.sent = bs + es; in.text = bx + ex; in.sev = bv + ev; in.runes = br + er;
      in.e_pos = below ? jp : ce_pos; in.e_tok = below ? jt : ce_tok; in.e_runes = below ? jr : ce_runes;
      A.seg_in[s] = in;
    }
    if (mE) {
      const int jl = highest(mE);
      ce_pos = __shfl(v.e_pos, jl); ce_tok = __shfl(bt + et + v.e_tok, jl); ce_runes = __shfl(br + er + v.e_runes, jl);
      any_e = true;
    }
    bt += tt; bs += ts; bx += tx; bv += tv; br += tr;
  }
  // The double array keeps its window over an EOT (datok.go:1019-1030): its carries are only
  // closed-form in documents without one; the others are compacted by their first segment alone.
  if (lane == 0) A.doc_seq[d] = (A.kind != DTK_KIND_MATRIX && any_e) ? 1u : 0u;
}


// ------------------------------------------------------- exclusive scan (x3)
//
// Turns the per-document counts into CSR row offsets (totals at [n_docs]) and
// counts flagged documents.  One 1024-thread block; each thread
// owns a contiguous slice, a block-level scan links the slices.

// One block: per-document counts -> CSR offsets (three arrays) + the number of flagged documents.
// Each thread adds up a few consecutive documents, the 16 waves scan with shuffles, one barrier
// links them.  With `fix` set it also does k_spec_fix's per-document step (one launch less on the
// batch's critical path).
#define SCAN1_TB 1024u
__global__ __launch_bounds__(SCAN1_TB) void k_scan3(const uint64_t *ca, const uint64_t *cb, const uint64_t *cc,
                                                uint64_t *a, uint64_t *b, uint64_t *c, uint32_t n,
                                                uint64_t *totals, const uint32_t *status, DtkSpecArgs S,
                                                uint32_t *redo_out, uint32_t *n_bad, int fix, const uint32_t *skip_if) {
  if (skip_if && *skip_if != 0u) return;
  __shared__ uint64_t wsum[3][SCAN1_TB / WAVE];
  __shared__ uint32_t wfl[SCAN1_TB / WAVE];
  const uint32_t T = blockDim.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  const uint32_t per = (n + T - 1) / T;
  const uint32_t lo = tid * per < n ? tid * per : n;
  const uint32_t hi = lo + per < n ? lo + per : n;
  uint64_t sa = 0, sb = 0, sc = 0;
  uint32_t fl = 0;
  // four documents at a time: their loads first, one wait -- a loop of dependent round trips was most of this kernel's
  // 15 us.  (Not all eight of a thread in registers: a 1024-thread block with 100 VGPRs per lane has to wait for a
  // whole CU to drain while other batches' walks fill the chip -- three batches in flight lost 10 %.)
  constexpr uint32_t G = 4u;
  for (uint32_t i0 = lo; i0 < hi; i0 += G) {
    uint64_t va[G], vb[G], vc[G];
    uint32_t vs[G], vf[G];
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
      const uint32_t i = i0 + j;
      const bool ok = i < hi;
      va[j] = ok ? ca[i] : 0ull; vb[j] = ok ? cb[i] : 0ull; vc[j] = ok ? cc[i] : 0ull;
      vs[j] = ok ? status[i] : 0u;
      vf[j] = (ok && fix) ? S.fail_lane[i] : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
      sa += va[j]; sb += vb[j]; sc += vc[j]; fl += vs[j] != 0u;
      if (fix && i0 + j < hi) {
        const uint32_t bad = ~vf[j];
        if (bad == 0xFFFFFFFFu) redo_out[i0 + j] = 0xFFFFFFFFu; else mark_redo(S, i0 + j, bad, redo_out, n_bad);
      }
    }
  }
  uint64_t xa = sa, xb = sb, xc = sc;
  uint32_t xf = fl;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const uint64_t ya = __shfl_up(xa, o), yb = __shfl_up(xb, o), yc = __shfl_up(xc, o);
    const uint32_t yf = __shfl_up(xf, o);
    if ((int)lane >= o) { xa += ya; xb += yb; xc += yc; xf += yf; }
  }
  if (lane == WAVE - 1) { wsum[0][wid] = xa; wsum[1][wid] = xb; wsum[2][wid] = xc; wfl[wid] = xf; }
  __syncthreads();
  uint64_t ba = 0, bb = 0, bc = 0;
  uint32_t bf = 0;
  for (uint32_t w = 0; w < wid; w++) { ba += wsum[0][w]; bb += wsum[1][w]; bc += wsum[2][w]; bf += wfl[w]; }
  uint64_t ra = ba + xa - sa, rb = bb + xb - sb, rc = bc + xc - sc;
  for (uint32_t i0 = lo; i0 < hi; i0 += G) {
    uint64_t va[G], vb[G], vc[G];
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
      const uint32_t i = i0 + j;
      const bool ok = i < hi;
      va[j] = ok ? ca[i] : 0ull; vb[j] = ok ? cb[i] : 0ull; vc[j] = ok ? cc[i] : 0ull;
    }
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
      const uint32_t i = i0 + j;
      if (i < hi) { a[i] = ra; b[i] = rb; c[i] = rc; }
      ra += va[j]; rb += vb[j]; rc += vc[j];
    }
  }
  if (tid == T - 1) {
    a[n] = ba + xa; b[n] = bb + xb; c[n] = bc + xc;
    totals[0] = ba + xa; totals[1] = bb + xb; totals[2] = bc + xc;
    totals[3] = bf + xf;
  }
}

// ------------------------------------------------------------------ results to the host
//
// One launch behind the compaction: every selected result array goes to its page-locked host buffer (DtkToHostArgs).
// Sources and destinations are 16-byte aligned (hipMalloc / hipHostMalloc); a wave writes 1 KiB of consecutive bytes
// per instruction -- posted writes over the link, nothing waits for them but the end of the wave.  The last block
// to finish (one counter add per block) ... is not needed: the host only looks after the stream's event.
// The kernel must not get in the way of the walks of other batches: with 2048 waves storing as fast as they could the
// stores queued up in every CU's memory pipeline and a k_spec_both beside it took 0.49 instead of 0.13 ms.  So few
// waves (one per block, spread over the CUs), each with ONE 1 KiB store instruction in flight: 128 KiB under way on
// the link at any time is what ~50 GB/s x 2 us of round trip need.
#ifndef DTK_TOHOST_WAVES
#define DTK_TOHOST_WAVES 128u
#endif
__global__ __launch_bounds__(WAVE) void k_to_host(DtkToHostArgs A) {
  if (A.skip_if && *A.skip_if != 0u) return;
  // (every wave decides alike: a count beyond its buffer means the host has to grow buffers and copy by itself)
  bool fits = true;
  for (uint32_t i = 0; i < A.n; i++)
    if (A.count_from[i] >= 0 && A.totals[A.count_from[i]] > A.cap[i]) fits = false;
  if (!fits) return;
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = 0; i < A.n; i++) {
    const uint64_t bytes = A.count_from[i] >= 0 ? A.totals[A.count_from[i]] * A.bytes[i] : A.bytes[i];
    const uint4 *__restrict__ s16 = reinterpret_cast<const uint4 *>(A.src[i]);
    uint4 *__restrict__ d16 = reinterpret_cast<uint4 *>(A.dst[i]);
    const uint64_t n16 = bytes >> 4;
    // pieces of 64 x 16 B, dealt round-robin to the waves (rotated by the array's number: the short arrays do not
    // all land on wave 0)
    for (uint64_t p = (blockIdx.x + gridDim.x - i % gridDim.x) % gridDim.x; p * WAVE < n16; p += gridDim.x) {
      const uint64_t j = p * WAVE + lane;
      if (j < n16) {
        const uint4 v = s16[j];
        d16[j] = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // one store under way per wave
    }
    if (blockIdx.x == i % gridDim.x && lane < (bytes & 15u))
      reinterpret_cast<uint8_t *>(A.dst[i])[(n16 << 4) + lane] = reinterpret_cast<const uint8_t *>(A.src[i])[(n16 << 4) + lane];
  }
  if (blockIdx.x == 0 && lane == 0) *A.done = A.epoch;
}

extern "C" int dtk_launch_to_host(const DtkToHostArgs *args, void *stream) {
  hipLaunchKernelGGL(k_to_host, dim3(DTK_TOHOST_WAVES), dim3(WAVE), 0, (hipStream_t)stream, *args);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- launchers

// ---- clears: the accumulator block and the two event arrays of a run in one launch (16-byte stores)
__global__ __launch_bounds__(256) void k_clear2(uint4 *__restrict__ a, uint64_t na16, uint4 *__restrict__ b,
                                                uint64_t nb16) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na16 + nb16; i += stride) {
    if (i < na16) a[i] = z; else b[i - na16] = z;
  }
}

// both pointers 16-byte aligned, both sizes multiples of 16
extern "C" int dtk_launch_clear2(void *a, uint64_t a_bytes, void *b, uint64_t b_bytes, void *stream) {
  const uint64_t n16 = (a_bytes + b_bytes) / 16;
  if (n16 == 0) return 0;
  uint64_t blocks = (n16 + 256ull * 4 - 1) / (256ull * 4);  // 4 stores per thread
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_clear2, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint4 *)a, a_bytes / 16,
                     (uint4 *)b, b_bytes / 16);
  return (int)hipGetLastError();
}

extern "C" int dtk_launch_symbolize(const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs,
                                    uint64_t total, const DtkSigmaDev *sig, void *sym, int padded,
                                    const uint32_t *blk_doc, unsigned long long *n_invalid, uint32_t *rs_bits,
                                    uint32_t *ev_bits, uint32_t bit_words, void *acc, uint64_t acc_bytes,
                                    uint64_t epoch, void *stream) {
  if (total == 0 || n_docs == 0) return 0;
  const uint32_t blocks = (uint32_t)((total + SYM_BLOCK_BYTES - 1) / SYM_BLOCK_BYTES);
  // ALIGNED4 may read up to 3 bytes past `total`: true for the batch's own (padded) buffer;
  // a caller-owned device buffer only qualifies when its size is a multiple of 4
  const bool al4 = (((uintptr_t)text) & 3u) == 0 && (padded || (total & 3u) == 0);
  auto go = [&](auto k) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(SYM_THREADS), 0, (hipStream_t)stream, text, doc_off, n_docs,
                       total, *sig, sym, blk_doc, n_invalid, rs_bits, ev_bits, bit_words, (uint4 *)acc, (uint32_t)(acc_bytes / 16),
                       (unsigned long long)epoch);
  };
  if (sig->n_codes) { if (al4) go(k_symbolize<true, true>); else go(k_symbolize<false, true>); }
  else { if (al4) go(k_symbolize<true, false>); else go(k_symbolize<false, false>); }
  return (int)hipGetLastError();
}

// codes: the symbol stream holds codes (DtkSym::lut) -- what the lean loop reads
template <typename F>
static int with_trans(const DtkTableDev *tab, bool codes, F &&f) {
  if (tab->kind == DTK_KIND_MATRIX) {
    if (tab->fused) {
      MatrixFusedTrans tr{(const uint32_t *)tab->tab, tab->stride, tab->n_eps, tab->start, tab->ident_guard};
      // (da_dense: a double-array tokenizer laid out as a fused matrix -- the table's walk, datok.go's EOT rules)
      auto call = [&](auto t) { if (tab->da_dense) f(t, std::false_type{}); else f(t, std::true_type{}); };
      if (tab->ident_guard == 0xFFFFFFFFu && !tab->plain_walk && codes) {  // the lean loop applies
        MatrixLeanTrans lt;
        static_cast<MatrixFusedTrans &>(lt) = tr;
        call(lt);
      } else {
        call(tr);
      }
    } else if (tab->entry_bytes == 2) {
      MatrixTrans<uint16_t> tr{(const uint16_t *)tab->tab, tab->stride, tab->n_eps, tab->start};
      f(tr, std::true_type{});
    } else {
      MatrixTrans<uint32_t> tr{(const uint32_t *)tab->tab, tab->stride, tab->n_eps, tab->start};
      f(tr, std::true_type{});
    }
  } else {
    DaTrans tr{(const uint2 *)tab->tab, tab->da_len, tab->da_size, tab->da_base1};
    f(tr, std::false_type{});
  }
  return (int)hipGetLastError();
}

extern "C" int dtk_launch_walk(const DtkTableDev *tab, const DtkWalkArgs *args, void *stream) {
  if (args->n_docs == 0) return 0;
  const uint32_t blocks = (args->n_docs + WAVE - 1) / WAVE;
  hipStream_t s = (hipStream_t)stream;
  return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
    using TR = decltype(tr);
    hipLaunchKernelGGL((k_walk_doc<TR, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, tr, *args,
                       tab->epsilon, tab->unknown, tab->identity);
  });
}

// stage: 6 start records + walk, 7 link + verify, 4 fix (first pass; or 0 start records, 1 link, 2 walk, 3 verify, 4);
//        5 clear, 6 plan, 2 walk, 7 check (repair rounds, spec->redo_from set)
extern "C" int dtk_launch_spec(const DtkTableDev *tab, const DtkWalkArgs *args, const DtkSpecArgs *spec,
                               int stage, uint32_t cmp_mask, uint32_t *redo_out, uint32_t *n_bad,
                               void *stream) {
  if (args->n_docs == 0 || spec->n_lanes == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t lane_blocks = (spec->n_lanes + WAVE - 1) / WAVE;
  const uint32_t lane_blocks256 = (spec->n_lanes + 255) / 256;
  const uint32_t doc_blocks = (args->n_docs + 255) / 256;
  switch (stage) {
    case 0:
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_start<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE), 0, s,
                           tr, *args, *spec, tab->epsilon, tab->unknown, tab->identity);
      });
    case 6:  // first pass: start records and chunk walk in one launch
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_both<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE),
                           3u * spec->lds_words * sizeof(uint32_t), s, tr, *args, *spec, tab->epsilon, tab->unknown,
                           tab->identity);
      });
    case 1:
      hipLaunchKernelGGL(k_spec_link, dim3(lane_blocks256), dim3(256), 0, s, *spec);
      return (int)hipGetLastError();
    case 2:
      return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
        using TR = decltype(tr);
        hipLaunchKernelGGL((k_spec_walk<TR, decltype(is_matrix)::value>), dim3(lane_blocks), dim3(WAVE),
                           3u * spec->lds_words * sizeof(uint32_t), s, tr, *args, *spec, tab->epsilon, tab->unknown,
                           tab->identity);
      });
    case 3:
      hipLaunchKernelGGL(k_spec_verify, dim3((spec->n_lanes + VERIFY_TB - 1u) / VERIFY_TB), dim3(VERIFY_TB), 0, s, *args, *spec, cmp_mask, 0u);
      return (int)hipGetLastError();
    case 7:  // first pass behind k_spec_both: link + verify in one
      hipLaunchKernelGGL(k_spec_verify, dim3((spec->n_lanes + VERIFY_TB - 1u) / VERIFY_TB), dim3(VERIFY_TB), 0, s, *args, *spec, cmp_mask, 1u);
      return (int)hipGetLastError();
    case 4:
      hipLaunchKernelGGL(k_spec_fix, dim3(doc_blocks), dim3(256), 0, s, *args, *spec, redo_out, n_bad);
      return (int)hipGetLastError();
    case 5:  // repair round, before link / walk / verify / fix
      hipLaunchKernelGGL(k_redo_spread, dim3(lane_blocks256), dim3(256), 0, s, *spec);
      hipLaunchKernelGGL(k_redo_reset, dim3(doc_blocks), dim3(256), 0, s, *args, *spec);
      return (int)hipGetLastError();
  }
  return -1;
}

extern "C" int dtk_launch_redo_clear(const DtkWalkArgs *args, const DtkSpecArgs *spec, void *stream) {
  if (args->bit_words == 0) return 0;
  hipLaunchKernelGGL(k_redo_clear, dim3((args->bit_words + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *args, *spec);
  return (int)hipGetLastError();
}

// the exact pass over the listed documents (one lane each; the general loop for every table kind)
extern "C" int dtk_launch_exact(const DtkTableDev *tab, const DtkExactArgs *args, void *stream) {
  if (args->n == 0) return 0;
  const uint32_t blocks = (args->n + WAVE - 1) / WAVE;
  hipStream_t s = (hipStream_t)stream;
  return with_trans(tab, args->sym.lut != nullptr, [&](auto tr, auto is_matrix) {
    using TR = decltype(tr);
    if constexpr (TR::LEAN) {
      const MatrixFusedTrans base = tr;
      hipLaunchKernelGGL((k_exact_doc<MatrixFusedTrans, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, base, *args,
                         tab->epsilon, tab->unknown, tab->identity);
    } else {
      hipLaunchKernelGGL((k_exact_doc<TR, decltype(is_matrix)::value>), dim3(blocks), dim3(WAVE), 0, s, tr, *args,
                         tab->epsilon, tab->unknown, tab->identity);
    }
  });
}

// small_max: documents of at most that many bytes are compacted by one lane each (k_compact_small; 0: none);
// big_docs / n_big: the other documents (the wave-per-document grid then covers only those; segment mode: all segments)
// which: 1 the documents without an EOT call (and the lane-per-document kernel), 2 the documents with one, 3 both
extern "C" int dtk_launch_compact(const DtkCompactArgs *args, uint32_t small_max, const uint32_t *big_docs, uint32_t n_big,
                                  int which, void *stream) {
  if (args->n_docs == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (small_max && (which & 1))
    hipLaunchKernelGGL(k_compact_small, dim3((args->n_docs + 255u) / 256u), dim3(256), 0, s, *args, small_max);
  const uint32_t grid = args->seg_doc ? args->n_segs : (small_max ? n_big : args->n_docs);
  const uint32_t *list = args->seg_doc ? nullptr : (small_max ? big_docs : nullptr);
  if (grid && (which & 1)) hipLaunchKernelGGL(k_compact_plain, dim3(grid), dim3(WAVE), 0, s, *args, small_max, list);
  if (grid && (which & 2)) hipLaunchKernelGGL(k_compact_eot, dim3(grid), dim3(WAVE), 0, s, *args, small_max, list);
  return (int)hipGetLastError();
}

// the carries of the segments of long documents (before dtk_launch_compact in segment mode)
extern "C" int dtk_launch_seg_prepare(const DtkCompactArgs *args, const uint32_t *doc_seg0, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_seg_sum, dim3(args->n_segs), dim3(WAVE), 0, s, *args);
  hipLaunchKernelGGL(k_seg_scan, dim3(args->n_docs), dim3(WAVE), 0, s, *args, doc_seg0);
  return (int)hipGetLastError();
}

// Many documents: the same scan in three launches (tile sums, scan of the sums, tiles).
#define SCAN_TB 256u
#define SCAN_PER 8u
#define SCAN_TILE (SCAN_TB * SCAN_PER)

__device__ __forceinline__ uint64_t scan_block_excl(uint64_t v, uint64_t *sh, uint64_t &total) {
  const uint32_t tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (uint32_t o = 1; o < SCAN_TB; o <<= 1) {
    const uint64_t x = tid >= o ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += x;
    __syncthreads();
  }
  total = sh[SCAN_TB - 1];
  const uint64_t ex = sh[tid] - v;
  __syncthreads();
  return ex;
}

__global__ __launch_bounds__(SCAN_TB) void k_scan3_sums(const uint64_t *ca, const uint64_t *cb, const uint64_t *cc,
                                                        const uint32_t *status, uint32_t n, uint64_t *ws, const uint32_t *skip_if) {
  if (skip_if && *skip_if != 0u) return;
  __shared__ uint64_t sh[SCAN_TB];
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
  uint64_t s[4] = {0, 0, 0, 0};
  for (uint32_t i = i0; i < i0 + SCAN_PER && i < n; i++) { s[0] += ca[i]; s[1] += cb[i]; s[2] += cc[i]; s[3] += status[i] != 0; }
  for (int j = 0; j < 4; j++) {
    uint64_t tot;
    (void)scan_block_excl(s[j], sh, tot);
    if (threadIdx.x == 0) ws[4u * blockIdx.x + j] = tot;
  }
}

__global__ __launch_bounds__(SCAN_TB) void k_scan3_mid(uint64_t *ws, uint32_t nb, uint64_t *a, uint64_t *b, uint64_t *c,
                                                       uint32_t n, uint64_t *totals, const uint32_t *skip_if) {
  if (skip_if && *skip_if != 0u) return;
  __shared__ uint64_t sh[SCAN_TB];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nb + SCAN_TB - 1) / SCAN_TB;
  const uint32_t lo = min(tid * per, nb), hi = min(lo + per, nb);
  for (int j = 0; j < 4; j++) {
    uint64_t sm = 0;
    for (uint32_t i = lo; i < hi; i++) sm += ws[4u * i + j];
    uint64_t tot;
    uint64_t run = scan_block_excl(sm, sh, tot);
    for (uint32_t i = lo; i < hi; i++) { const uint64_t v = ws[4u * i + j]; ws[4u * i + j] = run; run += v; }
    if (tid == 0) {
      totals[j] = tot;
      if (j == 0) a[n] = tot;
      if (j == 1) b[n] = tot;
      if (j == 2) c[n] = tot;
    }
  }
}

__global__ __launch_bounds__(SCAN_TB) void k_scan3_apply(const uint64_t *ca, const uint64_t *cb, const uint64_t *cc,
                                                         uint64_t *a, uint64_t *b, uint64_t *c, uint32_t n,
                                                         const uint64_t *ws, const uint32_t *skip_if) {
  if (skip_if && *skip_if != 0u) return;
  __shared__ uint64_t sh[SCAN_TB];
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
  const uint64_t *src[3] = {ca, cb, cc};
  uint64_t *dst[3] = {a, b, c};
  for (int j = 0; j < 3; j++) {
    uint64_t v[SCAN_PER], sm = 0;
    for (uint32_t q = 0; q < SCAN_PER; q++) { v[q] = i0 + q < n ? src[j][i0 + q] : 0; sm += v[q]; }
    uint64_t tot;
    uint64_t run = ws[4u * blockIdx.x + j] + scan_block_excl(sm, sh, tot);
    for (uint32_t q = 0; q < SCAN_PER; q++) {
      if (i0 + q < n) dst[j][i0 + q] = run;
      run += v[q];
    }
  }
}

// fix_spec != nullptr: also k_spec_fix's per-document step (only done in the one-block case: returns 1 if it was)
extern "C" int dtk_launch_scan3(const uint64_t *ca, const uint64_t *cb, const uint64_t *cc, uint64_t *a,
                                uint64_t *b, uint64_t *c, uint32_t n_docs, uint64_t *totals,
                                const uint32_t *status, uint64_t *ws, const DtkSpecArgs *fix_spec,
                                uint32_t *redo_out, uint32_t *n_bad, const uint32_t *skip_if, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_docs <= 8192u || !ws) {
    DtkSpecArgs none{};
    // 1024 threads: the kernel's time is the threads' serial loops over their documents (512: 23 instead of 16 us
    // for 4096 documents, 256: 32 us)
    hipLaunchKernelGGL(k_scan3, dim3(1), dim3(SCAN1_TB), 0, s, ca, cb, cc, a, b, c, n_docs, totals, status,
                       fix_spec ? *fix_spec : none, redo_out, n_bad, fix_spec ? 1 : 0, skip_if);
  } else {
    if (fix_spec) return -1;  // the caller runs k_spec_fix itself for that many documents
    const uint32_t nb = (n_docs + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan3_sums, dim3(nb), dim3(SCAN_TB), 0, s, ca, cb, cc, status, n_docs, ws, skip_if);
    hipLaunchKernelGGL(k_scan3_mid, dim3(1), dim3(SCAN_TB), 0, s, ws, nb, a, b, c, n_docs, totals, skip_if);
    hipLaunchKernelGGL(k_scan3_apply, dim3(nb), dim3(SCAN_TB), 0, s, ca, cb, cc, a, b, c, n_docs, ws, skip_if);
  }
  return (int)hipGetLastError();
}
