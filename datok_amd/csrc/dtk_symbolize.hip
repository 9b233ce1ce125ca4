// dtk_symbolize.hip -- gfx950 kernel 1 of the batch tokenizer: bytes -> symbol stream (+ rune-start bitmap, + the
// clears of the run's event bitmaps and accumulators).  See dtk_walk.hip for the pipeline.
#include "dtk_device.h"

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
    // (block 0 from word 0: leading empty documents move d_lo, and with it `ga`, past the words of their positions)
    const uint32_t wa = blockIdx.x == 0 ? 0u : (uint32_t)(ga >> 5);
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
      // (a loop over the lane's set bits -- as many rounds as the lane with the most bytes >= 0x80 has, two or three in
      //  European text -- instead of eight conditional stores: 22.1 -> 21.6 us per 16 MiB saturated)
      for (uint32_t m = rare; __builtin_amdgcn_ballot_w64(m != 0u) != 0ull; m &= m - 1u)
        if (m) s_q[slot++] = (uint16_t)(i0 + (uint32_t)__builtin_ctz(m));
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

// ---------------------------------------------------------------- launcher

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

