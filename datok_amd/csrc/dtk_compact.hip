// dtk_compact.hip -- gfx950 kernel 3 of the batch tokenizer: event bitmaps -> the offset arrays NewTokenWriter
// (token_writer.go:36-175) would have collected; the scan that sizes the CSR rows; the clears; the copy of the results
// into page-locked host memory.
#include "dtk_device.h"

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
          if (A.tok_bstart) { A.tok_bstart[tok_base + k] = startP; A.tok_bend[tok_base + k] = P; }
          if (A.tok_rstart) { A.tok_rstart[tok_base + k] = rstart; A.tok_rend[tok_base + k] = rend; }
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
          if (A.tok_bstart) { A.tok_bstart[k] = sp; A.tok_bend[k] = P; }
          if (A.tok_rstart) { A.tok_rstart[k] = (int32_t)(sR - base); A.tok_rend[k] = (int32_t)(R - base); }
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
            // (a caller that reads one kind of offsets does not pay for the stores of the other: DTK_NO_*_OFFSETS)
            if (A.tok_bstart) { A.tok_bstart[k] = (v.x & 0xFFFFu) + pb; A.tok_bend[k] = (v.x >> 16) + pb; }
            if (A.tok_rstart) { A.tok_rstart[k] = (int32_t)((v.y & 0xFFFFu) + rb); A.tok_rend[k] = (int32_t)((v.y >> 16) + rb); }
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
          if (A.tok_bstart) { A.tok_bstart[tok_base + n_tok] = cs; A.tok_bend[tok_base + n_tok] = p; }
          if (A.tok_rstart) { A.tok_rstart[tok_base + n_tok] = rs; A.tok_rend[tok_base + n_tok] = posC; }
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
// per instruction -- posted writes over the link.  The host only looks after the stream's event (`done` tells whether
// the copy was made at all: not when a count exceeds its buffer or documents are still to be repaired).
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

// ---- DTK_R_TOK_RUNE16: a token's two rune offsets as the halves of one word (start low), for the way over the link
__global__ __launch_bounds__(256) void k_pack_r16(const int32_t *__restrict__ rs, const int32_t *__restrict__ re,
                                                  uint32_t *__restrict__ out, uint64_t n) {
  const uint64_t i4 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;  // (all three arrays are hipMalloc'ed: 16-byte loads)
  if (i4 + 4u <= n) {
    const int4 a = *reinterpret_cast<const int4 *>(rs + i4), b = *reinterpret_cast<const int4 *>(re + i4);
    *reinterpret_cast<uint4 *>(out + i4) =
        make_uint4(((uint32_t)a.x & 0xFFFFu) | ((uint32_t)b.x << 16), ((uint32_t)a.y & 0xFFFFu) | ((uint32_t)b.y << 16),
                   ((uint32_t)a.z & 0xFFFFu) | ((uint32_t)b.z << 16), ((uint32_t)a.w & 0xFFFFu) | ((uint32_t)b.w << 16));
  } else {
    for (uint64_t i = i4; i < n; i++) out[i] = ((uint32_t)rs[i] & 0xFFFFu) | ((uint32_t)re[i] << 16);
  }
}

extern "C" int dtk_launch_pack_r16(const int32_t *rs, const int32_t *re, uint32_t *out, uint64_t n, void *stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pack_r16, dim3((unsigned)((n + 1023u) / 1024u)), dim3(256), 0, (hipStream_t)stream, rs, re, out, n);
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

