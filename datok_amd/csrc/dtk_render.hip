// dtk_render.hip -- NewTokenWriter's byte output on the device (gfx950).
//
// token_writer.go:36-175 restated as a data-parallel gather: what the reference prints call by
// call is, per text,
//     [ surface "\n" per Token (TOKENS) , "\n" per SentenceEnd (SENTENCES) , in call order ]
//     [ token positions  "s e s e ...\n" (TOKEN_POS) ] [ sentence positions (SENTENCE_POS) ]
//     or, without a position flag, one "\n" per TextEnd.
// Every piece has a size that depends only on the compacted arrays, so three exclusive scans
// (surface bytes, token-position digits, sentence-position digits) plus the per-token count of
// earlier SentenceEnd calls give every piece its offset; all separators that are not written
// explicitly are newlines, so the buffer is pre-filled with '\n'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dtk_internal.h"

namespace {

constexpr uint32_t RB = 256;          // threads per block
constexpr uint32_t RI = 4;            // items per thread
constexpr uint32_t RTILE = RB * RI;   // items per block

// strconv.Itoa: length and digits
__device__ __forceinline__ uint32_t dec_len(int32_t v) {
  const uint32_t neg = v < 0 ? 1u : 0u;
  const uint32_t u = neg ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
  return 1u + (u >= 10u) + (u >= 100u) + (u >= 1000u) + (u >= 10000u) + (u >= 100000u) + (u >= 1000000u) +
         (u >= 10000000u) + (u >= 100000000u) + (u >= 1000000000u) + neg;
}

__device__ __forceinline__ void dec_put(uint8_t *o, int32_t v, uint32_t n) {
  uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
  uint32_t i = n;
  do { o[--i] = (uint8_t)('0' + u % 10u); u /= 10u; } while (u != 0u && i > 0u);
  if (v < 0) o[0] = '-';
}

struct Pair { uint64_t a, p; };

// largest d in [0, n) with off[d] <= x   (off is non-decreasing, off[0] == 0)
__device__ __forceinline__ uint32_t row_of(const uint64_t *off, uint32_t n, uint64_t x) {
  uint32_t lo = 0, hi = n;  // invariant: off[lo] <= x, (hi == n or off[hi] > x)
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}


// string(buf[offset:]) re-encodes the window's runes (token_writer.go:85): a byte that did not
// decode (U+FFFD, width 1) leaves as EF BF BD.  Such bytes are the rune starts of width 1 whose
// rune is >= 256 in the symbol stream.
__device__ __forceinline__ bool is_invalid_byte(const DtkSym &S, uint64_t i) {
  const uint32_t e = dtk_sym_entry(S, i);
  return DTK_SYM_WIDTH(e) == 1u && ((e >> DTK_SYM_CLS_SHIFT) & 3u) >= 2u;
}

__device__ __forceinline__ uint32_t invalid_in(const DtkRenderArgs &R, uint64_t k, uint32_t d) {
  const uint64_t sy = R.doc_off[d];
  uint32_t n = 0;
  for (uint32_t q = R.bstart[k]; q < R.bend[k]; q++) n += is_invalid_byte(R.sym, sy + q) ? 1u : 0u;
  return n;
}

__device__ __forceinline__ Pair tok_weight(const DtkRenderArgs &R, uint64_t k) {
  Pair w{0, 0};
  if (k < R.n_tok) {
    if (R.flags & 1u) {  // surface + '\n'
      w.a = (uint64_t)(R.bend[k] - R.bstart[k]) + 1u;
      if (R.sym.base) w.a += 2u * invalid_in(R, k, row_of(R.tok_off, R.n_docs, k));
    }
    if (R.flags & 4u) w.p = (uint64_t)dec_len(R.rstart[k]) + dec_len(R.rend[k]) + 2u;  // "s e "
  }
  return w;
}

__device__ __forceinline__ uint64_t sent_weight(const DtkRenderArgs &R, uint64_t s) {
  return (s < R.n_sent && (R.flags & 8u)) ? (uint64_t)dec_len(R.sent[s]) + 1u : 0u;
}

// block-wide inclusive scan of one value per thread (RB threads); returns the exclusive prefix
// of this thread and the block total
__device__ __forceinline__ uint64_t block_excl(uint64_t v, uint64_t *sh, uint64_t &total) {
  const uint32_t tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (uint32_t o = 1; o < RB; o <<= 1) {
    const uint64_t x = tid >= o ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += x;
    __syncthreads();
  }
  total = sh[RB - 1];
  const uint64_t ex = sh[tid] - v;
  __syncthreads();
  return ex;
}

// phase 1: per-tile sums.  grid.x = tiles over tokens, then tiles over sentence ints.
__global__ __launch_bounds__(RB) void k_render_sums(DtkRenderArgs R, uint32_t tok_tiles) {
  __shared__ uint64_t sh[RB];
  const uint32_t b = blockIdx.x;
  uint64_t tot;
  if (b < tok_tiles) {
    Pair s{0, 0};
    const uint64_t k0 = (uint64_t)b * RTILE + threadIdx.x * RI;
    for (uint32_t i = 0; i < RI; i++) { const Pair w = tok_weight(R, k0 + i); s.a += w.a; s.p += w.p; }
    (void)block_excl(s.a, sh, tot);
    if (threadIdx.x == 0) R.blkA[b] = tot;
    (void)block_excl(s.p, sh, tot);
    if (threadIdx.x == 0) R.blkP[b] = tot;
  } else {
    const uint32_t c = b - tok_tiles;
    uint64_t s = 0;
    const uint64_t s0 = (uint64_t)c * RTILE + threadIdx.x * RI;
    for (uint32_t i = 0; i < RI; i++) s += sent_weight(R, s0 + i);
    (void)block_excl(s, sh, tot);
    if (threadIdx.x == 0) R.blkQ[c] = tot;
  }
}

// phase 2: exclusive scans of the tile sums and of the per-document SentenceEnd counts
// (one block; slices per thread linked by a block scan)
__device__ void scan_inplace_u64(uint64_t *x, uint64_t n, uint64_t *sh, uint64_t *total_out) {
  const uint32_t tid = threadIdx.x;
  const uint64_t per = (n + RB - 1) / RB;
  const uint64_t lo = min((uint64_t)tid * per, n), hi = min(lo + per, n);
  uint64_t s = 0;
  for (uint64_t i = lo; i < hi; i++) s += x[i];
  uint64_t tot;
  uint64_t run = block_excl(s, sh, tot);
  for (uint64_t i = lo; i < hi; i++) { const uint64_t v = x[i]; x[i] = run; run += v; }
  if (total_out && tid == 0) *total_out = tot;
}

__global__ __launch_bounds__(RB) void k_render_scan_tiles(DtkRenderArgs R, uint32_t tok_tiles, uint32_t sent_tiles) {
  __shared__ uint64_t sh[RB];
  scan_inplace_u64(R.blkA, tok_tiles, sh, nullptr);
  scan_inplace_u64(R.blkP, tok_tiles, sh, nullptr);
  scan_inplace_u64(R.blkQ, sent_tiles, sh, nullptr);
  // ns_off[d] = SentenceEnd calls in documents before d; [n_docs] = all
  {
    const uint32_t tid = threadIdx.x;
    const uint64_t n = R.n_docs;
    const uint64_t per = (n + RB - 1) / RB;
    const uint64_t lo = min((uint64_t)tid * per, n), hi = min(lo + per, n);
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s += R.doc_ns[i];
    uint64_t tot;
    uint64_t run = block_excl(s, sh, tot);
    for (uint64_t i = lo; i < hi; i++) { R.ns_off[i] = run; run += R.doc_ns[i]; }
    if (tid == 0) R.ns_off[n] = tot;
  }
}

// phase 3: the scans themselves (A, P over tokens; Q over sentence ints), n+1 entries each
__global__ __launch_bounds__(RB) void k_render_offsets(DtkRenderArgs R, uint32_t tok_tiles) {
  __shared__ uint64_t sh[RB];
  const uint32_t b = blockIdx.x;
  uint64_t tot;
  if (b < tok_tiles) {
    Pair w[RI], s{0, 0};
    const uint64_t k0 = (uint64_t)b * RTILE + threadIdx.x * RI;
    for (uint32_t i = 0; i < RI; i++) { w[i] = tok_weight(R, k0 + i); s.a += w[i].a; s.p += w[i].p; }
    uint64_t ra = R.blkA[b] + block_excl(s.a, sh, tot);
    uint64_t rp = R.blkP[b] + block_excl(s.p, sh, tot);
    for (uint32_t i = 0; i < RI; i++) {
      if (k0 + i <= R.n_tok) { R.A[k0 + i] = ra; R.P[k0 + i] = rp; }
      ra += w[i].a; rp += w[i].p;
    }
  } else {
    const uint32_t c = b - tok_tiles;
    uint64_t w[RI], s = 0;
    const uint64_t s0 = (uint64_t)c * RTILE + threadIdx.x * RI;
    for (uint32_t i = 0; i < RI; i++) { w[i] = sent_weight(R, s0 + i); s += w[i]; }
    uint64_t rq = R.blkQ[c] + block_excl(s, sh, tot);
    for (uint32_t i = 0; i < RI; i++) {
      if (s0 + i <= R.n_sent) R.Q[s0 + i] = rq;
      rq += w[i];
    }
  }
}

// per text g (and the sentinel g == n_text): where its three regions start
__global__ __launch_bounds__(RB) void k_render_texts(DtkRenderArgs R) {
  const uint64_t g = (uint64_t)blockIdx.x * RB + threadIdx.x;
  if (g > R.n_text) return;
  const uint64_t S = (R.flags & 2u) ? 1u : 0u;
  const uint64_t NP = (R.flags & 12u) ? 0u : 1u;  // without position flags TextEnd prints "\n"
  if (g == R.n_text) {
    R.tx_base[g] = R.A[R.n_tok] + R.P[R.n_tok] + R.Q[R.n_sent] + S * R.ns_off[R.n_docs] + NP * g;
    return;
  }
  const uint32_t d = row_of(R.text_off, R.n_docs, g);
  const bool first = g == R.text_off[d];
  const uint64_t TB = R.tok_off[d] + (first ? 0u : R.ttok[g - 1]), TE = R.tok_off[d] + R.ttok[g];
  const uint64_t QB = R.sent_off[d] + (first ? 0u : R.tsent[g - 1]);
  const uint64_t SS = R.ns_off[d] + (first ? 0u : R.ts_end[g - 1]), SE = R.ns_off[d] + R.ts_end[g];
  const uint64_t base = R.A[TB] + R.P[TB] + R.Q[QB] + S * SS + NP * g;
  const uint64_t pos_start = base + (R.A[TE] - R.A[TB]) + S * (SE - SS);
  const uint64_t sent_start = pos_start + (R.P[TE] - R.P[TB]);
  R.tx_base[g] = base;
  R.tx_stream[g] = base - R.A[TB] - S * SS;  // + A[K] + S * (SentenceEnd calls before token K)
  R.tx_pos[g] = pos_start - R.P[TB];         // + P[K]
  R.tx_sent[g] = sent_start - R.Q[QB];       // + Q[s]
}

__global__ __launch_bounds__(RB) void k_render_doc_offsets(DtkRenderArgs R) {
  const uint64_t d = (uint64_t)blockIdx.x * RB + threadIdx.x;
  if (d > R.n_docs) return;
  R.out_off[d] = R.tx_base[d == R.n_docs ? R.n_text : R.text_off[d]];
}

// one thread per token: surface bytes and/or its two position ints
__global__ __launch_bounds__(RB) void k_render_tokens(DtkRenderArgs R) {
  const uint64_t K = (uint64_t)blockIdx.x * RB + threadIdx.x;
  if (K >= R.n_tok) return;
  const uint32_t d = row_of(R.tok_off, R.n_docs, K);
  const uint32_t k = (uint32_t)(K - R.tok_off[d]);
  // text of the token: first record of the document whose token end lies behind k
  uint64_t lo = R.text_off[d], hi = R.text_off[d + 1];
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (R.ttok[mid] > k) hi = mid; else lo = mid + 1;
  }
  if (lo >= R.text_off[d + 1]) return;  // token behind the last TextEnd (flagged document)
  const uint64_t g = lo;
  if (R.flags & 1u) {
    const uint64_t S = (R.flags & 2u) ? 1u : 0u;
    const uint64_t o = R.tx_stream[g] + R.A[K] + S * (R.ns_off[d] + R.sbefore[K]);
    const uint32_t b0 = R.bstart[K], n = R.bend[K] - b0;
    if (o + n <= R.out_total) {
      const uint8_t *src = R.text + R.doc_off[d] + b0;
      uint8_t *dst = R.out + o;
      if (!R.sym.base) {
        for (uint32_t i = 0; i < n; i++) dst[i] = src[i];
      } else {  // the batch has bytes that print as U+FFFD
        const uint64_t sy = R.doc_off[d] + b0;
        const uint64_t lim = R.out_total - o;
        uint64_t w = 0;
        for (uint32_t i = 0; i < n; i++) {
          if (is_invalid_byte(R.sym, sy + i)) {
            if (w + 3u <= lim) { dst[w] = 0xEF; dst[w + 1] = 0xBF; dst[w + 2] = 0xBD; }
            w += 3u;
          } else {
            if (w < lim) dst[w] = src[i];
            w++;
          }
        }
      }
    }
  }
  if (R.flags & 4u) {
    uint64_t o = R.tx_pos[g] + R.P[K];
    const int32_t rs = R.rstart[K], re = R.rend[K];
    const uint32_t n1 = dec_len(rs), n2 = dec_len(re);
    if (o + n1 + n2 + 2u <= R.out_total) {
      dec_put(R.out + o, rs, n1);
      R.out[o + n1] = ' ';
      dec_put(R.out + o + n1 + 1u, re, n2);
      if (k + 1u != R.ttok[g]) R.out[o + n1 + 1u + n2] = ' ';  // the line's last int keeps the '\n'
    }
  }
}

// one thread per sentence int (SENTENCE_POS)
__global__ __launch_bounds__(RB) void k_render_sents(DtkRenderArgs R) {
  const uint64_t s = (uint64_t)blockIdx.x * RB + threadIdx.x;
  if (s >= R.n_sent) return;
  const uint32_t d = row_of(R.sent_off, R.n_docs, s);
  const uint32_t i = (uint32_t)(s - R.sent_off[d]);
  uint64_t lo = R.text_off[d], hi = R.text_off[d + 1];
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (R.tsent[mid] > i) hi = mid; else lo = mid + 1;
  }
  if (lo >= R.text_off[d + 1]) return;
  const uint64_t g = lo;
  const uint64_t o = R.tx_sent[g] + R.Q[s];
  const int32_t v = R.sent[s];
  const uint32_t n = dec_len(v);
  if (o + n + 1u <= R.out_total) {
    dec_put(R.out + o, v, n);
    if (i + 1u != R.tsent[g]) R.out[o + n] = ' ';
  }
}

}  // namespace

static inline uint32_t tiles_of(uint64_t n) { return (uint32_t)((n + 1 + RTILE - 1) / RTILE); }  // n+1 entries

extern "C" uint32_t dtk_render_tiles(uint64_t n) { return tiles_of(n); }

// stage 0: sizes (sums, scans, per-text regions, per-document offsets); stage 1: bytes
extern "C" int dtk_launch_render(const DtkRenderArgs *R, int stage, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t tt = tiles_of(R->n_tok), st = tiles_of(R->n_sent);
  if (stage == 0) {
    hipLaunchKernelGGL(k_render_sums, dim3(tt + st), dim3(RB), 0, s, *R, tt);
    hipLaunchKernelGGL(k_render_scan_tiles, dim3(1), dim3(RB), 0, s, *R, tt, st);
    hipLaunchKernelGGL(k_render_offsets, dim3(tt + st), dim3(RB), 0, s, *R, tt);
    hipLaunchKernelGGL(k_render_texts, dim3((uint32_t)((R->n_text + 1 + RB - 1) / RB)), dim3(RB), 0, s, *R);
    hipLaunchKernelGGL(k_render_doc_offsets, dim3((R->n_docs + 1 + RB - 1) / RB), dim3(RB), 0, s, *R);
  } else {
    if (R->n_tok && (R->flags & 5u))
      hipLaunchKernelGGL(k_render_tokens, dim3((uint32_t)((R->n_tok + RB - 1) / RB)), dim3(RB), 0, s, *R);
    if (R->n_sent && (R->flags & 8u))
      hipLaunchKernelGGL(k_render_sents, dim3((uint32_t)((R->n_sent + RB - 1) / RB)), dim3(RB), 0, s, *R);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
