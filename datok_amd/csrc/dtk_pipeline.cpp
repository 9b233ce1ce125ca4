// dtk_pipeline.cpp -- a corpus larger than one batch: slices cut at document boundaries run through a few
// dtk_batch objects in turn, so that the upload of one slice (pinned host memory -> HBM on that batch's own HIP
// stream) overlaps the walk of the slices before it and nothing waits for the host except the hand-over of
// finished slices, in order.  The reference streams its input through a bufio.Reader (matrix.go:372); this is the
// batch-speed counterpart for many documents (one TransduceTokenWriter call each).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/datok_gpu.h"

struct dtk_pipeline {
  uint64_t slice_bytes = 0;
  uint32_t slice_docs = 0;
  std::vector<dtk_batch *> slots;
  std::vector<uint32_t> first, count;  // the slice each slot holds
  std::vector<uint8_t> touched;        // an upload into the slot may be under way (error paths wait for it)
  std::vector<uint64_t> off;           // rebased offsets of the slice being submitted
  uint32_t fields = 0;                 // DTK_R_*: result arrays brought to the host for every slice (0: none)
  // Three streams for the whole pipeline, whatever its depth: uploads, kernels, downloads (the HIP runtime maps streams
  // onto four hardware queues; streams that share a queue serialise -- a stream per slot made depth 4 slower than 3).
  // Slices take turns on each: the link carries slice i + 2 in and slice i out while slice i + 1 is walked.
  hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
};

extern "C" void *dtk_pinned_alloc(size_t n) {
  void *p = nullptr;
  if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

extern "C" void dtk_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int dtk_pipeline_create(uint64_t slice_bytes, uint32_t slice_docs, uint32_t depth, dtk_pipeline **out) {
  if (!out || slice_bytes == 0 || slice_docs == 0 || depth == 0 || depth > 16) return DTK_E_ARG;
  *out = nullptr;
  dtk_pipeline *p = new dtk_pipeline();
  p->slice_bytes = slice_bytes;
  p->slice_docs = slice_docs;
  if (hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking) != hipSuccess) {
    dtk_pipeline_free(p);
    return DTK_E_HIP;
  }
  for (uint32_t i = 0; i < depth; i++) {
    dtk_batch *b = nullptr;
    int rc = dtk_batch_create(slice_bytes, slice_docs, &b);
    if (rc == DTK_OK) {
      p->slots.push_back(b);
      rc = dtk_batch_set_streams(b, p->s_run, p->s_up);
      if (rc == DTK_OK) rc = dtk_batch_set_download_stream(b, p->s_down);
    }
    if (rc != DTK_OK) { dtk_pipeline_free(p); return rc; }
  }
  p->first.assign(depth, 0);
  p->count.assign(depth, 0);
  p->touched.assign(depth, 0);
  *out = p;
  return DTK_OK;
}

extern "C" void dtk_pipeline_free(dtk_pipeline *p) {
  if (!p) return;
  for (dtk_batch *b : p->slots) dtk_batch_free(b);
  for (hipStream_t st : {p->s_up, p->s_run, p->s_down})
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  delete p;
}

extern "C" int dtk_pipeline_set_chunking(dtk_pipeline *p, uint32_t chunk_bytes, uint32_t warm_bytes) {
  if (!p) return DTK_E_ARG;
  for (dtk_batch *b : p->slots) {
    const int rc = dtk_batch_set_chunking(b, chunk_bytes, warm_bytes);
    if (rc != DTK_OK) return rc;
  }
  return DTK_OK;
}

extern "C" int dtk_pipeline_set_result_fields(dtk_pipeline *p, uint32_t fields) {
  if (!p) return DTK_E_ARG;
  for (dtk_batch *b : p->slots) {
    const int rc = dtk_batch_set_result_fields(b, fields ? fields : (uint32_t)DTK_R_ALL);
    if (rc != DTK_OK) return rc;
  }
  p->fields = fields;
  return DTK_OK;
}

// Slices whose kernels have finished start their way home now (nothing here blocks; the copies queue up on the download
// stream in slice order): the download of slice i runs under the walk of slice i + 1 and the upload of slice i + 2.
static void begin_downloads(dtk_pipeline *p, uint32_t oldest) {
  if (!p->fields) return;
  const uint32_t depth = (uint32_t)p->slots.size();
  for (uint32_t q = 0; q < depth; q++) {
    const uint32_t slot = (oldest + q) % depth;
    if (p->count[slot] == 0) continue;
    if (!dtk_batch_done(p->slots[slot])) break;
    if (dtk_batch_download_begin(p->slots[slot]) != DTK_OK) break;  // (deliver() reports the error)
  }
}

static int deliver(dtk_pipeline *p, uint32_t slot, dtk_slice_fn fn, void *user) {
  if (p->count[slot] == 0) return DTK_OK;
  begin_downloads(p, slot);
  dtk_totals t;
  int rc = dtk_batch_totals(p->slots[slot], &t);  // waits for the slice; repairs, capacity check
  if (rc == DTK_OK && p->fields) rc = dtk_batch_download_begin(p->slots[slot]);
  begin_downloads(p, slot);  // (the next slices' copies behind this one's, before the callback waits for its own)
  const uint32_t first = p->first[slot], n = p->count[slot];
  p->count[slot] = 0;
  p->touched[slot] = 0;
  if (rc != DTK_OK) return rc;
  return fn ? fn(user, first, n, p->slots[slot]) : DTK_OK;
}

extern "C" int dtk_pipeline_run(dtk_pipeline *p, const dtk_model *m, const uint8_t *text, const uint64_t *doc_off,
                                uint32_t n_docs, uint32_t flags, dtk_slice_fn fn, void *user) {
  if (!p || !m || !doc_off || (n_docs && doc_off[n_docs] && !text)) return DTK_E_ARG;
  const uint32_t depth = (uint32_t)p->slots.size();
  const uint64_t total = n_docs ? doc_off[n_docs] - doc_off[0] : 0;
  // Asynchronous uploads need page-locked memory.  Memory from dtk_pinned_alloc (or any registered range) is
  // used as it is; anything else is registered for the duration of the call (if that fails the copies are
  // staged by the runtime: correct, but the upload then blocks the submitting thread).
  bool registered = false;
  if (total) {
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, text + doc_off[0]) == hipSuccess &&
                        (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged);
    (void)hipGetLastError();
    if (!pinned) {
      registered = hipHostRegister((void *)(text + doc_off[0]), total, hipHostRegisterDefault) == hipSuccess;
      (void)hipGetLastError();
    }
  }
  int rc = DTK_OK;
  uint32_t i = 0, k = 0;
  while (i < n_docs && rc == DTK_OK) {
    // the slice [i, j): as many documents as fit
    const uint64_t lim = doc_off[i] + p->slice_bytes;
    uint32_t j = (uint32_t)(std::upper_bound(doc_off + i, doc_off + n_docs + 1, lim) - doc_off) - 1u;
    if (j > i + p->slice_docs) j = i + p->slice_docs;
    if (j == i) { rc = DTK_E_CAPACITY; break; }  // one document larger than a slice
    const uint32_t slot = k % depth;
    if ((rc = deliver(p, slot, fn, user)) != DTK_OK) break;
    p->off.resize((size_t)(j - i) + 1);
    for (uint32_t d = i; d <= j; d++) p->off[d - i] = doc_off[d] - doc_off[i];
    p->touched[slot] = 1;
    if ((rc = dtk_batch_set_input(p->slots[slot], text + doc_off[i], p->off.data(), j - i)) != DTK_OK) break;
    if ((rc = dtk_batch_run(m, p->slots[slot], flags)) != DTK_OK) break;
    p->first[slot] = i;
    p->count[slot] = j - i;
    i = j;
    k++;
  }
  // the slices still in flight, oldest first
  for (uint32_t q = 0; q < depth; q++) {
    const uint32_t slot = (k + q) % depth;
    if (rc == DTK_OK) rc = deliver(p, slot, fn, user);
    // (an error: nothing is handed over any more, but no copy from the caller's text may outlive the call)
    if (rc != DTK_OK && p->touched[slot]) { (void)dtk_batch_sync(p->slots[slot]); p->count[slot] = 0; p->touched[slot] = 0; }
  }
  if (registered) { (void)hipHostUnregister((void *)(text + doc_off[0])); (void)hipGetLastError(); }
  return rc;
}
