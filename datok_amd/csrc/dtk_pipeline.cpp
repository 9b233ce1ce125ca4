// dtk_pipeline.cpp -- a corpus larger than one batch: slices cut at document boundaries run through a few
// dtk_batch objects in turn, so that the upload of one slice (pinned host memory -> HBM on that batch's own HIP
// stream) overlaps the walk of the slices before it and nothing waits for the host except the hand-over of
// finished slices, in order.  The reference streams its input through a bufio.Reader (matrix.go:372); this is the
// batch-speed counterpart for many documents (one TransduceTokenWriter call each).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
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
  // (portable: every device of a dtk_multi uploads from it)
  if (hipHostMalloc(&p, n ? n : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
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
  if (rc != DTK_OK) { (void)dtk_batch_sync(p->slots[slot]); return rc; }  // (no copy from the caller's text outlives the call)
  return fn ? fn(user, first, n, p->slots[slot]) : DTK_OK;
}

// The slices of a corpus: [first, first + n) each, as many documents as fit slice_bytes / slice_docs.
struct Slice { uint32_t first, n; };
static int cut_slices(const uint64_t *doc_off, uint32_t n_docs, uint64_t slice_bytes, uint32_t slice_docs, std::vector<Slice> &out) {
  uint32_t i = 0;
  while (i < n_docs) {
    const uint64_t lim = doc_off[i] + slice_bytes;
    uint32_t j = (uint32_t)(std::upper_bound(doc_off + i, doc_off + n_docs + 1, lim) - doc_off) - 1u;
    if (j > i + slice_docs) j = i + slice_docs;
    if (j == i) return DTK_E_CAPACITY;  // one document larger than a slice
    out.push_back(Slice{i, j - i});
    i = j;
  }
  return DTK_OK;
}

// Page-locks the corpus for the duration of a run unless it is page-locked already (dtk_pinned_alloc, or registered
// by the caller): asynchronous uploads need it (if that fails the copies are staged by the runtime: correct, but the
// upload then blocks the submitting thread).
struct HostLock {
  void *p = nullptr;
  HostLock(const uint8_t *text, uint64_t total) {
    if (!total) return;
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, text) == hipSuccess &&
                        (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeManaged);
    (void)hipGetLastError();
    if (!pinned && hipHostRegister((void *)text, total, hipHostRegisterPortable) == hipSuccess) p = (void *)text;
    (void)hipGetLastError();
  }
  ~HostLock() { if (p) { (void)hipHostUnregister(p); (void)hipGetLastError(); } }
};

// The listed slices through the pipeline's slots, in order.  `stop` (may be null): checked before every submission,
// a dtk_multi's way to end its workers when one of them has failed.
static int run_slices(dtk_pipeline *p, const dtk_model *m, const uint8_t *text, const uint64_t *doc_off, const Slice *sl,
                      size_t n_slices, uint32_t flags, dtk_slice_fn fn, void *user, const volatile int *stop) {
  const uint32_t depth = (uint32_t)p->slots.size();
  int rc = DTK_OK;
  size_t k = 0;
  for (; k < n_slices && rc == DTK_OK; k++) {
    if (stop && *stop) { rc = DTK_E_STATE; break; }
    const uint32_t i = sl[k].first, n = sl[k].n;
    const uint32_t slot = (uint32_t)(k % depth);
    if ((rc = deliver(p, slot, fn, user)) != DTK_OK) break;
    p->off.resize((size_t)n + 1);
    for (uint32_t d = 0; d <= n; d++) p->off[d] = doc_off[i + d] - doc_off[i];
    p->touched[slot] = 1;
    if ((rc = dtk_batch_set_input(p->slots[slot], text + doc_off[i], p->off.data(), n)) != DTK_OK) break;
    if ((rc = dtk_batch_run(m, p->slots[slot], flags)) != DTK_OK) break;
    p->first[slot] = i;
    p->count[slot] = n;
  }
  // the slices still in flight, oldest first
  for (uint32_t q = 0; q < depth; q++) {
    const uint32_t slot = (uint32_t)((k + q) % depth);
    if (rc == DTK_OK) rc = deliver(p, slot, fn, user);
    // (an error: nothing is handed over any more, but no copy from the caller's text may outlive the call)
    if (rc != DTK_OK && p->touched[slot]) { (void)dtk_batch_sync(p->slots[slot]); p->count[slot] = 0; p->touched[slot] = 0; }
  }
  return rc;
}

extern "C" int dtk_pipeline_run(dtk_pipeline *p, const dtk_model *m, const uint8_t *text, const uint64_t *doc_off,
                                uint32_t n_docs, uint32_t flags, dtk_slice_fn fn, void *user) {
  if (!p || !m || !doc_off || (n_docs && doc_off[n_docs] && !text)) return DTK_E_ARG;
  const uint64_t total = n_docs ? doc_off[n_docs] - doc_off[0] : 0;
  std::vector<Slice> sl;
  int rc = cut_slices(doc_off, n_docs, p->slice_bytes, p->slice_docs, sl);
  if (rc != DTK_OK) return rc;
  HostLock lock(total ? text + doc_off[0] : nullptr, total);
  return run_slices(p, m, text, doc_off, sl.data(), sl.size(), flags, fn, user, nullptr);
}

// -------------------------------------------------------------------------------------------------- several devices
//
// dtk_multi: the documents of a corpus sharded over the GPUs of a node behind the C-ABI (the reference's caller,
// fomafile.go:29-33, has no torch.distributed to do it for him).  Documents are independent (matrix.go:349-381: all
// walk state is per call), so there is nothing to exchange: one worker thread per listed device with its own replica
// of the model and its own dtk_pipeline; the slices of the corpus are dealt round-robin; every worker runs its slices
// through its pipeline and hands each finished slice to the calling thread, which calls `fn` in corpus order -- from
// the owning device's page-locked buffers if result fields are selected.  A device may be listed more than once
// (two workers on one GPU: how the one-GPU test box exercises this).
struct dtk_multi {
  struct Worker {
    int device = 0;
    dtk_model *model = nullptr;
    dtk_pipeline *pipe = nullptr;
    std::thread th;
    std::vector<Slice> mine;      // this run's slices of this worker, in corpus order
    // hand-over of one finished slice to the calling thread
    size_t ready_k = 0;           // mine[ready_k - 1] is on offer (0: none yet)
    dtk_batch *ready_batch = nullptr;
    size_t taken_k = 0;           // the calling thread is done with mine[taken_k - 1]
    int rc = DTK_OK;
    bool finished = false;
  };
  std::vector<Worker> w;
  std::mutex mu;
  std::condition_variable cv;
  // the run in progress
  const uint8_t *text = nullptr;
  const uint64_t *doc_off = nullptr;
  uint32_t flags = 0;
  volatile int stop = 0;
  uint64_t run_no = 0;            // workers wait for the next run (or for quit)
  bool quit = false;
};

namespace {
struct WorkerCtx { dtk_multi *mp; dtk_multi::Worker *me; size_t k; };

// runs on the worker's thread for every finished slice: offer it, wait until the calling thread has delivered it
int worker_slice(void *user, uint32_t, uint32_t, dtk_batch *b) {
  WorkerCtx *c = (WorkerCtx *)user;
  std::unique_lock<std::mutex> lk(c->mp->mu);
  c->me->ready_batch = b;
  c->me->ready_k = ++c->k;
  c->mp->cv.notify_all();
  c->mp->cv.wait(lk, [&] { return c->me->taken_k >= c->k || c->mp->stop; });
  return c->mp->stop ? DTK_E_STATE : DTK_OK;
}

void worker_main(dtk_multi *mp, size_t wi) {
  dtk_multi::Worker &me = mp->w[wi];
  (void)hipSetDevice(me.device);
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(mp->mu);
      mp->cv.wait(lk, [&] { return mp->quit || mp->run_no != seen; });
      if (mp->quit) return;
      seen = mp->run_no;
    }
    WorkerCtx ctx{mp, &me, 0};
    const int rc = run_slices(me.pipe, me.model, mp->text, mp->doc_off, me.mine.data(), me.mine.size(), mp->flags,
                              worker_slice, &ctx, &mp->stop);
    std::unique_lock<std::mutex> lk(mp->mu);
    me.rc = rc;
    me.finished = true;
    if (rc != DTK_OK) mp->stop = 1;
    mp->cv.notify_all();
  }
}
}  // namespace

extern "C" void dtk_multi_free(dtk_multi *mp) {
  if (!mp) return;
  {
    std::unique_lock<std::mutex> lk(mp->mu);
    mp->quit = true;
    mp->stop = 1;
    mp->cv.notify_all();
  }
  for (auto &wk : mp->w)
    if (wk.th.joinable()) wk.th.join();
  for (auto &wk : mp->w) {
    (void)hipSetDevice(wk.device);
    dtk_pipeline_free(wk.pipe);
    dtk_model_free(wk.model);
  }
  delete mp;
}

extern "C" int dtk_multi_create(const char *model_path, const int *devices, uint32_t n_devices, uint64_t slice_bytes,
                                uint32_t slice_docs, uint32_t depth, dtk_multi **out) {
  if (!out || !model_path || !devices || n_devices == 0 || n_devices > 64) return DTK_E_ARG;
  *out = nullptr;
  const int have = dtk_device_count();
  if (have <= 0) return DTK_E_NO_DEVICE;
  int prev = 0;
  (void)hipGetDevice(&prev);
  dtk_multi *mp = new dtk_multi();
  mp->w.resize(n_devices);
  int rc = DTK_OK;
  for (uint32_t i = 0; i < n_devices && rc == DTK_OK; i++) {
    dtk_multi::Worker &wk = mp->w[i];
    wk.device = devices[i];
    if (wk.device < 0 || wk.device >= have) { rc = DTK_E_ARG; break; }
    if ((rc = dtk_set_device(wk.device)) != DTK_OK) break;
    if ((rc = dtk_model_load(model_path, &wk.model)) != DTK_OK) break;
    rc = dtk_pipeline_create(slice_bytes, slice_docs, depth, &wk.pipe);
  }
  (void)hipSetDevice(prev);
  if (rc != DTK_OK) { dtk_multi_free(mp); return rc; }
  for (uint32_t i = 0; i < n_devices; i++) mp->w[i].th = std::thread(worker_main, mp, (size_t)i);
  *out = mp;
  return DTK_OK;
}

extern "C" const char *dtk_multi_type(const dtk_multi *mp) { return mp && !mp->w.empty() ? dtk_model_type(mp->w[0].model) : ""; }

extern "C" int dtk_multi_set_result_fields(dtk_multi *mp, uint32_t fields) {
  if (!mp) return DTK_E_ARG;
  for (auto &wk : mp->w) {
    const int rc = dtk_pipeline_set_result_fields(wk.pipe, fields);
    if (rc != DTK_OK) return rc;
  }
  return DTK_OK;
}

extern "C" int dtk_multi_set_chunking(dtk_multi *mp, uint32_t chunk_bytes, uint32_t warm_bytes) {
  if (!mp) return DTK_E_ARG;
  for (auto &wk : mp->w) {
    const int rc = dtk_pipeline_set_chunking(wk.pipe, chunk_bytes, warm_bytes);
    if (rc != DTK_OK) return rc;
  }
  return DTK_OK;
}

extern "C" int dtk_multi_run(dtk_multi *mp, const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs, uint32_t flags,
                             dtk_slice_fn fn, void *user) {
  if (!mp || !doc_off || (n_docs && doc_off[n_docs] && !text)) return DTK_E_ARG;
  const size_t nw = mp->w.size();
  std::vector<Slice> sl;
  // (all workers' pipelines were created alike)
  int rc = cut_slices(doc_off, n_docs, mp->w[0].pipe->slice_bytes, mp->w[0].pipe->slice_docs, sl);
  if (rc != DTK_OK) return rc;
  const uint64_t total = n_docs ? doc_off[n_docs] - doc_off[0] : 0;
  HostLock lock(total ? text + doc_off[0] : nullptr, total);
  {
    std::unique_lock<std::mutex> lk(mp->mu);
    for (size_t i = 0; i < nw; i++) {
      auto &wk = mp->w[i];
      wk.mine.clear();
      wk.ready_k = wk.taken_k = 0; wk.ready_batch = nullptr; wk.rc = DTK_OK; wk.finished = false;
    }
    for (size_t j = 0; j < sl.size(); j++) mp->w[j % nw].mine.push_back(sl[j]);  // slices dealt round-robin
    mp->text = text; mp->doc_off = doc_off; mp->flags = flags; mp->stop = 0;
    mp->run_no++;
    mp->cv.notify_all();
  }
  // finished slices in corpus order: slice j is the (j / nw + 1)-th of worker j % nw
  for (size_t j = 0; j < sl.size() && rc == DTK_OK; j++) {
    auto &wk = mp->w[j % nw];
    const size_t k = j / nw + 1;
    dtk_batch *b = nullptr;
    {
      std::unique_lock<std::mutex> lk(mp->mu);
      mp->cv.wait(lk, [&] { return wk.ready_k >= k || wk.finished || mp->stop; });
      if (wk.ready_k >= k) b = wk.ready_batch; else rc = wk.rc != DTK_OK ? wk.rc : DTK_E_STATE;
    }
    if (rc == DTK_OK && fn) rc = fn(user, sl[j].first, sl[j].n, b);
    std::unique_lock<std::mutex> lk(mp->mu);
    if (rc != DTK_OK) mp->stop = 1;
    wk.taken_k = k;
    mp->cv.notify_all();
  }
  // all workers back at their wait (their slots are synchronised by run_slices before it returns)
  std::unique_lock<std::mutex> lk(mp->mu);
  mp->cv.wait(lk, [&] { return std::all_of(mp->w.begin(), mp->w.end(), [](const dtk_multi::Worker &x) { return x.finished; }); });
  for (auto &wk : mp->w)
    if (rc == DTK_OK && wk.rc != DTK_OK) rc = wk.rc;
  return rc;
}
