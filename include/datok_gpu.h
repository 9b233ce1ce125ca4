/*
 * datok_gpu.h -- C-ABI of the MI355X batch tokenizer (libdatok_gpu.so).
 *
 * This is the drop-in boundary for the one hot path of KorAP/Datok: the
 * matrix / double-array FSA walk behind
 *     LoadTokenizerFile            (fomafile.go:452-484)
 *     Tokenizer.Transduce          (matrix.go:340-342, datok.go:769-771)
 *     Tokenizer.TransduceTokenWriter (matrix.go:348-698, datok.go:781-1135)
 *     TokenWriter / NewTokenWriter (token_writer.go:27-33, 36-175)
 * A cgo shim binds exactly these entry points; the C++ and
 * Python host mirrors in this repo sit on the same ABI.  Plain pointers and
 * sizes only: no torch, no HIP types in any signature (a HIP stream is passed
 * as void*).
 *
 * Every compute entry point runs on the GPU.  There is no CPU fallback: if no
 * HIP device is usable the calls return DTK_E_NO_DEVICE.
 */
#ifndef DATOK_GPU_H
#define DATOK_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (negative); dtk_strerror() renders them ---- */
enum {
  DTK_OK = 0,
  DTK_E_IO = -1,        /* cannot open / read (matrix.go:215-219) */
  DTK_E_FORMAT = -2,    /* not gzip, bad magic, bad version, short file (matrix.go:253-334) */
  DTK_E_NO_DEVICE = -3, /* no usable HIP device */
  DTK_E_HIP = -4,       /* a HIP runtime call failed; dtk_last_hip_error() has the text */
  DTK_E_ARG = -5,       /* invalid argument */
  DTK_E_MODEL = -6,     /* model outside device limits (>= 2048 symbols, epsilon cycle, ids out of range),
                           or a Foma net ParseFoma rejects (fomafile.go:159-167,283-310) */
  DTK_E_CAPACITY = -7,  /* batch larger than the dtk_batch was created for */
  DTK_E_STATE = -8,     /* call order (e.g. result requested before a run) */
  DTK_E_NOMEM = -9      /* host allocation failed */
};

/* ---- Bits, token_writer.go:17-25 (same values) ---- */
enum {
  DTK_TOKENS = 1,
  DTK_SENTENCES = 2,
  DTK_TOKEN_POS = 4,
  DTK_SENTENCE_POS = 8,
  DTK_NEWLINE_AFTER_EOT = 16,
  DTK_SIMPLE = 3,
  /* dtk_batch_run only (not a reference bit): the caller wants the offset arrays only; the
   * per-token bookkeeping of dtk_batch_render_* is not written (a render then returns DTK_E_STATE) */
  DTK_OFFSETS_ONLY = 256,
  /* dtk_batch_run only, each implies DTK_OFFSETS_ONLY: a caller that reads one kind of token offsets does not pay for
   * the other (the compaction otherwise writes four 32-bit words per token where the north star's arrays have two;
   * tok_rstart / tok_rend resp. tok_bstart / tok_bend then come back NULL).  The closure replay and the device
   * renderer need the byte offsets. */
  DTK_NO_BYTE_OFFSETS = 512,
  DTK_NO_RUNE_OFFSETS = 1024
};

/* ---- per-document status bits: inputs on which the reference panics or
 *      that the device representation cannot express.  Offsets of a document
 *      with status != 0 are out of contract. ---- */
enum {
  DTK_ST_WINDOW_OVERFLOW = 1, /* > 1024 buffered runes: matrix.go:365/406 index panic */
  DTK_ST_EMPTY_TEXT = 2,      /* SentenceEnd/TextEnd with no token in the text:
                                 token_writer.go:108,135,145 panic in position modes */
  DTK_ST_BAD_MODEL = 4,       /* walk left the table */
  DTK_ST_IRREGULAR = 8,       /* internal: calls not in position order; such a document is re-walked by the
                                 exact pass (dtk_result_view.n_exact) and never reported with this bit */
  DTK_ST_STEP_LIMIT = 16,     /* safety cap on lookups hit */
  DTK_ST_INTERNAL = 32,       /* internal consistency check failed (a bug; never expected) */
  DTK_ST_BAD_OFFSET = 64      /* a Token call whose offset lies behind its buffer (after a hard fail with the
                                 epsilon slot behind the token start, e.g. "x ....\n\n" at the end of a document):
                                 string(buf[offset:]) panics in the reference when surfaces are printed
                                 (token_writer.go:85,93); in position-only modes it yields start > end */
};

typedef struct dtk_model dtk_model;
typedef struct dtk_batch dtk_batch;

/* ---- devices ---- */
int dtk_device_count(void);
int dtk_set_device(int device);          /* device used by subsequent loads / batches of this thread */
const char *dtk_strerror(int code);
const char *dtk_last_hip_error(void);

/* Test hook.  The library never reads the environment; the code paths it otherwise picks by model, batch shape or
 * history (general loop instead of the lean one, 16-bit stream entries, the double array's pairs instead of its
 * dense layout, two-launch first pass, ...) can be forced through this call so that the test suite runs over each
 * of them.  key: the name in dtk_host.cpp's table, with or without the "DATOK_" prefix the tests' environment
 * variables carry (datok_amd/_lib.py forwards those); value: an integer as text (NULL = 1).  No switch changes a
 * result.  Set before the models / batches it should affect are created. */
int dtk_debug_configure(const char *key, const char *value);

/* ---- model: replaces LoadTokenizerFile (fomafile.go:452-484), LoadMatrixFile
 *      (matrix.go:214-231), LoadDatokFile (datok.go:600-617).  gunzip, sniff
 *      "MATOK"/"DATOK", parse, build the device tables.  Immutable afterwards,
 *      shareable between batches and threads. ---- */
int dtk_model_load(const char *path, dtk_model **out);
int dtk_model_load_mem(const void *gz_bytes, size_t n, dtk_model **out); /* ParseMatrix/ParseDatok on a gzip blob */
void dtk_model_free(dtk_model *m);
/* Both loaders also accept a gzip'd Foma text net ("##foma-net ..."): LoadFomaFile + Automaton.ToMatrix
 * (fomafile.go:56-450, matrix.go:30-99), i.e. what the reference's tests build their tiny tokenizers with.
 * dtk_foma_to_matok is `datok convert` without --double-array (cmd/datok.go:50-70): the same conversion
 * followed by MatrixTokenizer.Save (matrix.go:107-210) into a gzip image (*out, free with dtk_free).
 * Host only: needs no device.
 * dtk_foma_to_datok is `datok convert --double-array`: Automaton.ToDoubleArray (datok.go:82-238, Mizobuchi et al. 2000
 * with the xCheckSkipNiu search) followed by DaTokenizer.Save (datok.go:485-596).  The reference lays the states out
 * in the order Go's map iteration hands it their symbols (fomafile.go:488-495) -- two runs give two arrays; here the
 * symbols are taken in ascending order, so the image is one the reference can produce, not a particular one. */
int dtk_foma_to_matok(const void *gz_bytes, size_t n, void **out, size_t *out_n);
int dtk_foma_to_datok(const void *gz_bytes, size_t n, void **out, size_t *out_n);
const char *dtk_model_type(const dtk_model *m); /* Tokenizer.Type(): "MATOK" / "DATOK" (matrix.go:102, datok.go:252) */

typedef struct {
  int32_t kind;          /* 0 matrix, 1 double array */
  int32_t epsilon, unknown, identity, final_state, sigma_count;
  uint32_t state_count;  /* matrix: stateCount; double array: array[1].check */
  uint64_t array_len;    /* matrix: u32 cells; double array: {base,check} pairs */
  uint32_t n_eps_states; /* states with an epsilon arc */
  uint32_t max_eps_chain;/* longest path of epsilon arcs */
  uint32_t entry_bytes;  /* device table cell size (2 or 4 matrix, 8 double array) */
  uint64_t device_bytes; /* HBM held by the model */
  uint32_t unknown_used; /* 1 if any state has an arc on the unknown symbol */
  uint32_t dense_states; /* double array only: its states, if its transitions were laid out as a matrix on the
                          * device at load (entry_bytes is then 4); 0: the {base, check} pairs are walked */
  uint32_t stream_codes; /* distinct symbol-stream entries of the model if they fit a byte (the stream is then one
                          * code per input byte and the lean loop may apply); 0: 16-bit entries, general loop */
} dtk_model_info;
int dtk_model_get_info(const dtk_model *m, dtk_model_info *out);

/* ---- batch: one data-parallel TransduceTokenWriter over n_docs documents.
 *      Each document is one reference call with a fresh writer
 *      (matrix.go:348 / datok.go:781).  A dtk_batch owns its device buffers
 *      (sized once at creation) and one HIP stream; no allocation happens in
 *      dtk_batch_run. ---- */
int dtk_batch_create(uint64_t max_bytes, uint32_t max_docs, dtk_batch **out);
void dtk_batch_free(dtk_batch *b);

/* Host input: text = concatenated documents, doc_off[n_docs+1] byte offsets
 * (doc_off[0] == 0).  Copies to the device on the batch's stream. */
int dtk_batch_set_input(dtk_batch *b, const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs);
/* Device-resident input (e.g. a torch tensor's data_ptr): no copy is made, the
 * buffers must stay valid until the run has been synchronised. */
int dtk_batch_set_input_device(dtk_batch *b, const void *d_text, const void *d_doc_off,
                               uint32_t n_docs, uint64_t total_bytes);

/* Walk strategy.  chunk_bytes = 0: one lane per document.  Otherwise documents are
 * cut into chunks of chunk_bytes; every chunk is walked by its own lane from a
 * speculative start found warm_bytes earlier, and a check pass proves that each
 * lane arrived exactly where its successor started (mismatches are repaired, the
 * result is always exact).  0xFFFFFFFF (the default) picks the chunk size from
 * the batch size.  warm_bytes defaults to 8 (the start then moves back to the previous blank, see below). */
int dtk_batch_set_chunking(dtk_batch *b, uint32_t chunk_bytes, uint32_t warm_bytes);

/* A speculative start that falls inside a long blank-free token (a URL) would invent
 * token ends; the start is therefore moved back from chunk - warm_bytes to the previous
 * blank, by at most max_bytes (default 240; 0 = keep the fixed distance, which tests use
 * to force mispredictions).  Speed only: the result is exact either way. */
int dtk_batch_set_warm_extend(dtk_batch *b, uint32_t max_bytes);

/* Launches the whole path on the batch's stream (asynchronous):
 * symbolise -> walk -> count -> scan -> compact.  flags: DTK_NEWLINE_AFTER_EOT
 * is the only bit that changes the numbers (token_writer.go:66-68); DTK_OFFSETS_ONLY
 * skips what only the device renderer needs. */
int dtk_batch_run(const dtk_model *m, dtk_batch *b, uint32_t flags);
int dtk_batch_sync(dtk_batch *b);
/* Lend the batch HIP streams instead of the one it owns: `compute` for its kernels (NULL: keep), `upload` for the
 * copies of dtk_batch_set_input (NULL: on the compute stream).  The batches of a dtk_pipeline share one of each: the
 * HIP runtime has four hardware queues, streams that share one serialise, and a pipeline of any depth then needs three
 * (uploads, kernels, downloads).  The streams must outlive the batch's use of them. */
int dtk_batch_set_streams(dtk_batch *b, void *compute, void *upload);
int dtk_batch_done(dtk_batch *b); /* 1 if everything dtk_batch_run enqueued has finished, 0 if not; never blocks */
void *dtk_batch_stream(dtk_batch *b); /* hipStream_t, for event timing by the caller */

/* Optional per-stage timing with HIP events recorded on the batch's stream
 * around each kernel of dtk_batch_run (no host synchronisation is added).
 * dtk_batch_stage_ms() synchronises and returns the milliseconds of the last
 * run: [0] clears [1] symbolise [2] start records [3] link [4] chunk walk (or
 * the one-lane-per-document walk) [5] verify [6] fix [7] scan [8] compaction. */
#define DTK_N_STAGES 9
int dtk_batch_set_profiling(dtk_batch *b, int enable);
int dtk_batch_stage_ms(dtk_batch *b, float ms[DTK_N_STAGES]);

/* Totals of the last run (synchronises). */
typedef struct {
  uint32_t n_docs;
  uint64_t n_bytes;
  uint64_t n_tokens;      /* Token calls */
  uint64_t n_sent;        /* ints in the flat sentence list (token_writer.go:78,108) */
  uint64_t n_texts;       /* TextEnd calls */
  uint64_t n_flagged;     /* documents with status != 0 */
  uint64_t walk_steps;    /* table lookups performed by the walk kernels (warm-ups included) */
  uint32_t n_lanes;       /* lanes the walk ran on */
  uint32_t chunk_bytes;   /* chunk size used (0: one lane per document) */
  uint32_t repair_rounds; /* speculation repair rounds (normally 0) */
} dtk_totals;
int dtk_batch_totals(dtk_batch *b, dtk_totals *out);

/*
 * Result arrays of the last run.  CSR over documents:
 *   tokens of doc d:  [tok_off[d], tok_off[d+1])
 *     tok_rstart/tok_rend : rune offsets relative to the current text, exactly the
 *                           pos[] pairs of token_writer.go:72-81
 *     tok_bstart/tok_bend : byte offsets relative to the document (surface slices)
 *   sentence ints of d: [sent_off[d], sent_off[d+1])  -- the flat sent[] list of
 *                           token_writer.go:75-79,104-109 (start of first token after a
 *                           sentence/text end, end of last token at a SentenceEnd)
 *   texts of d:        [text_off[d], text_off[d+1])   -- one per TextEnd:
 *     text_tok_end[i]  = tokens of the document emitted before it
 *     text_sent_end[i] = sentence ints emitted before it
 *   status[d]          = DTK_ST_* bits
 * Pointers are DEVICE pointers owned by the batch, valid until the next run.
 */
/* One call of the reference into the TokenWriter, for documents walked by the exact pass (below):
 * kind 0 = Token(offset, buf): a = byte position of buf[0], b = of buf[offset], c = end of buf
 *          (document relative; b > c only in DTK_ST_BAD_OFFSET documents);
 * kind 1 = SentenceEnd(a);  kind 2 = TextEnd(a) -- a is the int the reference passes
 *          (matrix.go:575,597,600,684,691: buffc; datok.go:1015,1026,1119,1127: 0, :1023: buffc). */
typedef struct { uint32_t kind; int32_t a; uint32_t b, c; } dtk_call;

typedef struct {
  const uint64_t *tok_off, *sent_off, *text_off; /* n_docs+1 each */
  const int32_t *tok_rstart, *tok_rend;
  const uint32_t *tok_bstart, *tok_bend;
  const int32_t *sent;
  const uint32_t *text_tok_end, *text_sent_end;
  const uint32_t *status;
  /* Raw walk output, for replay into TokenWriter closures: one bitmap per kind of event over the cursor
   * positions p = 0..len of every document.  Position p of document d is bit DTK_EVENT_BIT(doc_off[d], d) + p;
   * bitmap k (DTK_EVB_*) is ev_bits[k * ev_words .. (k + 1) * ev_words).  Calls at one cursor position are
   * fired in the order SEOT, TEOT, END (the token whose first byte is the last START bit below), SEPS.  The
   * final SentenceEnd / TextEnd (matrix.go:683-691) are in doc_tail[d] = cursor << 2 | DTK_TAIL_S | DTK_TAIL_E.
   * The k-th END bit of a document belongs to its k-th token (tok_bstart / tok_bend). */
  const uint32_t *ev_bits;
  uint64_t ev_words;
  const uint32_t *doc_tail;
  /* The exact pass.  The event bytes order the calls by cursor position.  Two constructs of the
   * reference break that order (or put two calls on one bit): the double array consuming one EOT rune twice (it keeps its window
   * over an EOT, datok.go:1019-1030, so a later backtrack, :916-926, re-reads it: SentenceEnd /
   * TextEnd, then a Token that ends BEFORE them, then both again), and more than two epsilon
   * SentenceEnds at one cursor (matrix.go:573-576 has no limit).  No shipped model does either on
   * any test corpus; a document that does is walked again by a single lane in the reference's own
   * order, which writes its rows of the arrays above (bit-exact like all others) and lists its
   * calls here.  For the n_exact documents exact_doc[i] (ascending) a replay must use
   * calls[exact_off[i] .. exact_off[i+1]) instead of the event bitmaps. */
  uint32_t n_exact;
  const uint32_t *exact_doc;
  const uint64_t *exact_off; /* n_exact + 1 */
  const dtk_call *calls;
  /* DTK_R_TOK_RUNE16 (host results only): (uint16_t)tok_rstart[i] | (uint32_t)(uint16_t)tok_rend[i] << 16, or NULL */
  const uint32_t *tok_r16;
} dtk_result_view;
int dtk_batch_result_device(dtk_batch *b, dtk_result_view *out);
/* status words of the first n documents, copied to the caller's array */
int dtk_batch_status_host(dtk_batch *b, uint32_t *status, uint32_t n);
/* The results on the host.  The reference hands every token to a host closure (matrix.go:528,569,677 -> w.Token,
 * token_writer.go:72-88); here the arrays come over in one chain of asynchronous copies into page-locked memory owned
 * by the batch, on a HIP stream of their own (the batch's stream and the upload direction of the link stay free).
 *   dtk_batch_set_result_fields: which arrays a caller wants (DTK_R_*, default all): a caller that only reads the
 *       rune offsets moves 1.1 B per input byte instead of 2.8; one that replays the events into closures needs
 *       DTK_R_EVENTS | DTK_R_TOK_BYTE | DTK_R_CSR | DTK_R_STATUS.
 *   dtk_batch_download_begin: completes the run like dtk_batch_totals, enqueues the copies and returns at once
 *       (dtk_pipeline uses it to bring slice i home under the walk of slice i + 1 and the upload of slice i + 2).
 *   dtk_batch_result_host: begins the download if nobody has, waits for it, and returns host pointers (valid until the
 *       batch's next run / free); arrays that were not selected are NULL. */
enum {
  DTK_R_CSR = 1,       /* tok_off, sent_off, text_off */
  DTK_R_TOK_RUNE = 2,  /* tok_rstart, tok_rend */
  DTK_R_TOK_BYTE = 4,  /* tok_bstart, tok_bend */
  DTK_R_SENT = 8,      /* sent */
  DTK_R_TEXTS = 16,    /* text_tok_end, text_sent_end */
  DTK_R_STATUS = 32,   /* status */
  DTK_R_EVENTS = 64,   /* ev_bits, doc_tail */
  DTK_R_ALL = 127,
  /* tok_r16: the rune offsets of a token as the two halves of one 32-bit word -- start in the low half, end in the
   * high half, both int16 -- half the bytes of tok_rstart / tok_rend on the link (the download is the longer leg of a
   * host-to-host pass: 1.2 B per input byte against 1 B of upload).  Offsets never exceed a document's length, so the
   * narrow form exists when no document of the batch is longer than 32 767 bytes; for a batch with a longer one
   * tok_r16 comes back NULL and tok_rstart / tok_rend are delivered in its place.  Not part of DTK_R_ALL. */
  DTK_R_TOK_RUNE16 = 128,
  /* with any of the above: the selected arrays leave for the host inside dtk_batch_run -- a kernel behind the
   * compaction reads the row counts on the device and streams the rows into the batch's page-locked buffers, so no
   * host round trip stands between the walk and the copy.  dtk_batch_result_host then finds them there (a run that
   * needed a repair round, larger arrays or the exact pass copies again by itself). */
  DTK_R_EAGER = 256
};
int dtk_batch_set_result_fields(dtk_batch *b, uint32_t fields);
int dtk_batch_download_begin(dtk_batch *b);
/* The HIP stream the result copies run on: the batch creates one with its first download; a caller with several
 * batches should lend them one stream between them (dtk_pipeline does): the HIP runtime maps streams onto four
 * hardware queues, and two streams on one queue serialise. */
int dtk_batch_set_download_stream(dtk_batch *b, void *hip_stream);
void *dtk_batch_download_stream(dtk_batch *b);
int dtk_batch_result_host(dtk_batch *b, dtk_result_view *out);

/* bit of position 0 of document d in the bitmaps of dtk_result_view.ev_bits */
#define DTK_EVENT_BIT(doc_off_d, d) (((uint64_t)(doc_off_d)) + (uint64_t)(d))

/* the bitmaps of dtk_result_view.ev_bits, in the order the calls fire at one cursor position (START is no call) */
enum {
  DTK_EVB_END = 0,   /* a token ends before this byte: Token(offset, buf) */
  DTK_EVB_START = 1, /* ... and this is its first byte */
  DTK_EVB_SEPS = 2,  /* SentenceEnd from an epsilon arc on an empty token (matrix.go:574-575), after an END here */
  DTK_EVB_TEOT = 3,  /* TextEnd fired by the EOT rune before this byte (matrix.go:599-600), before an END here */
  DTK_EVB_SEOT = 4,  /* SentenceEnd fired by that EOT rune (matrix.go:595-598), before its TextEnd */
  DTK_EVB_KINDS = 5
};
enum { DTK_TAIL_S = 1, DTK_TAIL_E = 2 }; /* doc_tail: final SentenceEnd (matrix.go:683-684) / TextEnd (:690-691) */

/* ---- NewTokenWriter(w, bits) (token_writer.go:36-175) for every document of the batch, rendered on
 *      the device: bytes[doc_off[d] .. doc_off[d+1]) is exactly what the reference writes to w for
 *      document d (surfaces + "\n", "\n" per sentence / text end, position lines) -- SIMPLE gives the
 *      stream of Transduce (matrix.go:340-342).  Call after dtk_batch_run; `bits` may differ from the
 *      run's flags except for DTK_NEWLINE_AFTER_EOT.  The view is owned by the batch (valid until the
 *      next run / render / free).  Documents flagged DTK_ST_EMPTY_TEXT make the reference panic in
 *      position modes; their bytes are unspecified. ---- */
typedef struct {
  const uint8_t *bytes;
  const uint64_t *doc_off; /* n_docs + 1 */
  uint64_t total;
} dtk_render_view;
int dtk_batch_render_device(dtk_batch *b, uint32_t bits, dtk_render_view *out);
int dtk_batch_render_host(dtk_batch *b, uint32_t bits, dtk_render_view *out);

/* ---- streaming: a corpus larger than one batch.  The reference streams any amount of input through a
 *      bufio.Reader (matrix.go:372); for many documents (one TransduceTokenWriter call each) the counterpart
 *      here is a pipeline of `depth` batches: the corpus is cut into slices of at most slice_bytes /
 *      slice_docs at document boundaries, slice i + 1 is uploaded (asynchronously, from page-locked host
 *      memory, on its batch's own HIP stream) while slice i is walked, and finished slices are handed to
 *      `fn` in order on the calling thread -- read their results there (dtk_batch_totals / result_* /
 *      render_*), they are valid until fn returns.  Text in memory from dtk_pinned_alloc is uploaded
 *      straight from there; other memory is page-locked for the duration of the call. ---- */
typedef struct dtk_pipeline dtk_pipeline;
typedef int (*dtk_slice_fn)(void *user, uint32_t first_doc, uint32_t n_docs, dtk_batch *slice); /* DTK_OK to go on */
int dtk_pipeline_create(uint64_t slice_bytes, uint32_t slice_docs, uint32_t depth, dtk_pipeline **out);
void dtk_pipeline_free(dtk_pipeline *p);
int dtk_pipeline_set_chunking(dtk_pipeline *p, uint32_t chunk_bytes, uint32_t warm_bytes);
/* fields != 0 (DTK_R_*): every slice's selected result arrays are brought to the host -- the copies of slice i start as
 * soon as its kernels have finished and run under the walk of slice i + 1 and the upload of slice i + 2; `fn` finds them
 * through dtk_batch_result_host.  0 (the default): results stay in HBM unless fn asks. */
int dtk_pipeline_set_result_fields(dtk_pipeline *p, uint32_t fields);
int dtk_pipeline_run(dtk_pipeline *p, const dtk_model *m, const uint8_t *text, const uint64_t *doc_off,
                     uint32_t n_docs, uint32_t flags, dtk_slice_fn fn, void *user);
/* ---- several GPUs of one node.  Documents are independent (matrix.go:349-381: all walk state is per call), so a corpus
 *      shards by slices with nothing to exchange: dtk_multi keeps one worker thread per listed device, each with its
 *      own replica of the model (loaded from model_path on that device) and its own dtk_pipeline; dtk_multi_run deals
 *      the slices of the corpus round-robin, every device works through its share, and finished slices are handed to
 *      `fn` on the calling thread in corpus order -- their results, if fields are selected, in the owning device's
 *      page-locked buffers (dtk_batch_result_host inside fn).  A device may be listed more than once.  This is the
 *      Go caller's multi-GPU entry point: fomafile.go:29-33 has no torch.distributed; the Python harness of bench.py
 *      (one process per GPU, RCCL gather) measures the same sharding from the other side. ---- */
typedef struct dtk_multi dtk_multi;
int dtk_multi_create(const char *model_path, const int *devices, uint32_t n_devices, uint64_t slice_bytes,
                     uint32_t slice_docs, uint32_t depth, dtk_multi **out);
void dtk_multi_free(dtk_multi *mp);
const char *dtk_multi_type(const dtk_multi *mp); /* Tokenizer.Type() */
int dtk_multi_set_result_fields(dtk_multi *mp, uint32_t fields);
int dtk_multi_set_chunking(dtk_multi *mp, uint32_t chunk_bytes, uint32_t warm_bytes);
int dtk_multi_run(dtk_multi *mp, const uint8_t *text, const uint64_t *doc_off, uint32_t n_docs, uint32_t flags,
                  dtk_slice_fn fn, void *user);
void *dtk_pinned_alloc(size_t n); /* page-locked host memory (hipHostMalloc); NULL on failure */
void dtk_pinned_free(void *p);

/* ---- drop-in for Tokenizer.Transduce / TransduceTokenWriter with a stock
 *      NewTokenWriter(w, bits): what the reference writes to w for ONE stream
 *      (matrix.go:340-348, token_writer.go:36-175).  dtk_transduce walks and renders on the
 *      GPU; dtk_transduce_replay walks on the GPU and replays the event bytes into the
 *      closures of the C++ mirror of NewTokenWriter (include/datok.hpp) -- the path of a custom
 *      TokenWriter.  Both give the same bytes.  *out is malloc'd; free with
 *      dtk_free.  Returns 0 (the reference's `true`) or a negative code. ---- */
int dtk_transduce(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                  char **out, size_t *out_len, uint32_t *status);
int dtk_transduce_replay(const dtk_model *m, const uint8_t *text, size_t n, uint32_t bits,
                         char **out, size_t *out_len, uint32_t *status);
/* Both keep one dtk_batch per calling thread between calls (sized for the largest input seen);
 * this frees the calling thread's. */
void dtk_transduce_release(void);
/* The walk of one stream on that per-thread batch: host pointers, valid until the thread's next
 * dtk_transduce* call -- what TransduceTokenWriter with a custom writer replays from. */
int dtk_transduce_result(const dtk_model *m, const uint8_t *text, size_t n, uint32_t flags, dtk_result_view *view);
void dtk_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
