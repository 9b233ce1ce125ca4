"""datok_amd -- MI355X-native batch finite-state tokenizer, drop-in for the hot
path of KorAP/Datok (LoadTokenizerFile / Tokenizer.TransduceTokenWriter /
TokenWriter).

Python host mirror of the reference's Go surface; the walk runs in hand-written
HIP kernels behind the C-ABI of libdatok_gpu.so (include/datok_gpu.h):

    Go (reference)                       here
    ------------------------------------ -----------------------------------------
    Bits / TOKENS ... SIMPLE             TOKENS ... SIMPLE        (token_writer.go:17-25)
    TokenWriter, NewTokenWriter(w, f)    TokenWriter, new_token_writer(w, f)  (token_writer.go:27-175)
    LoadTokenizerFile(file) Tokenizer    load_tokenizer_file(file) -> Tokenizer | None (fomafile.go:452-484)
    tok.Transduce(r, w) bool             Tokenizer.transduce(r, w)            (matrix.go:340-342)
    tok.TransduceTokenWriter(r, tw) bool Tokenizer.transduce_token_writer(r, tw) (matrix.go:348-698)
    tok.Type() string                    Tokenizer.type()                     (matrix.go:102)
    LoadFomaFile(f).ToMatrix()           load_foma_file(f)                    (fomafile.go:56-450, matrix.go:30-99)
    `datok convert` (matrix)             foma_to_matok(bytes) -> bytes        (cmd/datok.go:50-70)
    `datok convert --double-array`       foma_to_datok(bytes) -> bytes        (datok.go:82-238)
    --                                   Batch: many documents per launch (addition)
    --                                   Pipeline: a corpus larger than a batch, uploads overlapped (addition)
    --                                   MultiPipeline: ... sharded over several GPUs of a node (addition)
"""
from ._lib import (DatokGpuError, ST_BAD_MODEL, ST_BAD_OFFSET, ST_EMPTY_TEXT, ST_IRREGULAR, ST_STEP_LIMIT,  # noqa: F401
                   ST_WINDOW_OVERFLOW, build, lib)
from .host import (NEWLINE_AFTER_EOT, NO_BYTE_OFFSETS, NO_RUNE_OFFSETS, OFFSETS_ONLY, SENTENCE_POS, SENTENCES, SIMPLE, TOKEN_POS, TOKENS, Batch,  # noqa: F401
                   BatchResult, MultiPipeline, PinnedBuffer, Pipeline, TokenWriter, Tokenizer, foma_to_datok, foma_to_matok, load_foma_file,
                   load_tokenizer_file, new_token_writer, replay)
