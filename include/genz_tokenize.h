/* genz_tokenize.h -- C ABI of the MI355X-native genz-tokenize hot path.
 *
 * This is the drop-in boundary underneath the reference's Python surface.  The
 * reference (DVNghiem/genz-tokenize v1.2.7) is pure Python and has no FFI of its
 * own, so every entry point below names the reference function it replaces
 * (paths relative to the reference checkout, genz_tokenize/tokenize.py).
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes in;
 *   - every function returns GZ_OK (0) or a negative GZ_E_* code; no C++
 *     exception crosses this boundary; gz_last_error() gives the message;
 *   - a gz_ctx is bound to ONE GPU (one process per GPU is the multi-GPU model)
 *     and is not re-entrant; distinct contexts are independent;
 *   - there is NO CPU fallback: without a usable gfx950 device gz_create fails.
 *
 * Text is packed UTF-8: document i is text[text_off[i] .. text_off[i+1]).
 * Python's `None` inside sequence_id / token_type_ids is carried as GZ_NONE.
 */
#ifndef GENZ_TOKENIZE_H
#define GENZ_TOKENIZE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GZ_VERSION 0x010100   /* 1.1: gz_expand_block takes the block's entry total (the block layout itself is round 5's) */

#define GZ_OK            0
#define GZ_E_INVALID    -1   /* bad argument */
#define GZ_E_UTF8       -2   /* table file is not valid UTF-8 (reference: UnicodeDecodeError at tokenize.py:45/54) */
#define GZ_E_HIP        -3   /* HIP runtime error */
#define GZ_E_NOTABLES   -4   /* gz_load_tables has not succeeded on this context */
#define GZ_E_CAPACITY   -5   /* ragged output does not fit `capacity`; row_off[n_docs] holds the size needed */
#define GZ_E_LIMIT      -6   /* table exceeds 2^20-2 symbol ids (every merge line takes one: merge lines + other symbols) */
#define GZ_E_NOMEM      -7
#define GZ_E_RCCL       -8
#define GZ_E_NODEVICE   -9   /* no gfx950 device / HIP runtime unusable */

/* flags of gz_encode_batch*: the keyword arguments of Tokenize.__call__ (tokenize.py:184-190) */
#define GZ_PADDING       0x1u   /* padding=True                                   */
#define GZ_TRUNCATION    0x2u   /* truncation=True                                */
#define GZ_MAX_LEN_NONE  0x4u   /* max_len=None (the max_len argument is ignored) */
#define GZ_TIMING        0x100u /* record HIP events around the kernels (gz_timing) */
#define GZ_NO_WORD_TABLE 0x200u /* do not consult the whole-word table: every word runs the merge loop (same results) */
#define GZ_KEEP_WORDS    0x400u /* keep the per-word records of this call for gz_word_token_counts (return_offset=True): small
                                   batches otherwise run in ONE fused launch that keeps nothing per word */

#define GZ_NONE (-1)            /* Python None in sequence_id / token_type_ids */

typedef struct gz_ctx gz_ctx;

int  gz_version(void);

/* Tokenize.__init__ part 1 (tokenize.py:7-37): create a context on HIP device `device_id`. */
int  gz_create(int device_id, gz_ctx **out);
void gz_destroy(gz_ctx *ctx);
const char *gz_last_error(gz_ctx *ctx);   /* ctx may be NULL: error of the last failed gz_create on this thread */

/* add_vocab_file + add_bpe_file (tokenize.py:44-57) on RAW FILE BYTES, followed by the build of the
 * device-side tables (pair -> rank hash, rank -> merged symbol, symbol -> vocab ids, code point -> symbol).
 * specials_utf8 = { pad, bos, eos, mask, unk } as NUL-terminated UTF-8 (tokenize.py:7-12, :31-37).
 * Loader quirks L1-L8 of SURVEY.md are reproduced exactly. */
int  gz_load_tables(gz_ctx *ctx, const uint8_t *vocab, size_t vocab_len,
                    const uint8_t *bpe, size_t bpe_len, const char *const specials_utf8[5]);

/* The table cache behind gz_load_tables (reference: Tokenize.__init__ re-parses both files on every construction,
 * tokenize.py:39-42, and fromFile does so twice, :261-267).  The finished table images are kept in
 * $GZ_TABLE_CACHE (default ~/.cache/genz_tokenize_amd, created 0700; "0" / "off" disables) under the SHA-256 of the file
 * bytes, the specials, the file-format version and the BUILD IDENTITY of this library (a hash of its sources: images made by
 * another build are never loaded); a file whose key, length, checksum or section sizes do not fit, or that holds an index
 * outside the tables it indexes, is refused and the tables are rebuilt; a directory other users can write to is not used.
 * gz_table_cache_status: what the last gz_load_tables of this context did -- 0 no cache, 1 hit, 2 miss (built and written),
 * 3 a cache file was there and was refused (rebuilt and rewritten), 4 built, but the file could not be written.
 * gz_table_digest: SHA-256 over every device table image + the host-visible dictionaries (equal for built and cached loads). */
int  gz_table_cache_status(gz_ctx *ctx);
int  gz_table_digest(gz_ctx *ctx, uint8_t out[32]);

/* vocab_size() (tokenize.py:59-60), encoder[pad|bos|eos|mask|unk] looked up at call time
 * (tokenize.py:134,143,145,151,164-165), len(bpe_ranks), number of interned symbols. */
int  gz_table_info(gz_ctx *ctx, int32_t *vocab_size, int32_t special_ids[5],
                   int32_t *n_ranks, int32_t *n_symbols);

/* Enumerate the encoder dict built by gz_load_tables in insertion order (so the Python shim can expose
 * `encoder` / `decoder`, tokenize.py:31-40): entry i -> (utf8 pointer valid until the next gz_load_tables,
 * byte length, id).  Returns GZ_E_INVALID when i is out of range. */
int  gz_vocab_entry(gz_ctx *ctx, int64_t i, const uint8_t **utf8, int32_t *len, int32_t *id);
/* Same for bpe_ranks (tokenize.py:53-57): key i of the dict in insertion order.  The tuple's fields are
 * returned joined by single '\n' bytes (fields never contain whitespace); n_fields may be 0. */
int  gz_merge_entry(gz_ctx *ctx, int64_t i, const uint8_t **utf8, int32_t *len, int32_t *n_fields, int32_t *rank);

/* Tokenize.__call__ (tokenize.py:184-259) over a batch of documents, host buffers in, host buffers out.
 *
 *   text/text_off          packed UTF-8 and n_docs+1 byte offsets
 *   pair/pair_off          NULL,NULL -> pair_text=None for every document; otherwise every document is a pair
 *   max_len, flags         keyword arguments (see GZ_* flags)
 *
 * Output layout.  With GZ_PADDING|GZ_TRUNCATION, no GZ_MAX_LEN_NONE and max_len >= 1 every row has exactly
 * max_len entries ("dense": row i starts at i*max_len, capacity must be >= n_docs*max_len).  Otherwise rows are
 * ragged: row i occupies [row_off[i], row_off[i+1]) of each flat array; if row_off[n_docs] > capacity nothing is
 * written to the flat arrays and GZ_E_CAPACITY is returned with row_off filled in.
 *
 *   input_ids, attention_mask       always written (tokenize.py:250-251)
 *   token_type_ids, sequence_id     pair mode only (may be NULL otherwise); both start at row_off[i] like the
 *                                   input_ids row and never exceed it: sequence_id row i holds pair_len[2i]
 *                                   entries, token_type_ids row i holds pair_len[2i+1] entries (tokenize.py:252-258;
 *                                   values 0 / 1 / GZ_NONE / pad id / eos id, rules P1-P6)
 *   row_off  [n_docs+1]             may be NULL in dense mode
 *   pair_len [2*n_docs]             pair mode; may be NULL otherwise
 *   n_real   [n_docs]               entries of the row that are not padding added by rule D1 (may be NULL)
 *   status   [n_docs]               0 = ok, 1 = the reference raises ValueError("None is not in list") for
 *                                   this document (tokenize.py:157-160, rule P3); may be NULL in single mode
 */
int  gz_encode_batch(gz_ctx *ctx,
                     const uint8_t *text, const int64_t *text_off,
                     const uint8_t *pair, const int64_t *pair_off,
                     int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                     int32_t *input_ids, int32_t *attention_mask,
                     int32_t *token_type_ids, int32_t *sequence_id,
                     int64_t *row_off, int32_t *pair_len, int32_t *n_real, int32_t *status);

/* The same call with every pointer a DEVICE pointer on the context's GPU (inputs already resident in HBM,
 * outputs left in HBM).  Work is enqueued on the context's stream; gz_sync waits for it and returns the
 * deferred error of the enqueued work, if any.  Only the dense layout and the ragged layout with a
 * sufficient capacity are available here (GZ_E_CAPACITY is reported by gz_sync).
 * The input buffers must be COMPLETE when the call is made (whatever wrote them has been synchronised): the library reads
 * them on its own streams, which know nothing of the caller's, and part of that reading starts at once. */
int  gz_encode_batch_device(gz_ctx *ctx,
                            const uint8_t *text, const int64_t *text_off,
                            const uint8_t *pair, const int64_t *pair_off,
                            int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                            int32_t *input_ids, int32_t *attention_mask,
                            int32_t *token_type_ids, int32_t *sequence_id,
                            int64_t *row_off, int32_t *pair_len, int32_t *n_real, int32_t *status);
/* The same when the caller also has the offsets on the host (it usually built them there): the library then knows the
 * byte sizes of the batch without reading them back from the device, i.e. without a host round trip before the
 * kernels are enqueued. */
int  gz_encode_batch_device_h(gz_ctx *ctx, const uint8_t *text, const int64_t *text_off, const uint8_t *pair,
                              const int64_t *pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                              int32_t *input_ids, int32_t *attention_mask, int32_t *token_type_ids, int32_t *sequence_id,
                              int64_t *row_off, int32_t *pair_len, int32_t *n_real, int32_t *status,
                              const int64_t *text_off_host, const int64_t *pair_off_host);
int  gz_sync(gz_ctx *ctx);

/* Token count of every word of the LAST encode call (for `return_offset=True`, tokenize.py:105,111-117,225-244):
 * which_text 0 = text, 1 = pair text.  counts[w] = pieces of word w (words of all documents, in order),
 * doc_first[d] = index of document d's first word (n_docs+1 entries).  Returns GZ_E_CAPACITY (and *n_words)
 * when the batch has more than `capacity` words.  Valid until the next encode call on the context.  The encode call
 * must have been made with GZ_KEEP_WORDS (GZ_E_INVALID otherwise). */
int  gz_word_token_counts(gz_ctx *ctx, int which_text, int32_t *counts, int64_t capacity,
                          int64_t *doc_first, int64_t *n_words);

/* Tokenize.bpe(token) (tokenize.py:62-101) for one word: the pieces as interned symbol ids.
 * pieces[k] >= 0 is a symbol id (string via gz_symbol_utf8); pieces[k] < 0 is -(code point)-1 for a code
 * point that occurs in no merge and no vocab entry.  Returns the number of pieces or a GZ_E_* code
 * (GZ_E_CAPACITY if more than `cap`).  The last piece still carries the "</w>" marker. */
int64_t gz_bpe_word(gz_ctx *ctx, const uint8_t *word_utf8, int64_t len, int32_t *pieces, int64_t cap);
int  gz_symbol_utf8(gz_ctx *ctx, int32_t symbol, const uint8_t **utf8, int32_t *len);

/* Host path with the copies overlapped (SURVEY.md 8(d) timing (ii); replaces a loop of Tokenize.__call__ with
 * max_len, padding=True, truncation=True over host strings, tokenize.py:184-259): the result comes back in CSR form --
 * n_real[d] = entries of row d after truncation (tokenize.py:141-146), and the rows' real entries back to back in
 * `tokens` (uint16 when bits == 16 and every id of the vocabulary fits, else int32; capacity in entries).  The dense
 * [N, max_len] input_ids / attention_mask are `row padded with the pad id` and `1 for the first n_real[d] positions`
 * unless a real token equals the pad id (then mask = ids != pad, :148-152): a caller rebuilds them only where needed.
 * The batch is cut into sub-batches: text H2D, kernels and the D2H of the compact rows run on three streams.
 * `text` / `tokens` / `n_real` from gz_host_alloc (pinned) make both copies true DMA; pageable memory works, slower.
 * *total = entries written (or needed: GZ_E_CAPACITY).  Single texts only.
 * gz_host_free does not use the context (a pinned block may outlive it): ctx may be NULL there. */
int  gz_host_alloc(gz_ctx *ctx, size_t bytes, void **ptr);
int  gz_host_free(gz_ctx *ctx, void *ptr);
int  gz_encode_batch_csr(gz_ctx *ctx, const uint8_t *text, const int64_t *text_off, int64_t n_docs, int32_t max_len,
                         uint32_t flags, void *tokens, int64_t capacity, int32_t bits, int32_t *n_real, int64_t *total);

/* Device memory helpers so that a Python host needs nothing but ctypes (no torch in the product path). */
int  gz_device_alloc(gz_ctx *ctx, size_t bytes, void **dptr);
int  gz_device_free(gz_ctx *ctx, void *dptr);
int  gz_memcpy_h2d(gz_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int  gz_memcpy_d2h(gz_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);

/* With GZ_TIMING: milliseconds (HIP events on the context's stream) of the kernels of the LAST encode call:
 * out[0] = the kernel pipeline of the call (split: brk / classify / scan / docw0; words; misses: scan / miss / miss_wide /
 * long; rows: rows1 | rows | assemble | rowsr's count pass), out[1] = ragged layouts: row lengths, scan, and the copy out of the raw
 * area or -- single texts without padding -- the rows written at their places, out[2] = pair type-id
 * kernel, out[3] = whole call on the stream.  Unused slots are 0. */
int  gz_timing(gz_ctx *ctx, double out_ms[4]);
/* GZ_TIMING calls can be chained without gz_sync in between (a dense call followed by a call that brings host copies
 * of its offsets is enqueued right behind it).  gz_timing_history synchronises and returns the duration of the main
 * kernels of the last (up to 1024, up to `max`) timed calls, oldest first, then forgets them. */
int  gz_timing_history(gz_ctx *ctx, double *out_ms, int32_t max, int32_t *n_out);

/* Offline / diagnostic table build on the HOST only (no GPU needed): the same builder gz_load_tables runs, with
 * the integer tables it would upload exposed read-only.  `which` (0 is retired -- the linear-probing pair table of rounds 1-3 -- and answers GZ_E_INVALID like any unknown value, with *data = NULL and *count = 0): 1 merges
 * (uint32 x4 [n_lines]: left,right,merged,0), 2 symbol ids (int32 x2 [n_symbols]: non-final, final), 3 BMP code
 * point table (uint32 x2 [65536]: plain, final), 4 astral table (uint32 x4 [slots]: cp,plain,final,0; may be
 * empty), 5 special ids (int32 [5]); the perfectly hashed pair table of the big pipeline's merge kernel: 6 entries (uint32 x2
 * [slots]: left | right << 20, right >> 12 | alias flag << 8 | rank << 9; 0xFFFFFFFF x2: empty), 7 displacement array (uint16
 * [buckets]), 8 its description (uint32 [7]: buckets, bucket shift, slot shift, slots, the two seeded multipliers, keys in
 * overflow buckets), 9 the hot set staged in LDS (uint32 x2 [4096], same entry form).  Pointers stay valid until
 * gz_host_tables_destroy (8: until the next call of this function on this thread). */
typedef struct gz_host_tables gz_host_tables;
int  gz_host_tables_create(const uint8_t *vocab, size_t vocab_len, const uint8_t *bpe, size_t bpe_len,
                           const char *const specials_utf8[5], gz_host_tables **out);
void gz_host_tables_destroy(gz_host_tables *t);
int  gz_host_tables_array(gz_host_tables *t, int which, const void **data, int64_t *count);
int  gz_host_tables_vocab_entry(gz_host_tables *t, int64_t i, const uint8_t **utf8, int32_t *len, int32_t *id);
int  gz_host_tables_merge_entry(gz_host_tables *t, int64_t i, const uint8_t **utf8, int32_t *len, int32_t *n_fields, int32_t *rank);
int  gz_host_tables_symbol(gz_host_tables *t, int32_t symbol, const uint8_t **utf8, int32_t *len);

/* Byte counts the library's 32-bit device paths take (no context, no GPU needed).  which = 0: bytes of ONE text (text, or pair
 * text) of an encode call from which the call answers GZ_E_LIMIT -- split the batch; 1: input bytes from which the text pre-pass
 * (gz_preprocess_batch[_device], which has no size limit of its own) leaves its 32-bit length scan for the 64-bit one.  Both are
 * below 2^32.  Unknown `which`: -1. */
int64_t gz_limit(int which);

/* Switches for tests and experiments: typed, named, range-checked -- the library reads NONE of them from the environment (its
 * documented environment is GZ_TABLE_CACHE, GZ_LOAD_TIMING, and for the Python package GZ_LIBRARY and GZ_PACK_THREADS).
 * ctx == NULL sets the process-wide defaults, which every context created afterwards copies and which the table builder reads
 * (the "builder" keys act only there); ctx != NULL sets one context.  The defaults are the product's behaviour.  Keys
 * (value range; default):
 *   small (0..1; 1)            small batches in one launch; 0: everything through the kernel pipeline
 *   small_wgs (1..2^20; 768)   workgroups the one-launch kernel aims for
 *   host_direct (0..2^20; 4096) host calls whose inputs and outputs both fit this many bytes are computed straight on the pinned staging
 *                              block (no copy in, no copy back); 0: never
 *   assemble (1..3; 3)         row writer of dense single texts: 3 rows1, 2 the pair-mode kernel, 1 the ragged layouts' scatter kernel;
 *                              single texts without padding: 3 counted, scanned, written once, < 3 through the raw area
 *   word_table (0..1; 1)       0: every word through the merge loop (as GZ_NO_WORD_TABLE on every call)
 *   pp_fused (0..1; 1)         0: the text pre-pass filter by filter for every document
 *   sub_batches (1..8; 1)      dense batches cut into document ranges on two streams
 *   docs_per_wave (0..16; 0)   documents per wave of the ragged row writer (0: by the batch's shape)
 *   side, brk_side (0..1; 1)   0: no side stream / control words prepared on the main stream
 *   scan_multi (>= 0; 8192)    block counts from which the chained multi-workgroup scan runs (0: always)
 *   near_limit (0..2^25; 2^25) token places below this get near records
 *   hot_wgs, hot_miss_wgs (0..65536; 0)   grids of the word / merge kernels (0: as many workgroups as the chip holds)
 *   m2_split_min (>= 0; 65536), m2_split_always (0..1; 0)   when the merge kernel's two instances share a launch
 *   builder: tab_slack (2..64; 16), ph_force_overflow (>= 0; 0), ph_hot_slots (0..8192; 1024), word_weights (0..2; 0)
 *   host_threads (0..256; 0)   worker threads of a large host call (0: by the processors this process may use, at most 32)
 *   dense_csr (0..1; 1)        a large dense single-text gz_encode_batch brings only the rows' real entries over the bus and pads them
 *                              into the caller's arrays on the host; 0: the dense rows cross
 *   host_hints (0..3; 0)       fresh output arrays of a large host call: bit 0 MADV_HUGEPAGE on them, bit 1 MADV_POPULATE_WRITE per piece
 *   inject_bad_alloc (>= 0; 0) test hook: the k-th allocation site reached from now on throws std::bad_alloc (the call answers GZ_E_NOMEM)
 *   diagnostic build only: diag_poison (0..1), rows_dpw, rows_dbg, ablate, diag_guard (0..2: every device buffer its own mapping
 *                              between unmapped granules, no slack -- 1 the buffer ends at its mapping's last byte, 2 it starts at the first),
 *                              diag_exact (0..1: hipMalloc of exactly the bytes asked for), diag_fresh (0..256: v > 0 fills every fresh
 *                              allocation with byte v - 1), diag_fresh_only (-1 | k: ... only the k-th allocation of the context)
 * Returns GZ_OK, or GZ_E_INVALID for an unknown key or a value out of range (nothing is changed then). */
int  gz_debug_set(gz_ctx *ctx, const char *key, int64_t value);

/* ---- batch decode ------------------------------------------------------------------------
 * gz_decoder_snapshot: build the id -> word map from the tables loaded so far, the way the reference builds
 *   `decoder` ONCE in __init__ (tokenize.py:40: {v: k for k, v in encoder.items()} -- on an id collision the last
 *   word wins).  Later gz_load_tables calls do not change the snapshot.
 * gz_decode_batch: Tokenize.decode (tokenize.py:137-139) + __convert_token_to_string (:123-124) for n_rows id lists:
 *     ' '.join(decoder.get(i, unk) for i in row).replace('@@ ', '')
 *   ids      int32, rows packed back to back; row r = ids[row_off[r] .. row_off[r+1])
 *   unk      the caller's unk_token string (the reference reads self.unk_token at call time)
 *   out      UTF-8 bytes of all rows back to back; out_off[n_rows+1] their byte offsets
 *   Returns GZ_E_CAPACITY when `capacity` is too small; out_off is valid then (out_off[n_rows] = bytes needed).
 * gz_decode_batch_device: the same with ids / row_off / out / out_off resident in HBM (total_host = bytes needed or
 *   written); out_dev may be NULL to size the output only.  As in every *_device entry point the offsets are
 *   ABSOLUTE from the base pointer: row r = ids_dev[row_off_dev[r] .. row_off_dev[r+1]), row_off_dev[0] need not be 0. */
int  gz_decoder_snapshot(gz_ctx *ctx);
int  gz_decode_batch(gz_ctx *ctx, const int32_t *ids, const int64_t *row_off, int64_t n_rows, const uint8_t *unk, int32_t unk_len,
                     uint8_t *out, int64_t capacity, int64_t *out_off);
int  gz_decode_batch_device(gz_ctx *ctx, const int32_t *ids_dev, const int64_t *row_off_dev, int64_t n_rows, const uint8_t *unk,
                            int32_t unk_len, uint8_t *out_dev, int64_t capacity, int64_t *out_off_dev, int64_t *total_host);

/* ---- text pre-pass: the string filters of genz_tokenize/preprocess.py on packed documents ----
 *   GZ_PP_HTML     remove_html          preprocess.py:5-9     re.sub(r'<[^>]*>', '', txt)
 *   GZ_PP_UNICODE  convert_unicode      preprocess.py:16-36   base letter + combining tone mark -> precomposed letter
 *   GZ_PP_PUNCT    remove_punctuations  preprocess.py:39-44   drop string.punctuation
 *   GZ_PP_EMOJI    remove_emoji         preprocess.py:47-72   drop the listed ranges, then ' '.join(s.split())
 *   GZ_PP_URL      remove_URL           preprocess.py:75-80   re.sub(r'http\S+', '', txt)
 * `ops[n_ops]` are applied in order to every document (vncore_tokenize, preprocess.py:83-89, talks to an external Java
 * server and is not part of this library).  Input and output use the packed layout of gz_encode_batch (the output can
 * be handed straight to it).  No filter grows a document, so capacity = input bytes always suffices; GZ_E_CAPACITY
 * is returned otherwise and out_off is still valid.  No tables are needed.
 * gz_preprocess_batch_device: the same with text / offsets / outputs resident in HBM; text_bytes = bytes of the input
 * text = text_off_dev[n_docs] - text_off_dev[0]; *total_host = bytes written (or needed); out_dev may be NULL to size the
 * output only.  Offsets are ABSOLUTE from the base pointer (document d = text_dev[text_off_dev[d] ..), exactly like
 * gz_encode_batch_device; text_off_dev[0] need not be 0.  The output offsets start at 0. */
#define GZ_PP_HTML    1
#define GZ_PP_UNICODE 2
#define GZ_PP_PUNCT   3
#define GZ_PP_EMOJI   4
#define GZ_PP_URL     5
int  gz_preprocess_batch(gz_ctx *ctx, const int32_t *ops, int32_t n_ops, const uint8_t *text, const int64_t *text_off,
                         int64_t n_docs, uint8_t *out, int64_t capacity, int64_t *out_off);
int  gz_preprocess_batch_device(gz_ctx *ctx, const int32_t *ops, int32_t n_ops, const uint8_t *text_dev, const int64_t *text_off_dev,
                                int64_t n_docs, int64_t text_bytes, uint8_t *out_dev, int64_t capacity, int64_t *out_off_dev,
                                int64_t *total_host);

/* ---- model-feed hand-off --------------------------------------------------------------------
 * Zero-copy export of an output buffer to a framework through DLPack.  gz_block_create takes ownership of a pointer
 * obtained from gz_device_alloc (refcount 1); gz_block_dlpack returns a malloc'ed DLManagedTensor (dlpack.h layout,
 * device type kDLROCM) that holds one more reference and whose deleter is a C function of this library, safe to call
 * at interpreter shutdown; the HBM allocation is freed when the last reference goes.  The fields a consumer sees
 * are the reference's DataCollection names (models/bert/dataset.py:7-28): input_ids, attention_mask, token_type_ids. */
typedef struct gz_block gz_block;
int   gz_block_create(gz_ctx *ctx, void *dptr, gz_block **out);
void  gz_block_release(gz_block *block);
void *gz_block_dlpack(gz_block *block, int32_t ndim, const int64_t *shape, int32_t dtype_code, int32_t dtype_bits);
/* Destructor for a PyCapsule named "dltensor" that wraps gz_block_dlpack's result (pass its address to PyCapsule_New):
 * a capsule nobody consumed gives its reference to the block back; a consumed one ("used_dltensor") is left alone.
 * (No reference counterpart: models/bert/dataset.py:7-28 builds tf tensors from Python lists.) */
void gz_dlpack_capsule_destructor(void *capsule);

/* Multi-GPU exchange step (one process per GPU, RCCL over xGMI).  rank 0 creates an id, every rank calls
 * gz_comm_init with it; gz_gather_rows sends each rank's [n_rows, row_len] int32 device block to `root`,
 * which receives them back to back in rank order (grouped ncclSend/ncclRecv: each peer uses its own link). */
int  gz_comm_unique_id(uint8_t id_out[128]);
/* The exchange operations (gz_compact_rows, gz_gather_rows, gz_expand_rows) run on a stream of their own: they start
 * when the encode call they belong to has finished and overlap the kernels of LATER encode calls; an encode call in
 * turn waits for every exchange operation issued before it (it may overwrite the buffers they read).  By default they
 * belong to the most recent encode call; gz_exchange_select(ctx, 1) makes the following ones belong to the call
 * before it (double buffering: enqueue step k, then exchange step k-1).  gz_sync waits for both streams. */
int  gz_exchange_select(gz_ctx *ctx, int back);
int  gz_comm_init(gz_ctx *ctx, const uint8_t id[128], int rank, int world);
int  gz_gather_rows(gz_ctx *ctx, const int32_t *send_dev, int64_t n_rows_local, int32_t row_len,
                    int32_t *recv_dev, const int64_t *rows_per_rank, int root);
/* Every gz_gather_rows is bracketed by a pair of events on the exchange stream: this call synchronises that stream and
 * returns, oldest first, the time from the moment the gather COULD start (the encode call it belongs to was done, earlier
 * exchange operations had drained) to its last byte received / sent, of the last (up to 64, up to `max`) gathers, then
 * forgets them. */
int  gz_exchange_timing_history(gz_ctx *ctx, double *out_ms, int32_t max, int32_t *n_out);

/* Compact form of dense rows for the exchange step: most of a [n_rows, row_len] block is padding, so a rank sends
 * only the n_real[i] leading entries of every row (+ the counts) and the root re-creates padding and mask.
 *   gz_compact_rows   rows_dev [n_rows,row_len] + n_real_dev [n_rows] -> out_dev (concatenated leading entries);
 *                     *total_host = number of entries written (the call synchronises the context's stream)
 *   gz_expand_rows    the inverse on the receiving side: compact_dev + n_real_dev -> ids_dev, mask_dev
 *                     ([n_rows,row_len] each; padding = the pad id of the loaded tables, mask = id != pad) */
int  gz_compact_rows(gz_ctx *ctx, const int32_t *rows_dev, const int32_t *n_real_dev, int64_t n_rows, int32_t row_len,
                     int32_t *out_dev, int64_t *total_host);
int  gz_expand_rows(gz_ctx *ctx, const int32_t *compact_dev, const int32_t *n_real_dev, int64_t n_rows, int32_t row_len,
                    int32_t *ids_dev, int32_t *mask_dev);
/* The same with 16-bit entries (half the bytes on the link): only when every id of the loaded vocabulary fits 16 bits
 * (GZ_E_LIMIT otherwise); lossless. */
int  gz_compact_rows16(gz_ctx *ctx, const int32_t *rows_dev, const int32_t *n_real_dev, int64_t n_rows, int32_t row_len,
                       uint16_t *out_dev, int64_t *total_host);
int  gz_expand_rows16(gz_ctx *ctx, const uint16_t *compact_dev, const int32_t *n_real_dev, int64_t n_rows, int32_t row_len,
                      int32_t *ids_dev, int32_t *mask_dev);
/* One rank's whole message of the exchange step as ONE block (one ncclSend per peer and shard):
 *     block_dev = [ int32 n_real[n_rows] | uint32 first[n_rows] | the rows' real entries, `bits` (16 | 32) bits each ]
 * row r's entries are the n_real[r] ones from entry first[r] on (round 4: no `first` -- the receiver scanned n_real again).
 * gz_compact_block writes it from dense rows (total_host receives the number of entries; the block is 2 n_rows + ceil(total * bits /
 * 32) int32 words long), gz_expand_block is the inverse on the receiving side: `total_entries` is the entry count the block was
 * announced with (what gz_block_total / gz_compact_block returned on the sending side).  A block comes from another process: a row
 * that claims more than row_len entries, or entries beyond total_entries -- a truncated block, one written in another layout -- is
 * written as padding only, never followed, and the next gz_sync answers GZ_E_INVALID.
 * gz_encode_emit_block arms the NEXT encode call (a dense one: GZ_E_INVALID from that call otherwise) to leave its block in block_dev
 * (2 n_docs + ceil(n_docs * max_len * bits / 32) words at most) as part of the call itself: the scan of the row lengths and the
 * compact kernel run right behind the call's kernels on the call's own stream (beside the NEXT call's kernels, on the exchange
 * stream, the same work costs those kernels more than it takes here), and nothing of it waits for the host.  gz_block_total(back)
 * waits for the encode call `back` (0..2) calls ago and returns its block's number of entries (GZ_E_INVALID when that call emitted
 * none).  An arming holds for ONE gz_encode_batch_device[_h] call -- whether that call makes the block or fails; block_dev == NULL disarms. */
int  gz_encode_emit_block(gz_ctx *ctx, int32_t *block_dev, int32_t bits);
int  gz_block_total(gz_ctx *ctx, int32_t back, int64_t *total_host);
int  gz_compact_block(gz_ctx *ctx, const int32_t *rows_dev, const int32_t *n_real_dev, int64_t n_rows, int32_t row_len,
                      int32_t bits, int32_t *block_dev, int64_t *total_host);
int  gz_expand_block(gz_ctx *ctx, const int32_t *block_dev, int32_t bits, int64_t n_rows, int32_t row_len, int64_t total_entries,
                     int32_t *ids_dev, int32_t *mask_dev);

#ifdef __cplusplus
}
#endif
#endif /* GENZ_TOKENIZE_H */
