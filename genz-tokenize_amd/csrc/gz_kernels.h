// Kernel argument blocks and launchers (gz_kernels.hip) used by the C ABI (gz_api.cpp).
#pragma once
#include "gz_common.h"
#include <hip/hip_runtime.h>

constexpr int GZ_WAVES_PER_BLOCK = 2;
constexpr int GZ_MAX_DOCS_PER_WAVE = 16;  // a wave owns up to this many consecutive documents
constexpr int32_t GZ_NONE_ = -1;        // == GZ_NONE of the public header

// max_len / padding / truncation of Tokenize.__call__ (tokenize.py:184-190)
struct GzShape {
    int32_t max_len;      // meaningful only when pad_mode
    int32_t pad_mode;     // `max_len is not None and padding` (tokenize.py:247, :256)
    int32_t truncation;
};

struct GzFinalizeArgs {
    const int64_t* text_off; const int64_t* pair_off;
    int64_t n_docs;
    GzShape S;
    const int32_t* raw; const int32_t* n_raw;
    int64_t* row_off;     // [n_docs+1] (written by the scan that precedes the finalize kernel)
    int64_t capacity;
    int32_t* ids; int32_t* mask; int32_t* n_real;
    int32_t* error_flag;  // set to 1 when row_off[n_docs] > capacity
};

struct GzPairArgs {
    int64_t n_docs;
    GzShape S;            // dense when row_off == nullptr (rows of max_len)
    const int64_t* row_off; int64_t capacity;
    const int32_t* ids;
    int32_t* seq; int32_t* tt; int32_t* pair_len; int32_t* status;
};

// ---- multi-kernel pipeline (gz_pipeline.inc) ---------------------------------------------------------------------
struct GzTextBufs {              // one text (A or B) of a batch and its per-call workspace
    const uint8_t* tb;           // first byte of the batch (text + off[0])
    const int64_t* off;          // [n_docs+1] absolute offsets
    int64_t B;                   // bytes of the batch
    int64_t nblk;                // 4-KiB blocks covering positions 0 .. B
    uint16_t* brk;               // bitmaps, 16 bits per 16 bytes: document starts / word starts / word ends
    uint16_t* st;
    uint16_t* en;
    uint32_t* blkcnt;            // [nblk+1] words per block, then (scanned in place) index of each block's first word
    uint32_t* docw0;             // [n_docs+1] index of each document's first word
    uint32_t* wtok;              // [words] id, or a merged word's record (far: MISS | token count; near: count and place in one word)
    uint32_t* waux;              // [words] far records only: where the word's tokens are (mtok[waux ...]; wide / long words: their byte offset)
    uint4* mlist;                // [words] the misses of block b, compact, at mlist[blkcnt[b] ...]: {word index, byte offset, pending record, 0}
    uint32_t* blkmiss;           // [nblk+1] number of misses per block; scanned in place before the merge pre-pass ([nblk] = total)
    uint32_t* grpblk;            // [words/64 + 2] block that holds miss number 64 g (written by the scan)
    uint32_t* tcnt;              // [words/1024 + 8] per 3 072-miss tile of mq: records the merge kernel takes | its chunks of 64 with a word of > 8 symbols << 16 (gz_mpre_kernel)
    int64_t wmax;                // upper bound of the number of words (sizes of the per-word arrays)
    uint32_t* ctl;               // [64] zeroed per call: [0] words for gz_long_kernel, [1] unused, [2] [3] ticket counters of the
                                 // chained scans, [4] cursor of the compact token area; wlist == ctl + 64
    uint4* mq;                   // [words] the misses, tile by tile (1 024) sorted by symbol count: {word index, byte offset, record, 0}
    uint64_t* lookback;          // [nblk / 4 + 2] chained-scan words of gz_scan32m_kernel {status:2, call:30, value:32}; never cleared
    uint32_t epoch;              // call number written into / expected in the chained-scan words
    uint32_t near_lim;           // places of the compact token area below this get near records (2^25; switch near_limit: smaller, for tests)
    uint32_t* blklong;           // [nblk] block holds a word for gz_long_kernel (zeroed per call)
    uint32_t* wlist;             // [0] count, then the words (indices) that need 32 or 64 lanes (zeroed count per call); from the END of the
                                 // array down (wlist[wmax + 6 - k]): the words gz_long_kernel takes
    uint16_t* tilecnt;           // [4 * nblk] word starts of the block that lie before each of its four 1-KiB tiles
    int32_t* mtok;               // [2 (B + 32)]: [0, B + 32) tokens of wide / long words, at the word's byte offset; from B + 32 on the
                                 // compact token area of gz_miss2_kernel (places handed out by gz_mpre_kernel, cursor: ctl[4])
};

struct GzAsmArgs {
    GzTextBufs X[2];
    int32_t n_texts;             // 1, or 2 in pair mode
    int64_t n_docs;
    int32_t dense, max_len;
    int32_t* ids; int32_t* mask; int32_t* raw; int32_t* n_real;
    int32_t docs_per_wave;
    // gz_rowsr_kernel (single texts without padding): the rows' places (write pass), the caller's capacity and the flag raised beyond it
    const int64_t* row_off; int64_t capacity; int32_t* error_flag;
};

// T_host: the host copy of the table descriptor (table sizes decide launch shapes)
// side / ev_fork0 / ev_fork / ev_join (may be null): a second stream on which gz_docw0_kernel runs beside the word kernel and
// the rare wide-word kernels beside the merge kernel
void gz_launch_pipeline_text(const GzOptions& O, const GzDeviceTables* T_dev, const GzDeviceTables& T_host, const GzTextBufs& X, int64_t n_docs, int use_words,
                             int32_t* long_flag /* device int, zeroed by the caller */, hipStream_t s,
                             hipStream_t side = nullptr, hipEvent_t ev_fork0 = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr,
                             hipEvent_t ev_brk = nullptr /* non-null: X.off is readable NOW (no copy of it is queued on s): the document-start
                                                            bits are prepared on the side stream, under whatever s is still running */);
void gz_launch_pick(const int64_t* off, const int64_t* off2, int64_t n_docs, int nsub, int64_t* out /* 2*(nsub+1) */, hipStream_t s);
void gz_launch_row_offsets(const int32_t* n_real, int64_t n_rows, uint32_t* off, hipStream_t s);
// first (may be null): receives where each row's entries start (= off[r]): the second array of an exchange block
void gz_launch_compact(const int32_t* rows, const uint32_t* off, int64_t n_rows, int32_t row_len, void* out, int bits /* 32 | 16 */, uint32_t* first,
                       hipStream_t s);
// row r: n_real[r] entries from first[r] on (scanned offsets, or a block's own array: the rows then lie in any order)
void gz_launch_expand(const void* compact, int bits, const uint32_t* first, const int32_t* n_real, int64_t n_rows, int32_t row_len, int32_t pad_id,
                      int32_t* ids, int32_t* mask, uint32_t total /* entries the compact array holds */, int32_t* bad /* set when a row does not fit */, hipStream_t s);
void gz_launch_assemble(const GzOptions& O, const GzDeviceTables* T_dev, const GzAsmArgs& A, hipStream_t s);
void gz_launch_rows_ragged(const GzDeviceTables* T_dev, const GzAsmArgs& A, int pass, int64_t text_bytes, hipStream_t s);
// small batches, one launch (gz_small.inc): G documents per workgroup, G <= GZ_SMALL_DOCS_PER_WG and every group of G
// documents (A and B texts together) <= GZ_SMALL_DOC_BYTES.  poff == nullptr: single texts.  dense: rows of max_len into
// ids / mask; else: unpadded rows into the raw area `ids` (document d at (bytes before d) + 2 d per text) and their
// lengths into n_real.  arena: [text bytes + pair bytes + 32] words of scratch for very long words.
constexpr int GZ_SMALL_DOC_BYTES = 4096, GZ_SMALL_DOCS_PER_WG = 64;
// layout 2 (rows without padding, ONE workgroup: (n_docs + G - 1) / G == 1): the kernel also makes row_off[n_docs + 1] and writes
// ids / mask at their final places (no rowlen / scan / finalize behind it); beyond `capacity` entries: *error_flag = 1, nothing written
void gz_launch_small(const GzDeviceTables* T_dev, const uint8_t* text0, const int64_t* off, int64_t base, const uint8_t* pair0, const int64_t* poff,
                     int64_t pbase, int64_t n_docs, int G, int layout /* 1 dense, 0 raw area, 2 placed */, int max_len, int use_words, int32_t* ids,
                     int32_t* mask, int32_t* n_real, int32_t* arena, int64_t* row_off, int64_t capacity, int32_t* error_flag, hipStream_t s);

void gz_launch_rowscan(const GzFinalizeArgs& F, int64_t* row_len_tmp, hipStream_t s);
void gz_launch_finalize(const GzDeviceTables& T, const GzFinalizeArgs& F, hipStream_t s);
void gz_launch_pair(const GzDeviceTables& T, const GzPairArgs& P, hipStream_t s);
void gz_launch_bpe_word(const GzDeviceTables* T_dev, const uint8_t* word, int64_t nbytes, uint32_t* arena,
                        int32_t* out, int32_t cap, int32_t* n_out, hipStream_t s);
// batch decode: out == nullptr -> row_bytes[n_rows] + out_off[n_rows + 1] (exclusive scan); else write the text
void gz_launch_decode(const GzDecTable& D, const int32_t* ids, const int64_t* row_off, int64_t n_rows, int64_t* row_bytes,
                      int64_t* out_off, uint8_t* out, int64_t capacity, hipStream_t s);

// text pre-pass (gz_preproc.inc): one filter over documents that sit in the slots of a packed text
struct GzPpArgs {
    const uint8_t* in; const int64_t* in_off;   // slot of document d = in[in_off[d] - in_off[0] ... (in_abs: see below)
    const int64_t* in_len;                      // ... its current length (nullptr: the whole slot, in_off[d+1] - in_off[d])
    int32_t in_abs;                             // 1: `in` is the caller's text, document d at in[in_off[d] ...] (absolute, as gz_encode_batch_device)
    int64_t n_docs;
    uint8_t* out;                               // pass 1: same slots, another buffer
    int64_t* out_len;                           // pass 1: the new lengths
    int64_t* aux;                               // [n_docs] html: position of the unclosed '<' (pass 0 -> pass 1)
    int32_t op;                                 // GZ_PP_*
    int64_t skip_upto;                          // documents whose INPUT slot is at most this long are left alone (gz_pp_fused_kernel did them); -1: none
};
struct GzPpFusedArgs {                          // the whole chain for short documents, on chip (gz_pp_fused_kernel)
    const uint8_t* in; const int64_t* in_off;   // the caller's text, absolute offsets
    int64_t n_docs;
    uint8_t* out; int64_t* out_len;             // the documents' slots of the chain's LAST buffer, their final lengths
    int32_t n_ops; int32_t ops[16];
    uint32_t* n_long;                           // += documents too long for this kernel (zeroed by the caller)
    uint32_t* out_len32;                        // the same lengths as 32-bit words (0 for a document left to the chain): input of the chained scan
};
#ifndef GZ_PPF_CAP
#define GZ_PPF_CAP 4096
#endif
constexpr int GZ_PP_FUSED_MAX_BYTES = GZ_PPF_CAP;     // == PPF_CAP (gz_preproc.inc)
void gz_launch_preprocess(const GzPpArgs& A, int pass, hipStream_t s);
void gz_launch_preprocess_fused(const GzPpFusedArgs& A, hipStream_t s);
void gz_launch_scan64(const int64_t* len, int64_t n, int64_t* out_off /* n+1 */, hipStream_t s);
void gz_launch_pp_pack(const uint8_t* in, const int64_t* in_off, const int64_t* len, int64_t n_docs, uint8_t* out, const int64_t* out_off,
                       hipStream_t s);
// the tail of the pre-pass when gz_pp_fused_kernel did every document: exclusive scan of the 32-bit lengths IN PLACE (chained scan
// over many workgroups, total at [n_docs]; lb / ctl / epoch as gz_scan32m_kernel wants them: ctl[1] = ticket counter, ctl[2] = time-out
// flag, both zeroed by the caller), then the slots -> the packed text (out may be null; documents that end beyond capacity are not
// written), four documents per wave, and out_off as 64-bit offsets
void gz_launch_pp_tail(const uint8_t* slots, const int64_t* in_off, uint32_t* len32, int64_t n_docs, uint8_t* out, int64_t capacity, int64_t* out_off,
                       unsigned long long* lb, uint32_t* ctl, uint32_t epoch, hipStream_t s);
