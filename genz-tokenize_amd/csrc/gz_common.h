// Shared between the host table builder (gz_tables.cpp), the kernels (gz_kernels.hip) and the C ABI (gz_api.cpp).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GZ_HD __host__ __device__ __forceinline__
#else
#define GZ_HD inline
#endif

// ---------------------------------------------------------------------------------------------------------
// Symbol values as the kernels hold them
//   bit 31 clear : interned symbol id (< GZ_MAX_SYMBOLS).  A symbol is a STRING of the reference's bpe()
//                  (tokenize.py:62-101): two merge histories that spell the same string are one symbol.
//   bit 31 set   : a code point that is in no merge and no vocab entry (low 21 bits = code point,
//                  bit 30 = it is the word's last character).  It can never merge and maps to the unk id.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t GZ_MAX_SYMBOLS = (1u << 20) - 2;   // ids 0 .. 2^20-3; 0xFFFFF is reserved (empty key)
constexpr uint32_t GZ_MAX_RANKS   = (1u << 23) - 1;   // merge-file line numbers (23 bits in a GzPair8 entry)
constexpr uint32_t GZ_SYM_UNKNOWN = 0x80000000u;
constexpr uint32_t GZ_SYM_LASTBIT = 0x40000000u;
constexpr uint32_t GZ_NO_SYMBOL   = 0xFFFFFFFFu;      // table entry: code point has no interned symbol
constexpr uint32_t GZ_RANK_NONE   = 0xFFFFFFFFu;      // "pair is not in bpe_ranks" (the float('inf') of tokenize.py:71)
constexpr uint32_t GZ_PAIR_EMPTY  = 0xFFFFFFFFu;      // `left` of an empty pair slot

// Byte counts the 32-bit paths take (gz_limit, include/genz_tokenize.h).  One text of an encode call: positions, word starts and
// token places are 32-bit on the device (GZ_E_LIMIT from this size on: split the batch).  The text pre-pass has no such limit of
// its own: up to this many INPUT bytes it scans its output lengths in 32 bits (filters never grow a document, so the output
// offsets and their total fit too); from this size on it takes the 64-bit scan and pack kernels.
constexpr int64_t GZ_TEXT_BYTES_LIMIT = 0xFFFF0000ll;
constexpr int64_t GZ_PP_TAIL32_LIMIT = 0xFFFF0000ll;
inline bool gz_pp_tail_is_32bit(int64_t in_bytes) { return in_bytes >= 0 && in_bytes < GZ_PP_TAIL32_LIMIT; }

GZ_HD uint32_t gz_cp_hash(uint32_t cp)
{
    uint32_t h = cp * 0x9E3779B1u;
    return h ^ (h >> 16);
}

// ---------------------------------------------------------------------------------------------------------
// The library's switches for tests and experiments (gz_debug_set, include/genz_tokenize.h).  Typed, named, with the product's
// behaviour as the default; NOTHING here is read from the environment.  A context copies the process-wide defaults when it is
// created (gz_debug_set(NULL, ...) changes those; gz_debug_set(ctx, ...) one context).  The keys marked "builder" act where the
// tables are built, which has no context: they are read from the process-wide defaults at that moment.
// ---------------------------------------------------------------------------------------------------------
#ifndef GZ_TAB_SLACK
#define GZ_TAB_SLACK 16
#endif
struct GzOptions {
    // ---- which kernels a call takes
    int32_t small = 1;                // 1: small dense / ragged batches in ONE launch (gz_small_kernel); 0: everything through the kernel pipeline
    int32_t small_wgs = 768;          // workgroups the one-launch kernel aims for
    int32_t host_direct = 4096;       // host calls whose inputs and outputs both fit this many bytes are computed straight out of / into the pinned
                                      // staging block (the kernels read and write host memory over the bus: no copy in, no copy back); 0: never
    int32_t assemble = 3;             // row writer of dense single texts: 3 gz_rows1_kernel, 2 the pair-mode kernel (rows through LDS), 1 the ragged layouts' scatter kernel
    int32_t word_table = 1;           // 0: every word through the merge loop (as GZ_NO_WORD_TABLE on every call)
    int32_t pp_fused = 1;             // 0: the text pre-pass filter by filter for every document
    int32_t sub_batches = 1;          // dense batches cut into this many document ranges on two streams (1 .. 8)
    int32_t docs_per_wave = 0;        // documents per wave of gz_assemble_kernel (0: by the batch's shape)
    // ---- how the pipeline of one text is scheduled
    int32_t side = 1;                 // 0: no side stream (docw0, wide / long words on the main stream)
    int32_t brk_side = 1;             // 0: document-start bits and control words prepared on the main stream
    int64_t scan_multi = 8192;        // block counts from which the chained multi-workgroup scan is used (0: always)
    int64_t near_limit = 1 << 25;     // token places below this get near records (smaller: the far form on small batches)
    int32_t hot_wgs = 0;              // grid of gz_words2_kernel (0: as many workgroups as the chip holds)
    int32_t hot_miss_wgs = 0;         // grid of gz_miss2_kernel (0: as above)
    int64_t m2_split_min = 65536;     // chunks of misses from which the merge kernel's two instances share a launch (0: never)
    int32_t m2_split_always = 0;      // 1: ... also with the whole-word tables on
    // ---- builder (process-wide: read when tables are built)
    int32_t tab_slack = GZ_TAB_SLACK; // long-key whole-word table: slots >= this x entries (2 .. 64)
    int32_t ph_force_overflow = 0;    // k > 0: the perfect-hash builder refuses every k-th bucket (overflow paths of the kernels)
    int32_t ph_hot_slots = 1024;      // slots at the head of the whole-word table reserved for the most frequent words (0 .. 8192)
    int32_t word_weights = 0;         // frequency estimate of whole words: 0 auto, 1 the vocab file's counts, 2 merge ranks
    // ---- the host paths (gz_api.cpp)
    int32_t host_threads = 0;         // threads that move a large host call's rows into the caller's arrays (0: by the host's processors, at most 16)
    int32_t dense_csr = 1;            // 1: a large dense single-text host call brings only the rows' real entries over the bus and pads them on the host; 0: the dense rows cross
    int32_t host_hints = 0;           // fresh output arrays of a large host call: bit 0 MADV_HUGEPAGE on them, bit 1 MADV_POPULATE_WRITE per piece before it is written
                                      // (0: neither -- on the GPU box 16 threads fill 2 GB of fresh numpy arrays in 27 ms without them, 29-31 ms with)
    int32_t inject_bad_alloc = 0;     // test hook: k > 0 makes the k-th allocation site reached from now on throw std::bad_alloc (counts down to 0)
    // ---- diagnostic build only (results are WRONG with ablate / rows_dbg)
    int32_t diag_poison = 0, rows_dpw = 0, rows_dbg = 0, ablate = 0;
    int32_t diag_fresh = 0;           // v > 0: every FRESH device allocation is filled with byte v - 1 before it is used (fresh memory is usually zero:
                                      // a kernel that reads a word nobody wrote then reads garbage every time, not on the day the memory is reused)
    int32_t diag_fresh_only = -1;     // >= 0: ... only the allocation with this running number (of the context), and its name goes to stderr
    int32_t diag_exact = 0;           // 1: hipMalloc of exactly the bytes asked for (16-byte rounded), no slack, no 4-KiB rounding: buffers are re-allocated
                                      // as often as with the guard allocator, in ordinary memory
    int32_t diag_guard = 0;           // device buffers from a guard-granule allocator: 1 the buffer ENDS at the last byte of its mapping (an unmapped granule behind it),
                                      // 2 it STARTS at the first byte (an unmapped granule before it); no slack behind a buffer in either mode
};
GzOptions& gz_default_options();                                          // gz_host_api.cpp
int gz_option_set(GzOptions& o, const char* key, int64_t value);           // GZ_OK, or GZ_E_INVALID for an unknown key / a value out of range

// The long-key whole-word table (words of 17..32 bytes, probed once per MISS) is a plain open-addressing table: slots >=
// tab_slack x entries, rounded up to a power of two.  (The parity suite runs its golden batches once more at half load, where probe
// chains are long, so that the continue-probing paths stay exercised.)
inline size_t gz_tab_slack() { return (size_t)gz_default_options().tab_slack; }

// ---------------------------------------------------------------------------------------------------------
// Static perfect hashing (hash and displace) for the tables the two hot kernels probe once per word / once per
// adjacent pair.  The tables never change after gz_load_tables, so the host places every key where ONE load finds it:
//     bucket = ha(key) >> bshift;   d = disp[bucket];   slot = ((hb(key) ^ d) * GZ_PH_MUL) >> sshift
// `disp` (16 bits per bucket, <= GZ_PH_LDS_BUCKETS buckets) is staged in LDS once per workgroup: a probe is one
// ds_read_u16 and ONE global load from a DENSE table (load <= 0.8) that stays resident in every XCD's L2 -- no probe
// loop, no tail, no empty-slot slack.  The key is always compared in full, so what a probe returns never depends on
// the hash functions.  A bucket the builder could not place (never seen on real tables; forced by the tests through
// the switch ph_force_overflow) gets d = GZ_PH_OVERFLOW: its keys are inserted by linear probing from their slot, and only
// lookups that land in such a bucket ever probe further.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t GZ_PH_MUL = 0x2C1B3C6Du;
constexpr uint32_t GZ_PH_OVERFLOW = 0xFFFFu;
constexpr uint32_t GZ_PH_LDS_BUCKETS = 16384;                 // displacement entries a workgroup stages in LDS (32 KB)
constexpr uint32_t GZ_PAIR_HOT_SLOTS = 4096, GZ_PAIR_HOT_SHIFT = 20;      // 32 KB of LDS
constexpr uint32_t GZ_WORD_HOT_SLOTS = 1024, GZ_WORD_HOT_SHIFT = 22;      // 16 KB of LDS: the word kernel's hot words (16-byte entries)
struct GzPh {
    const uint16_t* disp;                                     // [nbuckets] (device copy)
    uint32_t nbuckets, bshift;                                // bucket = ha >> bshift      (nbuckets = 2^(32 - bshift) >= 16)
    uint32_t sshift, mask;                                    // slot = (...) >> sshift     (slots = mask + 1 = 2^(32 - sshift) >= 16)
    uint32_t k1, k2;                                          // seeded multipliers of ha (the builder retries with other seeds)
};
GZ_HD uint32_t gz_ph_slot(uint32_t hb, uint32_t d, uint32_t sshift) { return ((hb ^ d) * GZ_PH_MUL) >> sshift; }
// pair (a, b), both < 2^20.  (ha, hb) is an invertible linear map of (a, b) mod 2^32 for odd k1, k2 (the determinant
// k1 * 0x27D4EB2F - k2 * 0xC2B2AE36 is odd): two pairs never share both hashes
GZ_HD uint32_t gz_pair_ha(uint32_t a, uint32_t b, uint32_t k1, uint32_t k2) { return a * k1 + b * k2; }
GZ_HD uint32_t gz_pair_hb(uint32_t a, uint32_t b) { return a * 0xC2B2AE36u + b * 0x27D4EB2Fu; }
// Pair table entry, 8 bytes: lo = left | right << 20 (low 12 bits of right), hi = right >> 12 | flag << 8 | rank << 9.
// Symbols are numbered so that the string a merge produces has the merge's rank as its id (gz_tables.cpp): the merged
// symbol IS the rank unless two merge lines spell the same string -- then the flag is set and merges[rank].merged holds it.
struct GzPair8 { uint32_t lo, hi; };                         // empty: {0xFFFFFFFF, 0xFFFFFFFF} (symbol 0xFFFFF is never interned)
constexpr uint32_t GZ_PAIR8_ALIAS = 0x100u;                   // (hi >> 8 = rank << 1 | flag: ordered like the ranks)
struct GzMergeInfo { uint32_t left, right, merged, pad; };   // indexed by rank
struct GzSymIds    { int32_t nonfinal, final_; };            // vocab id of  sym+"@@"  /  sym minus "</w>"
struct GzCpSyms    { uint32_t plain, final_; };              // symbol of  c  /  c+"</w>"   (GZ_NO_SYMBOL if none)
struct GzAstral    { uint32_t cp, plain, final_, pad; };     // open-addressing entry, cp == GZ_NO_SYMBOL -> empty
// Whole-word tables: word bytes -> the single vocab id bpe() yields for that word.  Built at table-load time by
// running the GPU merge path itself over every candidate word (gz_api.cpp).
// long-key whole-word table: words of 17..32 bytes (64-byte slots, linear probing), probed once per miss by the merge pre-pass
struct GzWordSlot2 { uint64_t k[4]; uint32_t len; int32_t id; uint32_t pad[6]; };    // len == 0 -> empty
GZ_HD uint32_t gz_word_hash2(const uint64_t k[4], uint32_t len)
{
    uint32_t h = len * 0x165667B1u;
    for (int i = 0; i < 4; ++i) {
        h ^= (uint32_t)k[i] * 0x9E3779B1u + (uint32_t)(k[i] >> 32) * 0x85EBCA6Bu;
        h = (h << 13) | (h >> 19);
        h *= 0xC2B2AE35u;
    }
    h ^= h >> 15;
    return h;
}

// Whole-word table of the big pipeline's word kernel: words of <= 16 bytes (99.4 % of running text) in 32-byte slots,
// perfectly hashed: key bytes 0..11 (zero padded), meta = len:5 | id << 5 (0: empty), key bytes 12..15.  The FIRST 16 bytes of
// a slot answer a word of <= 12 bytes (97 % of the probes): one 16-byte load per lane; only lanes with a longer word load the
// slot's fifth dword (a wave's scattered loads cost the CU's L1 a look-up per lane and load: 128 per round before, ~ 67 now).
struct GzWordSlot1 { uint32_t k[3]; uint32_t meta; uint32_t k3; uint32_t pad[3]; };
struct GzWordHot { uint32_t k[3]; uint32_t meta; };         // a hot word: the first 16 bytes of its GzWordSlot1 (meta 0: empty)
GZ_HD uint64_t gz_slot1_lo(const GzWordSlot1& e) { return (uint64_t)e.k[0] | ((uint64_t)e.k[1] << 32); }
GZ_HD uint64_t gz_slot1_hi(const GzWordSlot1& e) { return (uint64_t)e.k[2] | ((uint64_t)e.k3 << 32); }
// The hashes of such a key, from ONE mix h0 (four multiplies, three xors) with the builder's seeds k1, k2:
//     bucket: the top bits of ha = finalised h0;   slot: ((h0 ^ displacement) * GZ_PH_MUL) >> sshift   (hb = h0)
// Two keys with the same h0 would share bucket AND slot under every displacement: the builder then tries other seeds (a
// 32-bit collision among the few ten thousand words of a vocabulary is rare), and what it cannot place goes to an overflow bucket.
GZ_HD uint32_t gz_word1_h0(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t len, uint32_t k1, uint32_t k2)
{
    return a * k1 ^ b * k2 ^ c * 0xC2B2AE35u ^ (d + len) * 0x165667B1u;
}
GZ_HD uint32_t gz_word1_ha_of(uint32_t h0)
{
    h0 ^= h0 >> 15;
    return h0 * 0x2C1B3C6Du;                                  // (only the top bits are used: the bucket)
}
GZ_HD uint32_t gz_word1_hb(uint64_t lo, uint64_t hi, uint32_t len, uint32_t k1, uint32_t k2)
{
    return gz_word1_h0((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), len, k1, k2);
}
GZ_HD uint32_t gz_word1_ha(uint64_t lo, uint64_t hi, uint32_t len, uint32_t k1, uint32_t k2)
{
    return gz_word1_ha_of(gz_word1_hb(lo, hi, len, k1, k2));
}

// slots at the head of the whole-word table that the builder reserves for the most frequent words (0: no steering)
inline uint32_t gz_word_hot_slots() { return (uint32_t)gz_default_options().ph_hot_slots; }

// Device-resident tables, passed to kernels by value.
struct GzDeviceTables {
    const GzMergeInfo* merges;      uint32_t n_ranks;
    const GzSymIds*    sym_ids;     uint32_t n_symbols;
    const GzCpSyms*    bmp;                                   // 65536 entries
    const GzAstral*    astral;      uint32_t astral_mask;    // slots-1; astral == nullptr when no astral symbol exists
    int32_t pad_id, bos_id, eos_id, unk_id;
    const GzWordSlot2* words2;      uint32_t word2_mask;     // whole-word table, 17..32 bytes (may be nullptr)
    // perfect-hash tables (one family for every kernel: the big pipeline stages the displacement arrays in LDS, the
    // one-launch small-batch kernel and the wide / long word kernels read them from memory)
    const GzPair8*     pair8;       GzPh pair_ph;             // pair -> rank (= merged symbol)
    const GzWordSlot1* words0p;     GzPh word0_ph;            // whole-word table, <= 16 bytes (nullptr until built)
    uint32_t pair_ovf, word0_ovf;                             // keys in overflow buckets (0 on every real table: the kernels' fast forms)
    // hot set of the merge kernel, staged in LDS (same entry layout, direct-mapped, slot = top bits of ha): a probe that
    // hits there costs no memory traffic -- a random probe of a table in L2 moves a whole 128-byte line for 8 bytes
    const GzPair8*     pair_hot;    uint32_t pair_hot_shift;  // [2^(32 - shift)] the most frequent merges (smallest ranks)
    // hot words of the word kernel, staged in LDS: the first 16 bytes of a whole-word slot (key bytes 0..11, len | id << 5; keys of
    // <= 12 bytes only), direct-mapped by the top bits of ha; the most frequent word keeps a contested place.  A SUBSET of
    // words0p: it changes where a probe is answered, never what it answers.
    const GzWordHot*   word_hot;                              // [GZ_WORD_HOT_SLOTS]
};

// Host side of a perfect hash: hashes in, displacement array + the slot of every key out (gz_tables.cpp).
struct GzPhHost {
    std::vector<uint16_t> disp;
    uint32_t nbuckets = 0, bshift = 0, sshift = 0, slots = 0, k1 = 0, k2 = 0;
    uint32_t n_overflow = 0;                                  // keys placed by linear probing (0 on every table seen so far)
};
// hashes(ctx, key index, k1, k2, &ha, &hb) gives the two hashes of a key under the seeds of an attempt; slot_of[key] receives
// the key's slot.  Always succeeds.
// hot (optional, [n]): 1 = a key that is looked up very often.  Such keys are steered into the first `hot_slots` slots of the
// table -- the builder is free to choose any displacement that works, so it prefers one that lands a bucket's hot key there
// and keeps every other key out -- and so share a few cache lines that stay resident in every CU's L1 (a perfect hash
// otherwise puts the few thousand words running text consists of on as many different lines).  Placement only: what a probe
// returns never depends on it.
void gz_ph_build(size_t n, void (*hashes)(const void* ctx, size_t i, uint32_t k1, uint32_t k2, uint32_t* ha, uint32_t* hb), const void* ctx,
                 GzPhHost& out, std::vector<uint32_t>& slot_of, const uint8_t* hot = nullptr, uint32_t hot_slots = 0);

// Host-side result of the loader (tokenize.py:31-57) and of the table build.
// ---- decoder snapshot (id -> word bytes), tokenize.py:40 -------------------------------------------------------------
// 16 bytes per id, ONE load: len:24 | flags, then -- GZ_DEC_INLINE: words of <= 12 bytes without an inner "@@ ", nearly all --
// the word's bytes themselves (zero padded), else w[0] = the word's offset in the byte arena
struct GzDecEntry { uint32_t len_flags; uint32_t w[3]; };
constexpr uint32_t GZ_DEC_LEN_MASK  = 0x00FFFFFFu;
constexpr uint32_t GZ_DEC_ENDS_ATAT = 0x01000000u;              // the word ends in "@@"
constexpr uint32_t GZ_DEC_INNER     = 0x02000000u;              // the word contains "@@ " inside: filter byte by byte
constexpr uint32_t GZ_DEC_INLINE    = 0x04000000u;              // the word's bytes are in the entry
constexpr uint32_t GZ_DEC_ABSENT    = 0xFFFFFFFFu;              // no word has this id
struct GzDecTable {
    const GzDecEntry* entries;      // [n_ids + 1]; entries[n_ids] = the unk string of the current call
    const uint8_t*    bytes;
    int32_t           n_ids;
};

struct GzHostTables {
    // encoder in insertion order (Python dict order): word bytes, id (ids can repeat: rule L3)
    std::vector<std::string> enc_words;
    std::vector<int32_t>     enc_ids;
    int32_t special_ids[5] = {0, 0, 0, 0, 0};
    // bpe_ranks in insertion order: fields joined by '\n', field count, rank
    std::vector<std::string> rank_keys;
    std::vector<int32_t>     rank_nfields;
    std::vector<int32_t>     rank_vals;
    // interned symbols
    std::vector<std::string> symbols;
    // device images
    std::vector<GzMergeInfo> merges;
    std::vector<GzSymIds>    sym_ids;
    std::vector<GzCpSyms>    bmp;
    std::vector<GzAstral>    astral;     // empty when unused
    std::vector<GzPair8>     pair8;      // pair -> rank, perfectly hashed
    GzPhHost                 pair_ph;
    std::vector<GzPair8>     pair_hot;   // direct-mapped hot set (GZ_PAIR_HOT_SLOTS entries)
    std::vector<uint64_t>    enc_count;  // per encoder entry: the count the vocab line carries (0: none) -- only ever a PLACEMENT hint:
                                         // the most frequent words share the first lines of the whole-word table (gz_ph_build)
};

// Returns GZ_OK / GZ_E_UTF8 / GZ_E_LIMIT; `err` receives a message.
int gz_build_tables(const uint8_t* vocab, size_t vocab_len, const uint8_t* bpe, size_t bpe_len,
                    const char* const specials[5], GzHostTables& out, std::string& err);

// error text of C-ABI calls that have no context (gz_create, gz_host_tables_create, ...); thread-local (gz_host_api.cpp)
std::string& gz_create_err();

// true when the n bytes are valid UTF-8 holding no whitespace code point (so the regex "\S+" sees ONE word)
bool gz_is_plain_word(const uint8_t* p, size_t n);
