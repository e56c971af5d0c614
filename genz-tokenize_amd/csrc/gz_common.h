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
constexpr uint32_t GZ_MAX_RANKS   = (1u << 24) - 1;   // merge-file line numbers
constexpr uint32_t GZ_SYM_UNKNOWN = 0x80000000u;
constexpr uint32_t GZ_SYM_LASTBIT = 0x40000000u;
constexpr uint32_t GZ_NO_SYMBOL   = 0xFFFFFFFFu;      // table entry: code point has no interned symbol
constexpr uint32_t GZ_RANK_NONE   = 0xFFFFFFFFu;      // "pair is not in bpe_ranks" (the float('inf') of tokenize.py:71)
constexpr uint32_t GZ_PAIR_EMPTY  = 0xFFFFFFFFu;      // `left` of an empty pair slot

// pair -> rank hash table entry: GzPairSlot {left, right, merged, rank} below (16 bytes, 32-bit compares only)
// slot of the pair (a, b) in a table of 2^(32 - shift) slots: multiplicative hashing, the TOP bits of one multiply-add
// (three VALU instructions in the merge loop, which runs one probe per iteration)
GZ_HD uint32_t gz_pair_slot(uint32_t a, uint32_t b, uint32_t shift) { return (a * 0x9E3779B1u + b * 0x85EBCA6Bu) >> shift; }
GZ_HD uint32_t gz_cp_hash(uint32_t cp)
{
    uint32_t h = cp * 0x9E3779B1u;
    return h ^ (h >> 16);
}

// Hash tables (pair table, whole-word tables): slots >= GZ_TAB_SLACK x entries, rounded up to a power of two.  A wave
// probes 64 keys at once and loops until its LAST lane is done, so the tail of the probe-length distribution is what
// costs.  Measured (us per launch of 1 M documents) for 2 / 4 / 8 / 16 / 32: word kernel 416 / 374 / 369 / 346 / 361,
// merge kernel 494 / 400 / 369 / 354 / 357.
#ifndef GZ_TAB_SLACK
#define GZ_TAB_SLACK 16
#endif
// (the environment variable GZ_TAB_SLACK = 2 .. 64 overrides it when the tables are built: the parity suite runs its
// golden batches once more on tables at half load, where probe chains are long, so that the continue-probing paths of the
// kernels stay exercised)
inline size_t gz_tab_slack()
{
    static const size_t v = [] {
        const char* e = getenv("GZ_TAB_SLACK");
        const long k = e ? atol(e) : 0;
        return (size_t)(k >= 2 && k <= 64 ? k : GZ_TAB_SLACK);
    }();
    return v;
}
struct GzPairSlot  { uint32_t left, right, merged, rank; };              // left == GZ_PAIR_EMPTY -> empty; 32-bit compares only
struct GzMergeInfo { uint32_t left, right, merged, pad; };   // indexed by rank
struct GzSymIds    { int32_t nonfinal, final_; };            // vocab id of  sym+"@@"  /  sym minus "</w>"
struct GzCpSyms    { uint32_t plain, final_; };              // symbol of  c  /  c+"</w>"   (GZ_NO_SYMBOL if none)
struct GzAstral    { uint32_t cp, plain, final_, pad; };     // open-addressing entry, cp == GZ_NO_SYMBOL -> empty
// Whole-word tables: word bytes -> the single vocab id bpe() yields for that word.  Built at table-load time by
// running the GPU merge path itself over every candidate word (gz_api.cpp).
// first whole-word table: words of <= 12 bytes (96 % of running text) in 16-byte slots -> ONE 16-byte load per probe
struct GzWordSlot0 { uint64_t lo; uint32_t hi; uint32_t meta; };        // meta = len:4 | id << 4;  0 -> empty
GZ_HD uint32_t gz_word_hash0(uint64_t lo, uint32_t hi, uint32_t len)
{
    uint32_t h = (uint32_t)lo * 0x9E3779B1u ^ (uint32_t)(lo >> 32) * 0x85EBCA6Bu ^ hi * 0xC2B2AE35u ^ len * 0x165667B1u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}
// second whole-word table: words of 13..32 bytes (64-byte slots), probed by the miss kernel
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

// Device-resident tables, passed to kernels by value.
struct GzDeviceTables {
    const GzPairSlot*  pair_tab;    uint32_t pair_mask;      // slots-1 (power of two; >= 16 slots)
    uint32_t           pair_shift;  uint32_t pair_pad;       // 32 - log2(slots): gz_pair_slot(a, b, pair_shift)
    const GzMergeInfo* merges;      uint32_t n_ranks;
    const GzSymIds*    sym_ids;     uint32_t n_symbols;
    const GzCpSyms*    bmp;                                   // 65536 entries
    const GzAstral*    astral;      uint32_t astral_mask;    // slots-1; astral == nullptr when no astral symbol exists
    int32_t pad_id, bos_id, eos_id, unk_id;
    const GzWordSlot0* words0;      uint32_t word0_mask;     // whole-word table, <= 12 bytes (nullptr until built)
    const GzWordSlot2* words2;      uint32_t word2_mask;     // whole-word table, 13..32 bytes (may be nullptr)
};

// Host-side result of the loader (tokenize.py:31-57) and of the table build.
// ---- decoder snapshot (id -> word bytes), tokenize.py:40 -------------------------------------------------------------
struct GzDecEntry { uint32_t off; uint32_t len_flags; };       // len:24 | flags
constexpr uint32_t GZ_DEC_LEN_MASK  = 0x00FFFFFFu;
constexpr uint32_t GZ_DEC_ENDS_ATAT = 0x01000000u;              // the word ends in "@@"
constexpr uint32_t GZ_DEC_INNER     = 0x02000000u;              // the word contains "@@ " inside: filter byte by byte
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
    std::vector<GzPairSlot>  pair_tab;
    std::vector<GzMergeInfo> merges;
    std::vector<GzSymIds>    sym_ids;
    std::vector<GzCpSyms>    bmp;
    std::vector<GzAstral>    astral;     // empty when unused
    uint32_t max_probe = 0;
};

// Returns GZ_OK / GZ_E_UTF8 / GZ_E_LIMIT; `err` receives a message.
int gz_build_tables(const uint8_t* vocab, size_t vocab_len, const uint8_t* bpe, size_t bpe_len,
                    const char* const specials[5], GzHostTables& out, std::string& err);

// error text of C-ABI calls that have no context (gz_create, gz_host_tables_create, ...); thread-local (gz_host_api.cpp)
std::string& gz_create_err();

// true when the n bytes are valid UTF-8 holding no whitespace code point (so the regex "\S+" sees ONE word)
bool gz_is_plain_word(const uint8_t* p, size_t n);
