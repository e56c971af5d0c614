// HIP kernels of the genz-tokenize hot path for gfx950 (MI355X, wave64).
//
//   gz_encode_kernel     K1+K2+K3 fused: one wavefront per document.  Coalesced 16-B loads stage 1-KiB tiles of
//                        the packed UTF-8 through LDS; per-lane SWAR-free classification + wave prefix sums give
//                        the word list (Unicode whitespace split with the "\S+\n?" glue rule, tokenize.py:106);
//                        one lane per word runs the BPE merge loop (tokenize.py:62-101) against the pair->rank
//                        hash in HBM/L2 with its symbols in LDS; ids come from the symbol->vocab-id table
//                        (tokenize.py:120-121) and are framed, truncated, padded and masked in place
//                        (tokenize.py:126-152, :184-251).  Words longer than 16 symbols run wave-cooperatively
//                        (ballot + prefix-sum compaction), in LDS up to 1024 symbols, in a global arena beyond.
//   gz_rowlen_kernel / gz_scan_kernel / gz_finalize_kernel   ragged layouts only (padding=False, truncation=False,
//                        max_len None or < 1): row lengths, exclusive scan, copy + pad/cut (tokenize.py:141-146).
//   gz_pair_kernel       sequence_id / token_type_ids of sentence pairs (tokenize.py:154-182, :252-258).
//   gz_bpe_word_kernel   Tokenize.bpe(token) for one word (symbols out).
//
// Integer / byte work only: no MFMA.  Everything is bit-exact with the reference by construction; see DESIGN.md.
#include "gz_kernels.h"

namespace {

constexpr int WAVE = 64;
constexpr int WPB = GZ_WAVES_PER_BLOCK;      // waves per workgroup (each wave owns a group of documents)
constexpr int TILE = 1024;                   // bytes classified per tile: 16 per lane
constexpr int MAXSYM = 16;                   // symbols a lane-per-word merge holds
constexpr int LONGCAP = WAVE * MAXSYM;       // symbols the wave-cooperative LDS path holds (same LDS region)
constexpr int MAXWORDS = TILE / 2 + 1;       // a word needs >= 1 byte + >= 1 whitespace byte
constexpr int GMAX = GZ_MAX_DOCS_PER_WAVE;   // documents per wave
constexpr int RECCAP = 512;                  // word records buffered between two flushes
constexpr int MISSCAP = WAVE;                // words waiting for the merge loop
constexpr int MU = 1;                        // interleaved merge passes (4 words each)
constexpr uint32_t REC_MISS = 0x80000000u;   // record = [miss:1][doc:5][payload:26]  payload = vocab id | miss slot

struct alignas(16) WaveLds {
    uint32_t bytes[(TILE + 16) / 4];         // tile bytes + 16 look-ahead bytes
    uint32_t sym[LONGCAP];                   // lane-per-word: sym[k*64 + lane]; long path: sym[i]
    uint32_t rec[RECCAP];
    uint32_t miss_off[MISSCAP];              // word start, relative to the first byte of the wave's range
    uint32_t brk[TILE / 32];                 // document boundaries inside the current tile
    uint32_t doc_rel[GMAX + 1];              // document starts, relative to the first byte of the wave's range
    int32_t  doc_ntok[GMAX];                 // raw tokens emitted so far per document
    int64_t  row_base[GMAX];                 // element offset of each document's row
    uint32_t endmap[TILE / 32];              // bit p set = a word ends before byte p of the current tile
    uint16_t wstart[MAXWORDS + 3];
    uint16_t miss_len[MISSCAP];              // bytes | glue << 15
    uint8_t  miss_ntok[MISSCAP];
    uint8_t  pad_hit[GMAX];                  // a real token of this document equals the pad id (mask needs the slow path)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// streaming (write-once) 16-byte store that does not displace the tables from L2
__device__ __forceinline__ void nt_store4(int32_t* p, int32_t a, int32_t b, int32_t c, int32_t d)
{
    typedef int __attribute__((ext_vector_type(4))) v4i;
    v4i v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(p));
}
__device__ __forceinline__ uint64_t lt_mask(int lane) { return (1ull << lane) - 1ull; }

__device__ __forceinline__ int wave_excl_sum(int v, int lane, int& total)
{
    int x = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int y = __shfl_up(x, d, WAVE);
        if (lane >= d) x += y;
    }
    total = __shfl(x, WAVE - 1, WAVE);
    return x - v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)v, d, WAVE);
        v = y < v ? y : v;
    }
    return v;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// bpe_ranks.get(pair, inf)  (tokenize.py:70-71)
__device__ __forceinline__ uint32_t probe_rank(const GzDeviceTables& T, uint32_t a, uint32_t b)
{
    if ((a | b) & 0xFFF00000u) return GZ_RANK_NONE;          // a code point outside every table never merges
    uint32_t h = gz_pair_hash(a, b) & T.pair_mask;
    const uint64_t key = gz_pair_key(a, b);
    for (;;) {
        uint64_t e = T.pair_tab[h].keyrank;
        if ((e >> 24) == key) return (uint32_t)e & 0xFFFFFFu;
        if (e == GZ_PAIR_EMPTY) return GZ_RANK_NONE;
        h = (h + 1) & T.pair_mask;
    }
}

// the same probe returning the merged symbol too (one 16-byte load)
__device__ __forceinline__ uint32_t probe_pair(const GzDeviceTables& T, uint32_t a, uint32_t b, uint32_t& merged)
{
    if ((a | b) & 0xFFF00000u) return GZ_RANK_NONE;
    uint32_t h = gz_pair_hash(a, b) & T.pair_mask;
    const uint64_t key = gz_pair_key(a, b);
    for (;;) {
        const uint4 v = *reinterpret_cast<const uint4*>(&T.pair_tab[h]);
        const uint64_t e = ((uint64_t)v.y << 32) | v.x;
        if ((e >> 24) == key) { merged = v.z; return v.x & 0xFFFFFFu; }
        if (e == GZ_PAIR_EMPTY) return GZ_RANK_NONE;
        h = (h + 1) & T.pair_mask;
    }
}

// minimum over the 16 lanes of a DPP row (quad swaps, then half-row and row mirrors)
__device__ __forceinline__ uint32_t row16_min(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = t < v ? t : v;   // quad_perm [1,0,3,2]
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = t < v ? t : v;   // quad_perm [2,3,0,1]
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); v = t < v ? t : v;  // row_half_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); v = t < v ? t : v;  // row_mirror
    return v;
}

// tuple(token) / word[-1] + "</w>"  (tokenize.py:63-64): code point -> initial symbol
__device__ __forceinline__ uint32_t initial_symbol(const GzDeviceTables& T, uint32_t cp, bool last)
{
    uint32_t s = GZ_NO_SYMBOL;
    if (cp < 0x10000u) {
        GzCpSyms e = T.bmp[cp];
        s = last ? e.final_ : e.plain;
    } else if (T.astral != nullptr) {
        uint32_t h = gz_cp_hash(cp) & T.astral_mask;
        for (;;) {
            GzAstral e = T.astral[h];
            if (e.cp == cp) { s = last ? e.final_ : e.plain; break; }
            if (e.cp == GZ_NO_SYMBOL) break;
            h = (h + 1) & T.astral_mask;
        }
    }
    return s == GZ_NO_SYMBOL ? (GZ_SYM_UNKNOWN | cp) : s;
}

// encoder.get(piece, encoder.get(unk))  (tokenize.py:120-121)
__device__ __forceinline__ int32_t token_id(const GzDeviceTables& T, uint32_t s, bool final_piece)
{
    if (s & GZ_SYM_UNKNOWN) return T.unk_id;
    GzSymIds e = T.sym_ids[s];
    return final_piece ? e.final_ : e.nonfinal;
}

// Structural UTF-8 decode of the code point whose lead byte is at index i; never reads at or past `end`.
template <class ByteAt>
__device__ __forceinline__ uint32_t decode_cp(ByteAt&& at, int64_t i, int64_t end, int& len)
{
    uint32_t b0 = at(i);
    int want = b0 < 0x80 ? 1 : b0 < 0xE0 ? 2 : b0 < 0xF0 ? 3 : 4;
    if (i + want > end) want = (int)(end - i);
    len = want;
    if (b0 < 0x80) return b0;
    uint32_t cp = want == 2 ? (b0 & 0x1F) : want == 3 ? (b0 & 0x0F) : want == 4 ? (b0 & 0x07) : (b0 & 0x3F);
    for (int k = 1; k < want; ++k) cp = (cp << 6) | (at(i + k) & 0x3F);
    return cp;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-wave encoder state
// ---------------------------------------------------------------------------------------------------------------
struct Emit {
    int32_t* ids;        // destination row (dense: final input_ids row; ragged: raw row in the workspace)
    int32_t* mask;       // dense only
    int32_t* symout;     // gz_bpe_word: raw symbols instead of ids
    int limit;           // positions >= limit are dropped (dense: max_len-1)
    int stop;            // stop tokenizing once ntok >= stop (dense: max_len)
    int ntok;            // wave-uniform count of raw tokens so far
    int32_t pad_id;
    uint8_t* pad_hit;    // set to 1 when an emitted id equals the pad id (nullptr: not tracked)
};

__device__ __forceinline__ void emit_at(const Emit& E, int pos, int32_t id)
{
    if (pos < E.limit) {
        E.ids[pos] = id;
        if (E.mask) E.mask[pos] = id != E.pad_id ? 1 : 0;
        if (E.pad_hit && id == E.pad_id) *E.pad_hit = 1;
    }
}

__device__ __forceinline__ void emit_uniform(Emit& E, int32_t id, int lane)
{
    if (lane == 0) emit_at(E, E.ntok, id);
    E.ntok += 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-cooperative merge loop over S[0..n) (LDS or global scratch).  tokenize.py:69-98.
// ---------------------------------------------------------------------------------------------------------------
__device__ __noinline__ int wave_merge(const GzDeviceTables* Tp, uint32_t* S, int n, int lane, bool global_scratch)
{
    const GzDeviceTables& T = *Tp;
    while (n > 1) {
        uint32_t best = GZ_RANK_NONE;
        for (int i = lane; i < n - 1; i += WAVE) {
            uint32_t r = probe_rank(T, S[i], S[i + 1]);
            best = r < best ? r : best;
        }
        best = wave_min_u32(best);                         // min(pairs, key=rank)
        if (best == GZ_RANK_NONE) break;                   // "bigram not in self.bpe_ranks"
        const GzMergeInfo mi = T.merges[best];
        int out = 0;
        bool skip0 = false;                                // S[base] is the `second` of a pair merged in the previous chunk
        for (int base = 0; base < n; base += WAVE) {
            const int i = base + lane;
            const uint32_t s = i < n ? S[i] : GZ_NO_SYMBOL;
            const uint32_t s1 = i + 1 < n ? S[i + 1] : GZ_NO_SYMBOL;
            const uint64_t valid = __ballot(i < n);
            uint64_t m = __ballot(i + 1 < n && s == mi.left && s1 == mi.right);
            if (skip0) m &= ~1ull;
            uint64_t pick = m;
            if (mi.left == mi.right) {                     // overlapping candidates: greedy left to right
                pick = 0;
                uint64_t rem = m;
                while (rem) {
                    uint64_t low = rem & (0 - rem);
                    pick |= low;
                    rem &= ~(low | (low << 1));
                }
            }
            const uint64_t keep = valid & ~((pick << 1) | (skip0 ? 1ull : 0ull));
            skip0 = (pick >> 63) & 1ull;
            const uint32_t val = ((pick >> lane) & 1ull) ? mi.merged : s;
            const int dest = out + __popcll(keep & lt_mask(lane));
            if ((keep >> lane) & 1ull) S[dest] = val;
            out += __popcll(keep);
        }
        n = out;
        if (global_scratch) __threadfence();
    }
    return n;
}

// One word too long for a lane (or for a tile): bytes g[0..nbytes) in global memory, optional glued '\n'.
// Returns the new token count, or -1 when the word needs the global arena and none was given (deferred).
__device__ __noinline__ int long_word(const GzDeviceTables* Tp, uint32_t* lds_scratch, int lds_cap, const uint8_t* g,
                                     int64_t nbytes, bool glue, uint32_t* arena_slot, Emit E, int lane)
{
    const GzDeviceTables& T = *Tp;
    auto at = [&](int64_t i) -> uint32_t { return g[i]; };
    int leads = 0;
    for (int64_t i = lane; i < nbytes; i += WAVE) leads += (g[i] & 0xC0) != 0x80;
    const int64_t nsym64 = (int64_t)wave_sum(leads) + (glue ? 1 : 0);
    uint32_t* S;
    bool global_scratch = false;
    if (nsym64 <= lds_cap) S = lds_scratch;
    else if (arena_slot != nullptr) { S = arena_slot; global_scratch = true; }
    else return -1;
    int n = (int)nsym64;
    int symbase = 0;
    for (int64_t base = 0; base < nbytes; base += WAVE) {
        const int64_t i = base + lane;
        const bool lead = i < nbytes && (g[i] & 0xC0) != 0x80;
        const uint64_t m = __ballot(lead);
        if (lead) {
            int len;
            const uint32_t cp = decode_cp(at, i, nbytes, len);
            const int idx = symbase + __popcll(m & lt_mask(lane));
            S[idx] = initial_symbol(T, cp, !glue && idx == n - 1);
        }
        symbase += __popcll(m);
    }
    if (glue && lane == 0) S[n - 1] = initial_symbol(T, 0x0Au, true);
    if (global_scratch) __threadfence();
    if (n > 1) n = wave_merge(Tp, S, n, lane, global_scratch);
    for (int base = 0; base < n; base += WAVE) {
        const int i = base + lane;
        if (i < n) {
            if (E.symout) { if (E.ntok + i < E.limit) E.symout[E.ntok + i] = (int32_t)S[i]; }
            else emit_at(E, E.ntok + i, token_id(T, S[i], i == n - 1));
        }
    }
    return E.ntok + n;
}

// ---------------------------------------------------------------------------------------------------------------
// Tile classification: lane owns bytes [16*lane, 16*lane+16) of the tile (w[0..3]) and sees 4 more (w4).
//   start bit j : a word starts at byte 16*lane+j   (non-whitespace lead byte whose previous byte is whitespace)
//   end   bit j : a word ends before byte 16*lane+j (whitespace lead byte whose previous byte is not whitespace)
// `prev_ws0`: the byte before the tile counts as whitespace (tile begins at a word boundary); then leading
// continuation bytes (tail of a whitespace code point cut by the previous tile) count as whitespace too.
// `brk16`: bit j set = a document starts at byte 16*lane+j.
// ---------------------------------------------------------------------------------------------------------------
// SWAR helpers on 4 packed bytes: results have bit 7 of a byte set where the predicate holds
__device__ __forceinline__ uint32_t swar_eq(uint32_t x, uint32_t c)
{
    const uint32_t t = x ^ (c * 0x01010101u);
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
}
__device__ __forceinline__ uint32_t swar_ge7(uint32_t x7o, uint32_t c)      // x7o = (x & 0x7F..) | 0x80..;  low 7 bits >= c
{
    return (x7o - c * 0x01010101u) & 0x80808080u;
}
__device__ __forceinline__ uint32_t swar_bits(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3)   // MSB masks -> 16 bits
{
    const uint32_t n0 = ((m0 >> 7) * 0x00204081u) >> 21 & 0xFu, n1 = ((m1 >> 7) * 0x00204081u) >> 21 & 0xFu;
    const uint32_t n2 = ((m2 >> 7) * 0x00204081u) >> 21 & 0xFu, n3 = ((m3 >> 7) * 0x00204081u) >> 21 & 0xFu;
    return n0 | (n1 << 4) | (n2 << 8) | (n3 << 12);
}

__device__ __forceinline__ void classify(const uint32_t w[4], uint32_t w4, int lane, bool prev_ws0, uint32_t brk16,
                                         uint32_t& start16, uint32_t& end16)
{
    uint32_t leadm[4], wsm[4], ws23m[4], c3m[4];
    uint32_t any3 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = w[i], nx = i < 3 ? w[i + 1] : w4;
        const uint32_t y = (x >> 8) | (nx << 24);                         // the following byte, aligned
        const uint32_t hi = x & 0x80808080u, x7o = x | 0x80808080u;
        leadm[i] = ~(x & ~(x << 1)) & 0x80808080u;                        // not 10xxxxxx
        // ASCII whitespace: 09..0D, 1C..20
        const uint32_t a1 = ((swar_ge7(x7o, 0x09u) & ~swar_ge7(x7o, 0x0Eu)) | (swar_ge7(x7o, 0x1Cu) & ~swar_ge7(x7o, 0x21u))) & ~hi;
        // C2 85 / C2 A0
        const uint32_t a2 = swar_eq(x, 0xC2u) & (swar_eq(y, 0x85u) | swar_eq(y, 0xA0u));
        // candidates for the 3-byte ones: E1 9A .. / E2 80|81 .. / E3 80 ..
        const uint32_t eE = swar_ge7(x7o, 0x61u) & ~swar_ge7(x7o, 0x64u) & hi;   // E1..E3
        const uint32_t b0 = (x << 7) & 0x80808080u, b1 = (x << 6) & 0x80808080u;
        const uint32_t y80 = swar_eq(y, 0x80u);
        const uint32_t c3 = eE & ((b0 & ~b1 & swar_eq(y, 0x9Au)) | (~b0 & b1 & (y80 | swar_eq(y, 0x81u))) | (b0 & b1 & y80));
        wsm[i] = a1 | a2;
        ws23m[i] = a2;
        c3m[i] = c3;
        any3 |= c3;
    }
    uint32_t lead = swar_bits(leadm[0], leadm[1], leadm[2], leadm[3]);
    uint32_t ws = swar_bits(wsm[0], wsm[1], wsm[2], wsm[3]);
    uint32_t ws23 = swar_bits(ws23m[0], ws23m[1], ws23m[2], ws23m[3]);
    uint32_t ws3 = 0;
    if (__ballot(any3 != 0)) {                                           // rare: look at the third byte
        uint32_t c = swar_bits(c3m[0], c3m[1], c3m[2], c3m[3]);
        for (; c; c &= c - 1) {
            const int j = __ffs(c) - 1;
            auto byte = [&](int k) -> uint32_t {
                const uint32_t ww = k < 4 ? w[0] : k < 8 ? w[1] : k < 12 ? w[2] : k < 16 ? w[3] : w4;
                return (ww >> ((k & 3) * 8)) & 0xFFu;
            };
            const uint32_t b = byte(j), n1 = byte(j + 1), n2 = byte(j + 2);
            const bool a3 = (b == 0xE2u && ((n1 == 0x80u && ((n2 >= 0x80u && n2 <= 0x8Au) || n2 == 0xA8u || n2 == 0xA9u || n2 == 0xAFu)) ||
                                            (n1 == 0x81u && n2 == 0x9Fu))) ||
                            (b == 0xE1u && n1 == 0x9Au && n2 == 0x80u) || (b == 0xE3u && n1 == 0x80u && n2 == 0x80u);
            ws3 |= (uint32_t)a3 << j;
        }
        ws |= ws3;
        ws23 |= ws3;
    }
    uint32_t full = ws | (ws23 << 1) | (ws3 << 2);           // every byte of a whitespace code point (18 bits)
    uint32_t carry = (uint32_t)__shfl_up((int)(full >> 16), 1, WAVE);
    uint32_t prevbit = (uint32_t)__shfl_up((int)((full >> 15) & 1u), 1, WAVE);
    if (lane == 0) {
        carry = 0;
        prevbit = prev_ws0 ? 1u : 0u;
        if (prev_ws0) {
            const uint32_t c0 = ~lead & 1u, c1 = c0 & (~lead >> 1) & 1u, c2 = c1 & (~lead >> 2) & 1u;
            carry = c0 | (c1 << 1) | (c2 << 2);
        }
    }
    full = (full | carry) & 0xFFFFu;
    const uint32_t prev_ws = ((full << 1) | prevbit) & 0xFFFFu;
    // a document boundary (brk16) ends the word before it and lets a word start at it
    start16 = lead & ~ws & (prev_ws | brk16) & 0xFFFFu;
    end16 = (ws | brk16) & ~prev_ws & 0xFFFFu;
}

// Load the tile [pos, pos+TILE+16) of a document that ends at `end` into registers (bytes past the end read as
// spaces, which terminates the last word and can never be a glued '\n') and into LDS.
__device__ __forceinline__ void load_tile(const uint8_t* base, int64_t pos, int64_t end, int64_t buf_end, int lane,
                                          uint32_t w[4], uint32_t& w4, WaveLds& L)
{
    typedef uint4 __attribute__((aligned(1))) uint4_u;
    const int64_t g = pos + 16 * lane;
    int64_t nv = end - g;
    nv = nv < 0 ? 0 : nv > 16 ? 16 : nv;
    uint32_t r[4] = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
    if (nv > 0) {
        if (g + 16 <= buf_end) {
            typedef unsigned __attribute__((ext_vector_type(4), aligned(1))) v4u_u;
            const auto v = __builtin_nontemporal_load(reinterpret_cast<const v4u_u*>(base + g));
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else {
            for (int k = 0; k < (int)nv; ++k) {
                const uint32_t sh = (k & 3) * 8;
                r[k >> 2] = (r[k >> 2] & ~(0xFFu << sh)) | ((uint32_t)base[g + k] << sh);
            }
        }
        if (nv < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int v = (int)nv - 4 * q;
                v = v < 0 ? 0 : v > 4 ? 4 : v;
                const uint32_t keep = v == 4 ? 0xFFFFFFFFu : ((1u << (8 * v)) - 1u);
                r[q] = (r[q] & keep) | (0x20202020u & ~keep);
            }
        }
    }
    w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; w[3] = r[3];
    uint32_t nxt = (uint32_t)__shfl_down((int)r[0], 1, WAVE);
    if (lane == WAVE - 1) {
        nxt = 0x20202020u;
        const int64_t g2 = pos + TILE;
        for (int k = 0; k < 4; ++k)
            if (g2 + k < end) nxt = (nxt & ~(0xFFu << (8 * k))) | ((uint32_t)base[g2 + k] << (8 * k));
    }
    w4 = nxt;
    uint4* dst = reinterpret_cast<uint4*>(L.bytes);
    dst[lane] = make_uint4(r[0], r[1], r[2], r[3]);
    if (lane == WAVE - 1) L.bytes[TILE / 4] = nxt;
}

// ---------------------------------------------------------------------------------------------------------------
// A wave owns a GROUP of consecutive documents and streams the bytes of the group (text A of every document,
// then text B) tile by tile.  Words of all documents of a tile share the 64 lanes; per word a RECORD is buffered
// (vocab id when the whole-word table answers, otherwise a slot in the miss list), misses are merged 64 at a
// time (one lane per word, tokenize.py:62-101), and a flush turns records into row positions with a segmented
// prefix sum per document (bos/eos framing and truncation: tokenize.py:126-146).
// ---------------------------------------------------------------------------------------------------------------
struct Group {
    const GzDeviceTables* Tp;
    WaveLds& L;
    int lane;
    int g;                    // documents in this group
    int32_t* out;             // dense: input_ids; ragged: raw rows
    int limit, stop;          // positions >= limit are dropped; a document with >= stop tokens needs no more work
    const uint8_t* base;      // text A or text B (absolute offsets index it)
    int64_t B0, B1, buf_end;  // byte range of the group in `base`, end of the whole buffer
    uint32_t* arena;          // nullptr or global scratch indexed like `base`
    int nrec, nmiss;          // buffered records / misses (wave-uniform)
    int ablate;
    const GzWordSlot* words;  // whole-word table or nullptr
    uint32_t word_mask;
};

__device__ __forceinline__ void group_emit(Group& G, int doc, int pos, int32_t id)
{
    if (pos < G.limit && !(G.ablate & 4)) {
        G.out[G.L.row_base[doc] + pos] = id;
        if (id == G.Tp->pad_id) G.L.pad_hit[doc] = 1;
    }
}

// document (index inside the group) that owns relative byte offset a
__device__ __forceinline__ int doc_of(const Group& G, uint32_t a)
{
    int j = 0;
#pragma unroll
    for (int step = GMAX / 2; step >= 1; step >>= 1) {
        const int k = j + step;
        if (k < G.g && G.L.doc_rel[k] <= a) j = k;
    }
    return j;
}

// merge loop of one lane's word held in L.sym[k*64+lane], k < n   (tokenize.py:69-98)
__device__ __forceinline__ int lane_merge(const GzDeviceTables& T, WaveLds& L, int lane, int n)
{
    while (n > 1) {
        uint32_t best = GZ_RANK_NONE;
        uint32_t a = L.sym[lane];
        for (int k = 1; k < n; ++k) {
            const uint32_t b = L.sym[k * WAVE + lane];
            const uint32_t r = probe_rank(T, a, b);
            best = r < best ? r : best;
            a = b;
        }
        if (best == GZ_RANK_NONE) break;
        const GzMergeInfo mi = T.merges[best];
        int j = 0, k = 0;
        while (k < n) {
            const uint32_t s = L.sym[k * WAVE + lane];
            if (k + 1 < n && s == mi.left && L.sym[(k + 1) * WAVE + lane] == mi.right) {
                L.sym[j * WAVE + lane] = mi.merged;
                k += 2;
            } else {
                L.sym[j * WAVE + lane] = s;
                k += 1;
            }
            ++j;
        }
        n = j;
    }
    return n;
}

// Turn the buffered records into tokens in their rows.
__device__ __forceinline__ void group_flush(Group& G)
{
    WaveLds& L = G.L;
    const GzDeviceTables& T = *G.Tp;
    const int lane = G.lane;
    // ---- the misses: decode, merge, look up (one lane per word) ---------------------------------------------------
    // Sixteen lanes (one DPP row) per word, four words per pass: every adjacent pair of a word is probed in the same
    // instruction, the row minimum picks the pair to merge (tokenize.py:70-73), every occurrence merges left to
    // right (:75-92) and the row is compacted through LDS.  One dependent table load per merge iteration.
    // MU such passes are interleaved so that their table loads are in flight together.
    if (G.ablate & 2) { if (lane < G.nmiss) L.miss_ntok[lane] = 1; }
    else for (int c0 = 0; c0 < G.nmiss; c0 += 4 * MU) {
        const int row = lane >> 4, k = lane & 15;
        int m[MU], nb[MU], n[MU], ncp[MU];
        bool mv[MU], glue[MU], act[MU];
        const uint8_t* g[MU];
        uint32_t* S[MU];
        uint32_t leadbits[MU][4], bytev[MU][4];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            m[u] = c0 + 4 * u + row;
            mv[u] = m[u] < G.nmiss;
            g[u] = G.base + G.B0 + (mv[u] ? L.miss_off[m[u]] : 0u);
            nb[u] = mv[u] ? (L.miss_len[m[u]] & 0x7FFF) : 0;
            glue[u] = mv[u] && (L.miss_len[m[u]] >> 15) != 0;
            S[u] = L.sym + 16 * (mv[u] ? m[u] : 0);
        }
        // -- initial symbols: lane k looks at bytes k, k+16, k+32, k+48 of its word
#pragma unroll
        for (int u = 0; u < MU; ++u)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int b = 16 * p + k;
                bytev[u][p] = b < nb[u] ? g[u][b] : 0x80u;
            }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            ncp[u] = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int b = 16 * p + k;
                const uint64_t bal = __ballot(b < nb[u] && (bytev[u][p] & 0xC0u) != 0x80u);
                leadbits[u][p] = (uint32_t)(bal >> (16 * row)) & 0xFFFFu;
                ncp[u] += __popc(leadbits[u][p]);
            }
            n[u] = ncp[u] + (glue[u] ? 1 : 0);
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            int before = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int b = 16 * p + k;
                if (b < nb[u] && (bytev[u][p] & 0xC0u) != 0x80u) {
                    const uint8_t* gg = g[u];
                    auto at = [&](int64_t i) -> uint32_t { return gg[i]; };
                    int len;
                    const uint32_t cp = decode_cp(at, b, nb[u], len);
                    const int idx = before + __popc(leadbits[u][p] & ((1u << k) - 1u));
                    S[u][idx] = initial_symbol(T, cp, !glue[u] && idx == ncp[u] - 1);
                }
                before += __popc(leadbits[u][p]);
            }
            if (mv[u] && glue[u] && k == 0) S[u][n[u] - 1] = initial_symbol(T, 0x0Au, true);
            act[u] = __ballot(mv[u]) != 0;
        }
        // -- merge iterations
        if (!(G.ablate & 1)) for (;;) {
            uint32_t s[MU], s1[MU], h[MU];
            uint4 ent[MU];
            bool vp[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                s[u] = (act[u] && mv[u] && k < n[u]) ? S[u][k] : GZ_NO_SYMBOL;
                s1[u] = (act[u] && mv[u] && k + 1 < n[u]) ? S[u][k + 1] : GZ_NO_SYMBOL;
                vp[u] = act[u] && mv[u] && k + 1 < n[u] && !((s[u] | s1[u]) & 0xFFF00000u);
                h[u] = gz_pair_hash(s[u], s1[u]) & T.pair_mask;
            }
#pragma unroll
            for (int u = 0; u < MU; ++u)
                ent[u] = vp[u] ? *reinterpret_cast<const uint4*>(&T.pair_tab[h[u]]) : make_uint4(~0u, ~0u, 0u, 0u);
            bool any = false;
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                if (!act[u]) continue;
                uint32_t rank = GZ_RANK_NONE, merged = 0;
                if (vp[u]) {
                    const uint64_t key = gz_pair_key(s[u], s1[u]);
                    uint4 v = ent[u];
                    uint32_t hh = h[u];
                    for (;;) {
                        const uint64_t e = ((uint64_t)v.y << 32) | v.x;
                        if ((e >> 24) == key) { merged = v.z; rank = v.x & 0xFFFFFFu; break; }
                        if (e == GZ_PAIR_EMPTY) break;
                        hh = (hh + 1) & T.pair_mask;
                        v = *reinterpret_cast<const uint4*>(&T.pair_tab[hh]);
                    }
                }
                const uint32_t best = row16_min(rank);
                const uint64_t M = __ballot(best != GZ_RANK_NONE && rank == best);
                if (M == 0) { act[u] = false; continue; }           // no row of this pass has a ranked pair left
                any = true;
                uint64_t pick = M;
                if (M & (M << 1)) {                                 // overlapping occurrences (first == second): greedy
                    pick = 0;
                    uint64_t rem = M;
                    while (rem) {
                        const uint64_t low = rem & (0 - rem);
                        pick |= low;
                        rem &= ~(low | (low << 1));
                    }
                }
                const bool picked = (pick >> lane) & 1ull;
                const bool consumed = lane > 0 && ((pick >> (lane - 1)) & 1ull);
                const bool keep = mv[u] && k < n[u] && !consumed;
                const uint32_t keep16 = (uint32_t)(__ballot(keep) >> (16 * row)) & 0xFFFFu;
                if (keep) S[u][__popc(keep16 & ((1u << k) - 1u))] = picked ? merged : s[u];
                n[u] -= __popc((uint32_t)(pick >> (16 * row)) & 0xFFFFu);
            }
            if (!any) break;
        }
        // -- ids
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            if (mv[u] && k < n[u]) S[u][k] = (uint32_t)token_id(T, S[u][k], k == n[u] - 1);
            if (mv[u] && k == 0) L.miss_ntok[m[u]] = (uint8_t)n[u];
        }
    }
    // ---- positions: segmented exclusive prefix sum of the token counts, one segment per document -----------------
    if (!(G.ablate & 32)) for (int c0 = 0; c0 < G.nrec; c0 += WAVE) {
        const int r = c0 + lane;
        const bool valid = r < G.nrec;
        const uint32_t rec = valid ? L.rec[r] : 0u;
        const int doc = (int)((rec >> 26) & 31u);
        const bool miss = (rec & REC_MISS) != 0;
        const uint32_t payload = rec & 0x03FFFFFFu;
        const int cnt = valid ? (miss ? (int)L.miss_ntok[payload] : 1) : 0;
        int total;
        const int excl = wave_excl_sum(cnt, lane, total);
        const int prevdoc = __shfl_up(doc, 1, WAVE);
        const uint64_t heads = __ballot(valid && (lane == 0 || prevdoc != doc));
        const uint64_t below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int f = below ? 63 - __clzll((long long)below) : 0;
        const int seg_excl = excl - __shfl(excl, f, WAVE);
        const int nextdoc = __shfl_down(doc, 1, WAVE);
        const bool tail = valid && (lane == 63 || r + 1 >= G.nrec || nextdoc != doc);
        if (valid) {
            const int pos = L.doc_ntok[doc] + seg_excl;
            if (!miss) group_emit(G, doc, pos, (int32_t)payload);
            else for (int k = 0; k < cnt; ++k) group_emit(G, doc, pos + k, (int32_t)L.sym[16 * payload + k]);
            if (tail) L.doc_ntok[doc] = pos + cnt;
        }
    }
    G.nrec = 0;
    G.nmiss = 0;
}

// a word that does not fit a lane: (everything before it has been flushed) merge it wave-cooperatively, emit directly
__device__ __forceinline__ bool group_long_word(Group& G, int64_t gpos, int64_t nbytes, bool glue)
{
    const int doc = doc_of(G, (uint32_t)(gpos - G.B0));
    Emit E;
    E.ids = G.out + G.L.row_base[doc];
    E.mask = nullptr; E.symout = nullptr;
    E.limit = G.limit; E.stop = G.stop; E.pad_id = G.Tp->pad_id;
    E.pad_hit = &G.L.pad_hit[doc];
    E.ntok = G.L.doc_ntok[doc];
    if (E.ntok < G.stop) {
        const int nt = long_word(G.Tp, G.L.sym, LONGCAP, G.base + gpos, nbytes, glue, G.arena ? G.arena + gpos : nullptr, E, G.lane);
        if (nt < 0) return false;
        if (G.lane == 0) G.L.doc_ntok[doc] = nt;
    }
    return true;
}

// load + classify the tile at `pos` (document boundaries included); word starts / ends as per-lane 16-bit masks
__device__ __forceinline__ void group_tile(Group& G, int64_t pos, bool prev_ws0, uint32_t& st16, uint32_t& en16)
{
    WaveLds& L = G.L;
    const int lane = G.lane;
    uint32_t w[4], w4;
    load_tile(G.base, pos, G.B1, G.buf_end, lane, w, w4, L);
    if (lane < TILE / 32) L.brk[lane] = 0;
    if (lane >= 1 && lane < G.g) {
        const int64_t r = (int64_t)L.doc_rel[lane] - (pos - G.B0);
        if (r > 0 && r < TILE) atomicOr(&L.brk[r >> 5], 1u << (r & 31));
    }
    const uint32_t brk16 = (L.brk[lane >> 1] >> ((lane & 1) * 16)) & 0xFFFFu;
    classify(w, w4, lane, prev_ws0, brk16, st16, en16);
}

// One text (A or B) of every document of the group: __tokenize + ids (tokenize.py:103-133).
// Returns false if the group must be deferred to the arena pass.
__device__ __forceinline__ bool group_text(Group& G)
{
    WaveLds& L = G.L;
    const int lane = G.lane;
    const uint8_t* lds_bytes = reinterpret_cast<const uint8_t*>(L.bytes);
    int64_t pos = G.B0;
    // a giant word (one that fills a whole tile) is handled by the same flush + long-word site as ordinary long
    // words: `giant_len` >= 0 asks for it at the top of the next round iteration
    while (pos < G.B1) {
        uint32_t st16, en16;
        group_tile(G, pos, true, st16, en16);
        if (G.ablate & 16) { st16 = 0; en16 = 0; }
        if (G.ablate & 128) { st16 = (pos + 16 * lane < G.B1) ? 0x1111u : 0; en16 = st16 << 2; }   // fake words, no classify
        int n_starts;
        int sidx = wave_excl_sum(__popc(st16), lane, n_starts);
        for (uint32_t m = st16; m; m &= m - 1) L.wstart[sidx++] = (uint16_t)(16 * lane + __ffs(m) - 1);
        reinterpret_cast<uint16_t*>(L.endmap)[lane] = (uint16_t)en16;
        const uint32_t tile_rel = (uint32_t)(pos - G.B0);
        // the last word of the tile is cut when no word end follows its start
        const uint64_t endlanes = __ballot(en16 != 0);
        int e_last = -1;
        if (endlanes) {
            const int hl = 63 - __clzll((long long)endlanes);
            e_last = 16 * hl + 31 - __clz((int)__shfl((int)en16, hl, WAVE));
        }
        const int s_last = n_starts > 0 ? (int)L.wstart[n_starts - 1] : -1;
        const bool cut_word = n_starts > 0 && e_last <= s_last;
        const int nw = cut_word ? n_starts - 1 : n_starts;      // complete words of this tile

        // a word that fills the whole tile: find its end by scanning forward, then treat it as one long word
        int64_t giant_len = -1;
        bool giant_glue = false;
        if (cut_word && nw == 0 && s_last == 0) {
            int64_t q = pos + TILE;
            int64_t wend_abs = G.B1;
            bool at_break = false;
            while (q < G.B1) {
                uint32_t s2, e2;
                group_tile(G, q, false, s2, e2);
                const uint64_t any = __ballot(e2 != 0);
                if (any) {
                    const int fl = __ffsll((unsigned long long)any) - 1;
                    const uint32_t eb = (uint32_t)__shfl((int)e2, fl, WAVE);
                    const int p = 16 * fl + __ffs(eb) - 1;
                    wend_abs = q + p;
                    at_break = ((L.brk[p >> 5] >> (p & 31)) & 1u) != 0;
                    break;
                }
                q += TILE;
            }
            if (wend_abs > G.B1) wend_abs = G.B1;
            giant_glue = wend_abs < G.B1 && !at_break && G.base[wend_abs] == 0x0A;
            giant_len = wend_abs - pos;
        }

        const int nrounds = giant_len >= 0 ? 1 : (nw + WAVE - 1) / WAVE;
        for (int rd = 0; rd < nrounds; ++rd) {
            const int wi = rd * WAVE + lane;
            bool have = giant_len < 0 && wi < nw;
            const int ws = have ? L.wstart[wi] : 0;
            int we = 0;
            if (have) {                                             // first word end after the start
                int q = (ws + 1) >> 5;
                uint32_t bits = L.endmap[q] & (~0u << ((ws + 1) & 31));
                while (bits == 0) bits = L.endmap[++q];
                we = 32 * q + __ffs(bits) - 1;
            }
            const bool glue = have && lds_bytes[we] == 0x0Au && !((L.brk[we >> 5] >> (we & 31)) & 1u);   // "\S+\n?"
            const int nb = we - ws;
            const uint32_t a_rel = tile_rel + (uint32_t)ws;
            const int doc = have ? doc_of(G, a_rel) : 0;
            if (have && L.doc_ntok[doc] >= G.stop) have = false;       // row already full: the word cannot matter
            bool is_long = false;
            if (have && nb + (glue ? 1 : 0) > MAXSYM) {
                int leads = glue ? 1 : 0;
                for (int i = ws; i < we; ++i) leads += (lds_bytes[i] & 0xC0u) != 0x80u;
                is_long = leads > MAXSYM || nb > 4 * MAXSYM;
            }
            // whole-word table: a plain word of <= 16 bytes whose bpe() is known to be ONE piece needs no merge loop
            bool hit = false;
            uint32_t hit_id = 0;
            if (G.ablate & 64) { hit = have; hit_id = 7; }
            else if (G.words != nullptr && have && !glue && nb <= 16) {
                const uint64_t* q = reinterpret_cast<const uint64_t*>(L.bytes) + (ws >> 3);
                const uint64_t x0 = q[0], x1 = q[1], x2 = q[2];
                const int sh = (ws & 7) * 8;
                uint64_t lo = sh ? (x0 >> sh) | (x1 << (64 - sh)) : x0;
                uint64_t hi = sh ? (x1 >> sh) | (x2 << (64 - sh)) : x1;
                if (nb <= 8) { hi = 0; if (nb < 8) lo &= (1ull << (8 * nb)) - 1ull; }
                else if (nb < 16) hi &= (1ull << (8 * (nb - 8))) - 1ull;
                uint32_t h = gz_word_hash(lo, hi, (uint32_t)nb) & G.word_mask;
                for (;;) {
                    const GzWordSlot e = G.words[h];
                    if (e.len == (uint32_t)nb && e.lo == lo && e.hi == hi) { hit = true; hit_id = (uint32_t)e.id; break; }
                    if (e.len == 0) break;
                    h = (h + 1) & G.word_mask;
                }
            }
            // append in word order; a lane that does not fit (or is long) cuts the round: flush, then go on
            int lo = 0;
            bool need_flush = giant_len >= 0;
            int64_t lw_pos = pos, lw_len = giant_len;            // pending long word (after the flush)
            bool lw_glue = giant_glue;
            for (;;) {
                if (need_flush) {
                    if (G.nrec > 0) group_flush(G);
                    need_flush = false;
                    if (lw_len >= 0) {
                        if (!group_long_word(G, lw_pos, lw_len, lw_glue)) return false;
                        lw_len = -1;
                    }
                    if (giant_len >= 0) break;
                }
                const bool rem = have && lane >= lo;
                const bool missr = rem && !hit && !is_long;
                const uint64_t remb = __ballot(rem), missb = __ballot(missr);
                const int rp = __popcll(remb & lt_mask(lane)), mp = __popcll(missb & lt_mask(lane));
                const bool fits = rem && !is_long && (G.nmiss + mp + (missr ? 1 : 0) <= MISSCAP) && (G.nrec + rp < RECCAP);
                const uint64_t nofit = __ballot(rem && !fits);
                const int cut = nofit ? __ffsll((unsigned long long)nofit) - 1 : WAVE;
                const bool app = rem && lane < cut;
                if (app) {
                    uint32_t rec = (uint32_t)doc << 26;
                    if (missr) {
                        const int m = G.nmiss + mp;
                        L.miss_off[m] = a_rel;
                        L.miss_len[m] = (uint16_t)(nb | (glue ? 0x8000 : 0));
                        rec |= REC_MISS | (uint32_t)m;
                    } else {
                        rec |= hit_id;
                    }
                    L.rec[G.nrec + rp] = rec;
                }
                G.nrec += __popcll(__ballot(app));
                G.nmiss += __popcll(__ballot(app && missr));
                if (cut == WAVE) break;
                need_flush = true;
                if (__shfl((int)is_long, cut, WAVE)) {
                    lw_pos = pos + __shfl(ws, cut, WAVE);
                    lw_len = __shfl(nb, cut, WAVE);
                    lw_glue = __shfl((int)glue, cut, WAVE) != 0;
                    lo = cut + 1;
                } else {
                    lo = cut;
                }
            }
        }

        if (giant_len >= 0) pos += giant_len + (giant_glue ? 1 : 0);
        else if (cut_word) pos += s_last;                           // re-tile at the start of the cut word (> 0 here)
        else pos += TILE;
    }
    return true;
}

}  // namespace

// =================================================================================================================
// gz_encode_kernel
// =================================================================================================================
__device__ __forceinline__ void encode_body(const GzDeviceTables* __restrict__ Tp, const GzEncodeArgs& A)
{
    __shared__ WaveLds lds[WPB];
    const GzDeviceTables& T = *Tp;
    const int lane = lane_id();
    const int wv = threadIdx.x / WAVE;
    const int64_t d0 = ((int64_t)blockIdx.x * WPB + wv) * A.docs_per_wave;
    if (d0 >= A.n_docs) return;
    if (A.huge_pass && A.n_real[d0] != GZ_DEFERRED) return;
    WaveLds& L = lds[wv];
    const int g = (int)(A.n_docs - d0 < A.docs_per_wave ? A.n_docs - d0 : A.docs_per_wave);

    Group G{Tp, L, lane, g, A.dense ? A.ids : A.raw, 0, 0, nullptr, 0, 0, 0, nullptr, 0, 0, A.ablate,
            A.use_words ? T.words : nullptr, T.word_mask};
    if (A.dense) { G.limit = A.max_len - 1; G.stop = A.max_len; }       // position max_len-1 is eos or padding
    else { G.limit = 0x7FFFFFFF; G.stop = 0x7FFFFFFF; }

    // per-document state: row base, bos (tokenize.py:134-135)
    int64_t my_a = 0, my_p = 0;
    if (lane <= g) {
        my_a = A.text_off[d0 + lane];
        if (A.pair) my_p = A.pair_off[d0 + lane];
    }
    if (lane < g) {
        const int64_t d = d0 + lane;
        int64_t rb;
        if (A.dense) rb = d * (int64_t)A.max_len;
        else rb = (my_a - A.text_off[0]) + 2 * d + (A.pair ? (my_p - A.pair_off[0]) + 2 * d : 0);
        L.row_base[lane] = rb;
        L.pad_hit[lane] = 0;
        L.doc_ntok[lane] = 1;
        if (0 < G.limit) { G.out[rb] = T.bos_id; if (T.bos_id == T.pad_id) L.pad_hit[lane] = 1; }
    }

    bool ok = true;
    const int ntexts = A.pair ? 2 : 1;
    for (int tx = 0; tx < ntexts && ok; ++tx) {
        const int64_t* off = tx ? A.pair_off : A.text_off;
        const int64_t beg = off[d0];
        G.base = tx ? A.pair : A.text;
        G.B0 = beg; G.B1 = off[d0 + g]; G.buf_end = off[A.n_docs];
        G.arena = nullptr;
        if (A.arena) G.arena = tx ? A.arena + (A.text_off[A.n_docs] - A.text_off[0]) - A.pair_off[0] : A.arena - A.text_off[0];
        if (lane <= g) L.doc_rel[lane] = (uint32_t)((tx ? my_p : my_a) - beg);
        ok = group_text(G);
        if (!ok) break;
        if (G.nrec > 0) group_flush(G);
        // A eos [eos B eos]   (tokenize.py:134-135, :237-239)
        const int reps = (A.pair && tx == 0) ? 2 : 1;
        if (lane < g) {
            const int p = L.doc_ntok[lane];
            for (int k = 0; k < reps; ++k) group_emit(G, lane, p + k, T.eos_id);
            L.doc_ntok[lane] = p + reps;
        }
    }
    if (!ok) {
        if (lane < g) A.n_real[d0 + lane] = GZ_DEFERRED;
        if (lane == 0) *A.defer_flag = 1;
        return;
    }
    if (!A.dense) {
        if (lane < g) A.n_real[d0 + lane] = L.doc_ntok[lane];
        return;
    }
    // ---- __padding (tokenize.py:141-146) + attention mask (:148-152), one row at a time, 16-byte stores ------------
    const int Lm = A.max_len;
    if (A.ablate & 8) return;
    for (int j = 0; j < g; ++j) {
        const int Tn = L.doc_ntok[j];
        const int t = Tn < Lm ? Tn : Lm;
        const bool cut = Tn >= Lm;
        int32_t* ids = A.ids + L.row_base[j];
        int32_t* mask = A.mask + L.row_base[j];
        const int32_t tailv = cut ? T.eos_id : T.pad_id;
        const int first = cut ? Lm - 1 : t;                                 // first position not holding a raw token
        const bool slow = L.pad_hit[j] != 0 || (cut && T.eos_id == T.pad_id);
        if ((Lm & 3) == 0) {
            for (int c = lane; c < Lm / 4; c += WAVE) {
                const int i0 = 4 * c;
                if (i0 >= first) nt_store4(ids + i0, tailv, tailv, tailv, tailv);
                else if (i0 + 4 > first) for (int i = first; i < i0 + 4; ++i) ids[i] = tailv;
                if (!slow) nt_store4(mask + i0, i0 < t, i0 + 1 < t, i0 + 2 < t, i0 + 3 < t);
            }
        } else {
            for (int i = lane; i < Lm; i += WAVE) {
                if (i >= first) ids[i] = tailv;
                if (!slow) mask[i] = i < t ? 1 : 0;
            }
        }
        if (slow) {                                                         // a real token equals the pad id
            __threadfence();
            for (int i = lane; i < Lm; i += WAVE) {
                const int32_t v = __builtin_nontemporal_load(ids + i);
                mask[i] = v != T.pad_id ? 1 : 0;
            }
        }
        if (lane == 0) A.n_real[d0 + j] = t;
    }
}

__global__ __launch_bounds__(WAVE * WPB) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gz_encode_kernel(const GzDeviceTables* __restrict__ Tp, GzEncodeArgs A) { encode_body(Tp, A); }

// the same code under its own name for the load-time build of the whole-word table, so that profiles of the hot
// path are not mixed with those two tiny launches
__global__ __launch_bounds__(WAVE * WPB) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gz_encode_kernel_tablebuild(const GzDeviceTables* __restrict__ Tp, GzEncodeArgs A) { encode_body(Tp, A); }

// =================================================================================================================
// Ragged layouts: row length, scan, finalize
// =================================================================================================================
__device__ __forceinline__ int cut_len(int n, int max_len)            // len(seq[:max_len-1])
{
    const int stop = max_len - 1;
    if (stop >= 0) return n < stop ? n : stop;
    const int k = n + stop;
    return k > 0 ? k : 0;
}

__device__ __forceinline__ int padded_len(int n, const GzShape& S)   // len(__padding(seq)) when it applies
{
    if (!S.pad_mode) return n;
    if (n < S.max_len) return S.max_len;
    if (S.truncation) return cut_len(n, S.max_len) + 1;
    return n;
}

__global__ void gz_rowlen_kernel(const int32_t* n_raw, int64_t n_docs, GzShape S, int64_t* row_len)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_docs) row_len[d] = padded_len(n_raw[d], S);
}

// single-workgroup exclusive scan of int64 (ragged layouts only; N/1024 trips)
__global__ __launch_bounds__(1024) void gz_scan_kernel(const int64_t* in, int64_t n, int64_t* out /* n+1 */)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n ? in[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
            const int64_t y = __shfl_up(x, dlt, WAVE);
            if (lane >= dlt) x += y;
        }
        if (lane == WAVE - 1) wsum[wv] = x;
        __syncthreads();
        int64_t pre = carry;
        for (int k = 0; k < wv; ++k) pre += wsum[k];
        if (i < n) out[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ __launch_bounds__(WAVE * WPB) void gz_finalize_kernel(GzDeviceTables T, GzFinalizeArgs F)
{
    const int lane = lane_id();
    const int64_t d = (int64_t)blockIdx.x * WPB + threadIdx.x / WAVE;
    if (d >= F.n_docs) return;
    if (F.row_off[F.n_docs] > F.capacity) {
        if (d == 0 && lane == 0) *F.error_flag = 1;
        return;
    }
    const int64_t ro = (F.text_off[d] - F.text_off[0]) + 2 * d + (F.pair_off ? (F.pair_off[d] - F.pair_off[0]) + 2 * d : 0);
    const int32_t* raw = F.raw + ro;
    const int n = F.n_raw[d];
    const int64_t o = F.row_off[d];
    const int R = (int)(F.row_off[d + 1] - o);
    const bool padded = F.S.pad_mode && n < F.S.max_len;
    const bool cut = F.S.pad_mode && !padded && F.S.truncation;
    const int kept = cut ? R - 1 : n;
    for (int i = lane; i < R; i += WAVE) {
        int32_t v;
        if (i < kept) v = raw[i];
        else v = cut ? T.eos_id : T.pad_id;
        F.ids[o + i] = v;
        F.mask[o + i] = v != T.pad_id ? 1 : 0;
    }
    if (lane == 0) F.n_real[d] = padded ? n : R;
}

// =================================================================================================================
// gz_pair_kernel: get_sequence_id (tokenize.py:163-182), get_token_type (:154-161), __padding of it (:256-258)
// =================================================================================================================
__global__ __launch_bounds__(WAVE * WPB) void gz_pair_kernel(GzDeviceTables T, GzPairArgs P)
{
    const int lane = lane_id();
    const int64_t d = (int64_t)blockIdx.x * WPB + threadIdx.x / WAVE;
    if (d >= P.n_docs) return;
    if (P.row_off && P.row_off[P.n_docs] > P.capacity) return;
    const int64_t o = P.row_off ? P.row_off[d] : d * (int64_t)P.S.max_len;
    const int R = P.row_off ? (int)(P.row_off[d + 1] - o) : P.S.max_len;
    const int32_t* ids = P.ids + o;
    const int32_t eos = T.eos_id, bos = T.bos_id;

    // p1: first eos (the loop at :168-173 stops there)
    int p1 = R;
    for (int base = 0; base < R; base += WAVE) {
        const int i = base + lane;
        const uint64_t m = __ballot(i < R && ids[i] == eos);
        if (m) { p1 = base + __ffsll((unsigned long long)m) - 1; break; }
    }
    // p2: first eos after p1 that directly follows a `1` entry (:176-179) -> the list ends there
    int seq_len = R;
    for (int base = (p1 + 2) & ~(WAVE - 1); base < R; base += WAVE) {
        const int i = base + lane;
        const bool hit = i < R && i >= p1 + 2 && ids[i] == eos && ids[i - 1] != eos;
        const uint64_t m = __ballot(hit);
        if (m) { seq_len = base + __ffsll((unsigned long long)m) - 1 + 1; break; }
    }
    auto raw_val = [&](int i) -> int32_t {                     // the list before get_token_type touches it
        const int32_t v = ids[i];
        if (i < p1) return v == bos ? GZ_NONE_ : 0;
        if (i == p1) return GZ_NONE_;
        return v == eos ? GZ_NONE_ : 1;
    };
    // get_token_type: [0]=0, [-1]=1, then the first two remaining None -> 0, 1
    int n1 = -1, n2 = -1;
    for (int base = 0; base < seq_len && n2 < 0; base += WAVE) {
        const int i = base + lane;
        const bool none = i > 0 && i < seq_len - 1 && raw_val(i) == GZ_NONE_;
        uint64_t m = __ballot(none);
        while (m && n2 < 0) {
            const int p = base + __ffsll((unsigned long long)m) - 1;
            if (n1 < 0) n1 = p; else n2 = p;
            m &= m - 1;
        }
    }
    const bool bad = n2 < 0;                                   // list.index(None) raises ValueError
    if (lane == 0) P.status[d] = bad ? 1 : 0;
    auto final_val = [&](int i) -> int32_t {
        if (i == seq_len - 1) return 1;
        if (i == 0) return 0;
        if (i == n1) return 0;
        if (i == n2) return 1;
        return raw_val(i);
    };
    int tt_len = seq_len;
    bool tt_pad = false, tt_cut = false;
    if (P.S.pad_mode) {
        if (seq_len < P.S.max_len) { tt_len = P.S.max_len; tt_pad = true; }
        else if (P.S.truncation) { tt_len = cut_len(seq_len, P.S.max_len) + 1; tt_cut = true; }
    }
    if (lane == 0) { P.pair_len[2 * d] = bad ? 0 : seq_len; P.pair_len[2 * d + 1] = bad ? 0 : tt_len; }
    if (bad) return;
    for (int i = lane; i < seq_len; i += WAVE) P.seq[o + i] = final_val(i);
    const int kept = tt_cut ? tt_len - 1 : seq_len;
    for (int i = lane; i < tt_len; i += WAVE) {
        int32_t v;
        if (i < kept) v = final_val(i);
        else v = tt_cut ? eos : T.pad_id;
        (void)tt_pad;
        P.tt[o + i] = v;
    }
}

// =================================================================================================================
// gz_bpe_word_kernel: Tokenize.bpe(token) -- the whole input is ONE word (no whitespace split, no glue)
// =================================================================================================================
__global__ __launch_bounds__(WAVE) void gz_bpe_word_kernel(const GzDeviceTables* __restrict__ Tp, const uint8_t* word, int64_t nbytes,
                                                            uint32_t* arena, int32_t* out, int32_t cap, int32_t* n_out)
{
    __shared__ WaveLds L;
    const int lane = lane_id();
    Emit E;
    E.ids = nullptr; E.mask = nullptr; E.pad_hit = nullptr; E.symout = out; E.limit = cap; E.stop = 0x7FFFFFFF; E.ntok = 0; E.pad_id = Tp->pad_id;
    const int nt = long_word(Tp, L.sym, LONGCAP, word, nbytes, false, arena, E, lane);
    if (lane == 0) *n_out = nt;
}

#include "gz_pipeline.inc"

// =================================================================================================================
// launchers
// =================================================================================================================
void gz_launch_encode(const GzDeviceTables* T, const GzEncodeArgs& A, hipStream_t s)
{
    const int64_t waves = (A.n_docs + A.docs_per_wave - 1) / A.docs_per_wave;
    const int64_t blocks = (waves + WPB - 1) / WPB;
    if (blocks <= 0) return;
    if (A.table_build) hipLaunchKernelGGL(gz_encode_kernel_tablebuild, dim3((unsigned)blocks), dim3(WAVE * WPB), 0, s, T, A);
    else hipLaunchKernelGGL(gz_encode_kernel, dim3((unsigned)blocks), dim3(WAVE * WPB), 0, s, T, A);
}

void gz_launch_rowscan(const GzFinalizeArgs& F, int64_t* row_len_tmp, hipStream_t s)
{
    if (F.n_docs <= 0) return;
    hipLaunchKernelGGL(gz_rowlen_kernel, dim3((unsigned)((F.n_docs + 255) / 256)), dim3(256), 0, s,
                       F.n_raw, F.n_docs, F.S, row_len_tmp);
    hipLaunchKernelGGL(gz_scan_kernel, dim3(1), dim3(1024), 0, s, (const int64_t*)row_len_tmp, F.n_docs, F.row_off);
}

void gz_launch_finalize(const GzDeviceTables& T, const GzFinalizeArgs& F, hipStream_t s)
{
    if (F.n_docs <= 0) return;
    hipLaunchKernelGGL(gz_finalize_kernel, dim3((unsigned)((F.n_docs + WPB - 1) / WPB)), dim3(WAVE * WPB), 0, s, T, F);
}

void gz_launch_pair(const GzDeviceTables& T, const GzPairArgs& P, hipStream_t s)
{
    if (P.n_docs <= 0) return;
    hipLaunchKernelGGL(gz_pair_kernel, dim3((unsigned)((P.n_docs + WPB - 1) / WPB)), dim3(WAVE * WPB), 0, s, T, P);
}

void gz_launch_bpe_word(const GzDeviceTables* T, const uint8_t* word, int64_t nbytes, uint32_t* arena,
                        int32_t* out, int32_t cap, int32_t* n_out, hipStream_t s)
{
    hipLaunchKernelGGL(gz_bpe_word_kernel, dim3(1), dim3(WAVE), 0, s, T, word, nbytes, arena, out, cap, n_out);
}
