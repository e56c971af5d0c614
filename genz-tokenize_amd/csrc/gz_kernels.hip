// HIP kernels of the genz-tokenize hot path for gfx950 (MI355X, wave64).
//
// This file holds the device helpers shared by every kernel (table probes, UTF-8 decode, SWAR byte classes, the
// wave-cooperative long-word merge) and the small kernels around the pipeline; the pipeline itself -- classify ->
// scan -> words -> misses -> assemble -- is in gz_pipeline.inc (included below, same translation unit).
//
//   gz_rowlen_kernel / gz_scan[64]_kernel   ragged layouts only (padding=False, truncation=False, max_len None or < 1): row lengths,
//                        their exclusive scan = the rows' places;
//   gz_finalize_kernel   ... and the copy out of the raw area + pad/cut (tokenize.py:141-146) for pairs and the padded ragged shapes
//                        (rows without padding of single texts are written straight to their places: gz_rowsr_kernel, gz_pipeline.inc).
//   gz_scan32x_*_kernel  the 32-bit row offsets of an exchange block / of the CSR host path, over many workgroups.
//   gz_pair_kernel       sequence_id / token_type_ids of sentence pairs (tokenize.py:154-182, :252-258).
//   gz_bpe_word_kernel   Tokenize.bpe(token) for one word (symbols out).
//
// Integer / byte work only: no MFMA.  Everything is bit-exact with the reference by construction; see DESIGN.md.
#include "gz_kernels.h"

namespace {

constexpr int WAVE = 64;
constexpr int WPB = GZ_WAVES_PER_BLOCK;      // waves per workgroup of the small kernels (finalize, pair)
constexpr int TILE = 1024;                   // bytes classified per tile: 16 per lane
constexpr int MAXWORDS = TILE / 2 + 1;       // a word needs >= 1 byte + >= 1 whitespace byte
constexpr int GMAX = GZ_MAX_DOCS_PER_WAVE;   // documents per wave of the assemble kernel

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Index assertions of the diagnostic build (-DGZ_DIAG): every scattered access of the pipeline whose index comes out of memory is
// checked against the size of the array it goes into; a failed check is RECORDED (the first one: its code, the index, the bound,
// the workgroup; and the number of failures) and the access is skipped -- the host reads the record when it synchronises
// (gz_diag_check, sync_locked) and fails the call.  In the product build GZ_CHK is the constant `true` and compiles to nothing.
// Codes: 1xx word kernel, 2xx merge pre-pass, 3xx merge kernel, 4xx wide / long kernels, 5xx row kernels, 6xx docw0.
#ifdef GZ_DIAG
__device__ unsigned int gz_diag_err[8];
__device__ __noinline__ bool gz_chk_fail(unsigned code, unsigned long long v, unsigned long long bound)
{
    if (atomicCAS(&gz_diag_err[0], 0u, code) == 0u) {
        gz_diag_err[1] = (unsigned)v; gz_diag_err[2] = (unsigned)(v >> 32); gz_diag_err[3] = (unsigned)bound; gz_diag_err[5] = blockIdx.x;
    }
    atomicAdd(&gz_diag_err[4], 1u);
    return false;
}
#define GZ_CHK(code, v, bound) (((unsigned long long)(v) < (unsigned long long)(bound)) ? true : gz_chk_fail((code), (unsigned long long)(v), (unsigned long long)(bound)))
#else
#define GZ_CHK(code, v, bound) true
#endif

// In-kernel stamps (diagnostic builds only, -DGZ_DIAG: tools/ab_diag.sh): cycles per phase, summed over waves
#ifdef GZ_DIAG
__device__ unsigned long long gz_prof[64];
#define PROF_DECL unsigned long long prof_t0 = __builtin_readcyclecounter(), prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t0; prof_t0 = t_; } while (0)
#define PROF_CNT(i, v) do { prof_acc[i] += (unsigned long long)(v); } while (0)
#define PROF_FLUSH(base) do { if (lane_id() == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&gz_prof[(base) + i_], prof_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_CNT(i, v)
#define PROF_FLUSH(base)
#endif

// streaming (write-once) 16-byte store that does not displace the tables from L2
__device__ __forceinline__ void nt_store4(int32_t* p, int32_t a, int32_t b, int32_t c, int32_t d)
{
    typedef int __attribute__((ext_vector_type(4))) v4i;
    v4i v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(p));
}
__device__ __forceinline__ uint64_t lt_mask(int lane) { return (1ull << lane) - 1ull; }
// ballot of a predicate (v_cmp straight into an SGPR pair) and "set bits of m below my lane" (v_mbcnt)
__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int below(uint64_t m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// exclusive prefix sum over the wave, all in DPP (no LDS traffic): Hillis-Steele inside each row of 16 lanes
// (row_shr 1, 2, 4, 8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.
__device__ __forceinline__ int wave_excl_sum(int v, int lane, int& total)
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(x, WAVE - 1);
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

// bpe_ranks.get(pair, inf) (tokenize.py:70-71) and first + second (:88) from the perfectly hashed pair table (gz_common.h): the
// bucket's displacement (16 KB array in memory: the kernels that call this are latency-bound chains, not the big pipeline's
// merge kernel, which stages it in LDS) -> ONE 8-byte load; the key is compared in full.  Returns the rank (GZ_RANK_NONE:
// no such merge) and the merged symbol: the rank itself unless another line spells the same string (alias flag).
// (ldisp: a copy of the displacement array in LDS, or nullptr: read it from memory)
__device__ __forceinline__ uint32_t probe_pair(const GzDeviceTables& T, uint32_t a, uint32_t b, uint32_t& merged, const uint16_t* ldisp = nullptr)
{
    typedef unsigned __attribute__((ext_vector_type(2))) v2u;
    typedef const v2u __attribute__((address_space(1)))* gp8_t;
    typedef const uint16_t __attribute__((address_space(1)))* g16_t;
    typedef const uint16_t __attribute__((address_space(3)))* l16_t;
    if ((a | b) & 0xFFF00000u) return GZ_RANK_NONE;          // a code point outside every table never merges
    const uint32_t bucket = gz_pair_ha(a, b, T.pair_ph.k1, T.pair_ph.k2) >> T.pair_ph.bshift;
    const uint32_t d = ldisp ? ((l16_t)ldisp)[bucket] : ((g16_t)T.pair_ph.disp)[bucket];
    uint32_t h = gz_ph_slot(gz_pair_hb(a, b), d, T.pair_ph.sshift);
    const uint32_t klo = a | (b << 20), khi = b >> 12;
    v2u q = ((gp8_t)T.pair8)[h];
    if (d == GZ_PH_OVERFLOW)                                  // a bucket the builder could not place: its keys sit further along
        while (q.x != 0xFFFFFFFFu && !(q.x == klo && (q.y & 0xFFu) == khi)) {
            h = (h + 1) & T.pair_ph.mask;
            q = ((gp8_t)T.pair8)[h];
        }
    if (!(q.x == klo && (q.y & 0xFFu) == khi)) return GZ_RANK_NONE;
    const uint32_t rank = q.y >> 9;
    merged = (q.y & GZ_PAIR8_ALIAS) ? T.merges[rank].merged : rank;
    return rank;
}
__device__ __forceinline__ uint32_t probe_rank(const GzDeviceTables& T, uint32_t a, uint32_t b)
{
    uint32_t merged;
    return probe_pair(T, a, b, merged);
}

// The whole-word table of words of <= 16 bytes (perfectly hashed, gz_common.h) for the kernels that do not stage its
// displacement array in LDS: key = the word's bytes, zero padded (lo, hi), nb bytes.  True: `id` is what bpe() + the vocab
// lookup yield for this word (tokenize.py:62-121).
__device__ __forceinline__ bool probe_word16(const GzDeviceTables& T, uint64_t lo, uint64_t hi, uint32_t nb, uint32_t& id)
{
    typedef unsigned __attribute__((ext_vector_type(4))) v4u;
    typedef const v4u __attribute__((address_space(1)))* gw_t;
    typedef const uint32_t __attribute__((address_space(1)))* g32_t;
    typedef const uint16_t __attribute__((address_space(1)))* g16_t;
    const GzPh& P = T.word0_ph;
    const uint32_t d = ((g16_t)P.disp)[gz_word1_ha(lo, hi, nb, P.k1, P.k2) >> P.bshift];
    uint32_t h = gz_ph_slot(gz_word1_hb(lo, hi, nb, P.k1, P.k2), d, P.sshift);
    const uint8_t* tab = reinterpret_cast<const uint8_t*>(T.words0p);
    for (;;) {
        const v4u e = *(gw_t)(tab + ((size_t)h << 5));       // key bytes 0..11, len | id << 5
        const uint32_t k3 = *(g32_t)(tab + ((size_t)h << 5) + 16);
        if ((e.w & 31u) == nb && e.x == (uint32_t)lo && e.y == (uint32_t)(lo >> 32) && e.z == (uint32_t)hi && k3 == (uint32_t)(hi >> 32)) {
            id = e.w >> 5;
            return true;
        }
        if (d != GZ_PH_OVERFLOW || e.w == 0u) return false;    // (an overflow bucket's keys sit further along, by linear probing)
        h = (h + 1) & P.mask;
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
    // (the table pointers come out of a struct in memory: say that they are global, or every lookup is a flat load)
    typedef unsigned __attribute__((ext_vector_type(2))) v2u;
    typedef unsigned __attribute__((ext_vector_type(4))) v4u;
    typedef const v2u __attribute__((address_space(1)))* bmp_t;          // GzCpSyms  {plain, final_}
    typedef const v4u __attribute__((address_space(1)))* astral_t;       // GzAstral  {cp, plain, final_, pad}
    uint32_t s = GZ_NO_SYMBOL;
    if (cp < 0x10000u) {
        const v2u e = ((bmp_t)T.bmp)[cp];
        s = last ? e.y : e.x;
    } else if (T.astral != nullptr) {
        uint32_t h = gz_cp_hash(cp) & T.astral_mask;
        for (;;) {
            const v4u e = ((astral_t)T.astral)[h];
            if (e.x == cp) { s = last ? e.z : e.y; break; }
            if (e.x == GZ_NO_SYMBOL) break;
            h = (h + 1) & T.astral_mask;
        }
    }
    return s == GZ_NO_SYMBOL ? (GZ_SYM_UNKNOWN | cp) : s;
}

// encoder.get(piece, encoder.get(unk))  (tokenize.py:120-121)
__device__ __forceinline__ int32_t token_id(const GzDeviceTables& T, uint32_t s, bool final_piece)
{
    typedef int __attribute__((ext_vector_type(2))) v2i;
    typedef const v2i __attribute__((address_space(1)))* symids_t;       // GzSymIds  {nonfinal, final_}
    if (s & GZ_SYM_UNKNOWN) return T.unk_id;
    const v2i e = ((symids_t)T.sym_ids)[s];
    return final_piece ? e.y : e.x;
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
            const uint64_t valid = wballot(i < n);
            uint64_t m = wballot(i + 1 < n && s == mi.left && s1 == mi.right);
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
            const int dest = out + below(keep);
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
        const uint64_t m = wballot(lead);
        if (lead) {
            int len;
            const uint32_t cp = decode_cp(at, i, nbytes, len);
            const int idx = symbase + below(m);
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
// Byte classification of a tile: lane owns bytes [16*lane, 16*lane+16) (w[0..3]) and sees 4 more (w4).
//   start bit j : a word starts at byte 16*lane+j   (non-whitespace lead byte whose previous byte is whitespace)
//   end   bit j : a word ends before byte 16*lane+j (whitespace lead byte whose previous byte is not whitespace)
// The streaming form that uses these helpers is classify_stream() in gz_pipeline.inc.
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

}  // namespace

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
    // 2 elements per thread and trip, one barrier per trip (double-buffered wave sums, the running carry lives in
    // every thread)
    __shared__ int64_t wsum[2][16];
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    int64_t carry = 0;
    int buf = 0;
    for (int64_t base = 0; base < n; base += 2048, buf ^= 1) {
        const int64_t i = base + 2 * (int64_t)threadIdx.x;
        const int64_t a0 = i < n ? in[i] : 0, a1 = i + 1 < n ? in[i + 1] : 0;
        // in-wave inclusive scan of the 64-bit pair sums
        int64_t x = a0 + a1;
        const int64_t mine = x;
#pragma unroll
        for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
            const int64_t y = __shfl_up(x, dlt, WAVE);
            if (lane >= dlt) x += y;
        }
        if (lane == WAVE - 1) wsum[buf][wv] = x;
        __syncthreads();
        int64_t pre = carry, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int64_t w = wsum[buf][k]; if (k < wv) pre += w; all += w; }
        carry += all;
        const int64_t e0 = pre + x - mine;
        if (i < n) out[i] = e0;
        if (i + 1 < n) out[i + 1] = e0 + a0;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// The same scan over many workgroups, for large n (row offsets of ragged layouts, batch decode, the pre-pass's general tail: the
// single workgroup above takes 0.69 ms for 1 M rows).  No workspace: the problem has n + 1 elements (the last one 0, so that
// out[n] = the total); (1) every workgroup scans its SC64 elements locally and leaves its total in the NEXT block's first slot
// (whose own local offset is 0 by definition); (2) one workgroup turns those first slots into running bases; (3) every other
// slot adds its block's base.
constexpr int SC64 = 2048;
__global__ __launch_bounds__(1024) void gz_scan64_local_kernel(const int64_t* in, int64_t n, int64_t* out /* n+1 */)
{
    __shared__ int64_t wsum[16];
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    const int64_t b0 = (int64_t)blockIdx.x * SC64, i = b0 + 2 * (int64_t)threadIdx.x;
    const int64_t a0 = i < n ? in[i] : 0, a1 = i + 1 < n ? in[i + 1] : 0;
    int64_t x = a0 + a1;
    const int64_t mine = x;
#pragma unroll
    for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
        const int64_t y = __shfl_up(x, dlt, WAVE);
        if (lane >= dlt) x += y;
    }
    if (lane == WAVE - 1) wsum[wv] = x;
    __syncthreads();
    int64_t pre = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int64_t w = wsum[k]; if (k < wv) pre += w; all += w; }
    const int64_t e0 = pre + x - mine;
    if (i <= n && i != b0) out[i] = e0;                          // (slot b0 belongs to the block before: its total; block 0: zero)
    if (i + 1 <= n) out[i + 1] = e0 + a0;
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) out[0] = 0;
        if (b0 + SC64 <= n) out[b0 + SC64] = all;
    }
}
__global__ __launch_bounds__(1024) void gz_scan64_bases_kernel(int64_t* out, int64_t n)
{
    __shared__ int64_t wsum[2][16];
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    const int64_t nb = n / SC64 + 1;                             // blocks of the (n + 1)-element problem
    int64_t carry = 0;
    int buf = 0;
    for (int64_t base = 0; base < nb; base += 1024, buf ^= 1) {
        const int64_t b = base + threadIdx.x;
        int64_t x = b < nb ? out[b * SC64] : 0;
#pragma unroll
        for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
            const int64_t y = __shfl_up(x, dlt, WAVE);
            if (lane >= dlt) x += y;
        }
        if (lane == WAVE - 1) wsum[buf][wv] = x;
        __syncthreads();
        int64_t pre = carry, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int64_t w = wsum[buf][k]; if (k < wv) pre += w; all += w; }
        carry += all;
        if (b < nb) out[b * SC64] = pre + x;                     // inclusive: everything before block b
    }
}
__global__ __launch_bounds__(256) void gz_scan64_add_kernel(int64_t* out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n || (i & (SC64 - 1)) == 0) return;
    out[i] += out[i & ~(int64_t)(SC64 - 1)];
}
// The same three steps for 32-bit values (row lengths -> the 32-bit row offsets of the exchange step's blocks and of the CSR host path:
// the single workgroup of gz_scan32_kernel takes 0.66 ms for the 1.25 M rows of a shard, on the stream the block is waiting on)
constexpr int SC32 = 4096;
__global__ __launch_bounds__(1024) void gz_scan32x_local_kernel(const int32_t* in, int64_t n, uint32_t* out /* n+1 */)
{
    __shared__ uint32_t wsum[16];
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    const int64_t b0 = (int64_t)blockIdx.x * SC32, i = b0 + 4 * (int64_t)threadIdx.x;
    uint32_t a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = i + k < n ? (uint32_t)in[i + k] : 0u;
    uint32_t x = a[0] + a[1] + a[2] + a[3];
    const uint32_t mine = x;
#pragma unroll
    for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, dlt, WAVE);
        if (lane >= dlt) x += y;
    }
    if (lane == WAVE - 1) wsum[wv] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t w = wsum[k]; if (k < wv) pre += w; all += w; }
    uint32_t e = pre + x - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k <= n && i + k != b0) out[i + k] = e;           // (slot b0 belongs to the block before: its total; block 0: zero)
        e += a[k];
    }
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) out[0] = 0;
        if (b0 + SC32 <= n) out[b0 + SC32] = all;
    }
}
__global__ __launch_bounds__(1024) void gz_scan32x_bases_kernel(uint32_t* out, int64_t n)
{
    __shared__ uint32_t wsum[2][16];
    const int lane = lane_id(), wv = threadIdx.x / WAVE;
    const int64_t nb = n / SC32 + 1;                             // blocks of the (n + 1)-element problem
    uint32_t carry = 0;
    int buf = 0;
    for (int64_t base = 0; base < nb; base += 1024, buf ^= 1) {
        const int64_t b = base + threadIdx.x;
        uint32_t x = b < nb ? out[b * SC32] : 0u;
#pragma unroll
        for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, dlt, WAVE);
            if (lane >= dlt) x += y;
        }
        if (lane == WAVE - 1) wsum[buf][wv] = x;
        __syncthreads();
        uint32_t pre = carry, all = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t w = wsum[buf][k]; if (k < wv) pre += w; all += w; }
        carry += all;
        if (b < nb) out[b * SC32] = pre + x;                     // inclusive: everything before block b
    }
}
__global__ __launch_bounds__(256) void gz_scan32x_add_kernel(uint32_t* out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n || (i & (SC32 - 1)) == 0) return;
    out[i] += out[i & ~(int64_t)(SC32 - 1)];
}

// exclusive scan of n int64 values into out[0 .. n] (out[n] = total)
static void launch_scan64(const int64_t* in, int64_t n, int64_t* out, hipStream_t s)
{
    if (n < 8 * SC64) { hipLaunchKernelGGL(gz_scan_kernel, dim3(1), dim3(1024), 0, s, in, n, out); return; }
    hipLaunchKernelGGL(gz_scan64_local_kernel, dim3((unsigned)(n / SC64 + 1)), dim3(1024), 0, s, in, n, out);
    hipLaunchKernelGGL(gz_scan64_bases_kernel, dim3(1), dim3(1024), 0, s, out, n);
    hipLaunchKernelGGL(gz_scan64_add_kernel, dim3((unsigned)(n / 256 + 1)), dim3(256), 0, s, out, n);
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
        const uint64_t m = wballot(i < R && ids[i] == eos);
        if (m) { p1 = base + __ffsll((unsigned long long)m) - 1; break; }
    }
    // p2: first eos after p1 that directly follows a `1` entry (:176-179) -> the list ends there
    int seq_len = R;
    for (int base = (p1 + 2) & ~(WAVE - 1); base < R; base += WAVE) {
        const int i = base + lane;
        const bool hit = i < R && i >= p1 + 2 && ids[i] == eos && ids[i - 1] != eos;
        const uint64_t m = wballot(hit);
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
        uint64_t m = wballot(none);
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
    __shared__ uint32_t lsym[1024];
    const int lane = lane_id();
    Emit E;
    E.ids = nullptr; E.mask = nullptr; E.pad_hit = nullptr; E.symout = out; E.limit = cap; E.stop = 0x7FFFFFFF; E.ntok = 0; E.pad_id = Tp->pad_id;
    const int nt = long_word(Tp, lsym, 1024, word, nbytes, false, arena, E, lane);
    if (lane == 0) *n_out = nt;
}

// off[(s * n_docs) / nsub] for s = 0 .. nsub: the byte positions where a batch is cut into sub-batches
__global__ void gz_pick_kernel(const int64_t* off, const int64_t* off2, int64_t n_docs, int nsub, int64_t* out)
{
    const int s = threadIdx.x;
    if (s > nsub) return;
    const int64_t d = (int64_t)s * n_docs / nsub;
    out[s] = off[d];
    out[nsub + 1 + s] = off2 ? off2[d] : 0;
}

void gz_launch_pick(const int64_t* off, const int64_t* off2, int64_t n_docs, int nsub, int64_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(gz_pick_kernel, dim3(1), dim3(64), 0, s, off, off2, n_docs, nsub, out);
}

#include "gz_pipeline.inc"
#include "gz_hot.inc"
#include "gz_small.inc"
#include "gz_decode.inc"
#include "gz_preproc.inc"

// =================================================================================================================
// launchers
// =================================================================================================================
void gz_launch_rowscan(const GzFinalizeArgs& F, int64_t* row_len_tmp, hipStream_t s)
{
    if (F.n_docs <= 0) return;
    hipLaunchKernelGGL(gz_rowlen_kernel, dim3((unsigned)((F.n_docs + 255) / 256)), dim3(256), 0, s,
                       F.n_raw, F.n_docs, F.S, row_len_tmp);
    launch_scan64((const int64_t*)row_len_tmp, F.n_docs, F.row_off, s);
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

void gz_launch_decode(const GzDecTable& D, const int32_t* ids, const int64_t* row_off, int64_t n_rows, int64_t* row_bytes,
                      int64_t* out_off, uint8_t* out, int64_t capacity, hipStream_t s)
{
    if (n_rows <= 0) return;
    const unsigned grid = (unsigned)((n_rows + 3) / 4);
    if (!out) {
        hipLaunchKernelGGL(gz_decode_kernel, dim3(grid), dim3(WAVE * 4), 0, s, D, ids, row_off, n_rows, row_bytes,
                           (const int64_t*)nullptr, (uint8_t*)nullptr, (int64_t)0);
        launch_scan64((const int64_t*)row_bytes, n_rows, out_off, s);
    } else {
        hipLaunchKernelGGL(gz_decode_kernel, dim3(grid), dim3(WAVE * 4), 0, s, D, ids, row_off, n_rows, (int64_t*)nullptr,
                           (const int64_t*)out_off, out, capacity);
    }
}

// one filter, one pass (0: html look-ahead only; 1: classify + write into the document's slot + new length)
void gz_launch_preprocess(const GzPpArgs& A, int pass, hipStream_t s)
{
    if (A.n_docs <= 0) return;
    const dim3 grid((unsigned)((A.n_docs + 3) / 4)), block(WAVE * 4);
    switch (A.op) {
    case PP_HTML:    hipLaunchKernelGGL(gz_pp_kernel<PP_HTML>, grid, block, 0, s, A, pass); break;
    case PP_UNICODE: hipLaunchKernelGGL(gz_pp_kernel<PP_UNICODE>, grid, block, 0, s, A, pass); break;
    case PP_PUNCT:   hipLaunchKernelGGL(gz_pp_kernel<PP_PUNCT>, grid, block, 0, s, A, pass); break;
    case PP_EMOJI:   hipLaunchKernelGGL(gz_pp_kernel<PP_EMOJI>, grid, block, 0, s, A, pass); break;
    case PP_URL:     hipLaunchKernelGGL(gz_pp_kernel<PP_URL>, grid, block, 0, s, A, pass); break;
    default: break;
    }
}

void gz_launch_scan64(const int64_t* len, int64_t n, int64_t* out_off /* n+1 */, hipStream_t s)
{
    launch_scan64(len, n, out_off, s);
}

void gz_launch_preprocess_fused(const GzPpFusedArgs& A, hipStream_t s)
{
    static_assert(PPF_CAP == GZ_PP_FUSED_MAX_BYTES, "gz_kernels.h and gz_preproc.inc disagree");
    if (A.n_docs <= 0) return;
    hipLaunchKernelGGL(gz_pp_fused_kernel, dim3((unsigned)A.n_docs), dim3(WAVE), 0, s, A);
}

void gz_launch_pp_tail(const uint8_t* slots, const int64_t* in_off, uint32_t* len32, int64_t n_docs, uint8_t* out, int64_t capacity, int64_t* out_off,
                       unsigned long long* lb, uint32_t* ctl, uint32_t epoch, hipStream_t s)
{
    if (n_docs <= 0) return;
    hipLaunchKernelGGL(gz_scan32m_kernel, dim3((unsigned)((n_docs + SCH) / SCH)), dim3(256), 0, s, len32, n_docs, (uint32_t*)nullptr, lb, ctl + 1, epoch,
                       reinterpret_cast<int32_t*>(ctl + 2));
    hipLaunchKernelGGL(gz_pp_pack4_kernel, dim3((unsigned)((n_docs + 4 * PACK_DOCS - 1) / (4 * PACK_DOCS))), dim3(WAVE * 4), 0, s, slots, in_off, (const uint32_t*)len32, n_docs, out, capacity, out_off);
}

void gz_launch_pp_pack(const uint8_t* in, const int64_t* in_off, const int64_t* len, int64_t n_docs, uint8_t* out, const int64_t* out_off, hipStream_t s)
{
    if (n_docs > 0)
        hipLaunchKernelGGL(gz_pp_pack_kernel, dim3((unsigned)((n_docs + 3) / 4)), dim3(WAVE * 4), 0, s, in, in_off, len, n_docs, out, out_off);
}
