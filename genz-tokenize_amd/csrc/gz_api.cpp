// S1 -- the C ABI (include/genz_tokenize.h): context, device tables, workspace, launches, RCCL gather.
// No C++ exception leaves this file; there is no CPU fallback for any entry point that computes.
#include "../../include/genz_tokenize.h"
#include "gz_kernels.h"

#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>
#include <unordered_map>
#include <algorithm>

namespace {

#define g_create_err gz_create_err()     // shared with gz_host_api.cpp (error text of calls that have no context)

struct DBuf {                       // grow-only device buffer
    void* p = nullptr;
    size_t cap = 0;
};

// ---- RCCL through dlopen (the library loads on machines without RCCL or without a GPU) -----------------------
struct Id128 { char b[128]; };      // ncclUniqueId
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* ncclUniqueId by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load()
{
    std::call_once(g_rccl_once, [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        g_rccl.h = h;
        g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
        g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
        g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
        g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        g_rccl.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
        g_rccl.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
        g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    });
    return g_rccl.h && g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.Send && g_rccl.Recv &&
           g_rccl.GroupStart && g_rccl.GroupEnd;
}

}  // namespace

struct gz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::mutex mu;

    bool have_tables = false;
    GzHostTables host;
    GzDeviceTables dev{};
    DBuf t_pair, t_merges, t_symids, t_bmp, t_astral, t_struct;   // t_struct: device copy of `dev`

    // workspace
    DBuf w_text, w_toff, w_pair, w_poff, w_ids, w_mask, w_tt, w_seq, w_rowoff, w_rowlen, w_pairlen, w_nreal,
        w_status, w_raw, w_arena, w_flags, w_word, w_wordout;
    int32_t* h_flags = nullptr;     // pinned: [1] capacity error, [2] bpe_word count

    // the enqueued call (for gz_sync's arena pass)
    struct Pending {
        bool active = false;
        bool ragged = false;
        bool ragged_direct = false;    // single texts without padding: count pass, scan, rows written once at their places (gz_rowsr_kernel)
        int32_t* emit_block = nullptr; int32_t emit_bits = 0, emit_slot = 0;      // the call also leaves its exchange block ...
        int32_t* emit_rows = nullptr; int32_t* emit_nreal = nullptr; int64_t emit_n = 0; int32_t emit_len = 0;
        int64_t text_bytes = 0;
        GzFinalizeArgs F{};
        bool pair = false;
        GzPairArgs P{};
        bool timing = false;
        std::vector<GzAsmArgs> subs;   // sub-batches of the call (contiguous document ranges)
        int use_words = 0;             // bit 0: whole-word table; bits 8..: timing diagnostics (switch `ablate`, diagnostic build)
        // small batches: ONE fused launch (gz_small_kernel) instead of the pipeline
        bool small = false, s_dense = true;
        bool small_placed = false;     // ... and ragged rows of single texts without padding, ONE workgroup: places and rows made by that launch
        int small_G = 0;
        const uint8_t* s_text0 = nullptr; const int64_t* s_off = nullptr; int64_t s_base = 0, s_docs = 0;
        const uint8_t* s_pair0 = nullptr; const int64_t* s_poff = nullptr; int64_t s_pbase = 0;
        int32_t s_max_len = 0; int32_t* s_ids = nullptr; int32_t* s_mask = nullptr; int32_t* s_nreal = nullptr; int32_t* s_arena = nullptr;
        bool keep_words = false;
        int t_slot = 0;                // timing: the call's pair of events in the ring
        bool chained = false;          // enqueued behind a call that has not been synchronised (its scan flag is kept)
        bool inputs_resident = false;  // the caller's device buffers are readable now (no copy of them is queued on the stream)
    } pend;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // timed calls that were chained without a host sync in between: start / end of the main kernels of the last RING
    // (events are created on first use: an untimed context never pays for them)
    static constexpr int RING = 1024;
    hipEvent_t ring[RING][2] = {};
    uint64_t ring_n = 0;
    double timing[4] = {0, 0, 0, 0};

    void* comm = nullptr;
    int rank = 0, world = 1;

    DBuf t_words2;                                           // long-key whole-word table (17..32 bytes)
    DBuf t_pair8, t_pair_disp, t_words0p, t_word0_disp;      // the perfectly hashed tables (gz_common.h)
    DBuf t_pair_hot, t_word_hot;                             // hot sets the merge / word kernels stage in LDS
    struct TextWs { DBuf brk, st, en, blkcnt, docw0, wtok, waux, mtok, mlist, blkmiss, tilecnt, wlist, grpblk, lookback, mq; } tw[2][2];
    uint32_t lb_epoch = 0;               // call number of the chained scans (gz_scan32m_kernel)   // [slot][text]
    hipStream_t stream2 = nullptr;       // sub-batches alternate between the two streams
    hipStream_t side = nullptr;          // the wide-word kernels of a text run here, beside the merge kernel
    bool flags_lazy = false;       // the device flags of the pending chain have not been copied to h_flags yet (sync_locked does it)
    bool caller_buffers = false;   // set by the device entry points around encode_device_locked: text / offsets are the caller's
                                   // own device buffers (readable now), not staging copies queued on the stream
    hipEvent_t ev_sf0[2][2] = {}, ev_sf[2][2] = {}, ev_sj[2][2] = {}, ev_sb[2][2] = {};    // [slot][text]: forks / joins of the side stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // exchange step (compact / gather / expand) on its own stream, so that it overlaps the next call's kernels
    hipStream_t xstream = nullptr;
    hipEvent_t ev_tok[4] = {nullptr, nullptr, nullptr, nullptr};   // end of the last four encode calls
    // exchange block made by the encode call itself (gz_encode_emit_block arms the next call): the block of the last four calls --
    // its number of entries goes through w_flags[8 + slot] into h_pick's tail (uint32 56 + slot) before ev_blk[slot]
    int32_t* emit_block = nullptr; int32_t emit_bits = 0;
    hipEvent_t ev_blk[4] = {nullptr, nullptr, nullptr, nullptr};
    bool blk_valid[4] = {false, false, false, false};
    hipEvent_t ev_x = nullptr;                                       // last exchange operation issued
    // gz_gather_rows, timed: a pair of events on the exchange stream around every gather (gz_exchange_timing_history)
    static constexpr int XRING = 64;
    hipEvent_t xring[XRING][2] = {};
    uint64_t xring_n = 0;
    uint64_t enc_seq = 0;
    int x_back = 0;                                                  // exchange ops depend on encode call (last - x_back)
    bool x_used = false;
    DBuf w_pick, w_rowoff32;
    int64_t* h_pick = nullptr;           // pinned
    int64_t n_words = 0;
    // decoder snapshot + decode workspace
    bool have_dec = false;
    int32_t dec_n_ids = 0;
    size_t dec_bytes_len = 0;
    std::string dec_unk;
    bool dec_unk_set = false;
    DBuf t_dec_entries, t_dec_bytes, w_dec_ids, w_dec_roff, w_dec_rb, w_dec_ooff, w_dec_out;
    DBuf w_pp[2], w_ppoff[2], w_pplen, w_ppaux, w_pp_in, w_pp_inoff, w_ppctl, w_pplen32, w_pplb;      // text pre-pass
    DBuf w_tiny[8][2];                                                     // texts of fewer than 16 bytes, see encode_device_locked
    // host path with copies overlapped (gz_encode_batch_csr): copy-in / copy-out streams, per-sub-batch events and buffers
    hipStream_t s_in = nullptr, s_out = nullptr;
    static constexpr int CSR_SUBS = 16;
    hipEvent_t ev_in[CSR_SUBS] = {}, ev_done[CSR_SUBS] = {};
    DBuf w_csr_ids[2], w_csr_mask[2], w_csr_comp, w_csr_nreal, w_csr_off32;
    uint32_t* h_tot = nullptr;           // pinned: compact size of every sub-batch
    uint8_t* h_stage = nullptr; size_t h_stage_cap = 0;      // pinned staging of small host calls (one copy in, one copy out)
    // pinned buffers of the host paths (gz_hostpath.h): the caller's memory is never handed to a HIP copy
    uint8_t* h_xfer[2] = {nullptr, nullptr}; size_t h_xfer_cap[2] = {0, 0};          // copy_in / copy_out: two pieces in flight
    hipEvent_t ev_xfer[2] = {nullptr, nullptr}; bool xfer_busy[2] = {false, false};
    uint8_t* h_tin[2] = {nullptr, nullptr}; size_t h_tin_cap[2] = {0, 0};            // text of the sub-batches of a large host call, on its way in
    uint8_t* h_cout[3] = {nullptr, nullptr, nullptr}; size_t h_cout_cap[3] = {0, 0, 0};    // ... their compact rows, on their way out
    DBuf w_stage;
    int cache_status = 0;                // of the last gz_load_tables: 0 no cache, 1 hit, 2 miss (written), 3 a file was refused (rebuilt, rewritten), 4 rebuilt but not written
    bool building_words = false;         // the whole-word table is being built: ignore diagnostics
    GzOptions opt;                       // test / experiment switches (gz_debug_set): a copy of the process-wide defaults at creation
    int n_fresh = 0;                     // diagnostic build: running number of this context's device allocations (switch diag_fresh_only)
};

namespace {

int fail(gz_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try { if (c) c->err = buf; else g_create_err = buf; } catch (...) { /* no memory even for the message: the code still says what happened */ }
    return code;
}

// No C++ exception crosses the C ABI: every extern "C" function of this file is a function-try-block that ends in one of
// these handlers (locks and buffers owned by the frame are released by the unwinding before the handler runs).
#define GZ_CATCH(c)                                                                                               \
    catch (const std::bad_alloc&) { return fail((c), GZ_E_NOMEM, "%s: out of host memory", __func__); }            \
    catch (const std::exception& e_) { return fail((c), GZ_E_HIP, "%s: unexpected C++ exception: %s", __func__, e_.what()); } \
    catch (...) { return fail((c), GZ_E_HIP, "%s: unexpected C++ exception", __func__); }
#define GZ_CATCH_VOID catch (...) { }
#define GZ_CATCH_NULL catch (...) { return nullptr; }

// Test hook (switch `inject_bad_alloc` = k > 0): the k-th allocation site of the entry points reached from now on throws
// std::bad_alloc, as the allocation behind it would on a host that is out of memory.  Sites: wherever an extern "C" body grows
// a std::vector / std::string.  0 (the default) costs one load and a branch per site.
void alloc_site(gz_ctx* c)
{
    if (c->opt.inject_bad_alloc > 0 && --c->opt.inject_bad_alloc == 0) throw std::bad_alloc();
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail((c), GZ_E_HIP, "%s: %s", #call, hipGetErrorString(e_));    \
    } while (0)

#ifdef GZ_DIAG
// ---- guard-granule allocator (diagnostic build, switch diag_guard) ---------------------------------------------------------
// The index assertions of this build cover indices that COME OUT OF MEMORY.  An index a kernel COMPUTES -- a tile's look-ahead
// load near the end of its buffer, the 16 bytes at a word's start -- overruns silently as long as the bytes behind the buffer are
// mapped, which with hipMalloc they almost always are (it carves buffers out of large blocks): such an overrun faults only on the
// day the allocator's layout puts an unmapped page there.  Here every workspace / table buffer is its OWN virtual-memory mapping
// with an unmapped granule on both sides and NO slack: diag_guard = 1 puts the buffer's END on the last byte of its mapping
// (16-byte aligned start), = 2 its START on the first byte (the granule is 4 KiB on this runtime).  One byte too far faults on the spot, every time, and the runtime's
// fault message (AMD_LOG_LEVEL=1 names the kernel) replaces a coin toss.
struct GuardMap { void* base; size_t reserved; void* map_at; size_t mapped; hipMemGenericAllocationHandle_t h; };
std::mutex g_guard_mu;
std::unordered_map<void*, GuardMap> g_guard;                 // user pointer -> its mapping

int guard_alloc(gz_ctx* c, void** out, size_t bytes, int mode)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || gran == 0) return fail(c, GZ_E_HIP, "diag_guard: hipMemGetAllocationGranularity: %s", hipGetErrorString(e));
    const size_t want = (bytes + 15) & ~(size_t)15;
    GuardMap g{};
    g.mapped = (want + gran - 1) / gran * gran;
    g.reserved = g.mapped + 2 * gran;
    if ((e = hipMemAddressReserve(&g.base, g.reserved, gran, nullptr, 0)) != hipSuccess) return fail(c, GZ_E_NOMEM, "diag_guard: hipMemAddressReserve(%zu): %s", g.reserved, hipGetErrorString(e));
    g.map_at = (uint8_t*)g.base + gran;
    if ((e = hipMemCreate(&g.h, g.mapped, &prop, 0)) != hipSuccess) { hipMemAddressFree(g.base, g.reserved); return fail(c, GZ_E_NOMEM, "diag_guard: hipMemCreate(%zu): %s", g.mapped, hipGetErrorString(e)); }
    if ((e = hipMemMap(g.map_at, g.mapped, 0, g.h, 0)) != hipSuccess) { hipMemRelease(g.h); hipMemAddressFree(g.base, g.reserved); return fail(c, GZ_E_HIP, "diag_guard: hipMemMap: %s", hipGetErrorString(e)); }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(g.map_at, g.mapped, &acc, 1)) != hipSuccess) {
        hipMemUnmap(g.map_at, g.mapped); hipMemRelease(g.h); hipMemAddressFree(g.base, g.reserved);
        return fail(c, GZ_E_HIP, "diag_guard: hipMemSetAccess: %s", hipGetErrorString(e));
    }
    void* user = mode == 1 ? (void*)((uint8_t*)g.map_at + g.mapped - want) : g.map_at;
    { std::lock_guard<std::mutex> lk(g_guard_mu); g_guard[user] = g; }
    *out = user;
    return GZ_OK;
}
bool guard_free(void* user)
{
    GuardMap g;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard.find(user);
        if (it == g_guard.end()) return false;
        g = it->second;
        g_guard.erase(it);
    }
    hipDeviceSynchronize();
    hipMemUnmap(g.map_at, g.mapped);
    hipMemRelease(g.h);
    // The address range is NOT given back (hipMemAddressFree): a freed buffer's addresses are never handed out again.  Measured on
    // this runtime (ROCm 7.2, gfx950; profiles/r06_ab_variants.txt series 0): a range that is reserved again right after it was
    // freed gets the old addresses, and the first copy / kernel into the new mapping can still go through the old translation --
    // rows came out wrong (no fault) in exactly the calls that re-allocated a buffer, and only with this allocator.  Address space
    // is free; a stale pointer into a freed buffer now also faults for good.
    return true;
}
void dev_free(void* p) { if (p && !guard_free(p)) hipFree(p); }
#else
void dev_free(void* p) { if (p) hipFree(p); }
#endif

// zero_new: a fresh allocation is cleared before anything uses it (the chained-scan words are validated by a per-context
// call number only: memory another context freed could otherwise carry words that look current)
#ifdef GZ_DIAG
// diagnostic build, switch diag_fresh: a fresh allocation is filled with a byte of the tester's choice (fresh memory is usually
// zero -- with hipMalloc; not with the guard allocator's mappings -- and zero hides a read of a word nobody wrote)
int fresh_fill(gz_ctx* c, void* p, size_t bytes, const char* what)
{
    const int k = c->n_fresh++;
    if (c->opt.diag_fresh <= 0 || (c->opt.diag_fresh_only >= 0 && c->opt.diag_fresh_only != k)) return GZ_OK;
    if (c->opt.diag_fresh_only >= 0) fprintf(stderr, "[diag_fresh] allocation %d: %s, %zu bytes, filled with 0x%02X\n", k, what, bytes, c->opt.diag_fresh - 1);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemset(p, c->opt.diag_fresh - 1, bytes));
    HIPCHK(c, hipDeviceSynchronize());
    return GZ_OK;
}
#define ensure(c, b, ...) ensure_named((c), (b), #b, __VA_ARGS__)
#else
#define ensure(c, b, ...) ensure_named((c), (b), nullptr, __VA_ARGS__)
#endif
int ensure_named(gz_ctx* c, DBuf& b, const char* what, size_t bytes, bool zero_new = false)
{
    (void)what;
    if (bytes <= b.cap && b.p) return GZ_OK;
    if (b.p) { dev_free(b.p); b.p = nullptr; b.cap = 0; }
#ifdef GZ_DIAG
    if (c->opt.diag_guard) {
        // exactly what was asked for (rounded up to 16 bytes): no slack, an unmapped granule on both sides
        const size_t want = ((bytes ? bytes : 16) + 15) & ~(size_t)15;
        int rc = guard_alloc(c, &b.p, want, c->opt.diag_guard);
        if (rc) { b.p = nullptr; return rc; }
        b.cap = want;
        if (zero_new) { if (hipMemset(b.p, 0, want) != hipSuccess) return fail(c, GZ_E_HIP, "hipMemset of a new workspace buffer failed"); }
        else if ((rc = fresh_fill(c, b.p, want, what))) return rc;
        return GZ_OK;
    }
#endif
    size_t want = bytes + 256;                      // slack: tile loads may touch up to 15 bytes past the text
    want = (want + 4095) & ~(size_t)4095;
#ifdef GZ_DIAG
    if (c->opt.diag_exact) want = ((bytes ? bytes : 16) + 15) & ~(size_t)15;
#endif
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) { b.p = nullptr; return fail(c, GZ_E_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
    b.cap = want;
    if (zero_new) { if (hipMemset(b.p, 0, want) != hipSuccess) return fail(c, GZ_E_HIP, "hipMemset of a new workspace buffer failed"); }
#ifdef GZ_DIAG
    else { int rc = fresh_fill(c, b.p, want, what); if (rc) return rc; }
#endif
    return GZ_OK;
}

void release(DBuf& b) { if (b.p) dev_free(b.p); b.p = nullptr; b.cap = 0; }

}  // namespace
#include "gz_hostpath.h"
namespace {

template <class V>
int upload(gz_ctx* c, DBuf& b, const V& v)
{
    size_t bytes = v.size() * sizeof(v[0]);
    int rc = ensure(c, b, bytes ? bytes : 16);
    if (rc) return rc;
    return copy_in(c, b.p, v.data(), bytes, c->stream);
}

bool is_dense(const GzShape& S) { return S.pad_mode && S.truncation && S.max_len >= 1; }

GzShape make_shape(int32_t max_len, uint32_t flags)
{
    GzShape S;
    S.max_len = max_len;
    S.pad_mode = (!(flags & GZ_MAX_LEN_NONE) && (flags & GZ_PADDING)) ? 1 : 0;
    S.truncation = (flags & GZ_TRUNCATION) ? 1 : 0;
    return S;
}

// Enqueue every kernel of one call.  All pointers are device pointers.  Sub-batches alternate between two streams so
// that the bandwidth-bound assemble kernel of one overlaps the issue-bound word / merge kernels of the next.
// The second text stream and the side stream of the batch pipeline are made when a call first needs them (a stream costs
// ~ 2.5 ms to create: a tokenizer that only ever sees small calls never pays for them).
static int need_side_streams(gz_ctx* c, bool second)
{
    if (!c->side) HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    if (second && !c->stream2) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    return GZ_OK;
}

int enqueue(gz_ctx* c)
{
    gz_ctx::Pending& p = c->pend;
    hipStream_t s = c->stream;
    const GzDeviceTables* T = (const GzDeviceTables*)c->t_struct.p;
    if (c->x_used) HIPCHK(c, hipStreamWaitEvent(s, c->ev_x, 0));    // output buffers may still be read by an exchange step
    const bool no_flags = p.small && !p.ragged;                 // (a dense one-launch call raises no flag)
    // [0] scan time-out, [1] capacity error, [3] a word needs the wide / long kernels.  Calls chained without a host
    // synchronisation keep [0]: it is only cleared when a chain starts, so a time-out in ANY call of the chain is still
    // there when the chain is closed (sync_locked)
    // A dense call uses [0] and [3] only, and both may stay as they are inside a chain: [0] is meant to, and a stale [3] only
    // keeps gz_long_kernel from leaving at once (its list is empty).  Such a call neither clears the flags inside a chain nor
    // copies them back: the copy is made once, when the chain is closed (sync_locked) -- between two launches that follow each
    // other the stream then has nothing to do but the next launch (each clear / copy is a kernel of its own: ~ 10 us of idle chip).
    const bool lazy_flags = !no_flags && !p.ragged;
    if (!no_flags) {
        if (!p.chained) HIPCHK(c, hipMemsetAsync(c->w_flags.p, 0, 16, s));
        else if (!lazy_flags) HIPCHK(c, hipMemsetAsync((int32_t*)c->w_flags.p + 1, 0, 12, s));
    }
    if (p.timing) {
        hipEvent_t* slot = c->ring[c->ring_n % gz_ctx::RING];
        if (!slot[0]) { HIPCHK(c, hipEventCreate(&slot[0])); HIPCHK(c, hipEventCreate(&slot[1])); }
        HIPCHK(c, hipEventRecord(slot[0], s));                  // (one timed event at each end of the kernels: every one costs the
                                                                //  stream ~ 5 us between two launches)
    }
    const bool two = p.subs.size() > 1;
    if (!p.subs.empty()) { int rs = need_side_streams(c, two); if (rs) return rs; }
    if (two) { HIPCHK(c, hipEventRecord(c->ev_fork, s)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0)); }
    if (p.small)
        gz_launch_small(T, p.s_text0, p.s_off, p.s_base, p.s_pair0, p.s_poff, p.s_pbase, p.s_docs, p.small_G, p.s_dense ? 1 : p.small_placed ? 2 : 0, p.s_max_len,
                        p.use_words, p.s_ids, p.s_mask, p.s_nreal, p.s_arena, p.F.row_off, p.F.capacity, (int32_t*)c->w_flags.p + 1, s);
    for (size_t k = 0; k < p.subs.size(); ++k) {
        hipStream_t sk = (k & 1) ? c->stream2 : s;
        const GzAsmArgs& S = p.subs[k];
        for (int tx = 0; tx < S.n_texts; ++tx)
            gz_launch_pipeline_text(c->opt, T, c->dev, S.X[tx], S.n_docs, p.use_words, (int32_t*)c->w_flags.p + 3, sk, c->side, c->ev_sf0[k & 1][tx], c->ev_sf[k & 1][tx], c->ev_sj[k & 1][tx],
                                    p.inputs_resident ? c->ev_sb[k & 1][tx] : nullptr);
        if (p.ragged_direct) gz_launch_rows_ragged(T, S, 0, p.text_bytes, sk);      // the row lengths
        else gz_launch_assemble(c->opt, T, S, sk);
    }
    if (two) { HIPCHK(c, hipEventRecord(c->ev_join, c->stream2)); HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0)); }
    if (p.timing) { p.t_slot = (int)(c->ring_n % gz_ctx::RING); HIPCHK(c, hipEventRecord(c->ring[p.t_slot][1], s)); c->ring_n++; }
    if (p.emit_block) {
        uint32_t* cur = (uint32_t*)c->w_flags.p + 8 + p.emit_slot;
        {
            // the block from the call's dense rows, right behind its kernels on the SAME stream: the row lengths, their scan, the compact
            // kernel.  (On the exchange stream -- beside the next call's kernels -- the same work stretched those kernels by more than it
            // takes here; a row kernel that writes the block itself was tried twice, r05_ab_variants.txt series 13.)
            uint32_t* off = (uint32_t*)c->w_rowoff32.p;
            if (p.emit_n) HIPCHK(c, hipMemcpyAsync(p.emit_block, p.emit_nreal, (size_t)p.emit_n * 4, hipMemcpyDeviceToDevice, s));
            gz_launch_row_offsets(p.emit_nreal, p.emit_n, off, s);
            gz_launch_compact(p.emit_rows, off, p.emit_n, p.emit_len, p.emit_block + 2 * p.emit_n, p.emit_bits, (uint32_t*)(p.emit_block + p.emit_n), s);
            HIPCHK(c, hipMemcpyAsync(cur, off + p.emit_n, 4, hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(c, hipMemcpyAsync(reinterpret_cast<uint32_t*>(c->h_pick) + 56 + p.emit_slot, cur, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_blk[p.emit_slot], s));
        c->blk_valid[p.emit_slot] = true;
    } else c->blk_valid[c->enc_seq & 3] = false;
    if (p.ragged && !p.small_placed) {
        gz_launch_rowscan(p.F, (int64_t*)c->w_rowlen.p, s);
        if (p.ragged_direct) gz_launch_rows_ragged(T, p.subs[0], 1, p.text_bytes, s);     // the rows, at their places
        else gz_launch_finalize(c->dev, p.F, s);
    }
    if (p.timing && (p.ragged || p.pair)) HIPCHK(c, hipEventRecord(c->ev[2], s));      // (dense single texts: nothing follows the row kernel)
    if (p.pair) gz_launch_pair(c->dev, p.P, s);
    if (p.timing && (p.ragged || p.pair)) HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (lazy_flags) c->flags_lazy = true;
    else if (!no_flags) { HIPCHK(c, hipMemcpyAsync(c->h_flags, c->w_flags.p, 8, hipMemcpyDeviceToHost, s)); c->flags_lazy = false; }
    else if (!p.chained) c->h_flags[0] = c->h_flags[1] = 0;
    HIPCHK(c, hipEventRecord(c->ev_tok[c->enc_seq & 3], s));
    c->enc_seq++;
    HIPCHK(c, hipGetLastError());
    return GZ_OK;
}

// exchange operations run on c->xstream, after the encode call they belong to (gz_exchange_select) and before any later
// encode call touches the buffers again (enqueue waits for ev_x)
int x_begin(gz_ctx* c)
{
    if (!c->xstream) HIPCHK(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));      // (a stream costs ~ 2.5 ms to create: made when first needed)
    if (c->enc_seq > (uint64_t)c->x_back) HIPCHK(c, hipStreamWaitEvent(c->xstream, c->ev_tok[(c->enc_seq - 1 - (uint64_t)c->x_back) & 3], 0));
    return GZ_OK;
}
int x_end(gz_ctx* c)
{
    HIPCHK(c, hipEventRecord(c->ev_x, c->xstream));
    c->x_used = true;
    return GZ_OK;
}

#ifdef GZ_DIAG
extern "C" int gz_diag_check(unsigned int* out8, int clear);      // gz_hot.inc: the record of the kernels' index assertions
#endif

int sync_locked(gz_ctx* c)
{
    if (c->flags_lazy) {                                         // the flags of a chain of dense calls: copied once, now
        c->flags_lazy = false;
        HIPCHK(c, hipMemcpyAsync(c->h_flags, c->w_flags.p, 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->x_used) {
        HIPCHK(c, hipStreamSynchronize(c->xstream));
        int32_t* const bad = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(c->h_pick) + 384);
        if (*bad) {
            *bad = 0;
            HIPCHK(c, hipMemset((int32_t*)c->w_flags.p + 12, 0, 4));
            c->pend.active = false;
            return fail(c, GZ_E_INVALID, "an exchange block did not fit what was announced for it (a row longer than a row, or entries beyond the block's total): truncated, or written in another layout");
        }
    }
#ifdef GZ_DIAG
    {
        // diagnostic build: an index the kernels took out of memory did not fit the array it was meant for (the access was skipped)
        unsigned int e[8] = {0};
        if (gz_diag_check(e, 1) == 0 && e[0] != 0u) {
            c->pend.active = false;
            return fail(c, GZ_E_HIP, "diagnostic build: index check %u failed: index %llu, bound %u, workgroup %u (%u failures in all)", e[0],
                        (unsigned long long)e[1] | ((unsigned long long)e[2] << 32), e[3], e[5], e[4]);
        }
    }
#endif
    gz_ctx::Pending& p = c->pend;
    if (!p.active) return GZ_OK;
    if (p.timing) {
        float ms = 0;
        hipEvent_t t0 = c->ring[p.t_slot][0], t1 = c->ring[p.t_slot][1];
        hipEventElapsedTime(&ms, t0, t1); c->timing[0] = ms;
        c->timing[1] = c->timing[2] = 0; c->timing[3] = ms;
        if (p.ragged || p.pair) {
            hipEventElapsedTime(&ms, t1, c->ev[2]); c->timing[1] = ms;
            hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); c->timing[2] = ms;
            hipEventElapsedTime(&ms, t0, c->ev[3]); c->timing[3] = ms;
        }
    }
    p.active = false;
    if (c->h_flags[0]) return fail(c, GZ_E_HIP, "internal: a chained scan (gz_scan32m_kernel look-back) timed out");
    if (c->h_flags[1]) return fail(c, GZ_E_CAPACITY, "ragged output larger than capacity");
    return GZ_OK;
}

// One text (A or B) of a (sub-)batch: its per-call workspace in slot W, sized from the byte and document counts.
// tb = first byte of the batch on the device, off = its [n_docs + 1] absolute offsets (device), Bt = its bytes.
int setup_text(gz_ctx* c, gz_ctx::TextWs& W, DBuf& tiny, const uint8_t* tb, const int64_t* off, int64_t Bt, int64_t n_docs,
               hipStream_t s, GzTextBufs& X)
{
    if (Bt < 0) return fail(c, GZ_E_INVALID, "offsets are not non-decreasing");
    if (Bt >= GZ_TEXT_BYTES_LIMIT) return fail(c, GZ_E_LIMIT, "a batch of 4 GiB or more of text must be split");
    X.tb = tb;
    if (Bt > 0 && Bt < 16) {
        // the kernels read the 16 bytes that END at the last byte of the text (load16 / load4_tail move a load
        // back instead of running past B): give a tiny text 16 bytes of lead-in and zeroed slack behind it
        // (one buffer per sub-batch: they are filled before any kernel runs)
        int rc3;
        if ((rc3 = ensure(c, tiny, 64))) return rc3;
        HIPCHK(c, hipMemsetAsync(tiny.p, 0, 64, s));
        HIPCHK(c, hipMemcpyAsync((uint8_t*)tiny.p + 16, X.tb, (size_t)Bt, hipMemcpyDeviceToDevice, s));
        X.tb = (const uint8_t*)tiny.p + 16;
    }
    X.off = off;
    X.B = Bt;
    {
        // near records (gz_pipeline.inc, W_NEAR): the option near_limit shrinks the range so that small test batches reach the far form
        X.near_lim = (uint32_t)c->opt.near_limit;
    }
    X.nblk = Bt / 4096 + 1;
    const size_t bm = (size_t)((Bt + 1024) / 1024) * 128 + 64 + 4096;
    int64_t wmax = Bt / 2 + n_docs + 2; if (wmax > Bt + 2) wmax = Bt + 2;
    const int64_t nbr = (X.nblk + 2 + 3) & ~(int64_t)3;                       // block flags, then the control words, then the wide-word list: one buffer
    int rc2;
    if ((rc2 = ensure(c, W.brk, bm)) || (rc2 = ensure(c, W.st, bm)) || (rc2 = ensure(c, W.en, bm)) ||
        (rc2 = ensure(c, W.blkcnt, (size_t)(X.nblk + 2) * 4)) || (rc2 = ensure(c, W.docw0, (size_t)(n_docs + 2) * 4)) ||
        (rc2 = ensure(c, W.wtok, (size_t)wmax * 4)) || (rc2 = ensure(c, W.waux, (size_t)wmax * 4)) ||
        (rc2 = ensure(c, W.mtok, (size_t)(Bt + 32) * 8 + 64)) /* [0, B + 32): wide words, by byte offset; behind it: the compact token area */ ||
        (rc2 = ensure(c, W.mlist, (size_t)wmax * 16)) ||
        (rc2 = ensure(c, W.grpblk, (size_t)(wmax / 64 + 4 + wmax / 1024 + 8) * 4)) ||
        (rc2 = ensure(c, W.blkmiss, (size_t)(X.nblk + 2) * 4)) ||
        (rc2 = ensure(c, W.tilecnt, (size_t)(X.nblk + 2) * 8)) || (rc2 = ensure(c, W.wlist, (size_t)(nbr + 64 + wmax + 8) * 4)) ||
        (rc2 = ensure(c, W.mq, (size_t)wmax * 16)) ||
        (rc2 = ensure(c, W.lookback, (size_t)(X.nblk / 4 + 4) * 8, /* zero a new allocation */ true)))
        return rc2;
#ifdef GZ_DIAG
    {
        // option diag_poison (diagnostic build): the whole per-call workspace is filled with 0xFF before every call, so that a kernel
        // that consumes a word nobody wrote in THIS call reads an impossible index or a pending record every time, not just when the
        // allocation happens to hold one (run with brk_side = 0: the start bits are then cleared on this stream, behind the fill).
        // The chained scans' words are left alone: they are validated by their call number.
        if (c->opt.diag_poison)
            for (DBuf* b : {&W.brk, &W.st, &W.en, &W.blkcnt, &W.docw0, &W.wtok, &W.waux, &W.mtok, &W.mlist, &W.grpblk, &W.blkmiss, &W.tilecnt, &W.wlist, &W.mq})
                HIPCHK(c, hipMemsetAsync(b->p, 0xFF, b->cap, s));
    }
#endif
    X.brk = (uint16_t*)W.brk.p; X.st = (uint16_t*)W.st.p; X.en = (uint16_t*)W.en.p;
    X.blkcnt = (uint32_t*)W.blkcnt.p; X.docw0 = (uint32_t*)W.docw0.p;
    X.wtok = (uint32_t*)W.wtok.p; X.waux = (uint32_t*)W.waux.p; X.mtok = (int32_t*)W.mtok.p;
    X.mlist = (uint4*)W.mlist.p; X.grpblk = (uint32_t*)W.grpblk.p; X.tcnt = X.grpblk + wmax / 64 + 4; X.wmax = wmax; X.blkmiss = (uint32_t*)W.blkmiss.p;
    X.tilecnt = (uint16_t*)W.tilecnt.p;
    X.blklong = (uint32_t*)W.wlist.p; X.ctl = X.blklong + nbr; X.wlist = X.ctl + 64;     // (ONE memset clears the flags, the control words and the list's count)
    X.mq = (uint4*)W.mq.p;
    X.lookback = (uint64_t*)W.lookback.p;
    // call numbers of the chained scans of this text: X.epoch + 1 and + 2 (gz_scan32m_kernel; X.epoch itself is unused); they are
    // 1, 2, 3 mod 4, so never 0 in their low 30 bits (0 is what a fresh allocation may hold)
    c->lb_epoch += 4;
    if ((c->lb_epoch & 0x3FFFFFFFu) < 4u) {                      // the 30-bit call number wrapped: words of 2^28 calls ago would look current
        HIPCHK(c, hipDeviceSynchronize());
        for (auto& slot : c->tw) for (auto& t : slot) if (t.lookback.p) HIPCHK(c, hipMemset(t.lookback.p, 0, t.lookback.cap));
        if (c->w_pplb.p) HIPCHK(c, hipMemset(c->w_pplb.p, 0, c->w_pplb.cap));
    }
    X.epoch = c->lb_epoch + 1;
    return GZ_OK;
}

bool ids_fit_16(gz_ctx* c)
{
    for (int32_t id : c->host.enc_ids) if (id < 0 || id > 0xFFFF) return false;
    return true;
}

int use_words_flags(gz_ctx* c, uint32_t flags)
{
#if defined(GZ_DIAG) || defined(GZ_ABLATE)
    const int ablate = c->building_words ? 0 : c->opt.ablate;       // timing diagnostics only (diagnostic builds): results are wrong when set
#else
    const int ablate = 0;
#endif
    const int use_words = (c->dev.words0p != nullptr && !(flags & GZ_NO_WORD_TABLE) && c->opt.word_table) ? 1 : 0;
    return use_words | (ablate << 8);
}

int encode_device_locked(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                         const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                         int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                         int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status,
                         const int64_t* h_text_off /* host copy of the offsets, or nullptr */, const int64_t* h_pair_off,
                         int32_t* emit_block = nullptr /* gz_encode_emit_block's arming, taken by the device entry points only */, int32_t emit_bits = 0)
{
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    if (n_docs < 0 || !text_off || (n_docs > 0 && !text)) return fail(c, GZ_E_INVALID, "bad text arguments");
    if ((pair == nullptr) != (pair_off == nullptr)) return fail(c, GZ_E_INVALID, "pair and pair_off must both be given or both be NULL");
    if (!input_ids || !attention_mask) return fail(c, GZ_E_INVALID, "input_ids / attention_mask are required");
    const bool is_pair = pair_off != nullptr;
    if (is_pair && (!token_type_ids || !sequence_id || !pair_len || !status))
        return fail(c, GZ_E_INVALID, "pair mode needs token_type_ids, sequence_id, pair_len and status");
    // A pending call is normally synchronised first.  A DENSE pending call has nothing the host must look at (no
    // capacity flag), so when this call also comes with host offsets (no size read-back) its kernels are simply
    // enqueued behind it: the GPU never waits for the host between steps.  gz_sync / gz_timing_history close the chain.
    if (c->pend.active && (c->pend.ragged || !h_text_off)) { int rc = sync_locked(c); if (rc) return rc; }

    const GzShape S = make_shape(max_len, flags);
    const bool dense = is_dense(S);
    if (dense && capacity < n_docs * (int64_t)max_len) return fail(c, GZ_E_CAPACITY, "capacity < n_docs*max_len");
    if (!dense && !row_off) return fail(c, GZ_E_INVALID, "row_off is required for ragged layouts");

    int rc;
    if (!n_real) { rc = ensure(c, c->w_nreal, (size_t)(n_docs + 1) * 4); if (rc) return rc; n_real = (int32_t*)c->w_nreal.p; }
    gz_ctx::Pending& p = c->pend;
    const bool chained = p.active;                             // (still pending: this call goes behind it on the stream)
    p = gz_ctx::Pending();
    p.chained = chained;
    p.inputs_resident = c->caller_buffers;
    p.timing = (flags & GZ_TIMING) != 0;

    // Sub-batches: contiguous document ranges (dense layouts of large batches only).  Their byte positions are the
    // only thing the host needs to know about the offsets: one tiny kernel + one 8*(2*nsub+2)-byte copy.
    // (measured on cfg 3: 2 sub-batches on two streams gain 1.5 %, 4 gain nothing, 8 lose 10 % -- the kernels of one
    // sub-batch already fill the chip -- so the default is one batch; the option sub_batches is kept for experiments)
    int nsub = (dense && n_docs >= 8 * (int64_t)c->opt.sub_batches) ? c->opt.sub_batches : 1;
    int64_t cutA[9], cutB[9];
    if (h_text_off) {
        for (int k = 0; k <= nsub; ++k) {
            const int64_t d = (int64_t)k * n_docs / nsub;
            cutA[k] = h_text_off[d];
            cutB[k] = is_pair ? h_pair_off[d] : 0;
        }
    } else {
        if ((rc = ensure(c, c->w_pick, 256))) return rc;
        gz_launch_pick(text_off, pair_off, n_docs, nsub, (int64_t*)c->w_pick.p, c->stream);
        HIPCHK(c, hipMemcpyAsync(c->h_pick, c->w_pick.p, (size_t)(2 * nsub + 2) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k <= nsub; ++k) { cutA[k] = c->h_pick[k]; cutB[k] = c->h_pick[nsub + 1 + k]; }
    }
    const int64_t text_bytes = cutA[nsub] - cutA[0], pair_bytes = is_pair ? cutB[nsub] - cutB[0] : 0;
    if (text_bytes < 0 || pair_bytes < 0) return fail(c, GZ_E_INVALID, "offsets are not non-decreasing");

    int docs_per_wave;
    {
        // documents per wave of the assemble kernel: ~4 KiB of text per wave, but keep >= 16 K waves when possible
        const int64_t avg = n_docs > 0 ? text_bytes / n_docs : 0;
        int64_t dpw = avg > 0 ? 4096 / avg : GZ_MAX_DOCS_PER_WAVE;
        const int64_t by_waves = n_docs / 16384;
        if (dpw > by_waves) dpw = by_waves;
        if (dpw < 1) dpw = 1;
        if (dpw > GZ_MAX_DOCS_PER_WAVE) dpw = GZ_MAX_DOCS_PER_WAVE;
        if (c->opt.docs_per_wave >= 1 && c->opt.docs_per_wave <= GZ_MAX_DOCS_PER_WAVE) dpw = c->opt.docs_per_wave;
        docs_per_wave = (int)dpw;
        p.use_words = use_words_flags(c, flags);
    }
    int32_t* raw = nullptr;
    int32_t* n_raw = n_real;
    // Rows without padding of single texts (max_len None, padding False) are counted first and then written once, at their final
    // places (gz_rowsr_kernel); pairs and the padded ragged shapes go through the raw area and the finalize kernel.  (The option
    // assemble < 3 keeps the raw-area path for these rows too: tests.)
    const bool direct_ok = !dense && !is_pair && !S.pad_mode && c->opt.assemble >= 3;
    const bool maybe_small = c->opt.small && h_text_off && n_docs <= (1 << 20) && text_bytes + pair_bytes <= (2ll << 20);
    if (!dense) {
        rc = ensure(c, c->w_rowlen, (size_t)(n_docs + 1) * 8); if (rc) return rc;
        if (!direct_ok || maybe_small) {
            const int64_t raw_elems = text_bytes + pair_bytes + (is_pair ? 4 : 2) * n_docs;
            rc = ensure(c, c->w_raw, (size_t)raw_elems * 4 + 16); if (rc) return rc;
            raw = (int32_t*)c->w_raw.p;
            // raw token counts live in a private buffer: n_real is rewritten by the finalize kernel
            rc = ensure(c, c->w_status, (size_t)(n_docs + 1) * 4); if (rc) return rc;
            n_raw = (int32_t*)c->w_status.p;
        }
    }
    p.keep_words = (flags & GZ_KEEP_WORDS) != 0;
    {
        // Small batches run in ONE launch that cuts the work by documents (gz_small.inc): single texts or pairs whose
        // longest document fits a workgroup's LDS (ragged layouts: the unpadded rows go to the raw area, finalize follows as in
        // the big path).  The host needs the document sizes for that, so only calls that bring host offsets qualify.  The option small = 0 switches the path off (tests run the golden batches both ways).
        if (c->opt.small && h_text_off && !p.keep_words && n_docs > 0 && n_docs <= (1 << 20) && text_bytes + pair_bytes <= (2ll << 20)) {
            int64_t maxdoc = 0, maxpair = 0;
            for (int64_t d = 0; d < n_docs; ++d) {
                const int64_t b = h_text_off[d + 1] - h_text_off[d]; if (b > maxdoc) maxdoc = b;
                if (is_pair) { const int64_t b2 = h_pair_off[d + 1] - h_pair_off[d]; if (b2 > maxpair) maxpair = b2; }
            }
            maxdoc += maxpair;                                   // a group of G documents holds at most G * (longest A + longest B) bytes
            if (maxdoc <= GZ_SMALL_DOC_BYTES) {
                int64_t G = maxdoc > 0 ? GZ_SMALL_DOC_BYTES / maxdoc : GZ_SMALL_DOCS_PER_WG;
                if (G > GZ_SMALL_DOCS_PER_WG) G = GZ_SMALL_DOCS_PER_WG;
                // a batch this small cannot fill the chip with full groups: more, smaller workgroups (each phase of a workgroup
                // is a dependent chain, so the call's time is the time of ONE workgroup)
                const int64_t wg_target = c->opt.small_wgs;
                const int64_t by_chip = (n_docs + wg_target - 1) / wg_target;
                if (G > by_chip) G = by_chip;
                gz_ctx::TextWs& W = c->tw[0][0];
                if ((rc = ensure(c, W.mtok, (size_t)(text_bytes + pair_bytes + 32) * 4))) return rc;      // arena of very long words
                p.small = true; p.small_G = (int)G;
                p.s_text0 = text + cutA[0]; p.s_off = text_off; p.s_base = cutA[0]; p.s_docs = n_docs;
                if (is_pair) { p.s_pair0 = pair + cutB[0]; p.s_poff = pair_off; p.s_pbase = cutB[0]; }
                p.s_dense = dense; p.s_max_len = max_len; p.s_arena = (int32_t*)W.mtok.p;
                p.s_ids = dense ? input_ids : raw; p.s_mask = dense ? attention_mask : raw; p.s_nreal = dense ? n_real : n_raw;
                // one workgroup holds the whole call (a single encode(), a handful of sentences) and the rows have no padding: it also
                // places and writes them -- three launches less behind it
                if (!dense && !S.pad_mode && c->opt.assemble >= 3 && (n_docs + G - 1) / G == 1) {       // (pairs too: gz_pair_kernel follows either way)
                    p.small_placed = true;
                    p.s_ids = input_ids; p.s_mask = attention_mask; p.s_nreal = n_real;
                }
                nsub = 0;
            }
        }
    }
    p.ragged_direct = direct_ok && !p.small;
    p.text_bytes = text_bytes;
    if (p.ragged_direct) n_raw = n_real;                       // (the count pass writes the row lengths where the caller wants them)
    alloc_site(c);
    p.subs.resize((size_t)nsub);
    for (int k = 0; k < nsub; ++k) {
        const int64_t lo = (int64_t)k * n_docs / nsub, hi = (int64_t)(k + 1) * n_docs / nsub;
        GzAsmArgs& S2 = p.subs[(size_t)k];
        S2.n_texts = is_pair ? 2 : 1;
        S2.n_docs = hi - lo; S2.dense = dense ? 1 : 0; S2.max_len = max_len;
        S2.ids = input_ids + (dense ? lo * (int64_t)max_len : 0);
        S2.mask = attention_mask + (dense ? lo * (int64_t)max_len : 0);
        S2.raw = raw; S2.n_real = n_raw + lo;
        S2.docs_per_wave = docs_per_wave;
        S2.row_off = row_off; S2.capacity = capacity; S2.error_flag = (int32_t*)c->w_flags.p + 1;
        for (int tx = 0; tx < S2.n_texts; ++tx) {
            const int64_t* cut = tx ? cutB : cutA;
            int rc2 = setup_text(c, c->tw[k & 1][tx], c->w_tiny[k & 7][tx], (tx ? pair : text) + cut[k], (tx ? pair_off : text_off) + lo,
                                 cut[k + 1] - cut[k], S2.n_docs, c->stream, S2.X[tx]);
            if (rc2) return rc2;
        }
    }
    if (emit_block) {
        // the call's exchange block: [n_real[n] | first[n] | entries]
        if (!dense) return fail(c, GZ_E_INVALID, "gz_encode_emit_block needs a dense call (padding, truncation, max_len >= 1)");
        p.emit_block = emit_block; p.emit_bits = emit_bits; p.emit_slot = (int32_t)(c->enc_seq & 3);
        p.emit_rows = input_ids; p.emit_nreal = n_real; p.emit_n = n_docs; p.emit_len = max_len;
        if (!c->ev_blk[p.emit_slot]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_blk[p.emit_slot], hipEventDisableTiming));
        if ((rc = ensure(c, c->w_rowoff32, (size_t)(n_docs + 2) * 4))) return rc;       // (the fallback's scan)
    }
    if (!dense) {
        p.ragged = true;
        GzFinalizeArgs& F = p.F;
        F.text_off = text_off; F.pair_off = pair_off; F.n_docs = n_docs; F.S = S;
        F.raw = raw; F.n_raw = n_raw; F.row_off = row_off; F.capacity = capacity;
        F.ids = input_ids; F.mask = attention_mask; F.n_real = n_real;
        F.error_flag = (int32_t*)c->w_flags.p + 1;
    }
    if (is_pair) {
        p.pair = true;
        GzPairArgs& P = p.P;
        P.n_docs = n_docs; P.S = S; P.row_off = dense ? nullptr : row_off; P.capacity = capacity;
        P.ids = input_ids; P.seq = sequence_id; P.tt = token_type_ids; P.pair_len = pair_len; P.status = status;
    }
    p.active = true;
    rc = enqueue(c);
    if (rc) p.active = false;
    return rc;
}

}  // namespace


static int encode_host_locked(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                    const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                    int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                    int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status);

#include "gz_cache.inc"

// Whole-word table.  Candidates are the merge-closure symbols that end a word (".....</w>", <= 16 bytes without the
// marker).  Each candidate is tokenized BY THE GPU MERGE PATH ITSELF (one tiny document per word, table disabled);
// a word whose result is exactly [bos, id, eos] is recorded as word -> id.  Nothing is computed on the CPU.
static int build_word_table(gz_ctx* c, WordImages& W)
{
    static const bool load_timing = getenv("GZ_LOAD_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!load_timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "  word tables: %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    W = WordImages();
    GzHostTables& H = c->host;
    std::vector<uint8_t> text;
    std::vector<int64_t> off(1, 0);
    std::vector<uint32_t> which;
    for (size_t s = 0; s < H.symbols.size(); ++s) {
        const std::string& str = H.symbols[s];
        if (str.size() < 5 || str.size() > 36 || str.compare(str.size() - 4, 4, "</w>") != 0) continue;
        const size_t n = str.size() - 4;
        if (!gz_is_plain_word((const uint8_t*)str.data(), n)) continue;     // contains whitespace: never one word
        text.insert(text.end(), str.begin(), str.begin() + n);
        off.push_back((int64_t)text.size());
        which.push_back((uint32_t)s);
    }
    const int64_t n = (int64_t)which.size();
    if (n == 0) return GZ_OK;
    const int64_t cap = (int64_t)text.size() + 2 * n;
    std::vector<int32_t> ids((size_t)cap), mask((size_t)cap), nreal((size_t)n);
    std::vector<int64_t> row((size_t)n + 1);
    phase("candidate words");
    c->building_words = true;
    int rc = encode_host_locked(c, text.data(), off.data(), nullptr, nullptr, n, 0, GZ_MAX_LEN_NONE | GZ_NO_WORD_TABLE, cap,
                                ids.data(), mask.data(), nullptr, nullptr, row.data(), nullptr, nreal.data(), nullptr);
    c->building_words = false;
    if (rc) return rc;
    phase("GPU merge path over the candidates");
    std::vector<GzWordSlot2> found2;                         // 17..32 bytes: the long-key table (probed once per miss)
    std::vector<GzWordSlot1> found1;                         // <= 16 bytes: perfectly hashed, one probe per word
    std::vector<uint32_t> found1_sym;                        // ... and the symbol (word + "</w>") each of them is
    for (int64_t i = 0; i < n; ++i) {
        if (row[i + 1] - row[i] != 3) continue;
        const int32_t id = ids[row[i] + 1];
        if (id < 0 || id >= (1 << 26)) continue;
        const uint32_t len = (uint32_t)(off[i + 1] - off[i]);
        uint8_t key[32] = {0};
        std::memcpy(key, text.data() + off[i], len);
        if (len <= 16) {
            GzWordSlot1 e1{};
            e1.meta = len | ((uint32_t)id << 5);
            std::memcpy(e1.k, key, 12); std::memcpy(&e1.k3, key + 12, 4);
            found1.push_back(e1);
            found1_sym.push_back(which[(size_t)i]);
        } else {
            GzWordSlot2 e{{0, 0, 0, 0}, len, id, {0, 0, 0, 0, 0, 0}};
            std::memcpy(e.k, key, 32);
            found2.push_back(e);
        }
    }
    if (found1.empty()) return GZ_OK;            // (the long-key table is only consulted for misses of the first)
    {
        std::vector<uint32_t> slot_of;
        auto hashes = [](const void* ctx, size_t i, uint32_t k1, uint32_t k2, uint32_t* ha, uint32_t* hb) {
            const GzWordSlot1& e = (*static_cast<const std::vector<GzWordSlot1>*>(ctx))[i];
            *hb = gz_word1_hb(gz_slot1_lo(e), gz_slot1_hi(e), e.meta & 31u, k1, k2);
            *ha = gz_word1_ha_of(*hb);
        };
        // the words running text consists of (by the counts the vocab file carries; without counts: the file's order) share the
        // table's first lines: a hint to the builder, see gz_ph_build
        phase("results");
        const uint32_t hot_slots = gz_word_hot_slots();
        std::vector<uint8_t> hotf;
        std::vector<std::pair<uint64_t, uint32_t>> by_weight;       // (weight, index into found1), heaviest first
        {
            // a recorded word's id IS its vocab entry's id (the GPU looked the whole word up): the weights go by id.  The counts a vocab
            // file carries are believed when the file is sorted by them (as the tool that writes such files leaves it); a file without
            // counts, or with counts in no order, says nothing about frequency -- the MERGES do: they were learned most frequent
            // first, and a whole word's symbol is numbered by the merge that completed it (symbols no merge produces -- single
            // characters -- count as most frequent)
            bool any = false;
            size_t rising = 0;
            for (size_t i = 0; i < H.enc_count.size(); ++i) { any |= H.enc_count[i] != 0; rising += i > 5 && H.enc_count[i] > H.enc_count[i - 1]; }
            bool by_count = any && rising * 100 <= H.enc_count.size();
            if (gz_default_options().word_weights) by_count = any && gz_default_options().word_weights == 1;       // (A/B: 1 counts, 2 merge ranks)
            std::vector<uint64_t> weight_of_id(H.enc_words.size(), 0);          // (ids are sizes of the dict at insertion: < its size)
            if (by_count)
                for (size_t i = 0; i < H.enc_words.size(); ++i) {
                    const int32_t id = H.enc_ids[i];
                    if (id >= 0 && (size_t)id < weight_of_id.size()) weight_of_id[(size_t)id] = H.enc_count[i];
                }
            const uint32_t unk = (uint32_t)H.special_ids[4];                    // (a word the vocab does not hold: no weight)
            const uint64_t n_lines = H.merges.size();
            std::vector<std::pair<uint64_t, uint32_t>> order(found1.size());
            for (size_t i = 0; i < found1.size(); ++i) {
                const uint32_t id = found1[i].meta >> 5;
                uint64_t w = 0;
                if (id != unk) w = by_count ? (id < weight_of_id.size() ? weight_of_id[id] : 0) : (found1_sym[i] < n_lines ? n_lines - found1_sym[i] : n_lines + 1);
                order[i] = {w, (uint32_t)i};
            }
            std::sort(order.begin(), order.end(), [](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
            if (hot_slots) {
                hotf.assign(found1.size(), 0);
                const size_t nhot = std::min<size_t>(found1.size(), (size_t)hot_slots * 7 / 8);     // (an eighth of the region stays free: the last ones still find a slot)
                for (size_t i = 0; i < nhot; ++i) hotf[order[i].second] = 1;
            }
            by_weight = std::move(order);
        }
        phase("weights");
        GzPhHost& ph = W.ph;
        gz_ph_build(found1.size(), hashes, &found1, ph, slot_of, hotf.empty() ? nullptr : hotf.data(), hot_slots);
        phase("perfect hash");
        W.tabp.assign(ph.slots, GzWordSlot1{});
        for (size_t i = 0; i < found1.size(); ++i) W.tabp[slot_of[i]] = found1[i];
        // the word kernel's LDS hot set: direct-mapped by the top bits of ha (the seeds the builder settled on), heaviest first
        W.hot.assign(GZ_WORD_HOT_SLOTS, GzWordHot{{0, 0, 0}, 0});
        for (const auto& o : by_weight) {
            const GzWordSlot1& e = found1[o.second];
            if ((e.meta & 31u) > 12u) continue;
            GzWordHot& h = W.hot[gz_word1_ha(gz_slot1_lo(e), gz_slot1_hi(e), e.meta & 31u, ph.k1, ph.k2) >> GZ_WORD_HOT_SHIFT];
            if (h.meta == 0) h = GzWordHot{{e.k[0], e.k[1], e.k[2]}, e.meta};
        }
    }
    if (!found2.empty()) {
        size_t slots2 = 16;
        while (slots2 < gz_tab_slack() * found2.size()) slots2 <<= 1;
        std::vector<GzWordSlot2>& tab2 = W.tab2;
        tab2.assign(slots2, GzWordSlot2{{0, 0, 0, 0}, 0, 0, {0, 0, 0, 0, 0, 0}});
        for (const GzWordSlot2& e : found2) {
            size_t h = gz_word_hash2(e.k, e.len) & (slots2 - 1);
            while (tab2[h].len != 0) h = (h + 1) & (slots2 - 1);
            tab2[h] = e;
        }
    }
    W.n_words = (int64_t)(found1.size() + found2.size());
    phase("hot set, long-key table");
    return GZ_OK;
}

// The whole-word tables onto the device (built just now, or read from the cache) and into the table descriptor.
static int install_word_tables(gz_ctx* c, const WordImages& W)
{
    int rc;
    if (W.tabp.empty()) return GZ_OK;
    if ((rc = upload(c, c->t_words0p, W.tabp)) || (rc = upload(c, c->t_word0_disp, W.ph.disp)) || (rc = upload(c, c->t_word_hot, W.hot))) return rc;
    if (!W.tab2.empty() && (rc = upload(c, c->t_words2, W.tab2))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));                  // (the images may die once this returns)
    GzDeviceTables& D = c->dev;
    D.words0p = (const GzWordSlot1*)c->t_words0p.p;
    D.word0_ph = GzPh{(const uint16_t*)c->t_word0_disp.p, W.ph.nbuckets, W.ph.bshift, W.ph.sshift, W.ph.slots - 1, W.ph.k1, W.ph.k2};
    D.word0_ovf = W.ph.n_overflow;
    D.word_hot = (const GzWordHot*)c->t_word_hot.p;
    if (!W.tab2.empty()) { D.words2 = (const GzWordSlot2*)c->t_words2.p; D.word2_mask = (uint32_t)W.tab2.size() - 1; }
    { int rcs = copy_in(c, c->t_struct.p, &c->dev, sizeof(GzDeviceTables), c->stream); if (rcs) return rcs; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_words = W.n_words;
    return GZ_OK;
}


// =================================================================================================================
extern "C" {

int gz_version(void) { return GZ_VERSION; }

int gz_create(int device_id, gz_ctx** out)
try {
    if (!out) return GZ_E_INVALID;
    *out = nullptr;
    static const bool load_timing = getenv("GZ_LOAD_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!load_timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "gz_create: %-38s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, GZ_E_NODEVICE, "no HIP device (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, GZ_E_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, GZ_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GZ_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    phase("device count and properties");
    gz_ctx* c = new (std::nothrow) gz_ctx();
    if (!c) return fail(nullptr, GZ_E_NOMEM, "out of host memory");
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->h_flags, 64, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return fail(nullptr, GZ_E_HIP, "stream / pinned memory creation failed");
    }
    std::memset(c->h_flags, 0, 64);
    phase("device, first stream, pinned flags");
    c->opt = gz_default_options();
    for (auto& ev : c->ev) hipEventCreate(&ev);
    hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    for (auto& a : c->ev_sf0) for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& a : c->ev_sb) for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& a : c->ev_sf) for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& a : c->ev_sj) for (auto& e : a) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : c->ev_tok) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming);
    phase("streams and events");
    if (hipHostMalloc((void**)&c->h_pick, 512, hipHostMallocDefault) != hipSuccess) { gz_destroy(c); return fail(nullptr, GZ_E_NOMEM, "pinned memory creation failed"); }
    std::memset(c->h_pick, 0, 512);
    phase("pinned pick buffer");
    if (ensure(c, c->w_flags, 64, /* zeroed: word 12 is only ever raised */ true) != GZ_OK) { g_create_err = c->err; gz_destroy(c); return GZ_E_NOMEM; }
    phase("device flags");
    *out = c;
    return GZ_OK;
} GZ_CATCH(nullptr)

void gz_destroy(gz_ctx* c)
try {
    if (!c) return;
    hipSetDevice(c->device);
    // every stream of the context drained BEFORE any event or stream goes (an event destroyed under a stream that still waits for it,
    // a stream destroyed under its last copy: nothing here should depend on how the runtime treats those)
    for (hipStream_t q : {c->stream, c->stream2, c->side, c->xstream, c->s_in, c->s_out}) if (q) hipStreamSynchronize(q);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    for (auto& e : c->ev_tok) if (e) hipEventDestroy(e);
    for (auto& e : c->ev_blk) if (e) hipEventDestroy(e);
    if (c->ev_x) hipEventDestroy(c->ev_x);
    if (c->xstream) hipStreamDestroy(c->xstream);
    if (c->side) hipStreamDestroy(c->side);
    for (auto& a : c->ev_sf0) for (auto& e : a) if (e) hipEventDestroy(e);
    for (auto& a : c->ev_sb) for (auto& e : a) if (e) hipEventDestroy(e);
    for (auto& a : c->ev_sf) for (auto& e : a) if (e) hipEventDestroy(e);
    for (auto& a : c->ev_sj) for (auto& e : a) if (e) hipEventDestroy(e);
    if (c->s_in) hipStreamDestroy(c->s_in);
    if (c->s_out) hipStreamDestroy(c->s_out);
    for (auto& e : c->ev_in) if (e) hipEventDestroy(e);
    for (auto& e : c->ev_done) if (e) hipEventDestroy(e);
    if (c->h_tot) hipHostFree(c->h_tot);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (auto& h : c->h_xfer) if (h) hipHostFree(h);
    for (auto& h : c->h_tin) if (h) hipHostFree(h);
    for (auto& h : c->h_cout) if (h) hipHostFree(h);
    for (auto& e : c->ev_xfer) if (e) hipEventDestroy(e);
    release(c->w_stage);
    for (DBuf* b : {&c->w_csr_ids[0], &c->w_csr_ids[1], &c->w_csr_mask[0], &c->w_csr_mask[1], &c->w_csr_comp, &c->w_csr_nreal, &c->w_csr_off32}) release(*b);
    if (c->h_pick) hipHostFree(c->h_pick);
    release(c->w_pick); release(c->w_rowoff32);
    for (auto& t2 : c->w_tiny) for (auto& t : t2) release(t);
    for (DBuf* b : {&c->w_pp[0], &c->w_pp[1], &c->w_ppoff[0], &c->w_ppoff[1], &c->w_pplen, &c->w_ppaux, &c->w_pp_in, &c->w_pp_inoff, &c->w_ppctl, &c->w_pplen32, &c->w_pplb}) release(*b);
    for (DBuf* b : {&c->t_dec_entries, &c->t_dec_bytes, &c->w_dec_ids, &c->w_dec_roff, &c->w_dec_rb, &c->w_dec_ooff, &c->w_dec_out}) release(*b);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    for (DBuf* b : {&c->t_pair8, &c->t_pair_disp, &c->t_words0p, &c->t_word0_disp, &c->t_pair_hot, &c->t_word_hot}) release(*b);
    for (DBuf* b : {&c->t_merges, &c->t_symids, &c->t_bmp, &c->t_astral, &c->t_struct, &c->t_words2, &c->w_text, &c->w_toff, &c->w_pair,
                    &c->w_poff, &c->w_ids, &c->w_mask, &c->w_tt, &c->w_seq, &c->w_rowoff, &c->w_rowlen, &c->w_pairlen,
                    &c->w_nreal, &c->w_status, &c->w_raw, &c->w_arena, &c->w_flags, &c->w_word, &c->w_wordout})
        release(*b);
    for (auto& slot : c->tw) for (auto& t : slot)
        for (DBuf* b : {&t.brk, &t.st, &t.en, &t.blkcnt, &t.docw0, &t.wtok, &t.waux, &t.mtok, &t.mlist, &t.blkmiss, &t.tilecnt, &t.wlist, &t.grpblk, &t.lookback, &t.mq}) release(*b);
    for (auto& ev : c->ev) if (ev) hipEventDestroy(ev);
    for (auto& pr : c->ring) { if (pr[0]) hipEventDestroy(pr[0]); if (pr[1]) hipEventDestroy(pr[1]); }
    for (auto& pr : c->xring) { if (pr[0]) hipEventDestroy(pr[0]); if (pr[1]) hipEventDestroy(pr[1]); }
    if (c->h_flags) hipHostFree(c->h_flags);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
} GZ_CATCH_VOID

const char* gz_last_error(gz_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int gz_load_tables(gz_ctx* c, const uint8_t* vocab, size_t vocab_len, const uint8_t* bpe, size_t bpe_len,
                   const char* const specials[5])
try {
    if (!c || (!vocab && vocab_len) || (!bpe && bpe_len) || !specials) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    for (int i = 0; i < 5; ++i) if (!specials[i]) return fail(c, GZ_E_INVALID, "special token %d is NULL", i);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->pend.active) { int rc = sync_locked(c); if (rc) return rc; }
    c->have_tables = false;
    int rc = GZ_OK;
    static const bool load_timing = getenv("GZ_LOAD_TIMING") != nullptr;    // phase times of this call on stderr (tools/t_load.py)
    auto t_last = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!load_timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "gz_load_tables: %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    static const uint8_t empty = 0;
    if (!vocab) vocab = &empty;
    if (!bpe) bpe = &empty;
    // ---- the table cache: the finished images of these very files, if a verified copy exists (gz_cache.inc)
    WordImages W;
    uint8_t key[32];
    const std::string cdir = cache_dir();
    std::string cpath;
    bool hit = false;
    c->cache_status = 0;
    try {
        if (!cdir.empty()) {
            cache_key(vocab, vocab_len, bpe, bpe_len, specials, key);
            cpath = cache_file(cdir, key);
            const int got = cache_dir_usable(cdir, false) ? cache_read(cpath, key, c->host, W) : 0;
            hit = got == 1;
            c->cache_status = got == 1 ? 1 : got < 0 ? 3 : 2;      // 1 hit, 2 miss, 3 a file was there and was refused
        }
        if (!hit) rc = gz_build_tables(vocab, vocab_len, bpe, bpe_len, specials, c->host, c->err);
    } catch (const std::bad_alloc&) {
        return fail(c, GZ_E_NOMEM, "out of host memory while building tables");
    } catch (...) {
        return fail(c, GZ_E_INVALID, "unexpected failure while building tables");
    }
    if (rc) return rc;
    phase(hit ? "cache read" : "host build (gz_build_tables)");
    GzHostTables& H = c->host;
    if ((rc = upload(c, c->t_merges, H.merges))) return rc;
    if ((rc = upload(c, c->t_symids, H.sym_ids))) return rc;
    if ((rc = upload(c, c->t_bmp, H.bmp))) return rc;
    if (!H.astral.empty()) { if ((rc = upload(c, c->t_astral, H.astral))) return rc; }
    if ((rc = upload(c, c->t_pair8, H.pair8)) || (rc = upload(c, c->t_pair_disp, H.pair_ph.disp)) || (rc = upload(c, c->t_pair_hot, H.pair_hot))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    GzDeviceTables& D = c->dev;
    D.merges = (const GzMergeInfo*)c->t_merges.p; D.n_ranks = (uint32_t)H.merges.size();
    D.sym_ids = (const GzSymIds*)c->t_symids.p;  D.n_symbols = (uint32_t)H.symbols.size();
    D.bmp = (const GzCpSyms*)c->t_bmp.p;
    D.astral = H.astral.empty() ? nullptr : (const GzAstral*)c->t_astral.p;
    D.astral_mask = H.astral.empty() ? 0 : (uint32_t)H.astral.size() - 1;
    D.pad_id = H.special_ids[0]; D.bos_id = H.special_ids[1]; D.eos_id = H.special_ids[2]; D.unk_id = H.special_ids[4];
    D.words2 = nullptr; D.word2_mask = 0;
    D.pair8 = (const GzPair8*)c->t_pair8.p;
    D.pair_ph = GzPh{(const uint16_t*)c->t_pair_disp.p, H.pair_ph.nbuckets, H.pair_ph.bshift, H.pair_ph.sshift, H.pair_ph.slots - 1, H.pair_ph.k1, H.pair_ph.k2};
    D.words0p = nullptr; D.word0_ph = GzPh{nullptr, 16, 28, 28, 15, 1, 1};
    D.pair_ovf = H.pair_ph.n_overflow; D.word0_ovf = 0;
    D.pair_hot = (const GzPair8*)c->t_pair_hot.p; D.pair_hot_shift = GZ_PAIR_HOT_SHIFT;
    D.word_hot = nullptr;
    if (c->host.enc_words.size() >= (1u << 26)) return fail(c, GZ_E_LIMIT, "vocab has 2^26 or more entries");
    if ((rc = ensure(c, c->t_struct, sizeof(GzDeviceTables)))) return rc;
    { int rcs = copy_in(c, c->t_struct.p, &c->dev, sizeof(GzDeviceTables), c->stream); if (rcs) return rcs; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_tables = true;
    phase("upload of the pair / symbol tables");
    if (!hit) {
        rc = build_word_table(c, W);                             // (runs the GPU merge path over every candidate word)
        if (rc) { c->have_tables = false; return rc; }
        phase("whole-word tables (GPU merge path + perfect hash)");
    }
    rc = install_word_tables(c, W);
    if (rc) { c->have_tables = false; return rc; }
    phase("upload of the whole-word tables");
    if (!hit && !cpath.empty()) {
        bool written = false;
        try { written = cache_write(cdir, cpath, key, c->host, W); } catch (...) { /* a cache that cannot be written is only a slower start */ }
        if (!written) c->cache_status = 4;                       // 4: rebuilt, and the file could NOT be written (directory missing / shared / full)
        phase("cache write");
    }
    return GZ_OK;
} GZ_CATCH(c)

int gz_debug_set(gz_ctx* c, const char* key, int64_t value)
try {
    // c == NULL: the process-wide defaults (contexts created from now on, and the table builder); else this context
    int rc;
    if (c) { std::lock_guard<std::mutex> lk(c->mu); rc = gz_option_set(c->opt, key, value); }
    else rc = gz_option_set(gz_default_options(), key, value);
    if (rc) return fail(c, GZ_E_INVALID, "gz_debug_set: unknown key or value out of range: %s = %lld", key ? key : "(null)", (long long)value);
    return GZ_OK;
} GZ_CATCH(c)

int gz_table_cache_status(gz_ctx* c)
try {
    return c ? c->cache_status : GZ_E_INVALID;
} GZ_CATCH(c)

// SHA-256 over every device-resident table image and the descriptor's scalar fields: two contexts whose tables were built
// from the same files -- one by the builder, one from the cache -- must give the same digest.
int gz_table_digest(gz_ctx* c, uint8_t out[32])
try {
    if (!c || !out) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->pend.active) { int rc = sync_locked(c); if (rc) return rc; }
    Sha256 sha;
    std::vector<uint8_t> buf;
    auto add = [&](const void* dptr, size_t bytes) -> int {
        const uint64_t n = bytes;
        sha.update(&n, 8);
        if (!bytes) return GZ_OK;
        alloc_site(c);
        buf.resize(bytes);
        { int rcs = copy_out(c, buf.data(), dptr, bytes, c->stream); if (rcs) return rcs; }
        sha.update(buf.data(), bytes);
        return GZ_OK;
    };
    const GzDeviceTables& D = c->dev;
    const GzHostTables& H = c->host;
    int rc;
    if ((rc = add(D.merges, (size_t)D.n_ranks * sizeof(GzMergeInfo))) ||
        (rc = add(D.sym_ids, (size_t)D.n_symbols * sizeof(GzSymIds))) || (rc = add(D.bmp, 65536 * sizeof(GzCpSyms))) ||
        (rc = add(D.astral, D.astral ? ((size_t)D.astral_mask + 1) * sizeof(GzAstral) : 0)) ||
        (rc = add(D.pair8, ((size_t)D.pair_ph.mask + 1) * sizeof(GzPair8))) || (rc = add(D.pair_ph.disp, (size_t)D.pair_ph.nbuckets * 2)) ||
        (rc = add(D.pair_hot, GZ_PAIR_HOT_SLOTS * sizeof(GzPair8))) ||
        (rc = add(D.words2, D.words2 ? ((size_t)D.word2_mask + 1) * sizeof(GzWordSlot2) : 0)) ||
        (rc = add(D.words0p, D.words0p ? ((size_t)D.word0_ph.mask + 1) * sizeof(GzWordSlot1) : 0)) ||
        (rc = add(D.word0_ph.disp, D.words0p ? (size_t)D.word0_ph.nbuckets * 2 : 0)) ||
        (rc = add(D.word_hot, D.word_hot ? GZ_WORD_HOT_SLOTS * sizeof(GzWordHot) : 0)))
        return rc;
    const uint32_t scal[] = {D.n_ranks, D.n_symbols, D.astral_mask, (uint32_t)D.pad_id, (uint32_t)D.bos_id, (uint32_t)D.eos_id,
                             (uint32_t)D.unk_id, D.word2_mask, D.pair_ph.nbuckets, D.pair_ph.bshift, D.pair_ph.sshift, D.pair_ph.mask,
                             D.pair_ph.k1, D.pair_ph.k2, D.word0_ph.nbuckets, D.word0_ph.bshift, D.word0_ph.sshift, D.word0_ph.mask, D.word0_ph.k1,
                             D.word0_ph.k2, D.pair_ovf, D.word0_ovf, D.pair_hot_shift};
    sha.update(scal, sizeof scal);
    // the host side of the tables (what the Python surface reads back)
    for (const std::string& w : H.enc_words) { const uint32_t l = (uint32_t)w.size(); sha.update(&l, 4); sha.update(w.data(), l); }
    if (!H.enc_ids.empty()) sha.update(H.enc_ids.data(), H.enc_ids.size() * 4);
    for (const std::string& w : H.rank_keys) { const uint32_t l = (uint32_t)w.size(); sha.update(&l, 4); sha.update(w.data(), l); }
    if (!H.rank_vals.empty()) sha.update(H.rank_vals.data(), H.rank_vals.size() * 4);
    for (const std::string& w : H.symbols) { const uint32_t l = (uint32_t)w.size(); sha.update(&l, 4); sha.update(w.data(), l); }
    sha.final(out);
    return GZ_OK;
} GZ_CATCH(c)

int gz_table_info(gz_ctx* c, int32_t* vocab_size, int32_t special_ids[5], int32_t* n_ranks, int32_t* n_symbols)
try {
    if (!c) return GZ_E_INVALID;
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    if (vocab_size) *vocab_size = (int32_t)c->host.enc_words.size();
    if (special_ids) for (int i = 0; i < 5; ++i) special_ids[i] = c->host.special_ids[i];
    if (n_ranks) *n_ranks = (int32_t)c->host.rank_keys.size();
    if (n_symbols) *n_symbols = (int32_t)c->host.symbols.size();
    return GZ_OK;
} GZ_CATCH(c)

int gz_vocab_entry(gz_ctx* c, int64_t i, const uint8_t** utf8, int32_t* len, int32_t* id)
try {
    if (!c || !c->have_tables) return c ? fail(c, GZ_E_NOTABLES, "no tables") : GZ_E_INVALID;
    if (i < 0 || i >= (int64_t)c->host.enc_words.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)c->host.enc_words[i].data();
    if (len) *len = (int32_t)c->host.enc_words[i].size();
    if (id) *id = c->host.enc_ids[i];
    return GZ_OK;
} GZ_CATCH(c)

int gz_merge_entry(gz_ctx* c, int64_t i, const uint8_t** utf8, int32_t* len, int32_t* n_fields, int32_t* rank)
try {
    if (!c || !c->have_tables) return c ? fail(c, GZ_E_NOTABLES, "no tables") : GZ_E_INVALID;
    if (i < 0 || i >= (int64_t)c->host.rank_keys.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)c->host.rank_keys[i].data();
    if (len) *len = (int32_t)c->host.rank_keys[i].size();
    if (n_fields) *n_fields = c->host.rank_nfields[i];
    if (rank) *rank = c->host.rank_vals[i];
    return GZ_OK;
} GZ_CATCH(c)

int gz_symbol_utf8(gz_ctx* c, int32_t symbol, const uint8_t** utf8, int32_t* len)
try {
    if (!c || !c->have_tables) return c ? fail(c, GZ_E_NOTABLES, "no tables") : GZ_E_INVALID;
    if (symbol < 0 || symbol >= (int32_t)c->host.symbols.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)c->host.symbols[symbol].data();
    if (len) *len = (int32_t)c->host.symbols[symbol].size();
    return GZ_OK;
} GZ_CATCH(c)

int gz_encode_batch_device(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                           const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                           int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                           int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    // (an arming of gz_encode_emit_block holds for ONE call of the DEVICE entry points, whether that call takes it or fails; host
    //  calls and gz_load_tables -- which encode through the same function -- never see it)
    int32_t* const blk = c->emit_block; c->emit_block = nullptr;
    struct Flag { bool& f; explicit Flag(bool& x) : f(x) { f = true; } ~Flag() { f = false; } } caller(c->caller_buffers);
    return encode_device_locked(c, text, text_off, pair, pair_off, n_docs, max_len, flags, capacity, input_ids,
                                attention_mask, token_type_ids, sequence_id, row_off, pair_len, n_real, status, nullptr, nullptr, blk, c->emit_bits);
} GZ_CATCH(c)

int gz_encode_batch_device_h(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                             const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                             int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                             int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status,
                             const int64_t* text_off_host, const int64_t* pair_off_host)
try {
    if (!c) return GZ_E_INVALID;
    if (!text_off_host || ((pair_off != nullptr) != (pair_off_host != nullptr)))
        return fail(c, GZ_E_INVALID, "host copies of the offsets are required for every text");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    int32_t* const blk = c->emit_block; c->emit_block = nullptr;        // (as above)
    struct Flag { bool& f; explicit Flag(bool& x) : f(x) { f = true; } ~Flag() { f = false; } } caller(c->caller_buffers);
    return encode_device_locked(c, text, text_off, pair, pair_off, n_docs, max_len, flags, capacity, input_ids,
                                attention_mask, token_type_ids, sequence_id, row_off, pair_len, n_real, status,
                                text_off_host, pair_off_host, blk, c->emit_bits);
} GZ_CATCH(c)

int gz_sync(gz_ctx* c)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return sync_locked(c);
} GZ_CATCH(c)

// ---- large host calls: sub-batches, copies overlapped with the kernels ---------------------------------------------------
// Host buffers in; out either CSR (gz_encode_batch_csr: the rows' real entries + row lengths) or DENSE (gz_encode_batch with
// padding + truncation: [n_docs, max_len] ids + mask).  The batch is cut into sub-batches of documents:
//   caller's thread: text of sub-batch k + 1 into a pinned buffer (worker threads), onto the copy-in stream, while
//   main stream    : the pipeline of sub-batch k runs (dense rows into one of two device slots), then the rows lose
//                    their padding (scan of n_real + gz_compact_kernel) into the sub-batch's region of a device buffer,
//   copy-out stream: the real entries of sub-batch k - 1 travel back into a pinned slot,
//   worker threads : sub-batch k - 2's entries go to their final place in the caller's array -- or, dense form, are PADDED ON
//                    THE HOST into the caller's [n_docs, max_len] arrays.
// What crosses PCIe either way is the text, 8 bytes per document of offsets, and 2 (or 4) bytes per real token + 4 per document:
// for the dense form 0.1 GB instead of the 2 GB of mostly padding its arrays hold (1 M documents, max_len 256).
// Buffers that came from gz_host_alloc skip the pinned staging (they are pinned): copies go straight in / out.
struct CsrOut {
    void* tokens = nullptr; int64_t capacity = 0;       // CSR form: entries back to back (capacity in entries) ...
    int32_t* n_real = nullptr;                          // ... and the row lengths (dense form: may be null)
    int32_t* ids = nullptr; int32_t* mask = nullptr;    // dense form
};

static int csr_core(gz_ctx* c, const uint8_t* text, const int64_t* text_off, int64_t n_docs, int32_t max_len, uint32_t flags, int bits,
                    const CsrOut& O, int64_t* total_out)
{
    const bool dense_out = O.ids != nullptr;
    const int64_t tb = text_off[n_docs] - text_off[0];
    if (c->pend.active) { int rc0 = sync_locked(c); if (rc0) return rc0; }
    if (c->x_used) HIPCHK(c, hipStreamSynchronize(c->xstream));
    if (!c->s_in) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (auto& e : c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : c->ev_done) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(c, hipHostMalloc((void**)&c->h_tot, sizeof(uint32_t) * gz_ctx::CSR_SUBS, hipHostMallocDefault));
    }
    // sub-batches of about 32 MB of text (equal document counts), at most CSR_SUBS
    int nsub = (int)((tb + (32ll << 20) - 1) / (32ll << 20));
    if (dense_out) {                                            // ... and of at most 256 MB of dense rows (short documents, long rows)
        const int64_t by_rows = (n_docs * (int64_t)max_len * 8 + (256ll << 20) - 1) / (256ll << 20);
        if (by_rows > nsub) nsub = (int)std::min<int64_t>(by_rows, gz_ctx::CSR_SUBS);
    }
    if (nsub < 1) nsub = 1;
    if (nsub > gz_ctx::CSR_SUBS) nsub = gz_ctx::CSR_SUBS;
    if ((int64_t)nsub > n_docs) nsub = (int)n_docs;
    const size_t esz = bits == 16 ? 2 : 4;
    auto lo_of = [&](int k) { return (int64_t)k * n_docs / nsub; };
    int64_t nmax = 0, bmax = 0;
    for (int k = 0; k < nsub; ++k) {
        nmax = std::max(nmax, lo_of(k + 1) - lo_of(k));
        bmax = std::max(bmax, text_off[lo_of(k + 1)] - text_off[lo_of(k)]);
    }
    // a document of b bytes has at most min(max_len, b + 2) entries: regions of the device compact buffer start at these bounds
    alloc_site(c);
    std::vector<int64_t> bound((size_t)nsub + 1, 0);
    for (int k = 0; k < nsub; ++k) {
        const int64_t lo = lo_of(k), hi = lo_of(k + 1);
        const int64_t by_bytes = (text_off[hi] - text_off[lo]) + 2 * (hi - lo), by_rows = (hi - lo) * (int64_t)max_len;
        int64_t b = by_bytes < by_rows ? by_bytes : by_rows;
        b = (b + 7) & ~(int64_t)7;                                          // regions stay 16-byte aligned for either entry size
        bound[(size_t)k + 1] = bound[(size_t)k] + b;
    }
    int rc;
    if ((rc = ensure(c, c->w_text, (size_t)tb + 16)) || (rc = ensure(c, c->w_toff, (size_t)(n_docs + 1) * 8)) ||
        (rc = ensure(c, c->w_csr_nreal, (size_t)(n_docs + 1) * 4)) || (rc = ensure(c, c->w_csr_off32, (size_t)(nmax + 2) * 4)) ||
        (rc = ensure(c, c->w_csr_comp, (size_t)bound[(size_t)nsub] * esz + 64)))
        return rc;
    for (int q = 0; q < (nsub > 1 ? 2 : 1); ++q)
        if ((rc = ensure(c, c->w_csr_ids[q], (size_t)nmax * (size_t)max_len * 4 + 64)) || (rc = ensure(c, c->w_csr_mask[q], (size_t)nmax * (size_t)max_len * 4 + 64)))
            return rc;
    const GzDeviceTables* T = (const GzDeviceTables*)c->t_struct.p;
    const uint8_t* d_text = (const uint8_t*)c->w_text.p - text_off[0];
    const int64_t* d_off = (const int64_t*)c->w_toff.p;
    const int use_words = use_words_flags(c, flags);
    hipStream_t s = c->stream;
    if ((rc = need_side_streams(c, false))) return rc;

    // which of the caller's buffers are pinned (gz_host_alloc): those are copied directly
    const bool text_direct = tb >= 65536 && is_pinned(text + text_off[0], (size_t)tb);
    const bool out_direct = !dense_out && is_pinned(O.n_real, (size_t)n_docs * 4) && (O.capacity == 0 || is_pinned(O.tokens, (size_t)O.capacity * esz));
    if (!text_direct)
        for (int q = 0; q < (nsub > 1 ? 2 : 1); ++q)
            if ((rc = pinned_need(c, c->h_tin[q], c->h_tin_cap[q], (size_t)bmax + 16))) return rc;
    // (worker threads only when something has to be moved by the host: pinned text in and pinned CSR rows out need none)
    HostPool pool((text_direct && out_direct) ? 0 : pool_threads(c, dense_out ? (size_t)n_docs * (size_t)max_len * 8 : (size_t)tb));
    alloc_site(c);
    std::vector<uint32_t> row_first[3];                          // dense form: where each row of the slot's sub-batch starts among its entries
    if (dense_out) {
        for (auto& v : row_first) v.resize((size_t)nmax + 1);
        if (c->opt.host_hints & 1) {
            hint_huge(O.ids, (size_t)n_docs * (size_t)max_len * 4);
            hint_huge(O.mask, (size_t)n_docs * (size_t)max_len * 4);
        }
    }
    // whatever happens below, nothing of this call is in flight when it returns: the streams are drained, then the pool's threads
    // finish what is queued and are joined (its destructor)
    struct Drain { gz_ctx* c; ~Drain() { hipStreamSynchronize(c->s_in); hipStreamSynchronize(c->stream); hipStreamSynchronize(c->s_out); } } drain{c};

    // sub-batch k's text onto the copy-in stream (ev_in[k] behind it)
    auto text_in = [&](int k) -> int {
        const int64_t lo = lo_of(k), hi = lo_of(k + 1);
        const int64_t b0 = text_off[lo], b1 = text_off[hi];
        if (b1 > b0) {
            uint8_t* dst = (uint8_t*)c->w_text.p + (b0 - text_off[0]);
            if (text_direct) HIPCHK(c, hipMemcpyAsync(dst, text + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, c->s_in));
            else {
                uint8_t* h = c->h_tin[k & 1];
                if (k >= 2) HIPCHK(c, hipEventSynchronize(c->ev_in[k - 2]));        // (the copy that last read this buffer)
                const uint8_t* src = text + b0;
                pool.parallel((size_t)(b1 - b0), (size_t)1 << 20, [=](size_t a, size_t b) { std::memcpy(h + a, src + a, b - a); });
                HIPCHK(c, hipMemcpyAsync(dst, h, (size_t)(b1 - b0), hipMemcpyHostToDevice, c->s_in));
            }
        }
        HIPCHK(c, hipEventRecord(c->ev_in[k], c->s_in));
        return GZ_OK;
    };
    // Pinned text: EVERY copy-in is queued before anything else -- the bus never waits for the host.  The offsets -- 8 bytes per
    // document, pageable as a rule: staged piece by piece -- go on the MAIN stream, in front of the kernels that read them, while the first
    // sub-batch's text is already on its way.
    if (text_direct) for (int k = 0; k < nsub; ++k) if ((rc = text_in(k))) return rc;
    if ((rc = copy_in(c, c->w_toff.p, text_off, (size_t)(n_docs + 1) * 8, s))) return rc;
    HIPCHK(c, hipMemsetAsync(c->w_flags.p, 0, 16, s));
    const int32_t pad_id = c->dev.pad_id;
    int64_t total = 0;
    int ret = GZ_OK;
    // sub-batches whose kernels are enqueued ahead of the one whose rows the host takes: all of them when the text is pinned, two when it
    // goes through the two pinned text buffers
    const int LAG = text_direct ? nsub : 2;
    for (int step = 0; step < nsub + LAG; ++step) {
        if (step < nsub) {
            // ---- sub-batch k: text in (unless it is on its way already), kernels
            const int k = step;
            const int64_t lo = lo_of(k), hi = lo_of(k + 1);
            if (!text_direct && (rc = text_in(k))) return rc;
            GzAsmArgs A{};
            A.n_texts = 1; A.n_docs = hi - lo; A.dense = 1; A.max_len = max_len;
            A.ids = (int32_t*)c->w_csr_ids[k & 1].p; A.mask = (int32_t*)c->w_csr_mask[k & 1].p;
            A.raw = nullptr; A.n_real = (int32_t*)c->w_csr_nreal.p + lo;
            A.docs_per_wave = GZ_MAX_DOCS_PER_WAVE;
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_in[k], 0));
            if ((rc = setup_text(c, c->tw[k & 1][0], c->w_tiny[k & 7][0], d_text + text_off[lo], d_off + lo, text_off[hi] - text_off[lo], A.n_docs, s, A.X[0]))) return rc;
            gz_launch_pipeline_text(c->opt, T, c->dev, A.X[0], A.n_docs, use_words, (int32_t*)c->w_flags.p + 3, s, c->side, c->ev_sf0[k & 1][0], c->ev_sf[k & 1][0], c->ev_sj[k & 1][0]);
            gz_launch_assemble(c->opt, T, A, s);
            uint32_t* off32 = (uint32_t*)c->w_csr_off32.p;
            gz_launch_row_offsets(A.n_real, A.n_docs, off32, s);
            gz_launch_compact(A.ids, off32, A.n_docs, max_len, (uint8_t*)c->w_csr_comp.p + (size_t)bound[(size_t)k] * esz, bits, nullptr, s);
            HIPCHK(c, hipMemcpyAsync(&c->h_tot[k], off32 + A.n_docs, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipEventRecord(c->ev_done[k], s));
            HIPCHK(c, hipGetLastError());
        }
        if (step >= LAG) {
            // ---- sub-batch j is compact: its entries come back
            const int j = step - LAG;
            const int64_t lo = lo_of(j), hi = lo_of(j + 1), nj = hi - lo;
            HIPCHK(c, hipEventSynchronize(c->ev_done[j]));
            const int64_t tk = c->h_tot[j];
            const uint8_t* comp = (const uint8_t*)c->w_csr_comp.p + (size_t)bound[(size_t)j] * esz;
            if (!dense_out && (ret != GZ_OK || total + tk > O.capacity)) { ret = GZ_E_CAPACITY; total += tk; continue; }
            HIPCHK(c, hipStreamWaitEvent(c->s_out, c->ev_done[j], 0));
            if (out_direct) {
                if (tk) HIPCHK(c, hipMemcpyAsync((uint8_t*)O.tokens + (size_t)total * esz, comp, (size_t)tk * esz, hipMemcpyDeviceToHost, c->s_out));
                HIPCHK(c, hipMemcpyAsync(O.n_real + lo, (int32_t*)c->w_csr_nreal.p + lo, (size_t)nj * 4, hipMemcpyDeviceToHost, c->s_out));
                total += tk;
                continue;
            }
            const int slot = j % 3;
            pool.wait_tag(slot);                                 // (the jobs that read this slot two sub-batches ago)
            const size_t ebytes = ((size_t)tk * esz + 15) & ~(size_t)15;
            if ((rc = pinned_need(c, c->h_cout[slot], c->h_cout_cap[slot], ebytes + (size_t)nj * 4 + 16))) return rc;
            uint8_t* hs = c->h_cout[slot];
            if (tk) HIPCHK(c, hipMemcpyAsync(hs, comp, (size_t)tk * esz, hipMemcpyDeviceToHost, c->s_out));
            HIPCHK(c, hipMemcpyAsync(hs + ebytes, (int32_t*)c->w_csr_nreal.p + lo, (size_t)nj * 4, hipMemcpyDeviceToHost, c->s_out));
            HIPCHK(c, hipStreamSynchronize(c->s_out));
            const int32_t* nr = (const int32_t*)(hs + ebytes);
            if (!dense_out) {
                uint8_t* dst = (uint8_t*)O.tokens + (size_t)total * esz;
                const size_t nb = (size_t)tk * esz, parts = std::max<size_t>(1, std::min<size_t>((size_t)pool.threads(), nb >> 20));
                for (size_t q = 0; q < parts; ++q) {
                    const size_t a = nb * q / parts, b = nb * (q + 1) / parts;
                    pool.submit(slot, [=] { std::memcpy(dst + a, hs + a, b - a); });
                }
                int32_t* nd = O.n_real + lo;
                pool.submit(slot, [=] { std::memcpy(nd, nr, (size_t)nj * 4); });
            } else {
                uint32_t* rf = row_first[slot].data();
                uint64_t acc = 0;
                for (int64_t r = 0; r < nj; ++r) { rf[r] = (uint32_t)acc; acc += (uint32_t)(nr[r] < 0 ? 0 : nr[r] > max_len ? max_len : nr[r]); }
                if (acc != (uint64_t)tk) return fail(c, GZ_E_HIP, "internal: the row lengths of a sub-batch add up to %llu entries, its compact rows hold %lld", (unsigned long long)acc, (long long)tk);
                int32_t* ids = O.ids + lo * (int64_t)max_len;
                int32_t* mask = O.mask + lo * (int64_t)max_len;
                const int64_t per_job = std::max<int64_t>(1, ((int64_t)2 << 20) / ((int64_t)max_len * 4));       // ~ 2 MB of each array per job
                for (int64_t r0 = 0; r0 < nj; r0 += per_job) {
                    const int64_t r1 = std::min(nj, r0 + per_job);
                    const bool populate = (c->opt.host_hints & 2) != 0;
                    pool.submit(slot, [=] {
                        if (populate) {
                            hint_populate(ids + r0 * (int64_t)max_len, (size_t)(r1 - r0) * (size_t)max_len * 4);
                            hint_populate(mask + r0 * (int64_t)max_len, (size_t)(r1 - r0) * (size_t)max_len * 4);
                        }
                        if (esz == 2) expand_rows_host((const uint16_t*)hs, rf, nr, r0, r1, max_len, pad_id, ids, mask);
                        else expand_rows_host((const int32_t*)hs, rf, nr, r0, r1, max_len, pad_id, ids, mask);
                    });
                }
                if (O.n_real) { int32_t* nd = O.n_real + lo; pool.submit(slot, [=] { std::memcpy(nd, nr, (size_t)nj * 4); }); }
            }
            total += tk;
        }
    }
    pool.wait_all();
    HIPCHK(c, hipStreamSynchronize(c->s_out));
    HIPCHK(c, hipStreamSynchronize(s));
    *total_out = total;
    // a look-back time-out in any sub-batch (sub-batches of 32 MB are exactly the size that takes the chained scan)
    HIPCHK(c, hipMemcpy(c->h_flags, c->w_flags.p, 8, hipMemcpyDeviceToHost));
    if (c->h_flags[0]) return fail(c, GZ_E_HIP, "internal: a chained scan (gz_scan32m_kernel look-back) timed out");
    if (ret) return fail(c, ret, "the batch has %lld real entries, capacity is %lld", (long long)total, (long long)O.capacity);
    return GZ_OK;
}


static int encode_host_locked(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                    const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                    int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                    int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status)
{
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    if (n_docs < 0 || !text_off) return fail(c, GZ_E_INVALID, "bad text arguments");
    if ((pair == nullptr) != (pair_off == nullptr) && n_docs > 0)
        return fail(c, GZ_E_INVALID, "pair and pair_off must both be given or both be NULL");
    if (!input_ids || !attention_mask) return fail(c, GZ_E_INVALID, "input_ids / attention_mask are required");
    const bool is_pair = pair_off != nullptr;
    if (is_pair && (!token_type_ids || !sequence_id || !pair_len || !status))
        return fail(c, GZ_E_INVALID, "pair mode needs token_type_ids, sequence_id, pair_len and status");
    for (int64_t i = 0; i < n_docs; ++i) {
        if (text_off[i + 1] < text_off[i]) return fail(c, GZ_E_INVALID, "text_off is not non-decreasing at %lld", (long long)i);
        if (is_pair && pair_off[i + 1] < pair_off[i]) return fail(c, GZ_E_INVALID, "pair_off is not non-decreasing at %lld", (long long)i);
    }
    const GzShape S = make_shape(max_len, flags);
    const bool dense = is_dense(S);
    if (dense && capacity < n_docs * (int64_t)max_len) return fail(c, GZ_E_CAPACITY, "capacity < n_docs*max_len");
    if (!dense && !row_off) return fail(c, GZ_E_INVALID, "row_off is required for ragged layouts");
    if (n_docs == 0) { if (row_off) row_off[0] = 0; return GZ_OK; }

    hipStream_t s = c->stream;
    {
        // Small calls (a README-sized __call__, a few thousand sentences): ONE pinned staging block in each direction --
        // [text_off | text | pair_off | pair] up, [row_off | input_ids | attention_mask | token_type_ids | sequence_id |
        // n_real | pair_len | status] down -- and one synchronisation, instead of up to four copies up, nine blocking
        // copies down and two synchronisations.  Ragged outputs come down at their bound and are cut on the host.
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const int64_t tbs = text_off[n_docs] - text_off[0], pbs = is_pair ? pair_off[n_docs] - pair_off[0] : 0;
        if ((tbs > 0 && !text) || (pbs > 0 && !pair)) return fail(c, GZ_E_INVALID, "text is NULL but the offsets are not empty");
        const size_t off_b = (size_t)(n_docs + 1) * 8;
        const size_t in_text = up16(off_b), in_poff = up16(in_text + (size_t)tbs + 16), in_pair = up16(in_poff + (is_pair ? off_b : 0));
        const size_t in_b = up16(in_pair + (size_t)pbs + 16);
        int64_t E = dense ? n_docs * (int64_t)max_len : tbs + pbs + (is_pair ? 4 : 2) * n_docs;
        if (!dense && S.pad_mode && max_len > 0) E += n_docs * (int64_t)max_len;
        const size_t cells = up16((size_t)E * 4 + 16);
        const size_t o_row = in_b, o_ids = up16(o_row + off_b), o_mask = o_ids + cells, o_tt = o_mask + cells,
                     o_seq = o_tt + (is_pair ? cells : 0), o_nreal = o_seq + (is_pair ? cells : 0), o_plen = up16(o_nreal + (size_t)n_docs * 4),
                     o_status = up16(o_plen + (is_pair ? (size_t)n_docs * 8 : 0)), o_end = up16(o_status + (is_pair ? (size_t)n_docs * 4 : 0));
        if (o_end <= ((size_t)1 << 20)) {
            int rc0;
            if (c->pend.active && (rc0 = sync_locked(c))) return rc0;
            const size_t need = o_end + 64;
            if (need > c->h_stage_cap) {
                if (c->h_stage) hipHostFree(c->h_stage);
                c->h_stage = nullptr; c->h_stage_cap = 0;
                const size_t want = need < ((size_t)1 << 16) ? ((size_t)1 << 16) : need * 2;
                if (hipHostMalloc((void**)&c->h_stage, want, hipHostMallocDefault) != hipSuccess) return fail(c, GZ_E_NOMEM, "pinned staging");
                c->h_stage_cap = want;
            }
            if ((rc0 = ensure(c, c->w_stage, need))) return rc0;
            uint8_t* H = c->h_stage;
            // A call this small is latency, not bandwidth: the kernels work straight on the pinned block (page-locked host memory is
            // device-accessible; a README-sized call reads ~ 100 bytes over the bus and writes as few) -- two copies and their stream
            // round trips less per call.  Only when the one-launch kernel will take it (it touches its text once).
            // (not for ragged pairs: their row scan, finalize and pair kernels read the rows several times -- 60 -> 70 us over the bus)
            const bool direct = c->opt.host_direct > 0 && c->opt.small && o_end <= (size_t)c->opt.host_direct && !(flags & GZ_KEEP_WORDS) && !(is_pair && !dense);
            uint8_t* D = direct ? H : (uint8_t*)c->w_stage.p;
            std::memcpy(H, text_off, off_b);
            if (tbs) std::memcpy(H + in_text, text + text_off[0], (size_t)tbs);
            if (is_pair) {
                std::memcpy(H + in_poff, pair_off, off_b);
                if (pbs) std::memcpy(H + in_pair, pair + pair_off[0], (size_t)pbs);
            }
            if (!direct) HIPCHK(c, hipMemcpyAsync(D, H, in_b, hipMemcpyHostToDevice, s));
            rc0 = encode_device_locked(c, D + in_text - text_off[0], (const int64_t*)D, is_pair ? D + in_pair - pair_off[0] : nullptr,
                                       is_pair ? (const int64_t*)(D + in_poff) : nullptr, n_docs, max_len, flags, E, (int32_t*)(D + o_ids),
                                       (int32_t*)(D + o_mask), is_pair ? (int32_t*)(D + o_tt) : nullptr, is_pair ? (int32_t*)(D + o_seq) : nullptr,
                                       (int64_t*)(D + o_row), is_pair ? (int32_t*)(D + o_plen) : nullptr, (int32_t*)(D + o_nreal),
                                       is_pair ? (int32_t*)(D + o_status) : nullptr, text_off, pair_off);
            if (rc0) return rc0;
            if (!direct) HIPCHK(c, hipMemcpyAsync(H + o_row, D + o_row, o_end - o_row, hipMemcpyDeviceToHost, s));
            if ((rc0 = sync_locked(c))) return rc0;
            int64_t total = n_docs * (int64_t)max_len;
            if (!dense) {
                std::memcpy(row_off, H + o_row, off_b);
                total = row_off[n_docs];
                if (total > capacity) return fail(c, GZ_E_CAPACITY, "ragged output needs %lld entries, capacity is %lld", (long long)total, (long long)capacity);
            } else if (row_off) {
                for (int64_t i = 0; i <= n_docs; ++i) row_off[i] = i * (int64_t)max_len;
            }
            std::memcpy(input_ids, H + o_ids, (size_t)total * 4);
            std::memcpy(attention_mask, H + o_mask, (size_t)total * 4);
            if (is_pair) {
                std::memcpy(token_type_ids, H + o_tt, (size_t)total * 4);
                std::memcpy(sequence_id, H + o_seq, (size_t)total * 4);
                std::memcpy(pair_len, H + o_plen, (size_t)n_docs * 8);
                std::memcpy(status, H + o_status, (size_t)n_docs * 4);
            } else if (status) {
                std::memset(status, 0, (size_t)n_docs * 4);
            }
            if (n_real) std::memcpy(n_real, H + o_nreal, (size_t)n_docs * 4);
            return GZ_OK;
        }
    }
    const int64_t tb = text_off[n_docs] - text_off[0];
    const int64_t pb = is_pair ? pair_off[n_docs] - pair_off[0] : 0;
    int rc;
    // Large dense calls of single texts -- the batch form of Tokenize.__call__(text, max_len=L) -- bring only the rows' real
    // entries over the bus and pad them into the caller's arrays on host threads (csr_core); the switch dense_csr = 0, timed
    // calls, kept word records and the load-time whole-word build take the plain path below.
    if (dense && !is_pair && c->opt.dense_csr && !(flags & (GZ_TIMING | GZ_KEEP_WORDS)) && !c->building_words) {
        CsrOut O;
        O.ids = input_ids; O.mask = attention_mask; O.n_real = n_real;
        int64_t total = 0;
        if ((rc = csr_core(c, text, text_off, n_docs, max_len, flags, ids_fit_16(c) ? 16 : 32, O, &total))) return rc;
        if (row_off) for (int64_t i = 0; i <= n_docs; ++i) row_off[i] = i * (int64_t)max_len;
        if (status) std::memset(status, 0, (size_t)n_docs * 4);
        return GZ_OK;
    }
    if ((rc = ensure(c, c->w_text, (size_t)tb + 16))) return rc;
    if ((rc = ensure(c, c->w_toff, (size_t)(n_docs + 1) * 8))) return rc;
    // device-side outputs: dense -> n_docs*max_len; ragged -> worst case is known only after the encode kernel,
    // so size them by the bound  rows <= max(raw tokens, max_len)  with raw tokens <= bytes + 2 (+2)
    const int64_t raw_elems = tb + pb + (is_pair ? 4 : 2) * n_docs;
    int64_t out_elems = dense ? n_docs * (int64_t)max_len : raw_elems;
    if (!dense && S.pad_mode && max_len > 0) out_elems += n_docs * (int64_t)max_len;
    HostPool pool(pool_threads(c, (size_t)tb + (size_t)pb + (size_t)out_elems * (is_pair ? 16 : 8)));
    {
        const uint8_t* src = text + text_off[0];
        if (tb && (rc = copy_in(c, c->w_text.p, src, (size_t)tb, s))) return rc;
    }
    if ((rc = copy_in(c, c->w_toff.p, text_off, (size_t)(n_docs + 1) * 8, s))) return rc;
    const uint8_t* d_text = (const uint8_t*)c->w_text.p - text_off[0];
    const uint8_t* d_pair = nullptr;
    if (is_pair) {
        if ((rc = ensure(c, c->w_pair, (size_t)pb + 16))) return rc;
        if ((rc = ensure(c, c->w_poff, (size_t)(n_docs + 1) * 8))) return rc;
        if (pb && (rc = copy_in(c, c->w_pair.p, pair + pair_off[0], (size_t)pb, s))) return rc;
        if ((rc = copy_in(c, c->w_poff.p, pair_off, (size_t)(n_docs + 1) * 8, s))) return rc;
        d_pair = (const uint8_t*)c->w_pair.p - pair_off[0];
    }
    if ((rc = ensure(c, c->w_ids, (size_t)out_elems * 4 + 16))) return rc;
    if ((rc = ensure(c, c->w_mask, (size_t)out_elems * 4 + 16))) return rc;
    if ((rc = ensure(c, c->w_nreal, (size_t)(n_docs + 1) * 4))) return rc;
    if ((rc = ensure(c, c->w_rowoff, (size_t)(n_docs + 1) * 8))) return rc;
    if (is_pair) {
        if ((rc = ensure(c, c->w_tt, (size_t)out_elems * 4 + 16))) return rc;
        if ((rc = ensure(c, c->w_seq, (size_t)out_elems * 4 + 16))) return rc;
        if ((rc = ensure(c, c->w_pairlen, (size_t)(n_docs + 1) * 8))) return rc;
    }
    // status needs its own buffer in ragged mode (w_status holds the raw counts there)
    struct Scratch { DBuf b; ~Scratch() { release(b); } } st2;
    if (is_pair) { if ((rc = ensure(c, st2.b, (size_t)(n_docs + 1) * 4))) return rc; }
    rc = encode_device_locked(c, d_text, (const int64_t*)c->w_toff.p, d_pair, is_pair ? (const int64_t*)c->w_poff.p : nullptr,
                              n_docs, max_len, flags, out_elems, (int32_t*)c->w_ids.p, (int32_t*)c->w_mask.p,
                              is_pair ? (int32_t*)c->w_tt.p : nullptr, is_pair ? (int32_t*)c->w_seq.p : nullptr,
                              (int64_t*)c->w_rowoff.p, is_pair ? (int32_t*)c->w_pairlen.p : nullptr,
                              (int32_t*)c->w_nreal.p, is_pair ? (int32_t*)st2.b.p : nullptr, text_off, pair_off);
    if (rc == GZ_OK) rc = sync_locked(c);
    if (rc) return rc;

    int64_t total = n_docs * (int64_t)max_len;
    if (!dense) {
        if ((rc = copy_out(c, row_off, c->w_rowoff.p, (size_t)(n_docs + 1) * 8, s, &pool))) return rc;
        total = row_off[n_docs];
        if (total > capacity) return fail(c, GZ_E_CAPACITY, "ragged output needs %lld entries, capacity is %lld", (long long)total, (long long)capacity);
    } else if (row_off) {
        for (int64_t i = 0; i <= n_docs; ++i) row_off[i] = i * (int64_t)max_len;
    }
    if (total) {
        if ((size_t)total * 4 >= ((size_t)64 << 20) && (c->opt.host_hints & 1)) { hint_huge(input_ids, (size_t)total * 4); hint_huge(attention_mask, (size_t)total * 4); }
        if ((rc = copy_out(c, input_ids, c->w_ids.p, (size_t)total * 4, s, &pool))) return rc;
        if ((rc = copy_out(c, attention_mask, c->w_mask.p, (size_t)total * 4, s, &pool))) return rc;
        if (is_pair) {
            if ((rc = copy_out(c, token_type_ids, c->w_tt.p, (size_t)total * 4, s, &pool))) return rc;
            if ((rc = copy_out(c, sequence_id, c->w_seq.p, (size_t)total * 4, s, &pool))) return rc;
        }
    }
    if (is_pair) {
        if ((rc = copy_out(c, pair_len, c->w_pairlen.p, (size_t)n_docs * 8, s, &pool))) return rc;
        if ((rc = copy_out(c, status, st2.b.p, (size_t)n_docs * 4, s, &pool))) return rc;
    } else if (status) {
        std::memset(status, 0, (size_t)n_docs * 4);
    }
    if (n_real && (rc = copy_out(c, n_real, c->w_nreal.p, (size_t)n_docs * 4, s, &pool))) return rc;
    return GZ_OK;
}


int gz_encode_batch(gz_ctx* c, const uint8_t* text, const int64_t* text_off, const uint8_t* pair,
                    const int64_t* pair_off, int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                    int32_t* input_ids, int32_t* attention_mask, int32_t* token_type_ids, int32_t* sequence_id,
                    int64_t* row_off, int32_t* pair_len, int32_t* n_real, int32_t* status)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return encode_host_locked(c, text, text_off, pair, pair_off, n_docs, max_len, flags, capacity, input_ids,
                              attention_mask, token_type_ids, sequence_id, row_off, pair_len, n_real, status);
} GZ_CATCH(c)

int gz_host_alloc(gz_ctx* c, size_t bytes, void** ptr)
try {
    if (!c || !ptr) return GZ_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *ptr = nullptr; return fail(c, GZ_E_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    pinned_registry().add(*ptr, bytes ? bytes : 1);             // (the host paths copy to / from such blocks directly: gz_hostpath.h)
    return GZ_OK;
} GZ_CATCH(c)

int gz_host_free(gz_ctx* c, void* ptr)
try {
    // Independent of the context's state on purpose: a pinned block may outlive the context that allocated it (a numpy
    // array finalized after Tokenize.close(), or at interpreter exit), so nothing of *c is touched; c may be NULL.
    (void)c;
    if (ptr) pinned_registry().remove(ptr);
    if (ptr && hipHostFree(ptr) != hipSuccess) return GZ_E_HIP;
    return GZ_OK;
} GZ_CATCH(c)

// Host buffers in, CSR out, copies overlapped with the kernels (csr_core above).
int gz_encode_batch_csr(gz_ctx* c, const uint8_t* text, const int64_t* text_off, int64_t n_docs, int32_t max_len, uint32_t flags,
                        void* tokens, int64_t capacity, int32_t bits, int32_t* n_real, int64_t* total_out)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    if (n_docs < 0 || !text_off || !n_real || !total_out || capacity < 0 || (capacity > 0 && !tokens) || (bits != 16 && bits != 32))
        return fail(c, GZ_E_INVALID, "bad arguments");
    const GzShape S = make_shape(max_len, flags);
    if (!is_dense(S)) return fail(c, GZ_E_INVALID, "gz_encode_batch_csr needs max_len >= 1 with padding and truncation (rows are cut to max_len)");
    if (bits == 16 && !ids_fit_16(c)) return fail(c, GZ_E_LIMIT, "the vocabulary has ids that do not fit 16 bits");
    for (int64_t i = 0; i < n_docs; ++i)
        if (text_off[i + 1] < text_off[i]) return fail(c, GZ_E_INVALID, "text_off is not non-decreasing at %lld", (long long)i);
    *total_out = 0;
    if (n_docs == 0) return GZ_OK;
    if (text_off[n_docs] - text_off[0] > 0 && !text) return fail(c, GZ_E_INVALID, "text is NULL");
    CsrOut O;
    O.tokens = tokens; O.capacity = capacity; O.n_real = n_real;
    return csr_core(c, text, text_off, n_docs, max_len, flags, bits, O, total_out);
} GZ_CATCH(c)


int gz_word_token_counts(gz_ctx* c, int which_text, int32_t* counts, int64_t capacity, int64_t* doc_first, int64_t* n_words)
try {
    if (!c || !counts || !doc_first || !n_words || which_text < 0 || which_text > 1) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->pend.active) { int rc = sync_locked(c); if (rc) return rc; }
    const std::vector<GzAsmArgs>& subs = c->pend.subs;
    if (!c->pend.keep_words) return fail(c, GZ_E_INVALID, "the last encode call was not made with GZ_KEEP_WORDS");
    if (subs.empty() || which_text >= subs[0].n_texts) return fail(c, GZ_E_INVALID, "no encode call with that text to report on");
    if (subs.size() > 2) return fail(c, GZ_E_INVALID, "word counts are kept for batches of fewer than 65536 documents only");
    int64_t wbase = 0, dbase = 0;
    for (const GzAsmArgs& S : subs) {
        const GzTextBufs& X = S.X[which_text];
        int rcs;
        alloc_site(c);
        std::vector<uint32_t> dw((size_t)S.n_docs + 1);
        if ((rcs = copy_out(c, dw.data(), X.docw0, dw.size() * 4, c->stream))) return rcs;
        uint32_t total = 0;                                      // the scanned block counts end with the word total
        if ((rcs = copy_out_small(c, &total, X.blkcnt + X.nblk, 4, c->stream))) return rcs;
        for (int64_t d = 0; d < S.n_docs; ++d) doc_first[dbase + d] = wbase + dw[(size_t)d];
        if (wbase + (int64_t)total > capacity) {
            *n_words = wbase + total;
            return fail(c, GZ_E_CAPACITY, "the batch has more than %lld words", (long long)capacity);
        }
        if (total) {
            alloc_site(c);
            std::vector<uint32_t> wt(total);
            if ((rcs = copy_out(c, wt.data(), X.wtok, (size_t)total * 4, c->stream))) return rcs;
            for (uint32_t w = 0; w < total; ++w) counts[wbase + w] = !(wt[w] & 0x80000000u) ? 1 : (wt[w] & 0x20000000u) ? (int32_t)((wt[w] >> 25) & 15u) + 1 : (int32_t)(wt[w] & 0x0FFFFFFFu);   // (record forms: gz_pipeline.inc, W_NEAR)
        }
        wbase += total; dbase += S.n_docs;
    }
    doc_first[dbase] = wbase;
    *n_words = wbase;
    return GZ_OK;
} GZ_CATCH(c)

int64_t gz_bpe_word(gz_ctx* c, const uint8_t* word, int64_t len, int32_t* pieces, int64_t cap)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    if (len <= 0 || !word || !pieces || cap <= 0 || len > 0x3FFFFFFF) return fail(c, GZ_E_INVALID, "bad arguments");
    if (c->pend.active) { int rc = sync_locked(c); if (rc) return rc; }
    int rc;
    if ((rc = ensure(c, c->w_word, (size_t)len + 16))) return rc;
    if ((rc = ensure(c, c->w_wordout, (size_t)len * 4 + 16))) return rc;
    if ((rc = ensure(c, c->w_arena, (size_t)len * 4 + 16))) return rc;
    hipStream_t s = c->stream;
    if ((rc = copy_in(c, c->w_word.p, word, (size_t)len, s))) return rc;
    gz_launch_bpe_word((const GzDeviceTables*)c->t_struct.p, (const uint8_t*)c->w_word.p, len, (uint32_t*)c->w_arena.p, (int32_t*)c->w_wordout.p,
                       (int32_t)len, (int32_t*)c->w_flags.p + 2, s);
    HIPCHK(c, hipMemcpyAsync(c->h_flags + 2, (int32_t*)c->w_flags.p + 2, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    const int64_t n = c->h_flags[2];
    if (n > cap) return fail(c, GZ_E_CAPACITY, "word has %lld pieces, capacity %lld", (long long)n, (long long)cap);
    if ((rc = copy_out(c, pieces, c->w_wordout.p, (size_t)n * 4, s))) return rc;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t v = (uint32_t)pieces[i];
        if (v & GZ_SYM_UNKNOWN) pieces[i] = -(int32_t)(v & 0x1FFFFFu) - 1;
    }
    return n;
} GZ_CATCH(c)

int gz_device_alloc(gz_ctx* c, size_t bytes, void** dptr)
try {
    if (!c || !dptr) return GZ_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
#ifdef GZ_DIAG
    // (guard mode: the caller's buffers -- the text it uploads, the rows it reads back -- end at an unmapped granule too: what the
    //  device entry points may touch of a caller's buffer is exactly the bytes the caller said it has)
    if (c->opt.diag_guard) { *dptr = nullptr; return guard_alloc(c, dptr, bytes ? bytes : 16, c->opt.diag_guard); }
#endif
    hipError_t e = hipMalloc(dptr, bytes + 256);
    if (e != hipSuccess) { *dptr = nullptr; return fail(c, GZ_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return GZ_OK;
} GZ_CATCH(c)

int gz_device_free(gz_ctx* c, void* dptr)
try {
    if (!c) return GZ_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
#ifdef GZ_DIAG
    if (dptr && guard_free(dptr)) return GZ_OK;
#endif
    if (dptr) HIPCHK(c, hipFree(dptr));
    return GZ_OK;
} GZ_CATCH(c)

int gz_memcpy_h2d(gz_ctx* c, void* dst, const void* src, size_t bytes)
try {
    if (!c || (bytes && (!dst || !src))) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    // (through the library's pinned buffers, on the context's stream; complete on return: device entry points read their inputs on
    //  other streams as well)
    int rc = copy_in(c, dst, src, bytes, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GZ_OK;
} GZ_CATCH(c)

int gz_memcpy_d2h(gz_ctx* c, void* dst, const void* src, size_t bytes)
try {
    if (!c || (bytes && (!dst || !src))) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->x_used) HIPCHK(c, hipStreamSynchronize(c->xstream));       // (what an exchange step wrote is complete too)
    HostPool pool(pool_threads(c, bytes));
    return copy_out(c, dst, src, bytes, c->stream, &pool);
} GZ_CATCH(c)

int gz_timing_history(gz_ctx* c, double* out_ms, int32_t max, int32_t* n_out)
try {
    if (!c || !out_ms || !n_out || max < 0) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->pend.active) { int rc = sync_locked(c); if (rc) return rc; }
    int n = (int)(c->ring_n < (uint64_t)gz_ctx::RING ? c->ring_n : (uint64_t)gz_ctx::RING);
    if (n > max) n = max;
    for (int i = 0; i < n; ++i) {
        const uint64_t k = c->ring_n - (uint64_t)n + (uint64_t)i;
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ring[k % gz_ctx::RING][0], c->ring[k % gz_ctx::RING][1]));
        out_ms[i] = ms;
    }
    *n_out = n;
    c->ring_n = 0;
    return GZ_OK;
} GZ_CATCH(c)

int gz_timing(gz_ctx* c, double out_ms[4])
try {
    if (!c || !out_ms) return GZ_E_INVALID;
    for (int i = 0; i < 4; ++i) out_ms[i] = c->timing[i];
    return GZ_OK;
} GZ_CATCH(c)

// ---- batch decode ----------------------------------------------------------------------------------------------------
namespace {
constexpr size_t GZ_DEC_UNK_MAX = 4096;

uint32_t dec_flags(const std::string& w)
{
    uint32_t f = (uint32_t)w.size();
    if (w.size() >= 2 && w[w.size() - 1] == '@' && w[w.size() - 2] == '@') f |= GZ_DEC_ENDS_ATAT;
    if (w.find("@@ ") != std::string::npos) f |= GZ_DEC_INNER;
    return f;
}

// the table entry of a word whose bytes are (or would be) at `off` of the byte arena
GzDecEntry dec_entry_of(const std::string& w, size_t off)
{
    GzDecEntry e{dec_flags(w), {(uint32_t)off, 0u, 0u}};
    if (w.size() <= 12 && !(e.len_flags & GZ_DEC_INNER)) {
        e.len_flags |= GZ_DEC_INLINE;
        e.w[0] = 0;
        for (size_t k = 0; k < w.size(); ++k) e.w[k >> 2] |= (uint32_t)(uint8_t)w[k] << (8 * (k & 3));
    }
    return e;
}

int dec_set_unk(gz_ctx* c, const uint8_t* unk, int32_t unk_len)
{
    if (unk_len < 0 || (size_t)unk_len > GZ_DEC_UNK_MAX) return fail(c, GZ_E_LIMIT, "unk string longer than %zu bytes", GZ_DEC_UNK_MAX);
    const std::string u((const char*)unk, (size_t)unk_len);
    if (c->dec_unk_set && u == c->dec_unk) return GZ_OK;
    int rcs;
    if (unk_len && (rcs = copy_in(c, (uint8_t*)c->t_dec_bytes.p + c->dec_bytes_len, unk, (size_t)unk_len, c->stream))) return rcs;
    const GzDecEntry e = dec_entry_of(u, c->dec_bytes_len);
    if ((rcs = copy_in(c, (GzDecEntry*)c->t_dec_entries.p + c->dec_n_ids, &e, sizeof e, c->stream))) return rcs;
    c->dec_unk = u;
    c->dec_unk_set = true;
    return GZ_OK;
}

// both passes; *total = bytes of the whole batch.  out_dev may be nullptr (sizes only).
int decode_device_locked(gz_ctx* c, const int32_t* ids_dev, const int64_t* row_off_dev, int64_t n_rows, const uint8_t* unk,
                         int32_t unk_len, uint8_t* out_dev, int64_t capacity, int64_t* out_off_dev, int64_t* total)
{
    if (!c->have_dec) return fail(c, GZ_E_NOTABLES, "gz_decoder_snapshot has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dec_set_unk(c, unk, unk_len))) return rc;
    if ((rc = ensure(c, c->w_dec_rb, (size_t)(n_rows + 1) * 8))) return rc;
    GzDecTable D{(const GzDecEntry*)c->t_dec_entries.p, (const uint8_t*)c->t_dec_bytes.p, c->dec_n_ids};
    gz_launch_decode(D, ids_dev, row_off_dev, n_rows, (int64_t*)c->w_dec_rb.p, out_off_dev, nullptr, 0, c->stream);
    *total = 0;
    int64_t* const h_total = reinterpret_cast<int64_t*>(reinterpret_cast<uint8_t*>(c->h_pick) + 256);       // (pinned scratch)
    *h_total = 0;
    if (n_rows > 0) HIPCHK(c, hipMemcpyAsync(h_total, out_off_dev + n_rows, 8, hipMemcpyDeviceToHost, c->stream));
    else HIPCHK(c, hipMemsetAsync(out_off_dev, 0, 8, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *total = *h_total;
    if (!out_dev) return GZ_OK;
    if (*total > capacity) return fail(c, GZ_E_CAPACITY, "decode needs %lld bytes, capacity is %lld", (long long)*total, (long long)capacity);
    gz_launch_decode(D, ids_dev, row_off_dev, n_rows, nullptr, out_off_dev, out_dev, capacity, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return GZ_OK;
}
}  // namespace

int gz_decoder_snapshot(gz_ctx* c)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    // {v: k for k, v in encoder.items()} (tokenize.py:40): the last word wins on an id collision
    int32_t n_ids = 0;
    for (int32_t id : c->host.enc_ids) if (id + 1 > n_ids) n_ids = id + 1;
    std::vector<int64_t> last((size_t)n_ids, -1);
    for (size_t i = 0; i < c->host.enc_ids.size(); ++i) if (c->host.enc_ids[i] >= 0) last[(size_t)c->host.enc_ids[i]] = (int64_t)i;
    std::vector<GzDecEntry> ent((size_t)n_ids + 1, GzDecEntry{GZ_DEC_ABSENT, {0u, 0u, 0u}});
    std::string bytes;
    for (int32_t id = 0; id < n_ids; ++id) {
        if (last[(size_t)id] < 0) continue;
        const std::string& w = c->host.enc_words[(size_t)last[(size_t)id]];
        if (w.size() > GZ_DEC_LEN_MASK || bytes.size() + w.size() > 0xFFFF0000ull) return fail(c, GZ_E_LIMIT, "vocabulary too large for the decoder arena");
        ent[(size_t)id] = dec_entry_of(w, bytes.size());
        bytes += w;
    }
    ent[(size_t)n_ids] = GzDecEntry{GZ_DEC_INLINE, {0u, 0u, 0u}};                  // (the unk string: set per call, dec_set_unk)
    int rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure(c, c->t_dec_entries, ent.size() * sizeof(GzDecEntry)))) return rc;
    if ((rc = ensure(c, c->t_dec_bytes, bytes.size() + GZ_DEC_UNK_MAX + 16))) return rc;
    if ((rc = copy_in(c, c->t_dec_entries.p, ent.data(), ent.size() * sizeof(GzDecEntry), c->stream))) return rc;
    if (!bytes.empty() && (rc = copy_in(c, c->t_dec_bytes.p, bytes.data(), bytes.size(), c->stream))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->dec_n_ids = n_ids;
    c->dec_bytes_len = bytes.size();
    c->dec_unk_set = false;
    c->have_dec = true;
    return GZ_OK;
} GZ_CATCH(c)

int gz_decode_batch_device(gz_ctx* c, const int32_t* ids_dev, const int64_t* row_off_dev, int64_t n_rows, const uint8_t* unk,
                           int32_t unk_len, uint8_t* out_dev, int64_t capacity, int64_t* out_off_dev, int64_t* total_host)
try {
    if (!c || !row_off_dev || !out_off_dev || !total_host || n_rows < 0 || (!unk && unk_len) || capacity < 0)
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    return decode_device_locked(c, ids_dev, row_off_dev, n_rows, unk, unk_len, out_dev, capacity, out_off_dev, total_host);
} GZ_CATCH(c)

int gz_decode_batch(gz_ctx* c, const int32_t* ids, const int64_t* row_off, int64_t n_rows, const uint8_t* unk, int32_t unk_len,
                    uint8_t* out, int64_t capacity, int64_t* out_off)
try {
    if (!c || !row_off || !out_off || n_rows < 0 || (!unk && unk_len) || capacity < 0 || (capacity > 0 && !out))
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    const int64_t n_ids = row_off[n_rows] - row_off[0];
    if (n_ids < 0 || (n_ids > 0 && !ids)) return fail(c, GZ_E_INVALID, "bad row offsets");
    for (int64_t r = 0; r < n_rows; ++r) if (row_off[r + 1] < row_off[r]) return fail(c, GZ_E_INVALID, "row offsets must not decrease");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->w_dec_ids, (size_t)(n_ids + 1) * 4))) return rc;
    if ((rc = ensure(c, c->w_dec_roff, (size_t)(n_rows + 1) * 8))) return rc;
    if ((rc = ensure(c, c->w_dec_ooff, (size_t)(n_rows + 1) * 8))) return rc;
    if (n_ids && (rc = copy_in(c, c->w_dec_ids.p, ids + row_off[0], (size_t)n_ids * 4, c->stream))) return rc;
    if ((rc = copy_in(c, c->w_dec_roff.p, row_off, (size_t)(n_rows + 1) * 8, c->stream))) return rc;
    int64_t total = 0;
    if ((rc = decode_device_locked(c, (const int32_t*)c->w_dec_ids.p - row_off[0], (const int64_t*)c->w_dec_roff.p, n_rows, unk, unk_len, nullptr, 0,
                                   (int64_t*)c->w_dec_ooff.p, &total))) return rc;
    if ((rc = copy_out(c, out_off, c->w_dec_ooff.p, (size_t)(n_rows + 1) * 8, c->stream))) return rc;
    if (total > capacity) return fail(c, GZ_E_CAPACITY, "decode needs %lld bytes, capacity is %lld", (long long)total, (long long)capacity);
    if (total == 0) return GZ_OK;
    if ((rc = ensure(c, c->w_dec_out, (size_t)total))) return rc;
    GzDecTable D{(const GzDecEntry*)c->t_dec_entries.p, (const uint8_t*)c->t_dec_bytes.p, c->dec_n_ids};
    gz_launch_decode(D, (const int32_t*)c->w_dec_ids.p - row_off[0], (const int64_t*)c->w_dec_roff.p, n_rows, nullptr, (int64_t*)c->w_dec_ooff.p,
                     (uint8_t*)c->w_dec_out.p, total, c->stream);
    {
        HostPool pool(pool_threads(c, (size_t)total));
        if ((rc = copy_out(c, out, c->w_dec_out.p, (size_t)total, c->stream, &pool))) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return GZ_OK;
} GZ_CATCH(c)

// ---- text pre-pass ----------------------------------------------------------------------------------------------------
namespace {
// Runs the filters one after the other.  Every document keeps its slot of the packed input and only shrinks inside
// it (ping-pong between two slot buffers, one classification pass per filter); the final lengths are scanned and the
// documents packed into out_dev.  in_bytes = bytes of the input.
int preprocess_device_locked(gz_ctx* c, const int32_t* ops, int32_t n_ops, const uint8_t* text_dev, const int64_t* off_dev,
                             int64_t n_docs, int64_t in_bytes, uint8_t* out_dev, int64_t capacity, int64_t* out_off_dev, int64_t* total)
{
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    for (int k = 0; k < n_ops; ++k) if (ops[k] < GZ_PP_HTML || ops[k] > GZ_PP_URL) return fail(c, GZ_E_INVALID, "unknown filter %d", ops[k]);
    if ((rc = ensure(c, c->w_ppaux, (size_t)(n_docs + 1) * 8))) return rc;
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure(c, c->w_pp[k], (size_t)in_bytes + 16))) return rc;
        if ((rc = ensure(c, c->w_ppoff[k], (size_t)(n_docs + 1) * 8))) return rc;          // lengths after filter k, k+2, ...
    }
    const uint8_t* in = text_dev;
    const int64_t* in_len = nullptr;
    // documents of at most 4 KiB: the whole chain in one kernel, on chip (option pp_fused = 0: filter by filter like the long ones)
    // (the fused kernel's tail -- lengths, their scan, the pack kernel's offsets, the total -- is 32-bit: filters never grow a
    // document, so it is exact while the INPUT stays under 4 GiB; a larger input takes the filter-by-filter chain with its 64-bit
    // scan and pack kernels)
    const bool fused = c->opt.pp_fused != 0 && gz_pp_tail_is_32bit(in_bytes);
    if (fused && n_docs > 0) {
        GzPpFusedArgs F{};
        F.in = text_dev; F.in_off = off_dev; F.n_docs = n_docs;
        F.out = (uint8_t*)c->w_pp[(n_ops - 1) & 1].p; F.out_len = (int64_t*)c->w_ppoff[(n_ops - 1) & 1].p;
        F.n_ops = n_ops;
        for (int k = 0; k < n_ops; ++k) F.ops[k] = ops[k];
        // control words: [0] documents the kernel leaves to the chain below (none, as a rule), [1] ticket counter and [2] time-out
        // flag of the chained scan; the 32-bit lengths the scan turns into offsets in place
        if ((rc = ensure(c, c->w_ppctl, 64)) || (rc = ensure(c, c->w_pplen32, (size_t)(n_docs + 4) * 4)) ||
            (rc = ensure(c, c->w_pplb, (size_t)(n_docs / 1024 + 4) * 8, /* zero a new allocation */ true))) return rc;
        uint32_t* ctl = (uint32_t*)c->w_ppctl.p;
        F.n_long = ctl; F.out_len32 = (uint32_t*)c->w_pplen32.p;
        HIPCHK(c, hipMemsetAsync(ctl, 0, 16, c->stream));
        gz_launch_preprocess_fused(F, c->stream);
        // ... and, expecting that no document was too long, straight on: offsets by the chained scan, the slots packed into
        // out_dev, ONE synchronisation for the whole pre-pass.  (Round 3: a one-workgroup scan of the 64-bit lengths, 0.69 ms for
        // 1 M documents, a pack kernel with one document per wave, 0.32 ms, and three synchronisations.)
        c->lb_epoch += 4;
        if ((c->lb_epoch & 0x3FFFFFFFu) < 4u) {                  // the 30-bit call number wrapped (see setup_text)
            HIPCHK(c, hipDeviceSynchronize());
            for (auto& slot : c->tw) for (auto& t : slot) if (t.lookback.p) HIPCHK(c, hipMemset(t.lookback.p, 0, t.lookback.cap));
            HIPCHK(c, hipMemset(c->w_pplb.p, 0, c->w_pplb.cap));
        }
        gz_launch_pp_tail(F.out, off_dev, F.out_len32, n_docs, out_dev, capacity, out_off_dev, (unsigned long long*)c->w_pplb.p, ctl, c->lb_epoch + 1, c->stream);
        uint32_t* const h = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(c->h_pick) + 272);       // (pinned scratch: 4 control words + the total)
        for (int q = 0; q < 5; ++q) h[q] = 0;
        HIPCHK(c, hipMemcpyAsync(h, ctl, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h + 4, F.out_len32 + n_docs, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t total32 = h[4];
        if (h[2]) return fail(c, GZ_E_HIP, "internal: the chained scan of the pre-pass timed out");
        if (h[0] == 0) {
            *total = (int64_t)total32;
            if (out_dev && *total > capacity) return fail(c, GZ_E_CAPACITY, "pre-pass output needs %lld bytes, capacity is %lld", (long long)*total, (long long)capacity);
            HIPCHK(c, hipGetLastError());
            return GZ_OK;
        }
        // some documents are longer than the fused kernel takes: the filter-by-filter chain does those (and only those), then the
        // general tail below
    }
    for (int k = 0; k < n_ops; ++k) {
        GzPpArgs A{};
        A.in = in; A.in_off = off_dev; A.in_len = in_len; A.n_docs = n_docs; A.op = ops[k];
        A.skip_upto = fused ? GZ_PP_FUSED_MAX_BYTES : -1;
        A.in_abs = k == 0 ? 1 : 0;                                                         // the caller's text; later filters read their own slot buffers
        A.out = (uint8_t*)c->w_pp[k & 1].p; A.out_len = (int64_t*)c->w_ppoff[k & 1].p; A.aux = (int64_t*)c->w_ppaux.p;
        if (ops[k] == GZ_PP_HTML) gz_launch_preprocess(A, 0, c->stream);                  // does the last '<' close?
        gz_launch_preprocess(A, 1, c->stream);
        in = A.out; in_len = A.out_len;
    }
    gz_launch_scan64(in_len, n_docs, out_off_dev, c->stream);
    *total = 0;
    int64_t* const h_total = reinterpret_cast<int64_t*>(reinterpret_cast<uint8_t*>(c->h_pick) + 320);       // (pinned scratch)
    *h_total = 0;
    if (n_docs > 0) HIPCHK(c, hipMemcpyAsync(h_total, out_off_dev + n_docs, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *total = *h_total;
    if (!out_dev) return GZ_OK;
    if (*total > capacity) return fail(c, GZ_E_CAPACITY, "pre-pass output needs %lld bytes, capacity is %lld", (long long)*total, (long long)capacity);
    gz_launch_pp_pack(in, off_dev, in_len, n_docs, out_dev, out_off_dev, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return GZ_OK;
}
}  // namespace

int gz_preprocess_batch_device(gz_ctx* c, const int32_t* ops, int32_t n_ops, const uint8_t* text_dev, const int64_t* text_off_dev,
                               int64_t n_docs, int64_t text_bytes, uint8_t* out_dev, int64_t capacity, int64_t* out_off_dev,
                               int64_t* total_host)
try {
    if (!c || !ops || n_ops < 1 || n_ops > 16 || !text_off_dev || !out_off_dev || !total_host || n_docs < 0 || text_bytes < 0 || capacity < 0)
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    return preprocess_device_locked(c, ops, n_ops, text_dev, text_off_dev, n_docs, text_bytes, out_dev, capacity, out_off_dev, total_host);
} GZ_CATCH(c)

int gz_preprocess_batch(gz_ctx* c, const int32_t* ops, int32_t n_ops, const uint8_t* text, const int64_t* text_off, int64_t n_docs,
                        uint8_t* out, int64_t capacity, int64_t* out_off)
try {
    if (!c || !ops || n_ops < 1 || n_ops > 16 || !text_off || !out_off || n_docs < 0 || capacity < 0 || (capacity > 0 && !out))
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    const int64_t nbytes = text_off[n_docs] - text_off[0];
    if (nbytes < 0 || (nbytes > 0 && !text)) return fail(c, GZ_E_INVALID, "bad text offsets");
    for (int64_t d = 0; d < n_docs; ++d) if (text_off[d + 1] < text_off[d]) return fail(c, GZ_E_INVALID, "text offsets must not decrease");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->w_pp_in, (size_t)nbytes + 16))) return rc;
    if ((rc = ensure(c, c->w_pp_inoff, (size_t)(n_docs + 1) * 8))) return rc;
    DBuf& fin = c->w_dec_out;                                       // final text (shared scratch with decode)
    DBuf& fino = c->w_dec_ooff;
    if ((rc = ensure(c, fin, (size_t)nbytes + 16))) return rc;
    if ((rc = ensure(c, fino, (size_t)(n_docs + 1) * 8))) return rc;
    if (nbytes && (rc = copy_in(c, c->w_pp_in.p, text + text_off[0], (size_t)nbytes, c->stream))) return rc;
    if ((rc = copy_in(c, c->w_pp_inoff.p, text_off, (size_t)(n_docs + 1) * 8, c->stream))) return rc;
    int64_t total = 0;
    if ((rc = preprocess_device_locked(c, ops, n_ops, (const uint8_t*)c->w_pp_in.p - text_off[0], (const int64_t*)c->w_pp_inoff.p, n_docs, nbytes,
                                       (uint8_t*)fin.p, nbytes, (int64_t*)fino.p, &total))) return rc;
    HostPool pool(pool_threads(c, (size_t)total));
    if ((rc = copy_out(c, out_off, fino.p, (size_t)(n_docs + 1) * 8, c->stream, &pool))) return rc;
    if (total > capacity) return fail(c, GZ_E_CAPACITY, "pre-pass output needs %lld bytes, capacity is %lld", (long long)total, (long long)capacity);
    if (total && (rc = copy_out(c, out, fin.p, (size_t)total, c->stream, &pool))) return rc;
    return GZ_OK;
} GZ_CATCH(c)

// ---- DLPack hand-off ---------------------------------------------------------------------------------------------------
// The deleter a consumer (torch, ...) calls may run while the Python interpreter shuts down, so it must be plain C:
// a refcounted block owns the HBM allocation, every exported DLManagedTensor holds one reference.
struct gz_block {
    std::atomic<int> refs{1};
    void* dptr = nullptr;
    int device = 0;
};
namespace {
struct DlDevice { int32_t device_type, device_id; };
struct DlDataType { uint8_t code, bits; uint16_t lanes; };
struct DlTensor { void* data; DlDevice device; int32_t ndim; DlDataType dtype; int64_t* shape; int64_t* strides; uint64_t byte_offset; };
struct DlManagedTensor { DlTensor dl_tensor; void* manager_ctx; void (*deleter)(DlManagedTensor*); };   // dlpack.h, legacy ABI

void block_release(gz_block* b)
{
    if (b && b->refs.fetch_sub(1) == 1) {
        if (b->dptr) dev_free(b->dptr);
        delete b;
    }
}
void dl_deleter(DlManagedTensor* mt)
{
    if (!mt) return;
    block_release((gz_block*)mt->manager_ctx);
    free(mt->dl_tensor.shape);
    free(mt);
}
}  // namespace

int gz_block_create(gz_ctx* c, void* dptr, gz_block** out)
try {
    if (!c || !dptr || !out) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    gz_block* b = new (std::nothrow) gz_block();
    if (!b) return fail(c, GZ_E_NOMEM, "out of memory");
    b->dptr = dptr; b->device = c->device;
    *out = b;
    return GZ_OK;
} GZ_CATCH(c)

void gz_block_release(gz_block* b) { block_release(b); }

void* gz_block_dlpack(gz_block* b, int32_t ndim, const int64_t* shape, int32_t dtype_code, int32_t dtype_bits)
try {
    if (!b || ndim < 0 || ndim > 8 || (ndim && !shape)) return nullptr;
    DlManagedTensor* mt = (DlManagedTensor*)calloc(1, sizeof(DlManagedTensor));
    int64_t* sh = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ndim ? ndim : 1));
    if (!mt || !sh) { free(mt); free(sh); return nullptr; }
    for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
    mt->dl_tensor.data = b->dptr;
    mt->dl_tensor.device = DlDevice{10 /* kDLROCM */, b->device};
    mt->dl_tensor.ndim = ndim;
    mt->dl_tensor.dtype = DlDataType{(uint8_t)dtype_code, (uint8_t)dtype_bits, 1};
    mt->dl_tensor.shape = sh;
    mt->dl_tensor.strides = nullptr;
    mt->dl_tensor.byte_offset = 0;
    b->refs.fetch_add(1);
    mt->manager_ctx = b;
    mt->deleter = dl_deleter;
    return mt;
} GZ_CATCH_NULL

// PyCapsule destructor for a capsule made from gz_block_dlpack's result: a capsule that was never consumed (its name is
// still "dltensor"; a consumer renames it to "used_dltensor" and takes over the deleter) releases its reference to
// the block.  Plain C, resolved against the running interpreter: this library does not link libpython.
void gz_dlpack_capsule_destructor(void* capsule)
try {
    typedef int (*isvalid_t)(void*, const char*);
    typedef void* (*getptr_t)(void*, const char*);
    static const isvalid_t isvalid = (isvalid_t)dlsym(RTLD_DEFAULT, "PyCapsule_IsValid");
    static const getptr_t getptr = (getptr_t)dlsym(RTLD_DEFAULT, "PyCapsule_GetPointer");
    if (!capsule || !isvalid || !getptr || !isvalid(capsule, "dltensor")) return;
    DlManagedTensor* mt = (DlManagedTensor*)getptr(capsule, "dltensor");
    if (mt && mt->deleter) mt->deleter(mt);
} GZ_CATCH_VOID

// ---- compact rows for the exchange step -----------------------------------------------------------------------------
namespace {
int compact_impl(gz_ctx* c, const int32_t* rows_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len, void* out_dev,
                 int bits, int64_t* total_host, uint32_t* first_dev = nullptr)
{
    if (!c || !rows_dev || !n_real_dev || !out_dev || !total_host || n_rows < 0 || row_len <= 0)
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (bits == 16 && (!c->have_tables || !ids_fit_16(c))) return fail(c, GZ_E_LIMIT, "the vocabulary has ids that do not fit 16 bits");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->w_rowoff32, (size_t)(n_rows + 2) * 4))) return rc;
    uint32_t* off = (uint32_t*)c->w_rowoff32.p;
    if ((rc = x_begin(c))) return rc;
    gz_launch_row_offsets(n_real_dev, n_rows, off, c->xstream);
    gz_launch_compact(rows_dev, off, n_rows, row_len, out_dev, bits, first_dev, c->xstream);
    uint32_t* total = reinterpret_cast<uint32_t*>(c->h_pick) + 60;           // pinned (h_pick is 256 bytes; its tail is free)
    HIPCHK(c, hipMemcpyAsync(total, off + n_rows, 4, hipMemcpyDeviceToHost, c->xstream));
    if ((rc = x_end(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->xstream));
    HIPCHK(c, hipGetLastError());
    *total_host = *total;
    return GZ_OK;
}
int expand_impl(gz_ctx* c, const void* compact_dev, int bits, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len,
                int32_t* ids_dev, int32_t* mask_dev, const uint32_t* first_dev = nullptr /* a block's own array: no scan */,
                int64_t total = 0xFFFFFFFFll /* entries the compact array holds (a block's announced total) */)
{
    if (!c || !compact_dev || !n_real_dev || !ids_dev || !mask_dev || n_rows < 0 || row_len <= 0 || total < 0 || total > 0xFFFFFFFFll)
        return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_tables) return fail(c, GZ_E_NOTABLES, "gz_load_tables has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const uint32_t* off = first_dev;
    if (!off) {
        if ((rc = ensure(c, c->w_rowoff32, (size_t)(n_rows + 2) * 4))) return rc;
        off = (const uint32_t*)c->w_rowoff32.p;
    }
    if ((rc = x_begin(c))) return rc;
    if (!first_dev) gz_launch_row_offsets(n_real_dev, n_rows, (uint32_t*)c->w_rowoff32.p, c->xstream);
    // (a row that does not fit its block raises w_flags[12]; the word travels to the pinned scratch behind every expansion and is
    //  reported -- and cleared -- by the next synchronisation: sync_locked)
    int32_t* bad = (int32_t*)c->w_flags.p + 12;
    gz_launch_expand(compact_dev, bits, off, n_real_dev, n_rows, row_len, c->dev.pad_id, ids_dev, mask_dev, (uint32_t)total, bad, c->xstream);
    HIPCHK(c, hipMemcpyAsync(reinterpret_cast<uint8_t*>(c->h_pick) + 384, bad, 4, hipMemcpyDeviceToHost, c->xstream));
    if ((rc = x_end(c))) return rc;
    HIPCHK(c, hipGetLastError());
    return GZ_OK;
}
}  // namespace

int gz_compact_rows(gz_ctx* c, const int32_t* rows_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len,
                    int32_t* out_dev, int64_t* total_host)
try {
    return compact_impl(c, rows_dev, n_real_dev, n_rows, row_len, out_dev, 32, total_host);
} GZ_CATCH(c)
int gz_compact_rows16(gz_ctx* c, const int32_t* rows_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len,
                      uint16_t* out_dev, int64_t* total_host)
try {
    return compact_impl(c, rows_dev, n_real_dev, n_rows, row_len, out_dev, 16, total_host);
} GZ_CATCH(c)
int gz_expand_rows(gz_ctx* c, const int32_t* compact_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len,
                   int32_t* ids_dev, int32_t* mask_dev)
try {
    return expand_impl(c, compact_dev, 32, n_real_dev, n_rows, row_len, ids_dev, mask_dev);
} GZ_CATCH(c)
int gz_expand_rows16(gz_ctx* c, const uint16_t* compact_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len,
                     int32_t* ids_dev, int32_t* mask_dev)
try {
    return expand_impl(c, compact_dev, 16, n_real_dev, n_rows, row_len, ids_dev, mask_dev);
} GZ_CATCH(c)

int gz_encode_emit_block(gz_ctx* c, int32_t* block_dev, int32_t bits)
try {
    if (!c || (bits != 16 && bits != 32)) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (block_dev && bits == 16 && (!c->have_tables || !ids_fit_16(c))) return fail(c, GZ_E_LIMIT, "the vocabulary has ids that do not fit 16 bits");
    c->emit_block = block_dev; c->emit_bits = bits;              // (null: disarm)
    return GZ_OK;
} GZ_CATCH(c)
int gz_block_total(gz_ctx* c, int32_t back, int64_t* total_host)
try {
    if (!c || !total_host || back < 0 || back > 2) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->enc_seq <= (uint64_t)back) return fail(c, GZ_E_INVALID, "no encode call %d calls back", back);
    const int slot = (int)((c->enc_seq - 1 - (uint64_t)back) & 3);
    if (!c->blk_valid[slot]) return fail(c, GZ_E_INVALID, "the encode call %d calls back emitted no block (gz_encode_emit_block before it)", back);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_blk[slot]));
    *total_host = reinterpret_cast<uint32_t*>(c->h_pick)[56 + slot];
    return GZ_OK;
} GZ_CATCH(c)

int gz_compact_block(gz_ctx* c, const int32_t* rows_dev, const int32_t* n_real_dev, int64_t n_rows, int32_t row_len, int32_t bits,
                     int32_t* block_dev, int64_t* total_host)
try {
    if (!c || !block_dev || !n_real_dev || n_rows < 0 || (bits != 16 && bits != 32)) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    {
        // the row lengths in front of the entries (on the exchange stream, behind the encode call they belong to)
        std::lock_guard<std::mutex> lk(c->mu);
        HIPCHK(c, hipSetDevice(c->device));
        int rc;
        if ((rc = x_begin(c))) return rc;
        if (n_rows) HIPCHK(c, hipMemcpyAsync(block_dev, n_real_dev, (size_t)n_rows * 4, hipMemcpyDeviceToDevice, c->xstream));
        if ((rc = x_end(c))) return rc;
    }
    return compact_impl(c, rows_dev, n_real_dev, n_rows, row_len, block_dev + 2 * n_rows, bits, total_host, (uint32_t*)(block_dev + n_rows));
} GZ_CATCH(c)
int gz_expand_block(gz_ctx* c, const int32_t* block_dev, int32_t bits, int64_t n_rows, int32_t row_len, int64_t total_entries, int32_t* ids_dev, int32_t* mask_dev)
try {
    if (!c || !block_dev || n_rows < 0 || (bits != 16 && bits != 32)) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    return expand_impl(c, block_dev + 2 * n_rows, bits, block_dev, n_rows, row_len, ids_dev, mask_dev, (const uint32_t*)(block_dev + n_rows), total_entries);
} GZ_CATCH(c)

// ---- multi-GPU exchange step --------------------------------------------------------------------------------------
int gz_comm_unique_id(uint8_t id_out[128])
try {
    if (!id_out) return GZ_E_INVALID;
    if (!rccl_load()) return fail(nullptr, GZ_E_RCCL, "librccl.so could not be loaded");
    int r = g_rccl.GetUniqueId(id_out);
    return r == 0 ? GZ_OK : fail(nullptr, GZ_E_RCCL, "ncclGetUniqueId failed (%d)", r);
} GZ_CATCH(nullptr)

int gz_comm_init(gz_ctx* c, const uint8_t id[128], int rank, int world)
try {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return c ? fail(c, GZ_E_INVALID, "bad arguments") : GZ_E_INVALID;
    if (!rccl_load()) return fail(c, GZ_E_RCCL, "librccl.so could not be loaded");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    Id128 uid;
    std::memcpy(uid.b, id, 128);
    int r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) { c->comm = nullptr; return fail(c, GZ_E_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"); }
    c->rank = rank; c->world = world;
    return GZ_OK;
} GZ_CATCH(c)

int gz_exchange_select(gz_ctx* c, int back)
try {
    if (!c || back < 0 || back > 2) return c ? fail(c, GZ_E_INVALID, "back must be 0, 1 or 2") : GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    c->x_back = back;
    return GZ_OK;
} GZ_CATCH(c)

int gz_gather_rows(gz_ctx* c, const int32_t* send_dev, int64_t n_rows_local, int32_t row_len, int32_t* recv_dev,
                   const int64_t* rows_per_rank, int root)
try {
    if (!c) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->comm) return fail(c, GZ_E_RCCL, "gz_comm_init has not been called");
    if (root < 0 || root >= c->world || row_len <= 0 || n_rows_local < 0 || !rows_per_rank)
        return fail(c, GZ_E_INVALID, "bad arguments");
    // Everything that can be refused is refused BEFORE the group is opened: a rank that leaves between ncclGroupStart and
    // ncclGroupEnd, or whose count differs from what the root expects, hangs every other rank.
    if (rows_per_rank[c->rank] != n_rows_local)
        return fail(c, GZ_E_INVALID, "rows_per_rank[%d] = %lld but n_rows_local = %lld", c->rank,
                    (long long)rows_per_rank[c->rank], (long long)n_rows_local);
    int64_t total_rows = 0;
    for (int q = 0; q < c->world; ++q) {
        if (rows_per_rank[q] < 0) return fail(c, GZ_E_INVALID, "rows_per_rank[%d] is negative", q);
        total_rows += rows_per_rank[q];
    }
    if (n_rows_local > 0 && !send_dev) return fail(c, GZ_E_INVALID, "send_dev is NULL");
    if (c->rank == root && total_rows > 0 && !recv_dev) return fail(c, GZ_E_INVALID, "recv_dev is NULL on the root");
    HIPCHK(c, hipSetDevice(c->device));
    const int ncclInt32 = 2;
    { int rc0 = x_begin(c); if (rc0) return rc0; }
    hipEvent_t* xe = c->xring[c->xring_n % gz_ctx::XRING];
    if (!xe[0]) { HIPCHK(c, hipEventCreate(&xe[0])); HIPCHK(c, hipEventCreate(&xe[1])); }
    HIPCHK(c, hipEventRecord(xe[0], c->xstream));                // (behind whatever the exchange stream waits for: the gather's own start)
    // the root's own block is a device-to-device copy (outside the group: it is not an RCCL operation)
    if (c->rank == root && n_rows_local > 0) {
        int64_t row0 = 0;
        for (int q = 0; q < root; ++q) row0 += rows_per_rank[q];
        int32_t* dst = recv_dev + row0 * (int64_t)row_len;
        if (dst != send_dev)
            HIPCHK(c, hipMemcpyAsync(dst, send_dev, (size_t)n_rows_local * (size_t)row_len * 4, hipMemcpyDeviceToDevice, c->xstream));
    }
    // direct gatherv: the root posts one receive per peer, each peer one send, all inside one group, so every
    // peer's block travels over its own xGMI link concurrently (no ring).  The group is always closed.
    int r = g_rccl.GroupStart();
    if (r == 0) {
        if (c->rank == root) {
            int64_t row0 = 0;
            for (int q = 0; q < c->world; ++q) {
                const size_t cnt = (size_t)rows_per_rank[q] * (size_t)row_len;
                if (q != root && cnt && r == 0)
                    r = g_rccl.Recv(recv_dev + row0 * (int64_t)row_len, cnt, ncclInt32, q, c->comm, c->xstream);
                row0 += rows_per_rank[q];
            }
        } else {
            const size_t cnt = (size_t)n_rows_local * (size_t)row_len;
            if (cnt) r = g_rccl.Send(send_dev, cnt, ncclInt32, root, c->comm, c->xstream);
        }
        const int r2 = g_rccl.GroupEnd();
        if (r == 0) r = r2;
    }
    if (r != 0)
        return fail(c, GZ_E_RCCL, "RCCL gather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
    HIPCHK(c, hipEventRecord(xe[1], c->xstream));
    c->xring_n++;
    { int rc1 = x_end(c); if (rc1) return rc1; }
    return GZ_OK;
} GZ_CATCH(c)

int gz_exchange_timing_history(gz_ctx* c, double* out_ms, int32_t max, int32_t* n_out)
try {
    if (!c || !out_ms || !n_out || max < 0) return GZ_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->xstream) HIPCHK(c, hipStreamSynchronize(c->xstream));
    int n = (int)(c->xring_n < (uint64_t)gz_ctx::XRING ? c->xring_n : (uint64_t)gz_ctx::XRING);
    if (n > max) n = max;
    for (int i = 0; i < n; ++i) {
        const uint64_t k = c->xring_n - (uint64_t)n + (uint64_t)i;
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->xring[k % gz_ctx::XRING][0], c->xring[k % gz_ctx::XRING][1]));
        out_ms[i] = ms;
    }
    *n_out = n;
    c->xring_n = 0;
    return GZ_OK;
} GZ_CATCH(c)

}  // extern "C"
