// Host side of the host entry points (included by gz_api.cpp, after gz_ctx / fail / HIPCHK / ensure): how bytes get from the
// caller's memory to the device and back.
//
// RULE: memory the CALLER owns is never handed to a HIP copy.  A caller's buffer is, as a rule, pageable (a numpy array, a
// std::vector): the runtime would pin it on the fly, or stage it through buffers of its own, inside the copy call -- slow
// (2 GB of dense rows came back at 15 GB/s), serialised with everything else, and a code path of the runtime that only host
// calls on fresh allocations ever take.  Here every copy between host and device has a PINNED buffer of this library on the host
// side; the caller's memory is read and written by plain host code (memcpy, or the row expansion below) on worker threads,
// while the next piece is on the bus.  Memory that came from gz_host_alloc is pinned already and is copied directly.
#include "gz_hostpool.h"

namespace {

constexpr size_t XFER_CHUNK = (size_t)4 << 20;       // pinned transfer buffers: two of these (grown on demand up to this size)

int pool_threads(gz_ctx* c, size_t bytes_to_move)
{
    int t = c->opt.host_threads > 0 ? c->opt.host_threads : host_cpus();
    const size_t by_size = bytes_to_move / ((size_t)8 << 20);         // a thread per 8 MB: small calls stay on the caller's thread
    if ((size_t)t > by_size) t = (int)by_size;
    return t <= 1 ? 0 : t;
}

int pinned_need(gz_ctx* c, uint8_t*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap && p) return GZ_OK;
    if (p) { hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return fail(c, GZ_E_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); }
    cap = want;
    return GZ_OK;
}

int xfer_need(gz_ctx* c, size_t bytes)
{
    const size_t want = bytes < XFER_CHUNK ? (bytes < 65536 ? 65536 : bytes) : XFER_CHUNK;
    for (int b = 0; b < 2; ++b) {
        if (c->xfer_busy[b]) { HIPCHK(c, hipEventSynchronize(c->ev_xfer[b])); c->xfer_busy[b] = false; }
        if (c->h_xfer_cap[b] < want) {
            if (c->h_xfer[b]) { hipHostFree(c->h_xfer[b]); c->h_xfer[b] = nullptr; c->h_xfer_cap[b] = 0; }
            const size_t cap = want >= XFER_CHUNK / 2 ? XFER_CHUNK : want * 2;
            if (hipHostMalloc((void**)&c->h_xfer[b], cap, hipHostMallocDefault) != hipSuccess) { c->h_xfer[b] = nullptr; return fail(c, GZ_E_NOMEM, "pinned transfer buffer (%zu bytes)", cap); }
            c->h_xfer_cap[b] = cap;
        }
        if (!c->ev_xfer[b]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_xfer[b], hipEventDisableTiming));
    }
    return GZ_OK;
}

// Host -> device, ordered on stream s.  On return the bytes have LEFT src (the caller may reuse it); they may still be on their
// way to the device.
int copy_in(gz_ctx* c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t s)
{
    if (!bytes) return GZ_OK;
    if (bytes >= 65536 && is_pinned(src_host, bytes)) {
        HIPCHK(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        return GZ_OK;
    }
    int rc = xfer_need(c, bytes);
    if (rc) return rc;
    const size_t chunk = std::min(c->h_xfer_cap[0], c->h_xfer_cap[1]);
    int b = 0;
    for (size_t o = 0; o < bytes; o += chunk, b ^= 1) {
        const size_t n = std::min(chunk, bytes - o);
        if (c->xfer_busy[b]) { HIPCHK(c, hipEventSynchronize(c->ev_xfer[b])); c->xfer_busy[b] = false; }
        std::memcpy(c->h_xfer[b], (const uint8_t*)src_host + o, n);
        HIPCHK(c, hipMemcpyAsync((uint8_t*)dst_dev + o, c->h_xfer[b], n, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipEventRecord(c->ev_xfer[b], s));
        c->xfer_busy[b] = true;
    }
    return GZ_OK;
}

// Device -> host, behind whatever stream s holds.  Every byte is in dst on return.  Piece k + 1 is on the bus while piece k is
// copied out of its pinned buffer (by the pool's threads when there is one).
int copy_out(gz_ctx* c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t s, HostPool* pool = nullptr)
{
    if (!bytes) return GZ_OK;
    if (bytes >= 65536 && is_pinned(dst_host, bytes)) {
        HIPCHK(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return GZ_OK;
    }
    int rc = xfer_need(c, bytes);                               // (both buffers are idle after this)
    if (rc) return rc;
    const size_t chunk = std::min(c->h_xfer_cap[0], c->h_xfer_cap[1]);
    const size_t pieces = (bytes + chunk - 1) / chunk;
    auto issue = [&](size_t k) -> int {
        const size_t o = k * chunk, n = std::min(chunk, bytes - o);
        HIPCHK(c, hipMemcpyAsync(c->h_xfer[k & 1], (const uint8_t*)src_dev + o, n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_xfer[k & 1], s));
        return GZ_OK;
    };
    if ((rc = issue(0))) return rc;
    for (size_t k = 0; k < pieces; ++k) {
        HIPCHK(c, hipEventSynchronize(c->ev_xfer[k & 1]));
        if (k + 1 < pieces && (rc = issue(k + 1))) return rc;
        const size_t o = k * chunk, n = std::min(chunk, bytes - o);
        uint8_t* d = (uint8_t*)dst_host + o;
        const uint8_t* h = c->h_xfer[k & 1];
        if (pool && pool->threads()) pool->parallel(n, (size_t)256 << 10, [=](size_t lo, size_t hi) { std::memcpy(d + lo, h + lo, hi - lo); });
        else std::memcpy(d, h, n);
    }
    return GZ_OK;
}

// a few bytes (a total, a flag word) from the device, through the context's pinned scratch words
int copy_out_small(gz_ctx* c, void* dst_host, const void* src_dev, size_t bytes, hipStream_t s)
{
    if (bytes > 64) return copy_out(c, dst_host, src_dev, bytes, s);
    uint8_t* h = reinterpret_cast<uint8_t*>(c->h_pick) + 448;           // (h_pick is 512 bytes: [0, 256) the picks and block totals, [256, 512) scratch)
    HIPCHK(c, hipMemcpyAsync(h, src_dev, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::memcpy(dst_host, h, bytes);
    return GZ_OK;
}

}  // namespace
