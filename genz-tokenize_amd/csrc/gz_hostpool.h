// The HIP-free part of the host paths (gz_hostpath.h includes it; tests/native/hostpool_main.cpp builds it alone under the thread and
// address sanitizers): worker threads for the length of one call, the registry of pinned blocks, the hints for fresh output arrays,
// and the padding of CSR rows into dense arrays.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

// ---- how many worker threads a host call may use: the processors this process may run on (affinity mask, cgroup quota), at
// most 32 -- the work is first-touch page faults and streaming stores: more threads than that gain nothing.
int host_cpus()
{
    static const int n = [] {
        int k = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) k = CPU_COUNT(&set);
        if (k <= 0) k = (int)sysconf(_SC_NPROCESSORS_ONLN);
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) {
                const long long quota = atoll(q);
                const int by_quota = (int)((quota + period - 1) / period);
                if (by_quota >= 1 && by_quota < k) k = by_quota;
            }
            fclose(f);
        }
        return k < 1 ? 1 : k > 32 ? 32 : k;
    }();
    return n;
}

// ---- a few worker threads for the length of ONE host call (made by the call, joined before it returns: nothing of the library
// runs when no call is in progress).  Jobs carry a tag (the pinned slot they read); wait_tag() returns when no job of that tag is
// queued or running.  With 0 threads submit() runs the job on the caller's thread.
class HostPool {
public:
    explicit HostPool(int n_threads)
    {
        // (a thread that cannot be made -- a process at its limit -- is simply not there: the call runs with the ones it got, or on
        //  the caller's thread.  Nothing may escape this constructor once a thread runs: the vector's destructor would meet a
        //  joinable thread, which is std::terminate)
        try {
            th_.reserve((size_t)(n_threads > 0 ? n_threads : 0));
            for (int i = 0; i < n_threads; ++i) th_.emplace_back([this] { run(); });
        } catch (...) {
        }
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;
    int threads() const { return (int)th_.size(); }
    void submit(int tag, std::function<void()> f)
    {
        if (th_.empty()) { f(); return; }
        { std::lock_guard<std::mutex> lk(m_); q_.emplace_back(tag, std::move(f)); ++busy_[tag & 7]; }
        cv_.notify_one();
    }
    void wait_tag(int tag)
    {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return busy_[tag & 7] == 0; });
    }
    void wait_all() { for (int t = 0; t < 8; ++t) wait_tag(t); }
    // [0, n) cut into pieces of at least `grain`, one job each, all finished on return
    template <class F>
    void parallel(size_t n, size_t grain, F f)
    {
        const size_t parts = th_.empty() ? 1 : std::max<size_t>(1, std::min<size_t>((size_t)th_.size(), n / (grain ? grain : 1)));
        if (parts <= 1) { if (n) f((size_t)0, n); return; }
        for (size_t p = 0; p < parts; ++p) {
            const size_t lo = n * p / parts, hi = n * (p + 1) / parts;
            submit(7, [=] { f(lo, hi); });
        }
        wait_tag(7);
    }
private:
    void run()
    {
        for (;;) {
            std::pair<int, std::function<void()>> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job.second();
            { std::lock_guard<std::mutex> lk(m_); --busy_[job.first & 7]; }
            done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<std::pair<int, std::function<void()>>> q_;
    int busy_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool stop_ = false;
};

// Is this host pointer inside a page-locked block made by gz_host_alloc?  (The library's own registry: asking the runtime about
// a pointer it has never seen -- any numpy array -- is an error there, and is logged as one.)  Pinned memory from elsewhere is
// treated like pageable memory: staged, which is always correct.
struct PinnedRegistry {
    std::mutex mu;
    std::map<uintptr_t, size_t> blocks;                          // start -> bytes
    void add(const void* p, size_t n) { std::lock_guard<std::mutex> lk(mu); blocks[(uintptr_t)p] = n; }
    void remove(const void* p) { std::lock_guard<std::mutex> lk(mu); blocks.erase((uintptr_t)p); }
    bool holds(const void* p, size_t n)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = blocks.upper_bound((uintptr_t)p);
        if (it == blocks.begin()) return false;
        --it;
        return (uintptr_t)p >= it->first && (uintptr_t)p + n <= it->first + it->second;
    }
};
PinnedRegistry& pinned_registry() { static PinnedRegistry r; return r; }
bool is_pinned(const void* p, size_t bytes = 1) { return p && pinned_registry().holds(p, bytes ? bytes : 1); }

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
// The arrays a large call fills are, as a rule, fresh allocations nobody has touched: every 4-KiB page of them costs a page
// fault at its first store.  Two hints, both harmless where they do not apply (errors are ignored): the range may be backed by
// huge pages (one fault per 2 MiB where the system allows it), and a piece about to be written is populated by ONE system call
// instead of a fault per page.
void hint_huge(void* p, size_t bytes)
{
    const uintptr_t lo = ((uintptr_t)p + 0x1FFFFF) & ~(uintptr_t)0x1FFFFF, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)0x1FFFFF;
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
}
void hint_populate(void* p, size_t bytes)
{
    const uintptr_t lo = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE);
}

// CSR rows -> dense [rows, max_len] int32 ids + mask in the caller's arrays: rows [r0, r1) of a sub-batch whose entries lie in
// `tok` (16- or 32-bit) from row_first[r] on.  Padding with the pad id (tokenize.py:141-146), mask = ids != pad (:148-152).
template <typename E>
void expand_rows_host(const E* tok, const uint32_t* row_first, const int32_t* n_real, int64_t r0, int64_t r1, int32_t max_len, int32_t pad,
                      int32_t* ids, int32_t* mask)
{
    for (int64_t r = r0; r < r1; ++r) {
        const E* src = tok + row_first[r];
        int32_t n = n_real[r];
        if (n < 0) n = 0;
        if (n > max_len) n = max_len;
        int32_t* di = ids + r * (int64_t)max_len;
        int32_t* dm = mask + r * (int64_t)max_len;
        for (int32_t i = 0; i < n; ++i) { const int32_t v = (int32_t)src[i]; di[i] = v; dm[i] = v != pad; }
        for (int32_t i = n; i < max_len; ++i) { di[i] = pad; dm[i] = 0; }
    }
}

}  // namespace
