// Test infrastructure: the HIP-free part of the library's host paths (csrc/gz_hostpool.h) on its own, built with -fsanitize=thread
// and with -fsanitize=address,undefined by tests/test_host_tables.py:
//   1. expand_rows_host through a HostPool the way csr_core uses it (sub-batches, three rotating slots, tagged jobs, wait_tag before a
//      slot is reused) against a straightforward restatement of the padding rule (tokenize.py:141-152), 16- and 32-bit entries, real
//      tokens equal to the pad id, rows cut at max_len, 0 / 1 / 5 worker threads;
//   2. the pool alone: parallel() beside tagged jobs, thousands of tiny jobs, destruction with jobs still queued;
//   3. the pinned-block registry under concurrent add / remove / holds.
#include "../../genz-tokenize_amd/csrc/gz_hostpool.h"

#include <atomic>
#include <random>

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAILED: %s (line %d)\n", #x, __LINE__); ++failures; } } while (0)

template <typename E>
static void expand_case(int threads, int64_t n_rows, int32_t max_len, int32_t pad, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<int32_t> n_real((size_t)n_rows);
    std::vector<uint32_t> first((size_t)n_rows + 1, 0);
    for (int64_t r = 0; r < n_rows; ++r) {
        n_real[(size_t)r] = (int32_t)(rng() % (uint32_t)(max_len + 1));
        first[(size_t)r + 1] = first[(size_t)r] + (uint32_t)n_real[(size_t)r];
    }
    std::vector<E> tok(first[(size_t)n_rows] + 1);
    for (size_t i = 0; i < tok.size(); ++i) tok[i] = (E)((i % 7 == 0) ? pad : (int32_t)(rng() % 60000));      // (real tokens equal to the pad id: mask 0 there)
    std::vector<int32_t> ids((size_t)n_rows * max_len, -1), mask((size_t)n_rows * max_len, -1);
    {
        HostPool pool(threads);
        // sub-batches of rows, three rotating slots (tags), jobs of a few rows each, a slot waited for before it is "reused"
        const int64_t per_sub = n_rows / 7 + 1, per_job = 37;
        int sub = 0;
        for (int64_t lo = 0; lo < n_rows; lo += per_sub, ++sub) {
            const int64_t hi = std::min(n_rows, lo + per_sub);
            const int slot = sub % 3;
            pool.wait_tag(slot);
            for (int64_t r0 = lo; r0 < hi; r0 += per_job) {
                const int64_t r1 = std::min(hi, r0 + per_job);
                const E* t = tok.data(); const uint32_t* f = first.data(); const int32_t* nr = n_real.data();
                int32_t* di = ids.data(); int32_t* dm = mask.data();
                pool.submit(slot, [=] { expand_rows_host(t, f, nr, r0, r1, max_len, pad, di, dm); });
            }
        }
        pool.wait_all();
    }
    size_t wrong = 0;
    for (int64_t r = 0; r < n_rows; ++r)
        for (int32_t i = 0; i < max_len; ++i) {
            const int32_t want = i < n_real[(size_t)r] ? (int32_t)tok[first[(size_t)r] + (uint32_t)i] : pad;
            wrong += ids[(size_t)r * max_len + i] != want;
            wrong += mask[(size_t)r * max_len + i] != (want != pad ? 1 : 0);
        }
    if (wrong) { std::printf("expand: %zu wrong cells (threads %d, rows %lld, max_len %d, %zu-byte entries)\n", wrong, threads, (long long)n_rows, max_len, sizeof(E)); ++failures; }
}

int main()
{
    for (int threads : {0, 1, 5}) {
        expand_case<uint16_t>(threads, 5000, 64, 0, 1u + (unsigned)threads);
        expand_case<int32_t>(threads, 3000, 10, 7, 11u + (unsigned)threads);
        expand_case<uint16_t>(threads, 1, 256, 3, 21u);
        expand_case<uint16_t>(threads, 257, 1, 0, 31u);
    }
    {
        // the pool alone: parallel() beside tagged jobs; many tiny jobs; a pool destroyed with jobs still queued runs them all
        std::atomic<long long> sum{0};
        {
            HostPool pool(4);
            for (int i = 0; i < 20000; ++i) pool.submit(i % 3, [&sum, i] { sum += i; });
            std::vector<int> v(100000, 0);
            int* d = v.data();
            pool.parallel(v.size(), 1000, [=](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) d[k] = (int)k; });
            bool ok = true;
            for (size_t k = 0; k < v.size(); ++k) ok = ok && v[k] == (int)k;
            CHECK(ok);
            for (int i = 0; i < 5000; ++i) pool.submit(5, [&sum] { sum += 1; });
        }                                                       // (destructor: the queue is drained, the threads joined)
        CHECK(sum.load() == 19999LL * 20000 / 2 + 5000);
        HostPool none(0);
        int ran = 0;
        none.submit(0, [&ran] { ++ran; });
        none.parallel(10, 1, [&ran](size_t lo, size_t hi) { ran += (int)(hi - lo); });
        none.wait_all();
        CHECK(ran == 11 && none.threads() == 0);
    }
    {
        PinnedRegistry reg;
        std::vector<std::thread> th;
        std::atomic<int> bad{0};
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&reg, &bad, t] {
                for (int i = 0; i < 2000; ++i) {
                    const uintptr_t base = 0x10000000u + (uintptr_t)(t * 4096 + (i % 16)) * 0x100000u;
                    reg.add((const void*)base, 4096);
                    if (!reg.holds((const void*)(base + 100), 200)) ++bad;
                    if (reg.holds((const void*)(base + 4000), 200)) ++bad;          // (runs past the block's end)
                    reg.remove((const void*)base);
                    if (reg.holds((const void*)(base + 100), 1)) ++bad;
                }
            });
        for (auto& x : th) x.join();
        CHECK(bad.load() == 0);
        CHECK(!reg.holds((const void*)0x1234, 1));
    }
    CHECK(host_cpus() >= 1 && host_cpus() <= 32);
    std::printf(failures ? "host pool driver: %d FAILURES\n" : "host pool driver ok\n", failures);
    return failures ? 1 : 0;
}
