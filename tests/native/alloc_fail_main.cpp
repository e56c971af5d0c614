// Test infrastructure: the host-only entry points of the C ABI (gz_host_tables_create: the table builder over untrusted
// file bytes, the largest allocator of the host side) under an operator new that FAILS at the k-th allocation, for k = 1, 2, 3 ...:
// whatever allocation runs out, the call must come back with an error code (GZ_E_NOMEM) -- an exception that escaped the
// extern "C" boundary would be std::terminate, i.e. the silent SIGABRT this file exists to rule out.  Linked with gz_tables.cpp +
// gz_host_api.cpp (-DGZ_HOST_ONLY); run by tests/test_host_tables.py.
//   usage: alloc_fail_main vocab.txt bpe.codes [stride]
#include "../../include/genz_tokenize.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

static std::atomic<long long> g_count{0}, g_fail_at{-1};

void* operator new(std::size_t n)
{
    const long long k = ++g_count;
    if (k == g_fail_at.load()) throw std::bad_alloc();
    if (void* p = std::malloc(n ? n : 1)) return p;
    throw std::bad_alloc();
}
void* operator new[](std::size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

static std::vector<unsigned char> slurp(const char* path)
{
    std::vector<unsigned char> v;
    if (FILE* f = std::fopen(path, "rb")) {
        unsigned char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        std::fclose(f);
    }
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::vector<unsigned char> vocab = slurp(argv[1]), bpe = slurp(argv[2]);
    const long long stride = argc > 3 ? std::atoll(argv[3]) : 1;
    const char* specials[5] = {"<pad>", "<s>", "</s>", "<mask>", "<unk>"};
    // how many allocations does a clean build make?
    g_count = 0;
    gz_host_tables* t = nullptr;
    int rc = gz_host_tables_create(vocab.data(), vocab.size(), bpe.data(), bpe.size(), specials, &t);
    const long long clean = g_count.load();
    if (rc != GZ_OK || !t) { std::printf("clean build failed: %d\n", rc); return 1; }
    gz_host_tables_destroy(t);
    long long tried = 0, nomem = 0, ok = 0, other = 0;
    for (long long k = 1; k <= clean + 2; k += (k < 400 ? 1 : stride)) {
        g_count = 0;
        g_fail_at = k;
        t = nullptr;
        rc = gz_host_tables_create(vocab.data(), vocab.size(), bpe.data(), bpe.size(), specials, &t);
        g_fail_at = -1;
        ++tried;
        if (rc == GZ_E_NOMEM && !t) ++nomem;
        else if (rc == GZ_OK && t) { ++ok; gz_host_tables_destroy(t); }       // (k beyond the allocations this build makes)
        else { ++other; std::printf("k = %lld: rc %d, tables %p\n", k, rc, (void*)t); }
    }
    std::printf("allocations of a clean build: %lld; failure points tried: %lld; GZ_E_NOMEM: %lld; completed: %lld; other: %lld\n", clean, tried, nomem, ok, other);
    return other == 0 && nomem > 0 ? 0 : 1;
}
