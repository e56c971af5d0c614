// Test driver (tests/test_host_tables.py::test_host_builder_threads_under_tsan): the host table builder runs its vocab side and the
// pair table's perfect hash on helper threads (gz_tables.cpp); built with g++ -fsanitize=thread together with gz_tables.cpp and run
// over the table files given on the command line, several times.  ThreadSanitizer reports go to stderr and make the exit code 66.
#include "gz_common.h"
#include "genz_tokenize.h"
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

static std::string slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    std::stringstream s;
    s << f.rdbuf();
    return s.str();
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string vocab = slurp(argv[1]), bpe = slurp(argv[2]);
    const char* specials[5] = {"<pad>", "<s>", "</s>", "<mask>", "<unk>"};
    for (int i = 0; i < 3; ++i) {
        GzHostTables T;
        std::string err;
        const int rc = gz_build_tables((const uint8_t*)vocab.data(), vocab.size(), (const uint8_t*)bpe.data(), bpe.size(), specials, T, err);
        if (rc != 0) { std::fprintf(stderr, "gz_build_tables: %d %s\n", rc, err.c_str()); return 1; }
        std::printf("symbols %zu merges %zu pair slots %zu\n", T.symbols.size(), T.merges.size(), T.pair8.size());
    }
    // a vocab file that is not UTF-8 beside a valid merge file: the error of the helper thread must come back
    {
        GzHostTables T;
        std::string err, bad = vocab;
        bad[bad.size() / 2] = (char)0xFF;
        const int rc = gz_build_tables((const uint8_t*)bad.data(), bad.size(), (const uint8_t*)bpe.data(), bpe.size(), specials, T, err);
        if (rc != GZ_E_UTF8) { std::fprintf(stderr, "expected GZ_E_UTF8, got %d\n", rc); return 1; }
    }
    std::printf("tsan driver ok\n");
    return 0;
}
