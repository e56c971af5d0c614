// Host-only part of the C ABI (include/genz_tokenize.h): the loader + table builder exposed without a GPU
// (gz_host_tables_*).  No HIP in this file: it is also built with g++ -fsanitize=address,undefined together with
// gz_tables.cpp (make asan -> libgenz_tokenize_host_asan.so) so that the CPU test-suite can run the parser of
// untrusted file bytes (loader rules L1-L8) under the sanitizers.
#include "../../include/genz_tokenize.h"
#include "gz_common.h"

#include <cstring>
#include <new>
#include <string>

std::string& gz_create_err()
{
    static thread_local std::string err;
    return err;
}

GzOptions& gz_default_options()
{
    static GzOptions o;
    return o;
}

int gz_option_set(GzOptions& o, const char* key, int64_t v)
{
    if (!key) return GZ_E_INVALID;
    struct Key { const char* name; int32_t GzOptions::*p32; int64_t GzOptions::*p64; int64_t lo, hi; };
    static const Key keys[] = {
        {"small", &GzOptions::small, nullptr, 0, 1}, {"small_wgs", &GzOptions::small_wgs, nullptr, 1, 1 << 20}, {"host_direct", &GzOptions::host_direct, nullptr, 0, 1 << 20},
        {"assemble", &GzOptions::assemble, nullptr, 1, 3}, {"word_table", &GzOptions::word_table, nullptr, 0, 1},
        {"pp_fused", &GzOptions::pp_fused, nullptr, 0, 1}, {"sub_batches", &GzOptions::sub_batches, nullptr, 1, 8},
        {"docs_per_wave", &GzOptions::docs_per_wave, nullptr, 0, 16}, {"side", &GzOptions::side, nullptr, 0, 1},
        {"brk_side", &GzOptions::brk_side, nullptr, 0, 1}, {"scan_multi", nullptr, &GzOptions::scan_multi, 0, (int64_t)1 << 40},
        {"near_limit", nullptr, &GzOptions::near_limit, 0, 1 << 25}, {"hot_wgs", &GzOptions::hot_wgs, nullptr, 0, 1 << 16},
        {"hot_miss_wgs", &GzOptions::hot_miss_wgs, nullptr, 0, 1 << 16}, {"m2_split_min", nullptr, &GzOptions::m2_split_min, 0, (int64_t)1 << 32},
        {"m2_split_always", &GzOptions::m2_split_always, nullptr, 0, 1},
        {"tab_slack", &GzOptions::tab_slack, nullptr, 2, 64}, {"ph_force_overflow", &GzOptions::ph_force_overflow, nullptr, 0, 1 << 20},
        {"ph_hot_slots", &GzOptions::ph_hot_slots, nullptr, 0, 8192}, {"word_weights", &GzOptions::word_weights, nullptr, 0, 2},
        {"diag_poison", &GzOptions::diag_poison, nullptr, 0, 1}, {"rows_dpw", &GzOptions::rows_dpw, nullptr, 0, 64},
        {"rows_dbg", &GzOptions::rows_dbg, nullptr, 0, 255}, {"ablate", &GzOptions::ablate, nullptr, 0, 1 << 20},
        {"diag_guard", &GzOptions::diag_guard, nullptr, 0, 2}, {"diag_exact", &GzOptions::diag_exact, nullptr, 0, 1}, {"diag_fresh", &GzOptions::diag_fresh, nullptr, 0, 256},
        {"diag_fresh_only", &GzOptions::diag_fresh_only, nullptr, -1, 1 << 20}, {"host_threads", &GzOptions::host_threads, nullptr, 0, 256},
        {"dense_csr", &GzOptions::dense_csr, nullptr, 0, 1}, {"host_hints", &GzOptions::host_hints, nullptr, 0, 3}, {"inject_bad_alloc", &GzOptions::inject_bad_alloc, nullptr, 0, 1 << 30},
    };
    for (const Key& k : keys)
        if (std::strcmp(k.name, key) == 0) {
            if (v < k.lo || v > k.hi) return GZ_E_INVALID;
            if (k.p32) o.*(k.p32) = (int32_t)v; else o.*(k.p64) = v;
            return GZ_OK;
        }
    return GZ_E_INVALID;
}

extern "C" {

#ifdef GZ_HOST_ONLY
// (no contexts in the sanitizer build: every key goes to the process-wide defaults, which the table builder reads)
int gz_debug_set(gz_ctx*, const char* key, int64_t value)
{
    const int rc = gz_option_set(gz_default_options(), key, value);
    if (rc) gz_create_err() = std::string("gz_debug_set: unknown key or value out of range: ") + (key ? key : "(null)");
    return rc;
}

// the sanitizer build has no contexts: only the two context-free entry points of the main library
int gz_version(void) { return GZ_VERSION; }
const char* gz_last_error(gz_ctx*) { return gz_create_err().c_str(); }
#endif

// ---- host-only table build (diagnostics, offline checks; no GPU) ------------------------------------------------------
struct gz_host_tables { GzHostTables T; };

int gz_host_tables_create(const uint8_t* vocab, size_t vocab_len, const uint8_t* bpe, size_t bpe_len,
                          const char* const specials[5], gz_host_tables** out)
{
    if (!out || !specials) return GZ_E_INVALID;
    *out = nullptr;
    gz_host_tables* h = new (std::nothrow) gz_host_tables();
    if (!h) return GZ_E_NOMEM;
    static const uint8_t empty = 0;
    int rc;
    try {
        std::string err;
        rc = gz_build_tables(vocab ? vocab : &empty, vocab_len, bpe ? bpe : &empty, bpe_len, specials, h->T, err);
        if (rc) gz_create_err() = err;
    } catch (...) {
        rc = GZ_E_NOMEM;
    }
    if (rc) { delete h; return rc; }
    *out = h;
    return GZ_OK;
}

void gz_host_tables_destroy(gz_host_tables* t) { delete t; }

int gz_host_tables_array(gz_host_tables* t, int which, const void** data, int64_t* count)
{
    if (!t || !data || !count) return GZ_E_INVALID;
    GzHostTables& H = t->T;
    switch (which) {
        case 1: *data = H.merges.data();   *count = (int64_t)H.merges.size(); break;
        case 2: *data = H.sym_ids.data();  *count = (int64_t)H.sym_ids.size(); break;
        case 3: *data = H.bmp.data();      *count = (int64_t)H.bmp.size(); break;
        case 4: *data = H.astral.data();   *count = (int64_t)H.astral.size(); break;
        case 5: *data = H.special_ids;     *count = 5; break;
        // the pair table, perfectly hashed: entries, displacement array, {nbuckets, bshift, sshift, slots, k1, k2, keys in
        // overflow buckets}, hot set
        case 6: *data = H.pair8.data();    *count = (int64_t)H.pair8.size(); break;
        case 7: *data = H.pair_ph.disp.data(); *count = (int64_t)H.pair_ph.disp.size(); break;
        case 8: {
            static thread_local uint32_t ph[7];
            const GzPhHost& P = H.pair_ph;
            const uint32_t v[7] = {P.nbuckets, P.bshift, P.sshift, P.slots, P.k1, P.k2, P.n_overflow};
            for (int i = 0; i < 7; ++i) ph[i] = v[i];
            *data = ph; *count = 7; break;
        }
        case 9: *data = H.pair_hot.data(); *count = (int64_t)H.pair_hot.size(); break;
        default: *data = nullptr; *count = 0; return GZ_E_INVALID;      // (0 included: the linear-probing pair table is gone)
    }
    return GZ_OK;
}

int64_t gz_limit(int which)
{
    switch (which) {
        case 0: return GZ_TEXT_BYTES_LIMIT;
        case 1: return GZ_PP_TAIL32_LIMIT;
        default: return -1;
    }
}

int gz_host_tables_vocab_entry(gz_host_tables* t, int64_t i, const uint8_t** utf8, int32_t* len, int32_t* id)
{
    if (!t || i < 0 || i >= (int64_t)t->T.enc_words.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)t->T.enc_words[i].data();
    if (len) *len = (int32_t)t->T.enc_words[i].size();
    if (id) *id = t->T.enc_ids[i];
    return GZ_OK;
}

int gz_host_tables_merge_entry(gz_host_tables* t, int64_t i, const uint8_t** utf8, int32_t* len, int32_t* n_fields, int32_t* rank)
{
    if (!t || i < 0 || i >= (int64_t)t->T.rank_keys.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)t->T.rank_keys[i].data();
    if (len) *len = (int32_t)t->T.rank_keys[i].size();
    if (n_fields) *n_fields = t->T.rank_nfields[i];
    if (rank) *rank = t->T.rank_vals[i];
    return GZ_OK;
}

int gz_host_tables_symbol(gz_host_tables* t, int32_t symbol, const uint8_t** utf8, int32_t* len)
{
    if (!t || symbol < 0 || symbol >= (int32_t)t->T.symbols.size()) return GZ_E_INVALID;
    if (utf8) *utf8 = (const uint8_t*)t->T.symbols[symbol].data();
    if (len) *len = (int32_t)t->T.symbols[symbol].size();
    return GZ_OK;
}

}  // extern "C"
