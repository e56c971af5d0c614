// K0 -- host table builder.
//
// Replaces Tokenize.__init__ / add_vocab_file / add_bpe_file (reference genz_tokenize/tokenize.py:31-57):
// parses the RAW BYTES of vocab.txt and bpe.codes with the reference's exact text-mode semantics (loader rules
// L1-L8 of SURVEY.md §8(a)) and then re-expresses the two Python dicts as integer tables for the GPU:
//
//   symbols   every string bpe() can hold in its `word` tuple: merge operands, merge results, and the
//             single-character forms the vocab knows (c  and  c+"</w>").  Interned by string identity.
//   pair8     perfectly hashed (hash and displace)  (left symbol, right symbol) -> rank = merged symbol   [bpe_ranks.get(pair)]
//   merges    rank -> (left, right, merged symbol)                              [first + second, tokenize.py:88]
//   sym_ids   symbol -> vocab id when emitted as a non-final piece (string + "@@") and as the final piece
//             (string minus "</w>")                                              [encoder.get(tok, unk), :120-121]
//   bmp / astral   code point -> initial symbol (plain, and with "</w>")         [tuple(token), :63-64]
#include "gz_common.h"
#include "../../include/genz_tokenize.h"

#include <algorithm>
#include <thread>
#include <cstring>
#include <unordered_map>

namespace {

using U32 = std::u32string;

bool is_space(char32_t c)
{
    // str.isspace() / regex \s on str: the 29 code points of SURVEY.md "hard part 3"
    return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

// codecs 'utf-8', errors='strict'
bool decode_utf8_strict(const uint8_t* p, size_t n, U32& out)
{
    out.clear();
    out.reserve(n);
    size_t i = 0;
    while (i < n) {
        uint8_t b = p[i];
        if (b < 0x80) { out.push_back(b); ++i; continue; }
        int len; char32_t cp;
        if (b >= 0xC2 && b <= 0xDF) { len = 2; cp = b & 0x1F; }
        else if (b >= 0xE0 && b <= 0xEF) { len = 3; cp = b & 0x0F; }
        else if (b >= 0xF0 && b <= 0xF4) { len = 4; cp = b & 0x07; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; ++k) {
            uint8_t c = p[i + k];
            if ((c & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (c & 0x3F);
        }
        if (len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        out.push_back(cp);
        i += len;
    }
    return true;
}

void append_utf8(std::string& s, char32_t c)
{
    if (c < 0x80) s.push_back((char)c);
    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) {
        s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (c & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (c >> 18))); s.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F)));
    }
}

std::string to_utf8(const char32_t* b, const char32_t* e)
{
    std::string s;
    for (; b != e; ++b) append_utf8(s, *b);
    return s;
}

// open(..., 'r') newline=None: "\r\n" and "\r" become "\n"
void universal_newlines(U32& s)
{
    size_t w = 0;
    for (size_t r = 0; r < s.size(); ++r) {
        char32_t c = s[r];
        if (c == '\r') {
            if (r + 1 < s.size() && s[r + 1] == '\n') ++r;
            c = '\n';
        }
        s[w++] = c;
    }
    s.resize(w);
}

// number of code points of a UTF-8 string that is known valid (or -1 cheaply bounded)
size_t count_cps(const std::string& s)
{
    size_t n = 0;
    for (unsigned char c : s) n += (c & 0xC0) != 0x80;
    return n;
}

char32_t first_cp(const std::string& s)
{
    auto u = [&](size_t i) -> uint32_t { return (unsigned char)s[i]; };     // (bytes as unsigned: no signed shifts)
    const uint32_t b = u(0);
    if (b < 0x80) return b;
    if (b < 0xE0) return ((b & 0x1Fu) << 6) | (u(1) & 0x3Fu);
    if (b < 0xF0) return ((b & 0x0Fu) << 12) | ((u(1) & 0x3Fu) << 6) | (u(2) & 0x3Fu);
    return ((b & 0x07u) << 18) | ((u(1) & 0x3Fu) << 12) | ((u(2) & 0x3Fu) << 6) | (u(3) & 0x3Fu);
}

bool ends_with(const std::string& s, const char* suf)
{
    size_t k = std::strlen(suf);
    return s.size() >= k && std::memcmp(s.data() + s.size() - k, suf, k) == 0;
}

// string -> its index in a vector of strings the caller keeps (open addressing, cached hashes: the loader makes half a
// million dictionary operations on short strings; std::unordered_map<std::string, ...> spent most of the table build in
// allocations and bucket walks)
struct StrIndex {
    std::vector<uint32_t> slot;   // index + 1 (0: empty)
    std::vector<uint32_t> hsh;
    size_t n = 0;
    static uint32_t hash(const char* p, size_t len)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (len * 0xFF51AFD7ED558CCDull);
        while (len >= 8) { uint64_t v; std::memcpy(&v, p, 8); h = (h ^ v) * 0xC2B2AE3D27D4EB4Full; h ^= h >> 29; p += 8; len -= 8; }
        uint64_t v = 0;
        if (len) std::memcpy(&v, p, len);
        h = (h ^ v) * 0x9FB21C651E98DF25ull;
        h ^= h >> 32;
        return (uint32_t)h | 1u;                                  // (never 0)
    }
    void reserve(size_t k)
    {
        size_t cap = 16;
        while (cap < 2 * k + 2) cap <<= 1;
        if (cap > slot.size()) rebuild(cap);
    }
    void rebuild(size_t cap)
    {
        std::vector<uint32_t> os, oh;
        os.swap(slot); oh.swap(hsh);
        slot.assign(cap, 0); hsh.assign(cap, 0);
        for (size_t i = 0; i < os.size(); ++i) if (os[i]) place(oh[i], os[i] - 1);
    }
    void place(uint32_t h, uint32_t idx)
    {
        size_t m = slot.size() - 1, q = h & m;
        while (slot[q]) q = (q + 1) & m;
        slot[q] = idx + 1; hsh[q] = h;
    }
    // index of the key, or UINT32_MAX
    uint32_t find(const std::vector<std::string>& keys, const char* p, size_t len, uint32_t h) const
    {
        if (slot.empty()) return 0xFFFFFFFFu;
        size_t m = slot.size() - 1, q = h & m;
        while (slot[q]) {
            if (hsh[q] == h) { const std::string& k = keys[slot[q] - 1]; if (k.size() == len && std::memcmp(k.data(), p, len) == 0) return slot[q] - 1; }
            q = (q + 1) & m;
        }
        return 0xFFFFFFFFu;
    }
    void insert(uint32_t h, uint32_t idx)                         // (the key is not in the index)
    {
        if (2 * (n + 1) + 2 > slot.size()) rebuild(slot.empty() ? 16 : slot.size() * 2);
        place(h, idx);
        ++n;
    }
};

struct OrderedDict {            // a Python dict[str, int]: insertion order, overwrite keeps position
    StrIndex pos;
    std::vector<std::string> keys;
    std::vector<int32_t> vals;
    std::vector<uint64_t> aux;  // a second value per key that is NOT part of the dict (a hint / a field count): the last set() wins
    void reserve(size_t n) { pos.reserve(n); keys.reserve(n); vals.reserve(n); aux.reserve(n); }
    void set(const std::string& k, int32_t v, uint64_t a = 0)
    {
        const uint32_t h = StrIndex::hash(k.data(), k.size());
        const uint32_t i = pos.find(keys, k.data(), k.size(), h);
        if (i == 0xFFFFFFFFu) { pos.insert(h, (uint32_t)keys.size()); keys.push_back(k); vals.push_back(v); aux.push_back(a); }
        else { vals[i] = v; aux[i] = a; }
    }
    const int32_t* get(const char* p, size_t len) const
    {
        const uint32_t i = pos.find(keys, p, len, StrIndex::hash(p, len));
        return i == 0xFFFFFFFFu ? nullptr : &vals[i];
    }
    const int32_t* get(const std::string& k) const { return get(k.data(), k.size()); }
    size_t size() const { return keys.size(); }
};

}  // namespace

bool gz_is_plain_word(const uint8_t* p, size_t n)
{
    U32 cps;
    if (n == 0 || !decode_utf8_strict(p, n, cps)) return false;
    for (char32_t c : cps) if (is_space(c)) return false;
    return true;
}

int gz_build_tables(const uint8_t* vocab, size_t vocab_len, const uint8_t* bpe, size_t bpe_len,
                    const char* const specials[5], GzHostTables& T, std::string& err)
{
    T = GzHostTables();
    // The two files are independent until the symbols meet the vocabulary: the vocab side (decode, encoder) runs on a second
    // thread beside the merge side (decode, bpe_ranks, the merges' symbols); later the pair table is hashed on a second thread
    // beside the symbol -> id table.  (Nothing below throws past a thread: allocation failures are caught and reported.)
    U32 vtext, btext;
    OrderedDict enc;
    int rc_v = GZ_OK;
    std::string err_v;
    auto vocab_side = [&]() {
      try {
        if (!decode_utf8_strict(vocab, vocab_len, vtext)) { err_v = "vocab file: invalid UTF-8"; rc_v = GZ_E_UTF8; return; }
        universal_newlines(vtext);
        // ---- encoder (tokenize.py:31-37, :44-51) ------------------------------------------------------------
        {
            size_t lines = 8;
            for (char32_t c : vtext) lines += c == '\n';
            enc.reserve(lines);
        }
        for (int i = 0; i < 5; ++i) enc.set(specials[i], i);
        {
            size_t i = 0, n = vtext.size();
            while (i < n) {                                   // readlines(): split after every '\n', no empty tail
                size_t j = i;
                while (j < n && vtext[j] != '\n') ++j;
                size_t a = i, b = j;                          // line without its '\n'
                while (a < b && is_space(vtext[a])) ++a;      // .strip()
                while (b > a && is_space(vtext[b - 1])) --b;
                // idx = line.rfind(' '); word = line[:idx]   (idx == -1 -> drop the last character)
                size_t cut;
                size_t k = b;
                while (k > a && vtext[k - 1] != ' ') --k;
                if (k > a) cut = k - 1;                       // position of the last ' '
                else cut = (b > a) ? b - 1 : a;               // no space: line[:-1]; empty line: ''
                std::string word = to_utf8(vtext.data() + a, vtext.data() + cut);
                // what follows the last space is ignored by the reference; when it is a number (the bundled vocab: the word's corpus
                // count) it is kept as a placement HINT for the whole-word table (its most frequent words share a few lines)
                uint64_t cnt = 0;
                bool digits = k > a && cut + 1 < b;
                for (size_t q = cut + 1; q < b && digits; ++q) {
                    if (vtext[q] < '0' || vtext[q] > '9' || cnt > (1ull << 56)) digits = false;
                    else cnt = cnt * 10 + (uint64_t)(vtext[q] - '0');
                }
                enc.set(word, (int32_t)enc.size(), digits ? cnt : 0);       // len(encoder) BEFORE the insertion
                i = (j < n) ? j + 1 : j;
            }
        }
        T.enc_words = enc.keys;
        T.enc_ids = enc.vals;
        T.enc_count = enc.aux;
        for (int i = 0; i < 5; ++i) T.special_ids[i] = *enc.get(specials[i]);
      } catch (...) { err_v = "out of memory while reading the vocab file"; rc_v = GZ_E_NOMEM; }
    };
    // (a thread that cannot be started -- std::system_error -- is no reason to fail: the work is then done here, in line)
    auto start = [](std::thread& t, auto& work) { try { t = std::thread(work); } catch (...) { work(); } };
    std::thread vocab_thread;
    start(vocab_thread, vocab_side);
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } vocab_join{vocab_thread};      // (every return path joins)
    // the reference reads the vocab file first (tokenize.py:44-51, then :53-57): its error wins over one of the merge side
    auto fail_merge_side = [&](int rc, const char* what) { if (vocab_thread.joinable()) vocab_thread.join(); if (rc_v) { err = err_v; return rc_v; } err = what; return rc; };
    if (!decode_utf8_strict(bpe, bpe_len, btext)) return fail_merge_side(GZ_E_UTF8, "bpe file: invalid UTF-8");
    universal_newlines(btext);

    // ---- bpe_ranks (tokenize.py:53-57) --------------------------------------------------------------------
    OrderedDict ranks;                                    // key = fields joined by '\n'; aux = the number of fields
    {
        // read().split('\n')[:-1]
        std::vector<std::pair<size_t, size_t>> rows;
        size_t i = 0, n = btext.size();
        for (;;) {
            size_t j = i;
            while (j < n && btext[j] != '\n') ++j;
            rows.emplace_back(i, j);
            if (j >= n) break;
            i = j + 1;
        }
        rows.pop_back();
        if (rows.size() > GZ_MAX_RANKS) return fail_merge_side(GZ_E_LIMIT, "bpe file: too many lines");
        ranks.reserve(rows.size());
        for (size_t r = 0; r < rows.size(); ++r) {
            std::string key;
            int32_t nf = 0;
            size_t p = rows[r].first, e = rows[r].second;
            while (p < e) {                               // str.split()
                while (p < e && is_space(btext[p])) ++p;
                if (p >= e) break;
                size_t q = p;
                while (q < e && !is_space(btext[q])) ++q;
                if (nf) key.push_back('\n');
                key += to_utf8(btext.data() + p, btext.data() + q);
                ++nf;
                p = q;
            }
            ranks.set(key, (int32_t)r, (uint64_t)nf);
        }
        T.rank_nfields.resize(ranks.size());
        for (size_t k = 0; k < ranks.size(); ++k) T.rank_nfields[k] = (int32_t)ranks.aux[k];
        T.rank_keys = std::move(ranks.keys);              // (the dict is complete: its insertion-ordered halves move out)
        T.rank_vals = std::move(ranks.vals);
        T.merges.assign(rows.size(), GzMergeInfo{0, 0, 0, 0});
    }

    // ---- symbols --------------------------------------------------------------------------------------------
    // Numbering: the string a merge produces gets the merge's RANK as its id (two lines that spell the same string: the
    // smaller rank), so that a pair-table entry needs no separate "merged symbol" field; every other string (merge
    // operands that no merge produces, single characters) is numbered from the number of lines up.  Ranks of lines that
    // are no merges stay unused ids (empty strings).
    const size_t n_lines = T.merges.size();
    struct Pair { uint32_t a, b, rank; };
    std::vector<Pair> pairs;
    struct Fields { uint32_t key, nl, rank; };                  // a two-field line: its key (first '\n' second), where the '\n' is, its rank
    std::vector<Fields> two;
    two.reserve(T.rank_keys.size());
    // the strings the merges produce (first + second, tokenize.py:88), each with the smallest rank that spells it
    std::vector<std::string> mstr;
    std::vector<uint32_t> mrank;
    StrIndex midx;
    mstr.reserve(n_lines); mrank.reserve(n_lines); midx.reserve(n_lines);
    std::string m;
    for (size_t k = 0; k < T.rank_keys.size(); ++k) {
        if (T.rank_nfields[k] != 2) continue;             // such a key can never equal a (first, second) pair
        const std::string& key = T.rank_keys[k];
        const size_t nl = key.find('\n');
        two.push_back(Fields{(uint32_t)k, (uint32_t)nl, (uint32_t)T.rank_vals[k]});
        m.assign(key, 0, nl).append(key, nl + 1, std::string::npos);
        const uint32_t h = StrIndex::hash(m.data(), m.size());
        const uint32_t i = midx.find(mstr, m.data(), m.size(), h);
        if (i == 0xFFFFFFFFu) { midx.insert(h, (uint32_t)mstr.size()); mstr.push_back(m); mrank.push_back(two.back().rank); }
        else if (two.back().rank < mrank[i]) mrank[i] = two.back().rank;
    }
    T.symbols.assign(n_lines, std::string());
    StrIndex sidx;                                                // symbol string -> symbol id (the strings live in T.symbols)
    sidx.reserve(2 * n_lines + 1024);
    for (size_t i = 0; i < mstr.size(); ++i) {
        sidx.insert(StrIndex::hash(mstr[i].data(), mstr[i].size()), mrank[i]);
        T.symbols[mrank[i]] = std::move(mstr[i]);
    }
    auto intern = [&](const char* p, size_t len) -> uint32_t {
        const uint32_t h = StrIndex::hash(p, len);
        const uint32_t i = sidx.find(T.symbols, p, len, h);
        if (i != 0xFFFFFFFFu) return i;
        const uint32_t id = (uint32_t)T.symbols.size();
        sidx.insert(h, id);
        T.symbols.emplace_back(p, len);
        return id;
    };
    for (const Fields& f : two) {
        const std::string& key = T.rank_keys[f.key];
        m.assign(key, 0, f.nl).append(key, f.nl + 1, std::string::npos);
        const uint32_t ia = intern(key.data(), f.nl), ib = intern(key.data() + f.nl + 1, key.size() - f.nl - 1), im = intern(m.data(), m.size());
        T.merges[f.rank] = GzMergeInfo{ia, ib, im, 0};
        pairs.push_back({ia, ib, f.rank});
    }
    // ---- pair -> rank, perfectly hashed: 8-byte entries, one load per probe ------------------------------------
    int rc_p = GZ_OK;
    auto pair_side = [&]() {
      try {
        std::vector<uint32_t> slot_of;
        auto hashes = [](const void* ctx, size_t i, uint32_t k1, uint32_t k2, uint32_t* ha, uint32_t* hb) {
            const Pair& p = (*static_cast<const std::vector<Pair>*>(ctx))[i];
            *ha = gz_pair_ha(p.a, p.b, k1, k2); *hb = gz_pair_hb(p.a, p.b);
        };
        gz_ph_build(pairs.size(), hashes, &pairs, T.pair_ph, slot_of);
        T.pair8.assign(T.pair_ph.slots, GzPair8{0xFFFFFFFFu, 0xFFFFFFFFu});
        for (size_t i = 0; i < pairs.size(); ++i) {
            const Pair& p = pairs[i];
            const uint32_t merged = T.merges[p.rank].merged;
            T.pair8[slot_of[i]] = GzPair8{p.a | (p.b << 20), (p.b >> 12) | (p.rank << 9) | (merged != p.rank ? GZ_PAIR8_ALIAS : 0u)};
        }
        // hot set: merges are learned most frequent first, so the smallest ranks are the pairs running text asks for most;
        // direct-mapped by the top bits of ha, the smaller rank keeps a contested slot
        T.pair_hot.assign(GZ_PAIR_HOT_SLOTS, GzPair8{0xFFFFFFFFu, 0xFFFFFFFFu});
        std::vector<uint32_t> by_rank(pairs.size());
        for (size_t i = 0; i < pairs.size(); ++i) by_rank[i] = (uint32_t)i;
        std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t x, uint32_t y) { return pairs[x].rank < pairs[y].rank; });
        for (uint32_t i : by_rank) {
            const Pair& p = pairs[i];
            GzPair8& h = T.pair_hot[gz_pair_ha(p.a, p.b, T.pair_ph.k1, T.pair_ph.k2) >> GZ_PAIR_HOT_SHIFT];
            if (h.lo == 0xFFFFFFFFu) h = T.pair8[slot_of[i]];
        }
      } catch (...) { rc_p = GZ_E_NOMEM; }
    };
    std::thread pair_thread;                                    // (reads pairs and T.merges, writes T.pair_*: nothing the code below touches)
    start(pair_thread, pair_side);
    Joiner pair_join{pair_thread};
    if (vocab_thread.joinable()) vocab_thread.join();            // ---- the vocabulary is needed from here on
    if (rc_v) { err = err_v; return rc_v; }
    const int32_t unk_id = T.special_ids[4];
    // single-character forms the vocab knows: token "c" is the final piece of symbol c+"</w>", token "c@@" is
    // the non-final piece of symbol c
    for (const std::string& w : enc.keys) {
        if (w.empty()) continue;
        if (count_cps(w) == 1) { m.assign(w).append("</w>"); intern(m.data(), m.size()); }
        if (ends_with(w, "@@") && w.size() > 2) {
            m.assign(w, 0, w.size() - 2);
            if (count_cps(m) == 1) intern(m.data(), m.size());
        }
    }
    if (T.symbols.size() > GZ_MAX_SYMBOLS) { err = "too many distinct symbols"; return GZ_E_LIMIT; }

    // ---- symbol -> vocab ids ----------------------------------------------------------------------------------
    T.sym_ids.resize(T.symbols.size());
    std::string probe;
    for (size_t s = 0; s < T.symbols.size(); ++s) {
        const std::string& str = T.symbols[s];
        probe.assign(str).append("@@");
        const int32_t* nf = enc.get(probe);
        int32_t fin = unk_id;
        if (ends_with(str, "</w>")) {
            probe.assign(str, 0, str.size() - 4);
            const int32_t* f = enc.get(probe);
            if (f) fin = *f;
        }
        T.sym_ids[s] = GzSymIds{nf ? *nf : unk_id, fin};
    }

    // ---- code point -> initial symbol ---------------------------------------------------------------------------
    T.bmp.assign(65536, GzCpSyms{GZ_NO_SYMBOL, GZ_NO_SYMBOL});
    std::vector<GzAstral> astral_list;
    auto astral_slot = [&](uint32_t cp) -> GzAstral& {
        for (auto& e : astral_list) if (e.cp == cp) return e;
        astral_list.push_back(GzAstral{cp, GZ_NO_SYMBOL, GZ_NO_SYMBOL, 0});
        return astral_list.back();
    };
    for (size_t s = 0; s < T.symbols.size(); ++s) {
        const std::string& str = T.symbols[s];
        if (str.empty()) continue;
        size_t ncp = count_cps(str);
        bool fin = false;
        if (ncp == 1) fin = false;
        else if (ncp == 5 && ends_with(str, "</w>")) fin = true;
        else continue;
        uint32_t cp = first_cp(str);
        if (cp < 0x10000) { if (fin) T.bmp[cp].final_ = (uint32_t)s; else T.bmp[cp].plain = (uint32_t)s; }
        else { GzAstral& e = astral_slot(cp); if (fin) e.final_ = (uint32_t)s; else e.plain = (uint32_t)s; }
    }
    if (!astral_list.empty()) {
        size_t slots = 4;
        while (slots < 2 * astral_list.size()) slots <<= 1;
        T.astral.assign(slots, GzAstral{GZ_NO_SYMBOL, GZ_NO_SYMBOL, GZ_NO_SYMBOL, 0});
        for (const GzAstral& e : astral_list) {
            size_t h = gz_cp_hash(e.cp) & (slots - 1);
            while (T.astral[h].cp != GZ_NO_SYMBOL) h = (h + 1) & (slots - 1);
            T.astral[h] = e;
        }
    }

    if (pair_thread.joinable()) pair_thread.join();
    if (rc_p) { err = "out of memory while hashing the pair table"; return rc_p; }
    return GZ_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Hash and displace.  Buckets are placed largest first; a bucket tries d = 0, 1, 2, ... until every one of its keys
// lands on a free slot.  Tables stay at load <= 0.8 and buckets at about 8 keys on average (16 bits of displacement are
// plenty: the bundled pair table needs d < 1000); a failed attempt is retried with other multipliers and then with
// more buckets.  Keys of buckets that still cannot be placed (two keys with the same hb in one bucket under every
// seed: adversarial tables only) are inserted by linear probing and their bucket is marked GZ_PH_OVERFLOW.
// ---------------------------------------------------------------------------------------------------------------------
void gz_ph_build(size_t n, void (*hashes)(const void* ctx, size_t i, uint32_t k1, uint32_t k2, uint32_t* ha, uint32_t* hb), const void* ctx,
                 GzPhHost& out, std::vector<uint32_t>& slot_of, const uint8_t* hot, uint32_t hot_slots)
{
    size_t slots = 16;
    while (slots * 4 < n * 5) slots <<= 1;                         // load <= 0.8
    uint32_t sshift = 32;
    while ((size_t(1) << (32 - sshift)) < slots) --sshift;
    size_t nb0 = 16;
    while (nb0 * 8 < n) nb0 <<= 1;
    const int force = gz_default_options().ph_force_overflow;      // (tests: every force-th bucket is refused)
    const uint32_t hzone = (hot && hot_slots < slots / 4) ? hot_slots : 0u;        // (a region that is a large part of the table steers nothing)
    GzPhHost best;
    std::vector<uint32_t> best_slots;
    bool have = false;
    for (int attempt = 0; attempt < 12; ++attempt) {
        // attempts 0..3: nb0 buckets with four seeds, 4..7: twice as many, 8..11: four times as many
        size_t nb = nb0 << (attempt / 4);
        if (nb > slots) nb = slots;
        uint32_t bshift = 32;
        while ((size_t(1) << (32 - bshift)) < nb) --bshift;
        const uint32_t k1 = 0x9E3779B1u + 0x3C6EF372u * (uint32_t)attempt, k2 = 0x85EBCA6Bu + 0x1B873592u * (uint32_t)attempt;   // odd
        std::vector<uint32_t> cnt(nb + 1, 0), bucket_of(n), hb(n);
        for (size_t i = 0; i < n; ++i) { uint32_t ha; hashes(ctx, i, k1, k2, &ha, &hb[i]); bucket_of[i] = ha >> bshift; ++cnt[bucket_of[i] + 1]; }
        for (size_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
        std::vector<uint32_t> keys(n), fill(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; ++i) keys[fill[bucket_of[i]]++] = (uint32_t)i;
        std::vector<uint32_t> order(nb);
        for (size_t b = 0; b < nb; ++b) order[b] = (uint32_t)b;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cnt[x + 1] - cnt[x] > cnt[y + 1] - cnt[y]; });
        std::vector<uint8_t> used(slots, 0);
        GzPhHost cur;
        cur.disp.assign(nb, 0);
        cur.nbuckets = (uint32_t)nb; cur.bshift = bshift; cur.sshift = sshift; cur.slots = (uint32_t)slots; cur.k1 = k1; cur.k2 = k2;
        std::vector<uint32_t> cur_slots(n, 0), spill;
        std::vector<uint32_t> tmp;
        size_t nonempty = 0;
        for (uint32_t b : order) {
            const uint32_t lo = cnt[b], hi = cnt[b + 1];
            if (lo == hi) break;
            ++nonempty;
            bool placed = false;
            if (!(force && nonempty % (size_t)force == 0)) {
                tmp.resize(hi - lo);
                // the hot region [0, hzone): a hot key must land inside, any other key outside.  A bucket that cannot be placed
                // that way (two hot keys, a full region) is placed with its hot keys treated like the others.
                bool any_hot = false;
                if (hzone) for (uint32_t k = lo; k < hi; ++k) any_hot |= hot[keys[k]] != 0;
                for (int pass = any_hot ? 0 : 1; pass < 2 && !placed; ++pass)
                    for (uint32_t d = 0; d < GZ_PH_OVERFLOW && !placed; ++d) {
                        bool ok = true;
                        for (uint32_t k = lo; k < hi && ok; ++k) {
                            const uint32_t sl = gz_ph_slot(hb[keys[k]], d, sshift);
                            const bool want_in = pass == 0 && hot[keys[k]] != 0;
                            if (used[sl] || (sl < hzone) != want_in) ok = false;
                            else { for (uint32_t j = lo; j < k; ++j) if (tmp[j - lo] == sl) { ok = false; break; } }
                            tmp[k - lo] = sl;
                        }
                        if (ok) {
                            for (uint32_t k = lo; k < hi; ++k) { used[tmp[k - lo]] = 1; cur_slots[keys[k]] = tmp[k - lo]; }
                            cur.disp[b] = (uint16_t)d;
                            placed = true;
                        }
                    }
            }
            if (!placed) { cur.disp[b] = (uint16_t)GZ_PH_OVERFLOW; for (uint32_t k = lo; k < hi; ++k) spill.push_back(keys[k]); }
        }
        for (uint32_t i : spill) {                                    // linear probing from the slot d = GZ_PH_OVERFLOW gives
            uint32_t sl = gz_ph_slot(hb[i], GZ_PH_OVERFLOW, sshift);
            while (used[sl]) sl = (sl + 1) & (uint32_t)(slots - 1);
            used[sl] = 1; cur_slots[i] = sl;
        }
        cur.n_overflow = (uint32_t)spill.size();
        if (!have || cur.n_overflow < best.n_overflow) { best = std::move(cur); best_slots = std::move(cur_slots); have = true; }
        if (best.n_overflow == 0 || force) break;
    }
    out = std::move(best);
    slot_of = std::move(best_slots);
}
