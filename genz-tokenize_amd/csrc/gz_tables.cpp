// K0 -- host table builder.
//
// Replaces Tokenize.__init__ / add_vocab_file / add_bpe_file (reference genz_tokenize/tokenize.py:31-57):
// parses the RAW BYTES of vocab.txt and bpe.codes with the reference's exact text-mode semantics (loader rules
// L1-L8 of SURVEY.md §8(a)) and then re-expresses the two Python dicts as integer tables for the GPU:
//
//   symbols   every string bpe() can hold in its `word` tuple: merge operands, merge results, and the
//             single-character forms the vocab knows (c  and  c+"</w>").  Interned by string identity.
//   pair_tab  open-addressing hash  (left symbol, right symbol) -> rank, merged symbol   [bpe_ranks.get(pair)]
//   merges    rank -> (left, right, merged symbol)                              [first + second, tokenize.py:88]
//   sym_ids   symbol -> vocab id when emitted as a non-final piece (string + "@@") and as the final piece
//             (string minus "</w>")                                              [encoder.get(tok, unk), :120-121]
//   bmp / astral   code point -> initial symbol (plain, and with "</w>")         [tuple(token), :63-64]
#include "gz_common.h"
#include "../../include/genz_tokenize.h"

#include <algorithm>
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

struct OrderedDict {            // a Python dict[str, int]: insertion order, overwrite keeps position
    std::unordered_map<std::string, size_t> pos;
    std::vector<std::string> keys;
    std::vector<int32_t> vals;
    void set(const std::string& k, int32_t v)
    {
        auto it = pos.find(k);
        if (it == pos.end()) { pos.emplace(k, keys.size()); keys.push_back(k); vals.push_back(v); }
        else vals[it->second] = v;
    }
    const int32_t* get(const std::string& k) const
    {
        auto it = pos.find(k);
        return it == pos.end() ? nullptr : &vals[it->second];
    }
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
    U32 vtext, btext;
    if (!decode_utf8_strict(vocab, vocab_len, vtext)) { err = "vocab file: invalid UTF-8"; return GZ_E_UTF8; }
    if (!decode_utf8_strict(bpe, bpe_len, btext)) { err = "bpe file: invalid UTF-8"; return GZ_E_UTF8; }
    universal_newlines(vtext);
    universal_newlines(btext);

    // ---- encoder (tokenize.py:31-37, :44-51) ------------------------------------------------------------
    OrderedDict enc;
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
            enc.set(word, (int32_t)enc.size());           // len(encoder) BEFORE the insertion
            i = (j < n) ? j + 1 : j;
        }
    }
    T.enc_words = enc.keys;
    T.enc_ids = enc.vals;
    for (int i = 0; i < 5; ++i) T.special_ids[i] = *enc.get(specials[i]);
    const int32_t unk_id = T.special_ids[4];

    // ---- bpe_ranks (tokenize.py:53-57) --------------------------------------------------------------------
    OrderedDict ranks;                                    // key = fields joined by '\n'
    std::unordered_map<std::string, int32_t> nfields;
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
        if (rows.size() > GZ_MAX_RANKS) { err = "bpe file: too many lines"; return GZ_E_LIMIT; }
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
            ranks.set(key, (int32_t)r);
            nfields[key] = nf;
        }
        T.rank_keys = ranks.keys;
        T.rank_vals = ranks.vals;
        T.rank_nfields.resize(ranks.size());
        for (size_t k = 0; k < ranks.size(); ++k) T.rank_nfields[k] = nfields[ranks.keys[k]];
        T.merges.assign(rows.size(), GzMergeInfo{0, 0, 0, 0});
    }

    // ---- symbols --------------------------------------------------------------------------------------------
    std::unordered_map<std::string, uint32_t> sym_of;
    auto intern = [&](const std::string& s) -> uint32_t {
        auto it = sym_of.find(s);
        if (it != sym_of.end()) return it->second;
        uint32_t id = (uint32_t)T.symbols.size();
        sym_of.emplace(s, id);
        T.symbols.push_back(s);
        return id;
    };
    struct Pair { uint32_t a, b, rank; };
    std::vector<Pair> pairs;
    for (size_t k = 0; k < ranks.size(); ++k) {
        if (T.rank_nfields[k] != 2) continue;             // such a key can never equal a (first, second) pair
        const std::string& key = ranks.keys[k];
        size_t nl = key.find('\n');
        std::string a = key.substr(0, nl), b = key.substr(nl + 1);
        uint32_t ia = intern(a), ib = intern(b), im = intern(a + b);
        uint32_t r = (uint32_t)ranks.vals[k];
        T.merges[r] = GzMergeInfo{ia, ib, im, 0};
        pairs.push_back({ia, ib, r});
    }
    // single-character forms the vocab knows: token "c" is the final piece of symbol c+"</w>", token "c@@" is
    // the non-final piece of symbol c
    for (const std::string& w : enc.keys) {
        if (w.empty()) continue;
        if (count_cps(w) == 1) intern(w + "</w>");
        if (ends_with(w, "@@") && w.size() > 2) {
            std::string body = w.substr(0, w.size() - 2);
            if (count_cps(body) == 1) intern(body);
        }
    }
    if (T.symbols.size() > GZ_MAX_SYMBOLS) { err = "too many distinct symbols"; return GZ_E_LIMIT; }

    // ---- symbol -> vocab ids ----------------------------------------------------------------------------------
    T.sym_ids.resize(T.symbols.size());
    for (size_t s = 0; s < T.symbols.size(); ++s) {
        const std::string& str = T.symbols[s];
        const int32_t* nf = enc.get(str + "@@");
        int32_t fin = unk_id;
        if (ends_with(str, "</w>")) {
            const int32_t* f = enc.get(str.substr(0, str.size() - 4));
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

    // ---- pair hash -------------------------------------------------------------------------------------------------
    {
        size_t slots = 16;
        while (slots < gz_tab_slack() * pairs.size()) slots <<= 1;
        uint32_t shift = 32;
        while ((size_t(1) << (32 - shift)) < slots) --shift;
        T.pair_tab.assign(slots, GzPairSlot{GZ_PAIR_EMPTY, 0, 0, 0});
        uint32_t worst = 0;
        for (const Pair& p : pairs) {
            size_t h = gz_pair_slot(p.a, p.b, shift);
            uint32_t probes = 1;
            while (T.pair_tab[h].left != GZ_PAIR_EMPTY) { h = (h + 1) & (slots - 1); ++probes; }
            T.pair_tab[h] = GzPairSlot{p.a, p.b, T.merges[p.rank].merged, p.rank};
            worst = std::max(worst, probes);
        }
        T.max_probe = worst;
    }
    return GZ_OK;
}
