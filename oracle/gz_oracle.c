/* gz_oracle.c -- plain-C restatement of the genz_tokenize `Tokenize.__call__` / `encode` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (like oracle/gz_oracle.py): it may be loaded by tests/, by __graft_entry__.smoke() and by
 * bench.py's cpu_baseline leg, never by the product (genz-tokenize_amd/).  It shares no code and no data structure
 * with the product: symbols are byte-string slices of the word, tables are string-keyed hash maps, the merge loop is
 * the naive O(n^2) one.  Parity status: PINNED -- tests/test_oracle_c.py checks it against every golden vector the
 * reference produced (tests/golden/, see make_golden.py) and against oracle/gz_oracle.py.
 *
 * Each function cites the reference lines (genz_tokenize/tokenize.py) it follows.
 *
 * Build: make -C oracle   ->  oracle/libgz_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ string map */
typedef struct {
    char **keys;      /* owned copies, NULL = empty */
    int32_t *klen;
    int32_t *vals;
    size_t cap, n;
} Map;

static uint64_t fnv(const char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}

static void map_init(Map *m, size_t cap)
{
    m->cap = 16;
    while (m->cap < cap * 2) m->cap <<= 1;
    m->keys = (char **)calloc(m->cap, sizeof(char *));
    m->klen = (int32_t *)calloc(m->cap, sizeof(int32_t));
    m->vals = (int32_t *)calloc(m->cap, sizeof(int32_t));
    m->n = 0;
}

static void map_free(Map *m)
{
    for (size_t i = 0; i < m->cap; ++i) free(m->keys[i]);
    free(m->keys); free(m->klen); free(m->vals);
}

static void map_grow(Map *m);

/* dict[key] = val; returns 1 when the key is new */
static int map_set(Map *m, const char *k, size_t n, int32_t v)
{
    if ((m->n + 1) * 2 > m->cap) map_grow(m);
    size_t h = fnv(k, n) & (m->cap - 1);
    while (m->keys[h]) {
        if ((size_t)m->klen[h] == n && memcmp(m->keys[h], k, n) == 0) { m->vals[h] = v; return 0; }
        h = (h + 1) & (m->cap - 1);
    }
    m->keys[h] = (char *)malloc(n + 1);
    memcpy(m->keys[h], k, n);
    m->keys[h][n] = 0;
    m->klen[h] = (int32_t)n;
    m->vals[h] = v;
    m->n++;
    return 1;
}

static void map_grow(Map *m)
{
    Map o = *m;
    map_init(m, o.cap);
    for (size_t i = 0; i < o.cap; ++i)
        if (o.keys[i]) { map_set(m, o.keys[i], (size_t)o.klen[i], o.vals[i]); free(o.keys[i]); }
    free(o.keys); free(o.klen); free(o.vals);
}

static const int32_t *map_get(const Map *m, const char *k, size_t n)
{
    size_t h = fnv(k, n) & (m->cap - 1);
    while (m->keys[h]) {
        if ((size_t)m->klen[h] == n && memcmp(m->keys[h], k, n) == 0) return &m->vals[h];
        h = (h + 1) & (m->cap - 1);
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------------ text helpers */
static int is_space(uint32_t c)      /* the 29 code points of str.isspace / regex \s */
{
    return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

/* next code point of structurally valid UTF-8 (Python str encoded with 'surrogatepass'); returns its byte length */
static int next_cp(const uint8_t *p, size_t n, uint32_t *cp)
{
    uint8_t b = p[0];
    int len = b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
    if ((size_t)len > n) len = (int)n;
    uint32_t c = len == 1 ? b : len == 2 ? (b & 0x1Fu) : len == 3 ? (b & 0x0Fu) : (b & 0x07u);
    for (int k = 1; k < len; ++k) c = (c << 6) | (p[k] & 0x3Fu);
    *cp = c;
    return len;
}

/* codecs 'utf-8' strict: 0 = ok */
static int utf8_strict(const uint8_t *p, size_t n)
{
    size_t i = 0;
    while (i < n) {
        uint8_t b = p[i];
        if (b < 0x80) { ++i; continue; }
        int len; uint32_t cp;
        if (b >= 0xC2 && b <= 0xDF) { len = 2; cp = b & 0x1F; }
        else if (b >= 0xE0 && b <= 0xEF) { len = 3; cp = b & 0x0F; }
        else if (b >= 0xF0 && b <= 0xF4) { len = 4; cp = b & 0x07; }
        else return -1;
        if (i + (size_t)len > n) return -1;
        for (int k = 1; k < len; ++k) { if ((p[i + k] & 0xC0) != 0x80) return -1; cp = (cp << 6) | (p[i + k] & 0x3F); }
        if (len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return -1;
        if (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) return -1;
        i += (size_t)len;
    }
    return 0;
}

/* open(..., 'r') universal newlines, in place; returns the new length */
static size_t universal_newlines(uint8_t *s, size_t n)
{
    size_t w = 0;
    for (size_t r = 0; r < n; ++r) {
        uint8_t c = s[r];
        if (c == '\r') { if (r + 1 < n && s[r + 1] == '\n') ++r; c = '\n'; }
        s[w++] = c;
    }
    return w;
}

/* ------------------------------------------------------------------------------------------------ the tokenizer */
typedef struct {
    Map enc;          /* encoder: token string -> id                         (tokenize.py:31-51) */
    Map ranks;        /* bpe_ranks: "first\0second" -> rank (2-field lines)   (tokenize.py:53-57) */
    int32_t pad, bos, eos, unk;
    int32_t n_rank_keys;
} Oracle;

/* strip(): leading / trailing whitespace code points of [a, b) */
static void strip_ws(const uint8_t *s, size_t *a, size_t *b)
{
    while (*a < *b) {
        uint32_t c; int l = next_cp(s + *a, *b - *a, &c);
        if (!is_space(c)) break;
        *a += (size_t)l;
    }
    while (*b > *a) {
        size_t q = *b - 1;
        while (q > *a && (s[q] & 0xC0) == 0x80) --q;          /* back to the lead byte */
        uint32_t c; next_cp(s + q, *b - q, &c);
        if (!is_space(c)) break;
        *b = q;
    }
}

void gzo_destroy(Oracle *o)
{
    if (!o) return;
    map_free(&o->enc); map_free(&o->ranks);
    free(o);
}

/* returns NULL and *err = 1 on invalid UTF-8 (the reference raises UnicodeDecodeError) */
Oracle *gzo_create(const uint8_t *vocab, size_t vlen, const uint8_t *bpe, size_t blen, const char *const specials[5], int *err)
{
    *err = 0;
    if (utf8_strict(vocab, vlen) || utf8_strict(bpe, blen)) { *err = 1; return NULL; }
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    map_init(&o->enc, 1 << 16);
    map_init(&o->ranks, 1 << 16);
    for (int i = 0; i < 5; ++i) map_set(&o->enc, specials[i], strlen(specials[i]), i);     /* :31-37 */

    uint8_t *v = (uint8_t *)malloc(vlen + 1);
    memcpy(v, vocab, vlen);
    size_t n = universal_newlines(v, vlen);
    size_t i = 0;
    while (i < n) {                                                 /* readlines(): :45-51 */
        size_t j = i;
        while (j < n && v[j] != '\n') ++j;
        size_t a = i, b = j;
        strip_ws(v, &a, &b);
        size_t k = b;
        while (k > a && v[k - 1] != ' ') --k;                        /* rfind(' ') */
        size_t cut;
        if (k > a) cut = k - 1;
        else if (b > a) { cut = b - 1; while (cut > a && (v[cut] & 0xC0) == 0x80) --cut; }   /* line[:-1]: drop one CHARACTER */
        else cut = a;
        map_set(&o->enc, (const char *)v + a, cut - a, (int32_t)o->enc.n);                  /* len(encoder) before the insert */
        i = j < n ? j + 1 : j;
    }
    free(v);

    uint8_t *m = (uint8_t *)malloc(blen + 1);
    memcpy(m, bpe, blen);
    n = universal_newlines(m, blen);
    /* read().split('\n')[:-1] : every line that is followed by a '\n' */
    int32_t rank = 0;
    i = 0;
    char *key = (char *)malloc(n + 2);
    Map allkeys; map_init(&allkeys, 1 << 16);                       /* distinct tuples, for len(bpe_ranks) */
    while (i < n) {
        size_t j = i;
        while (j < n && m[j] != '\n') ++j;
        if (j >= n) break;                                           /* the last element is dropped */
        size_t p = i, klen = 0;
        int nf = 0;
        while (p < j) {                                              /* str.split() */
            uint32_t c; int l = next_cp(m + p, j - p, &c);
            if (is_space(c)) { p += (size_t)l; continue; }
            size_t q = p;
            while (q < j) { l = next_cp(m + q, j - q, &c); if (is_space(c)) break; q += (size_t)l; }
            if (nf) key[klen++] = 0;
            memcpy(key + klen, m + p, q - p); klen += q - p;
            ++nf;
            p = q;
        }
        /* tag the key with its field count so that ("a",) and ("a","") style ambiguities cannot collide */
        key[klen] = (char)nf;
        map_set(&allkeys, key, klen + 1, rank);
        if (nf == 2) map_set(&o->ranks, key, klen, rank);            /* later duplicates overwrite (dict(zip(...))) */
        ++rank;
        i = j + 1;
    }
    o->n_rank_keys = (int32_t)allkeys.n;
    map_free(&allkeys);
    free(key); free(m);
    o->pad = *map_get(&o->enc, specials[0], strlen(specials[0]));    /* looked up at call time in the reference */
    o->bos = *map_get(&o->enc, specials[1], strlen(specials[1]));
    o->eos = *map_get(&o->enc, specials[2], strlen(specials[2]));
    o->unk = *map_get(&o->enc, specials[4], strlen(specials[4]));
    return o;
}

int32_t gzo_vocab_size(const Oracle *o) { return (int32_t)o->enc.n; }
int32_t gzo_n_ranks(const Oracle *o) { return o->n_rank_keys; }
void gzo_special_ids(const Oracle *o, int32_t out[4]) { out[0] = o->pad; out[1] = o->bos; out[2] = o->eos; out[3] = o->unk; }
int32_t gzo_lookup(const Oracle *o, const uint8_t *tok, size_t n) { const int32_t *p = map_get(&o->enc, (const char *)tok, n); return p ? *p : -1; }

typedef struct { int32_t *v; size_t n, cap; } Vec;
static void push(Vec *x, int32_t v)
{
    if (x->n == x->cap) { x->cap = x->cap ? x->cap * 2 : 256; x->v = (int32_t *)realloc(x->v, x->cap * sizeof(int32_t)); }
    x->v[x->n++] = v;
}

/* bpe() + vocab lookup of one word (tokenize.py:62-101, :110, :120-121).  `w` = the word's bytes (with the glued
 * '\n' if any).  Symbols are slices [s[k], s[k+1]) of  w + "</w>". */
static void bpe_word(const Oracle *o, const uint8_t *w, size_t wn, Vec *ids, char *buf, size_t *bounds, char *key)
{
    memcpy(buf, w, wn);
    memcpy(buf + wn, "</w>", 4);
    size_t ns = 0, i = 0;
    while (i < wn) { uint32_t c; bounds[ns++] = i; i += (size_t)next_cp(w + i, wn - i, &c); }
    bounds[ns] = wn + 4;                                             /* the last symbol carries "</w>" (:64) */
    if (ns == 1) {                                                   /* :66-67: the word itself */
        const int32_t *p = map_get(&o->enc, (const char *)w, wn);
        push(ids, p ? *p : o->unk);
        return;
    }
    while (ns > 1) {
        int32_t best = -1; size_t bi = 0;
        for (size_t k = 0; k + 1 < ns; ++k) {                        /* min(pairs, key=rank) (:70-71) */
            size_t la = bounds[k + 1] - bounds[k], lb = bounds[k + 2] - bounds[k + 1];
            memcpy(key, buf + bounds[k], la); key[la] = 0; memcpy(key + la + 1, buf + bounds[k + 1], lb);
            const int32_t *r = map_get(&o->ranks, key, la + 1 + lb);
            if (r && (best < 0 || *r < best)) { best = *r; bi = k; }
        }
        if (best < 0) break;                                         /* :72-73 */
        const size_t fa = bounds[bi], fl = bounds[bi + 1] - fa, sa = bounds[bi + 1], sl = bounds[bi + 2] - sa;
        size_t out = 0, k = 0;                                       /* merge every occurrence left to right (:75-92) */
        while (k < ns) {
            if (k + 1 < ns && bounds[k + 1] - bounds[k] == fl && memcmp(buf + bounds[k], buf + fa, fl) == 0 &&
                bounds[k + 2] - bounds[k + 1] == sl && memcmp(buf + bounds[k + 1], buf + sa, sl) == 0) {
                bounds[out++] = bounds[k]; k += 2;
            } else { bounds[out++] = bounds[k]; k += 1; }
        }
        bounds[out] = wn + 4;
        ns = out;
        /* NB: bounds[] was compacted in place left to right; the slices of the chosen pair are read before any
         * write can reach them only if they sit at or after the write cursor -- which holds because out <= k. */
    }
    for (size_t k = 0; k < ns; ++k) {                                /* "@@ ".join(word)[:-4].split(" ") (:99-100, :110) */
        size_t a = bounds[k], l = bounds[k + 1] - a;
        const int32_t *p;
        if (k + 1 < ns) { memcpy(key, buf + a, l); memcpy(key + l, "@@", 2); p = map_get(&o->enc, key, l + 2); }
        else p = map_get(&o->enc, buf + a, l - 4);
        push(ids, p ? *p : o->unk);
    }
}

/* __tokenize + ids of one text (tokenize.py:103-133); appends to ids */
static void encode_text(const Oracle *o, const uint8_t *t, size_t n, Vec *ids, char **buf, size_t **bounds, char **key, size_t *cap)
{
    size_t i = 0;
    while (i < n) {
        uint32_t c; int l = next_cp(t + i, n - i, &c);
        if (is_space(c)) { i += (size_t)l; continue; }
        size_t j = i;
        while (j < n) { l = next_cp(t + j, n - j, &c); if (is_space(c)) break; j += (size_t)l; }
        if (j < n && t[j] == '\n') ++j;                              /* \S+\n? (:106) */
        const size_t wn = j - i;
        if (wn + 16 > *cap) {
            *cap = (wn + 16) * 2;
            *buf = (char *)realloc(*buf, *cap); *key = (char *)realloc(*key, *cap * 2 + 8);
            *bounds = (size_t *)realloc(*bounds, (*cap + 2) * sizeof(size_t));
        }
        bpe_word(o, t + i, wn, ids, *buf, *bounds, *key);
        i = j;
    }
}

static int cut_len(int n, int max_len) { int stop = max_len - 1; if (stop >= 0) return n < stop ? n : stop; int k = n + stop; return k > 0 ? k : 0; }

/* Tokenize.__call__ (tokenize.py:184-259) over a batch; same array conventions as include/genz_tokenize.h, except
 * that rows are always written ragged: row i occupies [row_off[i], row_off[i+1]).  flags: 1 padding, 2 truncation,
 * 4 max_len is None.  Returns 0, or -5 when `capacity` is too small (row_off is still filled in). */
int gzo_call_batch(const Oracle *o, const uint8_t *text, const int64_t *toff, const uint8_t *pair, const int64_t *poff,
                   int64_t n_docs, int32_t max_len, uint32_t flags, int64_t capacity,
                   int32_t *ids_out, int32_t *mask_out, int32_t *tt_out, int32_t *seq_out,
                   int64_t *row_off, int32_t *pair_len, int32_t *status)
{
    const int pad_mode = !(flags & 4u) && (flags & 1u), trunc = (flags & 2u) != 0;
    Vec v = {0, 0, 0};
    size_t cap = 256;
    char *buf = (char *)malloc(cap), *key = (char *)malloc(cap * 2 + 8);
    size_t *bounds = (size_t *)malloc((cap + 2) * sizeof(size_t));
    int64_t at = 0;
    int overflow = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        v.n = 0;
        push(&v, o->bos);                                            /* :134-135 */
        encode_text(o, text + toff[d], (size_t)(toff[d + 1] - toff[d]), &v, &buf, &bounds, &key, &cap);
        push(&v, o->eos);
        if (pair) {                                                  /* :229-230, :237-239 */
            push(&v, o->eos);
            encode_text(o, pair + poff[d], (size_t)(poff[d + 1] - poff[d]), &v, &buf, &bounds, &key, &cap);
            push(&v, o->eos);
        }
        int n = (int)v.n;
        if (pad_mode) {                                              /* __padding (:141-146) */
            if (n < max_len) { while ((int)v.n < max_len) push(&v, o->pad); n = max_len; }
            else if (trunc) { n = cut_len(n, max_len); v.n = (size_t)n; push(&v, o->eos); n += 1; }
        }
        row_off[d] = at;
        if (at + n > capacity) overflow = 1;
        if (!overflow) {
            for (int i = 0; i < n; ++i) { ids_out[at + i] = v.v[i]; mask_out[at + i] = v.v[i] != o->pad; }   /* :148-152 */
        }
        if (pair && !overflow) {
            /* get_sequence_id (:163-182) */
            int32_t *s = seq_out + at;
            int sl = 0;
            for (int i = 0; i < n; ++i) {
                if (v.v[i] == o->eos) { s[sl++] = -1; break; }
                s[sl++] = v.v[i] == o->bos ? -1 : 0;
            }
            for (int i = sl; i < n; ++i) {
                if (v.v[i] == o->eos) { s[sl++] = -1; if (s[i - 1] == 1) break; }
                else s[sl++] = 1;
            }
            /* get_token_type (:154-161) */
            s[0] = 0; s[sl - 1] = 1;
            int found = 0;
            for (int i = 0; i < sl && found < 2; ++i) if (s[i] == -1) { s[i] = found; ++found; }
            status[d] = found < 2;                                   /* list.index(None) raises ValueError */
            int tl = sl;
            int32_t *tt = tt_out + at;
            for (int i = 0; i < sl; ++i) tt[i] = s[i];
            if (pad_mode) {                                          /* :256-258 */
                if (sl < max_len) { for (int i = sl; i < max_len; ++i) tt[i] = o->pad; tl = max_len; }
                else if (trunc) { tl = cut_len(sl, max_len); tt[tl] = o->eos; tl += 1; }
            }
            pair_len[2 * d] = status[d] ? 0 : sl;
            pair_len[2 * d + 1] = status[d] ? 0 : tl;
        } else if (status) status[d] = 0;
        at += n;
    }
    row_off[n_docs] = at;
    free(v.v); free(buf); free(key); free(bounds);
    return overflow ? -5 : 0;
}
