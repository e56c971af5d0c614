/* Host-side string packing for the batch calls of genz_tokenize.Tokenize (encode_batch, decode_batch callers):
 * a list of Python str -> one UTF-8 buffer + int64 offsets[N + 1], in C instead of a Python loop (1 M sentences: about
 * a second in Python, most of encode_batch's time; tens of milliseconds here: the sizes and the encoding run on worker
 * threads with the GIL released, over raw buffers collected -- with a reference to every item -- while it was held).
 *
 * This is a CPython extension module (gcc, Python.h), separate from the C-ABI library, which stays free of Python.  It
 * only PACKS: no tokenizer logic lives here.  Encoding is UTF-8 with 'surrogatepass' semantics, as the Python fallback
 * in tokenize.py (`t.encode("utf-8", "surrogatepass")`): a lone surrogate becomes its three-byte form, so the bytes the
 * kernels see are the same either way.  Nothing is cached inside the str objects (no PyUnicode_AsUTF8AndSize).
 *
 *   _gz_pack.pack(seq) -> (bytes text, bytes offsets)      offsets: native int64[N + 1]
 *   _gz_pack.pack_into(seq, out, offsets) -> total bytes    the same into the caller's writable buffers (e.g. a pinned arena: no
 *       fresh allocation of the batch's size, no second copy); offsets must hold N + 1 int64; returns -(bytes needed) and writes
 *       nothing but the offsets when `out` is too small
 *   raises TypeError("expected string or bytes-like object") for a non-str item, like re.findall in the reference
 *   (tokenize.py:106) and _require_str in tokenize.py
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static Py_ssize_t utf8_size(int kind, const void* data, Py_ssize_t n)
{
    Py_ssize_t b = n;
    if (kind == PyUnicode_1BYTE_KIND) {
        const Py_UCS1* p = (const Py_UCS1*)data;
        for (Py_ssize_t i = 0; i < n; ++i) b += p[i] >> 7;
    } else if (kind == PyUnicode_2BYTE_KIND) {
        const Py_UCS2* p = (const Py_UCS2*)data;
        for (Py_ssize_t i = 0; i < n; ++i) b += (p[i] >= 0x80) + (p[i] >= 0x800);
    } else {
        const Py_UCS4* p = (const Py_UCS4*)data;
        for (Py_ssize_t i = 0; i < n; ++i) b += (p[i] >= 0x80) + (p[i] >= 0x800) + (p[i] >= 0x10000);
    }
    return b;
}

static inline unsigned char* put_cp(unsigned char* o, Py_UCS4 c)
{
    if (c < 0x80) { *o++ = (unsigned char)c; }
    else if (c < 0x800) { *o++ = (unsigned char)(0xC0 | (c >> 6)); *o++ = (unsigned char)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) {                                       /* surrogates included: 'surrogatepass' */
        *o++ = (unsigned char)(0xE0 | (c >> 12)); *o++ = (unsigned char)(0x80 | ((c >> 6) & 0x3F)); *o++ = (unsigned char)(0x80 | (c & 0x3F));
    } else {
        *o++ = (unsigned char)(0xF0 | (c >> 18)); *o++ = (unsigned char)(0x80 | ((c >> 12) & 0x3F));
        *o++ = (unsigned char)(0x80 | ((c >> 6) & 0x3F)); *o++ = (unsigned char)(0x80 | (c & 0x3F));
    }
    return o;
}

/* ---- worker threads (the GIL is released while they run: they touch only raw buffers collected under the GIL) ---- */
#include <pthread.h>
#include <stdlib.h>

typedef struct { const void* data; Py_ssize_t len; int kind; int ascii; } Item;
typedef struct {
    const Item* it; int64_t* off; unsigned char* out; Py_ssize_t lo, hi; int phase;     /* 0: sizes into off[i + 1]; 1: encode */
} PackJob;

static void* pack_worker(void* arg)
{
    PackJob* j = (PackJob*)arg;
    if (j->phase == 0) {
        for (Py_ssize_t i = j->lo; i < j->hi; ++i)
            j->off[i + 1] = j->it[i].ascii ? j->it[i].len : utf8_size(j->it[i].kind, j->it[i].data, j->it[i].len);
        return NULL;
    }
    for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
        unsigned char* o = j->out + j->off[i];
        const Item* t = &j->it[i];
        if (t->ascii) memcpy(o, t->data, (size_t)t->len);
        else if (t->kind == PyUnicode_1BYTE_KIND) { const Py_UCS1* p = (const Py_UCS1*)t->data; for (Py_ssize_t k = 0; k < t->len; ++k) o = put_cp(o, p[k]); }
        else if (t->kind == PyUnicode_2BYTE_KIND) { const Py_UCS2* p = (const Py_UCS2*)t->data; for (Py_ssize_t k = 0; k < t->len; ++k) o = put_cp(o, p[k]); }
        else { const Py_UCS4* p = (const Py_UCS4*)t->data; for (Py_ssize_t k = 0; k < t->len; ++k) o = put_cp(o, p[k]); }
    }
    return NULL;
}

#include <unistd.h>
static int n_threads(Py_ssize_t n_items)
{
    /* GZ_PACK_THREADS, else half the online processors, between 4 and 32 (the passes are memory-bound: beyond that nothing is gained) */
    const char* e = getenv("GZ_PACK_THREADS");
    long t = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN) / 2;
    if (!e) { if (t < 4) t = 4; if (t > 32) t = 32; }
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    if (n_items < 20000) t = 1;                                       /* small batches: thread start-up costs more than it saves */
    return (int)t;
}

/* runs `phase` over the items on nt threads (the calling thread takes the first share) */
static void run_jobs(PackJob* jobs, int nt)
{
    pthread_t th[64];
    int started[64];
    for (int k = 1; k < nt; ++k) started[k] = pthread_create(&th[k], NULL, pack_worker, &jobs[k]) == 0;
    pack_worker(&jobs[0]);
    for (int k = 1; k < nt; ++k) { if (started[k]) pthread_join(th[k], NULL); else pack_worker(&jobs[k]); }
}

/* pack_into(seq, out, offsets): see the header of this file */
static PyObject* gz_pack_into(PyObject* self, PyObject* args)
{
    (void)self;
    PyObject* arg;
    Py_buffer outb, offb;
    if (!PyArg_ParseTuple(args, "Ow*w*", &arg, &outb, &offb)) return NULL;
    PyObject* seq = PySequence_Fast(arg, "expected a sequence of str");
    if (!seq) { PyBuffer_Release(&outb); PyBuffer_Release(&offb); return NULL; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    PyObject* result = NULL;
    Item* it = (Item*)malloc((size_t)(n > 0 ? n : 1) * sizeof(Item));
    PyObject** held = (PyObject**)malloc((size_t)(n > 0 ? n : 1) * sizeof(PyObject*));
    Py_ssize_t got = 0;
    if (!it || !held) PyErr_NoMemory();
    else if (offb.len < (n + 1) * (Py_ssize_t)sizeof(int64_t) || ((uintptr_t)offb.buf & 7)) PyErr_SetString(PyExc_ValueError, "pack_into: offsets must hold N + 1 aligned int64");
    else {
        int64_t* off = (int64_t*)offb.buf;
        for (; got < n; ++got) {
            PyObject* s = items[got];
            if (got + 16 < n) __builtin_prefetch(items[got + 16]);      /* (a million str headers scattered over the heap: each one is a cache miss) */
            if (!PyUnicode_Check(s)) { PyErr_SetString(PyExc_TypeError, "expected string or bytes-like object"); break; }
            if (PyUnicode_READY(s) < 0) break;
            Py_INCREF(s);
            held[got] = s;
            it[got].data = PyUnicode_DATA(s); it[got].len = PyUnicode_GET_LENGTH(s); it[got].kind = PyUnicode_KIND(s); it[got].ascii = PyUnicode_IS_ASCII(s);
        }
        if (got == n) {
            const int nt = n_threads(n);
            PackJob jobs[64];
            for (int k = 0; k < nt; ++k) { jobs[k].it = it; jobs[k].off = off; jobs[k].out = NULL; jobs[k].lo = n * k / nt; jobs[k].hi = n * (k + 1) / nt; jobs[k].phase = 0; }
            off[0] = 0;
            int fits;
            Py_BEGIN_ALLOW_THREADS
            run_jobs(jobs, nt);
            for (Py_ssize_t i = 0; i < n; ++i) off[i + 1] += off[i];
            fits = off[n] <= (int64_t)outb.len;
            if (fits) {
                for (int k = 0; k < nt; ++k) { jobs[k].out = (unsigned char*)outb.buf; jobs[k].phase = 1; }
                run_jobs(jobs, nt);
            }
            Py_END_ALLOW_THREADS
            result = PyLong_FromLongLong(fits ? (long long)off[n] : -(long long)off[n]);
        }
    }
    for (Py_ssize_t i = 0; i < got; ++i) { if (i + 16 < got) __builtin_prefetch(held[i + 16], 1); Py_DECREF(held[i]); }
    free(it); free(held);
    Py_DECREF(seq);
    PyBuffer_Release(&outb); PyBuffer_Release(&offb);
    return result;
}

static PyObject* gz_pack(PyObject* self, PyObject* arg)
{
    (void)self;
    PyObject* seq = PySequence_Fast(arg, "expected a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    PyObject* offs = PyBytes_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    Item* it = (Item*)malloc((size_t)(n > 0 ? n : 1) * sizeof(Item));
    PyObject** held = (PyObject**)malloc((size_t)(n > 0 ? n : 1) * sizeof(PyObject*));
    if (!offs || !it || !held) { Py_XDECREF(offs); free(it); free(held); Py_DECREF(seq); return PyErr_NoMemory(); }
    int64_t* off = (int64_t*)PyBytes_AS_STRING(offs);
    /* under the GIL: type check, and a reference + the raw buffer of every item (the sequence may change under us later) */
    Py_ssize_t got = 0;
    for (; got < n; ++got) {
        PyObject* s = items[got];
        if (got + 16 < n) __builtin_prefetch(items[got + 16]);
        if (!PyUnicode_Check(s)) { PyErr_SetString(PyExc_TypeError, "expected string or bytes-like object"); break; }
        if (PyUnicode_READY(s) < 0) break;
        Py_INCREF(s);
        held[got] = s;
        it[got].data = PyUnicode_DATA(s); it[got].len = PyUnicode_GET_LENGTH(s); it[got].kind = PyUnicode_KIND(s); it[got].ascii = PyUnicode_IS_ASCII(s);
    }
    PyObject* text = NULL;
    if (got == n) {
        const int nt = n_threads(n);
        PackJob jobs[64];
        for (int k = 0; k < nt; ++k) { jobs[k].it = it; jobs[k].off = off; jobs[k].out = NULL; jobs[k].lo = n * k / nt; jobs[k].hi = n * (k + 1) / nt; jobs[k].phase = 0; }
        off[0] = 0;
        Py_BEGIN_ALLOW_THREADS
        run_jobs(jobs, nt);
        for (Py_ssize_t i = 0; i < n; ++i) off[i + 1] += off[i];
        Py_END_ALLOW_THREADS
        text = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)off[n]);
        if (text) {
            unsigned char* out = (unsigned char*)PyBytes_AS_STRING(text);
            for (int k = 0; k < nt; ++k) { jobs[k].out = out; jobs[k].phase = 1; }
            Py_BEGIN_ALLOW_THREADS
            run_jobs(jobs, nt);
            Py_END_ALLOW_THREADS
        }
    }
    for (Py_ssize_t i = 0; i < got; ++i) { if (i + 16 < got) __builtin_prefetch(held[i + 16], 1); Py_DECREF(held[i]); }
    free(it); free(held);
    Py_DECREF(seq);
    if (!text) { Py_DECREF(offs); return NULL; }
    PyObject* r = PyTuple_Pack(2, text, offs);
    Py_DECREF(text); Py_DECREF(offs);
    return r;
}

/* expand(tokens buffer, bits, n_real int32 buffer, max_len, pad_id, ids_out buffer, mask_out buffer): CSR rows -> dense [N, max_len]
 * int32 input_ids (padded with the pad id, tokenize.py:141-146) and attention_mask = ids != pad (:148-152), on host threads. */
typedef struct { const void* tok; int bits; const int32_t* nreal; const int64_t* row; int32_t L, pad; int32_t* ids; int32_t* mask; Py_ssize_t lo, hi; } ExpJob;

static void* exp_worker(void* arg)
{
    ExpJob* j = (ExpJob*)arg;
    for (Py_ssize_t r = j->lo; r < j->hi; ++r) {
        int32_t* di = j->ids + (size_t)r * (size_t)j->L;
        int32_t* dm = j->mask + (size_t)r * (size_t)j->L;
        const int32_t t = j->nreal[r] < j->L ? j->nreal[r] : j->L;
        const int64_t o = j->row[r];
        if (j->bits == 16) { const uint16_t* src = (const uint16_t*)j->tok + o; for (int32_t i = 0; i < t; ++i) { di[i] = src[i]; dm[i] = src[i] != (uint32_t)j->pad; } }
        else { const int32_t* src = (const int32_t*)j->tok + o; for (int32_t i = 0; i < t; ++i) { di[i] = src[i]; dm[i] = src[i] != j->pad; } }
        for (int32_t i = t; i < j->L; ++i) { di[i] = j->pad; dm[i] = 0; }
    }
    return NULL;
}

static PyObject* gz_expand(PyObject* self, PyObject* args)
{
    (void)self;
    Py_buffer tok, nr, row, ids, mask;
    int bits, L, pad;
    if (!PyArg_ParseTuple(args, "y*iy*y*iiw*w*", &tok, &bits, &nr, &row, &L, &pad, &ids, &mask)) return NULL;
    const Py_ssize_t n = nr.len / 4;
    int ok = (bits == 16 || bits == 32) && L > 0 && row.len == (n + 1) * 8 && ids.len == (Py_ssize_t)n * L * 4 && mask.len == ids.len;
    if (ok && n > 0) {
        /* the arrays index each other: offsets non-decreasing from 0 and inside the token buffer, every row's count >= 0 and
         * inside its own span (an exported function: its arguments are checked, not trusted) */
        const int64_t* ro = (const int64_t*)row.buf;
        const int32_t* nrp = (const int32_t*)nr.buf;
        ok = ro[0] == 0 && ro[n] >= 0 && ro[n] <= tok.len / (bits / 8);
        for (Py_ssize_t r = 0; ok && r < n; ++r)
            ok = ro[r] <= ro[r + 1] && nrp[r] >= 0 && (int64_t)(nrp[r] < L ? nrp[r] : L) <= ro[r + 1] - ro[r];
    }
    if (ok) {
        const int nt = n_threads(n);
        ExpJob jobs[64];
        pthread_t th[64];
        int started[64];
        for (int k = 0; k < nt; ++k) {
            ExpJob* j = &jobs[k];
            j->tok = tok.buf; j->bits = bits; j->nreal = (const int32_t*)nr.buf; j->row = (const int64_t*)row.buf; j->L = L; j->pad = pad;
            j->ids = (int32_t*)ids.buf; j->mask = (int32_t*)mask.buf; j->lo = n * k / nt; j->hi = n * (k + 1) / nt;
        }
        Py_BEGIN_ALLOW_THREADS
        for (int k = 1; k < nt; ++k) started[k] = pthread_create(&th[k], NULL, exp_worker, &jobs[k]) == 0;
        exp_worker(&jobs[0]);
        for (int k = 1; k < nt; ++k) { if (started[k]) pthread_join(th[k], NULL); else exp_worker(&jobs[k]); }
        Py_END_ALLOW_THREADS
    }
    PyBuffer_Release(&tok); PyBuffer_Release(&nr); PyBuffer_Release(&row); PyBuffer_Release(&ids); PyBuffer_Release(&mask);
    if (!ok) { PyErr_SetString(PyExc_ValueError, "expand: inconsistent buffers"); return NULL; }
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"pack", gz_pack, METH_O, "pack(seq of str) -> (utf-8 bytes, int64 offsets[N + 1] as bytes); 'surrogatepass' encoding; GZ_PACK_THREADS threads"},
    {"pack_into", gz_pack_into, METH_VARARGS, "pack_into(seq of str, out, offsets) -> bytes written, or -(bytes needed) when out is too small"},
    {"expand", gz_expand, METH_VARARGS, "expand(tokens, bits, n_real, row_off, max_len, pad_id, ids_out, mask_out): CSR rows -> dense [N, max_len] int32"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gz_pack", "string packing for genz_tokenize batch calls", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__gz_pack(void) { return PyModule_Create(&moddef); }
