/* Host-side string packing for the batch calls of genz_tokenize.Tokenize (encode_batch, decode_batch callers):
 * a list of Python str -> one UTF-8 buffer + int64 offsets[N + 1], in C instead of a Python loop (1 M sentences: about
 * a second in Python, most of encode_batch's time; tens of milliseconds here).
 *
 * This is a CPython extension module (gcc, Python.h), separate from the C-ABI library, which stays free of Python.  It
 * only PACKS: no tokenizer logic lives here.  Encoding is UTF-8 with 'surrogatepass' semantics, as the Python fallback
 * in tokenize.py (`t.encode("utf-8", "surrogatepass")`): a lone surrogate becomes its three-byte form, so the bytes the
 * kernels see are the same either way.  Nothing is cached inside the str objects (no PyUnicode_AsUTF8AndSize).
 *
 *   _gz_pack.pack(seq) -> (bytes text, bytes offsets)      offsets: native int64[N + 1]
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

static PyObject* gz_pack(PyObject* self, PyObject* arg)
{
    (void)self;
    PyObject* seq = PySequence_Fast(arg, "expected a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    PyObject* offs = PyBytes_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(int64_t));
    if (!offs) { Py_DECREF(seq); return NULL; }
    int64_t* off = (int64_t*)PyBytes_AS_STRING(offs);
    int64_t total = 0;
    off[0] = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* s = items[i];
        if (!PyUnicode_Check(s)) {
            Py_DECREF(offs); Py_DECREF(seq);
            PyErr_SetString(PyExc_TypeError, "expected string or bytes-like object");
            return NULL;
        }
        if (PyUnicode_READY(s) < 0) { Py_DECREF(offs); Py_DECREF(seq); return NULL; }
        const Py_ssize_t len = PyUnicode_GET_LENGTH(s);
        total += PyUnicode_IS_ASCII(s) ? len : utf8_size(PyUnicode_KIND(s), PyUnicode_DATA(s), len);
        off[i + 1] = total;
    }
    PyObject* text = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (!text) { Py_DECREF(offs); Py_DECREF(seq); return NULL; }
    unsigned char* o = (unsigned char*)PyBytes_AS_STRING(text);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* s = items[i];
        const Py_ssize_t len = PyUnicode_GET_LENGTH(s);
        const int kind = PyUnicode_KIND(s);
        const void* data = PyUnicode_DATA(s);
        if (PyUnicode_IS_ASCII(s)) { memcpy(o, data, (size_t)len); o += len; }
        else if (kind == PyUnicode_1BYTE_KIND) { const Py_UCS1* p = (const Py_UCS1*)data; for (Py_ssize_t k = 0; k < len; ++k) o = put_cp(o, p[k]); }
        else if (kind == PyUnicode_2BYTE_KIND) { const Py_UCS2* p = (const Py_UCS2*)data; for (Py_ssize_t k = 0; k < len; ++k) o = put_cp(o, p[k]); }
        else { const Py_UCS4* p = (const Py_UCS4*)data; for (Py_ssize_t k = 0; k < len; ++k) o = put_cp(o, p[k]); }
    }
    Py_DECREF(seq);
    PyObject* r = PyTuple_Pack(2, text, offs);
    Py_DECREF(text); Py_DECREF(offs);
    return r;
}

static PyMethodDef methods[] = {
    {"pack", gz_pack, METH_O, "pack(seq of str) -> (utf-8 bytes, int64 offsets[N + 1] as bytes); 'surrogatepass' encoding"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gz_pack", "string packing for genz_tokenize batch calls", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__gz_pack(void) { return PyModule_Create(&moddef); }
