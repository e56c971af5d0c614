/* Byte-scatter helper for corpus.py (bench/test infrastructure, not product).
 * out[dst[i] .. dst[i]+len[i]) = flat[src[i] .. src[i]+len[i])  for every piece i. */
#include <stdint.h>
#include <string.h>

void corpus_fill(uint8_t *out, const int64_t *dst, const uint8_t *flat,
                 const int64_t *src, const int64_t *len, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        memcpy(out + dst[i], flat + src[i], (size_t)len[i]);
}
