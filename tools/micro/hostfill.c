/* How fast can T threads FIRST-TOUCH a fresh anonymous mapping?  (What a large host call's output arrays -- np.empty, malloc:
 * untouched pages -- cost before a single result byte is in them.)  Variants: plain stores; MADV_HUGEPAGE on the range first;
 * MADV_POPULATE_WRITE per 2-MiB piece before the stores; both.  Prints GB/s per variant and thread count.
 *   gcc -O2 -pthread -o tools/micro/hostfill tools/micro/hostfill.c && tools/micro/hostfill [GiB] */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static uint8_t* base; static size_t total; static int nthreads, variant;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void* worker(void* arg)
{
    const size_t k = (size_t)(intptr_t)arg, piece = (size_t)2 << 20;
    const size_t pieces = total / piece;
    for (size_t p = k; p < pieces; p += (size_t)nthreads) {            /* interleaved 2-MiB pieces, like the row jobs */
        uint8_t* q = base + p * piece;
        if (variant & 2) madvise(q, piece, MADV_POPULATE_WRITE);
        memset(q, 7, piece);
    }
    return NULL;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 1.0;
    total = (size_t)(gib * (1u << 30)) & ~(((size_t)2 << 20) - 1);
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[128] = "?";
    if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", line);
    const int ts[] = {1, 2, 4, 8, 16, 32};
    for (variant = 0; variant < 4; ++variant)
        for (unsigned i = 0; i < sizeof ts / sizeof ts[0]; ++i) {
            nthreads = ts[i];
            base = mmap(NULL, total + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (base == MAP_FAILED) { perror("mmap"); return 1; }
            uint8_t* keep = base;
            base = (uint8_t*)(((uintptr_t)base + (((size_t)2 << 20) - 1)) & ~(uintptr_t)(((size_t)2 << 20) - 1));
            if (variant & 1) madvise(base, total, MADV_HUGEPAGE);
            pthread_t th[64];
            const double t0 = now();
            for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, worker, (void*)(intptr_t)k);
            for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
            const double dt = now() - t0;
            printf("%-28s threads %2d  %7.1f ms  %6.1f GB/s\n", variant == 0 ? "plain" : variant == 1 ? "MADV_HUGEPAGE" : variant == 2 ? "MADV_POPULATE_WRITE" : "HUGEPAGE + POPULATE_WRITE",
                   nthreads, dt * 1e3, total / dt / 1e9);
            munmap(keep, total + ((size_t)2 << 20));
        }
    return 0;
}
