/* tests/c_abi_consumer.c -- a plain-C consumer of liblsq_mi355x.so: no Python, no torch, no HIP headers.
 * It is what a foreign-language binding (the Julia ccall shim of INTEGRATION.md) does: load the library, hand
 * over host buffers in the reference's column-major layouts, read caller-allocated outputs.
 *
 *   c_abi_consumer <in.bin> <out.bin>
 * in.bin : int32 d, n, m, h, nr, icmiter, npert, randord; uint64 seed; int64 ilsiters[nr];
 *          float X[n*d]; int16 B[n*m]; float K[m*h*d]
 * out.bin: int16 Bs[nr*n*m]; float objs[nr]; double seconds (wall time of the lsq_encode_icm call)
 * Build: gcc tests/c_abi_consumer.c -Iinclude -Llocal-search-quantization_amd -llsq_mi355x -o c_abi_consumer
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "lsq_mi355x.h"

static void die(const char *what) {
    fprintf(stderr, "c_abi_consumer: %s: %s\n", what, lsq_last_error());
    exit(1);
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open input"); return 2; }
    int32_t hdr[8];
    uint64_t seed;
    if (fread(hdr, sizeof(int32_t), 8, f) != 8 || fread(&seed, sizeof seed, 1, f) != 1) return 2;
    const int d = hdr[0], m = hdr[2], h = hdr[3], nr = hdr[4], icmiter = hdr[5], npert = hdr[6], randord = hdr[7];
    const int64_t n = hdr[1];
    int64_t *ils = malloc(sizeof(int64_t) * (size_t)nr);
    float *X = malloc(sizeof(float) * (size_t)n * d);
    int16_t *B = malloc(sizeof(int16_t) * (size_t)n * m);
    float *K = malloc(sizeof(float) * (size_t)m * h * d);
    int16_t *Bs = malloc(sizeof(int16_t) * (size_t)nr * n * m);
    float *objs = malloc(sizeof(float) * (size_t)nr);
    if (fread(ils, sizeof(int64_t), (size_t)nr, f) != (size_t)nr) return 2;
    if (fread(X, sizeof(float), (size_t)n * d, f) != (size_t)n * d) return 2;
    if (fread(B, sizeof(int16_t), (size_t)n * m, f) != (size_t)n * m) return 2;
    if (fread(K, sizeof(float), (size_t)m * h * d, f) != (size_t)m * h * d) return 2;
    fclose(f);

    lsq_ctx *ctx = NULL;
    if (lsq_create(&ctx, 0) != LSQ_OK) die("lsq_create");
    /* one warm-up call so the timed one excludes workspace allocation, like a long-lived caller */
    if (lsq_encode_icm(ctx, X, B, K, d, n, m, h, ils, nr, icmiter, npert, randord, 2, seed, 0, 0, Bs, objs) != LSQ_OK) die("lsq_encode_icm");
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (lsq_encode_icm(ctx, X, B, K, d, n, m, h, ils, nr, icmiter, npert, randord, 2, seed, 0, 0, Bs, objs) != LSQ_OK) die("lsq_encode_icm");
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    lsq_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f) { perror("open output"); return 2; }
    fwrite(Bs, sizeof(int16_t), (size_t)nr * n * m, f);
    fwrite(objs, sizeof(float), (size_t)nr, f);
    fwrite(&secs, sizeof secs, 1, f);
    fclose(f);
    printf("c_abi_consumer: n=%lld d=%d m=%d  %.3f s  %.0f vectors/s  obj[last]=%g\n", (long long)n, d, m, secs, (double)n / secs, objs[nr - 1]);
    return 0;
}
