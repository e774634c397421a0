/* examples/roundtrip.c -- the C ABI in ~40 lines: encode a sequence, decode it back, report an
 * invalid base the way the reference does.
 *   gcc -Iinclude examples/roundtrip.c -Lbitnuc_amd -lbitnuc_hip -Wl,-rpath,$PWD/bitnuc_amd -o roundtrip
 * Needs an MI355X (gfx950); there is no CPU fallback. */
#include "bitnuc_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void) {
    bitnuc_ctx *ctx;
    bitnuc_err err;
    if (bitnuc_ctx_create(0, &ctx, &err) != BITNUC_OK) {
        fprintf(stderr, "no HIP device (hipError %d)\n", err.backend_code);
        return 1;
    }
    const char *seq = "ACGTACGTTTGACCAGTACGATCGATCGATTAGCAT"; /* 36 bases -> 2 words */
    size_t n = strlen(seq), nw = 0;
    uint64_t words[2];
    bitnuc_encode(ctx, (const uint8_t *)seq, n, words, &nw, &err); /* bitnuc::encode, src/utils/mod.rs:22-25 */
    printf("%zu bases -> %zu words: %016llx %016llx\n", n, nw, (unsigned long long)words[0], (unsigned long long)words[1]);

    char back[64] = {0};
    bitnuc_decode(ctx, words, nw, n, (uint8_t *)back, &err); /* bitnuc::decode, src/utils/mod.rs:60-62 */
    printf("decoded: %s (%s)\n", back, strcmp(back, seq) == 0 ? "round trip ok" : "MISMATCH");

    uint64_t kmer;
    if (bitnuc_as_2bit(ctx, (const uint8_t *)"ACGN", 4, &kmer, &err) == BITNUC_INVALID_BASE) /* packing/mod.rs:186-187 */
        printf("as_2bit(\"ACGN\") -> InvalidBase('%c') at index %llu\n", err.byte, (unsigned long long)err.index);
    bitnuc_ctx_destroy(ctx);
    return 0;
}
