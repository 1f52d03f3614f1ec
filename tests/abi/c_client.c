/* A plain C99 client of libgrmkmer.so (include/grm_kmer.h): what a non-Python host would link.
 * Runs without a GPU: version string, the no-device contract of grm_create, and the two writers on a
 * host-only matrix (grm_matrix_from_host -> grm_write_tsv).  Exit code 0 = all checks passed. */
#include <stdio.h>
#include <string.h>
#include "../../include/grm_kmer.h"

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    if (!grm_version() || !strstr(grm_version(), "gfx950")) return 3;
    /* 3 genomes, 2 columns, k = 3: k-mers AAC (code 0,0,1 = 1) and ACT (0,1,2 = 6), ascending */
    const uint64_t kmers[2] = {1, 6};
    /* genome 0 carries both, genome 1 the first, genome 2 the second: bit 63 - i of word-row 0 */
    const uint64_t data[2] = {(1ull << 63) | (1ull << 62), (1ull << 63) | (1ull << 61)};
    grm_matrix *m = NULL;
    if (grm_matrix_from_host(kmers, data, 2, 3, 3, &m) != GRM_OK || !m) return 4;
    if (grm_matrix_n_kmers(m) != 2 || grm_matrix_n_genomes(m) != 3 || grm_matrix_n_rows(m) != 1 || grm_matrix_k(m) != 3) return 5;
    const char *ids[3] = {"g0", "g1", "g2"};
    if (grm_write_tsv(m, ids, argv[1]) != GRM_OK) return 6;
    grm_matrix_free(m);
    FILE *f = fopen(argv[1], "r");
    if (!f) return 7;
    char buf[256];
    size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    if (strcmp(buf, "kmers\tg0\tg1\tg2\nAAC\t1\t1\t0\nACT\t1\t0\t1\n") != 0) { fputs(buf, stderr); return 8; }
    /* no device in this process: the context must not come up (no CPU fallback) -- when a GPU is
     * present the call succeeds and the context is simply destroyed again */
    grm_ctx *c = grm_create(0, 1);
    if (c) grm_destroy(c);
    if (grm_create(-1, 1) != NULL) return 9;
    puts("c client ok");
    return 0;
}
