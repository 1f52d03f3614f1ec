/*
 * grm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see grm_oracle.h).
 *
 * Restates, in plain C, what the reference's absent native tools compute:
 *   - DSK / multidsk   (invoked at bin/kover/core/kover/dataset/tools/kmer_count.py:28-53,
 *                       src/app.py:1372)            -> orc_count_*
 *   - dsk2kover        (invoked at .../tools/kmer_pack.py:28-36) -> orc_build_matrix
 *   - Ray Surveyor TSV (invoked at src/app.py:1310; layout read by
 *                       dataset/create.py:121-137,241)           -> orc_write_tsv
 *   - bit layout       (bin/kover/core/kover/utils.py:117-187)   -> orc_pack_bits, ...
 * "parity unpinned" for the [EXT] rules -- see the header of grm_oracle.h.
 *
 * Deliberately simple: per genome "collect all canonical k-mers, sort, run-length";
 * N-way heap merge for the matrix (the structure [EXT] dsk2kover uses).
 */
#define _GNU_SOURCE
#include "grm_oracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
/* k-mers of up to 128 bases (256 bits); k <= 32 keeps its own one-word fast path below */
typedef struct { u128 hi, lo; } u256;
static inline int u256_lt(u256 a, u256 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
static inline int u256_eq(u256 a, u256 b) { return a.hi == b.hi && a.lo == b.lo; }
static inline u256 u256_push_low(u256 v, unsigned code)           /* (v << 2) | code */
{
    v.hi = (v.hi << 2) | (v.lo >> 126);
    v.lo = (v.lo << 2) | (u128)code;
    return v;
}
static inline u256 u256_push_high(u256 v, unsigned code, int k)   /* (v >> 2) | code << 2(k-1) */
{
    v.lo = (v.lo >> 2) | (v.hi << 126);
    v.hi >>= 2;
    const int p = 2 * (k - 1);
    if (p >= 128) v.hi |= (u128)code << (p - 128);
    else v.lo |= (u128)code << p;
    return v;
}
static inline u256 u256_mask_k(int k)
{
    u256 m;
    const int b = 2 * k;
    if (b >= 256) { m.hi = ~(u128)0; m.lo = ~(u128)0; }
    else if (b > 128) { m.hi = ((u128)1 << (b - 128)) - 1; m.lo = ~(u128)0; }
    else if (b == 128) { m.hi = 0; m.lo = ~(u128)0; }
    else { m.hi = 0; m.lo = ((u128)1 << b) - 1; }
    return m;
}
static inline u256 u256_and(u256 a, u256 b) { a.hi &= b.hi; a.lo &= b.lo; return a; }
static inline int words_of_k(int k) { return (k + 31) / 32; }
/* `words` uint64 of a key, most significant first */
static inline void u256_store(u256 v, int words, uint64_t *out)
{
    const uint64_t w[4] = {(uint64_t)(v.hi >> 64), (uint64_t)v.hi, (uint64_t)(v.lo >> 64), (uint64_t)v.lo};
    for (int j = 0; j < words; j++) out[j] = w[4 - words + j];
}
static inline u256 u256_load(const uint64_t *in, int words)
{
    uint64_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < words; j++) w[4 - words + j] = in[j];
    u256 v;
    v.hi = ((u128)w[0] << 64) | w[1];
    v.lo = ((u128)w[2] << 64) | w[3];
    return v;
}

/* ---------------------------------------------------------------- primitives ------ */
int orc_base_code(unsigned char c) { return (c >> 1) & 3; }   /* A0 C1 T2 G3  [EXT] */
int orc_base_bad(unsigned char c)  { return (c >> 3) & 1; }   /* N,n,K,M,... [EXT] */

int orc_canonical_ascii(const char *s, int k, uint64_t *out)
{
    if (k < 1 || k > 128) return -1;
    u256 fwd = {0, 0}, rc = {0, 0};
    for (int i = 0; i < k; i++) {
        unsigned char c = (unsigned char)s[i];
        if (orc_base_bad(c)) return -1;
        unsigned code = (unsigned)orc_base_code(c);
        fwd = u256_push_low(fwd, code);
        rc  = u256_push_high(rc, code ^ 2u, k);             /* complement = code^2 */
    }
    fwd = u256_and(fwd, u256_mask_k(k));
    u256_store(u256_lt(fwd, rc) ? fwd : rc, words_of_k(k), out);
    return 0;
}

void orc_decode(const uint64_t *w, int k, char *out)
{
    static const char L[4] = {'A', 'C', 'T', 'G'};
    const int words = words_of_k(k);
    for (int i = 0; i < k; i++) {
        const int bit = 2 * (k - 1 - i);                     /* from the least significant end */
        const uint64_t word = w[words - 1 - bit / 64];
        out[i] = L[(int)((word >> (bit & 63)) & 3)];
    }
}

/* ------------------------------------------------------------- k-mer scanning ----- */
typedef struct { u256 *v; size_t n, cap; } kvec256;
typedef struct { uint64_t *v; size_t n, cap; } kvec64;

typedef struct {
    int k;
    u256 mask;
    u256 fwd, rc;
    int run;              /* consecutive valid symbols ending here */
    uint64_t nocc;
    int words;
    kvec64 a64;
    kvec256 a256;
    int oom;
} scanner;

static void sc_reset(scanner *s) { s->run = 0; s->fwd.hi = s->fwd.lo = 0; s->rc.hi = s->rc.lo = 0; }

static void sc_emit(scanner *s, u256 can)
{
    s->nocc++;
    if (s->words == 1) {
        if (s->a64.n == s->a64.cap) {
            size_t nc = s->a64.cap ? s->a64.cap * 2 : (1u << 16);
            uint64_t *nv = (uint64_t *)realloc(s->a64.v, nc * sizeof(uint64_t));
            if (!nv) { s->oom = 1; return; }
            s->a64.v = nv; s->a64.cap = nc;
        }
        s->a64.v[s->a64.n++] = (uint64_t)can.lo;
    } else {
        if (s->a256.n == s->a256.cap) {
            size_t nc = s->a256.cap ? s->a256.cap * 2 : (1u << 16);
            u256 *nv = (u256 *)realloc(s->a256.v, nc * sizeof(u256));
            if (!nv) { s->oom = 1; return; }
            s->a256.v = nv; s->a256.cap = nc;
        }
        s->a256.v[s->a256.n++] = can;
    }
}

static inline void sc_feed(scanner *s, unsigned char c)
{
    if (orc_base_bad(c)) { sc_reset(s); return; }
    unsigned code = (unsigned)orc_base_code(c);
    s->fwd = u256_and(u256_push_low(s->fwd, code), s->mask);
    s->rc  = u256_push_high(s->rc, code ^ 2u, s->k);
    if (++s->run >= s->k) sc_emit(s, u256_lt(s->fwd, s->rc) ? s->fwd : s->rc);
}

/* One file image.  FASTA: a line whose first byte is '>' is a header and starts a new
 * record; every other line contributes all of its bytes except '\r' as symbols.
 * FASTQ (first non-blank byte '@'): 4-line records, line 0 header, line 1 sequence.
 * k-mers never span records. */
static void sc_scan_buffer(scanner *s, const unsigned char *p, size_t len)
{
    size_t i = 0;
    while (i < len && (p[i] == '\n' || p[i] == '\r' || p[i] == ' ' || p[i] == '\t')) i++;
    int fastq = (i < len && p[i] == '@');
    size_t lineno = 0;
    sc_reset(s);
    i = 0;
    while (i < len) {
        const unsigned char *nl = (const unsigned char *)memchr(p + i, '\n', len - i);
        size_t e = nl ? (size_t)(nl - p) : len;
        if (!fastq) {
            if (e > i && p[i] == '>') sc_reset(s);
            else for (size_t j = i; j < e; j++) if (p[j] != '\r') sc_feed(s, p[j]);
        } else {
            int phase = (int)(lineno & 3);
            if (phase == 0) sc_reset(s);
            else if (phase == 1) for (size_t j = i; j < e; j++) if (p[j] != '\r') sc_feed(s, p[j]);
        }
        lineno++;
        i = e + 1;
    }
    sc_reset(s);
}

/* ------------------------------------------------------------------ sorting ------- */
static void radix_sort_u64(uint64_t *a, size_t n)
{
    if (n < 2) return;
    uint64_t *b = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!b) { /* fall back */
        for (size_t i = 1; i < n; i++) { uint64_t x = a[i]; size_t j = i; while (j && a[j-1] > x) { a[j] = a[j-1]; j--; } a[j] = x; }
        return;
    }
    uint64_t *src = a, *dst = b;
    for (int pass = 0; pass < 8; pass++) {
        size_t hist[256] = {0};
        int sh = pass * 8;
        for (size_t i = 0; i < n; i++) hist[(src[i] >> sh) & 255]++;
        int skip = 0;
        for (int d = 0; d < 256; d++) if (hist[d] == n) { skip = 1; break; }
        if (skip) continue;
        size_t sum = 0;
        for (int d = 0; d < 256; d++) { size_t c = hist[d]; hist[d] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[hist[(src[i] >> sh) & 255]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
    free(b);
}

static int cmp_u256(const void *x, const void *y)
{
    const u256 a = *(const u256 *)x, b = *(const u256 *)y;
    return u256_lt(a, b) ? -1 : (u256_lt(b, a) ? 1 : 0);
}

/* sort + run-length + abundance filter -> orc_set */
static int finish_set(scanner *s, uint32_t abundance_min, orc_set *out)
{
    memset(out, 0, sizeof(*out));
    out->k = s->k; out->words = s->words; out->n_occurrences = s->nocc;
    if (s->oom) return -2;
    size_t n = s->words == 1 ? s->a64.n : s->a256.n;
    if (s->words == 1) radix_sort_u64(s->a64.v, n);
    else qsort(s->a256.v, n, sizeof(u256), cmp_u256);
    /* count distinct */
    size_t nd = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        if (s->words == 1) while (j < n && s->a64.v[j] == s->a64.v[i]) j++;
        else while (j < n && u256_eq(s->a256.v[j], s->a256.v[i])) j++;
        if ((uint64_t)(j - i) >= abundance_min) nd++;
        i = j;
    }
    out->kmers = (uint64_t *)malloc((nd ? nd : 1) * s->words * sizeof(uint64_t));
    out->counts = (uint32_t *)malloc((nd ? nd : 1) * sizeof(uint32_t));
    if (!out->kmers || !out->counts) return -2;
    size_t o = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        if (s->words == 1) while (j < n && s->a64.v[j] == s->a64.v[i]) j++;
        else while (j < n && u256_eq(s->a256.v[j], s->a256.v[i])) j++;
        uint64_t c = j - i;
        if (c >= abundance_min) {
            if (s->words == 1) out->kmers[o] = s->a64.v[i];
            else u256_store(s->a256.v[i], s->words, out->kmers + (size_t)s->words * o);
            out->counts[o] = c > 0xffffffffu ? 0xffffffffu : (uint32_t)c;
            o++;
        }
        i = j;
    }
    out->n = nd;
    return 0;
}

static int sc_init(scanner *s, int k)
{
    if (k < 1 || k > 128) return -1;
    memset(s, 0, sizeof(*s));
    s->k = k; s->mask = u256_mask_k(k); s->words = words_of_k(k);
    return 0;
}
static void sc_free(scanner *s) { free(s->a64.v); free(s->a256.v); }

int orc_count_buffers(const unsigned char *const *bufs, const size_t *lens, int n_bufs,
                      int k, uint32_t abundance_min, orc_set *out)
{
    scanner s;
    if (sc_init(&s, k)) return -1;
    for (int i = 0; i < n_bufs; i++) sc_scan_buffer(&s, bufs[i], lens[i]);
    int rc = finish_set(&s, abundance_min ? abundance_min : 1, out);
    sc_free(&s);
    return rc;
}

static unsigned char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *b = (unsigned char *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!b) { fclose(f); return NULL; }
    size_t got = fread(b, 1, (size_t)sz, f);
    fclose(f);
    *len = got;
    return b;
}

int orc_count_files(const char *const *paths, int n_paths, int k, uint32_t abundance_min,
                    orc_set *out)
{
    scanner s;
    if (sc_init(&s, k)) return -1;
    for (int i = 0; i < n_paths; i++) {
        size_t len = 0;
        unsigned char *b = slurp(paths[i], &len);
        if (!b) { sc_free(&s); return -3; }
        sc_scan_buffer(&s, b, len);
        free(b);
    }
    int rc = finish_set(&s, abundance_min ? abundance_min : 1, out);
    sc_free(&s);
    return rc;
}

int orc_count_pooled_files(const char *const *paths, int n_paths, int k,
                           uint32_t abundance_min, orc_set *out)
{
    /* `dsk -file list` counts the union stream of all listed files as one bank. */
    return orc_count_files(paths, n_paths, k, abundance_min, out);
}

void orc_set_free(orc_set *s)
{
    if (!s) return;
    free(s->kmers); free(s->counts);
    memset(s, 0, sizeof(*s));
}

/* ------------------------------------------------------------- merge / pack ------- */
static inline u256 set_key(const orc_set *s, size_t i)
{
    return u256_load(s->kmers + (size_t)s->words * i, s->words);
}

typedef struct { u256 key; int g; } hnode;

static void heap_sift_down(hnode *h, size_t n, size_t i)
{
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && (u256_lt(h[l].key, h[m].key) || (u256_eq(h[l].key, h[m].key) && h[l].g < h[m].g))) m = l;
        if (r < n && (u256_lt(h[r].key, h[m].key) || (u256_eq(h[r].key, h[m].key) && h[r].g < h[m].g))) m = r;
        if (m == i) return;
        hnode t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

/* merge the sub-ranges [lo[g], hi[g]) of every set; append to growing outputs */
typedef struct {
    uint64_t *kmers; uint64_t *cols; /* cols: n_rows words per k-mer (column-major) */
    uint32_t *ng; size_t n, cap;
} mergeout;

static int merge_range(const orc_set *sets, int n_genomes, const size_t *lo, const size_t *hi,
                       int filter_singleton, size_t n_rows, int words, mergeout *o)
{
    hnode *heap = (hnode *)malloc(sizeof(hnode) * (size_t)(n_genomes ? n_genomes : 1));
    size_t *pos = (size_t *)malloc(sizeof(size_t) * (size_t)(n_genomes ? n_genomes : 1));
    uint64_t *bits = (uint64_t *)malloc(sizeof(uint64_t) * (n_rows ? n_rows : 1));
    if (!heap || !pos || !bits) { free(heap); free(pos); free(bits); return -2; }
    size_t hn = 0;
    for (int g = 0; g < n_genomes; g++) {
        pos[g] = lo[g];
        if (pos[g] < hi[g]) { heap[hn].key = set_key(&sets[g], pos[g]); heap[hn].g = g; hn++; }
    }
    for (size_t i = hn / 2; i-- > 0;) heap_sift_down(heap, hn, i);
    int rc = 0;
    while (hn) {
        u256 key = heap[0].key;
        memset(bits, 0, sizeof(uint64_t) * n_rows);
        uint32_t ng = 0;
        while (hn && u256_eq(heap[0].key, key)) {
            int g = heap[0].g;
            bits[g >> 6] |= (uint64_t)1 << (63 - (g & 63));          /* utils.py:133-156 */
            ng++;
            if (++pos[g] < hi[g]) heap[0].key = set_key(&sets[g], pos[g]);
            else heap[0] = heap[--hn];
            heap_sift_down(heap, hn, 0);
        }
        if (filter_singleton && ng == 1) continue;                 /* SURVEY 8(c)(6) */
        if (o->n == o->cap) {
            size_t nc = o->cap ? o->cap * 2 : 4096;
            uint64_t *a = (uint64_t *)realloc(o->kmers, nc * words * sizeof(uint64_t));
            if (a) o->kmers = a;
            uint64_t *b = (uint64_t *)realloc(o->cols, nc * (n_rows ? n_rows : 1) * sizeof(uint64_t));
            if (b) o->cols = b;
            uint32_t *c = (uint32_t *)realloc(o->ng, nc * sizeof(uint32_t));
            if (c) o->ng = c;
            if (!a || !b || !c) { rc = -2; break; }
            o->cap = nc;
        }
        u256_store(key, words, o->kmers + (size_t)words * o->n);
        memcpy(o->cols + o->n * n_rows, bits, n_rows * sizeof(uint64_t));
        o->ng[o->n] = ng;
        o->n++;
    }
    free(heap); free(pos); free(bits);
    return rc;
}

static size_t lower_bound_set(const orc_set *s, u256 key)
{
    size_t lo = 0, hi = s->n;
    while (lo < hi) { size_t m = (lo + hi) / 2; if (u256_lt(set_key(s, m), key)) lo = m + 1; else hi = m; }
    return lo;
}

typedef struct {
    const orc_set *sets; int n_genomes; int filter; size_t n_rows; int words;
    u256 lo_key, hi_key; int has_lo, has_hi;
    mergeout out; int rc;
} merge_job;

static void *merge_thread(void *arg)
{
    merge_job *j = (merge_job *)arg;
    size_t *lo = (size_t *)malloc(sizeof(size_t) * (size_t)(j->n_genomes ? j->n_genomes : 1));
    size_t *hi = (size_t *)malloc(sizeof(size_t) * (size_t)(j->n_genomes ? j->n_genomes : 1));
    for (int g = 0; g < j->n_genomes; g++) {
        lo[g] = j->has_lo ? lower_bound_set(&j->sets[g], j->lo_key) : 0;
        hi[g] = j->has_hi ? lower_bound_set(&j->sets[g], j->hi_key) : j->sets[g].n;
    }
    j->rc = merge_range(j->sets, j->n_genomes, lo, hi, j->filter, j->n_rows, j->words, &j->out);
    free(lo); free(hi);
    return NULL;
}

static int build_matrix_mt(const orc_set *sets, int n_genomes, int filter_singleton,
                           int n_threads, orc_matrix *out)
{
    memset(out, 0, sizeof(*out));
    if (n_genomes < 0) return -1;
    int words = n_genomes ? sets[0].words : 1, k = n_genomes ? sets[0].k : 0;
    for (int g = 0; g < n_genomes; g++) if (sets[g].words != words || sets[g].k != k) return -1;
    size_t n_rows = ((size_t)n_genomes + 63) / 64;
    if (n_threads < 1) n_threads = 1;
    /* splitters: quantiles of the largest set */
    int big = 0;
    for (int g = 1; g < n_genomes; g++) if (sets[g].n > sets[big].n) big = g;
    if (n_genomes == 0 || sets[big].n < (size_t)n_threads * 4) n_threads = 1;
    merge_job *jobs = (merge_job *)calloc((size_t)n_threads, sizeof(merge_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].sets = sets; jobs[t].n_genomes = n_genomes; jobs[t].filter = filter_singleton;
        jobs[t].n_rows = n_rows; jobs[t].words = words;
        if (t > 0) { jobs[t].has_lo = 1; jobs[t].lo_key = set_key(&sets[big], sets[big].n * (size_t)t / (size_t)n_threads); }
        if (t + 1 < n_threads) { jobs[t].has_hi = 1; jobs[t].hi_key = set_key(&sets[big], sets[big].n * (size_t)(t + 1) / (size_t)n_threads); }
    }
    if (n_threads == 1) merge_thread(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, merge_thread, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    size_t total = 0; int rc = 0;
    for (int t = 0; t < n_threads; t++) { total += jobs[t].out.n; if (jobs[t].rc) rc = jobs[t].rc; }
    out->n_kmers = total; out->n_rows = n_rows; out->n_genomes = n_genomes; out->words = words; out->k = k;
    out->kmers = (uint64_t *)malloc((total ? total : 1) * words * sizeof(uint64_t));
    out->matrix = (uint64_t *)calloc((total ? total : 1) * (n_rows ? n_rows : 1), sizeof(uint64_t));
    out->n_genomes_with = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    if (!out->kmers || !out->matrix || !out->n_genomes_with) rc = -2;
    size_t base = 0;
    for (int t = 0; t < n_threads && !rc; t++) {
        mergeout *o = &jobs[t].out;
        if (!o->n) continue;                     /* (a slice without k-mers has no arrays: memcpy must not be handed NULL, even for 0 bytes) */
        memcpy(out->kmers + base * words, o->kmers, o->n * words * sizeof(uint64_t));
        memcpy(out->n_genomes_with + base, o->ng, o->n * sizeof(uint32_t));
        for (size_t c = 0; c < o->n; c++)
            for (size_t r = 0; r < n_rows; r++)
                out->matrix[r * total + base + c] = o->cols[c * n_rows + r];
        base += o->n;
    }
    for (int t = 0; t < n_threads; t++) { free(jobs[t].out.kmers); free(jobs[t].out.cols); free(jobs[t].out.ng); }
    free(jobs); free(th);
    return rc;
}

int orc_build_matrix(const orc_set *sets, int n_genomes, int filter_singleton, orc_matrix *out)
{
    return build_matrix_mt(sets, n_genomes, filter_singleton, 1, out);
}

void orc_matrix_free(orc_matrix *m)
{
    if (!m) return;
    free(m->kmers); free(m->matrix); free(m->n_genomes_with);
    memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------------ kover bit layout ---- */
void orc_pack_bits(const uint8_t *bits, size_t n_rows, size_t n_cols, int pack_size, uint64_t *out)
{
    size_t prow = (n_rows + (size_t)pack_size - 1) / (size_t)pack_size;
    memset(out, 0, prow * n_cols * sizeof(uint64_t));
    for (size_t i = 0; i < n_rows; i++) {
        size_t pr = i / (size_t)pack_size;
        int sh = pack_size - (int)(i % (size_t)pack_size) - 1;       /* utils.py:151 */
        for (size_t c = 0; c < n_cols; c++)
            if (bits[i * n_cols + c]) out[pr * n_cols + c] |= (uint64_t)1 << sh;
    }
}

int orc_minimum_uint_bytes(uint64_t max_value)                       /* utils.py:117-130 */
{
    if (max_value <= 0xffu) return 1;
    if (max_value <= 0xffffu) return 2;
    if (max_value <= 0xffffffffu) return 4;
    return 8;
}

/* ------------------------------------------------------------------- TSV ---------- */
int orc_write_tsv(const orc_matrix *m, const char *const *genome_ids, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -3;
    fputs("kmers", f);                                               /* create.py:241 */
    for (int g = 0; g < m->n_genomes; g++) { fputc('\t', f); fputs(genome_ids[g], f); }
    fputc('\n', f);
    char *line = (char *)malloc((size_t)m->k + 2 * (size_t)m->n_genomes + 2);
    for (size_t c = 0; c < m->n_kmers; c++) {
        orc_decode(m->kmers + c * m->words, m->k, line);
        size_t p = (size_t)m->k;
        for (int g = 0; g < m->n_genomes; g++) {
            line[p++] = '\t';
            line[p++] = (m->matrix[(size_t)(g >> 6) * m->n_kmers + c] >> (63 - (g & 63))) & 1 ? '1' : '0';
        }
        line[p++] = '\n';
        fwrite(line, 1, p, f);
    }
    free(line);
    return fclose(f) ? -3 : 0;
}

/* ------------------------------------------------- threaded pipeline (baseline) --- */
typedef struct {
    const unsigned char *const *bufs; const size_t *lens; int n; int k; uint32_t amin;
    orc_set *sets; int next; pthread_mutex_t mu; int rc;
} count_pool;

static void *count_worker(void *arg)
{
    count_pool *p = (count_pool *)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        int g = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (g >= p->n) break;
        int rc = orc_count_buffers(&p->bufs[g], &p->lens[g], 1, p->k, p->amin, &p->sets[g]);
        if (rc) p->rc = rc;
    }
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int orc_pipeline_buffers(const unsigned char *const *bufs, const size_t *lens, int n_genomes,
                         int k, uint32_t abundance_min, int filter_singleton, int n_threads,
                         orc_matrix *out, double *count_s, double *merge_s, uint64_t *n_occ)
{
    if (n_threads < 1) n_threads = 1;
    orc_set *sets = (orc_set *)calloc((size_t)(n_genomes ? n_genomes : 1), sizeof(orc_set));
    count_pool pool = {bufs, lens, n_genomes, k, abundance_min, sets, 0, PTHREAD_MUTEX_INITIALIZER, 0};
    double t0 = now_s();
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, count_worker, &pool);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    double t1 = now_s();
    int rc = pool.rc;
    uint64_t occ = 0;
    for (int g = 0; g < n_genomes; g++) occ += sets[g].n_occurrences;
    if (!rc) rc = build_matrix_mt(sets, n_genomes, filter_singleton, n_threads, out);
    double t2 = now_s();
    for (int g = 0; g < n_genomes; g++) orc_set_free(&sets[g]);
    free(sets);
    if (count_s) *count_s = t1 - t0;
    if (merge_s) *merge_s = t2 - t1;
    if (n_occ) *n_occ = occ;
    return rc;
}
