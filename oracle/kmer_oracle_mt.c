/*
 * kmer_oracle_mt.c -- the counting path of kmer_oracle.c on ALL host cores.  TEST INFRASTRUCTURE ONLY:
 * bench.py's cpu_baseline leg times it beside the scalar port ("the same with all host
 * cores -- state nproc"); tests/ check it against pko_count_fasta.  Nothing under pykmer_amd/ calls it.
 *
 * Same semantics as pko_count_fasta (parse_fasta indexer.py:45-99, gen_kmers :130-160, canonical
 * :341, saturating count :239,262), restricted to inputs whose only blanks are line terminators
 * ('\n', '\r') outside header text: anything else (spaces or tabs in or around sequence lines, or in front
 * of a '>') makes it return 1 and the caller
 * stays with the scalar port.  That restriction is what makes a cheap split possible:
 *
 *   - every byte after a '\n' is a line start, and a line is a header iff its first byte is '>';
 *   - a thread that starts at a line start only needs (a) whether any header precedes it (text before
 *     the first header is dropped, indexer.py:80-82) and (b) the run of valid bases that ends right
 *     before it -- found by walking back over the previous sequence lines until k-1 bases are
 *     collected, or a non-base, a header line or the start of the file is met.
 *
 * Table updates are saturating byte increments done with compare-and-swap, so threads may meet on
 * the same address.  Per-record results (names, lengths) are not produced here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static inline int mt_conv(uint8_t c) {
    switch (c | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': return 3;
    default: return 4;
    }
}
static inline int mt_term(uint8_t c) { return c == '\n' || c == '\r'; }
/* str.strip() whitespace (ASCII) that is not a line terminator */
static inline int mt_other_blank(uint8_t c) { return c == ' ' || c == 9 || c == 11 || c == 12 || (c >= 28 && c <= 31); }

static inline void sat_inc(uint8_t *p) {
    uint8_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old != 255 && !__atomic_compare_exchange_n(p, &old, (uint8_t)(old + 1), 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
}

/* valid bases that end right before byte `at` (a line start), oldest first; returns how many (<= want) */
static int look_back(const uint8_t *f, uint64_t at, int want, uint8_t *bases) {
    int got = 0;
    uint64_t pos = at;
    while (got < want) {
        while (pos > 0 && mt_term(f[pos - 1])) pos--;               /* skip terminators / empty lines */
        if (pos == 0) break;
        uint64_t le = pos, ls = pos;
        while (ls > 0 && !mt_term(f[ls - 1])) ls--;
        if (f[ls] == '>') break;                                     /* a header line: the record starts after it */
        int stop = 0;
        for (uint64_t i = le; i > ls && got < want; i--) {
            int b = mt_conv(f[i - 1]);
            if (b > 3) { stop = 1; break; }
            bases[got++] = (uint8_t)b;                               /* newest first for now */
        }
        if (stop) break;
        pos = ls;
    }
    for (int i = 0; i < got / 2; i++) { uint8_t t = bases[i]; bases[i] = bases[got - 1 - i]; bases[got - 1 - i] = t; }
    return got;
}

int pko_count_fasta_mt(const uint8_t *fasta, uint64_t n_bytes, int k, uint8_t *table, uint64_t *num_kmers_out,
                       uint64_t *total_bp_out, int threads) {
    if (k <= 0 || (k & 1) == 0 || k > 31) return -1;                /* tools.py:165-167 */
    if (threads < 1) threads = 1;
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    const int top = 2 * (k - 1);
    const int T = threads;
    uint64_t *split = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(T + 1));
    uint64_t *hdrs = (uint64_t *)calloc((size_t)T + 1, sizeof(uint64_t));
    uint64_t *kmers = (uint64_t *)calloc((size_t)T, sizeof(uint64_t));
    uint64_t *bps = (uint64_t *)calloc((size_t)T, sizeof(uint64_t));
    int irregular = 0;
    split[0] = 0;
    for (int t = 1; t < T; t++) {
        uint64_t p = n_bytes / (uint64_t)T * (uint64_t)t;
        if (p < split[t - 1]) p = split[t - 1];
        const uint8_t *nl = p < n_bytes ? (const uint8_t *)memchr(fasta + p, '\n', n_bytes - p) : NULL;
        split[t] = nl ? (uint64_t)(nl - fasta) + 1 : n_bytes;
    }
    split[T] = n_bytes;

    /* pass 1: header lines per range, and a check that the input has no blanks besides terminators */
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(| : irregular)
    for (int t = 0; t < T; t++) {
        uint64_t h = 0;
        int at_start = 1, in_header = 0;
        for (uint64_t i = split[t]; i < split[t + 1]; i++) {
            const uint8_t c = fasta[i];
            if (mt_term(c)) { at_start = 1; in_header = 0; continue; }
            if (at_start && c == '>') { h++; in_header = 1; }
            if (!in_header && mt_other_blank(c)) irregular = 1;      /* header text may hold blanks, sequence lines may not */
            at_start = 0;
        }
        hdrs[t + 1] = h;
    }
    if (irregular) { free(split); free(hdrs); free(kmers); free(bps); return 1; }
    for (int t = 0; t < T; t++) hdrs[t + 1] += hdrs[t];              /* headers before range t+1 */

    /* pass 2: count */
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++) {
        uint64_t fwd = 0, rev = 0, nk = 0, bp = 0;
        int run = 0, have_rec = hdrs[t] != 0;
        if (t > 0 && split[t] < split[t + 1]) {
            uint8_t halo[32];
            const int got = look_back(fasta, split[t], k - 1, halo);
            for (int i = 0; i < got; i++) {
                fwd = ((fwd << 2) | halo[i]) & mask;
                rev = (rev >> 2) | ((uint64_t)(3 - halo[i]) << top);
            }
            run = got;
        }
        int at_start = 1, in_header = 0;
        for (uint64_t i = split[t]; i < split[t + 1]; i++) {
            const uint8_t c = fasta[i];
            if (mt_term(c)) { at_start = 1; in_header = 0; continue; }
            if (at_start) {
                at_start = 0;
                if (c == '>') { in_header = 1; have_rec = 1; run = 0; continue; }   /* indexer.py:66-82 */
            }
            if (in_header) continue;
            if (have_rec) bp++;                                      /* seq_len counts every character (indexer.py:77) */
            const int b = mt_conv(c);
            if (b > 3) { run = 0; continue; }
            fwd = ((fwd << 2) | (uint64_t)b) & mask;                 /* indexer.py:149 */
            rev = (rev >> 2) | ((uint64_t)(3 - b) << top);           /* indexer.py:150 */
            if (run < k) run++;
            if (run == k && have_rec) {
                sat_inc(&table[fwd < rev ? fwd : rev]);              /* indexer.py:341, 239, 262 */
                nk++;
            }
        }
        kmers[t] = nk; bps[t] = bp;
    }
    uint64_t nk = 0, bp = 0;
    for (int t = 0; t < T; t++) { nk += kmers[t]; bp += bps[t]; }
    if (num_kmers_out) *num_kmers_out = nk;
    if (total_bp_out) *total_bp_out = bp;
    free(split); free(hdrs); free(kmers); free(bps);
    return 0;
}

/*
 * pko_gram_mt -- the merger's pair loop on `threads` host threads, the way merger.py:137-153 spreads
 * pairs over a process pool: each (k, l) pair is one task that walks both tables
 * (Header.calculate_distance, tools.py:439-493: s_valid, o_valid, their sums and the sum of the AND).
 * Same output layout as pko_gram.  bench.py times it as the merge's CPU baseline (threads = 1 and all cores).
 */
int pko_gram_mt(const uint8_t *const *tables, int N, uint64_t n, int min_count, int max_count,
                uint64_t *matrix, int threads) {
    if (min_count < 1 || max_count > 255) return -1;                 /* merger.py:90-91 */
    const int n_pairs = N * (N - 1) / 2;
    int *pi = (int *)malloc(sizeof(int) * (size_t)(n_pairs > 0 ? n_pairs : 1)), *pj = (int *)malloc(sizeof(int) * (size_t)(n_pairs > 0 ? n_pairs : 1));
    int p = 0;
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++) { pi[p] = i; pj[p] = j; p++; }
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int q = 0; q < n_pairs; q++) {
        const uint8_t *a = tables[pi[q]], *b = tables[pj[q]];
        uint64_t ta = 0, tb = 0, sh = 0;
        for (uint64_t x = 0; x < n; x++) {                           /* tools.py:473-482 */
            const int va = a[x] >= min_count && a[x] <= max_count, vb = b[x] >= min_count && b[x] <= max_count;
            ta += (uint64_t)va; tb += (uint64_t)vb; sh += (uint64_t)(va & vb);
        }
        uint64_t *ij = matrix + ((uint64_t)pi[q] * N + pj[q]) * 3, *ji = matrix + ((uint64_t)pj[q] * N + pi[q]) * 3;
        ij[0] = ta; ij[1] = tb; ij[2] = sh;                          /* merger.py:175-176 */
        ji[0] = tb; ji[1] = ta; ji[2] = sh;
    }
    free(pi); free(pj);
    return 0;
}
