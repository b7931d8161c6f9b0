/*
 * kmer_oracle.c -- CPU restatement of the pykmer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the
 * product path (pykmer_amd/) never does.  Parity pinning: this file is checked against the
 * golden vectors under tests/golden/, which were produced by running the reference itself
 * (/root/reference, CPython + numpy) through oracle/gen_golden.py -- see DESIGN.md "Oracle".
 *
 * What it restates (citations into /root/reference):
 *   pko_count_fasta  parse_fasta          indexer.py:45-99   (strip, '>' records, CONV lookup)
 *                    gen_kmers            indexer.py:130-160 (fwd / rev values per window)
 *                    canonical + count    indexer.py:341-342,349-351
 *                    process_kmers        indexer.py:162-297 (net effect: table[a]=min(255,#a))
 *   pko_table_stats  Header.update_stats  tools.py:246-263
 *   pko_gram         Header.calculate_distance tools.py:439-493 for every pair, filled into the
 *                    matrix the way merger.merge does (merger.py:175-176)
 *
 * The reference walks each record as a Python tuple and recomputes fwd/rev in O(k) per window;
 * this restatement streams bytes once with a rolling update, which yields the same values:
 *   fwd = sum 4^(k-1-p) * b_p            -> fwd' = ((fwd << 2) | b) & (4^k - 1)
 *   rev = sum 4^p * (3 - b_p)            -> rev' = (rev >> 2) | ((3 - b) << 2(k-1))
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t name_off;      /* byte offset of the first header character after '>'            */
    uint64_t name_len;      /* header text length after strip()                               */
    uint64_t seq_len;       /* stripped sequence characters, valid or not (indexer.py:77,93)  */
    uint64_t n_valid_kmers; /* windows with no None (indexer.py:144)                          */
} pko_record;

/* str.strip() whitespace restricted to ASCII (indexer.py:56) */
static inline int is_ws(uint8_t c) {
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);
}
/* CONV (indexer.py:36-41): A/a 0, C/c 1, G/g 2, T/t 3, everything else None (=4 here) */
static inline int conv(uint8_t c) {
    switch (c | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': return 3;
    default: return 4;
    }
}

enum { LS_START = 0, LS_HEADER = 1, LS_SEQ = 2 };

/*
 * table: 4^k bytes, must be zero-initialised by the caller (lets a caller accumulate several
 * inputs, as the reference's memmap does across flushes).  Returns 0, or -1 on bad k.
 * Universal-newline text mode (indexer.py:110) makes '\r', '\n' and '\r\n' all line ends; since
 * empty lines are skipped (indexer.py:58-59) treating '\r' and '\n' as separate terminators is
 * equivalent.
 */
/* `table` may be NULL when only the list is wanted (k = 19: the table would be 256 GiB); `kmers_out` (nullable) receives
 * the canonical value of every valid window in text order, up to kmers_cap of them. */
int pko_count_fasta_ex(const uint8_t *fasta, uint64_t n_bytes, int k, uint8_t *table,
                       uint64_t *num_kmers_out, uint64_t *total_bp_out, pko_record *recs,
                       uint64_t recs_cap, uint64_t *n_recs_out, uint64_t *kmers_out, uint64_t kmers_cap) {
    if (k <= 0 || (k & 1) == 0 || k > 31) return -1;            /* tools.py:165-167 */
    const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int top = 2 * (k - 1);
    int ls = LS_START, have_rec = 0, run = 0;
    uint64_t pending_ws = 0, fwd = 0, rev = 0;
    uint64_t n_recs = 0, num_kmers = 0, total_bp = 0;
    pko_record cur = {0, 0, 0, 0};
    uint64_t name_end = 0;

    for (uint64_t pos = 0; pos < n_bytes; pos++) {
        uint8_t c = fasta[pos];
        if (c == '\n' || c == '\r') {
            pending_ws = 0;                                      /* trailing ws is stripped */
            ls = LS_START;
            continue;
        }
        if (ls == LS_START) {
            if (is_ws(c)) continue;                              /* leading ws is stripped  */
            if (c == '>') {                                      /* indexer.py:66-82        */
                if (have_rec) {
                    cur.name_len = name_end - cur.name_off;
                    if (n_recs < recs_cap) recs[n_recs] = cur;
                    n_recs++;
                    total_bp += cur.seq_len;
                }
                have_rec = 1;
                cur.name_off = pos + 1; cur.seq_len = 0; cur.n_valid_kmers = 0;
                name_end = pos + 1;
                run = 0;
                ls = LS_HEADER;
                continue;
            }
            ls = LS_SEQ;                                         /* falls into the SEQ case */
        }
        if (ls == LS_HEADER) {
            if (!is_ws(c)) name_end = pos + 1;
            continue;
        }
        /* LS_SEQ */
        if (is_ws(c)) { pending_ws++; continue; }
        if (pending_ws) {                                        /* interior ws: kept, maps to None */
            cur.seq_len += pending_ws; run = 0; pending_ws = 0;
        }
        cur.seq_len++;
        int b = conv(c);
        if (b > 3) { run = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)b) & mask;                 /* indexer.py:149 */
        rev = (rev >> 2) | ((uint64_t)(3 - b) << top);           /* indexer.py:150 */
        if (run < k) run++;
        if (run == k && have_rec) {                              /* lines before the 1st header are dropped (indexer.py:80-82) */
            uint64_t a = fwd < rev ? fwd : rev;                  /* indexer.py:341 */
            if (table && table[a] != 255) table[a]++;            /* indexer.py:239,262 */
            if (kmers_out && num_kmers < kmers_cap) kmers_out[num_kmers] = a;
            cur.n_valid_kmers++;
            num_kmers++;
        }
    }
    if (have_rec) {                                              /* indexer.py:86-95 */
        cur.name_len = name_end - cur.name_off;
        if (n_recs < recs_cap) recs[n_recs] = cur;
        n_recs++;
        total_bp += cur.seq_len;
    }
    if (num_kmers_out) *num_kmers_out = num_kmers;
    if (total_bp_out) *total_bp_out = total_bp;
    if (n_recs_out) *n_recs_out = n_recs;
    return 0;
}

int pko_count_fasta(const uint8_t *fasta, uint64_t n_bytes, int k, uint8_t *table,
                    uint64_t *num_kmers_out, uint64_t *total_bp_out, pko_record *recs,
                    uint64_t recs_cap, uint64_t *n_recs_out) {
    return pko_count_fasta_ex(fasta, n_bytes, k, table, num_kmers_out, total_bp_out, recs, recs_cap, n_recs_out, 0, 0);
}

/* tools.py:246-263.  hist[i] = #{a : table[a] == i+1}; vals = {sum, count(nonzero), min, max}. */
int pko_table_stats(const uint8_t *table, uint64_t n, uint64_t hist[255], uint64_t vals[4]) {
    uint64_t h[256];
    memset(h, 0, sizeof h);
    for (uint64_t i = 0; i < n; i++) h[table[i]]++;
    uint64_t sum = 0, cnt = 0, mn = 255, mx = 0;
    for (int v = 0; v < 256; v++) {
        if (!h[v]) continue;
        sum += (uint64_t)v * h[v];
        if (v) cnt += h[v];
        if ((uint64_t)v < mn) mn = (uint64_t)v;
        if ((uint64_t)v > mx) mx = (uint64_t)v;
    }
    if (n == 0) mn = 0;
    for (int v = 1; v < 256; v++) hist[v - 1] = h[v];
    vals[0] = sum; vals[1] = cnt; vals[2] = mn; vals[3] = mx;
    return 0;
}

/*
 * tools.py:473-482 for every pair k<l; matrix is N*N*3 row-major u64 filled as merger.py:175-176:
 * matrix[k][l] = (tot_k, tot_l, shared), matrix[l][k] = (tot_l, tot_k, shared).  The diagonal is
 * never assigned by the reference (merger.py:136) -- left as the caller initialised it.
 */
int pko_gram(const uint8_t *const *tables, int N, uint64_t n, int min_count, int max_count,
             uint64_t *matrix) {
    if (min_count < 1 || max_count > 255) return -1;             /* merger.py:90-91 */
    uint64_t *tot = (uint64_t *)calloc((size_t)N, sizeof(uint64_t));
    for (int i = 0; i < N; i++) {
        uint64_t t = 0;
        const uint8_t *a = tables[i];
        for (uint64_t x = 0; x < n; x++) t += (a[x] >= min_count && a[x] <= max_count);
        tot[i] = t;
    }
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++) {
            const uint8_t *a = tables[i], *b = tables[j];
            uint64_t s = 0;
            for (uint64_t x = 0; x < n; x++)
                s += ((a[x] >= min_count && a[x] <= max_count) & (b[x] >= min_count && b[x] <= max_count));
            uint64_t *ij = matrix + ((uint64_t)i * N + j) * 3, *ji = matrix + ((uint64_t)j * N + i) * 3;
            ij[0] = tot[i]; ij[1] = tot[j]; ij[2] = s;
            ji[0] = tot[j]; ji[1] = tot[i]; ji[2] = s;
        }
    free(tot);
    return 0;
}
