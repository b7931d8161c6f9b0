/*
 * synth.c -- deterministic synthetic FASTA generator (test / bench infrastructure).
 *
 * Integer-only (splitmix64 seeding + xoshiro256**), so the byte stream for a given
 * parameter set is identical on every machine.  Implements the synthetic inputs of
 * C1 (uniform + N runs + lowercase), C2 (repeat-rich
 * "genome": segmental duplicates, tandem repeats, N gaps, soft-masking) and the
 * C3/C5 family (genomes derived from one ancestor by substitutions + indels).
 *
 * This is neither the product path nor the oracle: it only manufactures inputs.
 * Build: gcc -O2 -fopenmp -shared -fPIC synth.c -o _build/libpksynth.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef struct { uint64_t s[4]; } rng_t;

static inline uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static void rng_seed(rng_t *r, uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
/* uniform in [0, n) -- multiply-shift, bias irrelevant for test data but deterministic */
static inline uint64_t rng_below(rng_t *r, uint64_t n) {
    return (uint64_t)(((__uint128_t)rng_next(r) * n) >> 64);
}

typedef struct {
    uint64_t seed;        /* ancestor seed                                           */
    uint64_t total_bp;    /* ancestor length summed over records                     */
    uint32_t n_records;
    uint32_t line_width;  /* 60                                                       */
    uint32_t pm_dup;      /* per-mille of bp in segmental duplicates                 */
    uint32_t pm_tandem;   /* per-mille in tandem / microsatellite runs               */
    uint32_t pm_ngap;     /* per-mille in N runs                                     */
    uint32_t pm_lower;    /* per-mille soft-masked lowercase                         */
    uint32_t big_gap;     /* length of one extra N gap in record 0 (0 = none)        */
    uint32_t crlf;        /* 1 = CRLF line ends                                      */
    uint64_t mut_seed;    /* derived genome: mutation stream seed (0 = ancestor)     */
    uint32_t sub_ppm;     /* substitutions per million bases                         */
    uint32_t indel_ppm;   /* indel events per million bases (len 1..8)               */
} pk_synth_params;

static const char UP[4] = {'A', 'C', 'G', 'T'};
static const char LO[4] = {'a', 'c', 'g', 't'};

/* Generate one ancestor record of exactly `len` characters into out. */
static void gen_record(const pk_synth_params *p, uint32_t rec, uint64_t len, uint8_t *out) {
    rng_t r;
    rng_seed(&r, p->seed * 0x100000001B3ULL + 0x51ED27ULL * (rec + 1));
    /* segment type weights ~ bp fraction / mean segment length (scaled integers) */
    uint64_t pm_uni = 1000 - (p->pm_dup + p->pm_tandem + p->pm_ngap + p->pm_lower);
    uint64_t w[5];
    w[0] = pm_uni        * 1000000ULL / 1100;   /* uniform  200..2000   */
    w[1] = p->pm_dup     * 1000000ULL / 25500;  /* dup     1000..50000  */
    w[2] = p->pm_tandem  * 1000000ULL / 1010;   /* tandem    20..2000   */
    w[3] = p->pm_ngap    * 1000000ULL / 250;    /* N run      1..500    */
    w[4] = p->pm_lower   * 1000000ULL / 2550;   /* lower    100..5000   */
    uint64_t wsum = w[0] + w[1] + w[2] + w[3] + w[4];
    uint64_t cur = 0;
    uint64_t gap_at = (rec == 0 && p->big_gap && len > 4ULL * p->big_gap) ? len / 2 : UINT64_MAX;
    while (cur < len) {
        if (cur >= gap_at) {                      /* the one long N gap */
            uint64_t n = p->big_gap; if (n > len - cur) n = len - cur;
            memset(out + cur, 'N', n); cur += n; gap_at = UINT64_MAX; continue;
        }
        uint64_t pick = rng_below(&r, wsum), t = 0;
        while (t < 4 && pick >= w[t]) { pick -= w[t]; t++; }
        uint64_t n;
        switch (t) {
        case 0: n = 200 + rng_below(&r, 1801); break;
        case 1: n = 1000 + rng_below(&r, 49001); break;
        case 2: n = 20 + rng_below(&r, 1981); break;
        case 3: n = 1 + rng_below(&r, 500); break;
        default: n = 100 + rng_below(&r, 4901); break;
        }
        if (n > len - cur) n = len - cur;
        if (t == 1 && cur < n + 1) t = 0;         /* nothing earlier to copy yet */
        switch (t) {
        case 0:
            for (uint64_t i = 0; i < n; ) {
                uint64_t v = rng_next(&r);
                for (int j = 0; j < 32 && i < n; j++, i++, v >>= 2) out[cur + i] = UP[v & 3];
            }
            break;
        case 1: {
            uint64_t src = rng_below(&r, cur - n + 1);
            for (uint64_t i = 0; i < n; i++) {
                uint8_t c = out[src + i];
                if (rng_below(&r, 50) == 0) c = UP[rng_next(&r) & 3];   /* 2 % substitutions */
                out[cur + i] = c;
            }
            break;
        }
        case 2: {
            static const char *motif[4] = {"A", "T", "AT", "AAG"};
            static const int mlen[4] = {1, 1, 2, 3};
            int m = (int)rng_below(&r, 4);
            for (uint64_t i = 0; i < n; i++) out[cur + i] = motif[m][i % mlen[m]];
            break;
        }
        case 3:
            memset(out + cur, 'N', n);
            break;
        default:
            for (uint64_t i = 0; i < n; ) {
                uint64_t v = rng_next(&r);
                for (int j = 0; j < 32 && i < n; j++, i++, v >>= 2) out[cur + i] = LO[v & 3];
            }
            break;
        }
        cur += n;
    }
}

/* Apply substitutions / indels; returns new length (out must hold len + len/4 + 64). */
static uint64_t mutate_record(const pk_synth_params *p, uint32_t rec, const uint8_t *in, uint64_t len,
                              uint8_t *out) {
    rng_t r;
    rng_seed(&r, p->mut_seed * 0x9E3779B1ULL + 0xA24BAED4963EE407ULL * (rec + 1));
    uint64_t o = 0;
    const uint64_t M = 1000000ULL;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t v = rng_next(&r);
        uint64_t a = (uint64_t)(((__uint128_t)(v & 0xFFFFFFFFULL) * M) >> 32);   /* [0,1e6) */
        uint64_t b = (uint64_t)(((__uint128_t)(v >> 32) * M) >> 32);
        uint8_t c = in[i];
        if (b < p->indel_ppm) {
            uint64_t w = rng_next(&r);
            uint32_t n = 1 + (uint32_t)((w >> 8) & 7);
            if (w & 1) {                               /* deletion of n bases */
                i += n - 1;
                continue;
            }
            for (uint32_t j = 0; j < n; j++, w >>= 2)  /* insertion before c  */
                out[o++] = UP[(w >> 16) & 3];
        }
        if (a < p->sub_ppm && c != 'N') {
            uint8_t nb = UP[(v >> 29) & 3];
            c = (c >= 'a') ? (uint8_t)(nb | 0x20) : nb;
        }
        out[o++] = c;
    }
    return o;
}

static uint64_t record_len(const pk_synth_params *p, uint32_t rec) {
    /* uneven, tomato-like chromosome sizes: weights 10 + 3*((rec*7)%5) */
    uint64_t wsum = 0, w = 0;
    for (uint32_t i = 0; i < p->n_records; i++) {
        uint64_t wi = 10 + 3 * ((i * 7) % 5);
        wsum += wi; if (i == rec) w = wi;
    }
    uint64_t len = p->total_bp / wsum * w + (p->total_bp % wsum) * w / wsum;
    if (rec == p->n_records - 1) {                     /* last record absorbs the rounding */
        uint64_t acc = 0;
        for (uint32_t i = 0; i + 1 < p->n_records; i++) {
            uint64_t wi = 10 + 3 * ((i * 7) % 5);
            acc += p->total_bp / wsum * wi + (p->total_bp % wsum) * wi / wsum;
        }
        len = p->total_bp - acc;
    }
    return len;
}

static uint64_t header_text(const pk_synth_params *p, uint32_t rec, uint64_t len, char *buf) {
    return (uint64_t)sprintf(buf, ">chr%02u synthetic seed=%llu mut=%llu len=%llu", rec + 1,
                             (unsigned long long)p->seed, (unsigned long long)p->mut_seed,
                             (unsigned long long)len);
}

/* Upper bound on the FASTA size for these parameters. */
uint64_t pk_synth_bound(const pk_synth_params *p) {
    uint64_t bp = p->total_bp + p->total_bp / 4 + 64ULL * p->n_records;
    uint64_t eol = p->crlf ? 2 : 1;
    return bp + (bp / p->line_width + 2ULL * p->n_records) * eol + 128ULL * p->n_records;
}

/*
 * Writes the FASTA into out (capacity cap).  Returns bytes written, 0 on failure.
 * total_bp_out (nullable) receives the number of sequence characters written.
 */
uint64_t pk_synth_fasta(const pk_synth_params *p, uint8_t *out, uint64_t cap, uint64_t *total_bp_out) {
    uint32_t R = p->n_records;
    if (R == 0 || p->line_width == 0) return 0;
    uint8_t **seq = (uint8_t **)calloc(R, sizeof(uint8_t *));
    uint64_t *len = (uint64_t *)calloc(R, sizeof(uint64_t));
    uint64_t *off = (uint64_t *)calloc(R + 1, sizeof(uint64_t));
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < R; i++) {
        uint64_t L = record_len(p, i);
        uint8_t *a = (uint8_t *)malloc(L + 1);
        if (!a) { fail = 1; continue; }
        gen_record(p, i, L, a);
        if (p->mut_seed) {
            uint8_t *b = (uint8_t *)malloc(L + L / 4 + 64);
            if (!b) { free(a); fail = 1; continue; }
            L = mutate_record(p, i, a, L, b);
            free(a); a = b;
        }
        seq[i] = a; len[i] = L;
    }
    uint64_t eol = p->crlf ? 2 : 1, W = p->line_width, bp = 0;
    char hb[160];
    for (uint32_t i = 0; i < R && !fail; i++) {
        uint64_t h = header_text(p, i, len[i], hb);
        uint64_t lines = (len[i] + W - 1) / W;
        off[i + 1] = off[i] + h + eol + len[i] + lines * eol;
        bp += len[i];
    }
    if (fail || off[R] > cap) {
        for (uint32_t i = 0; i < R; i++) free(seq[i]);
        free(seq); free(len); free(off);
        return 0;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < R; i++) {
        char hdr[160];
        uint8_t *o = out + off[i];
        uint64_t h = header_text(p, i, len[i], hdr);
        memcpy(o, hdr, h); o += h;
        if (p->crlf) *o++ = '\r';
        *o++ = '\n';
        for (uint64_t s = 0; s < len[i]; s += W) {
            uint64_t n = len[i] - s < W ? len[i] - s : W;
            memcpy(o, seq[i] + s, n); o += n;
            if (p->crlf) *o++ = '\r';
            *o++ = '\n';
        }
        free(seq[i]);
    }
    uint64_t total = off[R];
    if (total_bp_out) *total_bp_out = bp;
    free(seq); free(len); free(off);
    return total;
}
