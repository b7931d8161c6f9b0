// gram_scan.hip -- merger kernel for gfx950: all-pairs shared-k-mer tallies over N dense tables in
// ONE streaming pass.
//
// Replaces Header.calculate_distance (tools.py:439-493), which the reference runs once per PAIR
// (merger.py:139-153) re-reading both 4^k-byte tables each time: N(N-1) table reads.  Here every
// table byte is read from HBM exactly once: each lane turns 32 consecutive addresses of a table
// into one 32-bit validity mask ((count >= min) & (count <= max), tools.py:473-474; SWAR on
// dwords), and pair tallies are v_and + v_bcnt accumulates on those masks:
//   pair[i][i] = sum popc(m_i)          = total_i   (tools.py:480-481)
//   pair[i][j] = sum popc(m_i & m_j)    = shared_ij (tools.py:475,482)
// The kernel is HBM-bound (N * 4^k bytes); MFMA is deliberately not used.
//
//   k_gram_blk      N <= 128: masks of a 256-word tile staged in LDS, 8x8 table "pair blocks"
//                   spread over the waves of the workgroup, accumulators (1-2 x 64, or 2 x 32 packed) in registers.
//                   (An all-in-registers kernel for N <= 16 was measured slower: N=13 5.1 vs 6.2 TB/s.)
#include "pk_kernels.h"

namespace pk {

struct ValidParams {
    uint32_t lo_rep;   // (min & 0x7f) replicated to 4 bytes
    uint32_t lo_hi;    // min >= 128
    uint32_t up_rep;   // ((max+1) & 0x7f) replicated
    uint32_t up_hi;    // max+1 >= 128
    uint32_t has_up;   // max < 255
};

constexpr uint32_t H4 = 0x80808080u, L4 = 0x7f7f7f7fu;

// bit 7 of each byte set iff that byte is a valid count
template <bool FAST>
__device__ __forceinline__ uint32_t valid_bits(uint32_t x, const ValidParams &p) {
    if (FAST) return (((x & L4) + L4) | x) & H4;               // min=1, max=255: byte != 0
    uint32_t xl = x & L4, xh = x & H4;
    uint32_t gl = ((xl | H4) - p.lo_rep) & H4;                  // low 7 bits >= low 7 bits of min
    uint32_t ge = p.lo_hi ? (xh & gl) : (xh | gl);
    if (p.has_up) {
        uint32_t ul = ((xl | H4) - p.up_rep) & H4;
        uint32_t gu = p.up_hi ? (xh & ul) : (xh | ul);          // byte >= max+1
        ge &= ~gu;
    }
    return ge;
}

// 32 table bytes (8 dwords) -> one 32-bit mask.  Bit layout is the same for every table, which is
// all the tallies need.
template <bool FAST>
__device__ __forceinline__ uint32_t mask32(const uint4 &a, const uint4 &b, const ValidParams &p) {
    uint32_t m = valid_bits<FAST>(a.x, p) >> 7;
    m |= valid_bits<FAST>(a.y, p) >> 6;
    m |= valid_bits<FAST>(a.z, p) >> 5;
    m |= valid_bits<FAST>(a.w, p) >> 4;
    m |= valid_bits<FAST>(b.x, p) >> 3;
    m |= valid_bits<FAST>(b.y, p) >> 2;
    m |= valid_bits<FAST>(b.z, p) >> 1;
    m |= valid_bits<FAST>(b.w, p);
    return m;
}

// The 32 addresses of "word" w.  Words are numbered so that the 64 lanes of a wave (64 consecutive words)
// cover one 2 KiB block with two fully contiguous 1 KiB load instructions: lane l of block B takes bytes
// [B*2048 + l*16, +16) and [B*2048 + 1024 + l*16, +16).  Which 32 addresses share a word does not matter to
// the tallies as long as every table uses the same grouping.  Bytes at or beyond n read as 0 (never valid).
__device__ __forceinline__ uint64_t n_words_for(uint64_t n) { return ((n + 2047u) / 2048u) * 64u; }

__device__ __forceinline__ uint4 load_half(const uint8_t *t, uint64_t off, uint64_t n) {
    if (off + 16u <= n) {
        // every byte is read exactly once: stream it past the caches (nontemporal)
        // (the table pointers come out of a pointer array, so the compiler cannot tell their address space and would emit
        // FLAT loads, which count on the LDS counter as well and complete out of order: every LDS wait then also drains the
        // loads in flight.  They are global memory: say so.)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4 *gptr;
        const u32x4 x = __builtin_nontemporal_load((gptr)(t + off));
        return make_uint4(x.x, x.y, x.z, x.w);
    }
    uint32_t v[4] = {0, 0, 0, 0};
    const __attribute__((address_space(1))) uint8_t *tg = (const __attribute__((address_space(1))) uint8_t *)t;
    for (uint64_t i = off; i < n; i++) v[(i - off) >> 2] |= (uint32_t)tg[i] << (8u * ((i - off) & 3u));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void load_word(const uint8_t *t, uint64_t w, uint64_t n, uint4 &a, uint4 &b) {
    const uint64_t off = (w >> 6) * 2048u + (w & 63u) * 16u;
    a = load_half(t, off, n);
    b = load_half(t, off + 1024u, n);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// ------------------------------------------------------------------ any N: LDS-tiled ------------
constexpr int BLK = 8;            // tables per block
constexpr int TILE_WORDS = 256;   // 32-address words per tile (8 KiB of each table)
constexpr int MAX_PB = 24;        // pair blocks per launch: 12 waves x 2 slots

struct PairBlocks {               // which 8x8 table-block pairs this launch tallies
    int n;
    int8_t bi[MAX_PB], bj[MAX_PB];
};

// SLOTS pair blocks per wave: 1 (7-12 pair blocks, N 25-32: one wave each, 64 accumulators per lane, more
// waves sharing the load phase), 2 or 3 otherwise (see launch_gram).
// PACK: two 16-bit tallies per accumulator register (a lane adds at most 128 per tile to a tally, and the
// launcher keeps a workgroup below 448 tiles), halving the accumulator registers where three slots would spill.
template <int SLOTS, bool FAST, int MAXT, bool PACK>
__global__ __launch_bounds__(MAXT) void k_gram_blk(const uint8_t *const *__restrict__ tables, int N, uint64_t n, ValidParams vp,
                                                  PairBlocks pbs, unsigned long long *__restrict__ pair) {
    extern __shared__ uint32_t masks[];                 // [NB*BLK][TILE_WORDS]
    const int NB = (N + BLK - 1) / BLK;
    const int nthreads = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bi[SLOTS], bj[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        int pb = wave * SLOTS + s;
        bi[s] = pb < pbs.n ? pbs.bi[pb] : -1;
        bj[s] = pb < pbs.n ? pbs.bj[pb] : -1;
    }
    constexpr int AJ = PACK ? BLK / 2 : BLK;
    uint32_t acc[SLOTS][BLK][AJ];
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
#pragma unroll
        for (int i = 0; i < BLK; i++)
#pragma unroll
            for (int j = 0; j < AJ; j++) acc[s][i][j] = 0;

    const uint64_t n_words = n_words_for(n);
    const uint64_t n_tiles = (n_words + TILE_WORDS - 1) / TILE_WORDS;
    const int items = NB * BLK * TILE_WORDS;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = tile * TILE_WORDS;
        // phase 1: every (table, word) of the tile -> mask in LDS; consecutive lanes, consecutive words.
        // Ping-pong register sets keep 4 x 16-byte loads per thread in flight continuously: set B is
        // requested before set A is turned into masks and vice versa (the waits the compiler inserts are
        // counted, so only the older set is waited for).  The table index is wave-uniform (256 words per
        // table row), so the table pointer comes from a scalar load.
        auto fetch = [&](int q, uint4 &a, uint4 &b) -> bool {
            const int t = __builtin_amdgcn_readfirstlane(q / TILE_WORDS), w = q % TILE_WORDS;
            const bool live = q < items && t < N && w0 + w < n_words;
            a = make_uint4(0, 0, 0, 0); b = a;
            if (live) load_word(tables[t], w0 + w, n, a, b);
            return live;
        };
        {
            const int step = 2 * nthreads;
            uint4 a0, b0, a1, b1, c0, d0, c1, d1;
            int q = threadIdx.x;
            bool la0 = fetch(q, a0, b0), la1 = fetch(q + nthreads, a1, b1);
            for (; q < items; q += 2 * step) {
                const bool lc0 = fetch(q + step, c0, d0), lc1 = fetch(q + step + nthreads, c1, d1);
                if (q < items) masks[q] = la0 ? mask32<FAST>(a0, b0, vp) : 0u;
                if (q + nthreads < items) masks[q + nthreads] = la1 ? mask32<FAST>(a1, b1, vp) : 0u;
                la0 = fetch(q + 2 * step, a0, b0); la1 = fetch(q + 2 * step + nthreads, a1, b1);
                if (q + step < items) masks[q + step] = lc0 ? mask32<FAST>(c0, d0, vp) : 0u;
                if (q + step + nthreads < items) masks[q + step + nthreads] = lc1 ? mask32<FAST>(c1, d1, vp) : 0u;
            }
        }
        __syncthreads();
        // phase 2: each wave tallies its 8x8 pair blocks over the tile's words
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if (bi[s] < 0) continue;
            const uint32_t *mi = masks + bi[s] * BLK * TILE_WORDS, *mj = masks + bj[s] * BLK * TILE_WORDS;
#pragma unroll
            for (int r = 0; r < TILE_WORDS / 64; r++) {
                uint32_t a[BLK], b[BLK];
#pragma unroll
                for (int i = 0; i < BLK; i++) { a[i] = mi[i * TILE_WORDS + r * 64 + lane]; b[i] = mj[i * TILE_WORDS + r * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < BLK; i++)
#pragma unroll
                    for (int j = 0; j < AJ; j++) {
                        if (PACK) acc[s][i][j] += (uint32_t)__builtin_popcount(a[i] & b[2 * j]) + ((uint32_t)__builtin_popcount(a[i] & b[2 * j + 1]) << 16);
                        else acc[s][i][j] += __builtin_popcount(a[i] & b[j]);
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        if (bi[s] < 0) continue;
#pragma unroll
        for (int i = 0; i < BLK; i++)
#pragma unroll
            for (int j = 0; j < BLK; j++) {
                int gi = bi[s] * BLK + i, gj = bj[s] * BLK + j;
                const uint32_t mine = PACK ? (acc[s][i][j >> 1] >> (16 * (j & 1))) & 0xffffu : acc[s][i][PACK ? 0 : j];
                uint32_t v = wave_sum(mine);
                if (lane == 0 && v && gi < N && gj < N && gi <= gj) atomicAdd(&pair[gi * N + gj], (unsigned long long)v);
            }
    }
}

// ------------------------------------------------------------------ several windows in one pass --
// A threshold sweep (README.md:57-61: every --min-count / --max-count costs the reference a whole re-merge) asks for
// the same N x N tallies under W validity windows.  One pass over the tables serves them all: the 32 bytes a lane
// loads are turned into eight BIT PLANES (an 8 x 8 bit transpose across the eight dwords: 48 bit-field inserts), on
// which "count >= t" is a ripple comparator -- one three-input boolean per bit, t uniform -- so a window costs 8 or 17
// vector instructions per 32 addresses instead of the ~50 of the SWAR byte compares.  The masks of all windows go to
// LDS ([window][table][word]); a (window, pair block) combination is a wave's slot exactly like a pair block above.
// One workgroup per CU (the accumulators of two slots + the masks of up to 8 windows), so the next tile's loads are
// issued before the tallies of the current one and land during them.
struct WindowSet {     // byte w of each word belongs to window w (bytes of one scalar register: no indexed kernel arguments)
    int W;
    unsigned long long lo;     // min_count (>= 1)
    unsigned long long hi1;    // max_count + 1, or 0 when max_count = 255 (no upper bound)
    unsigned long long out;    // which N x N block of the output the window's tallies are added to
};

// rows x[0..7] (dwords), columns = bit positions inside each byte: transposed per byte lane, so that afterwards bit i
// of byte j of x[b] is bit b of byte j of the old x[i]
__device__ __forceinline__ void bit_transpose8(uint32_t (&x)[8]) {
    auto swap = [](uint32_t &r0, uint32_t &r1, uint32_t m, uint32_t s) {
        const uint32_t n0 = (r0 & m) | ((r1 << s) & ~m), n1 = ((r0 >> s) & m) | (r1 & ~m);
        r0 = n0; r1 = n1;
    };
    swap(x[0], x[1], 0x55555555u, 1); swap(x[2], x[3], 0x55555555u, 1); swap(x[4], x[5], 0x55555555u, 1); swap(x[6], x[7], 0x55555555u, 1);
    swap(x[0], x[2], 0x33333333u, 2); swap(x[1], x[3], 0x33333333u, 2); swap(x[4], x[6], 0x33333333u, 2); swap(x[5], x[7], 0x33333333u, 2);
    swap(x[0], x[4], 0x0f0f0f0fu, 4); swap(x[1], x[5], 0x0f0f0f0fu, 4); swap(x[2], x[6], 0x0f0f0f0fu, 4); swap(x[3], x[7], 0x0f0f0f0fu, 4);
}
// mask of the addresses whose count is >= t (t uniform, 1..255), from the bit planes: from the lowest bit up,
// ge = t_b ? (p_b & ge) : (p_b | ge)
__device__ __forceinline__ uint32_t planes_ge(const uint32_t (&p)[8], uint32_t t) {
    uint32_t ge = ~0u;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t nt = 0u - ((~t >> b) & 1u);                       // uniform: all ones where t_b = 0
        // majority(p_b, ge, nt) in one instruction; written out, the compiler selects between p & ge and p | ge (3 instructions)
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(ge) : "v"(p[b]), "v"(ge), "s"(nt));
    }
    return ge;
}

template <int SLOTS, int TW, int ITEMS, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT == 512 ? 4 : 1, 8))) void k_gram_mw(const uint8_t *const *__restrict__ tables, int N, uint64_t n, WindowSet ws,
                                                 unsigned long long *__restrict__ pair) {
    extern __shared__ uint32_t masks[];                 // [W][NBT][TW]
    constexpr bool PACK = SLOTS > 1;
    constexpr int RMAX = TW / 64;                        // 64-word rounds per tile
    const int NBT = ((N + BLK - 1) / BLK) * BLK;
    const int nthreads = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NB = NBT / BLK, n_pb = NB * (NB + 1) / 2;  // pair blocks (a, b), a <= b, row by row
    // A wave's slot is a (window, pair block) combination -- or, when there are fewer combinations than wave slots (few
    // windows), one ROUND of 64 words of a combination, so that the tallies of a tile are spread over all waves instead of
    // running as one long chain on a few of them (a 2-window sweep at N = 13 kept 3 of 12 waves busy in this phase).
    const int combos = ws.W * n_pb, n_waves = nthreads / 64;
    const int rs = (RMAX > 1 && combos * RMAX <= n_waves * SLOTS) ? RMAX : 1;      // rounds a combination is split into
    int bi[SLOTS], bj[SLOTS], win[SLOTS], r_lo[SLOTS], r_hi[SLOTS];
    uint32_t pm_lo[SLOTS], pm_hi[SLOTS];                 // which of the slot's 8 x 8 pairs exist (bit i * 8 + j; packed: pairs (2j, 2j+1) share bit i * 8 + 2j)
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const int u = wave * SLOTS + s;                  // work unit
        const int c = u / rs;                            // combination = (window, pair block)
        const bool on = c < combos;
        int a = 0, rem = c % n_pb;
        while (rem >= NB - a) { rem -= NB - a; a++; }
        const int ni = min(BLK, N - a * BLK), nj = min(BLK, N - (a + rem) * BLK);
        unsigned long long pm = 0;
        for (int i = 0; i < ni; i++)
            for (int j = 0; j < nj; j++)
                if (rem != 0 || j >= i) pm |= 1ull << (i * 8 + (PACK ? (j & ~1) : j));   // a diagonal block keeps its upper triangle
        win[s] = __builtin_amdgcn_readfirstlane(on ? c / n_pb : -1);     // wave-uniform: in scalar registers
        bi[s] = __builtin_amdgcn_readfirstlane(on ? a : -1);
        bj[s] = __builtin_amdgcn_readfirstlane(on ? a + rem : -1);
        r_lo[s] = __builtin_amdgcn_readfirstlane(rs > 1 ? u % rs : 0);
        r_hi[s] = __builtin_amdgcn_readfirstlane(rs > 1 ? u % rs + 1 : RMAX);
        pm_lo[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pm);
        pm_hi[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pm >> 32));
    }
    constexpr int AJ = PACK ? BLK / 2 : BLK;
    uint32_t acc[SLOTS][BLK][AJ];
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
#pragma unroll
        for (int i = 0; i < BLK; i++)
#pragma unroll
            for (int j = 0; j < AJ; j++) acc[s][i][j] = 0;

    const uint64_t n_words = n_words_for(n);
    const uint64_t n_tiles = (n_words + TW - 1) / TW;
    const int items = NBT * TW;
    // item m of this thread: table t_m (wave-uniform) and the word w_m of the tile -- both the same in every tile, and so is
    // the byte offset of the word inside the tile's 2 KiB blocks: a tile that lies wholly inside the tables is fetched
    // with a scalar base (table pointer + tile offset) and that constant lane offset, no per-lane address arithmetic
    // (the launcher's thread counts are multiples of TW, so the word -- and the offset -- is the same for all of a thread's items)
    const uint32_t w_mine = threadIdx.x % TW;
    const uint32_t voff = (w_mine >> 6) * 2048u + (w_mine & 63u) * 16u;
    uint4 ra[ITEMS], rb[ITEMS];
    // the items' table pointers, read once (scalar loads share the LDS counter: fetched inside the loop, each one would
    // wait for the mask stores in flight)
    const uint8_t *tab[ITEMS];
#pragma unroll
    for (int m = 0; m < ITEMS; m++) {
        const int q = threadIdx.x + m * nthreads;
        const int t = __builtin_amdgcn_readfirstlane(q / TW);
        const uint64_t pv = reinterpret_cast<uint64_t>(t < N ? tables[t] : nullptr);       // t < N implies q < items; all of it wave-uniform
        tab[m] = reinterpret_cast<const uint8_t *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pv >> 32)) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv));   // ... and told so: scalar registers
    }
    auto fetch_item = [&](uint64_t tile, int m) {
        const int q = threadIdx.x + m * nthreads;
        ra[m] = make_uint4(0, 0, 0, 0); rb[m] = ra[m];
        if (tab[m] == nullptr || tile >= n_tiles) return;                  // uniform
        const uint64_t byte0 = tile * (uint64_t)(RMAX * 2048);            // uniform
        if (byte0 + (uint64_t)(RMAX * 2048) <= n) {                        // uniform: nearly every tile
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) u32x4 *gptr;   // global, not flat: see load_half
            // scalar base + a 32-bit lane offset formed HERE: as a loop invariant "table + lane offset" was a register pair
            // per item, spilled in the 24-table shape, and its re-load from scratch waits for every load in flight -- the
            // prefetch of the items before it
            const uint8_t *base = tab[m] + byte0;
            uint32_t vo = voff;
            asm volatile("" : "+v"(vo));
            const u32x4 x = __builtin_nontemporal_load((gptr)(base + vo));
            const u32x4 y = __builtin_nontemporal_load((gptr)((base + vo) + 1024u));
            ra[m] = make_uint4(x.x, x.y, x.z, x.w); rb[m] = make_uint4(y.x, y.y, y.z, y.w);
        } else {
            const uint64_t w0 = tile * TW;
            const int w = q % TW;
            if (w0 + w < n_words) load_word(tab[m], w0 + w, n, ra[m], rb[m]);
        }
    };
#pragma unroll
    for (int m = 0; m < ITEMS; m++) fetch_item(blockIdx.x, m);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // phase 1: the tile's (table, word) items -> one mask per window in LDS.  An item's registers are free as soon as
        // its bit planes exist, so the SAME item of the next tile is requested right there: a whole iteration (the rest
        // of this phase, both barriers and the tallies) covers its latency, with no second set of registers.
#pragma unroll
        for (int m = 0; m < ITEMS; m++) {
            const int q = threadIdx.x + m * nthreads;
            uint32_t p[8] = {ra[m].x, ra[m].y, ra[m].z, ra[m].w, rb[m].x, rb[m].y, rb[m].z, rb[m].w};
            fetch_item(tile + gridDim.x, m);
            if (q >= items) continue;
            bit_transpose8(p);
            uint32_t ge_lo = 0, prev_lo = 0;
            // rolled: unrolled, the compiler hoists the 8 x 16 uniform bit masks of all thresholds out of the tile loop and
            // spills ~270 scalar registers; rolled, a window's masks are a dozen scalar instructions beside its 8 - 17 vector ones
#pragma unroll 1
            for (int w = 0; w < ws.W; w++) {
                const uint32_t lo = (uint32_t)(ws.lo >> (8 * w)) & 0xffu, hi1 = (uint32_t)(ws.hi1 >> (8 * w)) & 0xffu;
                if (lo != prev_lo) { ge_lo = planes_ge(p, lo); prev_lo = lo; }   // uniform: windows sorted by min share it
                uint32_t v = ge_lo;
                if (hi1) v &= ~planes_ge(p, hi1);
                masks[w * items + q] = v;
            }
        }
        __syncthreads();
        // phase 2: every wave tallies its slots over the tile's words.  Only pairs that exist are tallied: rows / columns
        // past the last table and, in a diagonal block, the lower triangle are skipped -- one scalar bit test per pair on
        // the slot's pair mask (N = 13: 91 of the 192 pair slots of its three blocks).  This phase is vector-issue bound.
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if (bi[s] < 0) continue;
            const uint32_t *mi = masks + win[s] * items + bi[s] * BLK * TW, *mj = masks + win[s] * items + bj[s] * BLK * TW;
            // opaque per tile: otherwise the 64 bit tests of the slot are hoisted out of the tile loop as 64 lane masks in
            // scalar registers, which spill (140 scalar spills, read back lane by lane inside the loop)
            uint32_t plo = pm_lo[s], phi = pm_hi[s];
            asm volatile("" : "+s"(plo), "+s"(phi));
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                if (r < r_lo[s] || r >= r_hi[s]) continue;                 // uniform: this unit's rounds
                uint32_t a[BLK], b[BLK];
#pragma unroll
                for (int i = 0; i < BLK; i++) { a[i] = mi[i * TW + r * 64 + lane]; b[i] = mj[i * TW + r * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < BLK; i++) {
                    const uint32_t row = ((i < 4 ? plo : phi) >> (8 * (i & 3))) & 0xffu;   // scalar
                    if (row == 0u) continue;
#pragma unroll
                    for (int j = 0; j < AJ; j++) {
                        if (PACK) {
                            if (!((row >> (2 * j)) & 1u)) continue;
                            acc[s][i][j] += (uint32_t)__builtin_popcount(a[i] & b[2 * j]) + ((uint32_t)__builtin_popcount(a[i] & b[2 * j + 1]) << 16);
                        } else {
                            if (!((row >> j) & 1u)) continue;
                            acc[s][i][j] += __builtin_popcount(a[i] & b[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        if (bi[s] < 0) continue;
        unsigned long long *out = pair + (size_t)((ws.out >> (8 * win[s])) & 0xffu) * N * N;
#pragma unroll
        for (int i = 0; i < BLK; i++)
#pragma unroll
            for (int j = 0; j < BLK; j++) {
                const int gi = bi[s] * BLK + i, gj = bj[s] * BLK + j;
                const uint32_t mine = PACK ? (acc[s][i][j >> 1] >> (16 * (j & 1))) & 0xffffu : acc[s][i][PACK ? 0 : j];
                const uint32_t v = wave_sum(mine);
                if (lane == 0 && v && gi < N && gj < N && gi <= gj) atomicAdd(&out[gi * N + gj], (unsigned long long)v);
            }
    }
}

// How many windows one pass of k_gram_mw takes for N tables (0: none -- use one single-window scan per window).
int gram_windows_per_pass(int N) { return N <= 16 ? 8 : N <= 24 ? 5 : N <= 32 ? 3 : 0; }

// W <= gram_windows_per_pass(N) windows (sorted by min_count: equal neighbours share their comparator); the tallies of
// window w are ADDED to the N x N block out_index[w] of dev_pair.
int launch_gram_windows(const uint8_t *const *dev_tables, int N, uint64_t n_slice, const int *min_counts, const int *max_counts,
                        const int *out_index, int W, unsigned long long *dev_pair, hipStream_t s) {
    if (N < 1 || W < 1 || W > gram_windows_per_pass(N)) return -1;
    if (n_slice == 0) return 0;
    WindowSet ws;
    ws.W = W;
    ws.lo = ws.hi1 = ws.out = 0;
    for (int w = 0; w < W; w++) {
        ws.lo |= (unsigned long long)(min_counts[w] & 0xff) << (8 * w);
        ws.hi1 |= (unsigned long long)(max_counts[w] < 255 ? max_counts[w] + 1 : 0) << (8 * w);
        ws.out |= (unsigned long long)(out_index[w] & 0xff) << (8 * w);
    }
    const int NB = (N + BLK - 1) / BLK;
    const uint64_t n_words = ((n_slice + 2047u) / 2048u) * 64u;
    const int combos = W * (NB * (NB + 1) / 2);
    static bool opted = false;
    if (!opted) {
        opted = true;
        hipFuncSetAttribute((const void *)k_gram_mw<1, 128, 2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute((const void *)k_gram_mw<2, 128, 3, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute((const void *)k_gram_mw<2, 64, 2, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    auto grid_for = [&](int tw, bool pack) {
        const uint64_t n_tiles = (n_words + tw - 1) / tw;
        uint64_t grid = n_tiles < 1024u ? n_tiles : 1024u;
        if (pack) {                                        // 16-bit packed tallies: a lane adds (tw / 64) * 32 per tile at most
            const uint64_t max_tiles = 65535u / ((uint64_t)(tw / 64) * 32u);
            if ((n_tiles + grid - 1) / grid > max_tiles) grid = (n_tiles + max_tiles - 1) / max_tiles;
        }
        return (uint32_t)grid;
    };
    // shapes (threads x loads per thread cover the tile's NBT x TW items; two slots per wave cover W x pair blocks):
    //   N <= 8    1 pair block  x 8 windows:  8 waves x 1 slot of 64 tallies,           tile of 128 words
    //   N <= 16   3 pair blocks x 8 windows: 12 waves x 2 slots of 64 packed tallies,  tile of 128 words
    //   N <= 24   6 pair blocks x 5 windows: 16 waves x 2 slots,                        tile of 64 words
    //   N <= 32  10 pair blocks x 3 windows: 16 waves x 2 slots,                        tile of 64 words
    const int NBT = NB * BLK;
    static_assert(512 % 128 == 0 && 768 % 128 == 0 && 1024 % 64 == 0, "k_gram_mw: thread counts are multiples of the tile width");
    if (N <= 8) {
        hipLaunchKernelGGL((k_gram_mw<1, 128, 2, 512>), dim3(grid_for(128, false)), dim3(512), (size_t)W * NBT * 128 * 4, s, dev_tables, N, n_slice, ws, dev_pair);
    } else if (N <= 16) {
        if (combos > 24) return -1;
        hipLaunchKernelGGL((k_gram_mw<2, 128, 3, 768>), dim3(grid_for(128, true)), dim3(768), (size_t)W * NBT * 128 * 4, s, dev_tables, N, n_slice, ws, dev_pair);
    } else {
        if (combos > 32) return -1;
        hipLaunchKernelGGL((k_gram_mw<2, 64, 2, 1024>), dim3(grid_for(64, true)), dim3(1024), (size_t)W * NBT * 64 * 4, s, dev_tables, N, n_slice, ws, dev_pair);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------ launcher --------------------
int launch_gram(const uint8_t *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count,
                unsigned long long *dev_pair, bool zero_first, hipStream_t s) {
    if (N < 1 || N > 128) return -1;                      // 128 tables x 256 words x 4 B = 128 KiB of LDS
    if (zero_first && hipMemsetAsync(dev_pair, 0, sizeof(unsigned long long) * (size_t)N * N, s) != hipSuccess) return -2;
    if (n_slice == 0) return 0;
    ValidParams vp;
    vp.lo_rep = (uint32_t)(min_count & 0x7f) * 0x01010101u;
    vp.lo_hi = min_count >= 128;
    vp.has_up = max_count < 255;
    vp.up_rep = (uint32_t)((max_count + 1) & 0x7f) * 0x01010101u;
    vp.up_hi = (max_count + 1) >= 128;
    const bool fast = (min_count == 1 && max_count == 255);
    const uint64_t n_words = ((n_slice + 2047u) / 2048u) * 64u;          // n_words_for(n_slice)
    {
        // 8x8 pair blocks, at most MAX_PB per launch; every launch streams all N tables once
        // (N <= 48: one launch; beyond that the extra launches re-read the tables).
        const int NB = (N + BLK - 1) / BLK;
        const size_t lds = (size_t)NB * BLK * TILE_WORDS * sizeof(uint32_t);
        uint64_t n_tiles = (n_words + TILE_WORDS - 1) / TILE_WORDS;
        uint32_t grid = (uint32_t)(n_tiles < 1024u ? n_tiles : 1024u);
        // packed 16-bit tallies (PACK): per tile a lane adds TILE_WORDS / 64 popcounts of at most 32 to one tally
        // (128 in all), so 448 tiles per workgroup stay below 2^16 and never carry into the neighbouring tally
        constexpr uint32_t MAX_TILES_PER_WG = 448;
        static_assert(MAX_TILES_PER_WG * (TILE_WORDS / 64) * 32 < 65536, "16-bit packed tallies would overflow");
        if ((n_tiles + grid - 1) / grid > MAX_TILES_PER_WG) grid = (uint32_t)((n_tiles + MAX_TILES_PER_WG - 1) / MAX_TILES_PER_WG);
        PairBlocks pbs;
        pbs.n = 0;
        auto flush = [&]() {
            if (!pbs.n) return;
            // Shape per number of pair blocks, measured at k=15 (TB/s):
            //    1 block   (N <= 8)    1 slot, 4 waves (3 of them only load)          5.5
            //    3 blocks  (N 9-16)    N <= 14: 1 slot, 3 waves 6.1; else 2 slots, 2 waves 6.0
            //    6 blocks  (N 17-24)   2 slots, 4 waves (one only loads)              5.8
            //   10 blocks  (N 25-32)   1 slot, 10 waves                               6.1   (2 slots: 4.2)
            //   15 blocks  (N 33-40)   2 slots, 8 waves                               5.5   (packed: 5.2)
            //   21 blocks  (N 41-48)   2 slots of packed 16-bit tallies, 11 waves     5.3   (3 slots of 32-bit tallies spill: 4.1)
            //   more: 24 blocks per launch, every launch streams all N tables again
            const int slots = pbs.n <= 3 ? (N <= 14 ? 1 : 2) : pbs.n <= 6 ? 2 : pbs.n <= 12 ? 1 : 2;
            const bool pack = pbs.n > 16;
            int waves = (pbs.n + slots - 1) / slots;
            if ((pbs.n == 1 || pbs.n == 6) && waves < 4) waves = 4;   // waves beyond the pair blocks take part in the load phase only
            int dev = 0;
            hipGetDevice(&dev);
            static size_t lds_opted[64] = {};                // per device: the dynamic-LDS limit already granted
            if (lds > 64u * 1024u && lds > lds_opted[dev & 63]) {   // opt in to more than 64 KiB of dynamic LDS, once
                lds_opted[dev & 63] = 128u * 1024u;
                const int lds_max = 128 * 1024;
                hipFuncSetAttribute((const void *)k_gram_blk<1, true, 768, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                hipFuncSetAttribute((const void *)k_gram_blk<1, false, 768, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                hipFuncSetAttribute((const void *)k_gram_blk<2, true, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                hipFuncSetAttribute((const void *)k_gram_blk<2, false, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                hipFuncSetAttribute((const void *)k_gram_blk<2, true, 768, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
                hipFuncSetAttribute((const void *)k_gram_blk<2, false, 768, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
            }
            if (slots == 1) {
                if (fast) hipLaunchKernelGGL((k_gram_blk<1, true, 768, false>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
                else hipLaunchKernelGGL((k_gram_blk<1, false, 768, false>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
            } else if (!pack) {
                if (fast) hipLaunchKernelGGL((k_gram_blk<2, true, 512, false>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
                else hipLaunchKernelGGL((k_gram_blk<2, false, 512, false>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
            } else {
                if (fast) hipLaunchKernelGGL((k_gram_blk<2, true, 768, true>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
                else hipLaunchKernelGGL((k_gram_blk<2, false, 768, true>), dim3(grid), dim3(64 * waves), lds, s, dev_tables, N, n_slice, vp, pbs, dev_pair);
            }
            pbs.n = 0;
        };
        for (int a = 0; a < NB; a++)
            for (int b = a; b < NB; b++) {
                pbs.bi[pbs.n] = (int8_t)a; pbs.bj[pbs.n] = (int8_t)b; pbs.n++;
                if (pbs.n == MAX_PB) flush();
            }
        flush();
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pk
