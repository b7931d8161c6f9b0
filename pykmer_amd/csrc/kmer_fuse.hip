// kmer_fuse.hip -- k-mer assembly fused with the level-1 partition: packed bases in, level-1 runs out.
//
// Replaces gen_kmers (indexer.py:130-160), the canonical min (indexer.py:341) and the first half of what
// np.unique + the fragment scatter do in process_kmers (indexer.py:162-297).  Input is the packed stream
// of kmer_pack.hip: per 16 KiB text chunk a slot of 2-bit codes and restart bits.  Because the stream
// holds nothing but valid bases, the k-mer ending at base j is a fixed-offset bit field of two dwords --
// no state is carried from thread to thread -- so one workgroup takes one slot, 16 or 32 bases per thread,
// and keeps its <= 16 K canonical k-mers IN REGISTERS: exactly the tile the LDS counting sort of
// part_common.h wants.  The records never exist in HBM unsorted (round 1 wrote 2.8 GB of them and read
// 3.3 GB back).
//
// Per base:  fwd = sum 4^(k-1-p) b_p   one v_alignbit on the pair-reversed dwords       (indexer.py:149)
//            rev = sum 4^p (3 - b_p)   one 64-bit shift of the complemented dwords       (indexer.py:150)
//            window valid iff no restart bit among the k-1 bases behind its first        (indexer.py:144)
// Tandem repeats (k-mers equal to the one 1, 2 or 3 bases earlier) are recognised on the packed words themselves
// (periodicity masks, see the walk loop) and tallied in the per-workgroup hot-key table of part_common.h instead of
// being emitted as records.
//
// Shape: k <= 15 runs 512 threads x 32 bases with 72 KiB of LDS, so TWO workgroups share a CU and one's
// k-mer assembly (vector ALU) overlaps the other's ranking / parking / run writes (LDS, HBM) -- with one
// 1024-thread workgroup per CU the phases ran back to back (2.4 ms instead of W + sort overlapped).  k = 17
// (64-bit k-mers, digit kept beside the record: 104 KiB) stays at 1024 x 16, one workgroup per CU.
//
// Where the runs go: bucket sizes are not known before the k-mers exist.  A sampling launch of the same
// kernel (COUNT, tally only: one wave's stretch of bases out of every 16 -- every second slot contributes an eighth of
// itself) estimates the size of every FINAL bucket (top b1+b2 address bits, <= 2^15 of them; k = 17: of every
// level-1 bucket, the final ones are then sized from a sample of the level-1 records, kmer_part.hip: k_sample2),
// k_provision lays the buckets of both levels out with 12.5 % + a constant of slack each (k_rooms2: 25 %), and the
// sorts claim room for every run from per-bucket cursors.
// A bucket that outgrows its room raises a flag (its runs go to a dump area, nothing is overwritten); every
// later kernel of the feed then returns at once and the host repeats from here with stride 1, i.e. with
// exact sizes.  Inputs below 1024 chunks are counted exactly straight away.
#include <cstdlib>
#include "part_common.h"

namespace pk {

// OR of (x << s) for s = 0 .. n-1 (n even, <= 16): which positions have a restart among the n before-or-at them
__device__ __forceinline__ uint32_t smear_up(uint32_t x, uint32_t n) {
    const uint32_t y1 = x | (x << 1), y2 = y1 | (y1 << 2), y3 = y2 | (y2 << 4);
    if (n >= 16u) return y3 | (y3 << 8);
    uint32_t acc = 0, off = 0;
    if (n & 8u) { acc |= y3; off = 8; }
    if (n & 4u) { acc |= y2 << off; off += 4; }
    if (n & 2u) { acc |= y1 << off; off += 2; }
    if (n & 1u) { acc |= x << off; }
    return acc;
}

// NT threads, PER bases each (NT * PER = 16384 = one slot); NB = LDS room for level-1 digits; HS = hot-key slots.
// COUNT: tally only -- `tally` counters in LDS ([n_tally], the final-bucket digit where there are two levels and
// <= 2^14 final buckets, the level-1 digit otherwise), written out as one row per workgroup.
// SLICED: the table holds one of 2^slice_bits address ranges; k-mers of the others are dropped, the rest are numbered
// inside the range.  DEEP (k = 19, 21; always sliced, 1024 x 16): the k-1 bases behind a thread's first one no longer
// fit one dword, so two are carried and the windows are cut from 96 bits; in front of a slot they come from the slots
// before it (the chunk state's 32 bits hold 16 bases).
template <typename KT, bool COUNT, int NT, int PER, int NB, uint32_t HS, bool SLICED, bool DEEP, uint32_t KC = 0>
__global__ __launch_bounds__(NT, 4) void k_walk_sort(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ restarts,
                                                  const uint32_t *__restrict__ n_bases, const L2 *__restrict__ chunk_l2_state,
                                                  PartPlan pl, uint32_t n_items, uint32_t stride, void *__restrict__ out,
                                                  uint32_t *__restrict__ cursor1, const uint32_t *__restrict__ cap_end, uint32_t dump,
                                                  uint32_t *__restrict__ flags, uint32_t tally_shift, uint32_t n_tally,
                                                  uint32_t *__restrict__ tally_rows, unsigned long long *__restrict__ side,
                                                  unsigned long long *__restrict__ side_n, uint64_t side_cap) {
    static_assert(NT * PER == TILE && PER % 16 == 0, "one slot per workgroup");
    static_assert(!DEEP || (PER == 16 && sizeof(KT) == 8 && SLICED), "deep windows: 64-bit k-mers, one code dword per thread, sliced table");
    constexpr int NW = PER / 16;                                          // code dwords per thread
    constexpr bool WIDE = sizeof(KT) == 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ScatterLdsT<NB> &L = *reinterpret_cast<ScatterLdsT<NB> *>(smem);
    uint32_t *tally = reinterpret_cast<uint32_t *>(smem);                 // COUNT only: the sort's LDS is not used then
    __shared__ HotTable<COUNT ? 2u : HS> hot;
    const uint32_t k = KC ? KC : pl.k, km1 = k - 1;                       // KC: k as a literal (the k = 15 instantiation)
    if (COUNT) {
        for (uint32_t i = threadIdx.x; i < n_tally; i += NT) tally[i] = 0u;
    } else {
        for (uint32_t i = threadIdx.x; i < HS; i += NT) { hot.key[i] = 0ull; hot.val[i] = 0u; }
        if (threadIdx.x == 0) hot.used = 0;
        for (uint32_t i = threadIdx.x; i < NB; i += NT) { L.hist[i] = 0; L.run[i] = 0; }
    }
    // unsliced k = 15 / 17 tables always split 7 + rest / 9 + rest (make_part_plan): digit position and count as literals
    // k as a literal pays (k = 15: 1.39 -> 1.33 ms, k = 17: 2.25 -> 2.03); the digit position as a literal on top of it
    // pays for the 64-bit kernel only (k = 15: 1.33 -> 1.37 with it)
    // (the same literals in the level-2 kernel change nothing: 1.213 ms either way)
#ifndef PK_LIT15
#define PK_LIT15 0
#endif
    const uint32_t B = KC == 17 ? 512u : (PK_LIT15 && KC == 15) ? 128u : pl.B1, shift = KC == 17 ? 25u : (PK_LIT15 && KC == 15) ? 23u : pl.addr_bits - pl.b1;
    const uint32_t low_mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
    const bool out16 = (KC == 17 || (PK_LIT15 && KC == 15)) ? false : pl.b2 == 0;
    const KT mask = (KT)((2u * k >= sizeof(KT) * 8u) ? ~(KT)0 : (((KT)1 << (2u * k)) - 1));
    const KT local_mask = (KT)((pl.addr_bits >= sizeof(KT) * 8u) ? ~(KT)0 : (((KT)1 << pl.addr_bits) - 1));   // SLICED: address inside the range
    uint32_t t = threadIdx.x;                                             // COUNT: the thread's place in the slot its WAVE samples (see locate)
    __syncthreads();
    // items: this workgroup's slots (persistent over a contiguous range), or every stride-th slot when sampling
    const uint32_t i_lo = COUNT ? blockIdx.x : blockIdx.x * pl.G1, i_hi = COUNT ? n_items : min(i_lo + pl.G1, n_items);
    const uint32_t i_step = COUNT ? gridDim.x : 1u;
    // One slot's worth of input per thread: the base count, the thread's NW code dwords and the one before them, the two
    // restart dwords, the chunk's start state.  All of it is requested one tile AHEAD (before the current tile is
    // assembled and sorted) and taken delivery of right before the current tile's runs are stored (settle): on this
    // ISA loads and stores share one in-order counter, so a load waited for after the stores would also wait for the
    // stores to be acknowledged by HBM -- a round trip per tile with nothing else in flight.  Addresses are in range
    // for every thread whatever the slot holds; what lies past the base count is masked later.
    struct Fetched { uint32_t nb, prev0, pprev0, r_here, r_before, st_flags, st_bits, cur[NW]; };
    auto fetch = [&](uint32_t c, uint32_t t, Fetched &f) {
        const uint32_t *cw = codes + (uint64_t)c * SLOT_CODE_WORDS;
        const uint32_t *rw = restarts + (uint64_t)c * SLOT_RST_WORDS;
        f.nb = n_bases[c];
#pragma unroll
        for (int w = 0; w < NW; w++) f.cur[w] = cw[NW * t + w];
        f.prev0 = cw[t ? NW * t - 1 : 0];
        f.pprev0 = cw[NW * t > 1 ? NW * t - 2 : 0];                       // the dword before prev0: deep windows, and one more base of history for the repeat test
        const uint32_t ri = NW == 2 ? t : (t >> 1);
        f.r_here = rw[ri];
        f.r_before = rw[ri ? ri - 1 : 0];
        f.st_flags = chunk_l2_state[c].flags;
        f.st_bits = chunk_l2_state[c].bits;
    };
    Fetched nxt;
    auto settle = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0); lgkmcnt / expcnt untouched
        asm volatile("" : "+v"(nxt.nb), "+v"(nxt.prev0), "+v"(nxt.pprev0), "+v"(nxt.r_here), "+v"(nxt.r_before), "+v"(nxt.st_flags), "+v"(nxt.st_bits));
#pragma unroll
        for (int w = 0; w < NW; w++) asm volatile("" : "+v"(nxt.cur[w]));
    };
    // Which slot, and which thread's place in it, this thread works on in round `it`.  The sort takes whole slots.  The
    // sampling launch takes one wave's STRETCH (64 threads' bases) of a slot per wave: stretch number g * stride + g % stride
    // of all stretches, so every second slot contributes an eighth of itself instead of every 16th slot all of itself.
    // Whole slots made a coarse sample -- a repeat array of a few hundred kbp is one or two sampled slots or none, its
    // final buckets then outgrow their room and the feed pays the exact re-layout.  Nothing in the tally loop is a
    // workgroup-wide operation, so the waves may go different ways.
    auto locate = [&](uint32_t it, uint32_t &c, uint32_t &tt) -> bool {
        if (!COUNT) { c = it; tt = threadIdx.x; return true; }
        constexpr uint32_t WPW = NT / 64;
        const uint64_t g = (uint64_t)it * WPW + (threadIdx.x >> 6);
        const uint64_t q = stride == 1u ? g : g * stride + g % stride;
        c = (uint32_t)(q / WPW); tt = (uint32_t)(q % WPW) * 64u + (threadIdx.x & 63u);
        return q / WPW < pl.n_chunks;
    };
    uint32_t c_n = 0, t_n = threadIdx.x;
    bool ok_n = false;
    if (i_lo < i_hi) { ok_n = locate(i_lo, c_n, t_n); if (ok_n) fetch(c_n, t_n, nxt); }
    settle();
#ifdef PK_PHASE_PROF
    unsigned long long phase_prof[5] = {0, 0, 0, 0, 0};                   // thread 0: assembly, count, scan, park, store (cycles)
#endif
    for (uint32_t it = i_lo; it < i_hi; it += i_step) {
#ifdef PK_PHASE_PROF
        const unsigned long long tile_t0 = __builtin_readcyclecounter();
#endif
        const Fetched me = nxt;
        const uint32_t c = c_n;
        const bool ok = ok_n;
        if (COUNT) t = t_n;
        if (it + i_step < i_hi) { ok_n = locate(it + i_step, c_n, t_n); if (ok_n) fetch(c_n, t_n, nxt); }
        if (COUNT && !ok) { settle(); continue; }                            // wave-uniform: past the last slot
        const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)me.nb);           // uniform
        if (nb == 0) { settle(); continue; }
        // ---- this thread's PER bases (NW code dwords), the 16 before them, and the restart bits of all of them
        const bool live = (uint32_t)PER * t < nb;
        uint32_t cur[NW], prev0 = me.prev0, rbits[NW];                    // rbits[w]: restart bits of the 16 bases before word w (low half) and of word w (high half)
#pragma unroll
        for (int w = 0; w < NW; w++) { cur[w] = live ? me.cur[w] : 0u; rbits[w] = 0; }
        if (NW == 2) {
            rbits[0] = (me.r_here << 16) | (me.r_before >> 16);
            rbits[NW - 1] = me.r_here;
        } else {
            rbits[0] = (t & 1u) ? me.r_here : ((me.r_here << 16) | (me.r_before >> 16));
        }
        uint32_t pprev0 = me.pprev0;
        // Is the dword before `prev` usable as three more bases of history for the repeat test?  Only where it is real
        // stream data of this slot and no restart sits on its last three bases (the restart bits a thread holds reach
        // 16 bases back; a repeat of period 3 at k = 17 looks 18 back).
        bool have_pp[NW];
        if (NW == 2) {
            have_pp[0] = t != 0 && ((me.r_before >> 13) & 7u) == 0u;
            have_pp[NW - 1] = t != 0 && ((me.r_before >> 29) & 7u) == 0u;
        } else {
            const unsigned long long rb = ((((unsigned long long)me.r_here) << 32) | me.r_before) >> (16u * (t & 1u));
            have_pp[0] = t > 1 && (((uint32_t)rb >> 13) & 7u) == 0u;
        }
        unsigned long long rb64 = 0;                                      // DEEP: restart bits of the 32 bases before the thread's (low half) and of its 16 (bits 32-47)
        if (!DEEP) {
            if (t == 0) {
                // the k-1 bases in front of the slot: the chunk's start state (newest base lowest) in stream order
                const uint32_t len = (me.st_flags >> 8) & 0xffu;          // l2_len
                prev0 = revpairs32(me.st_bits);
                rbits[0] &= 0xffff0000u;
                if (len < km1) rbits[0] |= 1u << (16u - len);             // nothing older than those `len` bases may be used
            }
        } else {
            uint32_t r_before = me.r_before;
            if (t < 2) {
                // the 32 bases in front of the slot, of which the run state says how many (len <= k-1) may be used.
                // Up to 16 are in the state's bits; more than that are the last bases of the slots before this one.
                const uint32_t len = (me.st_flags >> 8) & 0xffu;
                unsigned long long before = (unsigned long long)revpairs32(me.st_bits) << 32;     // newest base at the top
                if (len > 16u) {
                    uint32_t have = 0;
                    before = 0;
                    for (uint32_t cc = c; cc > 0 && have < 32u;) {
                        cc--;
                        const uint32_t n_cc = n_bases[cc];
                        const uint32_t take = min(n_cc, 32u - have);
                        if (take == 0) continue;
                        const uint32_t p0 = n_cc - take, d0 = p0 >> 4, sh = 2u * (p0 & 15u);
                        const uint32_t *pw = codes + (uint64_t)cc * SLOT_CODE_WORDS;
                        const unsigned long long w01 = ((unsigned long long)pw[min(d0 + 1u, SLOT_CODE_WORDS - 1u)] << 32) | pw[d0];
                        const unsigned long long w2 = pw[min(d0 + 2u, SLOT_CODE_WORDS - 1u)];
                        unsigned long long x = sh ? ((w01 >> sh) | (w2 << (64u - sh))) : w01;
                        if (take < 32u) x &= (1ull << (2u * take)) - 1ull;
                        before |= x << (2u * (32u - have - take));                                  // older bases lower
                        have += take;
                    }
                }
                const uint32_t va = (uint32_t)(before >> 32), vb = (uint32_t)before;              // the 16 right before the slot, the 16 before those
                if (t == 0) { prev0 = va; pprev0 = vb; } else { pprev0 = va; }
                r_before = (len != 0u && len < km1) ? (1u << (32u - len)) : 0u;
            }
            rb64 = ((((unsigned long long)me.r_here) << 32) | r_before) >> (16u * (t & 1u));
        }
        const uint32_t n_mine = live ? min((uint32_t)PER, nb - (uint32_t)PER * t) : 0u;

        KT r[PER];
        uint32_t okm = 0;                                                 // bit j: r[j] is a record
        // Hot keys.  Tandem repeats (poly-A/T, (AT)n, (AAG)n ...) put tens of millions of identical canonical k-mers
        // on a handful of addresses; routed like everything else they would all land in ONE final bucket, i.e. on one
        // CU.  A k-mer that equals the k-mer p bases earlier (p = 1, 2, 3) is therefore not emitted again but tallied:
        // that is the case exactly when base i-s equals base i-s-p for s = 0 .. k-1, which the packed stream shows as a
        // run of k equal-fields in  B ^ (B << 2p)  -- an AND over k shifted copies per period, for 16 bases at once,
        // instead of a cache of recent k-mers probed at every base (round 1 / earlier this round: 19 of the 27 vector
        // instructions per base).  The first p k-mers of a run are emitted, every later one adds to a tally for its
        // residue class, and the tallies go to the workgroup's LDS hash table -> side list -> k_apply_side.  Either
        // route counts each k-mer exactly once.  Where the history a thread sees (the 16 bases before its own) is too
        // short to prove a repeat, the k-mer is simply emitted.
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t prev = w ? cur[w ? w - 1 : 0] : prev0;
            const uint32_t cnt = n_mine > 16u * w ? min(16u, n_mine - 16u * w) : 0u;
            // window ending at base j of this word is void iff a restart lies among the k-1 bases after its first
            uint32_t has_x;                                                                // bit 16 + j: a restart lies among the k-1 positions before-or-at base j
            if (DEEP) {
                const unsigned long long y1 = rb64 | (rb64 << 1), y2 = y1 | (y1 << 2), y3 = y2 | (y2 << 4), y4 = y3 | (y3 << 8);
                unsigned long long acc = 0;
                uint32_t off = 0;
                if (km1 & 16u) { acc |= y4; off = 16; }
                if (km1 & 8u) { acc |= y3 << off; off += 8; }
                if (km1 & 4u) { acc |= y2 << off; off += 4; }
                if (km1 & 2u) { acc |= y1 << off; }
                has_x = (uint32_t)(acc >> 16);                                              // positions 16 back .. own, like the 32-bit form
            } else {
                has_x = smear_up(rbits[w], km1);
            }
            const uint32_t hasmask = ~(has_x >> 16) & ((1u << cnt) - 1u);
            const uint32_t fprev = revpairs32(prev), fcur = revpairs32(cur[w]);
            const unsigned long long fwd64 = ((unsigned long long)fprev << 32) | fcur;      // first base highest
            const uint32_t fpp = DEEP ? revpairs32(pprev0) : 0u;
            // complemented, first base lowest, shifted once so that the window ending at base j starts at bit 2j
            unsigned long long rev64;
            uint32_t rev_top = 0;                                                          // DEEP: what lies above rev64 after that shift
            if (DEEP) {
                const uint32_t pre = 2u * (33u - k);                                       // 28 (k = 19), 24 (k = 21)
                const unsigned long long lo = ~(((unsigned long long)prev << 32) | pprev0);
                const uint32_t top = ~cur[w];
                rev64 = (lo >> pre) | ((unsigned long long)top << (64u - pre));
                rev_top = top >> pre;
            } else {
                rev64 = (~(((unsigned long long)cur[w] << 32) | prev)) >> (2u * (17u - k));
            }
            const uint32_t rlo = (uint32_t)rev64, rhi = (uint32_t)(rev64 >> 32);
            // ---- repeats of period 1-3 among this word's 16 positions (fields: base b at bits 2b, 16 before + 16 own)
            const unsigned long long B64 = ((unsigned long long)cur[w] << 32) | prev;
            const uint32_t pp = w ? prev0 : pprev0;                                        // the 16 bases before `prev`
            auto repeats = [&](uint32_t p, uint32_t xr) -> uint32_t {                      // xr: restart among the k-2+p positions before-or-at
                // field b: base b ^ base b-p; for b < p the partner is one of the last bases of `pp`
                const unsigned long long x = B64 ^ ((B64 << (2u * p)) | (pp >> (32u - 2u * p)));
                const unsigned long long m = ~(x | (x >> 1)) & 0x5555555555555555ull & (have_pp[w] ? ~0ull : ~((1ull << (2u * p)) - 1ull));
                const unsigned long long a1 = m & (m << 2), a2 = a1 & (a1 << 4), a3 = a2 & (a2 << 8);
                unsigned long long acc = ~0ull;                                             // AND of m << 2s, s = 0 .. k-1
                uint32_t off = 0;
                if (k & 16u) { acc &= a3 & (a3 << 16); off = 16; }
                if (k & 8u) { acc &= a3 << (2u * off); off += 8; }
                if (k & 4u) { acc &= a2 << (2u * off); off += 4; }
                if (k & 2u) { acc &= a1 << (2u * off); off += 2; }
                if (k & 1u) { acc &= m << (2u * off); }
                uint32_t e = (uint32_t)(acc >> 32);                                         // this word's fields; gather the even bits
                e = (e | (e >> 1)) & 0x33333333u; e = (e | (e >> 2)) & 0x0f0f0f0fu; e = (e | (e >> 4)) & 0x00ff00ffu; e = (e | (e >> 8)) & 0xffffu;
                return e & ~xr & hasmask;
            };
            const uint32_t x1 = has_x | (has_x << 1), x2 = x1 | (x1 << 1), x3 = x2 | (x2 << 1);   // restart smear over k-1+p positions
            // (Measured and kept out, twice: a wave-uniform early-out of these three tests -- first with their own first half
            // as the pre-test, 1.284 -> 1.300 ms; then with an 8-run test on the 32-bit code words, once per thread and
            // tile, a tenth of their instructions: 1.349 -> 1.351 ms, k = 17 2.01 -> 2.08.  The tests are not what the kernel waits for.)
            const uint32_t d1 = repeats(1, x1 >> 16);
            const uint32_t d2 = repeats(2, x2 >> 16) & ~d1;
            const uint32_t d3 = repeats(3, x3 >> 16) & ~d1 & ~d2;
            const uint32_t dup = d1 | d2 | d3;
            uint32_t emit = hasmask & ~dup, in_slice = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                KT f, rv;
                if (sizeof(KT) == 4) {
                    f = (KT)__builtin_amdgcn_alignbit(fprev, fcur, 2u * (15u - j)) & mask;
                    rv = (KT)__builtin_amdgcn_alignbit(rhi, rlo, 2u * j) & mask;
                } else if (!DEEP) {
                    f = (KT)(fwd64 >> (2u * (15u - j))) & mask;
                    rv = (KT)(rev64 >> (2u * j)) & mask;
                } else {
                    f = (KT)((fwd64 >> (2u * (15u - j))) | (j < 15 ? ((unsigned long long)fpp << (64u - 2u * (15u - j))) : 0ull)) & mask;
                    rv = (KT)((rev64 >> (2u * j)) | (j > 0 ? ((unsigned long long)rev_top << (64u - 2u * j)) : 0ull)) & mask;
                }
                KT canon = f < rv ? f : rv;                                                 // indexer.py:341
                bool has = true;
                if (SLICED) {
                    has = (uint32_t)((unsigned long long)canon >> pl.addr_bits) == pl.slice_index;
                    canon &= local_mask;
                }
                r[16 * w + j] = canon;
                if (SLICED) in_slice |= has ? (1u << j) : 0u;
            }
            if (SLICED) emit &= in_slice;
            okm |= emit << (16 * w);
            // tallies of the repeats: one entry per period and residue class that occurs in this word (rare path)
            if (!COUNT && __any(dup != 0u)) {
                uint32_t left = SLICED ? (dup & in_slice) : dup;
                while (left) {                                                             // lane-divergent: at most six rounds
                    const uint32_t q = (uint32_t)__builtin_ctz(left);
                    const uint32_t p = ((d1 >> q) & 1u) ? 1u : (((d2 >> q) & 1u) ? 2u : 3u);
                    const uint32_t dp = p == 1u ? d1 : (p == 2u ? d2 : d3);
                    uint32_t cls = 0;                                                       // q, q+p, q+2p ... while they repeat: one and the same k-mer
                    for (uint32_t x = q; x < 16u && ((dp >> x) & 1u); x += p) cls |= 1u << x;
                    // the k-mer at position q, cut from the packed words again (indexing the record registers by a
                    // lane-varying q would send them all to scratch memory)
                    unsigned long long fq = fwd64 >> (2u * (15u - q)), rq = rev64 >> (2u * q);
                    if (DEEP) {
                        if (q < 15u) fq |= (unsigned long long)fpp << (64u - 2u * (15u - q));
                        if (q > 0u) rq |= (unsigned long long)rev_top << (64u - 2u * q);
                    }
                    const KT kf = (KT)fq & mask, kr = (KT)rq & mask;
                    KT key = kf < kr ? kf : kr;
                    if (SLICED) key &= local_mask;
                    hot_insert(hot, (uint64_t)key, (uint32_t)__builtin_popcount(SLICED ? (cls & in_slice) : cls), side, side_n, side_cap);
                    left &= ~cls;
                }
            }
        }
        if (COUNT) {
#pragma unroll
            for (int j = 0; j < PER; j++)
                if ((okm >> j) & 1u) atomicAdd(&tally[(uint32_t)((uint64_t)r[j] >> tally_shift)], 1u);
            settle();
            continue;
        }
        // 32-bit k-mers with a second level to come: 3-byte records in two planes (part_common.h)
        uint8_t *out_hi = (!WIDE && !out16) ? reinterpret_cast<uint8_t *>(out) + level1_hi_plane_offset(pl.capacity1) : nullptr;
#ifdef PK_PHASE_PROF
        unsigned long long *prof = phase_prof;
        if (threadIdx.x == 0) { const unsigned long long pn = __builtin_readcyclecounter(); prof[0] += pn - tile_t0; }
#else
        unsigned long long *prof = nullptr;
#endif
        bool hot_full = false;                                            // uniform: see scatter_tile (`watch`)
        scatter_tile<KT, WIDE, NT, PER, NB, false, PK_PB_L1, (WIDE ? 0 : PK_SB_L1), !WIDE>(L, r, okm, ~0u, shift, B, low_mask, out16, out, settle, cursor1, cap_end, dump, flags, out_hi, prof,
                                                                                          &hot.used, HS / 2, &hot_full);
        if (hot_full) hot_flush(hot, side, side_n, side_cap);             // (starts and ends with a barrier of its own)
    }
    if (COUNT) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_tally; i += NT) tally_rows[(uint64_t)blockIdx.x * n_tally + i] = tally[i];
        return;
    }
    hot_flush(hot, side, side_n, side_cap);
#ifdef PK_PHASE_PROF
    if (threadIdx.x == 0)                                                  // behind side_n (u64) and the flags: words 4.. of the 256-byte flag area
        for (int i = 0; i < 5; i++) atomicAdd(side_n + 4 + i, phase_prof[i]);
#endif
}

// ------------------------------------------------------------------ bucket layout ---------------
// One workgroup of 1024 threads.  `tot` holds the sampled tallies: of the final buckets (n_tally = B1 * B2 > B1) or
// of the level-1 buckets (n_tally = B1).  Room for a bucket of estimated size e: e + e/8 + slack (stride 1: the
// tally itself, which is exact or an over-count).  With final tallies both levels are laid out: final bucket
// starts + write cursors + limits, and the level-1 buckets from the sums of their final buckets' estimates.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *wsum, uint32_t &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();                                       // wsum may still be read from an earlier call
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = 0;
    total = 0;
    for (int i = 0; i < 16; i++) { if (i < w) pre += wsum[i]; total += wsum[i]; }
    return pre + inc - v;
}

// room for a bucket whose sampled tally is h: the scaled estimate + 12.5 % + slack (stride 1: h itself, exact or an over-count)
__device__ __forceinline__ uint32_t room_for(unsigned long long h, double scale, uint32_t stride, uint32_t slack) {
    if (stride == 1) return (uint32_t)h;
    const unsigned long long est = (unsigned long long)((double)h * scale) + 1ull;       // h < 2^32, scale <= 16: exact enough, and deterministic
    return (uint32_t)(est + est / 8 + slack);
}

// Final buckets are laid out by waves: wave w owns the tallies [w * seg, (w + 1) * seg), seg a multiple of 64, and
// walks them in rows of 64 (coalesced loads, all issued up front; one wave scan per row; no barrier until the sixteen
// wave totals meet).  A first version gave every thread 32 consecutive tallies in a loop: 94 us at 2^15 buckets, nearly
// all of it load latency, one after the other.
constexpr int PROV_ROWS = 32;                               // 16 waves x 32 rows x 64 = 2^15 final buckets at most
__global__ __launch_bounds__(1024) void k_provision(const uint32_t *__restrict__ tot, uint32_t n_tally, PartPlan pl, uint32_t n_sampled,
                                                    uint32_t stride, uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ cursor1,
                                                    uint32_t *__restrict__ cap_end, uint32_t *__restrict__ final_start,
                                                    uint32_t *__restrict__ cursor2, uint32_t *__restrict__ cap2_end, uint32_t *flags) {
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long sum1[512];                // sampled tallies per level-1 bucket
    const uint32_t B1 = pl.B1;
    const bool two = n_tally > B1;                          // final-bucket tallies: lay out level 2 as well
    const double scale = (double)pl.n_chunks / (double)n_sampled;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 512; i += 1024) sum1[i] = 0ull;
    __syncthreads();
    const uint32_t seg = (((n_tally + 15u) / 16u) + 63u) & ~63u;
    const uint32_t rows = seg / 64u;                        // <= PROV_ROWS (launch_provision checks)
    uint32_t h[PROV_ROWS], room[PROV_ROWS];
#pragma unroll
    for (int r = 0; r < PROV_ROWS; r++) {
        const uint32_t i = w * seg + (uint32_t)r * 64u + lane;
        h[r] = ((uint32_t)r < rows && i < n_tally) ? tot[i] : 0u;
    }
    uint32_t wave_total = 0;                                // uniform
#pragma unroll
    for (int r = 0; r < PROV_ROWS; r++) {
        room[r] = 0;
        if ((uint32_t)r < rows) {                           // uniform
            const uint32_t i = w * seg + (uint32_t)r * 64u + lane;
            const bool in = i < n_tally;
            if (two && pl.b2 >= 6u) {
                // a row of 64 final buckets lies inside one level-1 bucket: one add per row, not 64 on one LDS address
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(in ? h[r] : 0u), 63);
                if (lane == 0 && row) atomicAdd(&sum1[i >> pl.b2], (unsigned long long)row);
            } else if (in && h[r]) atomicAdd(&sum1[two ? (i >> pl.b2) : i], (unsigned long long)h[r]);
            if (two) {
                const uint32_t mine = in ? ((room_for(h[r], scale, stride, 2048u) + 7u) & ~7u) : 0u;   // 16-bit records: starts stay 16-byte aligned
                const uint32_t inc = wave_incl_scan_u32(mine);
                room[r] = mine;
                h[r] = wave_total + inc - mine;            // start inside the wave's stretch
                wave_total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
    }
    if (two) {
        if (lane == 0) wsum[w] = wave_total;
        __syncthreads();
        uint32_t pre = 0, total2 = 0;
        for (uint32_t i = 0; i < 16; i++) { if (i < w) pre += wsum[i]; total2 += wsum[i]; }
#pragma unroll
        for (int r = 0; r < PROV_ROWS; r++) {
            const uint32_t i = w * seg + (uint32_t)r * 64u + lane;
            if ((uint32_t)r < rows && i < n_tally) {
                const uint32_t a = pre + h[r];
                final_start[i] = a; cursor2[i] = a; cap2_end[i] = a + room[r];
            }
        }
        if (threadIdx.x == 0) {
            final_start[n_tally] = total2;
            if ((uint64_t)total2 > pl.capacity2) flags[0] = 1u;
        }
    }
    __syncthreads();
    uint32_t room1 = 0;
    if (threadIdx.x < B1) room1 = (room_for(sum1[threadIdx.x], scale, stride, 4096u) + 3u) & ~3u;   // 16-byte aligned starts
    uint32_t total1;
    const uint32_t base = block_excl_scan_1024(room1, wsum, total1);
    if (threadIdx.x < B1) { bucket_base[threadIdx.x] = base; cursor1[threadIdx.x] = base; cap_end[threadIdx.x] = base + room1; }
    if (threadIdx.x == 0) {
        bucket_base[B1] = total1;
        if ((uint64_t)total1 > pl.capacity1) flags[0] = 1u;               // cannot happen with the bounds of make_part_plan; be loud if it does
    }
}

// after the level-1 sort: how full every bucket got, the level-2 work split (bucket d gets ceil(n_d / R2)
// workgroups) and, for the layouts that are compact again from here on, the exclusive scan of the sizes
__global__ __launch_bounds__(1024) void k_level1_finish(const uint32_t *__restrict__ cursor1, const uint32_t *__restrict__ bucket_base,
                                                        const uint32_t *__restrict__ cap_end, PartPlan pl, uint32_t *__restrict__ bucket_end,
                                                        uint32_t *__restrict__ compact_base, uint32_t *__restrict__ wg2_start,
                                                        const uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[16];
    if (flags[0]) return;
    const uint32_t d = threadIdx.x, B1 = pl.B1;
    uint32_t size = 0;
    if (d < B1) size = min(cursor1[d], cap_end[d]) - bucket_base[d];
    const uint32_t g = (uint32_t)(((uint64_t)size + pl.R2 - 1) / pl.R2);
    uint32_t sum_n, sum_g;
    const uint32_t cb = block_excl_scan_1024(size, wsum, sum_n);
    const uint32_t ws = block_excl_scan_1024(g, wsum, sum_g);
    if (d < B1) { bucket_end[d] = bucket_base[d] + size; compact_base[d] = cb; wg2_start[d] = ws; }
    if (d == 0) { compact_base[B1] = sum_n; wg2_start[B1] = sum_g; }
    // the same running count with the buckets in XCD-class order (b % 8 major): what the persistent level-2 launch walks
    if (B1 >= 8u) {
        const uint32_t per = B1 >> 3;
        uint32_t g2 = 0;
        if (d < B1) {
            const uint32_t b = (d % per) * 8u + d / per;
            const uint32_t size2 = min(cursor1[b], cap_end[b]) - bucket_base[b];
            g2 = (uint32_t)(((uint64_t)size2 + pl.R2 - 1) / pl.R2);
        }
        uint32_t sum_g2;
        const uint32_t ps = block_excl_scan_1024(g2, wsum, sum_g2);
        uint32_t *pos = wg2_start + B1 + 1;
        if (d < B1) pos[d] = ps;
        if (d == 0) pos[B1] = sum_g2;
    }
}

// column sums of the per-workgroup tally rows of the sampling launch: one thread per column, every row (coalesced across the
// workgroup), the sum stored -- not added, so nothing has to be zeroed first
__global__ __launch_bounds__(256) void k_tally_sum(const uint32_t *__restrict__ rows, uint32_t n_rows, uint32_t n_cols, uint32_t *__restrict__ tot) {
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    if (col >= n_cols) return;
    // sixteen rows in flight per thread: the launch is 64 workgroups of latency (256 rows -> 16 round trips instead of 64)
    uint32_t acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t r = 0;
    for (; r + 16 <= n_rows; r += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc[u] += rows[(uint64_t)(r + u) * n_cols + col];
    }
    for (; r < n_rows; r++) acc[0] += rows[(uint64_t)r * n_cols + col];
    uint32_t sum = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) sum += acc[u];
    tot[col] = sum;
}

// ------------------------------------------------------------------ launchers -------------------
// k <= 15: 512 threads x 32 bases, 128 level-1 digits at most, 512 hot-key slots -> 72 KiB, two workgroups per CU
typedef ScatterLdsT<128> FuseLdsNarrow;
constexpr size_t FUSE_LDS_NARROW = offsetof(FuseLdsNarrow, dig);

// The variants: 32-bit k-mers (k <= 15) as 512 x 32, 64-bit ones (k = 17) as 1024 x 16, each for a whole table or one
// address slice; k = 19 / 21 deep windows (always a slice).  V(COUNT, ...) names the kernel.
#define PK_WS_NARROW(COUNT, SLICED) k_walk_sort<uint32_t, COUNT, 512, 32, 128, 512, SLICED, false>
#define PK_WS_K15(COUNT) k_walk_sort<uint32_t, COUNT, 512, 32, 128, 512, false, false, 15>
#define PK_WS_K17(COUNT) k_walk_sort<uint64_t, COUNT, 1024, 16, 512, 1024, false, false, 17>
#define PK_WS_WIDE(COUNT, SLICED) k_walk_sort<uint64_t, COUNT, 1024, 16, 512, 1024, SLICED, false>
#define PK_WS_DEEP(COUNT) k_walk_sort<uint64_t, COUNT, 1024, 16, 512, 1024, true, true>

void fuse_set_attributes() {
    auto set = [](const void *f, size_t bytes) { hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
    set((const void *)PK_WS_NARROW(false, false), FUSE_LDS_NARROW); set((const void *)PK_WS_NARROW(false, true), FUSE_LDS_NARROW);
    set((const void *)PK_WS_NARROW(true, false), 131072); set((const void *)PK_WS_NARROW(true, true), 131072);
    set((const void *)PK_WS_K15(false), FUSE_LDS_NARROW); set((const void *)PK_WS_K15(true), 131072);
    set((const void *)PK_WS_K17(false), SCATTER_LDS_WIDE); set((const void *)PK_WS_K17(true), 65536);
    set((const void *)PK_WS_WIDE(false, false), SCATTER_LDS_WIDE); set((const void *)PK_WS_WIDE(false, true), SCATTER_LDS_WIDE);
    set((const void *)PK_WS_WIDE(true, false), 65536); set((const void *)PK_WS_WIDE(true, true), 65536);
    set((const void *)PK_WS_DEEP(false), SCATTER_LDS_WIDE); set((const void *)PK_WS_DEEP(true), 65536);
}

// PK_K15=0: the generic instantiations for k = 15 / 17 as well (comparison runs)
static bool pk_k15_enabled() { static const bool on = !(getenv("PK_K15") && atoi(getenv("PK_K15")) == 0); return on; }

template <bool COUNT, typename... Args>
static void launch_ws(const PartPlan &pl, uint32_t grid, size_t lds, hipStream_t s, Args... args) {
    const bool sliced = pl.slice_bits != 0;
    if (pl.k > 17) hipLaunchKernelGGL((PK_WS_DEEP(COUNT)), dim3(grid), dim3(1024), lds, s, args...);
    else if (pl.k > 15) {
        if (sliced) hipLaunchKernelGGL((PK_WS_WIDE(COUNT, true)), dim3(grid), dim3(1024), lds, s, args...);
        else if (pl.k == 17 && pl.b1 == 9 && pk_k15_enabled()) hipLaunchKernelGGL((PK_WS_K17(COUNT)), dim3(grid), dim3(1024), lds, s, args...);
        else hipLaunchKernelGGL((PK_WS_WIDE(COUNT, false)), dim3(grid), dim3(1024), lds, s, args...);
    } else {
        if (sliced) hipLaunchKernelGGL((PK_WS_NARROW(COUNT, true)), dim3(grid), dim3(512), lds, s, args...);
        else if (pl.k == 15 && pl.b1 == 7 && pk_k15_enabled()) hipLaunchKernelGGL((PK_WS_K15(COUNT)), dim3(grid), dim3(512), lds, s, args...);
        else hipLaunchKernelGGL((PK_WS_NARROW(COUNT, false)), dim3(grid), dim3(512), lds, s, args...);
    }
}

// sampling launch + bucket layout.  tally_rows: COUNT_WGS x n_tally words of scratch; tally_tot: n_tally words.
void launch_provision(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl,
                      uint32_t stride, uint32_t *tally_rows, uint32_t *tally_tot, uint32_t *bucket_base, uint32_t *cursor1,
                      uint32_t *cap_end, uint32_t *final_start, uint32_t *cursor2, uint32_t *cap2_end, uint32_t *flags, hipStream_t s) {
    const uint32_t n_sampled = (pl.n_chunks + stride - 1) / stride;
    const uint32_t grid = n_sampled < COUNT_WGS ? n_sampled : COUNT_WGS;
    const uint32_t n_tally = pl.n_tally;
    const uint32_t tally_shift = n_tally > pl.B1 ? pl.fb_bits : pl.addr_bits - pl.b1;
    launch_ws<true>(pl, grid, (size_t)n_tally * 4, s, codes, restarts, n_bases, st2, pl, n_sampled, stride, (void *)nullptr, (uint32_t *)nullptr,
                    (const uint32_t *)nullptr, 0u, flags, tally_shift, n_tally, tally_rows, (unsigned long long *)nullptr,
                    (unsigned long long *)nullptr, (uint64_t)0);
    hipLaunchKernelGGL(k_tally_sum, dim3((n_tally + 255u) / 256u), dim3(256), 0, s, (const uint32_t *)tally_rows, grid, n_tally, tally_tot);
    hipLaunchKernelGGL(k_provision, dim3(1), dim3(1024), 0, s, (const uint32_t *)tally_tot, n_tally, pl, n_sampled, stride, bucket_base, cursor1, cap_end,
                       final_start, cursor2, cap2_end, flags);
}

void launch_walk_sort(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl, void *out1,
                      uint32_t *cursor1, const uint32_t *cap_end, uint32_t *flags, const uint32_t *bucket_base, uint32_t *bucket_end,
                      uint32_t *compact_base, uint32_t *wg2_start, unsigned long long *side, unsigned long long *side_n, uint64_t side_cap,
                      hipStream_t s) {
    const uint32_t dump = (uint32_t)pl.capacity1;
    launch_ws<false>(pl, pl.n_wg1, pl.k > 15 ? SCATTER_LDS_WIDE : FUSE_LDS_NARROW, s, codes, restarts, n_bases, st2, pl, pl.n_chunks, 1u, out1, cursor1,
                     cap_end, dump, flags, 0u, 0u, (uint32_t *)nullptr, side, side_n, side_cap);
    hipLaunchKernelGGL(k_level1_finish, dim3(1), dim3(1024), 0, s, (const uint32_t *)cursor1, bucket_base, cap_end, pl, bucket_end, compact_base,
                       wg2_start, (const uint32_t *)flags);
}

}  // namespace pk
