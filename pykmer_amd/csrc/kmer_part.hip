// kmer_part.hip -- table update, version 2 ("partitioned"): no HBM atomics on the count table.
//
// Version 1 (kmer_count.hip, k_count) issues one global atomic per k-mer into a 4^k-entry table; on
// MI355X scattered device-scope atomics execute at the memory side and top out near 4.6 G/s
// (measured, profiles/round01_v1_direct_*), i.e. ~4.7 Gbp/s however good the parser is.  Here the
// canonical k-mers are instead routed to the workgroup that owns their slice of the address space:
//
//   K0 k_walk_flat   FASTA -> canonical k-mers (same parser as v1), ballot-compacted per wave into
//                    fixed per-(chunk,wave) regions of a flat record array; per-workgroup histogram
//                    of the level-1 digit (top b1 address bits)
//   K1 k_rows1_scan  column scan of those histograms -> exact output offset of every (workgroup, digit)
//   K2 k_scatter1    LDS counting sort of each 16K-record tile by digit, coalesced run writes; for
//                    k <= 15 it also tallies the size of every FINAL bucket (2^14 LDS counters per workgroup)
//   K3 k_fine_sum / k_fine_scan  (k <= 15) final bucket starts + write cursors from those tallies
//      k_count2 / k_rows2_scan   (k = 17)  per-workgroup histogram of the level-2 digit (next b2 bits)
//                    inside each level-1 bucket, per-bucket column scan -> offsets + final bucket starts
//   K5 k_scatter2    second pass, 16-bit records (address inside the final bucket).  k <= 15: every tile
//                    claims room for its runs from the cursors (atomicAdd), there is no counting pass
//   K6 k_bucket_count one workgroup per final bucket of 2^16 addresses: the slice of the u8 table
//                    lives in LDS as 16-bit counters, ds_add per record, clamp, slice written to HBM
//                    (read back first when an earlier feed already wrote it)
//   K7 k_apply_side  the few hot k-mers the walk kept out of the record stream (below)
//
// Every pass is a stream: FASTA 0.8 GB + records 2.8+3.3+3.0+2.7+1.6+1.3 GB + table 1 GiB at
// k=15 / 800 Mbp.  Bucket boundaries come from histograms + scans, so no pass needs a global atomic
// per record.  With claimed runs the order of records inside a final bucket depends on timing; the
// table -- a saturating sum per address -- does not.  Saturation is exact: min(255, .) is applied
// only when a bucket's counters leave LDS (indexer.py:239,262), and K6 folds the slice already in
// HBM back in, so several feeds accumulate exactly like the reference's flushes.
#include <cstddef>
#include <cstdlib>
#include "fasta_fsm.h"
#include "kmer_walk.h"
#include "pk_kernels.h"

namespace pk {

constexpr int SUB = 4096;            // record slots per (chunk, wave) region of the flat array = PIECE * 64 lanes
constexpr int SC_T = 1024;           // threads of the scatter / bucket-count workgroups
constexpr int SC_PER = 16;           // records per thread per tile
constexpr int TILE = SC_T * SC_PER;  // 16384 records per tile = one FASTA chunk's worth

// ------------------------------------------------------------------ hot keys ---------------------
// Tandem repeats (poly-A/T, (AT)n, (AAG)n ...) put tens of millions of identical canonical k-mers on a
// handful of addresses; routed like everything else they would all land in ONE final bucket, i.e. on
// one CU.  Each lane therefore remembers its last three distinct k-mers (periods 1-3 cover poly-N,
// dinucleotide and trinucleotide repeats): a k-mer is emitted the first time it is seen, repeats while
// it is remembered only bump a lane counter, and evicted counters are tallied in a per-workgroup LDS
// hash table (addr -> count; lanes holding the same address are merged with ballot + readlane first),
// which is appended to a global side list when the workgroup finishes (or the table half fills).
// k_apply_side folds the side list into the finished u8 table with saturating CAS adds -- a few
// thousand entries instead of 10^7..10^8 records.
constexpr uint32_t HOT_SLOTS = 1024;     // per-workgroup LDS hash slots
constexpr uint32_t HOT_PROBES = 16;
constexpr uint32_t SIDE_CNT_BITS = 28;   // side entry = (addr << 28) | count

struct HotTable {
    unsigned long long key[HOT_SLOTS];   // addr + 1, 0 = empty
    uint32_t val[HOT_SLOTS];
    uint32_t used, n_flush;
};

__device__ __forceinline__ void side_append_one(unsigned long long *side, unsigned long long *side_n, uint64_t side_cap,
                                                uint64_t addr, uint32_t cnt) {
    unsigned long long i = atomicAdd(side_n, 1ull);
    if (i < side_cap) side[i] = ((unsigned long long)addr << SIDE_CNT_BITS) | cnt;
}

__device__ __forceinline__ void hot_insert(HotTable &H, uint64_t addr, uint32_t cnt, unsigned long long *side,
                                           unsigned long long *side_n, uint64_t side_cap) {
    const unsigned long long key = addr + 1ull;
    uint32_t h = (uint32_t)((addr * 0x9E3779B97F4A7C15ull) >> 40) & (HOT_SLOTS - 1u);
#pragma unroll 1                                               // rare path, inlined sixteen times into the walk loop: keep it small
    for (uint32_t t = 0; t < HOT_PROBES; t++) {
        unsigned long long old = atomicCAS(&H.key[h], 0ull, key);
        if (old == 0ull || old == key) {
            if (old == 0ull) atomicAdd(&H.used, 1u);
            atomicAdd(&H.val[h], cnt);
            return;
        }
        h = (h + 1u) & (HOT_SLOTS - 1u);
    }
    side_append_one(side, side_n, side_cap, addr, cnt);          // table crowded: straight to the side list
}

// Whole wave (uniform call): every lane with n > 0 contributes (addr, n); lanes holding the same addr
// are summed with ballot + readlane and inserted once.  Deliberately not inlined: it runs a few times
// per piece at most and would otherwise be replicated through the unrolled walk loop.
__device__ __noinline__ void hot_insert_wave(HotTable *H, unsigned long long addr, uint32_t n, unsigned long long *side,
                                             unsigned long long *side_n, uint64_t side_cap) {
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(n != 0u);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const unsigned long long a = __shfl(addr, leader, 64);
        const bool mine = n != 0u && addr == a;
        const unsigned long long same = __ballot(mine);
        uint32_t tot = mine ? n : 0u;
        for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d, 64);
        if (lane == leader) hot_insert(*H, a, tot, side, side_n, side_cap);
        pending &= ~same;
    }
}

// all threads of the workgroup; appends every occupied slot to the side list and clears the table
__device__ __forceinline__ void hot_flush(HotTable &H, unsigned long long *side, unsigned long long *side_n, uint64_t side_cap) {
    __syncthreads();
    if (threadIdx.x == 0) H.n_flush = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) mine += H.key[i] != 0ull;
    uint32_t at = mine ? atomicAdd(&H.n_flush, mine) : 0u;
    __syncthreads();
    __shared__ unsigned long long base64;
    if (threadIdx.x == 0) base64 = H.n_flush ? atomicAdd(side_n, (unsigned long long)H.n_flush) : 0ull;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) {
        if (H.key[i] != 0ull) {
            unsigned long long dst = base64 + at++;
            if (dst < side_cap) side[dst] = ((H.key[i] - 1ull) << SIDE_CNT_BITS) | H.val[i];
            H.key[i] = 0ull; H.val[i] = 0u;
        }
    }
    if (threadIdx.x == 0) H.used = 0;
    __syncthreads();
}

// ------------------------------------------------------------------ K0: walk -> flat records ----
template <typename KT, typename REC0, bool DBG>
__global__ __launch_bounds__(WG) void k_walk_flat(const uint8_t *__restrict__ fasta, uint64_t n_bytes, uint64_t stream_off,
                                                  const LaneState *__restrict__ lane_state, const L2 *__restrict__ chunk_l2_state,
                                                  PartPlan pl, REC0 *__restrict__ flat, uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ hist1_rows, DevRec *__restrict__ recs, uint64_t recs_cap,
                                                  Carry *carry, unsigned long long *__restrict__ side,
                                                  unsigned long long *__restrict__ side_n, uint64_t side_cap) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[WG * LDS_STRIDE];
    __shared__ uint32_t hist1[512];
    __shared__ HotTable hot;
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += WG) { hot.key[i] = 0ull; hot.val[i] = 0u; }
    if (threadIdx.x == 0) hot.used = 0;
    const uint32_t k = pl.k, km1 = k - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t d = threadIdx.x; d < pl.B1; d += WG) hist1[d] = 0;
    const uint32_t shift1 = pl.addr_bits - pl.b1;
    __shared__ RecAcc racc;
    recacc_init(racc);
    Walker<KT> wk;
    wk.setup(k, recs, recs_cap, &racc);
    const uint32_t c_lo = blockIdx.x * pl.G, c_hi = min(c_lo + pl.G, pl.n_chunks);
    __syncthreads();
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint64_t base = (uint64_t)c * CHUNK;
        recacc_retarget(racc, chunk_l2_state[c].rec, recs, recs_cap);    // published by the barrier below
        stage_chunk(fasta, base, n_bytes, lds);
        __syncthreads();
        const uint32_t nb = piece_len(base, n_bytes);
        // exact parser state at this lane's first byte: chunk state . lane prefix (both from the structure pass)
        const LaneState lst = lane_state[(uint64_t)c * WG + threadIdx.x];
        const L2 st2 = l2_compose(chunk_l2_state[c], lane_state_l2(lst), km1);
        const uint32_t ls_in = lane_state_ls(lst);
        const bool walk_clean = __all(!lane_state_dirty(lst) && ls_in != LS_HEADER && st2.p_tail == 0);
        wk.begin(ls_in, st2, stream_off + base + (uint64_t)threadIdx.x * PIECE);

        // wave-uniform base of this wave's record region (kept in scalar registers: stores use saddr + lane offset)
        const uint64_t region_i = ((uint64_t)c * (WG / 64) + wave) * SUB;
        REC0 *region = flat + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(region_i >> 32)) << 32) |
                               __builtin_amdgcn_readfirstlane((uint32_t)region_i));
        uint32_t wcount = 0;                               // wave-uniform
        // wave-uniform call: ballot-compact this step's records into the wave's region (coalesced store)
        const uint32_t dbg = DBG ? pl.dbg : 0u;          // ablation bits exist only in the diagnostic instantiation
        auto wave_emit = [&](bool e, KT a) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(e);   // takes the lane mask as it is (__ballot goes through an int)
            if (e) {
                // inside the branch the execution mask IS the ballot: rank among the emitting lanes from it
                const unsigned long long live = __builtin_amdgcn_read_exec();
                const uint32_t p = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
                if (!(dbg & 4u)) *reinterpret_cast<REC0 *>(reinterpret_cast<char *>(region) + p * (uint32_t)sizeof(REC0)) = (REC0)a;   // 32-bit lane offset
                const uint32_t digit = sizeof(KT) == 4 ? (uint32_t)a >> (shift1 & 31u) : (uint32_t)((uint64_t)a >> shift1);
                if (!(dbg & 2u)) atomicAdd(&hist1[digit], 1u);
            }
            wcount += __popcll(m);
        };
        // The lane remembers its last three distinct k-mers.  A k-mer is emitted the first time it is seen;
        // seeing it again while remembered (tandem repeats of period 1-3: the contended buckets) only
        // bumps a counter, which goes to the workgroup's LDS table when the entry is evicted or the piece
        // ends.  Either route counts each k-mer exactly once.  a1, a2, a3 are pairwise distinct (an entry
        // is only ever inserted on a miss; the initial ~0 is no k-mer), so at most one compare hits.
        KT a1 = ~(KT)0, a2 = ~(KT)0, a3 = ~(KT)0;
        uint32_t nn = 0;
        auto route = [&](bool has, KT canon) {
            const bool e1 = canon == a1, e2 = canon == a2, e3 = canon == a3;
            const bool h1 = has & e1, h2 = has & e2, h3 = has & e3;
            const bool miss = has & !e1 & !e2 & !e3;
            // the three counters live in one register (7 bits each are plenty: a piece has 64 steps, and
            // the entries are drained after every piece): n1 | n2 << 8 | n3 << 16
            nn += h1 ? 1u : (h2 ? 0x100u : (h3 ? 0x10000u : 0u));
            const uint32_t ev_n = miss ? (nn >> 16) : 0u;
            const KT ev_a = a3;
            a3 = miss ? a2 : a3;
            a2 = miss ? a1 : a2;
            a1 = miss ? canon : a1;
            nn = miss ? ((nn << 8) & 0xffff00u) : nn;
            wave_emit((dbg & 1u) ? has : miss, canon);
            if (ev_n != 0u) hot_insert(hot, (uint64_t)ev_a, ev_n, side, side_n, side_cap);   // rare: leaving a tandem run
        };
        if (dbg & 8u) {
        } else if (walk_clean) {                             // the common case: plain sequence lines
            wk.walk_clean(lds, nb, route);
        } else {
            for_each_byte(lds, nb, [&](uint32_t i, uint32_t ch, bool act) {
                KT canon;
                const bool has = wk.step(i, ch, act, canon);
                route(has, canon);
            });
        }
        hot_insert_wave(&hot, (unsigned long long)a1, nn & 0xffu, side, side_n, side_cap);     // drain the lane's entries
        hot_insert_wave(&hot, (unsigned long long)a2, (nn >> 8) & 0xffu, side, side_n, side_cap);
        hot_insert_wave(&hot, (unsigned long long)a3, nn >> 16, side, side_n, side_cap);
        wk.flush_rec_wave();
        if (lane == 0) cnt[c * (WG / 64) + wave] = wcount;
        __syncthreads();                                   // pieces consumed; LDS may be restaged
        if (hot.used >= HOT_SLOTS / 2) hot_flush(hot, side, side_n, side_cap);   // uniform: read after the barrier
    }
    hot_flush(hot, side, side_n, side_cap);
    wk.finish();
    recacc_finish(racc, recs, recs_cap, carry);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < pl.B1; d += WG) hist1_rows[(uint64_t)blockIdx.x * pl.B1 + d] = hist1[d];
}

// ------------------------------------------------------------------ K1: level-1 column scan -----
// One 256-thread workgroup per digit d scans that column of the (workgroup x digit) histogram:
// rowoff[w][d] = records of digit d written by rows < w.  The workgroup that finishes last (ticket
// counter) turns the column totals into bucket_base[d] = start of bucket d and the level-2 work split:
// bucket d gets ceil(n_d / R2) workgroups.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *wsum, uint32_t &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();                                       // wsum may still be read from an earlier call
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int i = 0; i < w; i++) pre += wsum[i];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return pre + inc - v;
}

__global__ __launch_bounds__(256) void k_rows1_scan(const uint32_t *__restrict__ hist_rows, uint32_t *__restrict__ rowoff, PartPlan pl,
                                                    uint32_t *col_tot, unsigned int *ticket, uint32_t *__restrict__ bucket_base,
                                                    uint32_t *__restrict__ wg2_start, uint32_t *__restrict__ final_start) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t last;
    const uint32_t B1 = pl.B1, d = blockIdx.x, n = pl.n_wg0;           // n <= 1024 rows: at most 4 per thread
    const uint32_t r_lo = min(threadIdx.x * 4u, n), r_hi = min(r_lo + 4u, n);
    uint32_t v[4], acc = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        v[j] = r_lo + j < r_hi ? hist_rows[(uint64_t)(r_lo + j) * B1 + d] : 0u;
        acc += v[j];
    }
    uint32_t total;
    uint32_t a = block_excl_scan_256(acc, wsum, total);
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
        if (r_lo + j < r_hi) { rowoff[(uint64_t)(r_lo + j) * B1 + d] = a; a += v[j]; }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&col_tot[d], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == B1 - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last workgroup: exclusive scans of the column totals (bucket starts) and of the per-bucket
    // level-2 workgroup counts, two digits per thread (B1 <= 512)
    uint32_t t0 = 0, t1 = 0;
    const uint32_t i0 = threadIdx.x * 2u, i1 = i0 + 1u;
    if (i0 < B1) t0 = __hip_atomic_load(&col_tot[i0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i1 < B1) t1 = __hip_atomic_load(&col_tot[i1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g0 = (uint32_t)((t0 + pl.R2 - 1) / pl.R2), g1 = (uint32_t)((t1 + pl.R2 - 1) / pl.R2);
    uint32_t sum_n, sum_g;
    const uint32_t base = block_excl_scan_256(t0 + t1, wsum, sum_n);
    const uint32_t wgs = block_excl_scan_256(g0 + g1, wsum, sum_g);
    if (i0 < B1) { bucket_base[i0] = base; wg2_start[i0] = wgs; if (pl.b2 == 0) final_start[i0] = base; }
    if (i1 < B1) { bucket_base[i1] = base + t0; wg2_start[i1] = wgs + g0; if (pl.b2 == 0) final_start[i1] = base + t0; }
    if (threadIdx.x == 0) {
        bucket_base[B1] = sum_n;
        wg2_start[B1] = sum_g;
        if (pl.b2 == 0) final_start[B1] = sum_n;
    }
}

// ------------------------------------------------------------------ K2 / K5: scatter ------------
// One tile = up to 16384 records: rank within digit by LDS atomic, exclusive scan of the digit
// counts, records + digits parked in LDS in sorted order, then written as coalesced runs at
// run[d] (this workgroup's running output offset for digit d).
struct ScatterLds {
    uint32_t hist[512], off[512], run[512], gbase[512];
    uint32_t wsum[SC_T / 64];
    uint32_t rec[TILE];
    uint16_t dig[TILE];            // only when the digit does not fit beside the record (k = 17, level 1)
};
constexpr size_t SCATTER_LDS_NARROW = offsetof(ScatterLds, dig);   // 72 KiB: two workgroups per CU
constexpr size_t SCATTER_LDS_WIDE = sizeof(ScatterLds);            // 104 KiB

// WIDE = the digit is kept in its own LDS array; otherwise the record parked in LDS still carries its
// digit (digit << shift | rest fits 32 bits) and is masked on the way out.
// `claim` != nullptr: the tile's run for digit d starts where a global cursor says (atomicAdd of the
// run length), instead of at this workgroup's precomputed running offset L.run[d].
//
// `settle()` is called once the tile is parked, right before its runs are stored.  The callers use it to
// take delivery of the NEXT tile's prefetched records at that point.  On this ISA loads and stores share one
// in-order counter (vmcnt): a wait placed after the stores -- where the compiler would put it, at the top
// of the next tile -- also waits for the stores to be acknowledged by HBM, a full round trip of ~8 us per
// tile with nothing else in flight.  Waiting here costs nothing (the loads were issued a whole sort ago)
// and leaves the stores in flight through the next tile's ranking and parking.
template <typename RIN, bool WIDE, class Settle>
__device__ __forceinline__ void scatter_tile(ScatterLds &L, const RIN (&r)[SC_PER], const bool (&ok)[SC_PER], uint32_t n_tile,
                                             uint32_t shift, uint32_t B, uint32_t low_mask, bool out16, void *__restrict__ out,
                                             Settle &&settle, uint32_t *claim = nullptr) {
    uint32_t dr[SC_PER];                                   // digit (9 bits) | rank inside the tile << 9
#pragma unroll
    for (int j = 0; j < SC_PER; j++) {
        dr[j] = 0;
        if (ok[j]) {
            const uint32_t dg = (uint32_t)((uint64_t)r[j] >> shift) & (B - 1u);
            dr[j] = dg | (atomicAdd(&L.hist[dg], 1u) << 9);
        }
    }
    __syncthreads();
    // exclusive scan of hist[0..B) by the first B threads (B <= 512 <= SC_T)
    uint32_t my_off = 0, claimed = 0;
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t v = threadIdx.x < B ? L.hist[threadIdx.x] : 0u, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) L.wsum[w] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (int i = 0; i < w; i++) pre += L.wsum[i];
        my_off = pre + inc - v;
        if (threadIdx.x < B) {
            L.off[threadIdx.x] = my_off;
            // a claimed run start is only needed when the runs are written: the atomic's round trip to HBM
            // overlaps the parking of the records below
            if (claim) claimed = v ? atomicAdd(&claim[threadIdx.x], v) : 0u;
            else { L.gbase[threadIdx.x] = L.run[threadIdx.x] - my_off; L.run[threadIdx.x] += v; }
            L.hist[threadIdx.x] = 0;                          // ready for the next tile
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_PER; j++)
        if (ok[j]) {
            const uint32_t dg = dr[j] & 511u;
            const uint32_t p = L.off[dg] + (dr[j] >> 9);
            if (WIDE) { L.rec[p] = (uint32_t)((uint64_t)r[j] & low_mask); L.dig[p] = (uint16_t)dg; }
            else L.rec[p] = (uint32_t)r[j];
        }
    if (claim && threadIdx.x < B) L.gbase[threadIdx.x] = claimed - my_off;   // sorted position p of digit d goes to p + gbase[d]
    __syncthreads();
    settle();
    if (out16) {
        // 16-bit records: each thread takes pairs of neighbours in sorted order and writes them as one
        // dword when they fall in the same run and the destination is even (the common case)
        uint16_t *o16 = reinterpret_cast<uint16_t *>(out);
#pragma unroll
        for (int j = 0; j < SC_PER / 2; j++) {
            const uint32_t p = 2u * (threadIdx.x + j * SC_T);
            if (p < n_tile) {
                const uint32_t r0 = L.rec[p], r1 = p + 1 < n_tile ? L.rec[p + 1] : 0u;
                const uint32_t d0 = WIDE ? L.dig[p] : (r0 >> shift) & (B - 1u);
                const uint32_t d1 = p + 1 < n_tile ? (WIDE ? (uint32_t)L.dig[p + 1] : (r1 >> shift) & (B - 1u)) : ~0u;
                const uint32_t dst0 = p + L.gbase[d0];
                if (d0 == d1 && (dst0 & 1u) == 0u) {
                    *reinterpret_cast<uint32_t *>(o16 + dst0) = (r0 & low_mask) | ((r1 & low_mask) << 16);
                } else {
                    o16[dst0] = (uint16_t)(r0 & low_mask);
                    if (p + 1 < n_tile) o16[p + 1 + L.gbase[d1]] = (uint16_t)(r1 & low_mask);
                }
            }
        }
    } else {
        // 32-bit records: neighbours in sorted order leave as one 8-byte store when they share a run
        uint32_t *o32 = reinterpret_cast<uint32_t *>(out);
#pragma unroll
        for (int j = 0; j < SC_PER / 2; j++) {
            const uint32_t p = 2u * (threadIdx.x + j * SC_T);
            if (p < n_tile) {
                const uint32_t r0 = L.rec[p], r1 = p + 1 < n_tile ? L.rec[p + 1] : 0u;
                const uint32_t d0 = WIDE ? L.dig[p] : (r0 >> shift) & (B - 1u);
                const uint32_t d1 = p + 1 < n_tile ? (WIDE ? (uint32_t)L.dig[p + 1] : (r1 >> shift) & (B - 1u)) : ~0u;
                const uint32_t dst0 = p + L.gbase[d0];
                if (d0 == d1 && (dst0 & 1u) == 0u) {
                    *reinterpret_cast<uint2 *>(o32 + dst0) = make_uint2(r0 & low_mask, r1 & low_mask);
                } else {
                    o32[dst0] = r0 & low_mask;
                    if (p + 1 < n_tile) o32[p + 1 + L.gbase[d1]] = r1 & low_mask;
                }
            }
        }
    }
    __syncthreads();
}

// level 1: source = the flat (chunk, wave) regions written by k_walk_flat; workgroup w owns the same
// chunk range as walk workgroup w, so its row of offsets applies.
// FINE: also tally, per workgroup, how many records go to every FINAL bucket (top b1+b2 address bits, at
// most 2^14 of them: 64 KiB of LDS counters behind the tile) and write that row out; summed over the
// workgroups it gives the final bucket sizes, so level 2 needs no counting pass over the records.
template <typename REC0, bool FINE>
__global__ __launch_bounds__(SC_T) void k_scatter1(const REC0 *__restrict__ flat, const uint32_t *__restrict__ cnt,
                                                   const uint32_t *__restrict__ rowoff, const uint32_t *__restrict__ bucket_base,
                                                   PartPlan pl, void *__restrict__ out, uint32_t *__restrict__ fine_rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ScatterLds &L = *reinterpret_cast<ScatterLds *>(smem);
    uint32_t *fine = reinterpret_cast<uint32_t *>(smem + SCATTER_LDS_NARROW);      // [B1 * B2], FINE only
    const uint32_t n_fine = pl.B1 * pl.B2;
    if (FINE) for (uint32_t i = threadIdx.x; i < n_fine; i += SC_T) fine[i] = 0u;
    const uint32_t B = pl.B1, shift = pl.addr_bits - pl.b1;
    const uint32_t low_mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
    const bool out16 = pl.b2 == 0;
    if (threadIdx.x < 512) { L.hist[threadIdx.x] = 0; L.run[threadIdx.x] = 0; }
    __syncthreads();
    if (threadIdx.x < B) L.run[threadIdx.x] = bucket_base[threadIdx.x] + rowoff[(uint64_t)blockIdx.x * B + threadIdx.x];
    __syncthreads();
    const uint32_t c_lo = blockIdx.x * pl.G, c_hi = min(c_lo + pl.G, pl.n_chunks);
    // one tile = the four wave regions of one chunk (SUB slots each, n[s] of them filled).  Thread t takes
    // slots 4t .. 4t+3 of every region with one 16-byte (32-byte for 64-bit records) load; slots past
    // n[s] are allocated but hold no record -- they are read anyway and masked, which keeps the loads
    // branch-free.
    static_assert(SUB == 4 * SC_T && SC_PER == 16, "tile layout");
    typedef uint32_t Counts __attribute__((ext_vector_type(4)));         // records in the chunk's four regions
    auto meta = [&](uint32_t c) { return *reinterpret_cast<const Counts *>(cnt + (uint64_t)c * 4); };
    typedef REC0 Quad __attribute__((ext_vector_type(4)));               // one load, one register tuple, one asm operand
    auto fetch = [&](uint32_t c, Quad (&q)[4]) {
        const Quad *src = reinterpret_cast<const Quad *>(flat + (uint64_t)c * 4 * SUB) + threadIdx.x;
#pragma unroll
        for (int sreg = 0; sreg < 4; sreg++) q[sreg] = src[sreg * (SUB / 4)];
    };
    // The next chunk's loads are issued before the current tile is sorted, so they fly during its barriers;
    // settle() (see scatter_tile) takes delivery of them before the current tile's stores are issued.  The
    // empty asm makes the registers "defined here" for the compiler, so it adds no wait of its own later.
    Quad nxt[4] = {};
    Counts nxt_n = {0u, 0u, 0u, 0u};
    auto settle = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0); lgkmcnt / expcnt untouched
        asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt_n));
    };
    if (c_lo < c_hi) { nxt_n = meta(c_lo); fetch(c_lo, nxt); }
    settle();
    for (uint32_t c = c_lo; c < c_hi; c++) {
        REC0 r[SC_PER];
        const uint32_t tile_n[4] = {nxt_n.x, nxt_n.y, nxt_n.z, nxt_n.w};
        const uint32_t tile_total = tile_n[0] + tile_n[1] + tile_n[2] + tile_n[3];
#pragma unroll
        for (int j = 0; j < SC_PER; j++) r[j] = nxt[j >> 2][j & 3];
        if (c + 1 < c_hi) { nxt_n = meta(c + 1); fetch(c + 1, nxt); }
        if (tile_total == 0) { settle(); continue; }
        bool ok[SC_PER];
#pragma unroll
        for (int j = 0; j < SC_PER; j++) ok[j] = threadIdx.x * 4u + (j & 3) < tile_n[j >> 2];
        if (FINE) {
#pragma unroll
            for (int j = 0; j < SC_PER; j++)
                if (ok[j]) atomicAdd(&fine[(uint32_t)((uint64_t)r[j] >> pl.fb_bits)], 1u);
        }
        scatter_tile<REC0, sizeof(REC0) == 8>(L, r, ok, tile_total, shift, B, low_mask, out16, out, settle);
    }
    if (FINE) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_fine; i += SC_T) fine_rows[(uint64_t)blockIdx.x * n_fine + i] = fine[i];
    }
}

// column sums of the per-workgroup final-bucket tallies: grid (n_fine / 256, row groups)
__global__ __launch_bounds__(256) void k_fine_sum(const uint32_t *__restrict__ fine_rows, uint32_t n_rows, uint32_t n_fine,
                                                  uint32_t *__restrict__ fine_tot) {
    const uint32_t col = blockIdx.x * 256u + threadIdx.x;
    if (col >= n_fine) return;
    const uint32_t per = (n_rows + gridDim.y - 1) / gridDim.y;
    const uint32_t r_lo = blockIdx.y * per, r_hi = min(r_lo + per, n_rows);
    uint32_t acc = 0;
    for (uint32_t r = r_lo; r < r_hi; r++) acc += fine_rows[(uint64_t)r * n_fine + col];
    if (acc) atomicAdd(&fine_tot[col], acc);
}

// exclusive scan of the final bucket sizes -> final_start[0 .. n_fine] and the write cursors level 2 claims from
__global__ __launch_bounds__(1024) void k_fine_scan(const uint32_t *__restrict__ fine_tot, uint32_t n_fine,
                                                    uint32_t *__restrict__ final_start, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t wsum[16];
    const uint32_t per = (n_fine + 1023u) / 1024u;                       // <= 16
    const uint32_t lo = min(threadIdx.x * per, n_fine), hi = min(lo + per, n_fine);
    uint32_t acc = 0;
    for (uint32_t i = lo; i < hi; i++) acc += fine_tot[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t a = inc - acc;
    for (int i = 0; i < w; i++) a += wsum[i];
    for (uint32_t i = lo; i < hi; i++) { final_start[i] = a; cursor[i] = a; a += fine_tot[i]; }
    if (threadIdx.x == 1023) final_start[n_fine] = a;
}

// level-2 work split: bucket b is covered by workgroups wg2_start[b] .. wg2_start[b+1]-1, R2 records each
__device__ __forceinline__ bool wg2_range(const uint32_t *__restrict__ wg2_start, const uint32_t *__restrict__ bucket_base,
                                          const PartPlan &pl, uint32_t &b, uint32_t &lo, uint32_t &hi) {
    const uint32_t w = blockIdx.x;
    if (w >= wg2_start[pl.B1]) return false;
    uint32_t a = 0, z = pl.B1;                              // last b with wg2_start[b] <= w
    while (z - a > 1) { uint32_t m = (a + z) >> 1; if (wg2_start[m] <= w) a = m; else z = m; }
    b = a;
    uint64_t s = (uint64_t)bucket_base[b] + (uint64_t)(w - wg2_start[b]) * pl.R2;
    uint64_t e = s + pl.R2;
    if (e > bucket_base[b + 1]) e = bucket_base[b + 1];
    lo = (uint32_t)s; hi = (uint32_t)e;
    return true;
}

__global__ __launch_bounds__(WG) void k_count2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ wg2_start,
                                               const uint32_t *__restrict__ bucket_base, PartPlan pl, uint32_t *__restrict__ hist_rows) {
    __shared__ uint32_t h[512];
    for (uint32_t d = threadIdx.x; d < pl.B2; d += WG) h[d] = 0;
    __syncthreads();
    uint32_t b, lo, hi;
    const bool live = wg2_range(wg2_start, bucket_base, pl, b, lo, hi);
    const uint32_t shift = pl.fb_bits;
    if (live) {
        const uint32_t mask = pl.B2 - 1u;
        for (uint32_t i = (lo & ~3u) + threadIdx.x * 4u; i < hi; i += WG * 4u) {      // 16 B per lane; edges masked
            const uint4 v = *reinterpret_cast<const uint4 *>(in + i);
            const uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (i + q >= lo && i + q < hi) atomicAdd(&h[(r[q] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < pl.B2; d += WG) hist_rows[(uint64_t)blockIdx.x * pl.B2 + d] = live ? h[d] : 0u;
}

// K4: one workgroup per level-1 bucket, one thread per level-2 digit.
__global__ __launch_bounds__(512) void k_rows2_scan(const uint32_t *__restrict__ hist_rows, uint32_t *__restrict__ rowoff,
                                                    const uint32_t *__restrict__ wg2_start, const uint32_t *__restrict__ bucket_base,
                                                    PartPlan pl, uint32_t *__restrict__ final_start) {
    __shared__ uint32_t tot[512];
    const uint32_t b = blockIdx.x, d = threadIdx.x;
    uint32_t acc = 0;
    if (d < pl.B2)
        for (uint32_t w = wg2_start[b]; w < wg2_start[b + 1]; w++) {
            uint32_t v = hist_rows[(uint64_t)w * pl.B2 + d];
            rowoff[(uint64_t)w * pl.B2 + d] = acc;
            acc += v;
        }
    tot[d] = d < pl.B2 ? acc : 0;
    __syncthreads();
    if (d == 0) {
        uint32_t a = bucket_base[b];
        for (uint32_t i = 0; i < pl.B2; i++) { uint32_t n = tot[i]; final_start[(uint64_t)b * pl.B2 + i] = a; a += n; }
        if (b == pl.B1 - 1) final_start[(uint64_t)pl.B1 * pl.B2] = a;
    }
}

// CLAIM: no precomputed offsets; every tile claims room for its runs from the final buckets' cursors.  Where a
// record lands inside its final bucket then depends on timing -- the bucket's contents as a multiset do not.
template <bool CLAIM>
__global__ __launch_bounds__(SC_T) void k_scatter2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ wg2_start,
                                                   const uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ rowoff,
                                                   const uint32_t *__restrict__ final_start, PartPlan pl, void *__restrict__ out,
                                                   uint32_t *__restrict__ cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ScatterLds &L = *reinterpret_cast<ScatterLds *>(smem);
    uint32_t b, lo, hi;
    if (!wg2_range(wg2_start, bucket_base, pl, b, lo, hi)) return;      // uniform per workgroup
    const uint32_t B = pl.B2, shift = pl.fb_bits;
    const uint32_t low_mask = (1u << shift) - 1u;
    if (threadIdx.x < 512) { L.hist[threadIdx.x] = 0; L.run[threadIdx.x] = 0; }
    __syncthreads();
    if (!CLAIM && threadIdx.x < B) L.run[threadIdx.x] = final_start[(uint64_t)b * B + threadIdx.x] + rowoff[(uint64_t)blockIdx.x * B + threadIdx.x];
    __syncthreads();
    // 16-byte aligned windows of TILE records over [lo, hi); the first / last window are partly masked.
    // The next window's loads are issued before the current tile is sorted, so they fly during its barriers.
    typedef uint32_t Quad __attribute__((ext_vector_type(4)));
    auto fetch = [&](uint32_t win, Quad (&v)[SC_PER / 4]) {
        const uint32_t v_hi = min(hi, win + (uint32_t)TILE);
#pragma unroll
        for (int j = 0; j < SC_PER / 4; j++) {
            const uint32_t i = win + (threadIdx.x + j * SC_T) * 4u;
            v[j] = Quad{0u, 0u, 0u, 0u};
            if (i < v_hi) v[j] = *reinterpret_cast<const Quad *>(in + i);
        }
    };
    Quad nxt[SC_PER / 4];
    auto settle = [&]() {                                                // see scatter_tile / k_scatter1
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]));
    };
    fetch(lo & ~3u, nxt);
    settle();
    for (uint32_t win = lo & ~3u; win < hi; win += TILE) {
        const uint32_t v_lo = max(lo, win), v_hi = min(hi, win + (uint32_t)TILE);
        const uint32_t n_tile = v_hi - v_lo;
        uint32_t r[SC_PER];
        bool ok[SC_PER];
#pragma unroll
        for (int j = 0; j < SC_PER / 4; j++) {
            const uint32_t i = win + (threadIdx.x + j * SC_T) * 4u;
            const uint32_t q[4] = {nxt[j].x, nxt[j].y, nxt[j].z, nxt[j].w};
#pragma unroll
            for (int e = 0; e < 4; e++) { ok[j * 4 + e] = i + e >= v_lo && i + e < v_hi; r[j * 4 + e] = q[e]; }
        }
        if (win + TILE < hi) fetch(win + TILE, nxt);
        scatter_tile<uint32_t, false>(L, r, ok, n_tile, shift, B, low_mask, true, out, settle, CLAIM ? cursor + (uint64_t)b * B : nullptr);
    }
}

// ------------------------------------------------------------------ K6: count in LDS ------------
// One workgroup per final bucket (2^fb_bits addresses, fb_bits <= 16).  Counters are 16 bit, two per
// LDS dword; a bucket with more than 65024 records is folded in pieces with a clamp between them so a
// counter (<= 255 + 65024) can never carry into its neighbour.
constexpr uint32_t K6_PIECE = 65024;   // multiple of 8

// `fresh` = first feed after a reset: the table holds nothing yet (it is not even zeroed), so slices
// are not read back and buckets without records are written as zeros.
//
// The value histogram behind Header.update_stats (tools.py:246-263) is maintained here as well: each
// workgroup writes the net change it made as one row of 256 signed counters, k_hist_reduce adds the
// rows to the running histogram, and finish() needs no pass over the table (3.3 ms at k=17).  Two ways
// of getting that change, chosen per bucket:
//   sparse bucket (records < addresses/4, the k=17 case): from the record side -- the LDS add returns
//     the counter's previous value, so every add knows which bins it moves a k-mer between; counters
//     preloaded from an earlier feed's slice make the change relative to what the table already held;
//   dense bucket (the k=15 case): histogram of the slice stored minus histogram of the slice loaded,
//     tallied on the packed dwords (bytes equal to 1 and 2 by SWAR test + popcount).
// Bins 1 and 2 -- nearly everything -- live in two lane registers; the rest goes to 256 LDS bins.
__device__ __forceinline__ int wave_sum_i32(int v) {
    // rotate-and-add inside each row of 16 lanes (DPP row_ror 8/4/2/1), then add the four row sums
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

struct SliceTally {
    int d1 = 0, d2 = 0;
    __device__ __forceinline__ void add_dword(int *dh, uint32_t x, int sign) {
        const uint32_t z = swar_zero(x), one = swar_zero(x ^ 0x01010101u), two = swar_zero(x ^ 0x02020202u);
        d1 += sign * (int)__builtin_popcount(one);
        d2 += sign * (int)__builtin_popcount(two);
        uint32_t rest = ~(z | one | two) & 0x80808080u;
        while (rest) {
            const int bit = __ffs(rest) - 1;                             // 7, 15, 23 or 31
            atomicAdd(&dh[(x >> (bit - 7)) & 0xffu], sign);
            rest &= rest - 1u;
        }
    }
};

// `split_bits` > 0 (sparse tables, k=17): a bucket is shared by 2^split_bits workgroups, each reading all of the
// bucket's (few) records but counting only its own part of the address range -- the LDS counters shrink
// with the part, so several workgroups fit on a CU and hide each other's phases.
template <int T>
__device__ __forceinline__ void bucket_count_body(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start,
                                                  uint32_t fb_bits, uint32_t split_bits, uint8_t *__restrict__ table8, uint32_t fresh,
                                                  int *__restrict__ bucket_hist, uint8_t *smem, int *dh) {
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                  // 2^fb_bits / 2 dwords
    const uint32_t fb = blockIdx.x >> split_bits, part = blockIdx.x & ((1u << split_bits) - 1u);
    const uint32_t start = final_start[fb], end = final_start[fb + 1];
    const uint32_t part_bits = fb_bits - split_bits;
    const uint32_t n_addr = 1u << part_bits;                             // addresses this workgroup owns
    uint8_t *slice = table8 + ((uint64_t)fb << fb_bits) + (uint64_t)part * n_addr;
    if (start == end) {
        if (fresh) {                                                     // nothing counted here: the slice is all zero
            if (n_addr >= 16) for (uint32_t g = threadIdx.x; g < n_addr / 16; g += T) reinterpret_cast<uint4 *>(slice)[g] = make_uint4(0, 0, 0, 0);
            else for (uint32_t a = threadIdx.x; a < n_addr; a += T) slice[a] = 0;
        }
        for (uint32_t i = threadIdx.x; i < 256; i += T) bucket_hist[(uint64_t)blockIdx.x * 256 + i] = 0;   // no change to the histogram
        return;                                                          // otherwise the slice stays as it is
    }
    // the first 16 bytes of records every lane will need are requested before the counters are set up, so the
    // load's latency hides behind that phase (sparse buckets need no second load at all)
    const uint32_t base = start & ~7u;                                   // 16-byte aligned vector loads
    const uint32_t i_first = base + threadIdx.x * 8;
    uint4 v_first = make_uint4(0, 0, 0, 0);
    if (i_first < min(base + K6_PIECE, end)) v_first = *reinterpret_cast<const uint4 *>(recs + i_first);
    const bool by_rec = ((end - start) >> split_bits) < n_addr / 4 || n_addr < 16;         // sparse bucket: histogram change from the adds
    SliceTally tally;
    for (uint32_t i = threadIdx.x; i < 256; i += T) dh[i] = 0;
    if (fresh) {
        for (uint32_t a = threadIdx.x; a < max(n_addr / 2, 1u); a += T) cnt[a] = 0u;
    } else if (n_addr >= 16) {                                           // fold in what the slice already holds (earlier feeds)
        if (!by_rec) __syncthreads();                                    // dh zeroed before anyone subtracts from it
        for (uint32_t g = threadIdx.x; g < n_addr / 16; g += T) {
            uint4 v = reinterpret_cast<const uint4 *>(slice)[g];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            if (!by_rec) {
#pragma unroll
                for (int q = 0; q < 4; q++) tally.add_dword(dh, w[q], -1);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t lo = (w[q >> 1] >> (16 * (q & 1))) & 0xffu, hi = (w[q >> 1] >> (16 * (q & 1) + 8)) & 0xffu;
                cnt[g * 8 + q] = lo | (hi << 16);
            }
        }
    } else {
        for (uint32_t a = threadIdx.x; a < n_addr / 2; a += T) cnt[a] = slice[2 * a] | ((uint32_t)slice[2 * a + 1] << 16);
    }
    __syncthreads();
    auto bump = [&](uint32_t a, uint32_t n) {
        const uint32_t sh = 16u * (a & 1u);
        if (!by_rec) { atomicAdd(&cnt[a >> 1], n << sh); return; }
        int &d1 = tally.d1, &d2 = tally.d2;
        const uint32_t c = (atomicAdd(&cnt[a >> 1], n << sh) >> sh) & 0xffffu;
        const uint32_t oc = c > 255u ? 255u : c, nc = c + n > 255u ? 255u : c + n;
        if (oc == nc) return;                                            // already saturated
        if (nc == 1u) d1++;
        else if (nc == 2u && oc == 1u) { d2++; d1--; }
        else {
            atomicAdd(&dh[nc], 1);
            if (oc) atomicAdd(&dh[oc], -1);
        }
    };
    // the lane's 8 records starting at index i, equal neighbours merged (what is left of tandem runs arrives
    // back to back)
    auto count8 = [&](const uint4 &v, uint32_t i) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t pa = 0, pn = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = i + q;
            const uint32_t full = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            const bool in = idx >= start && idx < end && (full >> part_bits) == part;
            const uint32_t a = full & (n_addr - 1u);
            if (in && pn && a == pa) pn++;
            else {
                if (pn) bump(pa, pn);
                pa = a; pn = in ? 1u : 0u;
            }
        }
        if (pn) bump(pa, pn);
    };
    // The same for a sparse bucket (histogram from the record side): the eight adds are issued back to back without
    // looking at each other -- records outside the bucket or the part add zero to their counter -- and their
    // returned values are evaluated afterwards: one LDS round trip instead of eight dependent ones.  Equal
    // neighbours are not merged here; their adds return consecutive values, which moves the histogram the same way.
    auto count8_sparse = [&](const uint4 &v, uint32_t i) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t old[8];
        bool in[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = i + q;
            const uint32_t full = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            in[q] = idx >= start && idx < end && (full >> part_bits) == part;
            const uint32_t a = full & (n_addr - 1u), sh = 16u * (a & 1u);
            old[q] = (atomicAdd(&cnt[a >> 1], in[q] ? (1u << sh) : 0u) >> sh) & 0xffffu;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t c = old[q];
            if (!in[q] || c >= 255u) continue;                           // nothing added, or already saturated
            if (c == 0u) tally.d1++;
            else if (c == 1u) { tally.d2++; tally.d1--; }
            else { atomicAdd(&dh[c + 1u], 1); atomicAdd(&dh[c], -1); }
        }
    };
    constexpr int NIT = (K6_PIECE + T * 8 - 1) / (T * 8);               // 16-byte loads per lane and piece
    if (end - base <= (uint32_t)T * 8u) {                                // sparse bucket: the hoisted load was all of it
        if (i_first < end) { if (by_rec) count8_sparse(v_first, i_first); else count8(v_first, i_first); }
        __syncthreads();
    } else {
        for (uint32_t p0 = base; p0 < end; p0 += K6_PIECE) {
            const uint32_t p1 = min(p0 + K6_PIECE, end);
            // all of the piece's loads are issued before the first record is counted: one memory latency per
            // piece instead of one per 8 records
            uint4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const uint32_t i = p0 + (threadIdx.x + it * T) * 8;
                v[it] = make_uint4(0, 0, 0, 0);
                if (i < p1) v[it] = (it == 0 && p0 == base) ? v_first : *reinterpret_cast<const uint4 *>(recs + i);
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const uint32_t i = p0 + (threadIdx.x + it * T) * 8;
                if (i < p1) count8(v[it], i);
            }
            __syncthreads();
            if (p1 < end) {                                              // more to come: clamp so nothing can overflow
                for (uint32_t a = threadIdx.x; a < max(n_addr / 2, 1u); a += T) {
                    uint32_t x = cnt[a], lo = x & 0xffffu, hi = x >> 16;
                    cnt[a] = (lo > 255u ? 255u : lo) | ((hi > 255u ? 255u : hi) << 16);
                }
                __syncthreads();
            }
        }
    }
    // clamp to u8 and write the slice back, 16 addresses per lane
    if (n_addr >= 16) {
        for (uint32_t g = threadIdx.x; g < n_addr / 16; g += T) {
            uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t x = cnt[g * 8 + q], lo = x & 0xffffu, hi = x >> 16;
                lo = lo > 255u ? 255u : lo; hi = hi > 255u ? 255u : hi;
                o[q >> 1] |= (lo | (hi << 8)) << (16 * (q & 1));
            }
            if (!by_rec) {
#pragma unroll
                for (int q = 0; q < 4; q++) tally.add_dword(dh, o[q], 1);
            }
            reinterpret_cast<uint4 *>(slice)[g] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (uint32_t a = threadIdx.x; a < n_addr / 2; a += T) {
            uint32_t x = cnt[a], lo = x & 0xffffu, hi = x >> 16;
            slice[2 * a] = (uint8_t)(lo > 255u ? 255u : lo);
            slice[2 * a + 1] = (uint8_t)(hi > 255u ? 255u : hi);
        }
    }
    const int d1 = wave_sum_i32(tally.d1), d2 = wave_sum_i32(tally.d2);
    if ((threadIdx.x & 63) == 0) {
        if (d1) atomicAdd(&dh[1], d1);
        if (d2) atomicAdd(&dh[2], d2);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += T) bucket_hist[(uint64_t)blockIdx.x * 256 + i] = dh[i];
}

// Two entry points for the same body.  A whole bucket (dense tables) needs all 128 KiB of LDS a workgroup may
// have, so one workgroup sits on a CU whatever its register count -- no cap.  Half buckets (sparse tables) are
// meant to run two to a CU, which takes at most 64 vector registers.
template <int T>
__global__ __launch_bounds__(T) void k_bucket_count(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start,
                                                    uint32_t fb_bits, uint32_t split_bits, uint8_t *__restrict__ table8, uint32_t fresh,
                                                    int *__restrict__ bucket_hist) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int dh[256];
    bucket_count_body<T>(recs, final_start, fb_bits, split_bits, table8, fresh, bucket_hist, smem, dh);
}
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bucket_count_half(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start, uint32_t fb_bits, uint32_t split_bits,
                         uint8_t *__restrict__ table8, uint32_t fresh, int *__restrict__ bucket_hist) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int dh[256];
    bucket_count_body<T>(recs, final_start, fb_bits, split_bits, table8, fresh, bucket_hist, smem, dh);
}

// sums the per-bucket histogram rows into the running 256-bin histogram (signed deltas: two's complement adds)
__global__ __launch_bounds__(256) void k_hist_reduce(const int *__restrict__ bucket_hist, uint32_t n_rows, unsigned long long *__restrict__ hist) {
    long long acc = 0;
    uint32_t r = blockIdx.x;
    for (; r + 7 * gridDim.x < n_rows; r += 8 * gridDim.x) {             // eight independent row loads in flight
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = bucket_hist[(uint64_t)(r + q * gridDim.x) * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < 8; q++) acc += v[q];
    }
    for (; r < n_rows; r += gridDim.x) acc += bucket_hist[(uint64_t)r * 256 + threadIdx.x];
    if (acc) atomicAdd(&hist[threadIdx.x], (unsigned long long)acc);
}

// ------------------------------------------------------------------ K7: fold the side list in ---
// Side entries (addr, count) from the hot-key tables: aggregated once more per workgroup in LDS, then
// added to the finished u8 table with a saturating compare-and-swap on the containing dword.
constexpr uint32_t AS_SLOTS = 4096, AS_WGS = 64;

__device__ __forceinline__ void table_sat_add(uint8_t *table8, uint64_t addr, uint32_t cnt, int *dh) {
    unsigned int *word = reinterpret_cast<unsigned int *>(table8 + (addr & ~3ull));
    const uint32_t sh = (uint32_t)(addr & 3ull) * 8u;
    unsigned int old = *word;
    while (true) {
        uint32_t b = (old >> sh) & 0xffu;
        uint32_t nb = b + cnt > 255u ? 255u : b + cnt;
        if (nb == b) return;
        unsigned int want = (old & ~(0xffu << sh)) | (nb << sh);
        unsigned int prev = atomicCAS(word, old, want);
        if (prev == old) {                                               // the byte moved from b to nb: keep the histogram in step
            atomicAdd(&dh[nb], 1);
            if (b) atomicAdd(&dh[b], -1);
            return;
        }
        old = prev;
    }
}

__global__ __launch_bounds__(WG) void k_apply_side(const unsigned long long *__restrict__ side, const unsigned long long *__restrict__ side_n,
                                                   uint64_t side_cap, uint8_t *__restrict__ table8, unsigned long long *__restrict__ hist) {
    __shared__ unsigned long long key[AS_SLOTS];
    __shared__ uint32_t val[AS_SLOTS];
    __shared__ int dh[256];                                              // this workgroup's change to the value histogram
    unsigned long long n = *side_n;
    if (n > side_cap) n = side_cap;
    const unsigned long long per = (n + gridDim.x - 1) / gridDim.x;
    const unsigned long long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) return;
    for (uint32_t i = threadIdx.x; i < AS_SLOTS; i += WG) { key[i] = 0ull; val[i] = 0u; }
    dh[threadIdx.x & 255u] = 0;
    __syncthreads();
    for (unsigned long long i = lo + threadIdx.x; i < hi; i += WG) {
        const unsigned long long e = side[i];
        const uint64_t addr = e >> SIDE_CNT_BITS;
        const uint32_t cnt = (uint32_t)(e & ((1ull << SIDE_CNT_BITS) - 1ull));
        const unsigned long long kk = addr + 1ull;
        uint32_t h = (uint32_t)((addr * 0x9E3779B97F4A7C15ull) >> 40) & (AS_SLOTS - 1u);
        bool done = false;
        for (uint32_t t = 0; t < 32 && !done; t++) {
            unsigned long long old = atomicCAS(&key[h], 0ull, kk);
            if (old == 0ull || old == kk) {
                uint32_t before = atomicAdd(&val[h], cnt);
                if (before + cnt < before) atomicExch(&val[h], 0xffffffffu);      // saturate instead of wrapping
                done = true;
            }
            h = (h + 1u) & (AS_SLOTS - 1u);
        }
        if (!done) table_sat_add(table8, addr, cnt > 255u ? 255u : cnt, dh);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < AS_SLOTS; i += WG)
        if (key[i] != 0ull) table_sat_add(table8, key[i] - 1ull, val[i] > 255u ? 255u : val[i], dh);
    __syncthreads();
    if (threadIdx.x < 256 && dh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)(long long)dh[threadIdx.x]);
}

// ------------------------------------------------------------------ plan + launch sequence ------
PartPlan make_part_plan(uint32_t k, uint64_t n_bytes) {
    PartPlan pl;
    pl.k = k;
    pl.addr_bits = 2 * k;
    pl.fb_bits = pl.addr_bits < 16 ? pl.addr_bits : 16;
    const uint32_t bucket_bits = pl.addr_bits - pl.fb_bits;
    pl.b1 = bucket_bits <= 9 ? bucket_bits : (bucket_bits + 1) / 2;
    pl.b2 = bucket_bits - pl.b1;
    pl.B1 = 1u << pl.b1;
    pl.B2 = 1u << pl.b2;
    pl.n_chunks = (uint32_t)((n_bytes + CHUNK - 1) / CHUNK);
    pl.n_wg0 = pl.n_chunks < 1024u ? pl.n_chunks : 1024u;
    if (pl.n_wg0 == 0) pl.n_wg0 = 1;
    pl.G = (pl.n_chunks + pl.n_wg0 - 1) / pl.n_wg0;
    if (pl.G == 0) pl.G = 1;
    uint64_t r2 = (n_bytes + 1023) / 1024;
    pl.R2 = r2 < (uint64_t)TILE ? (uint64_t)TILE : ((r2 + TILE - 1) / TILE) * TILE;
    pl.n_wg2_max = (uint32_t)(n_bytes / pl.R2) + pl.B1 + 1;
    const char *dbg = getenv("PK_DEBUG_WALK");
    pl.dbg = dbg ? (uint32_t)atoi(dbg) : 0u;
    return pl;
}

size_t part_workspace_bytes(const PartPlan &pl, uint64_t n_bytes, PartWorkspace *lay) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t rec0 = pl.k > 15 ? 8 : 4;
    const uint64_t nfb = (uint64_t)pl.B1 * pl.B2;
    size_t o = 0;
    lay->flat = o; o += up((size_t)pl.n_chunks * 4 * SUB * rec0);
    lay->cnt = o; o += up((size_t)pl.n_chunks * 4 * 4);
    lay->hist1 = o; o += up((size_t)pl.n_wg0 * pl.B1 * 4);
    lay->rowoff1 = o; o += up((size_t)pl.n_wg0 * pl.B1 * 4);
    lay->bucket_base = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->wg2_start = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->col_tot = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->final_start = o; o += up((size_t)(nfb + 1) * 4);
    lay->out1 = o; o += up((size_t)(n_bytes + 64) * (pl.b2 ? 4 : 2));
    lay->hist2 = o; o += up((size_t)pl.n_wg2_max * pl.B2 * 4);
    lay->rowoff2 = o; o += up((size_t)pl.n_wg2_max * pl.B2 * 4);
    lay->out2 = o; o += up(pl.b2 ? (size_t)(n_bytes + 64) * 2 : 256);
    lay->side_cap = n_bytes + 16;                          // every side entry stands for >= 1 k-mer
    lay->side = o; o += up((size_t)lay->side_cap * 8);
    lay->side_n = o; o += 256;
    const bool fine = pl.b2 && nfb <= 16384 && pl.k <= 15;             // level 2 without a counting pass (see k_scatter1)
    lay->fine_rows = o; o += fine ? up((size_t)pl.n_wg0 * nfb * 4) : 0;
    lay->fine_tot = o; o += fine ? up((size_t)nfb * 4) : 0;
    lay->cursor = o; o += fine ? up((size_t)nfb * 4) : 0;
    lay->bucket_hist = o; o += up((size_t)nfb * 2 * 256 * 4);              // up to 2 workgroups per bucket
    return o;
}

int launch_partitioned(const uint8_t *fasta, uint64_t n, uint64_t stream_off, const LaneState *lane_state, const L2 *st2, const PartPlan &pl,
                       uint8_t *ws, const PartWorkspace &lay, uint8_t *table8, DevRec *recs, uint64_t recs_cap, Carry *carry,
                       hipStream_t s, hipEvent_t ev_walk_end, hipEvent_t ev_part_end, bool fresh, unsigned long long *hist) {
    uint32_t *cnt = (uint32_t *)(ws + lay.cnt), *hist1 = (uint32_t *)(ws + lay.hist1), *rowoff1 = (uint32_t *)(ws + lay.rowoff1);
    uint32_t *bucket_base = (uint32_t *)(ws + lay.bucket_base), *wg2_start = (uint32_t *)(ws + lay.wg2_start);
    uint32_t *final_start = (uint32_t *)(ws + lay.final_start), *hist2 = (uint32_t *)(ws + lay.hist2), *rowoff2 = (uint32_t *)(ws + lay.rowoff2);
    void *flat = ws + lay.flat, *out1 = ws + lay.out1, *out2 = ws + lay.out2;
    unsigned long long *side = (unsigned long long *)(ws + lay.side), *side_n = (unsigned long long *)(ws + lay.side_n);
    const uint32_t nfb = pl.B1 * pl.B2;
    const bool fine = pl.b2 && nfb <= 16384 && pl.k <= 15;
    const size_t lds_fine = SCATTER_LDS_NARROW + (size_t)nfb * 4;
    hipFuncSetAttribute((const void *)k_scatter1<uint32_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_scatter1<uint32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SCATTER_LDS_NARROW + 65536));
    hipFuncSetAttribute((const void *)k_scatter1<uint64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_WIDE);
    hipFuncSetAttribute((const void *)k_scatter2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_scatter2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_bucket_count<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)k_bucket_count_half<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (hipMemsetAsync(side_n, 0, 16, s) != hipSuccess) return -2;   // side-list length + the column scan's ticket counter
    if (pl.dbg && pl.k <= 15) {                              // diagnostic build of the walk (PK_DEBUG_WALK), timing only
        hipLaunchKernelGGL((k_walk_flat<uint32_t, uint32_t, true>), dim3(pl.n_wg0), dim3(WG), 0, s, fasta, n, stream_off, lane_state, st2, pl,
                           (uint32_t *)flat, cnt, hist1, recs, recs_cap, carry, side, side_n, lay.side_cap);
    } else if (pl.k <= 15) {
        hipLaunchKernelGGL((k_walk_flat<uint32_t, uint32_t, false>), dim3(pl.n_wg0), dim3(WG), 0, s, fasta, n, stream_off, lane_state, st2, pl,
                           (uint32_t *)flat, cnt, hist1, recs, recs_cap, carry, side, side_n, lay.side_cap);
    } else {
        hipLaunchKernelGGL((k_walk_flat<uint64_t, uint64_t, false>), dim3(pl.n_wg0), dim3(WG), 0, s, fasta, n, stream_off, lane_state, st2, pl,
                           (uint64_t *)flat, cnt, hist1, recs, recs_cap, carry, side, side_n, lay.side_cap);
    }
    if (ev_walk_end) hipEventRecord(ev_walk_end, s);
    if (pl.dbg && pl.k <= 15) {                              // ablation run: time the walk kernel only, results are garbage
        if (ev_part_end) hipEventRecord(ev_part_end, s);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    hipLaunchKernelGGL(k_rows1_scan, dim3(pl.B1), dim3(256), 0, s, hist1, rowoff1, pl, (uint32_t *)(ws + lay.col_tot),
                       (unsigned int *)(side_n + 1), bucket_base, wg2_start, final_start);
    uint32_t *fine_rows = (uint32_t *)(ws + lay.fine_rows), *fine_tot = (uint32_t *)(ws + lay.fine_tot), *cursor = (uint32_t *)(ws + lay.cursor);
    if (fine) {
        if (hipMemsetAsync(fine_tot, 0, (size_t)nfb * 4, s) != hipSuccess) return -2;
        hipLaunchKernelGGL((k_scatter1<uint32_t, true>), dim3(pl.n_wg0), dim3(SC_T), lds_fine, s, (const uint32_t *)flat, cnt, rowoff1,
                           bucket_base, pl, out1, fine_rows);
    } else if (pl.k <= 15)
        hipLaunchKernelGGL((k_scatter1<uint32_t, false>), dim3(pl.n_wg0), dim3(SC_T), SCATTER_LDS_NARROW, s, (const uint32_t *)flat, cnt, rowoff1,
                           bucket_base, pl, out1, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL((k_scatter1<uint64_t, false>), dim3(pl.n_wg0), dim3(SC_T), SCATTER_LDS_WIDE, s, (const uint64_t *)flat, cnt, rowoff1,
                           bucket_base, pl, out1, (uint32_t *)nullptr);
    const uint16_t *final_recs = (const uint16_t *)out1;
    if (fine) {
        const uint32_t row_groups = pl.n_wg0 < 16u ? 1u : 16u;
        hipLaunchKernelGGL(k_fine_sum, dim3((nfb + 255u) / 256u, row_groups), dim3(256), 0, s, (const uint32_t *)fine_rows, pl.n_wg0, nfb, fine_tot);
        hipLaunchKernelGGL(k_fine_scan, dim3(1), dim3(1024), 0, s, (const uint32_t *)fine_tot, nfb, final_start, cursor);
        hipLaunchKernelGGL(k_scatter2<true>, dim3(pl.n_wg2_max), dim3(SC_T), SCATTER_LDS_NARROW, s, (const uint32_t *)out1, wg2_start,
                           bucket_base, (const uint32_t *)nullptr, final_start, pl, out2, cursor);
        final_recs = (const uint16_t *)out2;
    } else if (pl.b2) {
        hipLaunchKernelGGL(k_count2, dim3(pl.n_wg2_max), dim3(WG), 0, s, (const uint32_t *)out1, wg2_start, bucket_base, pl, hist2);
        hipLaunchKernelGGL(k_rows2_scan, dim3(pl.B1), dim3(512), 0, s, hist2, rowoff2, wg2_start, bucket_base, pl, final_start);
        hipLaunchKernelGGL(k_scatter2<false>, dim3(pl.n_wg2_max), dim3(SC_T), SCATTER_LDS_NARROW, s, (const uint32_t *)out1, wg2_start,
                           bucket_base, rowoff2, final_start, pl, out2, (uint32_t *)nullptr);
        final_recs = (const uint16_t *)out2;
    }
    if (ev_part_end) hipEventRecord(ev_part_end, s);
    // sparse tables (few records per 2^16-address bucket, k=17): 2^split workgroups per bucket, see k_bucket_count
    const uint32_t split = (pl.fb_bits == 16 && n / nfb < 8192) ? 1u : 0u;
    const size_t part_addrs = (size_t)1 << (pl.fb_bits - split);
    const size_t lds6 = part_addrs * 2 < 64 ? 64 : part_addrs * 2;
    int *bucket_hist = (int *)(ws + lay.bucket_hist);
    const uint32_t n_rows6 = (uint32_t)(nfb << split);
    if (split)
        hipLaunchKernelGGL(k_bucket_count_half<1024>, dim3(n_rows6), dim3(1024), lds6, s, final_recs, final_start, pl.fb_bits, split, table8, fresh ? 1u : 0u, bucket_hist);
    else
        hipLaunchKernelGGL(k_bucket_count<1024>, dim3(n_rows6), dim3(1024), lds6, s, final_recs, final_start, pl.fb_bits, split, table8, fresh ? 1u : 0u, bucket_hist);
    hipLaunchKernelGGL(k_hist_reduce, dim3(n_rows6 < 16u ? 1u : (n_rows6 / 16u > 2048u ? 2048u : n_rows6 / 16u)), dim3(256), 0, s, (const int *)bucket_hist, n_rows6, hist);
    hipLaunchKernelGGL(k_apply_side, dim3(AS_WGS), dim3(WG), 0, s, side, side_n, lay.side_cap, table8, hist);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pk
