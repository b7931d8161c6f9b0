// part_common.h -- pieces shared by the partition kernels (kmer_fuse.hip: k-mer assembly + level-1 sort;
// kmer_part.hip: level-2 sort, bucket count, side list): the per-workgroup hot-key table and the LDS
// counting sort of one 16 K-record tile.
#pragma once
#include <cstddef>
#include "fasta_fsm.h"
#include "pk_kernels.h"

namespace pk {

constexpr int SC_T = 1024;           // threads of the scatter / bucket-count workgroups
constexpr int SC_PER = 16;           // records per thread per tile
constexpr int TILE = SC_T * SC_PER;  // 16384 records per tile = one FASTA chunk's worth (one base per byte at most)

// batch sizes of the sort tile's parking / write-out phases (scatter_tile; 0 = record by record), per level
#ifndef PK_PB_L1
#define PK_PB_L1 0
#endif
#ifndef PK_SB_L1
#define PK_SB_L1 8          // 32-bit k-mers only (the 64-bit kernel loses with it: 2.22 -> 2.29 ms)
#endif
#ifndef PK_PB_L2
#define PK_PB_L2 0
#endif
#ifndef PK_SB_L2
#define PK_SB_L2 8
#endif
// experiments (tools/build_variant.sh): PK_CNT0 = count phase without per-record branches (empty slots add zero);
// PK_UFULL = run write-out batches that lie wholly inside the tile skip the per-lane bounds test
#ifndef PK_CNT0
#define PK_CNT0 0
#endif
#ifndef PK_UFULL
#define PK_UFULL 1          // k = 15: level 1 1.314 -> 1.285 ms (PK_CNT0: 1.68 ms -- the holes' LDS adds cost more than their branches)
#endif

// ------------------------------------------------------------------ hot keys ---------------------
// Tandem repeats (poly-A/T, (AT)n, (AAG)n ...) put tens of millions of identical canonical k-mers on a
// handful of addresses; routed like everything else they would all land in ONE final bucket, i.e. on
// one CU.  The level-1 kernel (kmer_fuse.hip) finds k-mers that repeat the one 1, 2 or 3 bases earlier from the
// packed bases themselves, keeps them out of the record stream and tallies them in this per-workgroup LDS
// hash table (addr -> count), which is appended to a global side list when the workgroup finishes (or the
// table half fills).  k_apply_side folds the side list into the finished u8 table with saturating CAS
// adds -- a few thousand entries instead of 10^7..10^8 records.
constexpr uint32_t HOT_PROBES = 16;
constexpr uint32_t SIDE_CNT_BITS = 28;   // side entry = (addr << 28) | count

template <uint32_t SLOTS>               // per-workgroup LDS hash slots (a power of two)
struct HotTable {
    static constexpr uint32_t HOT_SLOTS = SLOTS;
    unsigned long long key[SLOTS];       // addr + 1, 0 = empty
    uint32_t val[SLOTS];
    uint32_t used, n_flush;
    unsigned long long flush_base;
};

__device__ __forceinline__ void side_append_one(unsigned long long *side, unsigned long long *side_n, uint64_t side_cap,
                                                uint64_t addr, uint32_t cnt) {
    unsigned long long i = atomicAdd(side_n, 1ull);
    if (i < side_cap) side[i] = ((unsigned long long)addr << SIDE_CNT_BITS) | cnt;
}

template <uint32_t SLOTS>
__device__ __forceinline__ void hot_insert(HotTable<SLOTS> &H, uint64_t addr, uint32_t cnt, unsigned long long *side,
                                           unsigned long long *side_n, uint64_t side_cap) {
    constexpr uint32_t HOT_SLOTS = SLOTS;
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

// all threads of the workgroup; appends every occupied slot to the side list and clears the table
template <uint32_t SLOTS>
__device__ __forceinline__ void hot_flush(HotTable<SLOTS> &H, unsigned long long *side, unsigned long long *side_n, uint64_t side_cap) {
    constexpr uint32_t HOT_SLOTS = SLOTS;
    __syncthreads();
    if (threadIdx.x == 0) H.n_flush = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += blockDim.x) mine += H.key[i] != 0ull;
    uint32_t at = mine ? atomicAdd(&H.n_flush, mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) H.flush_base = H.n_flush ? atomicAdd(side_n, (unsigned long long)H.n_flush) : 0ull;
    __syncthreads();
    const unsigned long long base64 = H.flush_base;
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


// Level-1 records of 32-bit k-mers (k <= 15) that go on to a second level need at most 23 bits (address bits below the
// level-1 digit, + that digit's lowest bit rides along): they are stored as a 16-bit plane and an 8-bit plane with the
// same record index -- 3 bytes per record instead of 4, written once and read once (0.7 GB less each way on the 800 Mbp
// genome).  The high plane follows the low plane (and its dump tile) in the same allocation.
__host__ __device__ __forceinline__ size_t level1_hi_plane_offset(uint64_t capacity1) {
    return (((size_t)(capacity1 + TILE + 64) * 2u) + 255u) & ~(size_t)255u;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    // DPP: shifts inside the rows of 16 lanes, then the row totals broadcast into the rows behind them (no LDS traffic;
    // the shuffle form went through ds_bpermute six times per scan, 64 scans per wave in k_provision)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    return v;
}

// ------------------------------------------------------------------ K2 / K5: scatter ------------
// One tile = up to 16384 records: rank within digit by LDS atomic, exclusive scan of the digit
// counts, records + digits parked in LDS in sorted order, then written as coalesced runs at
// run[d] (this workgroup's running output offset for digit d).
template <int NB>                  // NB = most destinations (digits) a tile is sorted into
struct ScatterLdsT {
    // [NB .. NB + 64): one scratch digit per lane -- register slots without a record rank and park there, so the
    // ranking and parking loops carry no branches (an LDS atomic behind a branch is waited for on the spot; sixteen
    // of them back to back cost one round trip).  Likewise rec / dig [TILE .. TILE + 64).
    uint32_t hist[NB + 64], off[NB + 64], run[NB], gbase[NB];
    uint32_t wsum[SC_T / 64];
    uint32_t total, pad_[3];       // records of the tile being sorted
    uint32_t rec[TILE + 64];
    uint16_t dig[TILE + 64];       // only when the digit does not fit beside the record (k = 17, level 1)
};
typedef ScatterLdsT<512> ScatterLds;
constexpr size_t SCATTER_LDS_NARROW = offsetof(ScatterLds, dig);   // 72 KiB: two workgroups per CU
constexpr size_t SCATTER_LDS_WIDE = sizeof(ScatterLds);            // 104 KiB

// WIDE = the digit is kept in its own LDS array; otherwise the record parked in LDS still carries its
// digit (digit << shift | rest fits 32 bits) and is masked on the way out.
// `claim` != nullptr: the tile's run for digit d starts where a global cursor says (atomicAdd of the
// run length), instead of at this workgroup's precomputed running offset L.run[d].  With `cap_end` too the
// destinations have provisioned capacities (level 1, see kmer_fuse.hip): a run that would end beyond
// cap_end[d] raises *overflow and is written to the `dump` area instead (one tile's worth of slots behind
// the buckets), so nothing is ever stored outside its bucket; the host then repeats the level with exact sizes.
// n_tile == ~0u: the number of records is not known to the caller; it is taken from the digit scan.
//
// `settle()` is called once the tile is parked, right before its runs are stored.  The callers use it to
// take delivery of the NEXT tile's prefetched records at that point.  On this ISA loads and stores share one
// in-order counter (vmcnt): a wait placed after the stores -- where the compiler would put it, at the top
// of the next tile -- also waits for the stores to be acknowledged by HBM, a full round trip of ~8 us per
// tile with nothing else in flight.  Waiting here costs nothing (the loads were issued a whole sort ago)
// and leaves the stores in flight through the next tile's ranking and parking.
// NT threads sort PER records each (NT * PER = TILE); bit j of `okm` tells whether r[j] holds a record (one
// register instead of PER lane masks: 32 booleans do not fit the scalar register file).
// FULL: every register slot holds a record (okm is not looked at).
//
// Ranking is count-then-claim: the digits are first only counted (LDS adds that return nothing), and after the scan
// a record finds its place when it is parked -- a second add, on its digit's running offset, hands out the positions
// (off[d] ends at the start of run d + 1; gbase keeps the starts).  Round 1 ranked with one returning add and kept
// digit and rank of every record in a register until the scan was done: 16 or 32 more live registers and three to
// four more vector instructions per record, in kernels that are bound by instruction issue.
// OFF32: every output index of this kernel is below 2^31 (the host cuts feeds of 32-bit k-mers so that both bucket areas
// stay below that: feed_piece), so byte offsets fit 32 bits and the stores take a scalar base + a 32-bit lane offset
// instead of a 64-bit address per record.
template <typename RIN, bool WIDE, int NT = SC_T, int PER = SC_PER, int NB = 512, bool FULL = false, int PB = 0, int SB = 0, bool OFF32 = false, class Settle>
__device__ __forceinline__ void scatter_tile(ScatterLdsT<NB> &L, const RIN (&r)[PER], uint32_t okm, uint32_t n_tile,
                                             uint32_t shift, uint32_t B, uint32_t low_mask, bool out16, void *__restrict__ out,
                                             Settle &&settle, uint32_t *claim = nullptr, const uint32_t *__restrict__ cap_end = nullptr,
                                             uint32_t dump = 0, uint32_t *overflow = nullptr, uint8_t *__restrict__ out_hi = nullptr,
                                             unsigned long long *prof = nullptr, const uint32_t *watch = nullptr, uint32_t watch_limit = 0,
                                             bool *watch_hit = nullptr) {
    static_assert(NT * PER == TILE, "tile shape");
    // PK_PHASE_PROF (experiment builds): thread 0 adds the cycles of this tile's phases -- count, scan, park, store, each up to
    // its closing barrier -- to prof[1..4]
#ifdef PK_PHASE_PROF
    unsigned long long pt = __builtin_readcyclecounter();
#define PK_PROF_MARK(i) do { if (prof && threadIdx.x == 0) { const unsigned long long pn = __builtin_readcyclecounter(); prof[i] += pn - pt; pt = pn; } } while (0)
#else
#define PK_PROF_MARK(i) do { } while (0)
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t dbits = (uint32_t)__builtin_ctz(B);
    auto digit_of = [&](const RIN &x) -> uint32_t {
        return sizeof(RIN) == 4 ? __builtin_amdgcn_ubfe((uint32_t)x, shift, dbits) : (uint32_t)((uint64_t)x >> shift) & (B - 1u);
    };
    // PER <= 16: the digit, or this lane's scratch digit for an empty slot.  64-bit records do not keep it (the k = 17
    // kernel is out of registers as it is: 28 were spilled to scratch memory) and cut it from the record again when parking.
    // <= 16 records per thread: empty slots take a scratch digit instead of a branch (measured the other way round for the
    // 64-bit kernel too: per-record tests cost it 2.22 -> 2.36 ms)
    constexpr bool BRANCH_FREE = PER <= 16;
    constexpr bool KEEP_DG = PER <= 16 && sizeof(RIN) == 4;
    uint32_t dg[KEEP_DG ? PER : 1];
    auto slot_digit = [&](int j) -> uint32_t { return (FULL || ((okm >> j) & 1u)) ? digit_of(r[j]) : (uint32_t)NB + lane; };
    if (BRANCH_FREE) {
        // branch-free: an LDS operation behind a branch is waited for on the spot, sixteen back to back cost one round trip
        if (KEEP_DG) {
#pragma unroll
            for (int j = 0; j < PER; j++) dg[KEEP_DG ? j : 0] = slot_digit(j);
        }
#pragma unroll
        for (int j = 0; j < PER; j++)
            __hip_atomic_fetch_add(&L.hist[KEEP_DG ? dg[KEEP_DG ? j : 0] : slot_digit(j)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        // 32 records per thread: every record under its own test (32 digits in registers spill; scratch digits instead of
        // the tests: 1.42 -> 1.60 ms; a wave-uniform path without the tests for waves whose slots are all full: 1.32 -> 1.36)
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (PK_CNT0 && !FULL) {
                // no branch: an empty slot adds zero to the counter its (stale but in-range) digit names
                __hip_atomic_fetch_add(&L.hist[digit_of(r[j])], (okm >> j) & 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (FULL || ((okm >> j) & 1u)) __hip_atomic_fetch_add(&L.hist[digit_of(r[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (!FULL && threadIdx.x < 64u) L.off[NB + threadIdx.x] = (uint32_t)TILE + threadIdx.x;   // where empty slots park (adds of zero)
    __syncthreads();
    PK_PROF_MARK(1);
    // `watch` (the hot-key table's fill): looked at by ONE thread between the tile's first two barriers -- nothing writes it
    // there -- and handed to all behind the second, so the caller's decision to flush is uniform although no barrier
    // closes the tile any more
    if (watch && threadIdx.x == 0) L.pad_[0] = *watch >= watch_limit ? 1u : 0u;
    // exclusive scan of hist[0..B) by the first B threads (B <= NB <= NT)
    uint32_t my_off = 0, claimed = 0, room_end = 0, run_len = 0;
    {
        // The thread index is made opaque to the compiler here.  What this block derives from it -- the wave's slot in
        // wsum, the thread's entries of claim and cap_end -- is invariant across the caller's tile loop; the compiler kept
        // those three addresses in registers, spilled them under pressure (k_scatter2: 9 registers), and every re-load
        // from scratch is followed by a wait for ALL vector memory operations in flight: the next tile's records,
        // requested before this tile and meant to arrive during it (settle), were waited for here, between two barriers.
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, w = tid >> 6;
        const uint32_t v = threadIdx.x < B ? L.hist[threadIdx.x] : 0u;
        const uint32_t inc = wave_incl_scan_u32(v);          // DPP: no trips through the LDS crossbar (six ds_bpermute before)
        if (lane == 63) L.wsum[w] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (int i = 0; i < w; i++) pre += L.wsum[i];
        my_off = pre + inc - v;
        if (threadIdx.x < B) {
            L.off[threadIdx.x] = my_off;
            if (threadIdx.x == B - 1u) L.total = my_off + v;
            // a claimed run start is only needed when the runs are written: the atomic's round trip to HBM
            // overlaps the parking of the records below
            if (claim) {
                // issued here, looked at only after the parking below (the room check included: testing the returned
                // value on the spot put the atomic's whole round trip to the memory side -- a third of this phase, 12 %
                // of a tile -- in front of the barrier, with the other six waves waiting there)
                claimed = v ? atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(claim) + tid * 4u), v) : 0u;
                if (cap_end) room_end = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(cap_end) + tid * 4u);
                run_len = v;
            } else { L.gbase[threadIdx.x] = L.run[threadIdx.x] - my_off; L.run[threadIdx.x] += v; }
            L.hist[threadIdx.x] = 0;                          // ready for the next tile
        }
    }
    __syncthreads();
    PK_PROF_MARK(2);
    if (watch_hit) *watch_hit = L.pad_[0] != 0u;
    if (n_tile == ~0u) n_tile = L.total;
    if (PK_UFULL) n_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_tile);   // uniform by construction; now in a scalar register
    // Parking.  PB == 0: record by record (returning add, then the write it places); every add is waited for on the spot
    // -- the compiler may not move an LDS write across the next atomic -- so a thread walks a chain of PER LDS round
    // trips, which the other waves of the CU cover.  PB > 0: PB returning adds back to back, then their PB writes --
    // measured and not used: level 1 1.47 -> 1.68 ms (PB = 8), 1.65 (PB = 4); level 2 unchanged.
    if constexpr (PB == 0) {
        if (BRANCH_FREE) {
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const uint32_t d = KEEP_DG ? dg[KEEP_DG ? j : 0] : slot_digit(j);
                const uint32_t p = atomicAdd(&L.off[d], (FULL || ((okm >> j) & 1u)) ? 1u : 0u);
                if (WIDE) { L.rec[p] = (uint32_t)((uint64_t)r[j] & low_mask); L.dig[p] = (uint16_t)d; }
                else L.rec[p] = (uint32_t)r[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++)
                if (FULL || ((okm >> j) & 1u)) {
                    const uint32_t d = digit_of(r[j]);
                    const uint32_t p = atomicAdd(&L.off[d], 1u);
                    if (WIDE) { L.rec[p] = (uint32_t)((uint64_t)r[j] & low_mask); L.dig[p] = (uint16_t)d; }
                    else L.rec[p] = (uint32_t)r[j];
                }
        }
    } else {
        static_assert(PB == 0 || PER % (PB ? PB : 1) == 0, "parking batch");
#pragma unroll
        for (int j0 = 0; j0 < PER; j0 += (PB ? PB : 1)) {
            uint32_t p[PB ? PB : 1], d[PB ? PB : 1];
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int j = j0 + u;
                const bool ok = FULL || ((okm >> j) & 1u);
                d[u] = KEEP_DG ? dg[KEEP_DG ? j : 0] : (ok ? digit_of(r[j]) : (uint32_t)NB + lane);
                p[u] = atomicAdd(&L.off[d[u]], ok ? 1u : 0u);         // empty slots: add zero on the lane's scratch digit, park behind the tile
            }
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int j = j0 + u;
                if (WIDE) { L.rec[p[u]] = (uint32_t)((uint64_t)r[j] & low_mask); L.dig[p[u]] = (uint16_t)d[u]; }
                else L.rec[p[u]] = (uint32_t)r[j];
            }
        }
    }
    if (claim && threadIdx.x < B) {
        if (cap_end && run_len && claimed + run_len > room_end) {            // provisioned room exhausted (rare): park the run aside
            *overflow = 1u;
            claimed = dump + my_off;
        }
        L.gbase[threadIdx.x] = claimed - my_off;                             // sorted position p of digit d goes to p + gbase[d]
    }
    __syncthreads();
    PK_PROF_MARK(3);
    settle();
#ifdef PK_PHASE_PROF
    if (prof && threadIdx.x == 0) { const unsigned long long pn = __builtin_readcyclecounter(); prof[0] += pn - pt; pt = pn; }   // (delivery of the next tile's loads: booked on phase 0)
#endif
    // Run write-out, one record per lane and store: 64 consecutive sorted positions are 64 consecutive records of a run
    // (or of two).  32-bit records keep what they had above `low_mask` (narrow: the digit; wide: nothing) -- every
    // reader of 32-bit records masks for itself.  (Storing neighbours pairwise as 8 bytes, as round 1 did, halves the
    // store instructions but costs 15 vector instructions per record against 5 here.)
    // SB == 0: position by position (parked record, its digit's run base, store: two LDS round trips each).  SB > 0: SB
    // positions at a time -- the reads are unconditional then: positions past n_tile hold stale records whose digit
    // field still indexes gbase in range; only the stores are masked.  SB = 8: level 1 (32-bit k-mers) 1.47 -> 1.42 ms, level 2
    // 1.27 -> 1.24; 4 about the same, 16 loses (1.58 / 1.54), and so does any batching for 64-bit k-mers.
    // out_hi != nullptr (and !out16): the record leaves as two planes, its low 16 bits at o[...] and bits 16-23 at out_hi[...]
    auto store_runs = [&](auto *o) {
        constexpr bool O16 = sizeof(*o) == 2;
        const bool planes = O16 && !out16;
        // OFF32: the byte offset is formed in 32 bits, so the store takes the scalar base + a 32-bit lane offset (the
        // compiler does not derive that from an assumption on the index: it needs to see the 32-bit arithmetic)
        auto put = [&](uint32_t at, uint32_t v) {
            if (OFF32) *reinterpret_cast<decltype(o)>(reinterpret_cast<uint8_t *>(o) + (uint32_t)(at * (uint32_t)sizeof(*o))) = (decltype(*o + 0))v;
            else o[at] = v;
        };
        auto idx32 = [](uint32_t at) -> uint32_t { return at; };
        if constexpr (SB == 0) {
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const uint32_t p = threadIdx.x + (uint32_t)j * NT;
                if (p < n_tile) {
                    const uint32_t r0 = L.rec[p];
                    const uint32_t d0 = WIDE ? (uint32_t)L.dig[p] : __builtin_amdgcn_ubfe(r0, shift, dbits);
                    const uint32_t at = idx32(p + L.gbase[d0]);
                    if (O16) { put(at, (uint16_t)(planes ? r0 : (r0 & low_mask))); if (planes) out_hi[at] = (uint8_t)(r0 >> 16); }
                    else put(at, r0);
                }
            }
        } else {
            static_assert(SB == 0 || PER % (SB ? SB : 1) == 0, "store batch");
#pragma unroll
            for (int j0 = 0; j0 < PER; j0 += (SB ? SB : 1)) {
                uint32_t r0[SB ? SB : 1], g0[SB ? SB : 1];
#pragma unroll
                for (int u = 0; u < SB; u++) r0[u] = L.rec[threadIdx.x + (uint32_t)(j0 + u) * NT];
#pragma unroll
                for (int u = 0; u < SB; u++) {
                    const uint32_t p = threadIdx.x + (uint32_t)(j0 + u) * NT;
                    const uint32_t d0 = WIDE ? ((uint32_t)L.dig[p] & (uint32_t)(NB - 1)) : __builtin_amdgcn_ubfe(r0[u], shift, dbits);
                    g0[u] = L.gbase[d0];
                }
                // PK_UFULL: a batch that ends inside the tile (uniform test) stores without the per-lane bounds test
                const bool whole = PK_UFULL && !FULL && (uint32_t)(j0 + SB) * NT <= n_tile;
                if (PK_UFULL && whole) {
#pragma unroll
                    for (int u = 0; u < SB; u++) {
                        const uint32_t p = threadIdx.x + (uint32_t)(j0 + u) * NT;
                        const uint32_t at = idx32(p + g0[u]);
                        if (O16) { put(at, (uint16_t)(planes ? r0[u] : (r0[u] & low_mask))); if (planes) out_hi[at] = (uint8_t)(r0[u] >> 16); }
                        else put(at, r0[u]);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < SB; u++) {
                        const uint32_t p = threadIdx.x + (uint32_t)(j0 + u) * NT;
                        if (FULL || p < n_tile) {
                            const uint32_t at = idx32(p + g0[u]);
                            if (O16) { put(at, (uint16_t)(planes ? r0[u] : (r0[u] & low_mask))); if (planes) out_hi[at] = (uint8_t)(r0[u] >> 16); }
                            else put(at, r0[u]);
                        }
                    }
                }
            }
        }
    };
    if (out16 || out_hi) store_runs(reinterpret_cast<uint16_t *>(out));
    else store_runs(reinterpret_cast<uint32_t *>(out));
    // No barrier here (round 3).  What the next tile touches before ITS first barrier -- its records in registers, the
    // digit counters (zeroed during this tile's scan, not read since), the scratch entries of `off` -- is not read by a
    // wave still storing this tile's runs; everything else of the next tile (scan: off / gbase / total; parking: rec / dig)
    // lies behind that barrier, which no wave passes before all have left this store loop.  Waves that are done start
    // assembling the next tile's k-mers while the slowest still stores: thread 0 spent 31 % of a tile waiting at the
    // first barrier for the slowest wave's assembly.
#ifndef PK_TAIL_BARRIER
#define PK_TAIL_BARRIER 0
#endif
    if (PK_TAIL_BARRIER) __syncthreads();
    PK_PROF_MARK(4);
}


}  // namespace pk
