// kmer_part.hip -- table update ("partitioned"): no HBM atomics on the count table.
//
// One global atomic per k-mer into a 4^k-entry table tops out near 7.6 G/s on MI355X (scattered device-scope
// atomics execute at the memory side; measured in round 1), i.e. ~7.5 Gbp/s however good the parser is.
// Here the canonical k-mers are instead routed to the workgroup that owns their slice of the address space:
//
//   kmer_pack.hip  k_squeeze     FASTA text -> packed valid bases (2-bit codes + restart bits), per-record tallies
//   kmer_fuse.hip  k_walk_sort   packed bases -> canonical k-mers (in registers) -> LDS counting sort by the level-1
//                                digit (top b1 address bits) -> coalesced run writes into provisioned buckets
//      k_count2 / k_rows2_scan   (k = 17)  per-workgroup histogram of the level-2 digit (next b2 bits)
//                    inside each level-1 bucket, per-bucket column scan -> offsets + final bucket starts
//   K5 k_scatter2    second pass, 16-bit records (address inside the final bucket).  k <= 15: the final buckets were
//                    laid out from the same sampled estimate as the level-1 ones; every tile claims room for its
//                    runs from their cursors (atomicAdd), there is no counting pass
//   K6 k_bucket_count one workgroup per final bucket of 2^16 addresses: the slice of the u8 table
//                    lives in LDS as 16-bit counters, ds_add per record, clamp, slice written to HBM
//                    (read back first when an earlier feed already wrote it)
//   K7 k_apply_side  the few hot k-mers the walk kept out of the record stream (part_common.h)
//
// Every pass is a stream.  Runs are claimed from cursors, so the order of records inside a bucket depends on
// timing; the table -- a saturating sum per address -- does not.  Saturation is exact: min(255, .) is applied
// only when a bucket's counters leave LDS (indexer.py:239,262), and K6 folds the slice already in
// HBM back in, so several feeds accumulate exactly like the reference's flushes.
//
// `flags[0]` is raised by the level-1 sort when a provisioned bucket ran out of room (kmer_fuse.hip); every
// kernel below then returns without touching anything and the host repeats the level with exact sizes.
#include <cstddef>
#include <cstdlib>
#include "part_common.h"

namespace pk {

// level-2 work split: bucket b is covered by workgroups wg2_start[b] .. wg2_start[b+1]-1, R2 records each
__device__ __forceinline__ bool wg2_range(const uint32_t *__restrict__ wg2_start, const uint32_t *__restrict__ bucket_base,
                                          const uint32_t *__restrict__ bucket_end, const PartPlan &pl, uint32_t &b, uint32_t &lo,
                                          uint32_t &hi) {
    const uint32_t w = blockIdx.x;
    if (w >= wg2_start[pl.B1]) return false;
    uint32_t a = 0, z = pl.B1;                              // last b with wg2_start[b] <= w
    while (z - a > 1) { uint32_t m = (a + z) >> 1; if (wg2_start[m] <= w) a = m; else z = m; }
    b = a;
    uint64_t s = (uint64_t)bucket_base[b] + (uint64_t)(w - wg2_start[b]) * pl.R2;
    uint64_t e = s + pl.R2;
    if (e > bucket_end[b]) e = bucket_end[b];              // the bucket's records; the rest of its room stays unused
    lo = (uint32_t)s; hi = (uint32_t)e;
    return true;
}

__global__ __launch_bounds__(WG) void k_count2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ wg2_start,
                                               const uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ bucket_end, PartPlan pl,
                                               uint32_t *__restrict__ hist_rows, const uint32_t *__restrict__ flags) {
    __shared__ uint32_t h[512];
    if (flags[0]) return;
    for (uint32_t d = threadIdx.x; d < pl.B2; d += WG) h[d] = 0;
    __syncthreads();
    uint32_t b, lo, hi;
    const bool live = wg2_range(wg2_start, bucket_base, bucket_end, pl, b, lo, hi);
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
// The level-2 output is compact again: final bucket starts count up from compact_base[b], the number of records in
// the level-1 buckets before b.
__global__ __launch_bounds__(512) void k_rows2_scan(const uint32_t *__restrict__ hist_rows, uint32_t *__restrict__ rowoff,
                                                    const uint32_t *__restrict__ wg2_start, const uint32_t *__restrict__ compact_base,
                                                    PartPlan pl, uint32_t *__restrict__ final_start, const uint32_t *__restrict__ flags) {
    __shared__ uint32_t tot[512];
    if (flags[0]) return;
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
        uint32_t a = compact_base[b];
        for (uint32_t i = 0; i < pl.B2; i++) { uint32_t n = tot[i]; final_start[(uint64_t)b * pl.B2 + i] = a; a += n; }
        if (b == pl.B1 - 1) final_start[(uint64_t)pl.B1 * pl.B2] = a;
    }
}

// ---- sample2: final-bucket sizes from a sample of the level-1 records (2^15 < final buckets <= 2^18: k = 17, slices of k = 19).
// One workgroup per level-1 bucket tallies the level-2 digit of every stride2-th group of 256 records in LDS (stride2 = 1:
// all of them, i.e. exact).  It replaces the exact counting pass over ALL level-1 records (k_count2: 0.66 ms at k = 17).
__global__ __launch_bounds__(1024) void k_sample2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ bucket_base,
                                                  const uint32_t *__restrict__ bucket_end, PartPlan pl, uint32_t stride2,
                                                  uint32_t *__restrict__ tally, uint32_t *__restrict__ sampled_n,
                                                  const uint32_t *__restrict__ flags) {
    __shared__ uint32_t h[512];
    __shared__ uint32_t n_seen;
    if (flags[0]) return;
    const uint32_t b = blockIdx.x, lo = bucket_base[b], hi = bucket_end[b];
    if (threadIdx.x < 512) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_seen = 0;
    __syncthreads();
    // The sample is every stride2-th GROUP of 256 records (one wave load).  Records of one stretch of text lie together in
    // a level-1 bucket, and a repeat family can put tens of thousands of records of ONE final bucket into such a
    // stretch: sampled in blocks of 1024 records (as at first) a stretch of 87 K records is 5 or 6 blocks -- an estimate
    // 19 % off, beyond the 12.5 % + 4096 of slack, and which blocks are hit depends on the order the tiles claimed
    // their runs in, i.e. on timing (seen as an occasional re-layout at k = 17 when the level-1 grid changed).
    // (a group is what one wave loads at 16 bytes per lane: 1 KiB of contiguous records; with 4 bytes per lane -- groups of 64
    // records, 256 bytes out of every 4 KiB -- the same sample took 0.15 ms instead of 0.06)
    const uint32_t n_grp = (hi - lo + 255u) / 256u, n_sgrp = (n_grp + stride2 - 1u) / stride2;
    const uint32_t mask = pl.B2 - 1u, shift = pl.fb_bits, w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t mine = 0;
    for (uint32_t s0 = 0; s0 < n_sgrp; s0 += 8u * 16u) {                 // sixteen waves, eight loads in flight per lane
        uint4 v[8];
        uint32_t n_ok[8];                                                // how many of the lane's four records exist
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t sg = s0 + (uint32_t)u * 16u + w;
            const uint64_t i = (uint64_t)lo + (uint64_t)sg * stride2 * 256u + lane * 4u;      // bucket starts are 16-byte aligned
            n_ok[u] = (sg < n_sgrp && i < hi) ? (uint32_t)min((uint64_t)4, (uint64_t)hi - i) : 0u;
            v[u] = n_ok[u] ? *reinterpret_cast<const uint4 *>(in + i) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t r4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (uint32_t e = 0; e < 4; e++)
                if (e < n_ok[u]) { atomicAdd(&h[(r4[e] >> shift) & mask], 1u); mine++; }
        }
    }
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&n_seen, mine);
    __syncthreads();
    if (threadIdx.x < pl.B2) tally[(uint64_t)b * pl.B2 + threadIdx.x] = h[threadIdx.x];
    if (threadIdx.x == 0) sampled_n[b] = n_seen;
}
// room of every final bucket (its tally scaled by its level-1 bucket's size / sampled records, + 12.5 % + slack; exact
// tallies as they are), its start inside its block of 1024 final buckets, the block totals
__global__ __launch_bounds__(1024) void k_rooms2(const uint32_t *__restrict__ tally, const uint32_t *__restrict__ sampled_n,
                                                 const uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ bucket_end, PartPlan pl,
                                                 uint32_t stride2, uint32_t n_final, uint32_t *__restrict__ final_start,
                                                 uint32_t *__restrict__ cap2_end, uint32_t *__restrict__ block_tot, const uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[16];
    if (flags[0]) return;
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    uint32_t room = 0;
    if (i < n_final) {
        const uint32_t b = i >> pl.b2, hcount = tally[i];
        if (stride2 == 1u) room = hcount;
        else {
            const uint32_t seen = sampled_n[b], n_b = bucket_end[b] - bucket_base[b];
            const unsigned long long est = seen ? (unsigned long long)((double)hcount * ((double)n_b / (double)seen)) + 1ull : 0ull;
            // 25 % + 4096 of slack: which records the sample sees depends on the order the level-1 tiles claimed their runs in,
            // i.e. on timing, and with 12.5 % about one k = 17 step in forty outgrew a bucket and paid the exact re-layout
            room = (uint32_t)(est + est / 4 + 4096u);
        }
        room = (room + 7u) & ~7u;                                        // 16-bit records: starts stay 16-byte aligned
    }
    uint32_t inc = room;                                                 // exclusive scan over the block's 1024 rooms
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
    for (int j = 0; j < 16; j++) { if (j < w) pre += wsum[j]; total += wsum[j]; }
    if (i < n_final) { final_start[i] = pre + inc - room; cap2_end[i] = room; }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
// block bases (one workgroup: <= 256 blocks), then absolute starts, cursors and limits
__global__ __launch_bounds__(1024) void k_bases2(uint32_t n_blocks, uint32_t n_final, PartPlan pl, uint32_t *__restrict__ block_tot,
                                                 uint32_t *__restrict__ final_start, uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[16];
    if (flags[0]) return;
    const uint32_t v = threadIdx.x < n_blocks ? block_tot[threadIdx.x] : 0u;
    uint32_t inc = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
    for (int j = 0; j < 16; j++) { if (j < w) pre += wsum[j]; total += wsum[j]; }
    if (threadIdx.x < n_blocks) block_tot[threadIdx.x] = pre + inc - v;
    // the sum of all rooms in 64 bits (the 32-bit scan above wraps silently if it ever passed 2^32; feed_piece keeps
    // capacity2 below that, so this can only fire on an internal error -- but then it does fire)
    __shared__ unsigned long long total64;
    if (threadIdx.x == 0) total64 = 0ull;
    __syncthreads();
    unsigned long long mine = v;
    for (int d = 32; d; d >>= 1) mine += __shfl_down(mine, d, 64);
    if (lane == 0 && mine) atomicAdd(&total64, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        final_start[n_final] = total;
        if (total64 > pl.capacity2) flags[0] = 1u;                       // cannot happen with the bounds of make_part_plan; be loud if it does
    }
}
__global__ __launch_bounds__(1024) void k_starts2(uint32_t n_final, const uint32_t *__restrict__ block_base, uint32_t *__restrict__ final_start,
                                                  uint32_t *__restrict__ cursor2, uint32_t *__restrict__ cap2_end, const uint32_t *__restrict__ flags) {
    if (flags[0]) return;
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    if (i >= n_final) return;
    const uint32_t a = block_base[blockIdx.x] + final_start[i];
    final_start[i] = a; cursor2[i] = a; cap2_end[i] = a + cap2_end[i];
}

// CLAIM: no precomputed offsets; every tile claims room for its runs from the final buckets' cursors, inside the
// room k_provision gave each of them (cap_end; a bucket that outgrows it raises flags[0], see part_common.h).  Where
// a record lands inside its final bucket then depends on timing -- the bucket's contents as a multiset do not.
// The CLAIM launch is persistent (two workgroups per CU) over work items of R2 records and XCD-affine: workgroup x
// runs on XCD x % 8 (round-robin dispatch), and takes items of the level-1 buckets b with b % 8 == x % 8 only.  All
// writers of one final bucket then share one L2: the partly written cache lines where one tile's run ends and the
// next one's begins are merged there instead of travelling to HBM twice, and a bucket's cursor lives in one L2.
// `pos` (wg2_start + B1 + 1) is the running item count over the buckets in that order (k_level1_finish).
// NT threads x PER records = one tile.  The CLAIM launch runs as 512 x 32: two workgroups share a CU (72 KiB of LDS and
// 128 registers each), so one sorts while the other waits at a barrier or for its stores.
// REC24: the level-1 records are 3-byte records in two planes (32-bit k-mers; part_common.h), `in` is the 16-bit plane.
// (Measured and kept out: the claiming launch as 1024 x 16 at 64 registers -- two workgroups = 32 waves per CU instead of 16:
// 1.14 -> 1.19 ms with write-out batches of four, 2.1 ms with eight (46 spilled registers).)
template <bool CLAIM, int NT, int PER, bool REC24 = false>
__global__ __launch_bounds__(NT, 4) void k_scatter2(const uint32_t *__restrict__ in, const uint32_t *__restrict__ wg2_start,
                                                   const uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ bucket_end,
                                                   const uint32_t *__restrict__ rowoff, const uint32_t *__restrict__ final_start, PartPlan pl,
                                                   void *__restrict__ out, uint32_t *__restrict__ cursor, const uint32_t *__restrict__ cap_end,
                                                   uint32_t dump, uint32_t *__restrict__ flags, uint32_t xcd_affine) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ScatterLds &L = *reinterpret_cast<ScatterLds *>(smem);
    if (flags[0]) return;
    const uint32_t B = pl.B2, shift = pl.fb_bits;
    const uint32_t low_mask = (1u << shift) - 1u;
    for (uint32_t i = threadIdx.x; i < 512; i += NT) { L.hist[i] = 0; L.run[i] = 0; }
    __syncthreads();
#ifdef PK_PHASE_PROF
    unsigned long long phase_prof[5] = {0, 0, 0, 0, 0};                   // thread 0: unpack, count, scan, park, store (cycles)
#endif
    // records [lo, hi) of level-1 bucket b: 16-byte aligned windows of TILE records, the first / last partly masked.
    // The next window's loads are issued before the current tile is sorted, so they fly during its barriers.
    typedef uint32_t Quad __attribute__((ext_vector_type(4)));
    typedef uint32_t Pair __attribute__((ext_vector_type(2)));
    const uint16_t *in_lo = reinterpret_cast<const uint16_t *>(in);
    const uint8_t *in_hi = reinterpret_cast<const uint8_t *>(in) + level1_hi_plane_offset(pl.capacity1);
    auto item = [&](uint32_t b, uint32_t lo, uint32_t hi) {
        // four records per lane and load: 16 bytes of 4-byte records, or 8 bytes of the low plane + 4 of the high plane
        // (x, y: the low halves of records 0-1 and 2-3; z: the four high bytes)
        auto fetch = [&](uint32_t win, Quad (&v)[PER / 4]) {
            const uint32_t v_hi = min(hi, win + (uint32_t)TILE);
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
                const uint32_t i = win + (threadIdx.x + j * NT) * 4u;
                v[j] = Quad{0u, 0u, 0u, 0u};
                if (i < v_hi) {
                    if (REC24) {
                        const Pair l = *reinterpret_cast<const Pair *>(in_lo + i);
                        v[j] = Quad{l.x, l.y, *reinterpret_cast<const uint32_t *>(in_hi + i), 0u};
                    } else v[j] = *reinterpret_cast<const Quad *>(in + i);
                }
            }
        };
        Quad nxt[PER / 4];
        auto settle = [&]() {                                            // see scatter_tile
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int j = 0; j < PER / 4; j++) asm volatile("" : "+v"(nxt[j]));
        };
        fetch(lo & ~3u, nxt);
        settle();
        for (uint32_t win = lo & ~3u; win < hi; win += TILE) {
#ifdef PK_PHASE_PROF
            const unsigned long long tile_t0 = __builtin_readcyclecounter();
#endif
            const uint32_t v_lo = max(lo, win), v_hi = min(hi, win + (uint32_t)TILE);
            const uint32_t n_tile = v_hi - v_lo;
            uint32_t r[PER];
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
                if (REC24) {                                             // record e = low half e | high byte e << 16: one byte permute each
                    r[j * 4] = __builtin_amdgcn_perm(nxt[j].z, nxt[j].x, 0x0c040100u);
                    r[j * 4 + 1] = __builtin_amdgcn_perm(nxt[j].z, nxt[j].x, 0x0c050302u);
                    r[j * 4 + 2] = __builtin_amdgcn_perm(nxt[j].z, nxt[j].y, 0x0c060100u);
                    r[j * 4 + 3] = __builtin_amdgcn_perm(nxt[j].z, nxt[j].y, 0x0c070302u);
                } else { r[j * 4] = nxt[j].x; r[j * 4 + 1] = nxt[j].y; r[j * 4 + 2] = nxt[j].z; r[j * 4 + 3] = nxt[j].w; }
            }
            const bool full = v_lo == win && n_tile == (uint32_t)TILE;   // uniform; nearly every tile: items start 16-byte aligned
            uint32_t okm = 0;
            if (!full) {
#pragma unroll
                for (int j = 0; j < PER / 4; j++) {
                    const uint32_t i = win + (threadIdx.x + j * NT) * 4u;
#pragma unroll
                    for (int e = 0; e < 4; e++) okm |= (i + e >= v_lo && i + e < v_hi) ? (1u << (j * 4 + e)) : 0u;
                }
            }
            if (win + TILE < hi) fetch(win + TILE, nxt);
            uint32_t *cl = CLAIM ? cursor + (uint64_t)b * B : nullptr;
            const uint32_t *ce = CLAIM ? cap_end + (uint64_t)b * B : nullptr;
#ifdef PK_PHASE_PROF
            if (threadIdx.x == 0) phase_prof[0] += __builtin_readcyclecounter() - tile_t0;
            unsigned long long *prof = phase_prof;
#else
            unsigned long long *prof = nullptr;
#endif
            if (full) scatter_tile<uint32_t, false, NT, PER, 512, true, PK_PB_L2, PK_SB_L2, REC24>(L, r, 0u, n_tile, shift, B, low_mask, true, out, settle, cl, ce, dump, flags, nullptr, prof);
            else scatter_tile<uint32_t, false, NT, PER, 512, false, PK_PB_L2, PK_SB_L2, REC24>(L, r, okm, n_tile, shift, B, low_mask, true, out, settle, cl, ce, dump, flags, nullptr, prof);
        }
    };
    if (!CLAIM) {
        uint32_t b, lo, hi;
        if (!wg2_range(wg2_start, bucket_base, bucket_end, pl, b, lo, hi)) return;  // uniform per workgroup
        if (threadIdx.x < B) L.run[threadIdx.x] = final_start[(uint64_t)b * B + threadIdx.x] + rowoff[(uint64_t)blockIdx.x * B + threadIdx.x];
        __syncthreads();
        item(b, lo, hi);
        return;
    }
    const uint32_t *pos = wg2_start + pl.B1 + 1;
    const uint32_t per = pl.B1 >> 3;                                     // level-1 buckets per XCD class (B1 >= 16 with two levels)
    uint32_t o_lo = 0, o_hi = pl.B1, first = blockIdx.x, step = gridDim.x;
    if (xcd_affine) { const uint32_t c = blockIdx.x & 7u; o_lo = c * per; o_hi = o_lo + per; first = blockIdx.x >> 3; step = gridDim.x >> 3; }
    const uint32_t w_end = pos[o_hi];
    for (uint32_t w = pos[o_lo] + first; w < w_end; w += step) {         // uniform per workgroup
        uint32_t a = o_lo, z = o_hi;                                     // last position with pos[a] <= w
        while (z - a > 1) { const uint32_t m = (a + z) >> 1; if (pos[m] <= w) a = m; else z = m; }
        const uint32_t b = (a % per) * 8u + a / per;
        const uint64_t s = (uint64_t)bucket_base[b] + (uint64_t)(w - pos[a]) * pl.R2;
        const uint64_t e = min(s + pl.R2, (uint64_t)bucket_end[b]);
        item(b, (uint32_t)s, (uint32_t)e);
    }
#ifdef PK_PHASE_PROF
    if (threadIdx.x == 0)
        for (int i = 0; i < 5; i++) atomicAdd(reinterpret_cast<unsigned long long *>(flags) - 1 + 9 + i, phase_prof[i]);   // behind the level-1 kernel's five
#endif
}

// ------------------------------------------------------------------ K6: count in LDS ------------
// One workgroup per final bucket (2^fb_bits addresses, fb_bits <= 16).  Counters are 16 bit, two per
// LDS dword; a bucket with more than 65024 records is folded in pieces with a clamp between them so a
// counter (<= 255 + 65024) can never carry into its neighbour.
constexpr uint32_t K6_PIECE = 65024;   // multiple of 8
constexpr int K6_BYTES_LDS = 65536 + 1024 + 128;   // k_bucket_count_bytes: slice image, histogram bins, wrap flag

// `fresh` = first feed after a reset: the table holds nothing yet (it is not even zeroed), so slices
// are not read back and buckets without records are written as zeros.
//
// The value histogram behind Header.update_stats (tools.py:246-263) is maintained here as well: each
// workgroup adds the net change it made (256 signed counters, a handful of them non-zero) into one of
// HIST_REPLICAS copies in HBM, k_hist_reduce adds the copies to the running histogram, and finish() needs no
// pass over the table (3.3 ms at k=17).  (Until late in round 2 every workgroup wrote its own row of 256: 0.5 GB
// written and read again at k=17.)  Two ways
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

// two 16-bit counters -> both clamped to 255 (one packed min), and the low bytes of four such halves -> one table dword
typedef unsigned short U16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t clamp255_pair(uint32_t x) {
    const U16x2 lim = {255, 255};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(U16x2, x), lim));
}
__device__ __forceinline__ uint32_t pack_low_bytes(uint32_t x0, uint32_t x1) {      // 0x00bb00aa, 0x00dd00cc -> 0xddccbbaa
    return __builtin_amdgcn_perm(x1, x0, 0x06040200u);
}

struct SliceTally {
    int d1 = 0, d2 = 0;
    // Four table bytes: how many equal 1, how many equal 2 (two bit planes and a population count each); whatever is 3 or
    // more goes to the LDS bins one byte at a time.  Bytes above 3 are first taken out of the planes.  (A third plane -- the
    // values 1 .. 7 by seven population counts, only bytes >= 8 through the bins -- was measured on the dense k = 15 table,
    // where 81 M addresses hold 3 or more: 0.76 -> 0.87 ms; five more live tallies in a kernel capped at 64 registers.)
    __device__ __forceinline__ void add_dword(int *dh, uint32_t x, int sign) {
        uint32_t p0 = x & 0x01010101u, p1 = (x >> 1) & 0x01010101u;
        uint32_t rest = p0 & p1;                                         // bytes equal to 3 (if nothing above)
        const uint32_t high = x & 0xfcfcfcfcu;
        if (high) {                                                      // some byte >= 4 (a few per cent of the dwords)
            const uint32_t big = (((high & 0x7f7f7f7fu) + 0x7f7f7f7fu) | high) & 0x80808080u;   // bit 7 of every such byte
            const uint32_t keep = ~(big >> 7);
            p0 &= keep; p1 &= keep;
            rest = (p0 & p1) | (big >> 7);
        }
        const int n1 = (int)__builtin_popcount(p0 & ~p1), n2 = (int)__builtin_popcount(p1 & ~p0);
        d1 += sign * n1; d2 += sign * n2;
        while (rest) {
            const int bit = __ffs(rest) - 1;                             // 0, 8, 16 or 24
            atomicAdd(&dh[(x >> bit) & 0xffu], sign);
            rest &= rest - 1u;
        }
    }
};

// `split_bits` > 0 (sparse tables, k=17): a bucket is shared by 2^split_bits workgroups, each reading all of the
// bucket's (few) records but counting only its own part of the address range -- the LDS counters shrink
// with the part, so several workgroups fit on a CU and hide each other's phases.
// LEAN (whole-bucket kernel, i.e. dense tables): records of a dense bucket are counted by eight no-return LDS adds and nothing
// else.  The half-bucket kernel (sparse tables, 64 registers) keeps the form it was tuned with: the same change there moved its
// register allocation and cost 11 % (5.3 -> 5.9 ms at k = 17).
template <int T, bool LEAN, bool NOREC = false>
__device__ __forceinline__ void bucket_count_body(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start,
                                                  const uint32_t *__restrict__ final_end, uint32_t fb_bits, uint32_t split_bits,
                                                  uint8_t *__restrict__ table8, uint32_t fresh, unsigned long long *__restrict__ hist_rep, uint8_t *smem,
                                                  int *dh, uint32_t wg) {
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                  // 2^fb_bits / 2 dwords
    const uint32_t fb = wg >> split_bits, part = wg & ((1u << split_bits) - 1u);
    // final buckets lie back to back (end = the next one's start) unless they ARE the provisioned level-1 buckets
    const uint32_t start = final_start[fb], end = final_end ? final_end[fb] : final_start[fb + 1];
    const uint32_t part_bits = fb_bits - split_bits;
    const uint32_t n_addr = 1u << part_bits;                             // addresses this workgroup owns
    uint8_t *slice = table8 + ((uint64_t)fb << fb_bits) + (uint64_t)part * n_addr;
    if (start == end) {
        if (fresh) {                                                     // nothing counted here: the slice is all zero
            if (n_addr >= 16) for (uint32_t g = threadIdx.x; g < n_addr / 16; g += T) reinterpret_cast<uint4 *>(slice)[g] = make_uint4(0, 0, 0, 0);
            else for (uint32_t a = threadIdx.x; a < n_addr; a += T) slice[a] = 0;
        }
        return;                                                          // otherwise the slice stays as it is
    }
    // the first 16 bytes of records every lane will need are requested before the counters are set up, so the
    // load's latency hides behind that phase (sparse buckets need no second load at all)
    const uint32_t base = start & ~7u;                                   // 16-byte aligned vector loads
    const uint32_t i_first = base + threadIdx.x * 8;
    uint4 v_first = make_uint4(0, 0, 0, 0);
    if (i_first < min(base + K6_PIECE, end)) v_first = *reinterpret_cast<const uint4 *>(recs + i_first);
    // sparse bucket: histogram change from the adds.  NOREC (k_bucket_count_half_lean: dense tables, 2^15-address buckets)
    // always takes the slice-difference route, which is right for any bucket of >= 16 addresses: with the choice made at
    // compile time the record-side code is not in that kernel at all (64 registers, 44 bytes of scratch -> 60, none)
    const bool by_rec = NOREC ? false : (((end - start) >> split_bits) < n_addr / 4 || n_addr < 16);
    SliceTally tally;
    for (uint32_t i = threadIdx.x; i < 256; i += T) dh[i] = 0;
    if (fresh) {
        if (n_addr >= 8) for (uint32_t g = threadIdx.x; g < n_addr / 8; g += T) reinterpret_cast<uint4 *>(cnt)[g] = make_uint4(0, 0, 0, 0);
        else for (uint32_t a = threadIdx.x; a < max(n_addr / 2, 1u); a += T) cnt[a] = 0u;
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
            for (int q = 0; q < 8; q++)                                  // bytes b, b + 1 -> the halves of one counter dword
                cnt[g * 8 + q] = __builtin_amdgcn_perm(0u, w[q >> 1], (q & 1) ? 0x0c030c02u : 0x0c010c00u);
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
        if (LEAN && !by_rec) {
            // Whole buckets only (split_bits == 0: every record of the bucket is below 2^fb_bits, the level-2 sort masked
            // it).  Four vector instructions and one LDS add per record: the counter's byte offset, and 1 or 0x10000 by
            // the address's lowest bit.  (Round 1 merged equal neighbours first, for what was left of tandem runs; the
            // level-1 sort now drops those, and the merge cost more instructions than the adds it saved.)
            const bool interior = i >= start && i + 8u <= end;
            uint8_t *cb = reinterpret_cast<uint8_t *>(cnt);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t x = w[q >> 1];
                const uint32_t off = (q & 1) ? ((x >> 15) & 0x1fffcu) : ((x << 1) & 0x1fffcu);
                const uint32_t low = (q & 1) ? ((x >> 16) & 1u) : (x & 1u);
                uint32_t val = low * 0xffffu + 1u;
                if (!interior) val = (i + q >= start && i + q < end) ? val : 0u;
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(cb + off), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
        }
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
        const bool interior = i >= start && i + 8u <= end;             // all eight records belong to the bucket (nearly every lane)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = i + q;
            const uint32_t full = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
            in[q] = (interior || (idx >= start && idx < end)) && (full >> part_bits) == part;
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
                    cnt[a] = clamp255_pair(cnt[a]);
                }
                __syncthreads();
            }
        }
    }
    // clamp to u8 and write the slice back, 16 addresses per lane
    if (n_addr >= 16) {
        for (uint32_t g = threadIdx.x; g < n_addr / 16; g += T) {
            const uint4 c0 = reinterpret_cast<const uint4 *>(cnt)[g * 2], c1 = reinterpret_cast<const uint4 *>(cnt)[g * 2 + 1];
            const uint32_t o[4] = {pack_low_bytes(clamp255_pair(c0.x), clamp255_pair(c0.y)), pack_low_bytes(clamp255_pair(c0.z), clamp255_pair(c0.w)),
                                   pack_low_bytes(clamp255_pair(c1.x), clamp255_pair(c1.y)), pack_low_bytes(clamp255_pair(c1.z), clamp255_pair(c1.w))};
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
    // the net change goes to one of HIST_REPLICAS copies of the 256 bins (a handful of non-zero bins per workgroup)
    for (uint32_t i = threadIdx.x; i < 256; i += T) {
        const int v = dh[i];
        if (v) atomicAdd(&hist_rep[(uint64_t)(blockIdx.x % HIST_REPLICAS) * 256 + i], (unsigned long long)(long long)v);
    }
}

// Two entry points for the same body.  A whole bucket (dense tables) needs all 128 KiB of LDS a workgroup may
// have, so one workgroup sits on a CU whatever its register count -- no cap.  Half buckets (sparse tables) are
// meant to run two to a CU, which takes at most 64 vector registers.
template <int T>
__global__ __launch_bounds__(T) void k_bucket_count(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start,
                                                    const uint32_t *__restrict__ final_end, uint32_t fb_bits, uint32_t split_bits,
                                                    uint8_t *__restrict__ table8, uint32_t fresh, unsigned long long *__restrict__ hist_rep,
                                                    const uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int dh[256];
    if (flags[0]) return;
    bucket_count_body<T, true>(recs, final_start, final_end, fb_bits, split_bits, table8, fresh, hist_rep, smem, dh, blockIdx.x);
}
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bucket_count_half(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start, const uint32_t *__restrict__ final_end,
                         uint32_t fb_bits, uint32_t split_bits, uint8_t *__restrict__ table8, uint32_t fresh, unsigned long long *__restrict__ hist_rep,
                         const uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int dh[256];
    if (flags[0]) return;
    bucket_count_body<T, false>(recs, final_start, final_end, fb_bits, split_bits, table8, fresh, hist_rep, smem, dh, blockIdx.x);
}

template <int T, bool FRESH>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bucket_count_half_lean(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start, const uint32_t *__restrict__ final_end,
                              uint32_t fb_bits, uint32_t split_bits, uint8_t *__restrict__ table8, uint32_t fresh, unsigned long long *__restrict__ hist_rep,
                              const uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int dh[256];
    if (flags[0]) return;
    // only ever launched for whole buckets of 2^15 addresses, and per state of the table (first feed after a reset or
    // not): as literals, the slice loops have compile-time trip counts and the other state's code is not in the kernel
    bucket_count_body<T, true, true>(recs, final_start, final_end, 15u, 0u, table8, FRESH ? 1u : 0u, hist_rep, smem, dh, blockIdx.x);
}

// Sparse tables (k = 17: ~3000 records per final bucket of 2^16 addresses) -- BYTE counters.  One workgroup per final
// bucket keeps the bucket's 64 KiB slice of the table in LDS *as it will lie in HBM*: a record adds 1 << 8 * (a & 3) to
// the dword holding its byte, the add returns the byte's previous value (which is what the histogram needs), and the
// slice leaves LDS by a straight 16-byte copy -- no 16-bit counters to clamp and pack, no second workgroup reading the
// same records, half the LDS zeroing per table byte (k_bucket_count_half: two workgroups per bucket, each with 2^15 16-bit
// counters, each reading all of the bucket's records).
// A byte cannot saturate: the add that finds 255 wraps it and carries into its neighbour.  That add SEES the 255, so it
// raises a flag, and the workgroup then throws its LDS image away and counts the bucket again the old way (both halves,
// one after the other, 16-bit counters with clamping).  Hot k-mers of period <= 3 never get here (side list), so this is
// the rare bucket that holds a k-mer more than 254 times (or one already saturated by an earlier feed).
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_bucket_count_bytes(const uint16_t *__restrict__ recs, const uint32_t *__restrict__ final_start, const uint32_t *__restrict__ final_end,
                          uint8_t *__restrict__ table8, uint32_t fresh, unsigned long long *__restrict__ hist_rep,
                          const uint32_t *__restrict__ flags) {
    // 64 KiB: the slice, then the 256 histogram bins and the wrap flag.  No static LDS in this kernel: it would sit in
    // front of the slice and push its base off a 128-byte line (a 4-byte static made it 1040), which the 16-byte LDS
    // accesses of the zeroing and the copy-out pay for with bank conflicts (6.6 -> 6.4 ms at the time; the large cost of
    // the first version was the lanes without records, see the record loop below).
    extern __shared__ __attribute__((aligned(128))) uint8_t smem[];
    constexpr uint32_t N_ADDR = 65536u;
    int *dh = reinterpret_cast<int *>(smem + N_ADDR);
    uint32_t *wrap_flag = reinterpret_cast<uint32_t *>(smem + N_ADDR + 1024);
    if (flags[0]) return;
    const uint32_t fb = blockIdx.x;
    const uint32_t start = final_start[fb], end = final_end ? final_end[fb] : final_start[fb + 1];
    uint4 *slice = reinterpret_cast<uint4 *>(table8 + ((uint64_t)fb << 16));
    uint4 *img = reinterpret_cast<uint4 *>(smem);
    if (start == end) {
        if (fresh) for (uint32_t g = threadIdx.x; g < N_ADDR / 16; g += T) slice[g] = make_uint4(0, 0, 0, 0);
        return;
    }
    // A bucket with many records is nearly always one hot address (what the period-1..3 test of the level-1 sort let
    // through of a repeat family): eight unmerged adds per lane on ONE LDS address, only to find the byte wrapped.  Such a
    // bucket goes to the 16-bit counters at once (they merge equal neighbours and clamp between pieces).
    bool recount = end - start >= 16384u;                                // uniform
    int d1 = 0, d2 = 0;
    if (!recount) {
        const uint32_t base = start & ~7u;                               // 16-byte aligned vector loads, 8 records per lane
        const uint32_t i_first = base + threadIdx.x * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i_first < end) v = *reinterpret_cast<const uint4 *>(recs + i_first);
        for (uint32_t i = threadIdx.x; i < 256; i += T) dh[i] = 0;
        if (threadIdx.x == 0) *wrap_flag = 0u;
        if (fresh) { for (uint32_t g = threadIdx.x; g < N_ADDR / 16; g += T) img[g] = make_uint4(0, 0, 0, 0); }
        else { for (uint32_t g = threadIdx.x; g < N_ADDR / 16; g += T) img[g] = slice[g]; }
        __syncthreads();
        uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);
        bool wrapped = false;
        for (uint32_t p0 = base; p0 < end; p0 += (uint32_t)T * 8u) {    // at most three rounds
            const uint32_t i = p0 + threadIdx.x * 8;
            // lanes without records stay out of LDS: their eight adds (of zero) would all meet on address 0, a 64-way
            // same-address conflict per wave instruction -- 1.5 G LDS conflict cycles per launch against 0.3 G, 6.4 ms against 3.6
            if (i >= end) continue;
            if (p0 != base) v = *reinterpret_cast<const uint4 *>(recs + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const bool interior = i >= start && i + 8u <= end;           // nearly every lane
            uint32_t old[8];
            bool in[8];
            // eight adds back to back (records outside the bucket add zero), their returned bytes looked at afterwards
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t a = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu, sh = 8u * (a & 3u);
                in[q] = interior || (i + q >= start && i + q < end);
                old[q] = (atomicAdd(&cnt[a >> 2], in[q] ? (1u << sh) : 0u) >> sh) & 0xffu;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t c = old[q];
                if (!in[q]) continue;
                if (c == 0u) d1++;
                else if (c == 1u) { d2++; d1--; }
                else if (c == 255u) wrapped = true;
                else { atomicAdd(&dh[c + 1u], 1); atomicAdd(&dh[c], -1); }
            }
        }
        if (wrapped) *wrap_flag = 1u;
        __syncthreads();
        recount = *wrap_flag != 0u;                                      // uniform
        __syncthreads();                                                 // the bins and counters may be set up again below
    }
    if (recount) {
        // count the bucket with 16-bit counters, one half of the address range after the other
        if (threadIdx.x == 0) atomicAdd(const_cast<uint32_t *>(&flags[1]), 1u);               // statistics: pk_indexer_timings [9]
        bucket_count_body<T, false>(recs, final_start, final_end, 16u, 1u, table8, fresh, hist_rep, smem, dh, 2u * fb);
        __syncthreads();
        bucket_count_body<T, false>(recs, final_start, final_end, 16u, 1u, table8, fresh, hist_rep, smem, dh, 2u * fb + 1u);
        return;
    }
    for (uint32_t g = threadIdx.x; g < N_ADDR / 16; g += T) slice[g] = img[g];
    d1 = wave_sum_i32(d1); d2 = wave_sum_i32(d2);
    if ((threadIdx.x & 63) == 0) {
        if (d1) atomicAdd(&dh[1], d1);
        if (d2) atomicAdd(&dh[2], d2);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += T) {
        const int x = dh[i];
        if (x) atomicAdd(&hist_rep[(uint64_t)(blockIdx.x % HIST_REPLICAS) * 256 + i], (unsigned long long)(long long)x);
    }
}

// sums the HIST_REPLICAS copies of the feed's histogram change into the running 256-bin histogram (signed deltas: two's
// complement adds) and leaves the copies zeroed for the next feed (they are zeroed once more when the workspace is
// allocated): called by the first workgroup of k_apply_side, 256 threads of it
__device__ __forceinline__ void hist_reduce(unsigned long long *__restrict__ hist_rep, unsigned long long *__restrict__ hist) {
    if (threadIdx.x >= 256) return;
    unsigned long long acc = 0;
    for (uint32_t r = 0; r < HIST_REPLICAS; r++) {
        acc += hist_rep[(uint64_t)r * 256 + threadIdx.x];
        hist_rep[(uint64_t)r * 256 + threadIdx.x] = 0ull;
    }
    if (acc) atomicAdd(&hist[threadIdx.x], acc);
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
                                                   uint64_t side_cap, uint8_t *__restrict__ table8, unsigned long long *__restrict__ hist,
                                                   unsigned long long *__restrict__ hist_rep, const uint32_t *__restrict__ flags) {
    __shared__ unsigned long long key[AS_SLOTS];
    if (flags[0]) return;
    static_assert(WG >= 256, "one thread per histogram bin");
    if (blockIdx.x == 0) hist_reduce(hist_rep, hist);       // the bucket-count kernel's histogram change (launched before this one)
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
PartPlan make_part_plan(uint32_t k, uint64_t n_bytes, uint32_t slice_bits, uint32_t slice_index) {
    PartPlan pl;
    pl.k = k;
    pl.slice_bits = slice_bits;                            // the table holds one of 2^slice_bits equal address ranges ...
    pl.slice_index = slice_index;                          // ... this one
    pl.addr_bits = 2 * k - slice_bits;
    pl.fb_bits = pl.addr_bits < 16 ? pl.addr_bits : 16;
    // Dense tables of 32-bit k-mers (>= 1 input byte per 8 addresses, e.g. a genome at k = 15): final buckets of 2^15
    // addresses, so that TWO bucket-count workgroups (64 KiB of counters each) share a CU and one's load / count /
    // write-back phases hide behind the other's.  Sparse tables get there by other means (k_bucket_count_half).
    static const uint32_t dense_shift = getenv("PK_DENSE_SHIFT") ? (uint32_t)atoi(getenv("PK_DENSE_SHIFT")) : 3u;
    if (k <= 15 && pl.addr_bits >= 24 && n_bytes >= (((uint64_t)1 << pl.addr_bits) >> dense_shift)) pl.fb_bits = 15;
    const uint32_t bucket_bits = pl.addr_bits - pl.fb_bits;
    // one level while its digits fit the LDS arrays of the sort kernel that runs level 1 (kmer_fuse.hip: 128 digits for
    // 32-bit k-mers, 512 for 64-bit ones), two levels of about equal width otherwise
    const uint32_t one_level_max = k <= 15 ? 7u : 9u;
    pl.b1 = bucket_bits <= one_level_max ? bucket_bits : (bucket_bits + 1) / 2;
    if (pl.b1 > one_level_max) pl.b1 = one_level_max;      // 15 bits = 7 + 8: level 2 sorts up to 512 ways
    pl.b2 = bucket_bits - pl.b1;
    pl.B1 = 1u << pl.b1;
    pl.B2 = 1u << pl.b2;
    pl.n_chunks = (uint32_t)((n_bytes + CHUNK - 1) / CHUNK);
    pl.n_wg0 = pl.n_chunks < 1024u ? pl.n_chunks : 1024u;
    if (pl.n_wg0 == 0) pl.n_wg0 = 1;
    pl.G = (pl.n_chunks + pl.n_wg0 - 1) / pl.n_wg0;
    if (pl.G == 0) pl.G = 1;
    static const uint32_t wg1_env = getenv("PK_WG1") ? (uint32_t)atoi(getenv("PK_WG1")) : 0u;
    const uint32_t wg1 = wg1_env ? wg1_env : 4096u;       // shorter stretches per workgroup even out the last round (1024: 1.42 ms, 4096: 1.39)
    pl.n_wg1 = pl.n_chunks < wg1 ? pl.n_chunks : wg1;
    if (pl.n_wg1 == 0) pl.n_wg1 = 1;
    pl.G1 = (pl.n_chunks + pl.n_wg1 - 1) / pl.n_wg1;
    if (pl.G1 == 0) pl.G1 = 1;
    uint64_t r2 = (n_bytes + 1023) / 1024;
    pl.R2 = r2 < (uint64_t)TILE ? (uint64_t)TILE : ((r2 + TILE - 1) / TILE) * TILE;
    pl.n_wg2_max = (uint32_t)(n_bytes / pl.R2) + pl.B1 + 1;
    // bucket sizes are estimated from every 16th slot; small inputs are counted exactly
    pl.sample_stride = pl.n_chunks >= 1024u ? 16u : 1u;
    const uint64_t nfb = (uint64_t)pl.B1 * pl.B2;
    pl.n_tally = (pl.b2 && nfb <= 32768 && pl.addr_bits <= 30) ? (uint32_t)nfb : pl.B1;
    pl.sample2 = (pl.b2 && pl.n_tally == pl.B1 && nfb <= 262144 && pl.B1 >= 8u) ? 1u : 0u;
    if (pl.n_tally > pl.B1 || pl.sample2) {                // level 2 claims its room tile by tile: small work items for a persistent grid
        pl.R2 = 4u * TILE;
        pl.n_wg2_max = (uint32_t)(n_bytes / pl.R2) + pl.B1 + 1;
    }
    // room for the buckets: the sampled estimate can reach the slot capacity (+1 per bucket from rounding up), each
    // bucket gets 12.5 % + a constant + alignment on top of it (k_provision)
    const uint64_t est1 = (uint64_t)pl.n_chunks * TILE + pl.B1, est2 = (uint64_t)pl.n_chunks * TILE + nfb;
    pl.capacity1 = est1 + est1 / 8 + (uint64_t)pl.B1 * 4100;
    pl.capacity2 = pl.sample2 ? est2 + est2 / 4 + nfb * 4104 : est2 + est2 / 8 + nfb * 2056;
    return pl;
}

size_t part_workspace_bytes(const PartPlan &pl, uint64_t n_bytes, PartWorkspace *lay) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const uint64_t nfb = (uint64_t)pl.B1 * pl.B2;
    const bool laid_out2 = pl.n_tally > pl.B1 || pl.sample2; // final buckets provisioned from an estimate
    size_t o = 0;
    lay->codes = o; o += up((size_t)pl.n_chunks * SLOT_CODE_WORDS * 4);
    lay->restarts = o; o += up((size_t)pl.n_chunks * SLOT_RST_WORDS * 4);
    lay->n_bases = o; o += up((size_t)pl.n_chunks * 4);
    lay->tally_rows = o; o += up((size_t)COUNT_WGS * pl.n_tally * 4);
    lay->tally_tot = o; o += up(pl.sample2 ? (size_t)nfb * 4 + (size_t)(pl.B1 + 1024) * 4 : (size_t)pl.n_tally * 4);   // sample2: + sampled counts, block totals
    lay->bucket_base = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->bucket_end = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->compact_base = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->cursor1 = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->cap_end = o; o += up((size_t)(pl.B1 + 1) * 4);
    lay->wg2_start = o; o += up((size_t)(pl.B1 + 1) * 8);   // by bucket, and in XCD-class order
    lay->final_start = o; o += up((size_t)(nfb + 1) * 4);
    lay->cursor2 = o; o += up((size_t)(nfb + 1) * 4);
    lay->cap2_end = o; o += up((size_t)(nfb + 1) * 4);
    // level-1 buckets + the dump tile: 2-byte records when level 1 is the only level, two planes of 2 + 1 bytes for 32-bit
    // k-mers with a second level (part_common.h), 4-byte records (the low 32 bits) for 64-bit k-mers
    lay->out1 = o; o += !pl.b2 ? up((size_t)(pl.capacity1 + TILE + 64) * 2)
                     : pl.k <= 15 ? level1_hi_plane_offset(pl.capacity1) + up((size_t)(pl.capacity1 + TILE + 64))
                                  : up((size_t)(pl.capacity1 + TILE + 64) * 4);
    lay->hist2 = o; o += up(laid_out2 ? 256 : (size_t)pl.n_wg2_max * pl.B2 * 4);          // per-workgroup digit counts: only without claims
    lay->rowoff2 = o; o += up(laid_out2 ? 256 : (size_t)pl.n_wg2_max * pl.B2 * 4);
    lay->out2 = o; o += up(!pl.b2 ? 256 : laid_out2 ? (size_t)(pl.capacity2 + TILE + 64) * 2 : (size_t)(n_bytes + 64) * 2);
    lay->side_cap = n_bytes + 16;                          // every side entry stands for >= 1 k-mer
    lay->side = o; o += up((size_t)lay->side_cap * 8);
    lay->side_n = o; o += 256;                             // side-list length (u64), then the flags word
    return o;
}

void part_set_attributes() {
    fuse_set_attributes();
    hipFuncSetAttribute((const void *)k_scatter2<false, SC_T, SC_PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_scatter2<true, 512, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_scatter2<true, 512, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_bucket_count<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)k_bucket_count_half<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)k_bucket_count_half_lean<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)k_bucket_count_half_lean<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)k_bucket_count_bytes<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, K6_BYTES_LDS);
}

// diagnostics (include/pykmer_hip.h; tools/occupancy.py): resident workgroups per CU of the bucket-count / level-2 kernels
// as the runtime computes them
extern "C" int pk_diag_occupancy(int which) {
    int n = -1;
    hipError_t e = hipErrorInvalidValue;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_bucket_count_half<1024>, 1024, 65536);
    if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_bucket_count_bytes<1024>, 1024, K6_BYTES_LDS);
    if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_bucket_count_half_lean<1024, true>, 1024, 65536);
    if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_scatter2<true, 512, 32>, 512, SCATTER_LDS_NARROW);
    return e == hipSuccess ? n : -(int)e;
}

static bool pk_bytes_enabled() {                                         // PK_K6_BYTES=0: the two-halves kernel for sparse tables (comparison runs)
    static const bool on = !(getenv("PK_K6_BYTES") && atoi(getenv("PK_K6_BYTES")) == 0);
    return on;
}

// Everything behind the squeeze pass for one feed: bucket layout (sampled with `stride`; 1 = exact), the fused
// k-mer assembly + level-1 sort, level 2, bucket count, side list.  Returns right after the launches; the
// caller reads the flags word back to learn whether the layout held (flags[0] == 0).
int launch_partitioned(const L2 *st2, uint64_t n_bytes, const PartPlan &pl, uint32_t stride, uint8_t *ws, const PartWorkspace &lay, uint8_t *table8,
                       hipStream_t s, hipEvent_t ev_sort_begin, hipEvent_t ev_sort_end, hipEvent_t ev_part_end, bool fresh,
                       unsigned long long *hist, unsigned long long *bucket_hist, bool armed) {
    const uint32_t *codes = (const uint32_t *)(ws + lay.codes), *restarts = (const uint32_t *)(ws + lay.restarts);
    const uint32_t *n_bases = (const uint32_t *)(ws + lay.n_bases);
    uint32_t *tally_rows = (uint32_t *)(ws + lay.tally_rows), *tally_tot = (uint32_t *)(ws + lay.tally_tot);
    uint32_t *bucket_base = (uint32_t *)(ws + lay.bucket_base), *bucket_end = (uint32_t *)(ws + lay.bucket_end);
    uint32_t *compact_base = (uint32_t *)(ws + lay.compact_base), *cursor1 = (uint32_t *)(ws + lay.cursor1), *cap_end = (uint32_t *)(ws + lay.cap_end);
    uint32_t *wg2_start = (uint32_t *)(ws + lay.wg2_start), *final_start = (uint32_t *)(ws + lay.final_start);
    uint32_t *cursor2 = (uint32_t *)(ws + lay.cursor2), *cap2_end = (uint32_t *)(ws + lay.cap2_end);
    uint32_t *hist2 = (uint32_t *)(ws + lay.hist2), *rowoff2 = (uint32_t *)(ws + lay.rowoff2);
    void *out1 = ws + lay.out1, *out2 = ws + lay.out2;
    unsigned long long *side = (unsigned long long *)(ws + lay.side), *side_n = (unsigned long long *)(ws + lay.side_n);
    uint32_t *flags = (uint32_t *)(side_n + 1);
    const uint32_t nfb = pl.B1 * pl.B2;
    const bool laid_out2 = pl.n_tally > pl.B1 || pl.sample2;
    if (pl.B1 > (pl.k <= 15 ? 128u : 512u) || pl.B2 > 512u || pl.fb_bits > 16u) return -3;   // what the kernels' LDS arrays are sized for
    if (pl.k <= 15 && pl.b2 && (!laid_out2 || pl.addr_bits - pl.b1 > 24u)) return -3;           // 3-byte level-1 records: only the claiming level 2 reads them
    if (!armed && hipMemsetAsync(side_n, 0, PART_FLAG_WORDS * 4, s) != hipSuccess) return -2;   // side-list length + flags
    launch_provision(codes, restarts, n_bases, st2, pl, stride, tally_rows, tally_tot, bucket_base, cursor1, cap_end, final_start, cursor2, cap2_end,
                     flags, s);
    if (ev_sort_begin) hipEventRecord(ev_sort_begin, s);
    launch_walk_sort(codes, restarts, n_bases, st2, pl, out1, cursor1, cap_end, flags, bucket_base, bucket_end, compact_base, wg2_start, side, side_n,
                     lay.side_cap, s);
    if (ev_sort_end) hipEventRecord(ev_sort_end, s);
    const uint16_t *final_recs = (const uint16_t *)out1;
    const uint32_t *k6_start = bucket_base, *k6_end = bucket_end;        // b2 == 0: the level-1 buckets are the final ones
    if (pl.sample2) {                                                    // final-bucket rooms from a sample of the level-1 records
        const uint32_t stride2 = stride == 1u ? 1u : 16u, n_blocks = (nfb + 1023u) / 1024u;
        uint32_t *sampled_n = tally_tot + nfb, *block_tot = sampled_n + pl.B1;
        hipLaunchKernelGGL(k_sample2, dim3(pl.B1), dim3(1024), 0, s, (const uint32_t *)out1, (const uint32_t *)bucket_base, (const uint32_t *)bucket_end,
                           pl, stride2, tally_tot, sampled_n, (const uint32_t *)flags);
        hipLaunchKernelGGL(k_rooms2, dim3(n_blocks), dim3(1024), 0, s, (const uint32_t *)tally_tot, (const uint32_t *)sampled_n,
                           (const uint32_t *)bucket_base, (const uint32_t *)bucket_end, pl, stride2, nfb, final_start, cap2_end, block_tot,
                           (const uint32_t *)flags);
        hipLaunchKernelGGL(k_bases2, dim3(1), dim3(1024), 0, s, n_blocks, nfb, pl, block_tot, final_start, flags);
        hipLaunchKernelGGL(k_starts2, dim3(n_blocks), dim3(1024), 0, s, nfb, (const uint32_t *)block_tot, final_start, cursor2, cap2_end,
                           (const uint32_t *)flags);
    }
    if (laid_out2) {
        static const uint32_t xcd_affine = getenv("PK_XCD") ? (uint32_t)atoi(getenv("PK_XCD")) : 1u;
        // 512 would be the resident two per CU (1.21 ms); more and shorter walks even out the end: 4096 -> 1.17 ms (k = 17: 1.44 -> 1.38).
        // Measured and kept out: level 2 launched in 2 / 4 / 8 parts (ranges of level-1 buckets) with the bucket count of part p
        // on a second stream beside level 2 of part p + 1 (one workgroup of each fits a CU's LDS): 1.91 ms for the two
        // stages back to back, 2.03 / 2.12 / 2.44 overlapped -- every part pays its own ragged end, and the two kernels
        // do not hide each other (both live on the LDS pipe).
        static const uint32_t grid2_env = getenv("PK_GRID2") ? (uint32_t)atoi(getenv("PK_GRID2")) : 4096u;
        const uint32_t grid2 = grid2_env < 8u ? 8u : (grid2_env & ~7u);    // a multiple of 8: every XCD class gets the same number of workgroups
        if (pl.k <= 15)
            hipLaunchKernelGGL((k_scatter2<true, 512, 32, true>), dim3(grid2), dim3(512), SCATTER_LDS_NARROW, s, (const uint32_t *)out1, wg2_start,
                               bucket_base, bucket_end, (const uint32_t *)nullptr, final_start, pl, out2, cursor2, (const uint32_t *)cap2_end,
                               (uint32_t)pl.capacity2, flags, xcd_affine);
        else
            hipLaunchKernelGGL((k_scatter2<true, 512, 32>), dim3(grid2), dim3(512), SCATTER_LDS_NARROW, s, (const uint32_t *)out1, wg2_start,
                               bucket_base, bucket_end, (const uint32_t *)nullptr, final_start, pl, out2, cursor2, (const uint32_t *)cap2_end,
                               (uint32_t)pl.capacity2, flags, xcd_affine);
        final_recs = (const uint16_t *)out2; k6_start = final_start; k6_end = cursor2;   // a final bucket ends where its cursor stopped
    } else if (pl.b2) {
        hipLaunchKernelGGL(k_count2, dim3(pl.n_wg2_max), dim3(WG), 0, s, (const uint32_t *)out1, wg2_start, bucket_base, bucket_end, pl, hist2,
                           (const uint32_t *)flags);
        hipLaunchKernelGGL(k_rows2_scan, dim3(pl.B1), dim3(512), 0, s, hist2, rowoff2, wg2_start, compact_base, pl, final_start, (const uint32_t *)flags);
        hipLaunchKernelGGL((k_scatter2<false, SC_T, SC_PER>), dim3(pl.n_wg2_max), dim3(SC_T), SCATTER_LDS_NARROW, s, (const uint32_t *)out1, wg2_start,
                           bucket_base, bucket_end, rowoff2, final_start, pl, out2, (uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, flags, 0u);
        final_recs = (const uint16_t *)out2; k6_start = final_start; k6_end = nullptr;
    }
    if (ev_part_end) hipEventRecord(ev_part_end, s);
    // sparse tables (few records per 2^16-address bucket, k=17): 2^split workgroups per bucket, see k_bucket_count
    static const uint64_t sparse_max = getenv("PK_SPARSE_MAX") ? (uint64_t)atoll(getenv("PK_SPARSE_MAX")) : 8192u;
    const bool sparse = pl.fb_bits == 16 && n_bytes / nfb < sparse_max;
    const uint32_t split = sparse ? 1u : 0u;
    const size_t part_addrs = (size_t)1 << (pl.fb_bits - split);
    const size_t lds6 = part_addrs * 2 < 64 ? 64 : part_addrs * 2;
    const uint32_t n_rows6 = (uint32_t)(nfb << split);
    if (split > 1u) return -3;                                           // the kernels are laid out for whole and half buckets
    if (pl.fb_bits == 15 && fresh)                                       // 64 KiB of counters: two workgroups per CU
        hipLaunchKernelGGL((k_bucket_count_half_lean<1024, true>), dim3(n_rows6), dim3(1024), lds6, s, final_recs, k6_start, k6_end, pl.fb_bits, split, table8,
                           fresh ? 1u : 0u, bucket_hist, (const uint32_t *)flags);
    else if (pl.fb_bits == 15)
        hipLaunchKernelGGL((k_bucket_count_half_lean<1024, false>), dim3(n_rows6), dim3(1024), lds6, s, final_recs, k6_start, k6_end, pl.fb_bits, split, table8,
                           fresh ? 1u : 0u, bucket_hist, (const uint32_t *)flags);
    else if (split && pk_bytes_enabled())
        hipLaunchKernelGGL(k_bucket_count_bytes<1024>, dim3(nfb), dim3(1024), K6_BYTES_LDS, s, final_recs, k6_start, k6_end, table8, fresh ? 1u : 0u, bucket_hist,
                           (const uint32_t *)flags);
    else if (split)
        hipLaunchKernelGGL(k_bucket_count_half<1024>, dim3(n_rows6), dim3(1024), lds6, s, final_recs, k6_start, k6_end, pl.fb_bits, split, table8,
                           fresh ? 1u : 0u, bucket_hist, (const uint32_t *)flags);
    else
        hipLaunchKernelGGL(k_bucket_count<1024>, dim3(n_rows6), dim3(1024), lds6, s, final_recs, k6_start, k6_end, pl.fb_bits, split, table8,
                           fresh ? 1u : 0u, bucket_hist, (const uint32_t *)flags);
    hipLaunchKernelGGL(k_apply_side, dim3(AS_WGS), dim3(WG), 0, s, side, side_n, lay.side_cap, table8, hist, bucket_hist, (const uint32_t *)flags);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace pk
