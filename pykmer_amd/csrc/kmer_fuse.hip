// kmer_fuse.hip -- k-mer assembly fused with the level-1 partition: packed bases in, level-1 runs out.
//
// Replaces gen_kmers (indexer.py:130-160), the canonical min (indexer.py:341) and the first half of what
// np.unique + the fragment scatter do in process_kmers (indexer.py:162-297).  Input is the packed stream
// of kmer_pack.hip: per 16 KiB text chunk a slot of 2-bit codes and restart bits.  Because the stream
// holds nothing but valid bases, the k-mer ending at base j is a fixed-offset bit field of two dwords --
// no state is carried from thread to thread -- so a workgroup of 1024 threads takes one slot, 16 bases per
// thread, and keeps its <= 16 K canonical k-mers IN REGISTERS: exactly the tile the LDS counting sort of
// part_common.h wants.  The records never exist in HBM unsorted (round 1 wrote 2.8 GB of them and read
// 3.3 GB back).
//
// Per base:  fwd = sum 4^(k-1-p) b_p   one v_alignbit on the pair-reversed dwords       (indexer.py:149)
//            rev = sum 4^p (3 - b_p)   one 64-bit shift of the complemented dwords       (indexer.py:150)
//            window valid iff no restart bit among the k-1 bases behind its first        (indexer.py:144)
// The per-lane three-entry cache of recent k-mers (part_common.h, hot keys) keeps tandem repeats out of the
// record stream as before.
//
// Where the runs go: level-1 bucket sizes are not known before the k-mers exist.  A sampling launch of the
// same kernel (COUNT: every 16th slot, tally only) estimates them, k_provision lays the buckets out with
// 12.5 % + 4096 records of slack each, and the sort claims room for every run from per-bucket cursors.  A
// bucket that outgrows its room raises a flag (its runs go to a dump area, nothing is overwritten); every
// later kernel of the feed then returns at once and the host repeats from here with stride 1, i.e. with
// exact sizes.  Inputs below 1024 chunks are counted exactly straight away.
#include <cstdlib>
#include "part_common.h"

namespace pk {

__device__ __forceinline__ uint32_t revpairs32(uint32_t x) {            // 2-bit field p -> field 15 - p
    const uint32_t y = __builtin_bitreverse32(x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

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

template <typename KT, bool COUNT>
__global__ __launch_bounds__(SC_T) void k_walk_sort(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ restarts,
                                                    const uint32_t *__restrict__ n_bases, const L2 *__restrict__ chunk_l2_state,
                                                    PartPlan pl, uint32_t n_items, uint32_t stride, void *__restrict__ out,
                                                    uint32_t *__restrict__ cursor1, const uint32_t *__restrict__ cap_end, uint32_t dump,
                                                    uint32_t *__restrict__ flags, uint32_t *__restrict__ fine_rows,
                                                    uint32_t *__restrict__ sample_hist, unsigned long long *__restrict__ side,
                                                    unsigned long long *__restrict__ side_n, uint64_t side_cap, uint32_t dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    ScatterLds &L = *reinterpret_cast<ScatterLds *>(smem);
    constexpr bool WIDE = sizeof(KT) == 8;
    uint32_t *fine = reinterpret_cast<uint32_t *>(smem + (WIDE ? SCATTER_LDS_WIDE : SCATTER_LDS_NARROW));   // [B1 * B2] when fine_rows
    __shared__ HotTable hot;
    const uint32_t k = pl.k, km1 = k - 1;
    const uint32_t n_fine = pl.B1 * pl.B2;
    const bool tally_fine = !COUNT && fine_rows != nullptr;
    if (!COUNT) {
        for (uint32_t i = threadIdx.x; i < HOT_SLOTS; i += SC_T) { hot.key[i] = 0ull; hot.val[i] = 0u; }
        if (threadIdx.x == 0) hot.used = 0;
        if (tally_fine) for (uint32_t i = threadIdx.x; i < n_fine; i += SC_T) fine[i] = 0u;
    }
    if (threadIdx.x < 512) { L.hist[threadIdx.x] = 0; L.run[threadIdx.x] = 0; }
    const uint32_t B = pl.B1, shift = pl.addr_bits - pl.b1;
    const uint32_t low_mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
    const bool out16 = pl.b2 == 0;
    const KT mask = (KT)((2u * k >= sizeof(KT) * 8u) ? ~(KT)0 : (((KT)1 << (2u * k)) - 1));
    const uint32_t t = threadIdx.x;
    __syncthreads();
    // items: this workgroup's slots (persistent over a contiguous range), or every stride-th slot when sampling
    const uint32_t i_lo = COUNT ? blockIdx.x : blockIdx.x * pl.G, i_hi = COUNT ? n_items : min(i_lo + pl.G, n_items);
    const uint32_t i_step = COUNT ? gridDim.x : 1u;
    auto no_settle = []() {};
    for (uint32_t it = i_lo; it < i_hi; it += i_step) {
        const uint32_t c = COUNT ? it * stride : it;
        const uint32_t nb = n_bases[c];                                   // uniform
        if (nb == 0) continue;
        // ---- this thread's 16 bases, the 16 before them, and the restart bits of both
        const uint32_t *cw = codes + (uint64_t)c * SLOT_CODE_WORDS;
        const uint32_t *rw = restarts + (uint64_t)c * SLOT_RST_WORDS;
        const bool live = 16u * t < nb;
        uint32_t cur = 0, prev = 0, rr = 0;
        if (live) {
            cur = cw[t];
            const uint32_t r_here = rw[t >> 1];
            if (t == 0) {
                // the k-1 bases in front of the slot: the chunk's start state (newest base lowest) in stream order
                const L2 st = chunk_l2_state[c];
                const uint32_t len = l2_len(st);
                prev = revpairs32(st.bits);
                rr = r_here << 16;
                if (len < km1) rr |= 1u << (16u - len);                   // nothing older than those `len` bases may be used
            } else {
                prev = cw[t - 1];
                rr = (t & 1u) ? r_here : ((r_here << 16) | (rw[(t >> 1) - 1] >> 16));
            }
        }
        const uint32_t cnt = live ? min(16u, nb - 16u * t) : 0u;
        // window ending at base j (bit 16 + j of rr) is void iff a restart lies among the k-1 bases after its first
        const uint32_t hasmask = ~(smear_up(rr, km1) >> 16) & ((1u << cnt) - 1u);
        const uint32_t fprev = revpairs32(prev), fcur = revpairs32(cur);
        const unsigned long long fwd64 = ((unsigned long long)fprev << 32) | fcur;          // first base highest
        const unsigned long long rev64 = ~(((unsigned long long)cur << 32) | prev);         // complemented, first base lowest
        const uint32_t rev_sh0 = 2u * (17u - k);                                             // + 2j per base; < 64 for every odd k <= 17

        KT r[SC_PER];
        bool ok[SC_PER];
        // The lane remembers its last three distinct k-mers.  A k-mer is emitted the first time it is seen;
        // seeing it again while remembered (tandem repeats of period 1-3: the contended buckets) only
        // bumps a counter, which goes to the workgroup's LDS table when the entry is evicted or the
        // thread's 16 bases end.  Either route counts each k-mer exactly once.  a1, a2, a3 are pairwise
        // distinct (an entry is only ever inserted on a miss; the initial ~0 is no k-mer), so at most one
        // compare hits.  The three counters share one register: n1 | n2 << 8 | n3 << 16.
        KT a1 = ~(KT)0, a2 = ~(KT)0, a3 = ~(KT)0;
        uint32_t nn = 0;
#pragma unroll
        for (int j = 0; j < SC_PER; j++) {
            KT f;
            if (sizeof(KT) == 4) f = (KT)__builtin_amdgcn_alignbit(fprev, fcur, 2u * (15u - j)) & mask;
            else f = (KT)(fwd64 >> (2u * (15u - j))) & mask;
            const KT rv = (KT)(rev64 >> (rev_sh0 + 2u * j)) & mask;
            const KT canon = f < rv ? f : rv;                                               // indexer.py:341
            const bool has = (hasmask >> j) & 1u;
            const bool e1 = canon == a1, e2 = canon == a2, e3 = canon == a3;
            const bool h1 = has & e1, h2 = has & e2, h3 = has & e3;
            const bool miss = has & !e1 & !e2 & !e3;
            nn += h1 ? 1u : (h2 ? 0x100u : (h3 ? 0x10000u : 0u));
            const uint32_t ev_n = miss ? (nn >> 16) : 0u;
            const KT ev_a = a3;
            a3 = miss ? a2 : a3;
            a2 = miss ? a1 : a2;
            a1 = miss ? canon : a1;
            nn = miss ? ((nn << 8) & 0xffff00u) : nn;
            r[j] = canon;
            ok[j] = (dbg & 8u) ? has : miss;
            if (!COUNT && ev_n != 0u) hot_insert(hot, (uint64_t)ev_a, ev_n, side, side_n, side_cap);   // rare: leaving a tandem run
        }
        // ---- hot keys leave the thread.  Tandem runs span many lanes: a lane whose predecessor (the 16 bases before)
        // ended with the same k-mers in its cache would emit them again -- once per lane instead of once per run.
        // Where the predecessor saw repeats at all, first occurrences it already holds become tallies too.
        {
            const uint32_t prev_nn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nn, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 gets 0
            if (__any(prev_nn != 0u)) {
                KT p1, p2, p3;
                if (sizeof(KT) == 4) {
                    p1 = (KT)__builtin_amdgcn_update_dpp(-1, (int)a1, 0x138, 0xf, 0xf, false);
                    p2 = (KT)__builtin_amdgcn_update_dpp(-1, (int)a2, 0x138, 0xf, 0xf, false);
                    p3 = (KT)__builtin_amdgcn_update_dpp(-1, (int)a3, 0x138, 0xf, 0xf, false);
                } else {
                    auto shr1 = [](unsigned long long v) {
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
                        return ((unsigned long long)hi << 32) | lo;
                    };
                    p1 = (KT)shr1((unsigned long long)a1); p2 = (KT)shr1((unsigned long long)a2); p3 = (KT)shr1((unsigned long long)a3);
                }
                if (prev_nn != 0u) {
#pragma unroll
                    for (int j = 0; j < SC_PER; j++) {
                        if (ok[j] && (r[j] == p1 || r[j] == p2 || r[j] == p3)) {
                            ok[j] = false;
                            if (!COUNT) hot_insert(hot, (uint64_t)r[j], 1u, side, side_n, side_cap);
                        }
                    }
                }
            }
            if (!COUNT && __any(nn != 0u)) {                                                  // this lane's own repeat tallies
                if (nn & 0xffu) hot_insert(hot, (uint64_t)a1, nn & 0xffu, side, side_n, side_cap);
                if (nn & 0xff00u) hot_insert(hot, (uint64_t)a2, (nn >> 8) & 0xffu, side, side_n, side_cap);
                if (nn >> 16) hot_insert(hot, (uint64_t)a3, nn >> 16, side, side_n, side_cap);
            }
        }
        if (COUNT) {
#pragma unroll
            for (int j = 0; j < SC_PER; j++)
                if (ok[j]) atomicAdd(&L.run[(uint32_t)((uint64_t)r[j] >> shift) & (B - 1u)], 1u);
            continue;
        }
        if (dbg & 4u) { uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < SC_PER; j++) x ^= ok[j] ? (uint32_t)r[j] : 0u;
            if (x == 0x12345u) flags[1] = 1; continue; }
        if (tally_fine && !(dbg & 1u)) {
#pragma unroll
            for (int j = 0; j < SC_PER; j++)
                if (ok[j]) atomicAdd(&fine[(uint32_t)((uint64_t)r[j] >> pl.fb_bits)], 1u);
        }
        scatter_tile<KT, WIDE>(L, r, ok, ~0u, shift, B, low_mask, out16, out, no_settle, cursor1, cap_end, dump, flags);
        if (hot.used >= HOT_SLOTS / 2) hot_flush(hot, side, side_n, side_cap);   // uniform: read after the barrier that ends the tile
    }
    if (COUNT) {
        __syncthreads();
        if (threadIdx.x < B && L.run[threadIdx.x]) atomicAdd(&sample_hist[threadIdx.x], L.run[threadIdx.x]);
        return;
    }
    hot_flush(hot, side, side_n, side_cap);
    if (tally_fine) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_fine; i += SC_T) fine_rows[(uint64_t)blockIdx.x * n_fine + i] = fine[i];
    }
}

// ------------------------------------------------------------------ bucket layout ---------------
__device__ __forceinline__ uint32_t block_excl_scan_512(uint32_t v, uint32_t *wsum, uint32_t &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();                                       // wsum may still be read from an earlier call
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t pre = 0;
    total = 0;
    for (int i = 0; i < 8; i++) { if (i < w) pre += wsum[i]; total += wsum[i]; }
    return pre + inc - v;
}

// one workgroup, one thread per level-1 bucket (B1 <= 512): room for every bucket from the sampled tallies
__global__ __launch_bounds__(512) void k_provision(const uint32_t *__restrict__ sample_hist, PartPlan pl, uint32_t n_sampled,
                                                   uint32_t stride, uint32_t capacity, uint32_t *__restrict__ bucket_base,
                                                   uint32_t *__restrict__ cursor1, uint32_t *__restrict__ cap_end, uint32_t *flags) {
    __shared__ uint32_t wsum[8];
    const uint32_t d = threadIdx.x;
    uint32_t room = 0;
    if (d < pl.B1) {
        const unsigned long long h = sample_hist[d];
        if (stride == 1) room = (uint32_t)h;                                                // exact
        else {
            const unsigned long long est = (h * pl.n_chunks + n_sampled - 1) / n_sampled;
            room = (uint32_t)(est + est / 8 + 4096);
        }
        room = (room + 3u) & ~3u;                                                           // bucket starts stay 16-byte aligned
    }
    uint32_t total;
    const uint32_t base = block_excl_scan_512(room, wsum, total);
    if (d < pl.B1) { bucket_base[d] = base; cursor1[d] = base; cap_end[d] = base + room; }
    if (d == 0) {
        bucket_base[pl.B1] = total;
        if (total > capacity) flags[0] = 1u;                                                // cannot happen with the bounds of part_workspace_bytes; be loud if it does
    }
}

// after the level-1 sort: how full every bucket got, the level-2 work split (bucket d gets ceil(n_d / R2)
// workgroups) and, for the layouts that are compact again from here on, the exclusive scan of the sizes
__global__ __launch_bounds__(512) void k_level1_finish(const uint32_t *__restrict__ cursor1, const uint32_t *__restrict__ bucket_base,
                                                       const uint32_t *__restrict__ cap_end, PartPlan pl, uint32_t *__restrict__ bucket_end,
                                                       uint32_t *__restrict__ compact_base, uint32_t *__restrict__ wg2_start,
                                                       const uint32_t *__restrict__ flags) {
    __shared__ uint32_t wsum[8];
    if (flags[0]) return;
    const uint32_t d = threadIdx.x, B1 = pl.B1;
    uint32_t size = 0;
    if (d < B1) size = min(cursor1[d], cap_end[d]) - bucket_base[d];
    const uint32_t g = (uint32_t)(((uint64_t)size + pl.R2 - 1) / pl.R2);
    uint32_t sum_n, sum_g;
    const uint32_t cb = block_excl_scan_512(size, wsum, sum_n);
    const uint32_t ws = block_excl_scan_512(g, wsum, sum_g);
    if (d < B1) { bucket_end[d] = bucket_base[d] + size; compact_base[d] = cb; wg2_start[d] = ws; }
    if (d == 0) { compact_base[B1] = sum_n; wg2_start[B1] = sum_g; }
}

// ------------------------------------------------------------------ launchers -------------------
static size_t fuse_lds(const PartPlan &pl, bool fine) {
    const size_t base = pl.k > 15 ? SCATTER_LDS_WIDE : SCATTER_LDS_NARROW;
    return base + (fine ? (size_t)pl.B1 * pl.B2 * 4 : 0);
}

void fuse_set_attributes() {
    hipFuncSetAttribute((const void *)k_walk_sort<uint32_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SCATTER_LDS_NARROW + 65536));
    hipFuncSetAttribute((const void *)k_walk_sort<uint32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_NARROW);
    hipFuncSetAttribute((const void *)k_walk_sort<uint64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_WIDE);
    hipFuncSetAttribute((const void *)k_walk_sort<uint64_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCATTER_LDS_WIDE);
}

// sampling launch + bucket layout
void launch_provision(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl,
                      uint32_t stride, uint32_t capacity, uint32_t *sample_hist, uint32_t *bucket_base, uint32_t *cursor1,
                      uint32_t *cap_end, uint32_t *flags, hipStream_t s) {
    const uint32_t n_sampled = (pl.n_chunks + stride - 1) / stride;
    const uint32_t grid = n_sampled < 2048u ? n_sampled : 2048u;
    hipMemsetAsync(sample_hist, 0, 512 * sizeof(uint32_t), s);
    if (pl.k <= 15)
        hipLaunchKernelGGL((k_walk_sort<uint32_t, true>), dim3(grid), dim3(SC_T), SCATTER_LDS_NARROW, s, codes, restarts, n_bases, st2, pl, n_sampled,
                           stride, (void *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, flags, (uint32_t *)nullptr, sample_hist,
                           (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint64_t)0, getenv("PK_DBG") ? (uint32_t)atoi(getenv("PK_DBG")) : 0u);
    else
        hipLaunchKernelGGL((k_walk_sort<uint64_t, true>), dim3(grid), dim3(SC_T), SCATTER_LDS_WIDE, s, codes, restarts, n_bases, st2, pl, n_sampled,
                           stride, (void *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, flags, (uint32_t *)nullptr, sample_hist,
                           (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint64_t)0, getenv("PK_DBG") ? (uint32_t)atoi(getenv("PK_DBG")) : 0u);
    hipLaunchKernelGGL(k_provision, dim3(1), dim3(512), 0, s, (const uint32_t *)sample_hist, pl, n_sampled, stride, capacity, bucket_base, cursor1,
                       cap_end, flags);
}

void launch_walk_sort(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl, void *out1,
                      uint32_t *cursor1, const uint32_t *cap_end, uint32_t dump, uint32_t *flags, uint32_t *fine_rows,
                      const uint32_t *bucket_base, uint32_t *bucket_end, uint32_t *compact_base, uint32_t *wg2_start,
                      unsigned long long *side, unsigned long long *side_n, uint64_t side_cap, hipStream_t s) {
    const size_t lds = fuse_lds(pl, fine_rows != nullptr);
    if (pl.k <= 15)
        hipLaunchKernelGGL((k_walk_sort<uint32_t, false>), dim3(pl.n_wg0), dim3(SC_T), lds, s, codes, restarts, n_bases, st2, pl, pl.n_chunks, 1u, out1,
                           cursor1, cap_end, dump, flags, fine_rows, (uint32_t *)nullptr, side, side_n, side_cap, getenv("PK_DBG") ? (uint32_t)atoi(getenv("PK_DBG")) : 0u);
    else
        hipLaunchKernelGGL((k_walk_sort<uint64_t, false>), dim3(pl.n_wg0), dim3(SC_T), lds, s, codes, restarts, n_bases, st2, pl, pl.n_chunks, 1u, out1,
                           cursor1, cap_end, dump, flags, fine_rows, (uint32_t *)nullptr, side, side_n, side_cap, 0u);
    hipLaunchKernelGGL(k_level1_finish, dim3(1), dim3(512), 0, s, (const uint32_t *)cursor1, bucket_base, cap_end, pl, bucket_end, compact_base,
                       wg2_start, (const uint32_t *)flags);
}

}  // namespace pk
