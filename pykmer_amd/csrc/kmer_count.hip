// kmer_count.hip -- the structure pass of the indexer for gfx950 and the table histogram.
//
// Structure pass: makes the reference's line-by-line parser (parse_fasta, indexer.py:45-99) data-parallel.
// Every 64-byte piece of text is summarised, the summaries are composed in prefix scans (see
// fasta_fsm.h), and every piece gets its exact parser state -- what the squeeze pass (kmer_pack.hip)
// starts from.  k_hist is Header.update_stats (tools.py:246-263) for tables that arrive from disk.
#include "fasta_fsm.h"
#include "pk_kernels.h"

#ifndef PK_LB_L2
#define PK_LB_L2 7   // waves per SIMD the structure kernel is compiled for (72 registers; 6: 0.425 ms on the genome and 0.73 on 400 k reads, 7: 0.412 and 0.66; 4, 5 and 8: 0.44)
#endif

namespace pk {

// ------------------------------------------------------------------ per-chunk summaries --------
// L1 summary of a whole chunk = (does it hold a line terminator, which line state does it end in when
// entered at a line start).  Only the chunk's last line matters: the state after its last terminator
// (or from its first byte, if it has none) is decided by the first non-blank byte that follows --
// '>' opens a header, anything else sequence text.  So one wave per chunk looks at the chunk's tail,
// 256 bytes at a time backwards to the last terminator, then forwards to the first non-blank; with
// ordinary line lengths that is one or two loads per lane instead of a pass over the 16 KiB.
__global__ __launch_bounds__(WG) void k_chunk_l1(const uint8_t *__restrict__ fasta, uint64_t n_bytes, uint32_t n_chunks,
                                                 L1 *__restrict__ chunk_l1) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;                                       // wave-uniform
    const uint64_t base = (uint64_t)chunk * CHUNK;
    const uint32_t nb = (uint32_t)(n_bytes - base < (uint64_t)CHUNK ? n_bytes - base : (uint64_t)CHUNK);
    // the lane's 4 bytes of window w (bytes past the chunk read as 0, which is neither blank nor terminator)
    auto window = [&](uint32_t w) -> uint32_t {
        const uint32_t off = w * 256u + lane * 4u;
        if (off + 4u <= nb) return *reinterpret_cast<const uint32_t *>(fasta + base + off);   // chunk bases are 16-byte aligned
        uint32_t v = 0;
        for (uint32_t j = 0; off + j < nb && j < 4u; j++) v |= (uint32_t)fasta[base + off + j] << (8u * j);
        return v;
    };
    const uint32_t n_win = (nb + 255u) / 256u;
    int last_term = -1;                                                  // byte index of the chunk's last terminator
    for (int w = (int)n_win - 1; w >= 0 && last_term < 0; w--) {
        const uint32_t v = window((uint32_t)w);
        uint32_t tm = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) tm |= is_term((v >> (8u * j)) & 0xffu) ? (1u << j) : 0u;
        const unsigned long long any = __ballot(tm != 0u);
        if (any) {
            const int hl = 63 - __builtin_clzll(any);
            const uint32_t tmh = (uint32_t)__builtin_amdgcn_readlane((int)tm, hl);
            last_term = w * 256 + hl * 4 + (31 - __builtin_clz(tmh));
        }
    }
    const uint32_t from = (uint32_t)(last_term + 1);
    uint32_t state = LS_START;
    bool found = false;
    for (uint32_t w = from / 256u; w < n_win && !found; w++) {
        const uint32_t v = window(w);
        uint32_t nz = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t idx = w * 256u + lane * 4u + j;
            nz |= (idx >= from && idx < nb && !is_ws((v >> (8u * j)) & 0xffu)) ? (1u << j) : 0u;
        }
        const unsigned long long any = __ballot(nz != 0u);
        if (any) {
            const int ll = __builtin_ctzll(any);
            const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)nz, ll);
            const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)v, ll);
            const uint32_t c = (word >> (8u * (uint32_t)__builtin_ctz(m))) & 0xffu;
            state = c == '>' ? (uint32_t)LS_HEADER : (uint32_t)LS_SEQ;
            found = true;
        }
    }
    if (lane == 0) chunk_l1[chunk] = nb ? l1_make(last_term >= 0, state) : 0u;
}

// Every piece is classified once (classify_piece, fasta_fsm.h): which bytes are bases, terminators, or something only the
// byte-wise machine understands; the bases' codes pushed together; their restart bits.  The L1 / L2 summaries of a piece
// of plain sequence text come from those masks, and the pack is handed to the squeeze pass, which does not read the text
// of such pieces again.  Pieces with header text whose lines begin plainly (header pieces) get their summaries from masks
// too (header_text).  What is left -- a blank or control byte outside header text, a line that begins with one, the partial
// last piece -- needs the byte-wise machines, which cost a wave the same for one lane as for 64: those pieces are queued
// and worked off 64 per wave pass.
__global__ __launch_bounds__(WG, PK_LB_L2) void k_chunk_l2(const uint8_t *__restrict__ fasta, uint64_t n_bytes,
                                                 const L1 *__restrict__ chunk_l1_state, L2 *__restrict__ chunk_l2,
                                                 LaneState *__restrict__ lane_state, PiecePack *__restrict__ packs,
                                                 uint32_t *__restrict__ chunk_odd, uint32_t km1) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[WG * LDS_STRIDE];
    __shared__ L1 sh1[WG / 64];
    __shared__ L2 sh2[WG / 64];
    __shared__ uint16_t queue[WG];
    __shared__ uint32_t n_queued[2], n_header_pieces;
    __shared__ uint32_t res1[WG];                          // queued pieces: L1 summary | dirty << 8
    __shared__ uint8_t ls_of[WG];                          // every piece's incoming line state (for the L2 pass over the queue)
    const uint64_t base = (uint64_t)blockIdx.x * CHUNK;
    if (threadIdx.x < 2) n_queued[threadIdx.x] = 0;
    if (threadIdx.x == 2) n_header_pieces = 0;
    stage_chunk(fasta, base, n_bytes, lds);
    __syncthreads();
    const uint32_t nb = piece_len(base, n_bytes);
    const uint8_t *mine = lds + threadIdx.x * LDS_STRIDE;
    PieceMasks pm;
    PiecePack pk;
    uint32_t cw[4];
    piece_scan(mine, nb, pm, cw);
    const bool full = nb == (uint32_t)PIECE;
    const bool plain = (pm.blank | pm.gt) == 0ull;                       // nothing but sequence characters and terminators
    {
        const unsigned long long S = ~pm.term;                           // the pack of the piece read as plain sequence text
        piece_compact(pm.valid, S & ~pm.valid, cw, plain, (uint32_t)__popcll(S), (S & 1ull) != 0ull, pk);
        uint4 *dst = reinterpret_cast<uint4 *>(packs + (uint64_t)blockIdx.x * WG + threadIdx.x);
        dst[0] = make_uint4((uint32_t)pk.c_lo, (uint32_t)(pk.c_lo >> 32), (uint32_t)pk.c_hi, (uint32_t)(pk.c_hi >> 32));
        dst[1] = make_uint4((uint32_t)pk.restart, (uint32_t)(pk.restart >> 32), pk.meta, 0u);
    }
    // ---- L1: line state.  Full pieces whose lines do not begin with a blank from the masks; the rest (and partial pieces)
    // go to the queue
    const bool masks1 = full && lines_start_plain(pm);
    L1 my1 = l1_of_piece(pm);                                            // not used if queued
    bool dirty = !plain;                                                 // for queued pieces: what the byte-wise walk says
    const bool q1 = !masks1;
    if (q1) queue[atomicAdd(&n_queued[0], 1u)] = (uint16_t)threadIdx.x;
    __syncthreads();
    for (uint32_t q0 = (threadIdx.x >> 6) * 64u; q0 < n_queued[0]; q0 += WG) {       // wave-uniform
        const uint32_t qi = q0 + (threadIdx.x & 63u);
        const bool work = qi < n_queued[0];
        const uint32_t pc = work ? queue[qi] : 0u;
        bool d;
        const L1 r = piece_l1_at(lds + pc * LDS_STRIDE, work ? piece_len_of(pc, base, n_bytes) : 0u, d);
        if (work) res1[pc] = r | (d ? 0x100u : 0u);
    }
    __syncthreads();
    if (q1) { my1 = res1[threadIdx.x] & 0xffu; dirty = (res1[threadIdx.x] >> 8) & 1u; }
    L1 tot1;
    const L1 st1 = wg_excl_scan_l1(my1, chunk_l1_state[blockIdx.x], sh1, &tot1);
    const uint32_t ls_in = l1_kind(st1);
    ls_of[threadIdx.x] = (uint8_t)ls_in;
    // ---- L2: record / run state.  Clean pieces and header pieces from their masks; the rest through the queue again
    const bool clean = !dirty && ls_in != LS_HEADER;
    // a header piece: header lines by masks (header_text), and no blank or control byte outside header text
    bool header_piece = false;
    unsigned long long H = 0;
    if (__any(!clean && masks1)) {
        H = header_text(pm, ls_in);
        header_piece = !clean && masks1 && first_byte_plain(pm, ls_in) && (pm.blank & ~H) == 0ull;
    }
    L2 my2 = l2_identity();
    if (clean) my2 = l2_of_clean_piece(pm, pk, nb, ls_in, km1);
    if (__any(header_piece)) {                                           // compaction again, without the header text
        const unsigned long long seq = ~pm.term & ~H, valid = pm.valid & seq;
        PiecePack hk;
        piece_compact(header_piece ? valid : 0ull, header_piece ? ((seq & ~valid) | H) : 0ull, cw, true, 0u, false, hk);
        if (header_piece)
            my2 = l2_of_masks(pm.term, valid, (seq & ~valid) | H, seq, (uint32_t)__popcll(header_starts(pm, ls_in)), hk, nb, ls_in, km1);
    }
    const bool q2 = !clean && !header_piece;
    if (q2) queue[atomicAdd(&n_queued[1], 1u)] = (uint16_t)threadIdx.x;   // the L1 queue was consumed before the scan's barriers
    const unsigned long long hp_wave = __ballot(header_piece);
    if ((threadIdx.x & 63u) == 0u && hp_wave) atomicAdd(&n_header_pieces, (uint32_t)__popcll(hp_wave));
    __syncthreads();
    for (uint32_t q0 = (threadIdx.x >> 6) * 64u; q0 < n_queued[1]; q0 += WG) {
        const uint32_t qi = q0 + (threadIdx.x & 63u);
        const bool work = qi < n_queued[1];
        const uint32_t pc = work ? queue[qi] : 0u;
        const L2 r = piece_l2_at(lds + pc * LDS_STRIDE, work ? piece_len_of(pc, base, n_bytes) : 0u, ls_of[pc], km1);
        if (work) {                                        // the result takes the place of the piece's own text, now used up
            uint32_t *out = reinterpret_cast<uint32_t *>(lds + pc * LDS_STRIDE);
            out[0] = r.flags; out[1] = r.bits; out[2] = r.rec; out[3] = (uint32_t)r.p_tail;     // within one piece: <= 64
        }
    }
    __syncthreads();
    if (q2) {
        const uint32_t *in = reinterpret_cast<const uint32_t *>(mine);
        my2.flags = in[0]; my2.bits = in[1]; my2.rec = in[2]; my2.p_tail = in[3];
    }
    L2 total;
    const L2 rel = wg_excl_scan_l2(my2, l2_identity(), sh2, &total, km1);     // prefix relative to the chunk start
    lane_state[(uint64_t)blockIdx.x * WG + threadIdx.x] = lane_state_pack(rel, ls_in, q2, header_piece);
    if (threadIdx.x == 0) {
        chunk_l2[blockIdx.x] = total;
        chunk_odd[blockIdx.x] = n_queued[1] + n_header_pieces;   // pieces whose text the squeeze pass has to see
    }
}

// ------------------------------------------------------------------ grid-level scans -----------
// Three tiny launches per level instead of one latency-bound workgroup: (1) every 1024-summary tile is
// reduced by its own workgroup, (2) one workgroup scans the <= a few hundred tile totals behind the
// carried stream state, (3) every tile is scanned again behind its seed and written out.
constexpr int SCAN_T = 1024;

template <typename S, class Compose, class Shfl>
__device__ __forceinline__ S tile_scan(const S &mine, const S &seed, S *sh, S *total, Compose compose, Shfl shfl_up1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    S inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        S o = shfl_up1(inc, d);
        if (lane >= d) inc = compose(o, inc);
    }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    S pre = seed, tot = seed;
    for (int j = 0; j < SCAN_T / 64; j++) { if (j == w) pre = tot; tot = compose(tot, sh[j]); }
    *total = tot;
    S up = shfl_up1(inc, 1);
    return lane == 0 ? pre : compose(pre, up);
}

__global__ __launch_bounds__(SCAN_T) void k_scan_l1_reduce(const L1 *__restrict__ in, uint32_t n, L1 *__restrict__ tile_tot) {
    __shared__ L1 sh[SCAN_T / 64];
    const uint32_t i = blockIdx.x * SCAN_T + threadIdx.x;
    L1 tot;
    tile_scan<L1>(i < n ? in[i] : 0u, 0u, sh, &tot, [](L1 a, L1 b) { return l1_compose(a, b); }, [](L1 v, int d) { return (L1)__shfl_up(v, d, 64); });
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}
// (also zeroes `n_zero` words at `zero_words`: the side-list length and the flags of the feed's partition passes -- one
// stream operation less per feed; everything that writes or reads them is launched behind this kernel)
__global__ __launch_bounds__(SCAN_T) void k_scan_l1_tiles(L1 *__restrict__ tile_tot, uint32_t n_tiles, Carry *carry, uint32_t *__restrict__ zero_words,
                                                          uint32_t n_zero) {
    __shared__ L1 sh[SCAN_T / 64];
    if (threadIdx.x < n_zero) zero_words[threadIdx.x] = 0u;
    L1 run = carry->l1;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += SCAN_T) {
        const uint32_t i = t0 + threadIdx.x;
        L1 tot;
        L1 ex = tile_scan<L1>(i < n_tiles ? tile_tot[i] : 0u, run, sh, &tot, [](L1 a, L1 b) { return l1_compose(a, b); },
                              [](L1 v, int d) { return (L1)__shfl_up(v, d, 64); });
        if (i < n_tiles) tile_tot[i] = ex;                 // now: the state before the tile
        run = tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) carry->l1 = run;
}
__global__ __launch_bounds__(SCAN_T) void k_scan_l1_apply(const L1 *__restrict__ in, uint32_t n, const L1 *__restrict__ tile_seed,
                                                          L1 *__restrict__ out_state) {
    __shared__ L1 sh[SCAN_T / 64];
    const uint32_t i = blockIdx.x * SCAN_T + threadIdx.x;
    L1 tot;
    L1 ex = tile_scan<L1>(i < n ? in[i] : 0u, tile_seed[blockIdx.x], sh, &tot, [](L1 a, L1 b) { return l1_compose(a, b); },
                          [](L1 v, int d) { return (L1)__shfl_up(v, d, 64); });
    if (i < n) out_state[i] = ex;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_l2_reduce(const L2 *__restrict__ in, uint32_t n, L2 *__restrict__ tile_tot, uint32_t km1) {
    __shared__ L2 sh[SCAN_T / 64];
    const uint32_t i = blockIdx.x * SCAN_T + threadIdx.x;
    L2 tot;
    tile_scan<L2>(i < n ? in[i] : l2_identity(), l2_identity(), sh, &tot, [km1](const L2 &a, const L2 &b) { return l2_compose(a, b, km1); },
                  [](const L2 &v, int d) { return shfl_up_l2(v, d); });
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_T) void k_scan_l2_tiles(L2 *__restrict__ tile_tot, uint32_t n_tiles, Carry *carry, uint32_t km1) {
    __shared__ L2 sh[SCAN_T / 64];
    L2 run = carry->l2;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += SCAN_T) {
        const uint32_t i = t0 + threadIdx.x;
        L2 tot;
        L2 ex = tile_scan<L2>(i < n_tiles ? tile_tot[i] : l2_identity(), run, sh, &tot,
                              [km1](const L2 &a, const L2 &b) { return l2_compose(a, b, km1); }, [](const L2 &v, int d) { return shfl_up_l2(v, d); });
        if (i < n_tiles) tile_tot[i] = ex;
        run = tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { carry->l2 = run; carry->n_recs = run.rec; }
}
__global__ __launch_bounds__(SCAN_T) void k_scan_l2_apply(const L2 *__restrict__ in, uint32_t n, const L2 *__restrict__ tile_seed,
                                                          L2 *__restrict__ out_state, uint32_t km1) {
    __shared__ L2 sh[SCAN_T / 64];
    const uint32_t i = blockIdx.x * SCAN_T + threadIdx.x;
    L2 tot;
    L2 ex = tile_scan<L2>(i < n ? in[i] : l2_identity(), tile_seed[blockIdx.x], sh, &tot,
                          [km1](const L2 &a, const L2 &b) { return l2_compose(a, b, km1); }, [](const L2 &v, int d) { return shfl_up_l2(v, d); });
    if (i < n) out_state[i] = ex;
}

// ------------------------------------------------------------------ histogram ------------------
// Histogram of an existing u8 table (Header.update_stats, tools.py:246-263).  Zero and one -- by far the
// commonest values of a k-mer table -- are counted in registers; the rest go through a per-wave LDS
// histogram so hot bins do not serialise on one LDS address per workgroup.
__global__ __launch_bounds__(WG) void k_hist(const uint8_t *__restrict__ src, uint64_t n, unsigned long long *__restrict__ hist /*[256]*/) {
    __shared__ uint32_t h[WG / 64][256];
    for (int i = threadIdx.x; i < (WG / 64) * 256; i += WG) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *myh = h[threadIdx.x >> 6];
    uint64_t ones = 0;
    const uint64_t n16 = n / 16;
    for (uint64_t g = (uint64_t)blockIdx.x * WG + threadIdx.x; g < n16; g += (uint64_t)gridDim.x * WG) {
        const uint4 a = reinterpret_cast<const uint4 *>(src)[g];
        const uint32_t t[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t v = (t[j >> 2] >> (8 * (j & 3))) & 0xffu;
            if (v == 1u) ones++;
            else if (v) atomicAdd(&myh[v], 1u);
        }
    }
    // tail (n not a multiple of 16): first workgroup, one element per lane
    if (blockIdx.x == 0) {
        for (uint64_t i = n16 * 16 + threadIdx.x; i < n; i += WG) {
            const uint32_t x = src[i];
            if (x == 1u) ones++;
            else if (x) atomicAdd(&myh[x], 1u);
        }
    }
    for (int d = 32; d; d >>= 1) ones += __shfl_down((unsigned long long)ones, d, 64);
    if ((threadIdx.x & 63) == 0 && ones) atomicAdd(&hist[1], (unsigned long long)ones);
    __syncthreads();
    for (int b = threadIdx.x; b < 256; b += WG) {
        unsigned long long s = 0;
        for (int w = 0; w < WG / 64; w++) s += h[w][b];
        if (s) atomicAdd(&hist[b], s);
    }
}

// ------------------------------------------------------------------ launchers ------------------
void launch_chunk_l1(const uint8_t *fasta, uint64_t n, L1 *chunk_l1, uint32_t n_chunks, hipStream_t s) {
    hipLaunchKernelGGL(k_chunk_l1, dim3((n_chunks + WG / 64 - 1) / (WG / 64)), dim3(WG), 0, s, fasta, n, n_chunks, chunk_l1);
}
// tile_ws: scratch for ceil(n_chunks / 1024) summaries of the respective type
void launch_scan_l1(const L1 *in, uint32_t n_chunks, Carry *carry, L1 *out, L1 *tile_ws, uint32_t *zero_words, uint32_t n_zero, hipStream_t s) {
    const uint32_t n_tiles = (n_chunks + SCAN_T - 1) / SCAN_T;
    hipLaunchKernelGGL(k_scan_l1_reduce, dim3(n_tiles), dim3(SCAN_T), 0, s, in, n_chunks, tile_ws);
    hipLaunchKernelGGL(k_scan_l1_tiles, dim3(1), dim3(SCAN_T), 0, s, tile_ws, n_tiles, carry, zero_words, n_zero);
    hipLaunchKernelGGL(k_scan_l1_apply, dim3(n_tiles), dim3(SCAN_T), 0, s, in, n_chunks, (const L1 *)tile_ws, out);
}
void launch_chunk_l2(const uint8_t *fasta, uint64_t n, const L1 *st1, L2 *chunk_l2, LaneState *lane_state, PiecePack *packs, uint32_t *chunk_odd, uint32_t n_chunks,
                     uint32_t k, hipStream_t s) {
    hipLaunchKernelGGL(k_chunk_l2, dim3(n_chunks), dim3(WG), 0, s, fasta, n, st1, chunk_l2, lane_state, packs, chunk_odd, k - 1);
}
void launch_scan_l2(const L2 *in, uint32_t n_chunks, Carry *carry, L2 *out, L2 *tile_ws, uint32_t k, hipStream_t s) {
    const uint32_t n_tiles = (n_chunks + SCAN_T - 1) / SCAN_T;
    hipLaunchKernelGGL(k_scan_l2_reduce, dim3(n_tiles), dim3(SCAN_T), 0, s, in, n_chunks, tile_ws, k - 1);
    hipLaunchKernelGGL(k_scan_l2_tiles, dim3(1), dim3(SCAN_T), 0, s, tile_ws, n_tiles, carry, k - 1);
    hipLaunchKernelGGL(k_scan_l2_apply, dim3(n_tiles), dim3(SCAN_T), 0, s, in, n_chunks, (const L2 *)tile_ws, out, k - 1);
}
static uint32_t stream_grid(uint64_t items_per_thread_units) {
    uint64_t g = (items_per_thread_units + WG - 1) / WG;
    if (g > 256u * 8u) g = 256u * 8u;
    if (g == 0) g = 1;
    return (uint32_t)g;
}
void launch_hist8(const uint8_t *table8, uint64_t n, unsigned long long *hist, hipStream_t s) {
    hipLaunchKernelGGL(k_hist, dim3(stream_grid(n / 16 + 1)), dim3(WG), 0, s, table8, n, hist);
}

}  // namespace pk
