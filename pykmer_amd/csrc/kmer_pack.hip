// kmer_pack.hip -- the squeeze pass: FASTA text -> packed stream of valid bases.
//
// The reference turns every record into a tuple of codes (0-3, None) and slides a k-window over it
// (parse_fasta indexer.py:45-99, gen_kmers :130-160).  Here the text is read once more after the
// structure pass (kmer_count.hip) has given every 64-byte piece its exact parser state, and what is kept
// of it is just what the k-mer windows need:
//
//   codes    2 bits per VALID base of a record (A 0, C 1, G 2, T 3; CONV, indexer.py:36-41), 16 per dword
//            from bit 0 up, in text order.  Line terminators, headers, blanks and characters that map to
//            None are gone;
//   restart  1 bit per base: the run of valid bases starts anew here (first base of a record, or
//            something that maps to None lay between it and the base before: indexer.py:144 voids every
//            window that would span it).
//
// Each 16 KiB chunk of text owns a fixed slot (4 KiB of codes, 2 KiB of restart bits, a base count), so no
// stream-wide compaction is needed: the k-1 bases in front of a slot come from the chunk's start state.
// Per-record tallies (seq_len, number of valid windows, header extent: indexer.py:75-95,349-351) are
// taken here too, where the text is in hand.  0.38 bytes written per base instead of one 4-byte record;
// kmer_fuse.hip assembles the k-mers from the slots with no dependence between its threads.
#include <cstdlib>
#include "fasta_fsm.h"
#include "kmer_walk.h"
#include "pk_kernels.h"

#ifndef PK_LB_SQ
#define PK_LB_SQ 4   // waves per SIMD the squeeze kernel is compiled for (3, 5, 6 and 8 all give 0.31 ms instead of 0.21)
#endif

namespace pk {

// exclusive scan of a small count over the 256 lanes of the workgroup (sh: 4 words; one barrier)
__device__ __forceinline__ uint32_t wg_excl_scan_u32(uint32_t v, uint32_t *sh, uint32_t &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int i = 0; i < w; i++) pre += sh[i];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    return pre + inc - v;
}

// OR `n_bits` bits of (lo, hi) into the LDS bit array `dst` at bit offset `at`
__device__ __forceinline__ void lds_or_bits(uint32_t *dst, uint32_t at, unsigned long long lo, unsigned long long hi, uint32_t n_bits) {
    if (n_bits == 0) return;
    const uint32_t w0 = at >> 5, sh = at & 31u;
    const uint32_t x[5] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), 0u};
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint32_t v = sh ? ((x[i] << sh) | carry) : x[i];
        carry = sh ? (x[i] >> (32u - sh)) : 0u;
        if (v) atomicOr(&dst[w0 + i], v);                    // words past the lane's bits are all zero and skipped
    }
}

// The valid windows that END at each of a piece's nv pushed-together bases (bit j: one ends at base j), given the bases'
// restart bits as the piece alone knows them (F; base 0's is completed here: the run was already broken when the piece
// began) and the length of the run carried in: no restart among the k-1 positions behind the window's first base, and --
// where no restart precedes -- enough bases carried in (indexer.py:144).
__device__ __forceinline__ unsigned long long window_ends(unsigned long long &F, uint32_t nv, uint32_t run, uint32_t km1) {
    if (run == 0u && nv) F |= 1ull;
    const unsigned long long keep = nv >= 64u ? ~0ull : ((1ull << nv) - 1ull);
    unsigned long long X = 0;
    {
        const unsigned long long y1 = F | (F << 1), y2 = y1 | (y1 << 2), y3 = y2 | (y2 << 4), y4 = y3 | (y3 << 8);
        uint32_t off = 0;                                                 // k - 1 <= 20 copies: 16 + 4 at most
        if (km1 & 16u) { X |= y4; off = 16; }
        if (km1 & 8u) { X |= y3 << off; off += 8; }
        if (km1 & 4u) { X |= y2 << off; off += 4; }
        if (km1 & 2u) { X |= y1 << off; off += 2; }
        if (km1 & 1u) { X |= F << off; }
    }
    const uint32_t short_by = run >= km1 ? 0u : km1 - run;                // leading positions the carried run cannot complete
    const unsigned long long lead = short_by >= 64u ? ~0ull : ((1ull << short_by) - 1ull);
    // a restart inside the piece takes over from the carried run: positions at or above the first restart obey X only
    const unsigned long long below_first = F ? ((F & (0ull - F)) - 1ull) : ~0ull;
    return ~X & ~(lead & below_first) & keep;
}

// ------------------------------------------------------------------ clean pieces ----
// A CLEAN piece holds nothing but sequence characters and line terminators and does not start inside a header line
// (the structure pass flags everything else): no record opens, nothing is stripped.  The structure pass has already
// classified its bytes and pushed its bases together (PiecePack, fasta_fsm.h); what is added here is what depends on
// the state the piece is entered with:
//   restart bit of base 0   the run was already broken when the piece began;
//   window count            positions with no restart among the k-1 before them (shift-or smear), minus those the
//                           incoming run is too short for (indexer.py:144);
//   tallies                 sequence characters (valid or not, indexer.py:77), pending blanks that turn out interior.
// `mine`: this lane's piece is a clean one; other lanes run along and leave no trace.
__device__ __forceinline__ void squeeze_apply(const PiecePack &pk, SeqWalker &wk, PieceBases &pb, bool mine) {
    const uint32_t nv = pack_n_valid(pk);
    if (mine) wk.seq_acc += pack_n_seq(pk);                               // indexer.py:77: valid or not
    // blanks pending from the piece before (that piece was not a clean one; this one is): interior if sequence text
    // follows -- each maps to None and the run breaks -- and stripped if the line ends here (indexer.py:56)
    if (mine && wk.pend) {
        if (pack_first_is_seq(pk)) { wk.seq_acc += wk.pend; wk.run = 0u; }
        wk.pend = 0;
    }
    const bool live = mine && wk.rec != 0;                                // text before the first header is dropped
    unsigned long long F = pk.restart;
    const unsigned long long has = window_ends(F, nv, wk.run, wk.k - 1u);
    wk.kmer_acc += live ? (uint64_t)__popcll(has) : 0ull;
    pb.code_lo = live ? pk.c_lo : 0ull; pb.code_hi = live ? pk.c_hi : 0ull;
    pb.restart = live ? F : 0ull;
    pb.n = live ? nv : 0u;
}

// ------------------------------------------------------------------ header pieces ----
// A HEADER PIECE (the structure pass's term, k_chunk_l2): a full piece that holds header text -- it opens records, or is
// entered inside a header line -- whose lines do not begin with a blank and that has no blank or control byte outside the
// header text.  Its header lines are found by masks (header_text, fasta_fsm.h): everything that is neither a terminator
// nor header text is a sequence character.  A read set has such a piece every kilobase; walking them byte by byte (64
// steps of ~100 instructions for the wave, whatever the number of lanes at work) was 86 % of this kernel's time there.
//
// Index one past the last byte of the header line `run` that str.strip() keeps (indexer.py:56,66), 0 if there is none:
// bytes from 0x21 up are kept; blanks and control bytes above the last of those are looked at one by one, from the top
// (a header rarely ends in one: zero rounds).
__device__ __forceinline__ uint32_t header_text_end(const uint8_t *text, unsigned long long run, unsigned long long blank) {
    const unsigned long long solid = run & ~blank;
    uint32_t end = solid ? 64u - (uint32_t)__builtin_clzll(solid) : 0u;
    unsigned long long todo = run & blank & (end >= 64u ? 0ull : ~0ull << end);
    while (todo) {
        const uint32_t p = 63u - (uint32_t)__builtin_clzll(todo);
        if (!is_ws(text[p])) { end = p + 1u; break; }
        todo ^= 1ull << p;
    }
    return end;
}

// All lanes of the wave; `mine`: this lane holds a header piece of a live record (wq.rec != 0) in `text`, with wq at the
// exact state of its first byte.  Leaves the piece's bases in rb and the tallies of the piece's LAST record in wq (the
// caller flushes them); the records that end inside the piece are flushed here.
__device__ __forceinline__ void squeeze_header_piece(const uint8_t *text, SeqWalker &wq, PieceBases &rb, bool mine) {
    PieceMasks pm;
    uint32_t cw[4];
    piece_scan(text, (uint32_t)PIECE, pm, cw);
    const uint32_t ls_in = wq.ls;
    const unsigned long long H = header_text(pm, ls_in), S = header_starts(pm, ls_in);
    const unsigned long long seq = ~pm.term & ~H, valid = pm.valid & seq;
    PiecePack hk;
    piece_compact(mine ? valid : 0ull, mine ? ((seq & ~valid) | H) : 0ull, cw, true, 0u, false, hk);     // loops as a wave
    if (!mine) return;
    // blanks pending from the piece before: interior if a sequence character comes first (indexer.py:56; step())
    if (wq.pend && ls_in == LS_SEQ && (seq & 1ull)) { wq.seq_acc += wq.pend; wq.run = 0u; }
    wq.pend = 0;
    unsigned long long F = hk.restart;
    const uint32_t nv = pack_n_valid(hk);
    const unsigned long long has = window_ends(F, nv, wq.run, wq.k - 1u);
    if (ls_in == LS_HEADER) {                                             // the line the piece is entered in goes on
        const uint32_t e = header_text_end(text, H & ~(H + 1ull), pm.blank);
        if (e) wq.name_end = wq.pos0 + e;
    }
    // one round per record opened: what lies below its '>' belongs to the record before
    unsigned long long below_prev = 0, cmp_prev = 0, todo = S;
    while (todo) {
        const uint32_t s = (uint32_t)__builtin_ctzll(todo);
        const unsigned long long bit = 1ull << s, below = bit - 1ull;
        const unsigned long long cmp_below = (1ull << (uint32_t)__popcll(valid & below)) - 1ull;     // <= 63 bases below a byte
        wq.seq_acc += (uint64_t)__popcll(seq & below & ~below_prev);
        wq.kmer_acc += (uint64_t)__popcll(has & cmp_below & ~cmp_prev);
        wq.flush_rec();
        wq.rec++;                                                         // indexer.py:66-82
        if (wq.rec <= wq.recs_cap) wq.recs[wq.rec - 1].name_off = wq.pos0 + s + 1u;
        wq.name_end = wq.pos0 + header_text_end(text, H & ~(H + bit), pm.blank);      // at least the '>' itself
        below_prev = below; cmp_prev = cmp_below;
        todo &= todo - 1ull;
    }
    wq.seq_acc += (uint64_t)__popcll(seq & ~below_prev);
    wq.kmer_acc += (uint64_t)__popcll(has & ~cmp_prev);
    rb.code_lo = hk.c_lo; rb.code_hi = hk.c_hi; rb.restart = F; rb.n = nv;
}

// ---- staging: the chunk's 16 KiB as a plain image in LDS (the lane's piece = 64 contiguous bytes; the 4-way bank
// conflict on its four 16-byte reads is noise here).  Full chunks arrive by LDS-DMA (global_load_lds: no registers,
// asynchronous -- the NEXT chunk's image is requested before the current one is squeezed and lands meanwhile); a chunk
// that reaches the end of the stream goes through registers so that bytes past the end are never read and arrive as 0.
__device__ __forceinline__ void stage_image_async(const uint8_t *__restrict__ fasta, uint64_t chunk_base, uint8_t *buf) {
    // the lane index is made opaque here: as loop invariants of the chunk loop the four lane addresses lived in registers,
    // were spilled, and each re-load from scratch waited for the DMA issued just before it (a memory round trip apiece)
    uint32_t lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < CHUNK / (WG * 16); i++) {
        const uint32_t p = i * WG + lane;                                    // 16-byte piece of the chunk; one wave-instruction = 1 KiB
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(fasta + chunk_base + p * 16u),
                                         (__attribute__((address_space(3))) void *)(buf + (p & ~63u) * 16u), 16, 0, 0);
    }
}
__device__ __forceinline__ void stage_image_tail(const uint8_t *__restrict__ fasta, uint64_t chunk_base, uint64_t n_bytes, uint8_t *buf) {
#pragma unroll
    for (int i = 0; i < CHUNK / (WG * 16); i++) {
        const uint32_t p = i * WG + threadIdx.x;
        const uint64_t g = chunk_base + (uint64_t)p * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16u <= n_bytes) {
            v = *reinterpret_cast<const uint4 *>(fasta + g);
        } else if (g < n_bytes) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < (uint32_t)(n_bytes - g); j++) w[j >> 2] |= (uint32_t)fasta[g + j] << (8u * (j & 3u));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(buf + p * 16u) = v;
    }
}
__device__ __forceinline__ void stage_image(const uint8_t *__restrict__ fasta, uint64_t chunk_base, uint64_t n_bytes, uint8_t *buf) {
    if (chunk_base + CHUNK <= n_bytes) stage_image_async(fasta, chunk_base, buf);     // uniform
    else stage_image_tail(fasta, chunk_base, n_bytes, buf);
}

template <uint32_t KC>                   // k as a literal (0: the argument)
__global__ __launch_bounds__(WG, PK_LB_SQ) void k_squeeze(const uint8_t *__restrict__ fasta, uint64_t n_bytes, uint64_t stream_off,
                                                const LaneState *__restrict__ lane_state, const PiecePack *__restrict__ packs,
                                                const L2 *__restrict__ chunk_l2_state,
                                                const uint32_t *__restrict__ chunk_odd, uint32_t k_arg, uint32_t n_chunks, uint32_t chunks_per_wg,
                                                uint32_t *__restrict__ codes,
                                                uint32_t *__restrict__ restarts, uint32_t *__restrict__ n_bases,
                                                DevRec *__restrict__ recs, uint64_t recs_cap, Carry *carry, uint32_t *__restrict__ flags) {
    // The record array was sized before this feed's records were counted (no host round trip between the structure pass and
    // this kernel).  The count is known here -- the structure pass's scan left it in `carry` -- so if the array is too
    // small every workgroup returns before touching anything, flags[0] = 2 keeps the later kernels of the feed away, and
    // the host grows the array and repeats from here.
    if (carry->n_recs > recs_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 2u;
        return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t image[2][CHUNK];   // text of chunks that hold queued pieces (header pieces, pieces for the byte-wise machine)
    __shared__ __attribute__((aligned(16))) uint32_t slot_codes[SLOT_CODE_WORDS + 8];   // + slack: lds_or_bits touches up to 5 words
    __shared__ __attribute__((aligned(16))) uint32_t slot_rst[SLOT_RST_WORDS + 8];
    __shared__ uint32_t scan_sh[WG / 64];
    __shared__ uint16_t queue[WG];                         // pieces that need their text
    __shared__ uint32_t n_queued;
    __shared__ RecAcc racc;
    const uint32_t k = KC ? KC : k_arg, km1 = k - 1;
    if (threadIdx.x == 0) n_queued = 0;
    for (uint32_t i = threadIdx.x; i < SLOT_CODE_WORDS + 8; i += WG) slot_codes[i] = 0;
    for (uint32_t i = threadIdx.x; i < SLOT_RST_WORDS + 8; i += WG) slot_rst[i] = 0;
    recacc_init(racc);
    SeqWalker wk;
    wk.setup(k, recs, recs_cap, &racc);
    const uint32_t c_lo = blockIdx.x * chunks_per_wg, c_hi = min(c_lo + chunks_per_wg, n_chunks);
    // Per chunk a lane needs its state (8 bytes) and its pack (32 bytes).  Both are requested one chunk AHEAD and taken
    // delivery of right before the current chunk's slot is stored (settle, as in the sort kernels: loads and stores
    // share one in-order counter, and a load waited for after the stores would also wait for the stores).  The text
    // itself is only fetched -- by LDS-DMA, also one chunk ahead -- for chunks with queued pieces.
    struct Fetched { uint32_t ls_flags, ls_rec_tail; uint4 p0, p1; };
    auto fetch = [&](uint32_t c, Fetched &f) {
        const LaneState l = lane_state[(uint64_t)c * WG + threadIdx.x];
        f.ls_flags = l.flags; f.ls_rec_tail = l.rec_tail;
        const uint4 *src = reinterpret_cast<const uint4 *>(packs + (uint64_t)c * WG + threadIdx.x);
        f.p0 = src[0]; f.p1 = src[1];
    };
    Fetched nxt;
    auto settle = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): also the next chunk's image, if one was requested
        asm volatile("" : "+v"(nxt.ls_flags), "+v"(nxt.ls_rec_tail), "+v"(nxt.p0.x), "+v"(nxt.p0.y), "+v"(nxt.p0.z), "+v"(nxt.p0.w),
                          "+v"(nxt.p1.x), "+v"(nxt.p1.y), "+v"(nxt.p1.z), "+v"(nxt.p1.w));
    };
    if (c_lo < c_hi) {
        if (chunk_odd[c_lo]) stage_image(fasta, (uint64_t)c_lo * CHUNK, n_bytes, image[0]);
        fetch(c_lo, nxt);
    }
    settle();
    __syncthreads();
#ifdef PK_PHASE_PROF      // where a workgroup's time goes, in cycles of its thread 0 (experiment builds; printed by pk_api.hip)
    unsigned long long sq_prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sq_t = __builtin_readcyclecounter();
#define SQ_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); sq_prof[i] += n_ - sq_t; sq_t = n_; } } while (0)
#else
#define SQ_MARK(i) do { } while (0)
#endif
    for (uint32_t c = c_lo; c < c_hi; c++) {
        const uint64_t base = (uint64_t)c * CHUNK;
        uint8_t *buf = image[(c - c_lo) & 1u];
        const bool all_clean = chunk_odd[c] == 0u;         // uniform over the workgroup; the usual chunk
        if (!all_clean) __syncthreads();                   // every wave's share of the image has landed (each waited for its own in settle)
        // the record tallies of a chunk count up from the record it starts in; when that moves on, what was gathered
        // is written out (a genome keeps one window for thousands of chunks, a read set moves it with every chunk)
        const uint32_t first_rec = chunk_l2_state[c].rec;
        if (first_rec != racc.rec0) {                      // uniform: rec0 has not changed since the barrier
            recacc_spill(racc, racc.rec0, recs, recs_cap);
            __syncthreads();
            if (threadIdx.x == 0) racc.rec0 = first_rec;
            __syncthreads();
        }
        SQ_MARK(0);                                        // image barrier + record-window move
        const Fetched me = nxt;
        if (c + 1 < c_hi) {
            // the other image is free: its readers passed the barriers of the chunk before this one
            if (chunk_odd[c + 1]) stage_image(fasta, base + CHUNK, n_bytes, image[(c - c_lo + 1) & 1u]);
            fetch(c + 1, nxt);
        }
        // exact parser state at this lane's first byte: chunk state . lane prefix (both from the structure pass)
        const L2 chunk_st = chunk_l2_state[c];
        LaneState lst; lst.flags = me.ls_flags; lst.rec_tail = me.ls_rec_tail;
        const L2 st2 = l2_compose(chunk_st, lane_state_l2(lst), km1);
        const uint32_t ls_in = lane_state_ls(lst);
        wk.begin(ls_in, st2, stream_off + base + (uint64_t)threadIdx.x * PIECE);
        uint8_t *piece = buf + threadIdx.x * PIECE;
        PieceBases pb;
        pb.clear();
        // Plain sequence text: the structure pass's pack plus this lane's state.  Every other piece needs the chunk's
        // text: header pieces take it by masks (squeeze_header_piece), pieces with a blank or control byte outside header
        // text byte by byte, which costs the same for one lane as for 64.  They are queued, and the queue is worked off 64
        // pieces per wave pass -- with a header every kilobase (read sets) that is one pass per workgroup instead of one
        // per wave.  A queued piece's result is left in the piece's own 64 bytes of the image.
        const bool clean = !lane_state_dirty(lst) && !lane_state_header_piece(lst) && ls_in != LS_HEADER;   // the structure pass's definition: chunk_odd counts the rest
        {
            PiecePack pk;
            pk.c_lo = ((unsigned long long)me.p0.y << 32) | me.p0.x; pk.c_hi = ((unsigned long long)me.p0.w << 32) | me.p0.z;
            pk.restart = ((unsigned long long)me.p1.y << 32) | me.p1.x; pk.meta = me.p1.z; pk.pad_ = 0;
            if (all_clean || __any(clean)) squeeze_apply(pk, wk, pb, clean);
        }
        SQ_MARK(1);                                        // state composition + clean pieces
        if (!all_clean) {
            if (!clean) queue[atomicAdd(&n_queued, 1u)] = (uint16_t)threadIdx.x;
            __syncthreads();
            const uint32_t n_q = n_queued;
            SQ_MARK(8);
            for (uint32_t q0 = (threadIdx.x >> 6) * 64u; q0 < n_q; q0 += WG) {        // wave-uniform
                const uint32_t qi = q0 + (threadIdx.x & 63u);
                const bool work = qi < n_q;
                const uint32_t pc = work ? queue[qi] : 0u;
                const LaneState l2s = lane_state[(uint64_t)c * WG + pc];
                SeqWalker wq;
                wq.setup(k, recs, recs_cap, &racc);
                wq.begin(lane_state_ls(l2s), l2_compose(chunk_st, lane_state_l2(l2s), km1), stream_off + base + (uint64_t)pc * PIECE);
                // header pieces by masks (text in front of the first record is dropped: that piece takes the byte-wise walk)
                const bool by_masks = work && lane_state_header_piece(l2s) && wq.rec != 0u;
                const uint32_t nbq = (work && !by_masks) ? piece_len_of(pc, base, n_bytes) : 0u;
                uint8_t *pq = buf + pc * PIECE;
                PieceBases rb;
                rb.clear();
                SQ_MARK(9);
                if (__any(by_masks)) squeeze_header_piece(pq, wq, rb, by_masks);
                SQ_MARK(10);
                if (__any(nbq != 0u))
                    for_each_byte_of(pq, nbq, [&](uint32_t i, uint32_t ch, bool act) {
                        uint32_t code;
                        bool rst;
                        const bool take = wq.step(i, ch, act, code, rst);
                        rb.push(take, code, rst);
                    });
                SQ_MARK(11);
                wq.flush_rec_wave();
                SQ_MARK(12);
                wk.seq_tot += wq.seq_tot; wk.kmer_tot += wq.kmer_tot;                  // stream totals travel with the lane that did the work
                if (work) {
                    unsigned long long *res = reinterpret_cast<unsigned long long *>(pq);
                    res[0] = rb.code_lo; res[1] = rb.code_hi; res[2] = rb.restart; res[3] = rb.n;
                }
            }
            SQ_MARK(13);
            __syncthreads();
            SQ_MARK(14);
            if (!clean) {
                const unsigned long long *res = reinterpret_cast<const unsigned long long *>(piece);
                pb.code_lo = res[0]; pb.code_hi = res[1]; pb.restart = res[2]; pb.n = (uint32_t)res[3];
            }
            if (threadIdx.x == 0) n_queued = 0;                                       // read again only after the next barrier
        }
        SQ_MARK(2);                                        // queued pieces (+ two barriers)
        wk.flush_rec_wave();
        // where the lane's bases go in the chunk's slot: exclusive prefix of the counts over the workgroup
        uint32_t total;
        const uint32_t at = wg_excl_scan_u32(pb.n, scan_sh, total);
        lds_or_bits(slot_codes, 2u * at, pb.code_lo, pb.code_hi, 2u * pb.n);
        lds_or_bits(slot_rst, at, pb.restart, 0ull, pb.n);
        __syncthreads();
        SQ_MARK(3);                                        // record tallies, scan, bit packing into the slot image
        settle();
        SQ_MARK(4);                                        // delivery of the next chunk's loads
        // slot -> HBM, 16 bytes per lane, only the words that hold bases; the LDS copy is cleared for the next chunk
        const uint32_t code_q = (total + 63u) / 64u, rst_q = (total + 127u) / 128u;      // uint4 groups in use
        // (the lane's byte offset is formed here, opaque to the compiler: as a loop invariant the two lane addresses were
        // kept in registers, spilled once the header-piece code raised the pressure, and re-loaded from scratch right in
        // front of the stores -- a memory round trip per chunk, 0.204 -> 0.245 ms on the 800 Mbp genome)
        uint32_t lane_off = threadIdx.x * 16u;
        asm volatile("" : "+v"(lane_off));
        uint8_t *gc = reinterpret_cast<uint8_t *>(codes + (uint64_t)c * SLOT_CODE_WORDS);
        uint8_t *gr = reinterpret_cast<uint8_t *>(restarts + (uint64_t)c * SLOT_RST_WORDS);
        if (threadIdx.x < code_q) {
            *reinterpret_cast<uint4 *>(gc + lane_off) = reinterpret_cast<uint4 *>(slot_codes)[threadIdx.x];
            reinterpret_cast<uint4 *>(slot_codes)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        }
        if (threadIdx.x < rst_q) {
            *reinterpret_cast<uint4 *>(gr + lane_off) = reinterpret_cast<uint4 *>(slot_rst)[threadIdx.x];
            reinterpret_cast<uint4 *>(slot_rst)[threadIdx.x] = make_uint4(0, 0, 0, 0);
        }
        if (threadIdx.x == 0) n_bases[c] = total;
        SQ_MARK(5);                                        // slot store
    }
#ifdef PK_PHASE_PROF
    if (threadIdx.x == 0)
        for (int i = 0; i < 16; i++) atomicAdd(reinterpret_cast<unsigned long long *>(flags) - 1 + 14 + i, sq_prof[i]);
#endif
    wk.finish();
    recacc_finish(racc, recs, recs_cap, carry);
}

void launch_squeeze(const uint8_t *fasta, uint64_t n, uint64_t stream_off, const LaneState *lane_state, const PiecePack *packs, const L2 *st2,
                    const uint32_t *chunk_odd, uint32_t k, uint32_t n_chunks, uint32_t n_wg, uint32_t chunks_per_wg, uint32_t *codes, uint32_t *restarts, uint32_t *n_bases,
                    DevRec *recs, uint64_t recs_cap, Carry *carry, uint32_t *flags, hipStream_t s) {
    static const bool lit = !(getenv("PK_K15") && atoi(getenv("PK_K15")) == 0);
#define PK_SQUEEZE(KC) hipLaunchKernelGGL(k_squeeze<KC>, dim3(n_wg), dim3(WG), 0, s, fasta, n, stream_off, lane_state, packs, st2, chunk_odd, k, n_chunks, \
                                          chunks_per_wg, codes, restarts, n_bases, recs, recs_cap, carry, flags)
    if (lit && k == 15) PK_SQUEEZE(15);
    else if (lit && k == 17) PK_SQUEEZE(17);
    else PK_SQUEEZE(0);
#undef PK_SQUEEZE
}

}  // namespace pk
