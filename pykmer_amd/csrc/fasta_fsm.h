// fasta_fsm.h -- device-side FASTA state machine shared by the structure-scan and k-mer kernels.
//
// The reference parses FASTA line by line on the host (indexer.py:45-99): strip() each line, skip
// empty ones, a line starting with '>' opens a record, every other line is sequence; k-mers run
// across line breaks inside a record and never across records; characters outside ACGTacgt map to
// None and void every window that contains them (indexer.py:36-41,144).
//
// On the GPU the byte stream is cut into fixed 64-byte lane pieces (256 lanes = one 16 KiB chunk
// per workgroup).  A piece can start anywhere -- mid-line, mid-header, mid-k-mer -- so each piece
// is first reduced to a small *summary* (how it transforms the parser state), summaries are
// combined with an associative compose() in prefix scans (lane -> wave -> workgroup -> grid), and
// only then does every lane re-walk its piece from its now exact start state.  Two levels:
//   L1  line state: are we at line start / inside a header line / inside a sequence line
//   L2  record index, pending (not yet classified) whitespace, and the last <= k-1 valid bases
// This is exact for any input: unwrapped 100 Mbp lines, CRLF, blank lines, interior blanks ...
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pk {

constexpr int WG = 256;            // threads per workgroup
constexpr int PIECE = 64;          // bytes walked by one lane
constexpr int CHUNK = WG * PIECE;  // bytes per workgroup (16 KiB)
constexpr int LDS_STRIDE = 80;     // lane piece stride in LDS: 20 banks -> conflict-free ds_read_b128

enum : uint32_t { LS_START = 0, LS_HEADER = 1, LS_SEQ = 2 };

__device__ __forceinline__ bool is_term(uint32_t c) { return c == 10u || c == 13u; }
// str.strip() whitespace, ASCII subset (indexer.py:56): \t \n \v \f \r, FS GS RS US, space
__device__ __forceinline__ bool is_ws(uint32_t c) { return c == 32u || (c - 9u) <= 4u || (c - 28u) <= 3u; }
// CONV (indexer.py:36-41): 0..3 for ACGT/acgt, 4 otherwise.  ASCII bits 1-2 of A,C,G,T are 00,01,11,10
// (same for lower case), so code = b ^ (b >> 1); validity is a bit test in the 32-letter block.
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
    const uint32_t b = (c >> 1) & 3u;
    const uint32_t code = b ^ (b >> 1);
    const bool valid = (c >> 6) == 1u && ((0x0010008Au >> (c & 31u)) & 1u);
    return valid ? code : 4u;
}

// ---------------------------------------------------------------- L1: line state -------------
// Encoding: 0 = identity; else bit3 set, bit0 = piece contains a line terminator, bits1-2 = kind:
// the line state at the end of the piece if it has a terminator, otherwise what LS_START becomes
// (LS_HEADER and LS_SEQ pass through a terminator-free piece unchanged).
typedef uint32_t L1;
__device__ __forceinline__ L1 l1_make(bool has_term, uint32_t kind) { return 8u | (has_term ? 1u : 0u) | (kind << 1); }
__device__ __forceinline__ L1 l1_state(uint32_t ls) { return l1_make(true, ls); }
__device__ __forceinline__ uint32_t l1_kind(L1 a) { return (a >> 1) & 3u; }
__device__ __forceinline__ L1 l1_compose(L1 a, L1 b) {   // a first, then b
    if (!b) return a;
    if (!a) return b;
    uint32_t ak = l1_kind(a), bk = l1_kind(b);
    uint32_t kind = (b & 1u) ? bk : (ak == LS_START ? bk : ak);
    return 8u | ((a | b) & 1u) | (kind << 1);
}

// ---------------------------------------------------------------- L2: record / run state -----
// flags: bit0 non-identity, bit1 p_reset (piece holds an event that zeroes pending whitespace),
// bit2 F (piece starts in a sequence line with a non-blank character: breaks the run iff the
// incoming pending-whitespace count is > 0), bit3 brk (run broken inside, or >= k-1 valid bases:
// incoming bases irrelevant), bits 8-15 len (valid bases after the last break, capped at k-1).
struct L2 {
    uint32_t flags;
    uint32_t bits;    // those `len` bases, 2 bits each, newest lowest
    uint32_t rec;     // record headers opened
    uint64_t p_tail;  // pending whitespace at the end (since the last reset, or since the start)
};
constexpr uint32_t F_NONID = 1u, F_PRESET = 2u, F_FRONT = 4u, F_BRK = 8u;

__device__ __forceinline__ uint32_t l2_len(const L2 &a) { return (a.flags >> 8) & 0xffu; }
__device__ __forceinline__ uint32_t bases_mask(uint32_t nb) { return nb >= 16u ? 0xffffffffu : ((1u << (2u * nb)) - 1u); }
__device__ __forceinline__ L2 l2_identity() { L2 z; z.flags = 0; z.bits = 0; z.rec = 0; z.p_tail = 0; return z; }

// A concrete parser state is the summary "whatever came before, the state is now this".
__device__ __forceinline__ L2 l2_state(uint64_t pending, uint32_t run, uint32_t bases, uint32_t rec) {
    L2 s; s.flags = F_NONID | F_PRESET | F_BRK | (run << 8); s.bits = bases; s.rec = rec; s.p_tail = pending; return s;
}

__device__ __forceinline__ L2 l2_compose(const L2 &a, const L2 &b, uint32_t km1) {   // a first, then b
    if (!(b.flags & F_NONID)) return a;
    if (!(a.flags & F_NONID)) return b;
    L2 c;
    c.rec = a.rec + b.rec;
    c.p_tail = (b.flags & F_PRESET) ? b.p_tail : a.p_tail + b.p_tail;
    uint32_t f = F_NONID | ((a.flags | b.flags) & F_PRESET) | (a.flags & F_FRONT);
    // b's first character closes a's trailing whitespace as *interior* whitespace -> run broken.
    // (If a has no reset and p_tail == 0 it holds no sequence-line event, so b cannot carry F.)
    bool front_break = (b.flags & F_FRONT) && (a.p_tail > 0);
    uint32_t blen = l2_len(b), alen = l2_len(a);
    if ((b.flags & F_BRK) || front_break || blen >= km1) {
        f |= F_BRK | (blen << 8);
        c.bits = b.bits;
    } else {
        uint32_t tot = alen + blen;
        if (tot > km1) tot = km1;
        f |= (a.flags & F_BRK) | (tot << 8);
        c.bits = ((blen >= 16u ? 0u : (a.bits << (2u * blen))) | b.bits) & bases_mask(km1);   // 32 bits hold the newest 16 (k = 19, 21: len runs to 20)
    }
    c.flags = f;
    return c;
}

// Per-lane hand-off from the structure pass to the squeeze pass: the lane's L2 prefix RELATIVE to its chunk start
// (compose it behind the chunk's state to get the exact state), its incoming line state and whether its piece needs
// the byte-wise machine.  8 bytes per 64-byte piece: the squeeze pass forms no k-mers, so the carried bases are left out
// (kmer_fuse.hip takes the bases in front of a slot from the CHUNK's state).
struct LaneState {
    uint32_t flags;     // L2 flags (incl. the run length) | ls_in << 16 | dirty << 18 | header piece << 19
    uint32_t rec_tail;  // records opened before the lane, within the chunk (<= 8192) | pending blanks << 16 (<= 16384)
};
// dirty: the piece needs the byte-wise machine; header piece: it holds header text but the mask path understands it
__device__ __forceinline__ LaneState lane_state_pack(const L2 &rel, uint32_t ls_in, bool dirty, bool header_piece) {
    LaneState o;
    o.flags = rel.flags | (ls_in << 16) | (dirty ? (1u << 18) : 0u) | (header_piece ? (1u << 19) : 0u);
    o.rec_tail = rel.rec | ((uint32_t)rel.p_tail << 16);
    return o;
}
__device__ __forceinline__ L2 lane_state_l2(const LaneState &o) {
    L2 r; r.flags = o.flags & 0xffffu; r.bits = 0; r.rec = o.rec_tail & 0xffffu; r.p_tail = o.rec_tail >> 16; return r;
}
__device__ __forceinline__ uint32_t lane_state_ls(const LaneState &o) { return (o.flags >> 16) & 3u; }
__device__ __forceinline__ bool lane_state_dirty(const LaneState &o) { return (o.flags >> 18) & 1u; }
__device__ __forceinline__ bool lane_state_header_piece(const LaneState &o) { return (o.flags >> 19) & 1u; }

// ---- wave / workgroup exclusive scans (64-wide wavefronts, non-commutative operator) ----------
__device__ __forceinline__ L1 wave_incl_scan_l1(L1 v, int lane) {
    // Short cut for the usual wave: no piece ends inside a header line, and a piece without a terminator ends in
    // sequence text (it is not all blanks).  l1_compose then never lets an earlier piece's state through -- a
    // terminator in b decides, and without one both START and SEQ before b give SEQ after it -- so the prefix
    // ending in lane i has lane i's own end state, and "has a terminator" is an OR over the lanes up to i.
    const uint32_t kind = l1_kind(v);
    if (__all(v != 0u && kind != LS_HEADER && ((v & 1u) || kind == LS_SEQ))) {
        const unsigned long long terms = __ballot((v & 1u) != 0u);
        const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        return (v & ~1u) | ((terms & upto) ? 1u : 0u);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        L1 o = __shfl_up(v, d, 64);
        if (lane >= d) v = l1_compose(o, v);
    }
    return v;
}
__device__ __forceinline__ L2 shfl_up_l2(const L2 &v, int d) {
    L2 o;
    o.flags = __shfl_up(v.flags, d, 64);
    o.bits = __shfl_up(v.bits, d, 64);
    o.rec = __shfl_up(v.rec, d, 64);
    o.p_tail = __shfl_up((unsigned long long)v.p_tail, d, 64);
    return o;
}
__device__ __forceinline__ L2 wave_incl_scan_l2(L2 v, int lane, uint32_t km1) {
    // Plain sequence text and header lines: every lane's piece restarts the window by itself (F_BRK: a break inside it,
    // or >= k-1 valid bases at its end) and leaves no pending blanks.  Then a prefix ending in lane i is lane i's own
    // summary, except that F_FRONT is inherited from the leftmost piece (l2_compose keeps a's) and the records opened
    // add up: no composition, and shuffles only in a wave that opens a record (a read set: every wave, and the
    // six-step composition was a fifth of the structure pass there).
    const uint32_t want = F_NONID | F_PRESET | F_BRK;
    if (__all((v.flags & want) == want && v.p_tail == 0ull)) {
        const uint32_t front0 = (uint32_t)__builtin_amdgcn_readlane((int)v.flags, 0) & F_FRONT;
        v.flags = (v.flags & ~F_FRONT) | front0;
        if (__any(v.rec != 0u)) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(v.rec, d, 64);
                if (lane >= d) v.rec += o;
            }
        }
        return v;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        L2 o = shfl_up_l2(v, d);
        if (lane >= d) v = l2_compose(o, v, km1);
    }
    return v;
}

// Exclusive scan over the 256 lanes of a workgroup, seeded with `seed` (the state before lane 0).
// Returns this lane's start state; *total (valid in every lane) = seed . lane0 . ... . lane255.
__device__ __forceinline__ L1 wg_excl_scan_l1(L1 mine, L1 seed, L1 *sh /*[4]*/, L1 *total) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    L1 inc = wave_incl_scan_l1(mine, lane);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    L1 pre = seed;
    for (int i = 0; i < w; i++) pre = l1_compose(pre, sh[i]);
    L1 tot = pre;
    for (int i = w; i < WG / 64; i++) tot = l1_compose(tot, sh[i]);
    *total = tot;
    L1 up = __shfl_up(inc, 1, 64);
    L1 res = (lane == 0) ? pre : l1_compose(pre, up);
    __syncthreads();
    return res;
}
__device__ __forceinline__ L2 wg_excl_scan_l2(const L2 &mine, const L2 &seed, L2 *sh /*[4]*/, L2 *total, uint32_t km1) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    L2 inc = wave_incl_scan_l2(mine, lane, km1);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    L2 pre = seed, tot;
    // the same short cut one level up (see wave_incl_scan_l2): wave totals that each restart the window
    // compose to "the last one, with the first one's F_FRONT"
    const uint32_t want = F_NONID | F_PRESET | F_BRK;
    bool plain = !(seed.flags & F_NONID);
    for (int i = 0; i < WG / 64; i++) plain = plain && (sh[i].flags & want) == want && sh[i].p_tail == 0ull;
    if (plain) {                                                         // uniform: sh[] is the same for every lane
        const uint32_t front0 = sh[0].flags & F_FRONT;
        uint32_t rec_before = 0, rec_all = 0;
        for (int i = 0; i < WG / 64; i++) { if (i < w) rec_before += sh[i].rec; rec_all += sh[i].rec; }
        if (w > 0) { pre = sh[w - 1]; pre.flags = (pre.flags & ~F_FRONT) | front0; pre.rec = rec_before; }
        tot = sh[WG / 64 - 1];
        tot.flags = (tot.flags & ~F_FRONT) | front0;
        tot.rec = rec_all;
    } else {
        for (int i = 0; i < w; i++) pre = l2_compose(pre, sh[i], km1);
        tot = pre;
        for (int i = w; i < WG / 64; i++) tot = l2_compose(tot, sh[i], km1);
    }
    *total = tot;
    L2 up = shfl_up_l2(inc, 1);
    L2 res = (lane == 0) ? pre : l2_compose(pre, up, km1);
    __syncthreads();
    return res;
}

// ---- staging: one 16 KiB chunk, coalesced 16 B per lane, into padded LDS pieces ----------------
// fasta must be 16-byte aligned.  Bytes at or beyond n_bytes are never read from memory.
// Bytes past the end of the stream are staged as 0, which every consumer treats like a terminator or
// skips by length (the clean-piece walk relies on "byte <= 13 is no sequence character").
__device__ __forceinline__ void stage_chunk(const uint8_t *__restrict__ fasta, uint64_t chunk_base, uint64_t n_bytes,
                                            uint8_t *lds) {
#pragma unroll
    for (int i = 0; i < CHUNK / (WG * 16); i++) {
        uint32_t p = i * WG + threadIdx.x;
        uint64_t g = chunk_base + (uint64_t)p * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16u <= n_bytes) {
            v = *reinterpret_cast<const uint4 *>(fasta + g);
        } else if (g < n_bytes) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < (uint32_t)(n_bytes - g); j++) w[j >> 2] |= (uint32_t)fasta[g + j] << (8u * (j & 3u));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint32_t bo = p * 16u;
        *reinterpret_cast<uint4 *>(lds + (bo / PIECE) * LDS_STRIDE + (bo % PIECE)) = v;
    }
}
__device__ __forceinline__ uint32_t piece_len_of(uint32_t piece, uint64_t chunk_base, uint64_t n_bytes) {
    uint64_t start = chunk_base + (uint64_t)piece * PIECE;
    if (start >= n_bytes) return 0;
    uint64_t left = n_bytes - start;
    return left < (uint64_t)PIECE ? (uint32_t)left : (uint32_t)PIECE;
}
__device__ __forceinline__ uint32_t piece_len(uint64_t chunk_base, uint64_t n_bytes) {
    uint64_t start = chunk_base + (uint64_t)threadIdx.x * PIECE;
    if (start >= n_bytes) return 0;
    uint64_t left = n_bytes - start;
    return left < (uint64_t)PIECE ? (uint32_t)left : (uint32_t)PIECE;
}

// Walks the lane's piece in LDS calling f(index, byte, active) for each of the PIECE byte slots.
// A ROLLED loop on purpose: the bodies below are 100+ instructions, and unrolling them 64x produced
// 30k-120k-instruction kernels that thrash the instruction cache (measured: 20x slower).  The trip
// count is uniform (callers may use wave-wide ballots inside f); `active` is false past the lane's
// last byte.  16 bytes are fetched per ds_read_b128 and shifted through four registers.
template <class Fn>
__device__ __forceinline__ void for_each_byte_of(const uint8_t *mine, uint32_t nb, Fn &&f) {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < (uint32_t)PIECE; i++) {
        if ((i & 15u) == 0u) {
            uint4 v = *reinterpret_cast<const uint4 *>(mine + i);
            w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
        } else if ((i & 3u) == 0u) {
            w0 = w1; w1 = w2; w2 = w3;
        }
        f(i, (w0 >> ((i & 3u) * 8u)) & 0xffu, i < nb);
    }
}

// L1 summary of the lane's piece.  `dirty` comes back true if the piece holds anything besides
// sequence characters and line terminators (a blank or control byte other than \n / \r, or a '>';
// the clean-piece loops rely on "byte > 13 means sequence character"): such pieces need
// the full state machine; pieces that are not dirty and do not start inside a header line are
// "clean" and take the short paths below and in kmer_walk.h.
__device__ __forceinline__ L1 piece_l1_at(const uint8_t *mine, uint32_t nb, bool &dirty) {   // mine: the piece's 64 bytes
    uint32_t st = LS_START;
    bool ht = false, d = false;
    for_each_byte_of(mine, nb, [&](uint32_t, uint32_t c, bool act) {
        const bool ws = is_ws(c), gt = c == '>';
        const bool term = act && is_term(c);
        const bool opens = act && st == LS_START && !ws;
        st = term ? (uint32_t)LS_START : opens ? (gt ? (uint32_t)LS_HEADER : (uint32_t)LS_SEQ) : st;
        ht |= term;
        d |= act && (((ws || c < 0x21u) && !term) || gt);
    });
    dirty = d;
    return nb ? l1_make(ht, st) : 0u;
}
// SWAR helpers: 0x80 in every byte of v that is zero / below n (n <= 128); exact, no cross-byte borrows
__device__ __forceinline__ uint32_t swar_zero(uint32_t v) { return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu); }
__device__ __forceinline__ uint32_t swar_less(uint32_t v, uint32_t n_rep) {
    return ~((((v & 0x7f7f7f7fu) | 0x80808080u) - n_rep) | v) & 0x80808080u;
}

// L2 summary of the lane's piece, given its exact incoming line state.
__device__ __forceinline__ L2 piece_l2_at(const uint8_t *mine, uint32_t nb, uint32_t ls_in, uint32_t km1) {   // mine: the piece's 64 bytes
    uint32_t ls = ls_in, flags = F_NONID, len = 0, bits = 0, rec = 0, pt = 0;
    const uint32_t bm = bases_mask(km1);
    for_each_byte_of(mine, nb, [&](uint32_t i, uint32_t c, bool act) {
        const bool term = is_term(c), ws = is_ws(c), gt = c == '>';
        const bool at_start = ls == LS_START;
        const bool hdr_start = act && at_start && !ws && gt;
        const bool seqchar = act && !ws && (ls == LS_SEQ || (at_start && !gt));
        const bool seqws = act && ws && !term && ls == LS_SEQ;
        const bool t = act && term;
        if (seqchar && i == 0 && ls_in == LS_SEQ) flags |= F_FRONT;
        const uint32_t code = base_code(c);
        const bool valid = seqchar && code < 4u;
        // a break: new record, invalid character, or whitespace inside the piece that turned out interior
        const bool brk = hdr_start || (seqchar && (code > 3u || pt != 0u));
        if (brk) { flags |= F_BRK; len = 0; bits = 0; }
        if (valid) { bits = ((bits << 2) | code) & bm; len = len < km1 ? len + 1 : len; }
        if (t || seqchar) flags |= F_PRESET;
        pt = (t || seqchar) ? 0u : pt + (seqws ? 1u : 0u);
        rec += hdr_start ? 1u : 0u;
        ls = t ? (uint32_t)LS_START : hdr_start ? (uint32_t)LS_HEADER : seqchar ? (uint32_t)LS_SEQ : ls;
    });
    if (nb == 0) return l2_identity();
    if (len >= km1) flags |= F_BRK;
    L2 s; s.flags = flags | (len << 8); s.bits = bits; s.rec = rec; s.p_tail = pt;
    return s;
}


// ---------------------------------------------------------------- one classification per piece ----
// The structure pass looks at every byte once and leaves, for every 64-byte piece, what the squeeze pass needs of it:
// the 2-bit codes of its valid bases pushed together (text order, base j at bits 2j), one restart bit per base (a
// character that maps to None lay between it and the base before, indexer.py:36-41,144) and three counts.  The squeeze
// pass then reads these 32 bytes instead of the 64 bytes of text -- it only adds what depends on the parser state the
// piece is entered with.  (Until the middle of round 2 both passes classified the text: 35 vector instructions per byte
// between them, three quarters of it the same SWAR tests twice.)  Pieces that are not clean (header text, a blank, a
// control byte; or entered inside a header line) carry a record too, but it is not used: the squeeze pass goes back to
// their text.
struct PiecePack {
    unsigned long long c_lo, c_hi;     // codes of bases 0-31, 32-63
    unsigned long long restart;        // bit j: base j restarts the run -- without what the incoming state adds at base 0
    uint32_t meta;                     // valid bases | sequence characters (valid or not) << 8 | first byte is a sequence character << 16
    uint32_t pad_;
};
static_assert(sizeof(PiecePack) == 32, "two 16-byte stores per piece");
__device__ __forceinline__ uint32_t pack_n_valid(const PiecePack &p) { return p.meta & 0xffu; }
__device__ __forceinline__ uint32_t pack_n_seq(const PiecePack &p) { return (p.meta >> 8) & 0xffu; }
__device__ __forceinline__ bool pack_first_is_seq(const PiecePack &p) { return (p.meta >> 16) & 1u; }

__device__ __forceinline__ uint32_t revpairs32(uint32_t x) {            // 2-bit field p -> field 15 - p
    const uint32_t y = __builtin_bitreverse32(x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}
__device__ __forceinline__ uint32_t movemask4(uint32_t flags80) {       // 0x80-per-byte flags -> 4 bits, byte 0 lowest
    return (flags80 * 0x00204081u) >> 28;
}

// What piece_scan finds in the piece's nb bytes (byte i <-> bit i); bytes past nb count as terminators.
struct PieceMasks {
    unsigned long long term, valid;    // line terminator; ACGTacgt
    unsigned long long blank, gt;      // blank or control byte that is no terminator (below 0x21); '>'
};

// `mine`: the piece's 64 bytes (16-byte aligned).  All lanes of the wave call it together.  cw: the 2-bit code of every
// byte (byte i -> bits 2i of the 128 bits), meaningful where `valid` is set.
__device__ __forceinline__ void piece_scan(const uint8_t *mine, uint32_t nb, PieceMasks &m, uint32_t (&cw)[4]) {
    const uint4 *quads = reinterpret_cast<const uint4 *>(mine);
    uint32_t vm[2] = {0, 0};
    cw[0] = cw[1] = cw[2] = cw[3] = 0;
#pragma unroll
    for (int q = 0; q < PIECE / 16; q++) {
        const uint4 v = quads[q];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = w4[j];
            const int d = q * 4 + j;                                     // dword index in the piece: bytes 4d .. 4d+3
            // The low three bits tell the four letters apart (A 0x41 -> 1, C 0x43 -> 3, T 0x54 -> 4, G 0x47 -> 7, either
            // case): a byte permute with them as selector is an eight-entry table for four bytes at once -- the letter
            // the byte would have to be (compared with the byte, case bit cleared: zero byte <=> that letter), and its
            // code (A 0, C 1, G 2, T 3; CONV, indexer.py:36-41).  13 vector instructions per dword; the bit arithmetic on
            // the letter bits 2-1 this replaced took 21.
            const uint32_t sel = w & 0x07070707u;
            const uint32_t diff = (w & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, sel);
            const uint32_t valid = swar_zero(diff);
            const uint32_t code = __builtin_amdgcn_perm(0x02000003u, 0x01000000u, sel);
            const uint32_t code8 = (code * 0x01041040u) >> 24;           // byte i -> bits 2i .. 2i+1 of one byte
            vm[d >> 3] |= movemask4(valid) << (4 * (d & 7));
            cw[d >> 2] |= code8 << (8 * (d & 3));
        }
    }
    const unsigned long long in_range = nb >= 64u ? ~0ull : ((1ull << nb) - 1ull);
    const unsigned long long V = ((((unsigned long long)vm[1]) << 32) | vm[0]) & in_range;
    // everything that is no base -- in plain sequence text one line terminator per piece -- is looked at byte by byte;
    // a wave that holds a piece with many of them (a run of N, a header) tests all bytes four at a time instead
    unsigned long long T = ~in_range, blank = 0, gt = 0, todo = ~V & in_range;
    if (__any(__popcll(todo) > 6)) {
        uint32_t tm[2] = {0, 0}, bm[2] = {0, 0}, gm[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < PIECE / 16; q++) {
            const uint4 v = quads[q];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int d = q * 4 + j;
                const uint32_t t = swar_zero(w4[j] ^ 0x0a0a0a0au) | swar_zero(w4[j] ^ 0x0d0d0d0du);
                tm[d >> 3] |= movemask4(t) << (4 * (d & 7));
                bm[d >> 3] |= movemask4(swar_less(w4[j], 0x21212121u) & ~t) << (4 * (d & 7));
                gm[d >> 3] |= movemask4(swar_zero(w4[j] ^ 0x3e3e3e3eu)) << (4 * (d & 7));
            }
        }
        T |= ((((unsigned long long)tm[1]) << 32) | tm[0]) & in_range;
        blank = ((((unsigned long long)bm[1]) << 32) | bm[0]) & in_range;
        gt = ((((unsigned long long)gm[1]) << 32) | gm[0]) & in_range;
    } else {
        while (__any(todo != 0ull)) {
            if (todo) {
                const uint32_t p = (uint32_t)__builtin_ctzll(todo);
                const uint32_t c = mine[p];
                const unsigned long long bit = 1ull << p;
                if (is_term(c)) T |= bit;
                else if (c < 0x21u) blank |= bit;
                else if (c == '>') gt |= bit;
                todo &= todo - 1ull;
            }
        }
    }
    m.term = T; m.valid = V; m.blank = blank; m.gt = gt;
}

// Pushes the bases `V` of a piece together and derives their restart bits.  `none`: the positions that break the run
// (characters that map to None; header text); everything that is not in V is a hole.  compact = false leaves the codes as
// they are (a piece whose pack nobody will read: header text is mostly holes).
__device__ __forceinline__ void piece_compact(unsigned long long V, unsigned long long none, const uint32_t (&cw)[4], bool compact,
                                              uint32_t n_seq, bool first_is_seq, PiecePack &pk) {
    // restart flags: carry from every breaking position through the non-base positions above it into the next base
    unsigned long long F = ((~V) + none) & V;
    const uint32_t nv = (uint32_t)__popcll(V);
    unsigned long long c_lo = ((unsigned long long)cw[1] << 32) | cw[0], c_hi = ((unsigned long long)cw[3] << 32) | cw[2];
    unsigned long long holes = (V && compact) ? (~V & ((1ull << (63 - __builtin_clzll(V))) - 1ull)) : 0ull;   // below the highest base only
    // A whole RUN of neighbouring holes goes at once (the Ns before the first base after a gap: up to 63 positions,
    // which one-at-a-time deletion turned into 63 rounds for the whole wave -- 0.3 ms of the structure pass on a genome
    // with 3 % N).  Plain text has one line terminator per piece, now and then two: two deletions are laid out straight
    // (a lane without a hole deletes nothing), and only a wave that holds a piece with more runs loops.
    auto delete_lowest_run = [&]() {
        const bool any = holes != 0ull;
        const uint32_t p = any ? (uint32_t)__builtin_ctzll(holes) : 0u;
        const uint32_t L = any ? (uint32_t)__builtin_ctzll(~(holes >> p)) : 0u;       // p + L <= 63: the highest base lies above
        const unsigned long long low = any ? ((1ull << p) - 1ull) : ~0ull;
        F = (F & low) | ((F >> L) & ~low);
        holes = (holes >> L) & ~low;
        // the same on the 128 bits of codes: fields below p stay, fields above move down by L
        const unsigned long long low_lo = !any || p >= 32u ? ~0ull : ((1ull << (2u * p)) - 1ull);
        const unsigned long long low_hi = !any ? ~0ull : (p <= 32u ? 0ull : ((1ull << (2u * (p - 32u))) - 1ull));
        const uint32_t sh = 2u * L;                                                   // 0 .. 126
        unsigned long long s_lo, s_hi;
        if (sh >= 64u) { s_lo = c_hi >> (sh - 64u); s_hi = 0ull; }
        else if (sh) { s_lo = (c_lo >> sh) | (c_hi << (64u - sh)); s_hi = c_hi >> sh; }
        else { s_lo = c_lo; s_hi = c_hi; }
        c_lo = (c_lo & low_lo) | (s_lo & ~low_lo);
        c_hi = (c_hi & low_hi) | (s_hi & ~low_hi);
    };
    delete_lowest_run();
    delete_lowest_run();
    while (__any(holes != 0ull)) delete_lowest_run();
    // holes above the highest base were left where they are; clear everything past the nv bases
    const unsigned long long keep = nv >= 64u ? ~0ull : ((1ull << nv) - 1ull);
    F &= keep;
    if (nv < 32u) { c_lo &= (1ull << (2u * nv)) - 1ull; c_hi = 0; }
    else if (nv < 64u) c_hi &= (1ull << (2u * (nv - 32u))) - 1ull;
    pk.c_lo = c_lo; pk.c_hi = c_hi; pk.restart = F;
    pk.meta = nv | (n_seq << 8) | ((first_is_seq ? 1u : 0u) << 16);
    pk.pad_ = 0;
}

// the pack of a piece read as plain sequence text: everything that is no terminator is a sequence character
__device__ __forceinline__ void classify_piece(const uint8_t *mine, uint32_t nb, PieceMasks &m, PiecePack &pk) {
    uint32_t cw[4];
    piece_scan(mine, nb, m, cw);
    const unsigned long long S = ~m.term;
    piece_compact(m.valid, S & ~m.valid, cw, (m.blank | m.gt) == 0ull, (uint32_t)__popcll(S), (S & 1ull) != 0ull, pk);
}

// Header lines by masks.  In a piece whose lines do not begin with a blank or control byte, a line is a header line exactly
// if its first byte is '>' (indexer.py:66: strip() has nothing to strip in front); the line the piece starts in is one if the
// piece is entered in state HEADER, or at a line start with '>' first.  Header text = from there up to the line's
// terminator: adding the start bits into the mask of non-terminators ripples a carry through exactly that stretch.
__device__ __forceinline__ unsigned long long header_starts(const PieceMasks &m, uint32_t ls_in) {       // '>' that open a record
    return ((m.term << 1) | (ls_in == LS_START ? 1ull : 0ull)) & m.gt;
}
__device__ __forceinline__ unsigned long long header_text(const PieceMasks &m, uint32_t ls_in) {
    const unsigned long long starts = header_starts(m, ls_in) | (ls_in == LS_HEADER ? 1ull : 0ull);
    const unsigned long long x = ~m.term;
    return ((x + starts) ^ x) & x;
}
// No line that BEGINS in the piece begins with a blank or control byte: then the L1 summary needs no byte-wise walk, and
// header lines are the lines whose first byte is '>'.  A blank as the piece's first byte is fine if a terminator follows
// somewhere (the summary is then decided behind the last terminator) -- it only matters when the piece turns out to be
// entered at a line start, which first_byte_plain() rules out once the line state is known.  (A header line with words in
// it crosses a piece seam at a blank in one case out of ten: a read set had such a piece in most of its chunks, and one
// piece for the byte-wise machine costs the workgroup as much as sixty-four.)
__device__ __forceinline__ bool lines_start_plain(const PieceMasks &m) {
    return ((m.term << 1) & m.blank) == 0ull && (!(m.blank & 1ull) || m.term != 0ull);
}
__device__ __forceinline__ bool first_byte_plain(const PieceMasks &m, uint32_t ls_in) { return !(m.blank & 1ull) || ls_in != LS_START; }

// L1 summary from the masks of a FULL piece with lines_start_plain(): the state after the last terminator is START; what
// follows it is a header line if it begins with '>' and sequence text otherwise; blanks further in change nothing.
__device__ __forceinline__ L1 l1_of_piece(const PieceMasks &m) {
    const unsigned long long T = m.term;
    uint32_t kind;
    if (T) {
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(T);
        kind = last == 63u ? (uint32_t)LS_START : (((m.gt >> (last + 1u)) & 1ull) ? (uint32_t)LS_HEADER : (uint32_t)LS_SEQ);
    } else {
        kind = (m.gt & 1ull) ? (uint32_t)LS_HEADER : (uint32_t)LS_SEQ;   // what START becomes; HEADER and SEQ pass through
    }
    return l1_make(T != 0ull, kind);
}

// L2 summary from masks: `valid` the bases that count, `brk` the positions that break the run (None characters, header
// text), `seq` the sequence characters, `n_hdr` record headers opened; pk = piece_compact(valid, brk).  What the byte-wise
// walk (piece_l2_at) finds for a piece without blanks outside header text.
__device__ __forceinline__ L2 l2_of_masks(unsigned long long term, unsigned long long valid, unsigned long long brk, unsigned long long seq,
                                          uint32_t n_hdr, const PiecePack &pk, uint32_t nb, uint32_t ls_in, uint32_t km1) {
    if (nb == 0) return l2_identity();
    const unsigned long long in_range = nb >= 64u ? ~0ull : ((1ull << nb) - 1ull);
    const uint32_t nv = pack_n_valid(pk);
    uint32_t tail = nv;
    if (brk) {
        const uint32_t hb = 63u - (uint32_t)__builtin_clzll(brk);
        tail = hb == 63u ? 0u : (uint32_t)__popcll(valid >> (hb + 1u));
    }
    uint32_t flags = F_NONID;
    if ((term & in_range) | seq) flags |= F_PRESET;                      // a terminator or a sequence character zeroes pending blanks
    if (ls_in == LS_SEQ && (seq & 1ull)) flags |= F_FRONT;
    if (brk || tail >= km1) flags |= F_BRK;
    const uint32_t len = tail < km1 ? tail : km1;
    const uint32_t n16 = len < 16u ? len : 16u;                          // 32 bits hold the newest 16
    uint32_t bits = 0;
    if (n16) {
        const uint32_t from = nv - n16;                                  // first of the n16 newest bases
        unsigned long long x = from >= 32u ? (pk.c_hi >> (2u * (from - 32u)))
                                           : (from ? ((pk.c_lo >> (2u * from)) | (pk.c_hi << (64u - 2u * from))) : pk.c_lo);
        uint32_t x32 = (uint32_t)x;
        if (n16 < 16u) x32 &= (1u << (2u * n16)) - 1u;
        bits = revpairs32(x32) >> (32u - 2u * n16);                      // newest base lowest
    }
    L2 s; s.flags = flags | (len << 8); s.bits = bits; s.rec = n_hdr; s.p_tail = 0;
    return s;
}
// a clean piece (plain sequence text, not entered inside a header line)
__device__ __forceinline__ L2 l2_of_clean_piece(const PieceMasks &m, const PiecePack &pk, uint32_t nb, uint32_t ls_in, uint32_t km1) {
    const unsigned long long seq = ~m.term;
    return l2_of_masks(m.term, m.valid, seq & ~m.valid, seq, 0u, pk, nb, ls_in, km1);
}

}  // namespace pk
