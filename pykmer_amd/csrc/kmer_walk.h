// kmer_walk.h -- the per-lane k-mer walker shared by both table-update versions.
//
// One Walker holds the exact parser state at a lane's first byte (from the L1/L2 scans) and advances
// it one byte per step(): line state, pending whitespace, the rolling forward / reverse-complement
// values (indexer.py:146-150, as rolling updates) and the per-record tallies the reference keeps
// (seq_len, valid windows, header text extent; indexer.py:75-95,349-351).  step() is written
// branch-light -- predicates and selects for everything that happens on most bytes, real branches
// only for the rare events (a header opens, interior whitespace resolves).
#pragma once
#include "fasta_fsm.h"
#include "pk_kernels.h"

namespace pk {

// Per-workgroup accumulator for the record most lanes are inside of (a 16 KiB chunk usually lies in
// one record): lanes add their tallies here, one lane writes them to HBM when the workgroup moves on
// to another record or finishes.  Without it every lane hit the same DevRec with global atomics --
// 25 M same-address atomics on an 800 Mbp genome, ~100 ms.
struct RecAcc {
    uint32_t rec;                          // 1-based record the sums belong to (0 = none)
    unsigned long long seq, kmers;         // pending sums for that record
    unsigned long long tot_seq, tot_kmers; // stream totals gathered by this workgroup
};

__device__ __forceinline__ void recacc_init(RecAcc &A) {
    if (threadIdx.x == 0) { A.rec = 0; A.seq = 0; A.kmers = 0; A.tot_seq = 0; A.tot_kmers = 0; }
}
// thread 0 only, with the workgroup quiescent (after a barrier): write the pending sums out
__device__ __forceinline__ void recacc_spill(RecAcc &A, DevRec *recs, uint64_t recs_cap) {
    if (A.rec && A.rec <= recs_cap) {
        if (A.seq) atomicAdd((unsigned long long *)&recs[A.rec - 1].seq_len, A.seq);
        if (A.kmers) atomicAdd((unsigned long long *)&recs[A.rec - 1].n_valid, A.kmers);
    }
    A.seq = 0; A.kmers = 0;
}
__device__ __forceinline__ void recacc_retarget(RecAcc &A, uint32_t rec, DevRec *recs, uint64_t recs_cap) {
    if (threadIdx.x == 0 && A.rec != rec) { recacc_spill(A, recs, recs_cap); A.rec = rec; }
}
__device__ __forceinline__ void recacc_finish(RecAcc &A, DevRec *recs, uint64_t recs_cap, Carry *carry) {
    __syncthreads();
    if (threadIdx.x == 0) {
        recacc_spill(A, recs, recs_cap);
        if (A.tot_seq) atomicAdd((unsigned long long *)&carry->total_bp, A.tot_seq);
        if (A.tot_kmers) atomicAdd((unsigned long long *)&carry->num_kmers, A.tot_kmers);
    }
}

template <typename KT>
struct Walker {
    // parser state
    uint32_t ls, run, rec;
    uint64_t pend;
    KT fwd, rev;
    // tallies for the record being walked, and lane totals
    uint64_t seq_acc, kmer_acc, name_end, seq_tot, kmer_tot;
    // constants
    uint32_t k, top;
    KT mask;
    uint64_t pos0;
    DevRec *recs;
    uint64_t recs_cap;
    RecAcc *acc;                           // LDS

    __device__ __forceinline__ void setup(uint32_t k_, DevRec *recs_, uint64_t recs_cap_, RecAcc *acc_) {
        k = k_; top = 2 * (k_ - 1);
        mask = (KT)((k_ >= sizeof(KT) * 4) ? ~(KT)0 : (((KT)1 << (2 * k_)) - 1));
        recs = recs_; recs_cap = recs_cap_; acc = acc_;
        seq_tot = 0; kmer_tot = 0;
    }

    // start of a piece: ls_in / st2 are this lane's exact incoming states
    __device__ __forceinline__ void begin(uint32_t ls_in, const L2 &st2, uint64_t pos_first_byte) {
        ls = ls_in; pend = st2.p_tail; run = l2_len(st2); rec = st2.rec;
        fwd = (KT)st2.bits; rev = 0;
        for (uint32_t i = 0; i < run; i++) {           // reverse-complement value of the carried bases
            uint32_t b = (st2.bits >> (2 * (run - 1 - i))) & 3u;
            rev = (rev >> 2) | ((KT)(3u - b) << top);
        }
        seq_acc = 0; kmer_acc = 0; name_end = 0;
        pos0 = pos_first_byte;
    }

    // May be called by any subset of lanes (a header opens mid-piece).
    __device__ __forceinline__ void flush_rec() {
        if (rec && rec <= recs_cap) {
            if (rec == acc->rec) {
                if (seq_acc) atomicAdd(&acc->seq, (unsigned long long)seq_acc);
                if (kmer_acc) atomicAdd(&acc->kmers, (unsigned long long)kmer_acc);
            } else {
                if (seq_acc) atomicAdd((unsigned long long *)&recs[rec - 1].seq_len, (unsigned long long)seq_acc);
                if (kmer_acc) atomicAdd((unsigned long long *)&recs[rec - 1].n_valid, (unsigned long long)kmer_acc);
            }
            if (name_end) atomicMax((unsigned long long *)&recs[rec - 1].name_end, (unsigned long long)name_end);
        }
        if (rec) { seq_tot += seq_acc; kmer_tot += kmer_acc; }   // text before the first header belongs to no record
        seq_acc = 0; kmer_acc = 0; name_end = 0;
    }

    // End of a piece, ALL lanes of the wave: lanes inside the workgroup's current record are summed
    // across the wave first (one LDS atomic per wave); the rest take the general path.
    __device__ __forceinline__ void flush_rec_wave() {
        const bool common = rec != 0 && rec == acc->rec && rec <= recs_cap;
        unsigned long long s = common ? seq_acc : 0ull, n = common ? kmer_acc : 0ull;
        if (common) { seq_tot += seq_acc; kmer_tot += kmer_acc; seq_acc = 0; kmer_acc = 0; }
        for (int d = 32; d; d >>= 1) { s += __shfl_down(s, d, 64); n += __shfl_down(n, d, 64); }
        if ((threadIdx.x & 63) == 0) {
            if (s) atomicAdd(&acc->seq, s);
            if (n) atomicAdd(&acc->kmers, n);
        }
        if (seq_acc | kmer_acc | name_end) flush_rec();
    }

    // A whole CLEAN piece (sequence characters and line terminators only, no pending blanks, not
    // inside a header): the line/record machinery drops out and the loop is kept deliberately lean --
    // 32-bit counters, integer flags, 16 bytes per iteration with constant shifts.  `sink(has, canon)`
    // is called once per byte slot in wave-uniform control flow.
    template <class Sink>
    __device__ __forceinline__ void walk_clean(const uint8_t *lds, uint32_t nb, Sink &&sink) {
        const uint4 *mine = reinterpret_cast<const uint4 *>(lds + threadIdx.x * LDS_STRIDE);
        KT f = fwd, r = rev;
        uint32_t n_seq = 0, n_kmer = 0;
        // k is wanted as a vector operand in every step (select against a lane mask); pin one copy in a
        // VGPR instead of letting the compiler re-create it from a spilled scalar each time
        uint32_t kk;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kk) : "s"(k));
        const bool live = rec != 0;
        int need = (int)(kk - run);                                      // bases still missing for a full window (<= 0: none)
        // 16 bytes per ds_read_b128, fetched one iteration ahead; the 16 byte steps are unrolled with
        // constant shifts (a rolled byte loop made the compiler issue one LDS read + full wait per byte).
        // Bytes past the end of the stream were staged as 0 (stage_chunk) and behave like terminators here,
        // so `nb` is not consulted.
        (void)nb;
        uint4 nxt = mine[0];
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)PIECE / 16u; q++) {
            const uint4 cur = nxt;
            if (q + 1u < (uint32_t)PIECE / 16u) nxt = mine[q + 1u];
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll 16
            for (uint32_t j = 0; j < 16u; j++) {
                const uint32_t c = (w[j >> 2] >> (8u * (j & 3u))) & 0xffu;
                const bool seq = c > 13u;                                // clean piece: anything but \n / \r (/ 0 fill) is sequence text
                // bits 1-2 of the letter pick the 2-bit code (A 0, C 1, T 3, G 2) and the letter it must be
                const uint32_t idx2 = c & 6u;
                const uint32_t code = (0xB4u >> idx2) & 3u;
                const uint32_t expect = (0x47544341u >> (idx2 << 2)) & 0xffu;       // 'A' 'C' 'T' 'G'
                const bool valid = (c & 0xDFu) == expect;                // either case; everything else is no base
                const KT nf = (KT)(((f << 2) | (KT)code) & mask);        // indexer.py:149
                const KT nr = (KT)((r >> 2) | ((KT)(3u ^ code) << top)); // indexer.py:150
                f = valid ? nf : f;
                r = valid ? nr : r;
                const int fewer = need - (valid ? 1 : 0);                // may run below zero; a piece is far too short to wrap
                need = (seq & !valid) ? (int)kk : fewer;                 // a non-base restarts the window, a terminator holds it
                const bool has = valid & live & (need <= 0);
                n_seq += seq ? 1u : 0u;
                n_kmer += has ? 1u : 0u;
                sink(has, (KT)(f < r ? f : r));
            }
        }
        const uint32_t rn = need <= 0 ? kk : kk - (uint32_t)need;
        fwd = f; rev = r; run = rn;
        seq_acc += n_seq; kmer_acc += n_kmer;
    }

    // One byte.  Returns true when a valid window ends here; canon = min(fwd, rev) (indexer.py:341).
    __device__ __forceinline__ bool step(uint32_t i, uint32_t c, bool act, KT &canon) {
        const bool term = is_term(c), ws = is_ws(c), gt = c == '>';
        const bool at_start = ls == LS_START, in_seq = ls == LS_SEQ;
        const bool hdr_start = act && at_start && !ws && gt;
        const bool seqchar = act && !ws && (in_seq || (at_start && !gt));
        const bool t = act && term;
        if (hdr_start) {                                           // indexer.py:66-82: a new record opens
            flush_rec();
            rec++;
            if (rec <= recs_cap) recs[rec - 1].name_off = pos0 + i + 1;
            name_end = pos0 + i + 1;
            run = 0;
        } else if (act && !ws && ls == LS_HEADER) {
            name_end = pos0 + i + 1;                               // header text extent after strip()
        }
        if (seqchar && pend) { seq_acc += pend; run = 0; pend = 0; }   // blanks were interior: each maps to None
        pend = t ? 0ull : pend + ((act && ws && !term && in_seq) ? 1ull : 0ull);
        ls = t ? (uint32_t)LS_START : hdr_start ? (uint32_t)LS_HEADER : seqchar ? (uint32_t)LS_SEQ : ls;
        seq_acc += seqchar ? 1ull : 0ull;                          // indexer.py:77: valid or not
        const uint32_t code = base_code(c);
        const bool valid = seqchar && code < 4u;
        const KT nf = (KT)(((fwd << 2) | (KT)(code & 3u)) & mask);                 // indexer.py:149
        const KT nr = (KT)((rev >> 2) | ((KT)(3u - (code & 3u)) << top));          // indexer.py:150
        fwd = valid ? nf : fwd;
        rev = valid ? nr : rev;
        run = valid ? (run < k ? run + 1 : run) : (seqchar ? 0u : run);
        const bool has = valid && run == k && rec != 0;            // text before the first header is dropped
        kmer_acc += has ? 1ull : 0ull;
        canon = fwd < rev ? fwd : rev;
        return has;
    }

    // after the last piece, all lanes: wave-reduce the lane totals into the workgroup accumulator
    __device__ __forceinline__ void finish() {
        for (int d = 32; d; d >>= 1) {
            seq_tot += __shfl_down((unsigned long long)seq_tot, d, 64);
            kmer_tot += __shfl_down((unsigned long long)kmer_tot, d, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            if (seq_tot) atomicAdd(&acc->tot_seq, (unsigned long long)seq_tot);
            if (kmer_tot) atomicAdd(&acc->tot_kmers, (unsigned long long)kmer_tot);
        }
    }
};

}  // namespace pk
