// kmer_walk.h -- the per-lane FASTA walker of the squeeze pass (kmer_pack.hip).
//
// One SeqWalker holds the exact parser state at a lane's first byte (from the L1/L2 scans) and advances
// it one byte per step(): line state, pending whitespace, the length of the current run of valid bases
// and the per-record tallies the reference keeps (seq_len, valid windows, header text extent;
// indexer.py:75-95,349-351).  It does not form k-mers: every valid base of a record is handed on as a
// 2-bit code plus a "restart" flag (the run of valid bases begins anew at this base: a record opened, or
// a character that maps to None came before it, indexer.py:36-41,144), and the k-mer values are
// assembled from that packed stream by kmer_fuse.hip.  step() is written branch-light -- predicates and
// selects for everything that happens on most bytes, real branches only for the rare events (a header
// opens, interior whitespace resolves).
#pragma once
#include "fasta_fsm.h"
#include "pk_kernels.h"

namespace pk {

// Per-workgroup accumulator for the records of the chunk being walked.  A 16 KiB chunk usually lies in one record,
// a read set puts a dozen in it: lanes add their tallies to the LDS entry of their record (records of a chunk are
// numbered consecutively from the one the chunk starts in), and once per chunk the entries are written to HBM --
// one atomic pair per record and chunk.  Without it every lane hit the DevRec with global atomics: 25 M same-address
// atomics on an 800 Mbp genome (~100 ms), and still two per lane and piece on a read set (~2 ms).
constexpr uint32_t RECACC_SLOTS = 64;
struct RecAcc {
    uint32_t rec0;                                        // 1-based record of slot 0 (0: the chunk starts before the first header)
    unsigned long long seq[RECACC_SLOTS], kmers[RECACC_SLOTS];   // pending sums for records rec0 .. rec0 + SLOTS - 1
    unsigned long long tot_seq, tot_kmers;                // stream totals gathered by this workgroup
};

__device__ __forceinline__ void recacc_init(RecAcc &A) {
    for (uint32_t i = threadIdx.x; i < RECACC_SLOTS; i += blockDim.x) { A.seq[i] = 0; A.kmers[i] = 0; }
    if (threadIdx.x == 0) { A.rec0 = 0; A.tot_seq = 0; A.tot_kmers = 0; }
}
// all threads, with the workgroup's tallies of the chunk complete (after a barrier) and a barrier before the entries are
// used again: write the pending sums out.  rec0 = the record of slot 0 as read while the chunk was being walked (thread 0
// may already be setting A.rec0 for the next chunk).
__device__ __forceinline__ void recacc_spill(RecAcc &A, uint32_t rec0, DevRec *recs, uint64_t recs_cap) {
    if (threadIdx.x < RECACC_SLOTS) {
        const uint64_t rec = (uint64_t)rec0 + threadIdx.x;
        const unsigned long long s = A.seq[threadIdx.x], n = A.kmers[threadIdx.x];
        if (rec && rec <= recs_cap) {
            if (s) atomicAdd((unsigned long long *)&recs[rec - 1].seq_len, s);
            if (n) atomicAdd((unsigned long long *)&recs[rec - 1].n_valid, n);
        }
        A.seq[threadIdx.x] = 0; A.kmers[threadIdx.x] = 0;
    }
}
__device__ __forceinline__ void recacc_finish(RecAcc &A, DevRec *recs, uint64_t recs_cap, Carry *carry) {
    __syncthreads();
    recacc_spill(A, A.rec0, recs, recs_cap);
    if (threadIdx.x == 0) {
        if (A.tot_seq) atomicAdd((unsigned long long *)&carry->total_bp, A.tot_seq);
        if (A.tot_kmers) atomicAdd((unsigned long long *)&carry->num_kmers, A.tot_kmers);
    }
}

// What one lane hands to the packed stream for its 64-byte piece: the valid bases of live records in
// text order, 2 bits each from bit 0 up (A 0, C 1, G 2, T 3 -- CONV, indexer.py:36-41), one restart
// bit per base, and how many there are (<= 64).
struct PieceBases {
    unsigned long long code_lo, code_hi;   // bases 0-31, 32-63
    unsigned long long restart;
    uint32_t n;
    __device__ __forceinline__ void clear() { code_lo = 0; code_hi = 0; restart = 0; n = 0; }
    __device__ __forceinline__ void push(bool take, uint32_t code, bool rst) {
        const uint32_t sh = (2u * n) & 63u;
        const unsigned long long v = take ? ((unsigned long long)code << sh) : 0ull;
        if (n < 32u) code_lo |= v; else code_hi |= v;
        restart |= (take && rst) ? (1ull << (n & 63u)) : 0ull;
        n += take ? 1u : 0u;
    }
};

struct SeqWalker {
    // parser state
    uint32_t ls, run, rec;
    uint64_t pend;
    // tallies for the record being walked, and lane totals
    uint64_t seq_acc, kmer_acc, name_end, seq_tot, kmer_tot;
    // constants
    uint32_t k;
    uint64_t pos0;
    DevRec *recs;
    uint64_t recs_cap;
    RecAcc *acc;                           // LDS

    __device__ __forceinline__ void setup(uint32_t k_, DevRec *recs_, uint64_t recs_cap_, RecAcc *acc_) {
        k = k_; recs = recs_; recs_cap = recs_cap_; acc = acc_;
        seq_tot = 0; kmer_tot = 0;
    }

    // start of a piece: ls_in / st2 are this lane's exact incoming states
    __device__ __forceinline__ void begin(uint32_t ls_in, const L2 &st2, uint64_t pos_first_byte) {
        ls = ls_in; pend = st2.p_tail; run = l2_len(st2); rec = st2.rec;
        seq_acc = 0; kmer_acc = 0; name_end = 0;
        pos0 = pos_first_byte;
    }

    // May be called by any subset of lanes (a header opens mid-piece).
    __device__ __forceinline__ void flush_rec() {
        if (rec && rec <= recs_cap) {
            const uint32_t slot = rec - acc->rec0;                       // records of a chunk count up from the one it starts in
            if (rec >= acc->rec0 && slot < RECACC_SLOTS) {
                if (seq_acc) atomicAdd(&acc->seq[slot], (unsigned long long)seq_acc);
                if (kmer_acc) atomicAdd(&acc->kmers[slot], (unsigned long long)kmer_acc);
            } else {
                if (seq_acc) atomicAdd((unsigned long long *)&recs[rec - 1].seq_len, (unsigned long long)seq_acc);
                if (kmer_acc) atomicAdd((unsigned long long *)&recs[rec - 1].n_valid, (unsigned long long)kmer_acc);
            }
            if (name_end) atomicMax((unsigned long long *)&recs[rec - 1].name_end, (unsigned long long)name_end);
        }
        if (rec) { seq_tot += seq_acc; kmer_tot += kmer_acc; }   // text before the first header belongs to no record
        seq_acc = 0; kmer_acc = 0; name_end = 0;
    }

    // End of a piece, ALL lanes of the wave: when the whole wave sits in one record (the usual case) the tallies are
    // summed across the wave first (one LDS atomic pair per wave); otherwise every lane takes the general path.
    __device__ __forceinline__ void flush_rec_wave() {
        const uint32_t rec_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec);
        if (__all(rec == rec_first)) {
            unsigned long long s = seq_acc, n = kmer_acc;
            for (int d = 32; d; d >>= 1) { s += __shfl_down(s, d, 64); n += __shfl_down(n, d, 64); }
            const bool lead = (threadIdx.x & 63) == 0;
            seq_acc = lead ? s : 0ull; kmer_acc = lead ? n : 0ull;      // lane 0 flushes the wave's sums; name_end stays per lane
        }
        if (seq_acc | kmer_acc | name_end) flush_rec();
    }

    // One byte.  Returns true when the byte is a valid base of a record (it joins the packed stream);
    // `code` is its 2-bit value and `restart` tells whether the run of valid bases starts anew with it.
    __device__ __forceinline__ bool step(uint32_t i, uint32_t c, bool act, uint32_t &code, bool &restart) {
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
        const uint32_t cd = base_code(c);
        const bool valid = seqchar && cd < 4u;
        restart = run == 0u;
        run = valid ? (run < k ? run + 1 : run) : (seqchar ? 0u : run);
        const bool live = rec != 0;                                // text before the first header is dropped (indexer.py:80-82)
        kmer_acc += (valid && run == k && live) ? 1ull : 0ull;     // a window without None ends here (indexer.py:144)
        code = cd & 3u;
        return valid && live;
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
