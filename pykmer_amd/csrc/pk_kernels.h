// pk_kernels.h -- device-side structs and kernel launchers shared between the .hip files and pk_api.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fasta_fsm.h"

namespace pk {

// One FASTA record while it is being accumulated in HBM (atomics from many lanes).
struct DevRec {
    uint64_t name_off;  // first byte after '>'
    uint64_t name_end;  // one past the last non-blank header byte (atomicMax)
    uint64_t seq_len;   // atomicAdd
    uint64_t n_valid;   // atomicAdd
};

// Parser state carried between feeds + running totals; lives in device memory so that consecutive
// feeds need no host round trip for it.
struct Carry {
    L2 l2;               // state after the last byte fed so far
    L1 l1;
    uint32_t pad;
    uint64_t n_recs;     // = l2.rec, mirrored for the host read-back
    uint64_t total_bp;   // sum of seq_len over all records
    uint64_t num_kmers;  // valid windows counted
};

void launch_chunk_l1(const uint8_t *fasta, uint64_t n, L1 *chunk_l1, uint32_t n_chunks, hipStream_t s);
void launch_scan_l1(const L1 *in, uint32_t n_chunks, Carry *carry, L1 *out, L1 *tile_ws, uint32_t *zero_words, uint32_t n_zero, hipStream_t s);
void launch_chunk_l2(const uint8_t *fasta, uint64_t n, const L1 *st1, L2 *chunk_l2, LaneState *lane_state, PiecePack *packs, uint32_t *chunk_odd, uint32_t n_chunks,
                     uint32_t k, hipStream_t s);
void launch_scan_l2(const L2 *in, uint32_t n_chunks, Carry *carry, L2 *out, L2 *tile_ws, uint32_t k, hipStream_t s);
void launch_hist8(const uint8_t *table8, uint64_t n, unsigned long long *hist, hipStream_t s);

// kmer_pack.hip -- the packed stream: one slot per 16 KiB text chunk
constexpr uint32_t SLOT_CODE_WORDS = 1024;   // 16384 bases x 2 bits
constexpr uint32_t SLOT_RST_WORDS = 512;     // 16384 restart bits
void launch_squeeze(const uint8_t *fasta, uint64_t n, uint64_t stream_off, const LaneState *lane_state, const PiecePack *packs, const L2 *st2,
                    const uint32_t *chunk_odd, uint32_t k, uint32_t n_chunks, uint32_t n_wg, uint32_t chunks_per_wg, uint32_t *codes, uint32_t *restarts, uint32_t *n_bases,
                    DevRec *recs, uint64_t recs_cap, Carry *carry, uint32_t *flags, hipStream_t s);

// kmer_fuse.hip / kmer_part.hip -- partitioned table update
struct PartPlan {
    uint32_t k, addr_bits;   // addr_bits = 2k - slice_bits: address bits of the table this indexer holds
    uint32_t slice_bits, slice_index;   // k-mers whose top slice_bits address bits differ from slice_index are not this table's
    uint32_t fb_bits;        // address bits inside a final bucket (<= 16)
    uint32_t b1, b2, B1, B2; // level-1 / level-2 digit widths and bucket counts
    uint32_t n_chunks;       // 16 KiB FASTA chunks in this feed
    uint32_t n_wg0, G;       // persistent workgroups of the squeeze kernel and chunks per workgroup
    uint32_t n_wg1, G1;      // the same for the level-1 sort (k_walk_sort)
    uint64_t R2;             // records per level-2 workgroup
    uint32_t n_wg2_max;      // upper bound on level-2 workgroups
    uint32_t sample_stride;  // every how-manieth slot is tallied to size the buckets (1 = all: exact)
    uint32_t n_tally;        // what the sampling launch tallies: B1 * B2 final buckets (both levels are then laid out from
                             // the estimate), or just the B1 level-1 buckets (one level, or too many final buckets: k = 17)
    uint32_t sample2;        // 1: the final buckets (too many to tally while sampling slots: 2^15 < B1 * B2 <= 2^18) are sized from a
                             // sample of the level-1 RECORDS, after the level-1 sort, and level 2 claims its room like the others
    uint64_t capacity1;      // record slots for all level-1 buckets together (the dump area starts there)
    uint64_t capacity2;      // the same for the final buckets, where they are laid out from the estimate
};
constexpr uint32_t COUNT_WGS = 256;   // workgroups (= tally rows) of the sampling launch
struct PartWorkspace {       // byte offsets into one device allocation
    size_t codes, restarts, n_bases, tally_rows, tally_tot, bucket_base, bucket_end, compact_base, cursor1, cap_end, wg2_start, final_start, cursor2,
        cap2_end, out1, hist2, rowoff2, out2, side, side_n;
    uint64_t side_cap;
};
PartPlan make_part_plan(uint32_t k, uint64_t n_bytes, uint32_t slice_bits, uint32_t slice_index);
size_t part_workspace_bytes(const PartPlan &pl, uint64_t n_bytes, PartWorkspace *lay);
void part_set_attributes();
void fuse_set_attributes();
void launch_provision(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl,
                      uint32_t stride, uint32_t *tally_rows, uint32_t *tally_tot, uint32_t *bucket_base, uint32_t *cursor1,
                      uint32_t *cap_end, uint32_t *final_start, uint32_t *cursor2, uint32_t *cap2_end, uint32_t *flags, hipStream_t s);
void launch_walk_sort(const uint32_t *codes, const uint32_t *restarts, const uint32_t *n_bases, const L2 *st2, const PartPlan &pl, void *out1,
                      uint32_t *cursor1, const uint32_t *cap_end, uint32_t *flags, const uint32_t *bucket_base, uint32_t *bucket_end,
                      uint32_t *compact_base, uint32_t *wg2_start, unsigned long long *side, unsigned long long *side_n, uint64_t side_cap,
                      hipStream_t s);
// `armed`: the side-list length and the flags word were already zeroed on the stream (launch_scan_l1 does it for the first
// attempt of a feed); a repeat after an overflow zeroes them itself.
int launch_partitioned(const L2 *st2, uint64_t n_bytes, const PartPlan &pl, uint32_t stride, uint8_t *ws, const PartWorkspace &lay, uint8_t *table8,
                       hipStream_t s, hipEvent_t ev_sort_begin, hipEvent_t ev_sort_end, hipEvent_t ev_part_end, bool fresh,
                       unsigned long long *hist, unsigned long long *hist_replicas, bool armed);
constexpr uint32_t HIST_REPLICAS = 64;   // copies of the 256-bin histogram change the bucket-count workgroups add into (zeroed by their reader, k_apply_side)
#ifdef PK_PHASE_PROF
constexpr uint32_t PART_FLAG_WORDS = 60;  // + 5 + 5 + 8 u64 phase-cycle counters (experiment builds)
#else
constexpr uint32_t PART_FLAG_WORDS = 6;   // side_n (u64) + flags[4], zeroed together
#endif

// gram_scan.hip
// tables: device array of N device pointers, each n_slice bytes (16-byte aligned).  pair: device N*N u64,
// zeroed by the launcher when zero_first; [i][i] += total_i, [i][j] (i<j) += shared_ij.
int launch_gram(const uint8_t *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count,
                unsigned long long *dev_pair, bool zero_first, hipStream_t s);

// Several validity windows in one pass over the tables (threshold sweeps): at most gram_windows_per_pass(N) windows
// per launch (0: this N is served by one launch_gram per window), sorted by min_count; window w's tallies are added to
// block out_index[w] (< 256) of dev_pair.
int gram_windows_per_pass(int N);
int launch_gram_windows(const uint8_t *const *dev_tables, int N, uint64_t n_slice, const int *min_counts, const int *max_counts,
                        const int *out_index, int W, unsigned long long *dev_pair, hipStream_t s);

}  // namespace pk
