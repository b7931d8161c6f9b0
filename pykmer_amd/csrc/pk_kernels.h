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
void launch_scan_l1(const L1 *in, uint32_t n_chunks, Carry *carry, L1 *out, L1 *tile_ws, hipStream_t s);
void launch_chunk_l2(const uint8_t *fasta, uint64_t n, const L1 *st1, L2 *chunk_l2, LaneState *lane_state, uint32_t n_chunks, uint32_t k,
                     hipStream_t s);
void launch_scan_l2(const L2 *in, uint32_t n_chunks, Carry *carry, L2 *out, L2 *tile_ws, uint32_t k, hipStream_t s);
void launch_count(const uint8_t *fasta, uint64_t n, uint64_t stream_off, const LaneState *lane_state, const L2 *st2, uint32_t n_chunks,
                  uint32_t k, uint32_t *table32, DevRec *recs, uint64_t recs_cap, Carry *carry, hipStream_t s);
void launch_finalize(const uint32_t *table32, uint8_t *table8, uint64_t n, unsigned long long *hist, hipStream_t s);
void launch_hist8(const uint8_t *table8, uint64_t n, unsigned long long *hist, hipStream_t s);
void launch_clamp32(uint32_t *table32, uint64_t n, hipStream_t s);

// kmer_part.hip -- partitioned table update (version 2)
struct PartPlan {
    uint32_t k, addr_bits;   // 2k
    uint32_t fb_bits;        // address bits inside a final bucket (<= 16)
    uint32_t b1, b2, B1, B2; // level-1 / level-2 digit widths and bucket counts
    uint32_t n_chunks;       // 16 KiB FASTA chunks in this feed
    uint32_t n_wg0, G;       // walk workgroups and chunks per workgroup (rows of the level-1 histogram)
    uint64_t R2;             // records per level-2 workgroup
    uint32_t n_wg2_max;      // upper bound on level-2 workgroups
    uint32_t dbg;            // PK_DEBUG_WALK ablation bits (timing diagnostics only; 0 in production)
};
struct PartWorkspace {       // byte offsets into one device allocation
    size_t flat, cnt, hist1, rowoff1, bucket_base, wg2_start, col_tot, final_start, out1, hist2, rowoff2, out2, side, side_n, bucket_hist, fine_rows, fine_tot, cursor;
    uint64_t side_cap;
};
PartPlan make_part_plan(uint32_t k, uint64_t n_bytes);
size_t part_workspace_bytes(const PartPlan &pl, uint64_t n_bytes, PartWorkspace *lay);
int launch_partitioned(const uint8_t *fasta, uint64_t n, uint64_t stream_off, const LaneState *lane_state, const L2 *st2, const PartPlan &pl,
                       uint8_t *ws, const PartWorkspace &lay, uint8_t *table8, DevRec *recs, uint64_t recs_cap, Carry *carry,
                       hipStream_t s, hipEvent_t ev_walk_end, hipEvent_t ev_part_end, bool fresh, unsigned long long *hist);

// gram_scan.hip
// tables: device array of N device pointers, each n_slice bytes (16-byte aligned).  pair: device N*N u64,
// zeroed by the launcher when zero_first; [i][i] += total_i, [i][j] (i<j) += shared_ij.
int launch_gram(const uint8_t *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count,
                unsigned long long *dev_pair, bool zero_first, hipStream_t s);

}  // namespace pk
