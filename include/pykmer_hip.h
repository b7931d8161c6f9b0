/*
 * pykmer_hip.h -- C-ABI of libpykmer_hip.so, the MI355X (gfx950) engine behind pykmer's two hot loops.
 *
 * The reference (sauloal/pykmer) is monolithic Python with no FFI; the boundary sits where it hands
 * work to its hot loops.  Each entry point names the reference code it replaces
 * (file:line into the reference repository).  Plain pointers and sizes only; the caller owns every
 * host buffer; the library owns device memory.  All functions return PK_OK (0) or a negative
 * PK_ERR_* code; pk_last_error() returns the message of the calling thread's last failure.
 * Calls are blocking and may be issued concurrently for different devices.
 */
#ifndef PYKMER_HIP_H
#define PYKMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_ABI_VERSION 3

enum {
    PK_OK = 0,
    PK_ERR_ARG = -1,      /* even / non-positive / unsupported k, bad min/max count, null pointer */
    PK_ERR_HIP = -2,      /* HIP runtime failure (no device, out of memory, launch error)        */
    PK_ERR_RECS_CAP = -3, /* more records than recs_cap; *n_recs_out holds the number needed      */
    PK_ERR_STATE = -4     /* handle used out of order                                            */
};

/* One FASTA record as the reference's parse_fasta yields it (indexer.py:45-99). The caller slices the
 * header text out of its own buffer and drops records with n_valid_kmers == 0 to reproduce the
 * `chromosomes` list (indexer.py:349-351). Offsets are relative to the first byte ever fed. */
typedef struct pk_record {
    uint64_t name_off;      /* first byte after '>'                                             */
    uint64_t name_len;      /* header length after strip() (indexer.py:56,80)                   */
    uint64_t seq_len;       /* stripped sequence characters, valid or not (indexer.py:77,93)    */
    uint64_t n_valid_kmers; /* windows without a None (indexer.py:144)                          */
} pk_record;

int pk_version(void);
int pk_last_error(char *buf, size_t n);
int pk_device_count(void);
/* Optional: brings the HIP runtime up on `device` and loads the kernels, so that a command-line host can do this in a
 * thread while it still parses arguments and maps its input (about 0.2 s that pk_indexer_create / pk_gram* would
 * otherwise spend).  Has no reference counterpart: indexer.py / merger.py start no device. */
int pk_warm(int device);

/* ---- plain device buffers, so a host in any language can stage tables in HBM one at a time (a k=17
 * merge holds N x 16 GiB on the device, never on the host) and hand slices to pk_gram_device_partial. */
int pk_dev_alloc(void **dev_out, uint64_t n_bytes, int device);
int pk_dev_free(void *dev, int device);
int pk_dev_upload(void *dev_dst, const void *host_src, uint64_t n_bytes, int device);
int pk_dev_download(void *host_dst, const void *dev_src, uint64_t n_bytes, int device);
/* Free / total HBM on `device` in bytes: lets the host size how many table slices it stages beside each other. */
int pk_dev_mem_info(uint64_t *free_out, uint64_t *total_out, int device);

/* ---- indexer: replaces gen_kmers + canonical min + process_kmers (indexer.py:130-160, 341, 162-297)
 * and the parser that feeds them (indexer.py:45-99).  k must be odd (tools.py:165-167); 1 <= k <= 17 in one table,
 * k = 19 and 21 through pk_indexer_create_slice.
 *
 * fasta       uncompressed FASTA text (what gzip.open(...,'rt') would hand the reference)
 * table_out   4^k bytes (host); receives table[a] = min(255, #canonical k-mers with value a): the
 *             exact content of the reference's .kin file (tools.py:333-341, indexer.py:262)
 * hist256_out nullable; 256 counters, hist256[v] = #{a : table[a] == v}.  Header.update_stats
 *             (tools.py:246-263) follows from it: hist = hist256[1:], vals_sum = sum v*hist256[v] ...
 */
int pk_count_fasta(const uint8_t *fasta, uint64_t n_bytes, int k, uint8_t *table_out,
                   uint64_t *num_kmers_out, uint64_t *total_bp_out, uint64_t hist256_out[256],
                   pk_record *recs_out, uint64_t recs_cap, uint64_t *n_recs_out, int device);

/* pk_count_fasta keeps its indexer (table + workspace in HBM) for the next call with the same k and device;
 * this frees it. */
int pk_count_release(void);

/* Streaming / device-resident form of the same path (inputs larger than host RAM, inputs that
 * already live in HBM, repeated timing).  One indexer owns one 4^k table in HBM on one device. */
typedef struct pk_indexer pk_indexer;
int pk_indexer_create(pk_indexer **out, int k, int device);
/* Address-range-sharded form (every shard streams the whole text and keeps the canonical k-mers of its own
 * address range; no reduction, the shards' tables concatenate to the 4^k-byte .kin).  n_slices is a power of two; the
 * indexer's table is bytes [slice_index, slice_index + 1) * 4^k / n_slices of the .kin image, and a slice may hold at
 * most 2^34 addresses -- so k = 19 (256 GiB, never run by the reference: README.md:51-52) is 16 slices of 16 GiB,
 * on one GPU after another or spread over several.  num_kmers / total_bp / records describe the whole input whatever the
 * slice; hist256 describes the slice. */
int pk_indexer_create_slice(pk_indexer **out, int k, int device, int slice_index, int n_slices);
int pk_indexer_reset(pk_indexer *ix);                       /* zero the table, forget parser state  */
/* Feed the next n_bytes of the FASTA text.  Chunks may split lines, records and k-mers anywhere.   */
int pk_indexer_feed(pk_indexer *ix, const uint8_t *host_fasta, uint64_t n_bytes);
/* Same, but the bytes already sit in device memory on the indexer's device (16-byte aligned).  Work runs on
 * the indexer's own stream; the call returns when the feed has been counted.  Any size: the library cuts the text into
 * pieces of at most 2 GiB (record positions inside one piece are 32-bit). */
int pk_indexer_feed_device(pk_indexer *ix, const void *dev_fasta, uint64_t n_bytes);
/* Close the last record and report the totals.  hist256_out[v] = number of table entries equal to v
 * (the histogram is kept in HBM while counting, so this makes no pass over the table).              */
int pk_indexer_finish(pk_indexer *ix, uint64_t *num_kmers_out, uint64_t *total_bp_out,
                      uint64_t hist256_out[256], uint64_t *n_recs_out);
int pk_indexer_records(pk_indexer *ix, pk_record *recs_out, uint64_t recs_cap);
int pk_indexer_table_to_host(pk_indexer *ix, uint8_t *table_out);
/* Table bytes [offset, offset + n_bytes) to the host: lets the caller take the .kin image in slices and hash / write
 * slice i while slice i+1 crosses PCIe (tools.py:280,283 hash the whole file afterwards). */
int pk_indexer_table_slice_to_host(pk_indexer *ix, uint8_t *dst, uint64_t offset, uint64_t n_bytes);
/* Device pointer of the finished u8 table (valid until reset/destroy) -- lets a merge run on tables
 * that never left HBM. */
int pk_indexer_table_device(pk_indexer *ix, const void **dev_table_out);
/* Copy table bytes [offset, offset+n_bytes) into another device buffer on the same device (keeps an
 * address-range slice for a sharded merge while the indexer goes on to the next sample). */
int pk_indexer_table_slice_to_device(pk_indexer *ix, void *dev_dst, uint64_t offset, uint64_t n_bytes);
/* Seconds spent since the last reset per stage, measured with HIP events on the indexer's stream:
 * [0] structure scans, [1] squeeze pass (text -> packed bases + record tallies), [2] finish, [3] reset,
 * [4] feeds (as a double), [5] everything between squeeze and bucket count (bucket layout, fused k-mer
 * assembly + level-1 sort, level 2), [6] bucket count + histogram rows + side list, [7] of [5]: the fused
 * k_walk_sort kernel alone, [8] how many feeds had to be laid out a second time with exact bucket sizes, [9] how many final buckets of a
 * sparse table were counted a second time because a byte counter wrapped (k_bucket_count_bytes). */
int pk_indexer_timings(pk_indexer *ix, double out[10]);
void pk_indexer_destroy(pk_indexer *ix);

/* ---- stats: replaces Header.update_stats (tools.py:246-263) on a host table of n bytes. */
int pk_table_stats(const uint8_t *table, uint64_t n, uint64_t hist256_out[256], int device);

/* ---- merger: replaces Header.calculate_distance (tools.py:439-493) for every pair of merger.merge
 * (merger.py:139-176) in ONE pass over the N tables.
 *
 * tables      N host pointers to n-byte tables (n = 4^k, equal sizes: tools.py:444)
 * min/max     validity window, 1 <= min, max <= 255 (merger.py:90-91)
 * matrix_out  N*N*3 u64 row-major: [i][j] = (total_i, total_j, shared_ij) for i != j
 *             (merger.py:175-176); the diagonal, which the reference leaves unassigned
 *             (merger.py:136), is written as (0,0,0).
 * devices     n_devices device ordinals; the address range is split evenly across them and the
 *             partial matrices are summed on the host (single-process form).  The multi-process
 *             RCCL form uses pk_gram_device_partial on each rank's slice.
 */
int pk_gram(const uint8_t *const *tables, int N, uint64_t n, int min_count, int max_count,
            uint64_t *matrix_out, const int *devices, int n_devices);

/* Device-resident slice form: dev_tables[i] points at n_slice bytes of table i in HBM on `device`.
 * pair_out (host) receives N*N u64: [i][i] = total_i over the slice, [i][j] = shared_ij.  Summing
 * pair_out over disjoint slices (ranks) and expanding with pk_gram_expand gives matrix_out.
 * dev_pair_out (nullable, device, N*N u64) receives the same numbers in HBM for an RCCL all-reduce. */
int pk_gram_device_partial(const void *const *dev_tables, int N, uint64_t n_slice, int min_count,
                           int max_count, uint64_t *pair_out, void *dev_pair_out, int device,
                           double *kernel_seconds_out);
/* Same scan, but the tallies are ADDED to dev_pair_accum (device, N*N u64, zeroed by the caller before the
 * first slice): a rank whose share of the address range does not fit HBM beside N tables scans it in
 * sub-slices, and the accumulator is what the RCCL all-reduce sums across ranks (merger.py:163-178). */
int pk_gram_device_accumulate(const void *const *dev_tables, int N, uint64_t n_slice, int min_count,
                              int max_count, void *dev_pair_accum, int device, double *kernel_seconds_out);
/* The same for n_windows validity windows at once (threshold sweeps: the reference re-runs the whole merge per
 * --min-count / --max-count, README.md:57-61; the compare itself is tools.py:473-475).  dev_pair_accum holds
 * n_windows x N x N u64, block w for (min_counts[w], max_counts[w]); the staged slices are streamed once per group of up
 * to eight windows (N <= 16; five for N <= 24, three for N <= 32, one beyond), not once per window. */
int pk_gram_device_accumulate_windows(const void *const *dev_tables, int N, uint64_t n_slice, const int *min_counts,
                                      const int *max_counts, int n_windows, void *dev_pair_accum, int device,
                                      double *kernel_seconds_out);
int pk_gram_expand(const uint64_t *pair, int N, uint64_t *matrix_out);

/* ---- BGZF on the host (no device work): the reference reads `.kin.bgz` tables and `.fa.gz` / `.bgz` inputs through one
 * Python gzip.open stream (tools.py:294-305, indexer.py:112-115); bgzip output (README.md:26) is a series of independent
 * gzip members, inflated here block-parallel on native threads straight into the caller's buffer.
 * pk_bgzf_scan: walks the member headers of `src` (no inflation): offset, size and ISIZE of each; PK_ERR_ARG if the
 * bytes are not BGZF; PK_ERR_RECS_CAP if more than `cap` blocks (*n_blocks_out = the count; call again with room).
 * pk_bgzf_inflate: block i (c_off[i], c_size[i]) -> dst + u_off[i], u_off[i + 1] - u_off[i] bytes (u_off has n_blocks + 1
 * entries); the block list is checked against src_bytes (it may come from a `.gzi` file), CRC32 and ISIZE of every
 * block against its payload (SAM spec 4.1). */
int pk_bgzf_scan(const uint8_t *src, uint64_t n_bytes, uint64_t cap, uint64_t *c_off_out, uint64_t *c_size_out,
                 uint64_t *isize_out, uint64_t *n_blocks_out);
int pk_bgzf_inflate(const uint8_t *src, uint64_t src_bytes, const uint64_t *c_off, const uint64_t *c_size,
                    const uint64_t *u_off, uint64_t n_blocks, uint8_t *dst, int threads);
/* The writer (the README's `bgzip -i -I x.gzi -l 9 -c x > x.bgz` step, README.md:26): every block_input (<= 0xff00) bytes
 * of src become one BGZF block, deflated on native threads; dst needs 65536 bytes per block, the blocks end up back to
 * back in its first *total_out bytes (no end-of-file block), c_sizes_out[i] = size of block i (the `.gzi` follows). */
int pk_bgzf_deflate(const uint8_t *src, uint64_t n_bytes, int level, uint32_t block_input, uint8_t *dst, uint64_t dst_cap,
                    uint64_t *c_sizes_out, uint64_t *total_out, int threads);

/* ---- diagnostics (tools/, tests/): no reference counterpart, no effect on any result.
 * pk_diag_occupancy: workgroups per CU the runtime grants kernel `which` (0 k_bucket_count_half, 1 k_bucket_count_bytes,
 * 2 k_bucket_count_half_lean, 3 k_scatter2<claim>); negative = HIP error.
 * pk_diag_plan: the partition plan of ONE feed of n_bytes (0 = the largest piece a feed is cut into) at kmer_len k:
 * out = { largest piece, level-1 record capacity, final-bucket record capacity, level-1 buckets, level-2 digits,
 * address bits per final bucket, 16 KiB chunks, 1 if every record position fits 32 bits }. */
int pk_diag_occupancy(int which);
int pk_diag_plan(int k, uint64_t n_bytes, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* PYKMER_HIP_H */
