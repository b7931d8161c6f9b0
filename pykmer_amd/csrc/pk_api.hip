// pk_api.hip -- host side of the C-ABI declared in include/pykmer_hip.h.
// Owns device memory, streams and events; sequences the kernels of kmer_count.hip / gram_scan.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pykmer_hip.h"
#include "pk_kernels.h"

using namespace pk;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace pk { int set_error(int code, const std::string &msg) { g_err = msg; return code; } }   // for the other translation units

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return fail(PK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

extern "C" int pk_version(void) { return PK_ABI_VERSION; }

extern "C" int pk_last_error(char *buf, size_t n) {
    if (!buf || n == 0) return PK_ERR_ARG;
    snprintf(buf, n, "%s", g_err.c_str());
    return PK_OK;
}

extern "C" int pk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pk_warm(int device) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(nullptr));                                // the context
    pk::part_set_attributes();                               // the code object (first use of a kernel symbol loads it)
    return PK_OK;
}

// tools.py:165-167: k > 0 and odd.  One indexer holds at most 2^34 addresses (a 16 GiB table): that is all of k <= 17; beyond
// (k = 19: 256 GiB, k = 21: 4 TiB -- README.md:51-52 marks both as never run) the address range is cut into 2^slice_bits
// slices and an indexer counts one of them.
static int check_k(int k, int slice_bits = 0, int slice_index = 0) {
    if (k <= 0 || (k % 2) == 0) return fail(PK_ERR_ARG, "kmer_len must be positive and odd (tools.py:165-167), got %d", k);
    if (k > 21) return fail(PK_ERR_ARG, "kmer_len %d not supported by the device path (max 21)", k);
    if (slice_bits < 0 || slice_bits > 2 * k || slice_bits > 16) return fail(PK_ERR_ARG, "bad number of address slices for kmer_len %d", k);
    if (2 * k - slice_bits > 34)
        return fail(PK_ERR_ARG, "kmer_len %d needs a table of 4^%d bytes; one indexer holds 2^34 (16 GiB): count it in %d address slices "
                                "(pk_indexer_create_slice)", k, k, 1 << (2 * k - 34));
    if (slice_index < 0 || slice_index >= (1 << slice_bits)) return fail(PK_ERR_ARG, "slice index %d outside 0..%d", slice_index, (1 << slice_bits) - 1);
    return PK_OK;
}

// ================================================================== host <-> HBM copies =========
// The C-ABI takes plain (pageable) host buffers.  One hipMemcpy from pageable memory is a single thread bouncing
// the bytes through a small pinned buffer; here several host threads each own a pinned bounce buffer (two halves)
// and a stream, so page-touching memcpy and PCIe DMA of different pieces overlap and the link is what limits.
#include <mutex>
#include <thread>
namespace {
constexpr size_t BOUNCE_HALF = 8u << 20;
constexpr int MAX_COPY_THREADS = 16;
struct Bouncer {
    std::mutex mu;
    int threads = 0;
    uint8_t *pinned[MAX_COPY_THREADS] = {};
    hipStream_t stream[MAX_COPY_THREADS] = {};
    hipEvent_t ev[MAX_COPY_THREADS][2] = {};
};
Bouncer g_bounce[64];

int bouncer_for(int device, Bouncer **out) {
    if (device < 0 || device >= 64) return fail(PK_ERR_ARG, "device ordinal %d out of range", device);
    Bouncer &b = g_bounce[device];
    static std::mutex init_mu;                             // several host threads may make their first copy at once
    std::lock_guard<std::mutex> init_lock(init_mu);
    if (!b.threads) {
        const char *env = getenv("PK_COPY_THREADS");
        int t = env ? atoi(env) : 8;
        t = std::max(1, std::min(t, MAX_COPY_THREADS));
        for (int i = 0; i < t; i++) {
            HIPCHK(hipHostMalloc((void **)&b.pinned[i], 2 * BOUNCE_HALF, hipHostMallocDefault));
            HIPCHK(hipStreamCreateWithFlags(&b.stream[i], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&b.ev[i][0], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b.ev[i][1], hipEventDisableTiming));
        }
        b.threads = t;
    }
    *out = &b;
    return PK_OK;
}

// to_device: host -> dev, else dev -> host.  Blocking.
int bounce_copy(void *dev, void *host, size_t n, bool to_device, int device) {
    if (n == 0) return PK_OK;
    HIPCHK(hipSetDevice(device));
    if (n < (4u << 20)) {
        HIPCHK(to_device ? hipMemcpy(dev, host, n, hipMemcpyHostToDevice) : hipMemcpy(host, dev, n, hipMemcpyDeviceToHost));
        return PK_OK;
    }
    Bouncer *b = nullptr;
    int rc = bouncer_for(device, &b);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(b->mu);
    const size_t n_pieces = (n + BOUNCE_HALF - 1) / BOUNCE_HALF;
    const int T = (int)std::min<size_t>((size_t)b->threads, n_pieces);
    std::vector<hipError_t> errs(T, hipSuccess);
    auto work = [&](int t) {
        hipError_t e = hipSetDevice(device);
        size_t pending_off[2] = {0, 0}, pending_len[2] = {0, 0};      // D2H: a half whose DMA is in flight and still has to reach the host buffer
        int h = 0;
        for (size_t p = (size_t)t; p < n_pieces && e == hipSuccess; p += (size_t)T, h ^= 1) {
            const size_t off = p * BOUNCE_HALF, len = std::min(BOUNCE_HALF, n - off);
            uint8_t *half = b->pinned[t] + (size_t)h * BOUNCE_HALF;
            if (to_device) {
                e = hipEventSynchronize(b->ev[t][h]);                  // the DMA that last read this half is done
                if (e != hipSuccess) break;
                memcpy(half, (const uint8_t *)host + off, len);
                e = hipMemcpyAsync((uint8_t *)dev + off, half, len, hipMemcpyHostToDevice, b->stream[t]);
                if (e == hipSuccess) e = hipEventRecord(b->ev[t][h], b->stream[t]);
            } else {
                if (pending_len[h]) {                                  // drain what this half held before reusing it
                    e = hipEventSynchronize(b->ev[t][h]);
                    if (e != hipSuccess) break;
                    memcpy((uint8_t *)host + pending_off[h], half, pending_len[h]);
                }
                e = hipMemcpyAsync(half, (const uint8_t *)dev + off, len, hipMemcpyDeviceToHost, b->stream[t]);
                if (e == hipSuccess) e = hipEventRecord(b->ev[t][h], b->stream[t]);
                pending_off[h] = off; pending_len[h] = len;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream[t]);
        if (!to_device && e == hipSuccess)
            for (int q = 0; q < 2; q++)
                if (pending_len[q]) memcpy((uint8_t *)host + pending_off[q], b->pinned[t] + (size_t)q * BOUNCE_HALF, pending_len[q]);
        errs[t] = e;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++)
        if (errs[t] != hipSuccess) return fail(PK_ERR_HIP, "host <-> device copy failed: %s", hipGetErrorString(errs[t]));
    return PK_OK;
}
}  // namespace

// ================================================================== device buffers =============
extern "C" int pk_dev_alloc(void **dev_out, uint64_t n_bytes, int device) {
    if (!dev_out) return fail(PK_ERR_ARG, "null output pointer");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc(dev_out, n_bytes + 64));            // slack: kernels read whole 16/32-byte words
    return PK_OK;
}
extern "C" int pk_dev_free(void *dev, int device) {
    if (!dev) return PK_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(dev));
    return PK_OK;
}
extern "C" int pk_dev_upload(void *dev_dst, const void *host_src, uint64_t n_bytes, int device) {
    if (n_bytes && (!dev_dst || !host_src)) return fail(PK_ERR_ARG, "null pointer");
    return bounce_copy(dev_dst, const_cast<void *>(host_src), n_bytes, true, device);
}
extern "C" int pk_dev_download(void *host_dst, const void *dev_src, uint64_t n_bytes, int device) {
    if (n_bytes && (!host_dst || !dev_src)) return fail(PK_ERR_ARG, "null pointer");
    return bounce_copy(const_cast<void *>(dev_src), host_dst, n_bytes, false, device);
}

extern "C" int pk_dev_mem_info(uint64_t *free_out, uint64_t *total_out, int device) {
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_out) *free_out = f;
    if (total_out) *total_out = t;
    return PK_OK;
}

// ================================================================== indexer ====================
struct pk_indexer {
    int k = 0, device = 0;
    int slice_bits = 0, slice_index = 0;   // the table holds addresses [slice_index, slice_index + 1) * 4^k / 2^slice_bits
    uint64_t n = 0;                  // table bytes: 4^k / 2^slice_bits
    hipStream_t stream = nullptr;
    uint8_t *table8 = nullptr;       // the .kin image
    // parser state + running totals, and the value histogram, side by side: one copy brings both to the host, one copy resets both
    struct Tail { Carry carry; unsigned long long hist[256]; };
    Tail *tail = nullptr, *tail0 = nullptr;   // tail0: the state of an empty stream (a reset is a device-to-device copy, no host wait)
    Carry *carry = nullptr;            // = &tail->carry
    struct Pinned { Tail tail; uint32_t flags[4]; } *pin = nullptr;   // pinned landing zone of the small read-backs
    bool tail_on_host = false;         // pin->tail is what the device holds (the last feed brought it along with its flags)
    bool zero_timed = true;            // t_zero of the last reset has been read from its events
    unsigned long long *hist = nullptr;
    unsigned long long *hist_rep = nullptr;   // HIST_REPLICAS copies of one feed's histogram change (zero between feeds)
    DevRec *recs = nullptr;
    uint64_t recs_cap = 0;
    L1 *c_l1 = nullptr, *c_l1s = nullptr;
    L2 *c_l2 = nullptr, *c_l2s = nullptr;
    LaneState *lane_state = nullptr;   // per 64-byte piece: start state relative to its chunk
    PiecePack *packs = nullptr;        // per 64-byte piece: its bases, classified and pushed together (structure pass -> squeeze pass)
    uint32_t *chunk_odd = nullptr;     // per chunk: pieces that are not plain sequence text
    L1 *t_l1 = nullptr;                // scan scratch: one summary per 1024 chunks
    L2 *t_l2 = nullptr;
    uint32_t chunk_cap = 0;
    uint8_t *staging[2] = {nullptr, nullptr};   // device copies of host-fed pieces (one counted while the next uploads)
    uint64_t staging_cap[2] = {0, 0};
    uint64_t bytes_fed = 0, n_recs = 0;
    bool finished = false;
    hipEvent_t ev[12] = {};
    double t_scan = 0, t_squeeze = 0, t_sort = 0, t_final = 0, t_zero = 0, t_part = 0, t_bucket = 0;
    int feeds = 0, relayouts = 0;
    uint64_t recounted = 0;                                // buckets whose byte counters wrapped and were counted again (k_bucket_count_bytes)
    bool table_fresh = true;         // no feed has written the u8 table since the last reset
    uint8_t *ws = nullptr;           // workspace of the partition passes
    size_t ws_cap = 0;
};

static int ix_reset(pk_indexer *ix) {
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipEventRecord(ix->ev[6], ix->stream));
    // the first feed writes every slice of the u8 table itself (k_bucket_count, fresh); the table is only
    // zeroed if nothing gets fed at all (see pk_indexer_finish).  Nothing here waits for the device: the stream orders
    // the reset behind whatever is still running, and its duration is read at the next point that waits anyway.
    HIPCHK(hipMemcpyAsync(ix->tail, ix->tail0, sizeof(pk_indexer::Tail), hipMemcpyDeviceToDevice, ix->stream));
    ix->tail_on_host = false;
    if (ix->recs) HIPCHK(hipMemsetAsync(ix->recs, 0, ix->recs_cap * sizeof(DevRec), ix->stream));
    HIPCHK(hipEventRecord(ix->ev[7], ix->stream));
    ix->zero_timed = false;
    ix->t_zero = 0;
    ix->bytes_fed = ix->n_recs = 0;
    ix->finished = false;
    ix->table_fresh = true;
    ix->t_scan = ix->t_squeeze = ix->t_sort = ix->t_final = ix->t_part = ix->t_bucket = 0;
    ix->feeds = ix->relayouts = 0; ix->recounted = 0;
    return PK_OK;
}

// after a wait on the stream: the duration of the last reset, if it has not been read yet
static void time_reset(pk_indexer *ix) {
    if (ix->zero_timed) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, ix->ev[6], ix->ev[7]) == hipSuccess) ix->t_zero = ms * 1e-3;
    ix->zero_timed = true;
}

extern "C" void pk_indexer_destroy(pk_indexer *ix) {
    if (!ix) return;
    hipSetDevice(ix->device);
    if (ix->stream) hipStreamSynchronize(ix->stream);
    hipFree(ix->table8); hipFree(ix->tail); hipFree(ix->tail0); hipFree(ix->hist_rep); hipFree(ix->recs);
    if (ix->pin) hipHostFree(ix->pin);
    hipFree(ix->c_l1); hipFree(ix->c_l1s); hipFree(ix->c_l2); hipFree(ix->c_l2s); hipFree(ix->lane_state); hipFree(ix->packs); hipFree(ix->chunk_odd); hipFree(ix->t_l1); hipFree(ix->t_l2); hipFree(ix->staging[0]); hipFree(ix->staging[1]); hipFree(ix->ws);
    for (auto &e : ix->ev) if (e) hipEventDestroy(e);
    if (ix->stream) hipStreamDestroy(ix->stream);
    delete ix;
}

extern "C" int pk_indexer_create(pk_indexer **out, int k, int device) { return pk_indexer_create_slice(out, k, device, 0, 1); }

extern "C" int pk_indexer_create_slice(pk_indexer **out, int k, int device, int slice_index, int n_slices) {
    if (!out) return fail(PK_ERR_ARG, "null output pointer");
    *out = nullptr;
    int slice_bits = 0;
    while (slice_bits < 30 && (1 << slice_bits) < n_slices) slice_bits++;
    if (n_slices < 1 || (1 << slice_bits) != n_slices) return fail(PK_ERR_ARG, "the number of address slices must be a power of two, got %d", n_slices);
    int rc = check_k(k, slice_bits, slice_index);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    pk_indexer *ix = new pk_indexer();
    ix->k = k; ix->device = device; ix->slice_bits = slice_bits; ix->slice_index = slice_index;
    ix->n = 1ULL << (2 * k - slice_bits);
    auto bail = [&](hipError_t e, const char *what) {
        int r = fail(PK_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        std::string keep = g_err;
        pk_indexer_destroy(ix);
        g_err = keep;
        return r;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    for (auto &ev : ix->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&ix->table8, std::max<uint64_t>(ix->n, 16))) != hipSuccess) return bail(e, "hipMalloc(u8 table)");
    if ((e = hipMalloc(&ix->tail, sizeof(pk_indexer::Tail))) != hipSuccess) return bail(e, "hipMalloc(carry)");
    if ((e = hipMalloc(&ix->tail0, sizeof(pk_indexer::Tail))) != hipSuccess) return bail(e, "hipMalloc(carry0)");
    ix->carry = &ix->tail->carry; ix->hist = ix->tail->hist;
    if ((e = hipHostMalloc(&ix->pin, sizeof(*ix->pin), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
    {
        Carry c;
        memset(&c, 0, sizeof c);
        c.l1 = 8u | 1u | (LS_START << 1);                    // l1_state(LS_START)
        c.l2.flags = F_NONID | F_PRESET | F_BRK;             // l2_state(0, 0, 0, 0)
        if ((e = hipMemset(ix->tail0, 0, sizeof(pk_indexer::Tail))) != hipSuccess) return bail(e, "hipMemset(carry0)");
        if ((e = hipMemcpy(&ix->tail0->carry, &c, sizeof c, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(carry0)");
    }
    if ((e = hipMalloc(&ix->hist_rep, (size_t)HIST_REPLICAS * 256 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(hist replicas)");
    if ((e = hipMemset(ix->hist_rep, 0, (size_t)HIST_REPLICAS * 256 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(hist replicas)");
    {
        // room for the records of small inputs from the start: the squeeze pass checks the capacity itself (see feed_piece)
        const uint64_t cap = 4096;
        if ((e = hipMalloc(&ix->recs, cap * sizeof(DevRec))) != hipSuccess) return bail(e, "hipMalloc(records)");
        ix->recs_cap = cap;
    }
    part_set_attributes();                               // dynamic-LDS opt-ins, once per process and device
    rc = ix_reset(ix);
    if (rc) { std::string keep = g_err; pk_indexer_destroy(ix); g_err = keep; return rc; }
    *out = ix;
    return PK_OK;
}

extern "C" int pk_indexer_reset(pk_indexer *ix) {
    if (!ix) return fail(PK_ERR_ARG, "null indexer");
    return ix_reset(ix);
}

static int ensure_chunks(pk_indexer *ix, uint32_t n_chunks) {
    if (n_chunks <= ix->chunk_cap) return PK_OK;
    hipFree(ix->c_l1); hipFree(ix->c_l1s); hipFree(ix->c_l2); hipFree(ix->c_l2s); hipFree(ix->lane_state); hipFree(ix->packs); hipFree(ix->chunk_odd); hipFree(ix->t_l1); hipFree(ix->t_l2);
    ix->c_l1 = ix->c_l1s = nullptr; ix->c_l2 = ix->c_l2s = nullptr; ix->lane_state = nullptr; ix->packs = nullptr; ix->chunk_odd = nullptr; ix->t_l1 = nullptr; ix->t_l2 = nullptr;
    ix->chunk_cap = 0;
    HIPCHK(hipMalloc(&ix->c_l1, n_chunks * sizeof(L1)));
    HIPCHK(hipMalloc(&ix->c_l1s, n_chunks * sizeof(L1)));
    HIPCHK(hipMalloc(&ix->c_l2, n_chunks * sizeof(L2)));
    HIPCHK(hipMalloc(&ix->c_l2s, n_chunks * sizeof(L2)));
    HIPCHK(hipMalloc(&ix->lane_state, (size_t)n_chunks * WG * sizeof(LaneState)));
    HIPCHK(hipMalloc(&ix->packs, (size_t)n_chunks * WG * sizeof(PiecePack)));
    HIPCHK(hipMalloc(&ix->chunk_odd, (size_t)n_chunks * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&ix->t_l1, ((size_t)n_chunks / 1024 + 1) * sizeof(L1)));
    HIPCHK(hipMalloc(&ix->t_l2, ((size_t)n_chunks / 1024 + 1) * sizeof(L2)));
    ix->chunk_cap = n_chunks;
    return PK_OK;
}

static int ensure_recs(pk_indexer *ix, uint64_t need) {
    if (need <= ix->recs_cap) return PK_OK;
    uint64_t cap = std::max<uint64_t>(need, std::max<uint64_t>(1024, ix->recs_cap * 2));
    DevRec *nr = nullptr;
    HIPCHK(hipMalloc(&nr, cap * sizeof(DevRec)));
    HIPCHK(hipMemsetAsync(nr, 0, cap * sizeof(DevRec), ix->stream));
    if (ix->recs && ix->recs_cap)
        HIPCHK(hipMemcpyAsync(nr, ix->recs, ix->recs_cap * sizeof(DevRec), hipMemcpyDeviceToDevice, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    hipFree(ix->recs);
    ix->recs = nr; ix->recs_cap = cap;
    return PK_OK;
}

// one feed of at most FEED_MAX bytes: structure pass -> squeeze -> bucket layout -> fused k-mer assembly + level-1
// sort -> level 2 -> bucket count.  Record positions are 32-bit: both bucket areas (their capacity + the dump tile) must
// end below 2^32 records.  The worst plan is k = 17 (2^18 final buckets x 4104 records of fixed slack + 25 % on the
// estimate): 2 GiB of text -> capacity2 = 3.76e9.  feed_piece checks the plan it actually got and refuses otherwise.
// 32-bit k-mers (k <= 15): 1 GiB pieces, record positions below 2^31 -- their sort kernels store through 32-bit byte
// offsets (part_common.h: OFF32).
static const uint64_t FEED_MAX = 2ULL << 30;
static uint64_t feed_max_for(int k) { return k <= 15 ? (1ULL << 30) : FEED_MAX; }

static bool plan_fits_u32(const PartPlan &pl) {
    const uint64_t lim = (pl.k <= 15 ? (1ULL << 31) : (1ULL << 32)) - (16384 + 64);   // the dump tile behind the buckets (part_common.h: TILE)
    return pl.capacity1 < lim && pl.capacity2 < lim;
}

// diagnostics (include/pykmer_hip.h): the partition plan of one feed of n_bytes at kmer_len k
extern "C" int pk_diag_plan(int k, uint64_t n_bytes, uint64_t out[8]) {
    if (!out) return fail(PK_ERR_ARG, "null output");
    int rc = check_k(k, k > 17 ? 2 * k - 34 : 0, 0);
    if (rc) return rc;
    if (n_bytes == 0) n_bytes = feed_max_for(k);
    const PartPlan pl = make_part_plan((uint32_t)k, n_bytes, k > 17 ? (uint32_t)(2 * k - 34) : 0u, 0u);
    out[0] = feed_max_for(k); out[1] = pl.capacity1; out[2] = pl.capacity2; out[3] = pl.B1; out[4] = pl.B2; out[5] = pl.fb_bits;
    out[6] = pl.n_chunks; out[7] = plan_fits_u32(pl) ? 1 : 0;
    return PK_OK;
}

static int feed_piece(pk_indexer *ix, const uint8_t *f, uint64_t n_bytes) {
    const uint32_t n_chunks = (uint32_t)((n_bytes + CHUNK - 1) / CHUNK);
    int rc = ensure_chunks(ix, n_chunks);
    if (rc) return rc;
    PartPlan pl = make_part_plan((uint32_t)ix->k, n_bytes, (uint32_t)ix->slice_bits, (uint32_t)ix->slice_index);
    if (!plan_fits_u32(pl)) return fail(PK_ERR_ARG, "feed of %llu bytes needs record positions beyond 2^32 (internal limit); split it", (unsigned long long)n_bytes);
    PartWorkspace lay;
    const size_t need = part_workspace_bytes(pl, n_bytes, &lay);
    if (need > ix->ws_cap) {
        hipFree(ix->ws); ix->ws = nullptr; ix->ws_cap = 0;
        HIPCHK(hipMalloc(&ix->ws, need));
        ix->ws_cap = need;
    }
    uint32_t *flag_words = (uint32_t *)(ix->ws + lay.side_n);              // side-list length (u64), then flags[4]
    uint32_t *flags = flag_words + 2;
    // Nothing between here and the last kernel of the feed waits for the device: the record array was sized from what the
    // feeds so far held (ensure_recs below, after the feed), the squeeze pass checks that against the count the structure
    // pass leaves in `carry` and backs out if it does not fit (flags[0] = 2), the sorts back out if a sampled bucket
    // room does not hold (flags[0] = 1), and the host reads flags + record count once, behind the last kernel.
    HIPCHK(hipEventRecord(ix->ev[0], ix->stream));
    launch_chunk_l1(f, n_bytes, ix->c_l1, n_chunks, ix->stream);
    launch_scan_l1(ix->c_l1, n_chunks, ix->carry, ix->c_l1s, ix->t_l1, flag_words, PART_FLAG_WORDS, ix->stream);
    launch_chunk_l2(f, n_bytes, ix->c_l1s, ix->c_l2, ix->lane_state, ix->packs, ix->chunk_odd, n_chunks, (uint32_t)ix->k, ix->stream);
    launch_scan_l2(ix->c_l2, n_chunks, ix->carry, ix->c_l2s, ix->t_l2, (uint32_t)ix->k, ix->stream);
    HIPCHK(hipEventRecord(ix->ev[1], ix->stream));
    float a = 0, b = 0, c = 0, d = 0, e = 0;
    bool armed = true;                                       // the scan kernel zeroed the flag words for the first attempt
    bool squeeze = true;
    uint32_t stride = pl.sample_stride;
    for (int attempt = 0;; attempt++) {
        if (squeeze) {
            if (!armed) HIPCHK(hipMemsetAsync(flag_words, 0, PART_FLAG_WORDS * 4, ix->stream));
            HIPCHK(hipEventRecord(ix->ev[2], ix->stream));
            launch_squeeze(f, n_bytes, ix->bytes_fed, ix->lane_state, ix->packs, ix->c_l2s, ix->chunk_odd, (uint32_t)ix->k, n_chunks, pl.n_wg0, pl.G, (uint32_t *)(ix->ws + lay.codes),
                           (uint32_t *)(ix->ws + lay.restarts), (uint32_t *)(ix->ws + lay.n_bases), ix->recs, ix->recs_cap, ix->carry, flags, ix->stream);
            HIPCHK(hipEventRecord(ix->ev[3], ix->stream));
            armed = true;
        }
        // the level-1 buckets are laid out from a sample of the slots; if one of them runs out of room every later kernel
        // returns untouched (flags[0]) and the passes behind the squeeze are repeated with exact sizes
        if (launch_partitioned(ix->c_l2s, n_bytes, pl, stride, ix->ws, lay, ix->table8, ix->stream, ix->ev[10], ix->ev[11], ix->ev[8],
                               ix->table_fresh, ix->hist, ix->hist_rep, armed))
            return fail(PK_ERR_HIP, "partition pipeline launch failed: %s", hipGetErrorString(hipGetLastError()));
        HIPCHK(hipEventRecord(ix->ev[9], ix->stream));
        volatile uint32_t *got = ix->pin->flags;
        // what the host needs of the feed, in two small copies behind the last kernel: the flags, and the stream totals +
        // value histogram (pk_indexer_finish then has nothing left to fetch)
        HIPCHK(hipMemcpyAsync(ix->pin->flags, flags, sizeof ix->pin->flags, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipMemcpyAsync(&ix->pin->tail, ix->tail, sizeof(pk_indexer::Tail), hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        HIPCHK(hipGetLastError());
        time_reset(ix);
#ifdef PK_PHASE_PROF
        {   // experiment builds: cycles thread 0 of every level-1 workgroup spent per phase, summed over workgroups and tiles
            unsigned long long pp[26];
            HIPCHK(hipMemcpy(pp, flag_words + 8, sizeof pp, hipMemcpyDeviceToHost));
            fprintf(stderr, "[phase prof] k_walk_sort assembly %llu count %llu scan %llu park %llu store %llu (cycles, all workgroups)\n", pp[0], pp[1], pp[2], pp[3], pp[4]);
            fprintf(stderr, "[phase prof] k_scatter2  unpack   %llu count %llu scan %llu park %llu store %llu\n", pp[5], pp[6], pp[7], pp[8], pp[9]);
            fprintf(stderr, "[phase prof] k_squeeze   window %llu clean %llu queued %llu tally+scan+pack %llu delivery %llu store %llu\n",
                    pp[10], pp[11], pp[12], pp[13], pp[14], pp[15]);
            fprintf(stderr, "[phase prof]   of queued: barrier1 %llu load+begin %llu masks %llu bytes %llu flush %llu write %llu barrier2 %llu\n", pp[18], pp[19], pp[20], pp[21], pp[22], pp[23], pp[24]);
        }
#endif
        if (!got[0]) { ix->recounted += got[1]; break; }
        if (attempt >= 3) return fail(PK_ERR_HIP, "the feed's layout did not settle (internal error, flag %u)", got[0]);
        armed = false;
        if (got[0] == 2u) {                                  // more records than the array holds: grow it, squeeze again
            rc = ensure_recs(ix, ix->pin->tail.carry.n_recs);
            if (rc) return rc;
            squeeze = true;
        } else {                                             // a bucket outgrew its sampled room: lay out again, exactly
            if (stride == 1) return fail(PK_ERR_HIP, "level-1 buckets overflowed an exact layout (internal error)");
            stride = 1;
            squeeze = false;
            ix->relayouts++;
        }
    }
    const uint64_t recs_before = ix->n_recs;
    ix->n_recs = ix->pin->tail.carry.n_recs;
    ix->tail_on_host = true;
    // room for the next feed's records before it arrives: as many again as this feed brought, and then some
    rc = ensure_recs(ix, ix->n_recs + 2 * (ix->n_recs - recs_before) + 1024);
    if (rc) return rc;
    ix->table_fresh = false;
    HIPCHK(hipEventElapsedTime(&a, ix->ev[0], ix->ev[1]));
    HIPCHK(hipEventElapsedTime(&b, ix->ev[2], ix->ev[3]));
    HIPCHK(hipEventElapsedTime(&c, ix->ev[3], ix->ev[8]));
    HIPCHK(hipEventElapsedTime(&d, ix->ev[8], ix->ev[9]));
    HIPCHK(hipEventElapsedTime(&e, ix->ev[10], ix->ev[11]));
    ix->t_scan += a * 1e-3; ix->t_squeeze += b * 1e-3; ix->t_part += c * 1e-3; ix->t_bucket += d * 1e-3; ix->t_sort += e * 1e-3;
    ix->feeds++;
    ix->bytes_fed += n_bytes;
    return PK_OK;
}

extern "C" int pk_indexer_feed_device(pk_indexer *ix, const void *dev_fasta, uint64_t n_bytes) {
    if (!ix) return fail(PK_ERR_ARG, "null indexer");
    if (ix->finished) return fail(PK_ERR_STATE, "indexer already finished; reset it first");
    if (n_bytes == 0) return PK_OK;
    if (!dev_fasta || ((uintptr_t)dev_fasta & 15u)) return fail(PK_ERR_ARG, "device FASTA pointer must be non-null and 16-byte aligned");
    if (n_bytes > (1ULL << 40)) return fail(PK_ERR_ARG, "feed of %llu bytes too large; split it", (unsigned long long)n_bytes);
    HIPCHK(hipSetDevice(ix->device));
    const uint8_t *f = (const uint8_t *)dev_fasta;
    const uint64_t piece_max = feed_max_for(ix->k);                 // a multiple of 16: pieces stay aligned
    for (uint64_t off = 0; off < n_bytes; off += piece_max) {
        int rc = feed_piece(ix, f + off, std::min(piece_max, n_bytes - off));
        if (rc) return rc;
    }
    return PK_OK;
}

// Host text arrives in pieces of FEED_PIECE bytes through two staging buffers in HBM: while the GPU counts piece i,
// the copy threads already move piece i+1 across PCIe (the upload, ~15 ms per 0.8 GB, is the longer of the two).
extern "C" int pk_indexer_feed(pk_indexer *ix, const uint8_t *host_fasta, uint64_t n_bytes) {
    if (!ix) return fail(PK_ERR_ARG, "null indexer");
    if (n_bytes == 0) return PK_OK;
    if (!host_fasta) return fail(PK_ERR_ARG, "null FASTA pointer");
    HIPCHK(hipSetDevice(ix->device));
    const char *env = getenv("PK_FEED_PIECE");
    uint64_t piece = env ? strtoull(env, nullptr, 10) : (256ULL << 20);
    piece = std::max<uint64_t>(1 << 20, std::min<uint64_t>(piece, 1ULL << 30)) & ~15ULL;
    const uint64_t n_pieces = (n_bytes + piece - 1) / piece;
    const uint64_t buf_bytes = std::min(piece, n_bytes) + 64;
    const int n_bufs = n_pieces > 1 ? 2 : 1;
    for (int i = 0; i < n_bufs; i++)
        if (buf_bytes > ix->staging_cap[i]) {
            hipFree(ix->staging[i]); ix->staging[i] = nullptr; ix->staging_cap[i] = 0;
            HIPCHK(hipMalloc(&ix->staging[i], buf_bytes));
            ix->staging_cap[i] = buf_bytes;
        }
    auto upload = [&](uint64_t p) -> int {
        const uint64_t off = p * piece, len = std::min(piece, n_bytes - off);
        return bounce_copy(ix->staging[p & 1], const_cast<uint8_t *>(host_fasta) + off, len, true, ix->device);
    };
    int rc = upload(0);
    if (rc) return rc;
    for (uint64_t p = 0; p < n_pieces; p++) {
        int up_rc = PK_OK;
        std::string up_err;
        std::thread next;
        if (p + 1 < n_pieces) next = std::thread([&]() { up_rc = upload(p + 1); if (up_rc) up_err = g_err; });
        const uint64_t off = p * piece, len = std::min(piece, n_bytes - off);
        rc = pk_indexer_feed_device(ix, ix->staging[p & 1], len);
        if (next.joinable()) next.join();
        if (rc) return rc;
        if (up_rc) { g_err = up_err; return up_rc; }
    }
    return PK_OK;
}

extern "C" int pk_indexer_finish(pk_indexer *ix, uint64_t *num_kmers_out, uint64_t *total_bp_out, uint64_t hist256_out[256],
                                 uint64_t *n_recs_out) {
    if (!ix) return fail(PK_ERR_ARG, "null indexer");
    HIPCHK(hipSetDevice(ix->device));
    if (!ix->finished) {
        if (ix->tail_on_host && !ix->table_fresh) {
            // the usual case: the last feed's read-back already holds the totals and the histogram (kept up to date by
            // k_bucket_count / k_apply_side: no pass over the table), and every kernel has finished -- nothing to do
            ix->t_final = 0;
        } else {
            HIPCHK(hipEventRecord(ix->ev[4], ix->stream));
            if (ix->table_fresh) {                           // nothing was fed: the table is all zero
                HIPCHK(hipMemsetAsync(ix->table8, 0, std::max<uint64_t>(ix->n, 16), ix->stream));
                ix->table_fresh = false;
            }
            HIPCHK(hipEventRecord(ix->ev[5], ix->stream));
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(&ix->pin->tail, ix->tail, sizeof(pk_indexer::Tail), hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            ix->tail_on_host = true;
            time_reset(ix);
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ix->ev[4], ix->ev[5]));
            ix->t_final = ms * 1e-3;
        }
        ix->finished = true;
    }
    const Carry &c = ix->pin->tail.carry;
    if (num_kmers_out) *num_kmers_out = c.num_kmers;
    if (total_bp_out) *total_bp_out = c.total_bp;
    if (n_recs_out) *n_recs_out = c.n_recs;
    if (hist256_out) {
        const unsigned long long *h = ix->pin->tail.hist;
        uint64_t nonzero = 0;
        for (int v = 1; v < 256; v++) { hist256_out[v] = h[v]; nonzero += h[v]; }
        hist256_out[0] = ix->n - nonzero;                // zeros are not tallied on the device
    }
    return PK_OK;
}

extern "C" int pk_indexer_records(pk_indexer *ix, pk_record *recs_out, uint64_t recs_cap) {
    if (!ix) return fail(PK_ERR_ARG, "null indexer");
    if (ix->n_recs > recs_cap) return fail(PK_ERR_RECS_CAP, "%llu records, capacity %llu", (unsigned long long)ix->n_recs, (unsigned long long)recs_cap);
    if (ix->n_recs == 0) return PK_OK;
    if (!recs_out) return fail(PK_ERR_ARG, "null records pointer");
    HIPCHK(hipSetDevice(ix->device));
    std::vector<DevRec> tmp(ix->n_recs);
    HIPCHK(hipMemcpy(tmp.data(), ix->recs, ix->n_recs * sizeof(DevRec), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < ix->n_recs; i++) {
        recs_out[i].name_off = tmp[i].name_off;
        recs_out[i].name_len = tmp[i].name_end > tmp[i].name_off ? tmp[i].name_end - tmp[i].name_off : 0;
        recs_out[i].seq_len = tmp[i].seq_len;
        recs_out[i].n_valid_kmers = tmp[i].n_valid;
    }
    return PK_OK;
}

extern "C" int pk_indexer_table_to_host(pk_indexer *ix, uint8_t *table_out) {
    if (!ix || !table_out) return fail(PK_ERR_ARG, "null argument");
    if (!ix->finished) return fail(PK_ERR_STATE, "call pk_indexer_finish first");
    return bounce_copy(ix->table8, table_out, ix->n, false, ix->device);
}

extern "C" int pk_indexer_table_slice_to_host(pk_indexer *ix, uint8_t *dst, uint64_t offset, uint64_t n_bytes) {
    if (!ix || !dst) return fail(PK_ERR_ARG, "null argument");
    if (!ix->finished) return fail(PK_ERR_STATE, "call pk_indexer_finish first");
    if (offset > ix->n || n_bytes > ix->n - offset) return fail(PK_ERR_ARG, "slice outside the table");
    return bounce_copy(ix->table8 + offset, dst, n_bytes, false, ix->device);
}

extern "C" int pk_indexer_table_device(pk_indexer *ix, const void **dev_table_out) {
    if (!ix || !dev_table_out) return fail(PK_ERR_ARG, "null argument");
    if (!ix->finished) return fail(PK_ERR_STATE, "call pk_indexer_finish first");
    *dev_table_out = ix->table8;
    return PK_OK;
}

extern "C" int pk_indexer_table_slice_to_device(pk_indexer *ix, void *dev_dst, uint64_t offset, uint64_t n_bytes) {
    if (!ix || !dev_dst) return fail(PK_ERR_ARG, "null argument");
    if (!ix->finished) return fail(PK_ERR_STATE, "call pk_indexer_finish first");
    if (offset > ix->n || n_bytes > ix->n - offset) return fail(PK_ERR_ARG, "slice outside the table");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemcpy(dev_dst, ix->table8 + offset, n_bytes, hipMemcpyDeviceToDevice));
    return PK_OK;
}

extern "C" int pk_indexer_timings(pk_indexer *ix, double out[10]) {
    if (!ix || !out) return fail(PK_ERR_ARG, "null argument");
    for (int i = 0; i < 10; i++) out[i] = 0;
    out[0] = ix->t_scan; out[1] = ix->t_squeeze; out[2] = ix->t_final; out[3] = ix->t_zero; out[4] = (double)ix->feeds;
    out[5] = ix->t_part; out[6] = ix->t_bucket; out[7] = ix->t_sort; out[8] = (double)ix->relayouts; out[9] = (double)ix->recounted;
    return PK_OK;
}

static pk_indexer *g_cached_indexer = nullptr;

extern "C" int pk_count_release(void) {
    if (g_cached_indexer) { pk_indexer_destroy(g_cached_indexer); g_cached_indexer = nullptr; }
    return PK_OK;
}

extern "C" int pk_count_fasta(const uint8_t *fasta, uint64_t n_bytes, int k, uint8_t *table_out, uint64_t *num_kmers_out,
                              uint64_t *total_bp_out, uint64_t hist256_out[256], pk_record *recs_out, uint64_t recs_cap,
                              uint64_t *n_recs_out, int device) {
    int rc = check_k(k);
    if (rc) return rc;
    if (!table_out) return fail(PK_ERR_ARG, "null table pointer");
    if (n_bytes && !fasta) return fail(PK_ERR_ARG, "null FASTA pointer");
    // one indexer (1 GiB .. 16 GiB table + workspace in HBM) is kept between calls for the same k and device: a caller
    // that counts sample after sample does not pay hipMalloc / hipFree of ~15 GB each time.  pk_count_release() frees it.
    static std::mutex cache_mu;
    std::lock_guard<std::mutex> cache_lock(cache_mu);
    pk_indexer *&ix = g_cached_indexer;
    if (ix && (ix->k != k || ix->device != device)) { pk_indexer_destroy(ix); ix = nullptr; }
    if (!ix) {
        rc = pk_indexer_create(&ix, k, device);
        if (rc) { ix = nullptr; return rc; }
    } else if ((rc = pk_indexer_reset(ix))) {
        return rc;
    }
    auto done = [&](int r) { return r; };
    if ((rc = pk_indexer_feed(ix, fasta, n_bytes))) return done(rc);
    uint64_t n_recs = 0;
    if ((rc = pk_indexer_finish(ix, num_kmers_out, total_bp_out, hist256_out, &n_recs))) return done(rc);
    if (n_recs_out) *n_recs_out = n_recs;
    if ((rc = pk_indexer_table_to_host(ix, table_out))) return done(rc);
    if (n_recs > recs_cap) return done(fail(PK_ERR_RECS_CAP, "%llu records, capacity %llu", (unsigned long long)n_recs, (unsigned long long)recs_cap));
    if ((rc = pk_indexer_records(ix, recs_out, recs_cap))) return done(rc);
    return done(PK_OK);
}

// ================================================================== stats ======================
extern "C" int pk_table_stats(const uint8_t *table, uint64_t n, uint64_t hist256_out[256], int device) {
    if (!hist256_out || (n && !table)) return fail(PK_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(device));
    unsigned long long *d_hist = nullptr;
    uint8_t *d_t = nullptr;
    HIPCHK(hipMalloc(&d_hist, 256 * sizeof(unsigned long long)));
    hipError_t e = hipMalloc(&d_t, std::max<uint64_t>(n, 16));
    if (e != hipSuccess) { hipFree(d_hist); return fail(PK_ERR_HIP, "hipMalloc(table) failed: %s", hipGetErrorString(e)); }
    int rc = PK_OK;
    unsigned long long h[256];
    if (hipMemset(d_hist, 0, 256 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemcpy(d_t, table, n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(PK_ERR_HIP, "upload failed");
    if (!rc) {
        launch_hist8(d_t, n, d_hist, 0);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PK_ERR_HIP, "histogram kernel failed: %s", hipGetErrorString(hipGetLastError()));
    }
    hipFree(d_t); hipFree(d_hist);
    if (rc) return rc;
    uint64_t nz = 0;
    for (int v = 1; v < 256; v++) { hist256_out[v] = h[v]; nz += h[v]; }
    hist256_out[0] = n - nz;
    return PK_OK;
}

// ================================================================== merger =====================
static int check_counts(int N, int min_count, int max_count) {
    if (N < 1) return fail(PK_ERR_ARG, "need at least one table");
    if (N > 128) return fail(PK_ERR_ARG, "at most 128 tables per call (got %d)", N);
    if (min_count < 1 || max_count > 255) return fail(PK_ERR_ARG, "min_count must be >= 1 and max_count <= 255 (merger.py:90-91)");
    return PK_OK;
}

extern "C" int pk_gram_expand(const uint64_t *pair, int N, uint64_t *matrix_out) {
    if (!pair || !matrix_out || N < 1) return fail(PK_ERR_ARG, "bad argument");
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            uint64_t *m = matrix_out + ((uint64_t)i * N + j) * 3;
            if (i == j) { m[0] = m[1] = m[2] = 0; continue; }              // merger.py:136: never assigned
            m[0] = pair[(uint64_t)i * N + i];                                // merger.py:175-176
            m[1] = pair[(uint64_t)j * N + j];
            m[2] = i < j ? pair[(uint64_t)i * N + j] : pair[(uint64_t)j * N + i];
        }
    return PK_OK;
}

// Per-device scan context: the table-pointer array in HBM, a stream and two events, created once and reused by
// every scan on that device (a 2 ms kernel should not pay for hipMalloc / hipEventCreate each call).
#include <mutex>
#include <thread>
namespace {
struct GramCtx {
    std::mutex mu;
    const uint8_t **d_ptrs = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    unsigned long long *d_pair = nullptr;      // scratch N x N for callers that only want the host copy
};
constexpr int MAX_DEVICES = 64;
GramCtx g_gram[MAX_DEVICES];

int gram_ctx(int device, GramCtx **out) {
    if (device < 0 || device >= MAX_DEVICES) return fail(PK_ERR_ARG, "device ordinal %d out of range", device);
    GramCtx &c = g_gram[device];
    static std::mutex init_mu;
    std::lock_guard<std::mutex> init_lock(init_mu);
    if (!c.d_ptrs) {
        HIPCHK(hipMalloc(&c.d_ptrs, 128 * sizeof(void *)));
        HIPCHK(hipMalloc(&c.d_pair, 128 * 128 * sizeof(unsigned long long)));
        HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&c.e0));
        HIPCHK(hipEventCreate(&c.e1));
    }
    *out = &c;
    return PK_OK;
}

// one scan of N device-resident slices; `accumulate` adds into dev_pair instead of overwriting it
int gram_scan_device(const void *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count, uint64_t *pair_out,
                     void *dev_pair, bool accumulate, int device, double *kernel_seconds_out) {
    int rc = check_counts(N, min_count, max_count);
    if (rc) return rc;
    if (!dev_tables) return fail(PK_ERR_ARG, "null table list");
    for (int i = 0; i < N; i++)
        if (!dev_tables[i] || ((uintptr_t)dev_tables[i] & 15u)) return fail(PK_ERR_ARG, "table %d: device pointer must be 16-byte aligned", i);
    HIPCHK(hipSetDevice(device));
    GramCtx *c = nullptr;
    if ((rc = gram_ctx(device, &c))) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    unsigned long long *d_pair = dev_pair ? (unsigned long long *)dev_pair : c->d_pair;
    if (!dev_pair) accumulate = false;
    HIPCHK(hipMemcpyAsync(c->d_ptrs, dev_tables, N * sizeof(void *), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->e0, c->stream));
    if (launch_gram(c->d_ptrs, N, n_slice, min_count, max_count, d_pair, !accumulate, c->stream))
        return fail(PK_ERR_HIP, "gram kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(hipEventRecord(c->e1, c->stream));
    if (pair_out) HIPCHK(hipMemcpyAsync(pair_out, d_pair, (size_t)N * N * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (kernel_seconds_out) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->e0, c->e1)); *kernel_seconds_out = ms * 1e-3; }
    return PK_OK;
}
}  // namespace

extern "C" int pk_gram_device_partial(const void *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count,
                                      uint64_t *pair_out, void *dev_pair_out, int device, double *kernel_seconds_out) {
    return gram_scan_device(dev_tables, N, n_slice, min_count, max_count, pair_out, dev_pair_out, false, device, kernel_seconds_out);
}

extern "C" int pk_gram_device_accumulate(const void *const *dev_tables, int N, uint64_t n_slice, int min_count, int max_count,
                                         void *dev_pair_accum, int device, double *kernel_seconds_out) {
    if (!dev_pair_accum) return fail(PK_ERR_ARG, "null accumulator");
    return gram_scan_device(dev_tables, N, n_slice, min_count, max_count, nullptr, dev_pair_accum, true, device, kernel_seconds_out);
}

// Several windows over the same staged slices: one pass per group of windows (k_gram_mw) where the kernel has room for
// them, one single-window scan each otherwise.  PK_GRAM_MW=0: always one scan per window (comparison runs).
extern "C" int pk_gram_device_accumulate_windows(const void *const *dev_tables, int N, uint64_t n_slice, const int *min_counts,
                                                 const int *max_counts, int n_windows, void *dev_pair_accum, int device,
                                                 double *kernel_seconds_out) {
    if (!dev_pair_accum) return fail(PK_ERR_ARG, "null accumulator");
    if (n_windows < 1 || n_windows > 255 || !min_counts || !max_counts) return fail(PK_ERR_ARG, "between 1 and 255 windows per call");
    int rc = PK_OK;
    for (int w = 0; w < n_windows; w++)
        if ((rc = check_counts(N, min_counts[w], max_counts[w]))) return rc;
    if (!dev_tables) return fail(PK_ERR_ARG, "null table list");
    for (int i = 0; i < N; i++)
        if (!dev_tables[i] || ((uintptr_t)dev_tables[i] & 15u)) return fail(PK_ERR_ARG, "table %d: device pointer must be 16-byte aligned", i);
    static const bool mw_on = !(getenv("PK_GRAM_MW") && atoi(getenv("PK_GRAM_MW")) == 0);
    const int per_pass = mw_on ? gram_windows_per_pass(N) : 0;
    HIPCHK(hipSetDevice(device));
    GramCtx *c = nullptr;
    if ((rc = gram_ctx(device, &c))) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    unsigned long long *acc = (unsigned long long *)dev_pair_accum;
    HIPCHK(hipMemcpyAsync(c->d_ptrs, dev_tables, N * sizeof(void *), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->e0, c->stream));
    std::vector<int> order(n_windows);
    for (int w = 0; w < n_windows; w++) order[w] = w;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return min_counts[a] < min_counts[b]; });
    for (int at = 0; at < n_windows;) {
        const int take = (per_pass >= 2 && n_windows - at >= 2) ? std::min(per_pass, n_windows - at) : 1;
        int lrc;
        if (take == 1) {
            const int w = order[at];
            lrc = launch_gram(c->d_ptrs, N, n_slice, min_counts[w], max_counts[w], acc + (size_t)w * N * N, false, c->stream);
        } else {
            int mn[8], mx[8], out[8];
            for (int i = 0; i < take; i++) { mn[i] = min_counts[order[at + i]]; mx[i] = max_counts[order[at + i]]; out[i] = order[at + i]; }
            lrc = launch_gram_windows(c->d_ptrs, N, n_slice, mn, mx, out, take, acc, c->stream);
        }
        if (lrc) return fail(PK_ERR_HIP, "gram kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        at += take;
    }
    HIPCHK(hipEventRecord(c->e1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (kernel_seconds_out) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->e0, c->e1)); *kernel_seconds_out = ms * 1e-3; }
    return PK_OK;
}

extern "C" int pk_gram(const uint8_t *const *tables, int N, uint64_t n, int min_count, int max_count, uint64_t *matrix_out,
                       const int *devices, int n_devices) {
    int rc = check_counts(N, min_count, max_count);
    if (rc) return rc;
    if (!tables || !matrix_out) return fail(PK_ERR_ARG, "null argument");
    int dev0 = 0;
    if (!devices || n_devices <= 0) { devices = &dev0; n_devices = 1; }
    // address range split into n_devices contiguous slices (multiples of 32 addresses); one host thread per
    // device stages its slice and scans it, all devices at once
    const uint64_t per = ((n + n_devices - 1) / n_devices + 31u) & ~31ULL;
    std::vector<std::vector<uint64_t>> parts(n_devices, std::vector<uint64_t>((size_t)N * N, 0));
    std::vector<int> rcs(n_devices, PK_OK);
    std::vector<std::string> errs(n_devices);
    auto work = [&](int d) {
        const uint64_t lo = std::min<uint64_t>(n, per * d), hi = std::min<uint64_t>(n, lo + per);
        if (hi <= lo) return;
        auto run = [&]() -> int {
            HIPCHK(hipSetDevice(devices[d]));
            // as many tables' slices as fit beside each other in free HBM; the rest in further rounds over
            // sub-slices of the address range (partials add)
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            uint64_t sub = hi - lo;
            const uint64_t budget = (uint64_t)(free_b * 0.8);
            if ((uint64_t)N * (sub + 64) > budget) sub = std::max<uint64_t>(1 << 20, (budget / N - 64) & ~2047ULL);
            std::vector<void *> dptr(N, nullptr);
            int r = PK_OK;
            for (int i = 0; i < N && !r; i++)
                if (hipMalloc(&dptr[i], std::min(sub, hi - lo) + 64) != hipSuccess) r = fail(PK_ERR_HIP, "hipMalloc(table slice) failed");
            std::vector<uint64_t> one((size_t)N * N);
            for (uint64_t a = lo; a < hi && !r; a += sub) {
                const uint64_t b = std::min(hi, a + sub);
                for (int i = 0; i < N && !r; i++)
                    if (hipMemcpy(dptr[i], tables[i] + a, b - a, hipMemcpyHostToDevice) != hipSuccess) r = fail(PK_ERR_HIP, "table upload failed");
                if (!r) r = pk_gram_device_partial((const void *const *)dptr.data(), N, b - a, min_count, max_count, one.data(), nullptr, devices[d], nullptr);
                if (!r) for (size_t i = 0; i < one.size(); i++) parts[d][i] += one[i];
            }
            for (auto p : dptr) hipFree(p);
            return r;
        };
        rcs[d] = run();
        if (rcs[d]) errs[d] = g_err;                       // g_err is thread-local: carry the message back
    };
    if (n_devices == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int d = 0; d < n_devices; d++) th.emplace_back(work, d);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < n_devices; d++)
        if (rcs[d]) { g_err = errs[d]; return rcs[d]; }
    std::vector<uint64_t> pair((size_t)N * N, 0);
    for (int d = 0; d < n_devices; d++)
        for (size_t i = 0; i < pair.size(); i++) pair[i] += parts[d][i];
    return pk_gram_expand(pair.data(), N, matrix_out);
}
