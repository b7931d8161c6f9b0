// bgzf_host.cpp -- host-side BGZF block inflate / scan on native threads (no GPU work here).
//
// The reference reads `.kin.bgz` tables and `.fa.gz` inputs through ONE Python `gzip.open` stream (tools.py:294-305,
// indexer.py:112-115).  A BGZF file (what `bgzip` writes: README.md:26, data/README.md:24) is a series of independent gzip
// members of <= 64 KiB, so the members inflate in parallel; from Python threads that stops scaling at a few hundred MB/s
// (per-block interpreter work under the GIL: 1 GiB took 1.0 s on 16 threads), which left a 13-table `.kin.bgz` merge or a
// bgzipped genome waiting on the host for longer than the GPU needs for the whole job.  Here the blocks are inflated by
// zlib on std::threads straight into the caller's buffer, CRC32 and ISIZE checked per block (SAM spec 4.1).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pykmer_hip.h"

namespace pk { int set_error(int code, const std::string &msg); }

namespace {
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
}  // namespace

// Walks the block headers of `src` (no inflation): for every gzip member with a BC extra field its offset, size and
// ISIZE.  Returns PK_ERR_ARG if the bytes are not BGZF, PK_ERR_RECS_CAP if there are more than `cap` blocks (then
// *n_blocks_out holds the count so far).
extern "C" int pk_bgzf_scan(const uint8_t *src, uint64_t n_bytes, uint64_t cap, uint64_t *c_off_out, uint64_t *c_size_out,
                            uint64_t *isize_out, uint64_t *n_blocks_out) {
    if (!n_blocks_out || (n_bytes && !src)) return pk::set_error(PK_ERR_ARG, "null argument");
    uint64_t pos = 0, nb = 0;
    while (pos < n_bytes) {
        if (n_bytes - pos < 18 || src[pos] != 0x1f || src[pos + 1] != 0x8b || src[pos + 2] != 8 || !(src[pos + 3] & 4))
            return pk::set_error(PK_ERR_ARG, "not a BGZF stream (no gzip member with an extra field at byte " + std::to_string(pos) + ")");
        const uint32_t xlen = rd16(src + pos + 10);
        uint64_t x = pos + 12, end = pos + 12 + xlen, bsize = 0;
        if (end > n_bytes) return pk::set_error(PK_ERR_ARG, "truncated BGZF header");
        while (x + 4 <= end) {
            const uint32_t slen = rd16(src + x + 2);
            if (src[x] == 66 && src[x + 1] == 67 && slen == 2) bsize = (uint64_t)rd16(src + x + 4) + 1;
            x += 4 + slen;
        }
        if (!bsize || pos + bsize > n_bytes || bsize < 12 + xlen + 8) return pk::set_error(PK_ERR_ARG, "not a BGZF stream (no BC field, or a truncated block)");
        if (nb < cap && c_off_out && c_size_out && isize_out) { c_off_out[nb] = pos; c_size_out[nb] = bsize; isize_out[nb] = rd32(src + pos + bsize - 4); }
        nb++;
        pos += bsize;
    }
    *n_blocks_out = nb;
    return nb > cap ? pk::set_error(PK_ERR_RECS_CAP, "more BGZF blocks than the caller's arrays hold") : PK_OK;
}

extern "C" int pk_bgzf_inflate(const uint8_t *src, uint64_t src_bytes, const uint64_t *c_off, const uint64_t *c_size, const uint64_t *u_off,
                               uint64_t n_blocks, uint8_t *dst, int threads) {
    if (n_blocks == 0) return PK_OK;
    if (!src || !c_off || !c_size || !u_off || !dst) return pk::set_error(PK_ERR_ARG, "null argument");
    for (uint64_t b = 0; b < n_blocks; b++)                   // the index may come from a `.gzi` file: trust nothing
        if (c_size[b] < 26 || c_off[b] > src_bytes || c_size[b] > src_bytes - c_off[b] || u_off[b + 1] < u_off[b] ||
            (uint64_t)rd16(src + c_off[b] + 10) + 20 > c_size[b])
            return pk::set_error(PK_ERR_ARG, "BGZF block " + std::to_string(b) + ": the block index does not fit the file");
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_blocks) threads = (int)n_blocks;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    std::string why[1];
    std::atomic_flag why_set = ATOMIC_FLAG_INIT;
    auto fail_at = [&](uint64_t b, const char *what) {
        failed.store(1);
        if (!why_set.test_and_set()) why[0] = "BGZF block " + std::to_string(b) + " (at byte " + std::to_string(c_off[b]) + "): " + what;
    };
    auto work = [&]() {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { fail_at(0, "inflateInit2 failed"); return; }
        constexpr uint64_t BATCH = 16;                        // blocks per claim
        for (;;) {
            const uint64_t b0 = next.fetch_add(BATCH);
            if (b0 >= n_blocks || failed.load()) break;
            for (uint64_t b = b0; b < b0 + BATCH && b < n_blocks; b++) {
                const uint8_t *p = src + c_off[b];
                const uint64_t size = c_size[b];
                const uint32_t xlen = rd16(p + 10);
                const uint32_t isize = rd32(p + size - 4), crc = rd32(p + size - 8);
                const uint64_t room = u_off[b + 1] - u_off[b];
                if (isize != room) { fail_at(b, "ISIZE does not match the block index"); break; }
                if (isize == 0) continue;
                inflateReset(&z);
                z.next_in = const_cast<Bytef *>(p + 12 + xlen);
                z.avail_in = (uInt)(size - 12 - xlen - 8);
                z.next_out = dst + u_off[b];
                z.avail_out = isize;
                const int rc = inflate(&z, Z_FINISH);
                if (rc != Z_STREAM_END || z.avail_out != 0) { fail_at(b, "deflate stream is corrupt"); break; }
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + u_off[b], isize) != crc) { fail_at(b, "CRC mismatch"); break; }
            }
        }
        inflateEnd(&z);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed.load()) return pk::set_error(PK_ERR_ARG, why[0]);
    return PK_OK;
}

// The writer's side (README.md:26: `bgzip -i -I x.gzi -l 9 -c x > x.bgz`): every `block_input` bytes of `src` become one
// BGZF block (gzip member with the BC field, raw deflate at `level`, CRC32 + ISIZE), deflated on native threads.  Block i
// is first written at dst + i * 65536 (a block never exceeds 64 KiB: input that does not compress is stored), then the
// blocks are moved together; the end-of-file block is NOT appended (the caller writes it).  dst needs n_blocks * 65536
// bytes.  c_sizes_out[i] = size of block i; *total_out = bytes of dst in use.
extern "C" int pk_bgzf_deflate(const uint8_t *src, uint64_t n_bytes, int level, uint32_t block_input, uint8_t *dst, uint64_t dst_cap,
                               uint64_t *c_sizes_out, uint64_t *total_out, int threads) {
    if (!total_out || (n_bytes && (!src || !dst || !c_sizes_out))) return pk::set_error(PK_ERR_ARG, "null argument");
    if (block_input == 0 || block_input > 0xff00u) return pk::set_error(PK_ERR_ARG, "a BGZF block holds at most 0xff00 input bytes");
    if (level < 0 || level > 9) return pk::set_error(PK_ERR_ARG, "deflate level 0..9");
    const uint64_t n_blocks = (n_bytes + block_input - 1) / block_input;
    *total_out = 0;
    if (n_blocks == 0) return PK_OK;
    if (dst_cap < n_blocks * 65536ull) return pk::set_error(PK_ERR_ARG, "destination too small: 65536 bytes per block");
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_blocks) threads = (int)n_blocks;
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        // two streams per thread: `level`, set up like Python's zlib.compressobj(level, DEFLATED, -15) (the bytes are those
        // the Python writer produced), and level 0 for a block that does not compress into 64 KiB (stored: always fits)
        z_stream z, z0;
        memset(&z, 0, sizeof z); memset(&z0, 0, sizeof z0);
        if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed.store(1); return; }
        if (deflateInit2(&z0, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { deflateEnd(&z); failed.store(1); return; }
        for (;;) {
            const uint64_t b = next.fetch_add(1);
            if (b >= n_blocks || failed.load()) break;
            const uint8_t *in = src + b * block_input;
            const uint32_t len = (uint32_t)std::min<uint64_t>(block_input, n_bytes - b * block_input);
            uint8_t *out = dst + b * 65536ull;
            static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 'B', 'C', 0x02, 0x00};
            memcpy(out, head, 16);
            uint32_t body = 0;
            bool done = false;
            for (z_stream *zs : {&z, &z0}) {
                deflateReset(zs);
                zs->next_in = const_cast<Bytef *>(in); zs->avail_in = len;
                zs->next_out = out + 18; zs->avail_out = 65536 - 18 - 8;
                if (deflate(zs, Z_FINISH) == Z_STREAM_END) { body = (uint32_t)(65536 - 18 - 8 - zs->avail_out); done = true; break; }
            }
            if (!done || body + 26 > 0x10000u) { failed.store(1); break; }
            const uint32_t bsize = body + 25;                                    // total block size - 1
            out[16] = (uint8_t)bsize; out[17] = (uint8_t)(bsize >> 8);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, len);
            uint8_t *tail = out + 18 + body;
            for (int i = 0; i < 4; i++) { tail[i] = (uint8_t)(crc >> (8 * i)); tail[4 + i] = (uint8_t)(len >> (8 * i)); }
            c_sizes_out[b] = body + 26;
        }
        deflateEnd(&z); deflateEnd(&z0);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed.load()) return pk::set_error(PK_ERR_ARG, "deflate failed");
    uint64_t at = 0;
    for (uint64_t b = 0; b < n_blocks; b++) {                                    // move the blocks together
        if (at != b * 65536ull) memmove(dst + at, dst + b * 65536ull, c_sizes_out[b]);
        at += c_sizes_out[b];
    }
    *total_out = at;
    return PK_OK;
}
