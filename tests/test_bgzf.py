"""CPU: BGZF writer / parallel reader -- the README's `bgzip -i -I x.gzi -l 9` step and its inverse."""
import gzip
import os
import struct

import numpy as np
import pytest

from pykmer_amd import bgzf


def _table(n, seed, density=0.3):
    rng = np.random.default_rng(seed)
    t = rng.integers(1, 6, size=n, dtype=np.uint8)
    t[rng.random(n) > density] = 0
    return t


@pytest.mark.parametrize("n", [0, 1, 65279, 65280, 65281, 4 ** 9, 1_000_003])
def test_roundtrip_and_gzip_compatibility(tmp_path, n):
    src = tmp_path / "t.07.kin"
    data = _table(n, n)
    data.tofile(src)
    dst, gzi = bgzf.compress_file(str(src), threads=4)
    assert dst == str(src) + ".bgz" and gzi == dst + ".gzi" and not os.path.exists(dst + ".tmp")
    raw = open(dst, "rb").read()
    assert raw.endswith(bgzf.EOF_BLOCK) and bgzf.is_bgzf(dst)
    assert gzip.decompress(raw) == data.tobytes()                    # any gzip reader (the reference's gzip.open) accepts it
    with gzip.open(dst, "rb") as fh:
        assert fh.read() == data.tobytes()
    back = bgzf.decompress_file(dst, expected_size=n, threads=4)
    assert np.array_equal(back, data)
    # block structure: every member carries BC, payload <= 0xFF00, sizes add up
    blocks = bgzf.scan_blocks(memoryview(raw))
    assert sum(s for _, s in blocks) == len(raw) and blocks[-1][1] == 28
    isizes = [struct.unpack_from("<I", raw, off + size - 4)[0] for off, size in blocks]
    assert max(isizes, default=0) <= 0xFF00 and sum(isizes) == n
    # .gzi: u64 count + (compressed, uncompressed) pairs for every data block but the first (gzireader.py:12-34)
    entries = bgzf.read_gzi(gzi)
    data_blocks = blocks[:-1]
    assert len(entries) == max(0, len(data_blocks) - 1)
    u = 0
    for (off, size), isz, want in zip(data_blocks[1:], isizes[1:], entries):
        u += 0xFF00
        assert want == (off, u)


def test_incompressible_blocks_fit(tmp_path):
    src = tmp_path / "r.bin"
    data = np.random.default_rng(1).integers(0, 256, size=300_000, dtype=np.uint8)
    data.tofile(src)
    dst, _ = bgzf.compress_file(str(src), index=False, threads=2)
    raw = open(dst, "rb").read()
    assert all(size <= 0x10000 for _, size in bgzf.scan_blocks(memoryview(raw)))
    assert np.array_equal(bgzf.decompress_file(dst), data)


def test_plain_gzip_is_not_bgzf_but_still_read(tmp_path):
    """What python's gzip writes (and the reference's tests call .bgz) has no BC field: sequential fallback."""
    p = tmp_path / "x.07.kin.bgz"
    data = _table(4 ** 7, 3)
    with gzip.open(p, "wb") as fh:
        fh.write(data.tobytes())
    assert not bgzf.is_bgzf(str(p)) and bgzf.scan_blocks(memoryview(open(p, "rb").read())) == []
    assert np.array_equal(bgzf.decompress_file(str(p), expected_size=4 ** 7), data)
    with pytest.raises(AssertionError):
        bgzf.decompress_file(str(p), expected_size=5)


def test_corrupt_block_detected(tmp_path):
    src = tmp_path / "c.bin"
    _table(200_000, 5).tofile(src)
    dst, _ = bgzf.compress_file(str(src), index=False)
    raw = bytearray(open(dst, "rb").read())
    raw[40] ^= 0xFF
    open(dst, "wb").write(bytes(raw))
    with pytest.raises((OSError, Exception)):
        bgzf.decompress_file(dst)


def test_header_reads_bgzf_table_and_cli(tmp_path, manifest):
    from pykmer_amd.header import Header
    from test_host_layer import _family_indexes
    path = _family_indexes(tmp_path, manifest, n=1)[0]
    want = np.fromfile(path, dtype=np.uint8)
    bgzf.main([path])                                                # the README step: compress + remove the .kin
    assert not os.path.exists(path) and os.path.exists(path + ".bgz") and os.path.exists(path + ".bgz.gzi")
    h = Header(path + ".bgz", index_file=path + ".bgz")
    assert h.index_file.endswith(".bgz") and np.array_equal(h.read_table(), want)
    assert bytes(list(h)[:100]) == want[:100].tobytes()              # Header.__iter__ streams the same bytes (tools.py:527-533)


def test_decompress_range_reads_only_covering_blocks(tmp_path):
    """Address-range-sharded merge: a rank inflates only the BGZF blocks that cover its slice, located through
    the .gzi (gzireader.py:12-34 layout) or, without one, by walking the block headers; plain gzip streams
    (what the reference's own tests call .bgz, tools.py:300-302) are read sequentially up to the slice end."""
    import gzip
    rng = np.random.default_rng(5)
    data = rng.integers(0, 4, size=1_000_000, dtype=np.uint8)
    src = tmp_path / "t.kin"
    data.tofile(src)
    dst, gzi = bgzf.compress_file(str(src), level=1, threads=2)
    n_blocks = (data.size + bgzf.BLOCK_INPUT - 1) // bgzf.BLOCK_INPUT
    for use_gzi in (True, False):
        if not use_gzi:
            os.remove(gzi)
        for lo, hi in ((0, 10), (0, data.size), (65279, 65281), (200_000, 700_001), (data.size - 5, data.size), (123, 123)):
            part, inflated = bgzf.decompress_range(dst, lo, hi, threads=2)
            assert np.array_equal(part, data[lo:hi]), (use_gzi, lo, hi)
            covering = 0 if hi == lo else (hi - 1) // bgzf.BLOCK_INPUT - lo // bgzf.BLOCK_INPUT + 1
            assert inflated <= covering * bgzf.BLOCK_INPUT and covering <= n_blocks
    plain = tmp_path / "p.kin.bgz"
    with gzip.open(plain, "wb") as fh:
        fh.write(data.tobytes())
    part, inflated = bgzf.decompress_range(str(plain), 300_000, 300_100)
    assert np.array_equal(part, data[300_000:300_100]) and inflated == 300_100


def test_gzi_and_bgz_as_the_reference_reads_them(tmp_path, manifest):
    """Pinned by the reference's own tools (oracle/gen_golden.py bgzf, build container): its gzireader.print_index
    printed `gzireader_stdout` for the .gzi this writer produced, and tools.py:300-302's gzip.open read the table
    back.  Here the same seeded table is compressed again: the index must list what the reference printed and
    python's gzip must return the table.  Byte identity of the compressed file holds for the same zlib only."""
    import hashlib
    import zlib
    import inputs
    case = manifest["bgzf"]["gzi_300k_level9"]
    table = inputs.make_input(case["input"])
    assert hashlib.sha256(table).hexdigest() == case["input_sha256"]
    raw = tmp_path / "bgzf_case.07.kin"
    raw.write_bytes(table)
    dst, gzi = bgzf.compress_file(str(raw), level=case["level"], threads=2)
    with gzip.open(dst, "rb") as fh:
        assert hashlib.sha256(fh.read()).hexdigest() == case["gzip_open_sha256"]
    entries = bgzf.read_gzi(gzi)
    size = os.path.getsize(dst)
    lines = [f"number_entries: {len(entries):15,d}", f"filesize      : {size:15,d}"]
    lines += [f"pos: {i:15,d} compressed_offset {c:15,d} uncompressed_offset {u:15,d}" for i, (c, u) in enumerate(entries)]
    lines += lines[:2]
    want = case["gzireader_stdout"].splitlines()
    assert [ln.split("compressed_offset")[0] for ln in lines] == [ln.split("compressed_offset")[0] for ln in want]
    assert [ln.split("uncompressed_offset")[-1] for ln in lines[2:-2]] == [ln.split("uncompressed_offset")[-1] for ln in want[2:-2]]
    if zlib.ZLIB_VERSION == case["zlib_version"]:
        assert lines == want
        assert hashlib.sha256(open(dst, "rb").read()).hexdigest() == case["bgz_sha256"]
        assert hashlib.sha256(open(gzi, "rb").read()).hexdigest() == case["gzi_sha256"]


def test_block_index_and_pieces(tmp_path, monkeypatch):
    """The block index (from the .gzi, or from one walk over the headers; cached) behind decompress_range, and the
    piecewise inflate the indexer feeds a bgzipped FASTA with (indexer.py:112-115 streams gzip.open)."""
    monkeypatch.setattr(bgzf, "BLOCK_INPUT", 4096)
    src = tmp_path / "t.bin"
    data = _table(1_000_003, 11)
    data.tofile(src)
    dst, gzi = bgzf.compress_file(str(src), threads=2)
    c_offs, c_sizes, u_offs = bgzf.block_index(dst)
    assert len(c_offs) == len(c_sizes) == len(u_offs) - 1 == -(-data.size // 4096)
    assert int(u_offs[-1]) == data.size and int(c_offs[0]) == 0 and (np.diff(c_offs) == c_sizes[:-1]).all()
    assert bgzf.block_index(dst) is bgzf.block_index(dst)                                   # cached
    os.remove(gzi)
    bgzf._INDEX_CACHE.clear()
    again = bgzf.block_index(dst)                                                            # the same index from the headers alone
    assert all(np.array_equal(a, b) for a, b in zip(again, (c_offs, c_sizes, u_offs)))
    rng = np.random.default_rng(3)
    for _ in range(30):
        lo = int(rng.integers(0, data.size))
        hi = int(rng.integers(lo, min(data.size, lo + 50_000) + 1))
        part, inflated = bgzf.decompress_range(dst, lo, hi)
        assert np.array_equal(part, data[lo:hi]) and inflated <= (hi - lo) + 2 * 4096
    for piece in (1, 4096, 10_000, 1 << 20, 1 << 30):
        parts = list(bgzf.iter_pieces(dst, piece, threads=3))
        assert np.array_equal(np.concatenate(parts), data)
        assert all(p.size % 4096 == 0 for p in parts[:-1])                                   # whole blocks
        if piece >= 4096:
            assert all(p.size <= piece for p in parts)
    empty = tmp_path / "e.bin"
    empty.write_bytes(b"")
    dst0, _ = bgzf.compress_file(str(empty))
    assert list(bgzf.iter_pieces(dst0, 100)) == []


def test_native_reader_rejects_a_lying_index_and_corrupt_blocks(tmp_path):
    """pk_bgzf_inflate is handed block lists that may come from a `.gzi` file: offsets past the file, sizes that do not match,
    a flipped payload bit (CRC) -- all must come back as errors, never as reads outside the buffers."""
    from pykmer_amd import _lib
    src = tmp_path / "t.bin"
    data = _table(300_000, 5)
    data.tofile(src)
    dst, _ = bgzf.compress_file(str(src))
    raw = np.fromfile(dst, dtype=np.uint8)
    c_offs, c_sizes, u_offs = bgzf.block_index(dst)
    out = np.empty(data.size, dtype=np.uint8)
    _lib.bgzf_inflate(raw, c_offs, c_sizes, u_offs, out, 3)
    assert np.array_equal(out, data)
    bad = c_offs.copy(); bad[-1] = raw.size - 5
    with pytest.raises(ValueError):
        _lib.bgzf_inflate(raw, bad, c_sizes, u_offs, out, 3)
    bad = c_sizes.copy(); bad[0] = 10
    with pytest.raises(ValueError):
        _lib.bgzf_inflate(raw, c_offs, bad, u_offs, out, 3)
    bad = u_offs.copy(); bad[1] -= 1                                     # ISIZE no longer matches the room
    with pytest.raises(ValueError):
        _lib.bgzf_inflate(raw, c_offs, c_sizes, bad, np.empty(data.size, np.uint8), 3)
    flipped = raw.copy(); flipped[int(c_offs[1]) + 40] ^= 0x10
    with pytest.raises(ValueError):
        _lib.bgzf_inflate(flipped, c_offs, c_sizes, u_offs, out, 3)
    # mixed content: an incompressible block between compressible ones is stored, the level is back for the next one
    mix = np.concatenate([data[:100_000], np.random.default_rng(2).integers(0, 256, 65280, dtype=np.uint8), data[:100_000]])
    (tmp_path / "m.bin").write_bytes(mix.tobytes())
    m_dst, _ = bgzf.compress_file(str(tmp_path / "m.bin"))
    assert gzip.decompress(open(m_dst, "rb").read()) == mix.tobytes()
    sizes = bgzf.block_index(m_dst)[1]
    assert sizes.max() <= 0x10000 and sizes[0] < 30_000 and sizes[-1] < 30_000
