"""The device-side gzip path (ghip_options.gz_device; galah_amd/csrc/gz_inflate.hip, ingest_gz.cpp) against the host path it
replaces (libdeflate / zlib + ghip_parse_fasta: galah_amd/csrc/ingest.cpp) and against the oracle: the resident streams, the
assembly statistics (reference src/genome_stats.rs:11-51), the sketches, and -- for everything the device declines -- the
same result or the same error as without it.  Reference behaviour: needletail auto-detects gzip behind finch::sketch_files
(reference src/finch.rs:69); the reference's own gz test is tests/test_cmdline.rs:612-629."""
import gzip
import io
import os
import struct
import zlib

import numpy as np
import pytest

import galah_amd
import oracle
from conftest import fasta, never_run_on_hardware

# (never_run_on_hardware: ordering only -- written in a round without GPU access, green under the emulator; runs behind the proven tests)
pytestmark = [pytest.mark.gpu, pytest.mark.emu, never_run_on_hardware]


def _fasta_text(rng, lengths, width=60, eol=b"\n", alphabet=b"ACGT", final_newline=True, name=b"rec"):
    letters = np.frombuffer(alphabet, dtype=np.uint8)
    out = []
    for i, n in enumerate(lengths):
        seq = letters[rng.integers(0, len(letters), int(n))].tobytes()
        out.append(b">" + name + b"%d a description" % i)
        out += [seq[j:j + width] for j in range(0, len(seq), width)]
    t = eol.join(out)
    return t + eol if final_newline else t


def _gz(data, level=6, name=None, extra=None, comment=None, hcrc=False):
    """A one-member gzip image with the optional header fields of RFC 1952 2.3.1."""
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03"
    if extra is not None:
        head += struct.pack("<H", len(extra)) + extra
    if name is not None:
        head += name + b"\0"
    if comment is not None:
        head += comment + b"\0"
    if hcrc:
        head += struct.pack("<H", zlib.crc32(head) & 0xffff)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def _bits(spec):
    """Bits in stream order ('1 10 0000001': the first bit of the stream first) -> bytes, the way DEFLATE packs them (LSB first)."""
    bits = [int(c) for c in spec if c in "01"]
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i // 8] |= b << (i % 8)
    return bytes(out)


def _raw_member(deflate, text, isize):
    return b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + deflate + struct.pack("<II", zlib.crc32(text) & 0xffffffff, isize)


def _ingest(ctx, opts, paths, device, threads=3):
    opts(gz_device=1 if device else 0)
    before = ctx.ingest_counters()
    g = ctx.genomes_from_files(paths, threads)
    after = ctx.ingest_counters()
    streams = [g.to_host(i).tobytes() for i in range(len(paths))]
    stats = [g.stats(i) for i in range(len(paths))]
    g.free()
    return streams, stats, {k: after[k] - before[k] for k in after}


def test_device_inflate_equals_host_inflate(ctx, tmp_path, opts):
    """Every kind of deflate block and gzip header field the device path takes: stored blocks (level 0), fixed Huffman (tiny
    inputs), dynamic blocks at levels 1 / 6 / 9, several blocks per member, long overlapping copies (runs of N: distance 1,
    length 258), far matches (a repeated 20 kb segment: distances up to the 32 KiB window), FEXTRA / FNAME / FCOMMENT, CRLF,
    lower case and IUPAC codes, empty records, no final newline, an empty text, a text of blank lines only.  Same resident
    streams and statistics as the host's inflate + parser, and as the oracle; the counters say the device took them all."""
    rng = np.random.default_rng(11)
    files = {}
    base = _fasta_text(rng, [70_000, 31, 0, 12_345])
    for lvl in (0, 1, 6, 9):
        files[f"lvl{lvl}.fna.gz"] = _gz(base, lvl)
    files["tiny_fixed.fna.gz"] = _gz(b">a\nACGTACGTAC\n", 6)
    files["one_byte_header.fna.gz"] = _gz(b">", 6)
    files["empty.fna.gz"] = _gz(b"", 6)
    files["blank_lines.fna.gz"] = _gz(b"\n\n\r\n\n", 6)
    files["blank_then_records.fna.gz"] = _gz(b"\n\n" + base, 6)
    files["nofinal.fna.gz"] = _gz(_fasta_text(rng, [5000, 77], final_newline=False), 6)
    files["crlf.fna.gz"] = _gz(_fasta_text(rng, [9000, 0, 300], eol=b"\r\n"), 6)
    files["iupac.fna.gz"] = _gz(_fasta_text(rng, [40_000], alphabet=b"ACGTacgtNnRYKMSWBDHV-.~UuX* \t"), 6)
    seg = _fasta_text(rng, [20_000], width=20_000)[1:]
    files["repeats.fna.gz"] = _gz(b">r\n" + b"N" * 3000 + b"\n" + seg * 6 + b"A" * 700 + b"\n" + b"ACGT" * 2000 + b"\n", 9)
    files["long_line.fna.gz"] = _gz(_fasta_text(rng, [300_000, 10], width=300_000), 6)
    files["many_records.fna.gz"] = _gz(_fasta_text(rng, [40] * 3000, width=80), 6)
    files["header_fields.fna.gz"] = _gz(base, 6, name=b"genome.fna", extra=b"AB\x02\x00xy", comment=b"made by a test")
    files["fname_only.fna.gz"] = gzip.compress(base, 6)
    buf = io.BytesIO()
    with gzip.GzipFile(filename="inner_name.fna", mode="wb", fileobj=buf, compresslevel=6) as f:
        f.write(base)
    files["python_gzipfile.fna.gz"] = buf.getvalue()
    # what pigz and flushing writers leave in a member: empty stored blocks between the chunks (sync / full flush markers 00 00 ff ff)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for j in range(0, len(base), 9973):
        parts.append(co.compress(base[j:j + 9973]) + co.flush(zlib.Z_FULL_FLUSH if (j // 9973) % 3 == 0 else zlib.Z_SYNC_FLUSH))
    files["flush_markers.fna.gz"] = b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + b"".join(parts) + co.flush() + struct.pack("<II", zlib.crc32(base) & 0xffffffff, len(base))
    paths = []
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    host_streams, host_stats, host_counts = _ingest(ctx, opts, paths, device=False)
    dev_streams, dev_stats, dev_counts = _ingest(ctx, opts, paths, device=True)
    assert host_counts["gz_device_files"] == 0
    assert dev_counts["gz_device_files"] == len(paths) and dev_counts["gz_host_files"] == 0, dev_counts
    for i, p in enumerate(paths):
        assert dev_streams[i] == host_streams[i], p
        assert dev_stats[i] == host_stats[i] == oracle.genome_stats(p), p
        assert dev_streams[i] == galah_amd.fasta_stream(p)[0].tobytes().replace(b"-", b"N"), p
    assert dev_stats[paths.index(str(tmp_path / "many_records.fna.gz"))][0] == 3000
    assert dev_streams[paths.index(str(tmp_path / "empty.fna.gz"))] == b""


def test_what_the_device_declines_goes_through_the_host(ctx, tmp_path, opts):
    """Further members, bytes behind the member, a header CRC, a text that is not FASTA, a '\\r' in front of the first header,
    a wrong CRC-32, a wrong ISIZE, a truncated member, a damaged deflate stream: with gz_device on, the outcome -- streams and
    statistics, or the error -- is the one of the host path, which is the only one to judge a file."""
    rng = np.random.default_rng(12)
    text = _fasta_text(rng, [30_000, 200, 4000])
    cut = text.index(b">rec1")
    good = _gz(text, 6)
    fine = {
        "multi.fna.gz": _gz(text[:cut], 6) + _gz(text[cut:], 6),
        "hcrc.fna.gz": _gz(text, 6, hcrc=True),
        "cr_first.fna.gz": _gz(b"\r" + text, 6),
        "plain_named_gz.fna.gz": text,                       # not gzip at all: the host reads it as the plain file it is
        "ok.fna.gz": good,
    }
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x10
    bad = {
        "trailing_garbage.fna.gz": good + b"garbage!",
        "wrong_crc.fna.gz": good[:-8] + struct.pack("<I", (zlib.crc32(text) ^ 1) & 0xffffffff) + good[-4:],
        "wrong_isize.fna.gz": good[:-4] + struct.pack("<I", len(text) + 1),
        "truncated.fna.gz": good[: len(good) // 2],
        "bit_flip.fna.gz": bytes(flipped),
        "not_fasta.fna.gz": _gz(b"ACGT\nACGT\n", 6),
        "isize_too_small.fna.gz": good[:-4] + struct.pack("<I", len(text) - 100),     # more text than the trailer promises
        "distance_before_the_text.fna.gz": _raw_member(_bits("1 10 0000001 00000 0000000"), b"\0\0\0", 3),   # fixed block: a match as the first symbol
        "block_type_3.fna.gz": _raw_member(_bits("1 11"), b"", 0),
        "stored_len_mismatch.fna.gz": _raw_member(_bits("1 00 00000") + b"\x05\x00\x00\x00>a\nAC", b">a\nAC", 5),   # NLEN is not ~LEN
    }
    paths = []
    for name, data in fine.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    host = _ingest(ctx, opts, paths, device=False)
    dev = _ingest(ctx, opts, paths, device=True)
    assert dev[0] == host[0] and dev[1] == host[1]
    assert dev[2]["gz_device_files"] == 1 and dev[2]["gz_host_files"] == len(paths) - 1, dev[2]   # only ok.fna.gz is the device's
    for name, data in bad.items():
        p = tmp_path / name
        p.write_bytes(data)
        outcome = []
        for device in (False, True):
            opts(gz_device=1 if device else 0)
            try:
                g = ctx.genomes_from_files([str(tmp_path / "ok.fna.gz"), str(p)], 2)
                outcome.append(("ok", g.to_host(1).tobytes(), g.stats(1)))
                g.free()
            except galah_amd.GalahHipError as e:
                outcome.append(("error", str(e)))
        assert outcome[0] == outcome[1], (name, outcome)
        assert outcome[0][0] == "error" or name == "bit_flip.fna.gz", (name, outcome[0][0])   # (a flipped bit may still inflate -- then the CRC-32 catches it)
    assert outcome is not None


def _bgzf(data, block=60_000):
    """What bgzip writes: one gzip member per block of text, each with its size in a 'BC' extra field; an empty member last."""
    out = []
    for j in list(range(0, len(data), block)) + [len(data)]:
        chunk = data[j:j + block] if j < len(data) else b""
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        size = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", size - 1) + body + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    return b"".join(out)


def test_bgzf_files_are_sized_exactly(ctx, tmp_path, opts):
    """A BGZF file (bgzip: one member per 64 KiB, the last one empty) names the size of every member in its header: the capacity
    hint walks them, so the file no longer outgrows the trailer's ISIZE -- as a plain concatenation of members still does (that file
    is then read a second time and the genomes already resident move into a layout of exact lengths: counted).  Same streams and statistics either way, with the device path on or off (it leaves
    multi-member files to the host)."""
    rng = np.random.default_rng(16)
    text = _fasta_text(rng, [150_000, 40_000, 900])
    paths = []
    for name, data in (("a.fna.gz", _gz(text, 6)), ("bgzf.fna.gz", _bgzf(text)), ("b.fna", text)):
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    cat = tmp_path / "cat.fna.gz"
    cut = text.index(b">rec1")
    cat.write_bytes(_gz(text[:cut], 6) + _gz(text[cut:], 6))
    want = galah_amd.fasta_stream(paths[0])
    for device in (0, 1):
        opts(gz_device=device)
        before = ctx.ingest_counters()
        g = ctx.genomes_from_files(paths, 3)
        after = ctx.ingest_counters()
        assert after["two_phase_repeats"] == before["two_phase_repeats"], device
        for i in range(3):
            assert g.to_host(i).tobytes() == want[0].tobytes() and g.stats(i) == tuple(int(x) for x in want[1]), (device, i)
        g.free()
        g = ctx.genomes_from_files(paths + [str(cat)], 3)
        assert ctx.ingest_counters()["two_phase_repeats"] == after["two_phase_repeats"] + 1, device
        assert g.to_host(3).tobytes() == want[0].tobytes() and g.stats(3) == tuple(int(x) for x in want[1])
        g.free()
    # several outgrown files, their texts NOT kept between the pass that finds their lengths and the pass that places them
    # (fault stage 6 = the tests' small limits: what a call with more than 1 GiB of such texts does), more of them than reader threads
    cats = []
    for x in range(5):
        c = tmp_path / f"cat{x}.fna.gz"
        c.write_bytes(_gz(text[:cut], 6) + _gz(text[cut:], 1 + x))
        cats.append(str(c))
    opts(gz_device=0, fault_stage=6)
    g = ctx.genomes_from_files([paths[2]] + cats + [paths[0]], 2)
    for i in range(7):
        assert g.to_host(i).tobytes() == want[0].tobytes() and g.stats(i) == tuple(int(x) for x in want[1]), i
    g.free()


def test_reference_fixtures_through_the_device_path(ctx, opts, golden_sketches):
    """The reference's FASTA test data (tests/golden/fasta/*.fna.gz, as gzip wrote them) inflated on the device: the sketches of
    tests/golden/sketches.npz, the oracle's statistics."""
    names = sorted(n[:-7] for n in os.listdir(os.path.dirname(fasta("x"))) if n.endswith(".fna.gz"))
    paths = [fasta(n) for n in names]
    opts(gz_device=1)
    before = ctx.ingest_counters()
    sk = ctx.sketch_files(paths, 21, 1000, 0, io_threads=4)
    took = ctx.ingest_counters()["gz_device_files"] - before["gz_device_files"]
    assert took == len(paths), (took, len(paths))
    hashes, lens = sk.to_host()
    for i, n in enumerate(names):
        want = oracle.sketch_file(paths[i])
        assert lens[i] == len(want) and np.array_equal(hashes[i, : lens[i]], want), n
        if n in golden_sketches:
            assert np.array_equal(hashes[i, : lens[i]], golden_sketches[n][: lens[i]]), n
    g = ctx.genomes_from_files(paths, 4)
    for i, p in enumerate(paths):
        assert g.stats(i) == oracle.genome_stats(p), p
    g.free()


def test_reference_cli_expectations_with_the_device_inflate(ctx, opts):
    """The reference's command-line expectations (tests/test_cmdline.rs:36-61, :262-352, :62-216, :1100-1125 as restated in
    tests/test_gpu_parity.py -- its fixtures are gzip files, :612-629 is the reference's own gz test) with the files inflated on
    the device: files in -> clusters out, through ghip_sketch_and_index_files, the same clusters and quality order."""
    import test_gpu_parity
    opts(gz_device=1)
    before = ctx.ingest_counters()
    test_gpu_parity.test_reference_cli_expectations_through_hip(ctx)
    after = ctx.ingest_counters()
    assert after["gz_device_files"] - before["gz_device_files"] >= 15 and after["gz_host_files"] == before["gz_host_files"], (before, after)


def test_threshold_and_mixed_input(ctx, tmp_path, opts):
    """gz_device = N sends a gzip file to the device when the call holds N files' worth of it (a launch takes as long as its largest
    file): three of one size pass at N = 3 and not at N = 4; a fourth, ten times their size, stays with the host at N = 3 and
    goes along at N = 1.  Plain files and small groups next to device-ingested files keep their places in the layout (a group of
    small plain files must not be shipped over a neighbour the device wrote)."""
    rng = np.random.default_rng(13)
    paths = []
    for i in range(9):
        gz = i % 3 == 1
        text = _fasta_text(rng, [1500, 700] if gz else [int(rng.integers(50, 4000)) for _ in range(int(rng.integers(1, 4)))])
        p = tmp_path / ("f%d.fna%s" % (i, ".gz" if gz else ""))
        p.write_bytes(_gz(text, 6) if gz else text)
        paths.append(str(p))
    big = tmp_path / "big.fna.gz"
    big.write_bytes(_gz(_fasta_text(rng, [15000, 7000]), 6))
    for files, cases in ((paths, ((4, 0), (3, 3), (1, 3))), (paths[:5] + [str(big)] + paths[5:], ((3, 3), (1, 4)))):
        want = [galah_amd.fasta_stream(p) for p in files]
        for threshold, device_files in cases:
            opts(gz_device=threshold)
            before = ctx.ingest_counters()
            g = ctx.genomes_from_files(files, 3)
            assert ctx.ingest_counters()["gz_device_files"] - before["gz_device_files"] == device_files, (threshold, len(files))
            for i, p in enumerate(files):
                assert g.to_host(i).tobytes() == want[i][0].tobytes().replace(b"-", b"N"), (threshold, p)
                assert g.stats(i) == tuple(int(x) for x in want[i][1]), (threshold, p)
            g.free()


def test_several_batches_in_flight_and_runs_that_find_no_room(ctx, tmp_path, opts):
    """The driver's batch pipeline with a handful of files (fault_stage = gz_small_batches cuts the batches at 3 files): more batches
    than the two that are in flight at a time; and, with fault_rank = 1, every run of files "finds no room" once -- it is retried
    when the batch in flight has given its memory back, else halved.  Same genomes and statistics, all from the device."""
    rng = np.random.default_rng(15)
    paths = []
    for i in range(11):
        p = tmp_path / ("g%02d.fna.gz" % i)
        p.write_bytes(_gz(_fasta_text(rng, [int(rng.integers(200, 9000)) for _ in range(int(rng.integers(1, 5)))]), int(rng.integers(1, 10))))
        paths.append(str(p))
    crumbs = tmp_path / "g11_crumbs.fna.gz"   # 400 records of 300 bp: more than the (test-sized) record pool of its batch holds -> the host's
    crumbs.write_bytes(_gz(_fasta_text(rng, [300] * 400), 6))
    paths.append(str(crumbs))
    host = _ingest(ctx, opts, paths, device=False)
    for no_room in (0, 1):
        opts(fault_stage="gz_small_batches", fault_rank=no_room)
        dev = _ingest(ctx, opts, paths, device=True, threads=2)
        assert dev[0] == host[0] and dev[1] == host[1], no_room
        assert dev[2]["gz_device_files"] == len(paths) - 1 and dev[2]["gz_host_files"] == 1, (no_room, dev[2])
    opts(fault_stage="none", fault_rank=0)
    dev = _ingest(ctx, opts, paths, device=True, threads=2)      # with the pool at its real size the crumbs are the device's too
    assert dev[0] == host[0] and dev[1] == host[1] and dev[2]["gz_device_files"] == len(paths), dev[2]


def test_a_genome_sized_member(ctx, tmp_path, opts):
    """One 2 Mb genome as gzip -6 writes it (some 25 dynamic blocks, 400 000 tokens, every match length and distance class):
    stream, statistics and sketch equal the host path's."""
    rng = np.random.default_rng(14)
    text = _fasta_text(rng, [1_400_000, 500_000, 100_000], width=80)
    p = tmp_path / "genome.fna.gz"
    p.write_bytes(_gz(text, 6))
    host = _ingest(ctx, opts, [str(p)], device=False)
    dev = _ingest(ctx, opts, [str(p)], device=True)
    assert dev[2]["gz_device_files"] == 1 and dev[2]["gz_device_us"] > 0
    assert dev[0] == host[0] and dev[1] == host[1] == [oracle.genome_stats(str(p))]
    opts(gz_device=1)
    hashes, lens = ctx.sketch_files([str(p)], 21, 1000, 0, io_threads=1).to_host()
    want = oracle.sketch_file(str(p))
    assert lens[0] == len(want) and np.array_equal(hashes[0, : lens[0]], want)


def test_randomised_deflate_streams():
    """A short run of tests/fuzz_gz.py: skewed / flat alphabets, runs, periodic and far repeats, every zlib strategy and level."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_gz.py"), "8", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
