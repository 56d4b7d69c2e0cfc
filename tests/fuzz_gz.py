"""Randomised differential test of the device-side inflate (galah_amd/csrc/gz_inflate.hip; needs a GPU, or the emulator): byte
streams built to reach every corner of DEFLATE -- skewed alphabets (Huffman codes of up to 15 bits: the decode tables'
sub-tables), flat ones (256 literals of 8-9 bits), runs (distance 1, length 258), periodic text (distance < length), far
repeats (the whole 32 KiB window), every zlib strategy (default, filtered, Huffman only, RLE, fixed codes) and level 0-9,
small memLevels (many short blocks) -- behind a '>' header line, so that they are also valid (if odd) FASTA.  The resident
stream and the statistics with ghip_options.gz_device on must equal the host path's (libdeflate / zlib + ghip_parse_fasta),
and the device must have taken every file.
usage: fuzz_gz.py [rounds=40] [seed=1]"""
import os, shutil, struct, sys, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import conftest  # noqa: E402,F401  (the test harness' emulator switch, tests/conftest.py: GALAH_TEST_EMU)
import galah_amd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = galah_amd.Context(0)


def body(kind, n):
    if kind == "dna":
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes()
    if kind == "dna_lines":
        s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, n, p=[.24, .24, .24, .24, .01, .01, .005, .005, .01])].tobytes()
        w = int(rng.choice([60, 70, 80]))
        return b"\n".join(s[j:j + w] for j in range(0, n, w))
    if kind == "flat":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "skewed":      # geometric symbol frequencies: code lengths up to the 15-bit limit
        p = 0.62 ** np.arange(256)
        sym = rng.permutation(256).astype(np.uint8)
        return sym[rng.choice(256, n, p=p / p.sum())].tobytes()
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(32, 127))]) * int(rng.choice([1, 2, 3, 4, 50, 258, 259, 1000, 70_000]))
        return bytes(out[:n])
    if kind == "periodic":
        unit = rng.integers(65, 91, int(rng.integers(2, 40)), dtype=np.uint8).tobytes()
        return (unit * (n // len(unit) + 1))[:n]
    if kind == "far":         # a segment that comes back just inside / at / beyond the window
        seg = rng.integers(65, 85, int(rng.integers(300, 3000)), dtype=np.uint8).tobytes()
        out = bytearray()
        while len(out) < n:
            out += seg + rng.integers(97, 123, int(rng.choice([100, 20_000, 32_768 - len(seg), 32_768, 40_000])), dtype=np.uint8).tobytes()
        return bytes(out[:n])
    raise ValueError(kind)


def gz(data, level, strategy, mem_level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    return b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


KINDS = ["dna", "dna_lines", "flat", "skewed", "runs", "periodic", "far"]
STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
checked = 0
for rnd in range(rounds):
    d = tempfile.mkdtemp(prefix="ghip_fuzz_gz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths, what = [], []
        for f in range(int(rng.integers(1, 7))):
            parts = []
            for r in range(int(rng.integers(1, 4))):
                kind = str(rng.choice(KINDS))
                n = int(rng.choice([0, 1, 2, 3, 64, 257, 258, 259, 4096, 16_383, 16_384, 16_385, 65_536, 200_000]))
                parts.append(b">r%d %s\n" % (r, kind.encode()) + body(kind, n).replace(b"\n>", b"\nN") + b"\n")
            data = b"".join(parts)
            level, strategy, mem = int(rng.integers(0, 10)), int(rng.choice(STRATEGIES)), int(rng.choice([1, 4, 8, 9]))
            path = os.path.join(d, "f%d.fna.gz" % f)
            with open(path, "wb") as fh:
                fh.write(gz(data, level, strategy, mem))
            paths.append(path)
            what.append((level, strategy, mem, len(data)))
        threads = int(rng.integers(1, 5))
        ctx.set_options(gz_device=0)
        g = ctx.genomes_from_files(paths, threads)
        want = [(g.to_host(i).tobytes(), g.stats(i)) for i in range(len(paths))]
        g.free()
        ctx.set_options(gz_device=1)
        before = ctx.ingest_counters()
        g = ctx.genomes_from_files(paths, threads)
        after = ctx.ingest_counters()
        assert after["gz_device_files"] - before["gz_device_files"] == len(paths), (rnd, what, before, after)
        for i, p in enumerate(paths):
            assert (g.to_host(i).tobytes(), g.stats(i)) == want[i], (rnd, p, what[i])
            checked += 1
        g.free()
    finally:
        shutil.rmtree(d, ignore_errors=True)
ctx.set_options(gz_device=0)
print(f"fuzz ok: {rounds} rounds, {checked} files checked")
