"""Randomised differential test of the file ingest (needs a GPU): random FASTA files -- record counts and lengths, line
widths, LF / CRLF, lower case, IUPAC codes, gaps, empty records, no final newline, plain / gzip / multi-member gzip -- go
through ghip_genomes_from_files in a randomly chosen form (2-bit packed over PCIe, ASCII, pageable, two-phase; gzip inflated on
the host or on the device); the
resident streams must equal the host parser's (ghip_fasta_stream) byte for byte and the statistics the oracle's.
usage: fuzz_ingest.py [rounds=60] [seed=1]"""
import gzip, os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import conftest  # noqa: E402,F401  (the test harness' emulator switch, tests/conftest.py: GALAH_TEST_EMU)
import galah_amd
import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = galah_amd.Context(0)
ALPHA = np.frombuffer(b"ACGTacgtNnRYKMSWBDHV-.~UuX*", dtype=np.uint8)
checked = 0
for rnd in range(rounds):
    d = tempfile.mkdtemp(prefix="ghip_fuzz_ingest_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = []
        for f in range(int(rng.integers(1, 9))):
            eol = b"\r\n" if rng.random() < 0.3 else b"\n"
            width = int(rng.choice([1, 7, 60, 61, 80, 100, 4093, 70_000]))
            clean = rng.random() < 0.5          # mostly A/C/G/T (the fast paths) or anything goes
            parts = [b"\n" * int(rng.integers(0, 3))] if rng.random() < 0.2 else []
            for r in range(int(rng.integers(1, 6))):
                n = int(rng.choice([0, 1, 3, 31, 32, 33, 500, 8191, 8192, 8193, 40_000, 250_000]))
                p = np.full(len(ALPHA), 0.02 if clean else 1.0)
                p[:4] = 20.0 if clean else 3.0
                seq = ALPHA[rng.choice(len(ALPHA), size=n, p=p / p.sum())].tobytes()
                parts.append(b">rec%d some text" % r + eol)
                parts += [seq[j:j + width] + eol for j in range(0, n, width)]
            data = b"".join(parts)
            if rng.random() < 0.3 and data.endswith(eol):
                data = data[: -len(eol)]
            kind = rng.choice(["plain", "gz", "multi"])
            path = os.path.join(d, "f%d.fna%s" % (f, "" if kind == "plain" else ".gz"))
            if kind == "plain":
                blob = data
            elif kind == "gz":
                blob = gzip.compress(data, 1)
            else:
                cut = data.find(b">", len(data) // 2)
                cut = cut if cut > 0 else len(data) // 2
                blob = gzip.compress(data[:cut], 1) + gzip.compress(data[cut:], 1)
            with open(path, "wb") as fh:
                fh.write(blob)
            paths.append(path)
        want = [galah_amd.fasta_stream(p) for p in paths]
        form = str(rng.choice(["packed", "packed", "ascii", "pageable", "two-phase"]))
        gz_device = int(rng.integers(0, 2))   # 1: the .gz files are inflated, parsed and packed on the device (gz_inflate.hip); what it declines (the multi-member ones) goes the host's way
        ctx.set_options(ingest_form=form, gz_device=gz_device)
        g = ctx.genomes_from_files(paths, int(rng.integers(1, 9)))
        for i, p in enumerate(paths):
            got = g.to_host(i).tobytes()
            # the resident form keeps "is it A/C/G/T" per position (2-bit codes + validity bits): every other stream byte
            # -- 'N', needletail's '-' for gaps -- reads back as 'N'
            assert got == want[i][0].tobytes().replace(b"-", b"N"), (rnd, form, gz_device, p, len(got), len(want[i][0]))
            assert g.stats(i) == tuple(int(x) for x in want[i][1]) == oracle.genome_stats(p), (rnd, form, gz_device, p)
            checked += 1
        g.free()
    finally:
        shutil.rmtree(d, ignore_errors=True)
ctx.set_options(ingest_form="packed", gz_device=0)
print(f"fuzz ok: {rounds} rounds, {checked} files checked, {ctx.ingest_counters()['gz_device_files']} of them inflated on the device")
