"""The issue-arithmetic model of the device-side inflate (docs/design/gzip_device.md "Cost model") fed with what a run counts:
one synthetic genome is written through zlib at several levels, ingested with ghip_options.gz_device on and GHIP_INGEST_DEBUG
set, and the counters the inflate kernel keeps per file (tokens, matches, batches, copy rounds, deflate blocks) go into

    lone wavefront:  cycles = tokens x (scalar/token x 4 + vector/token x 4 + look-ups/token x LDS) + batches x FAR + rounds x ROUND
    saturated:       genomes/s = 1 024 SIMDs x clock / (tokens x scalar/token x 4)

with the instruction counts of profiles/r05_gz_inflate_isa.txt (52 scalar per match, 24 per literal; 12 / 5 vector; 2 / 1
look-ups).  The counters come from the kernel's own bookkeeping, so the script runs wherever the library runs -- on a GPU
box, or under the emulator of tests/emu (GALAH_TEST_EMU=1 with tests/ on the path: how profiles/r05_gz_cost_model.txt was
made).  It prints a MODEL: the measurement is scripts/gz_device_bench.py.
usage: gz_cost_model.py [length=5000000] [levels=1,6,9]"""
import os, re, subprocess, sys, tempfile, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

CLOCK = 2.4e9                 # Hz
SCALAR = {"match": 52, "literal": 24}
VECTOR = {"match": 12, "literal": 5}
LOOKUPS = {"match": 2, "literal": 1}
LDS_ROUND_TRIP = 80           # cycles, ds_read + wait + v_readfirstlane
FAR_LOADS = 0.7e-6            # s per batch: the round trip of the loads of the matches whose source is older than the batch (L2)
COPY_ROUND = 0.15e-6          # s per round: a wave barrier and the LDS reads / writes of the group's stage (the first version of the kernel, whose
                              # rounds went through the text: a drained store queue + a round trip of loads, ~1.75e-6)
SIMDS = 1024
HOST_TEXT_PER_S = 10.1e9      # libdeflate on the boxes' 16-CPU quota: 0.50 s per 1 000 x 5 Mb of 80-column FASTA (profiles/r04a_bench.json)


def child(length, level):
    """runs in a process of its own so that the library's stderr can be read back"""
    if os.environ.get("GALAH_TEST_EMU") == "1":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import conftest  # noqa: F401
    import numpy as np
    import galah_amd
    ctx = galah_amd.Context(0)
    g = ctx.genomes_synthetic_range(42, 10, 0, 1, length, 0.0253)
    seq = g.to_host(0)
    g.free()
    pad = (-len(seq)) % 80
    body = np.concatenate([seq, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
    body = np.concatenate([body, np.full((body.shape[0], 1), ord("\n"), np.uint8)], axis=1).tobytes()
    data = b">genome0 synthetic\n" + body
    d = tempfile.mkdtemp(prefix="ghip_gzmodel_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    p = os.path.join(d, "g.fna.gz")
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    with open(p, "wb") as f:
        f.write(co.compress(data) + co.flush())
    print("GZBYTES", os.path.getsize(p), len(data), flush=True)
    ctx.set_options(gz_device=1, debug=1)
    gg = ctx.genomes_from_files([p], 1)
    assert ctx.ingest_counters()["gz_device_files"] == 1
    gg.free()
    os.remove(p)
    os.rmdir(d)


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)

length = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,6,9").split(",")]
print(f"# one synthetic genome of {length} bp as 80-column FASTA; instruction counts per token from profiles/r05_gz_inflate_isa.txt; {CLOCK / 1e9:.1f} GHz")
print("# level  gz B/base  blocks  tokens   matches  literals  B/token  batches  rounds/batch | lone wavefront: decode s + copy s = s per file | saturated: GB/s of text (x the host's 10.1 GB/s on 16 CPUs)")
for level in levels:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(length), str(level)], capture_output=True, text=True, timeout=3000)
    m = re.search(r"text (\d+) bytes in (\d+) deflate blocks: (\d+) tokens \((\d+) matches\) in (\d+) batches, (\d+) copy rounds", r.stderr)
    z = re.search(r"GZBYTES (\d+) (\d+)", r.stdout)
    if r.returncode != 0 or not m or not z:
        print(level, "failed:", (r.stdout + r.stderr)[-500:])
        continue
    text, blocks, tokens, matches, batches, rounds = (int(x) for x in m.groups())
    literals = tokens - matches
    cyc = sum(n * (SCALAR[k] * 4 + VECTOR[k] * 4 + LOOKUPS[k] * LDS_ROUND_TRIP) for k, n in (("match", matches), ("literal", literals)))
    decode_s, copy_s = cyc / CLOCK, batches * FAR_LOADS + rounds * COPY_ROUND
    sat = SIMDS * CLOCK / (4 * (matches * SCALAR["match"] + literals * SCALAR["literal"]))
    print(f"  {level}      {int(z.group(1)) / length:.3f}     {blocks:5d}  {tokens:8d} {matches:8d} {literals:8d}   {text / tokens:5.2f}   {batches:7d}     {rounds / max(batches, 1):.2f}      |"
          f" {decode_s:.3f} + {copy_s:.3f} = {decode_s + copy_s:.3f} | {sat * text / 1e9:6.1f} ({sat * text / HOST_TEXT_PER_S:.1f}x)")
