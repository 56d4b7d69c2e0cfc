#!/bin/bash
# AddressSanitizer + UBSan over the host-only parts of the library: the FASTA parser (edge files + every fixture) and
# the host clusterer (random graphs).  No GPU needed.  usage: bash scripts/sanitize/run.sh
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T=$(mktemp -d)
FLAGS="-std=c++17 -g -O1 -fsanitize=address,undefined -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -I$R/include -I$R/galah_amd/csrc"
cd "$T"
printf ">a\nACGT" > t1.fna; printf ">a" > t2.fna; printf ">" > t3.fna; printf "\n\n>x\nAC\n>y\n" > t4.fna
python3 - <<'PY'
import random
random.seed(5)
def rec(n): return "".join(random.choice("ACGTACGTACGTacgtNnRY-") for _ in range(n))
for name, eol, w in (("t9_mixed.fna", "\n", 60), ("t10_crlf.fna", "\r\n", 71), ("t11_long.fna", "\n", 20000)):
    with open(name, "w", newline="") as f:
        for i, n in enumerate((70001, 33, 0, 12345, 4097)):
            r = rec(n)
            f.write(">r%d x" % i + eol + eol.join(r[j:j + w] for j in range(0, len(r), w)) + eol)
PY
printf ">a\nACGT\n" | gzip > t5.fna.gz; head -c 20 t5.fna.gz > t6_trunc.fna.gz; printf "\x1f\x8bgarbagegarbagegarbage" > t7_bad.fna.gz; : > t8_empty.fna
g++ $FLAGS "$R/scripts/sanitize/ingest_main.cpp" "$R/galah_amd/csrc/ingest.cpp" -o ingest_t -lz -lpthread -ldl 2>&1 | grep -E "error|undefined" || true
./ingest_t t*.fna t*.fna.gz "$R"/tests/golden/fasta/*.gz | tail -4
printf '#include <cstdlib>\nextern "C" void ghip_free(void *p) { free(p); }\n' > stub.cpp
g++ $FLAGS "$R/scripts/sanitize/cluster_main.cpp" stub.cpp "$R/galah_amd/csrc/cluster.cpp" -o cluster_t -lpthread 2>&1 | grep -E "error|undefined" || true
./cluster_t
rm -rf "$T"
