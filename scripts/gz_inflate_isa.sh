#!/bin/bash
# Static facts of the device-side inflate (galah_amd/csrc/gz_inflate.hip) read off the compiler's output -- no GPU needed.
# usage: bash scripts/gz_inflate_isa.sh [output = profiles/r05_gz_inflate_isa.txt]
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05_gz_inflate_isa.txt}
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --offload-device-only galah_amd/csrc/gz_inflate.hip -o $T/gz.s 2>/dev/null || exit 1
{
  echo "# $(git rev-parse --short HEAD)  hipcc -O3 --offload-arch=gfx950 -S galah_amd/csrc/gz_inflate.hip"
  echo "## kernels: LDS bytes, scratch bytes, SGPRs, VGPRs"
  python3 - $T/gz.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
    lds, name, scratch, sg, vg = m.groups()
    print(f"{name:70s} lds {lds:>6s}  scratch {scratch:>4s}  sgpr {sg:>4s}  vgpr {vg:>4s}")
PY
  awk '/^_Z17gz_inflate_kernel/,/s_endpgm/' $T/gz.s > $T/inf_full.s
  grep -v "^\s*;" $T/inf_full.s > $T/inf.s
  echo "## gz_inflate_kernel: $(grep -vc '^\.' $T/inf.s) instructions in all"
  # the token loop (decode_batch): from its header label to the end of the last block the compiler marks as part of it
  python3 - $T/inf_full.s <<'PY'
import re, sys
lines = open(sys.argv[1]).read().splitlines()
def body_of(header):
    name = lines[header].split(":")[0].lstrip(".L")
    member = [i for i, l in enumerate(lines) if l.startswith(".LBB") and f"Header={name} " in l]
    if not member:
        return []
    end = next(i for i in range(max(member) + 1, len(lines)) if lines[i].startswith(".LBB"))
    return [l.split()[0] for l in lines[header:end] if l.startswith("\t") and not l.strip().startswith(";")]
headers = [i for i, l in enumerate(lines) if l.startswith(".LBB") and any("Loop Header: Depth=3" in x for x in lines[i + 1:i + 4])]
# decode_batch is the depth-3 loop with the table look-ups (v_readfirstlane) AND the tokens' placement (v_cndmask)
body = max((body_of(h) for h in headers), key=lambda b: (sum(op.startswith("v_cndmask") for op in b) > 0) * sum(op == "v_readfirstlane_b32" for op in b))
cls = {}
for op in body:
    k = "scalar (s_*)" if op.startswith("s_") else "LDS (ds_*)" if op.startswith("ds_") else "vector memory" if op.startswith(("global_", "flat_", "buffer_")) else "vector ALU (v_*)"
    cls[k] = cls.get(k, 0) + 1
print(f"token loop (decode_batch), all paths and exits: {len(body)} instructions:", ", ".join(f"{v} {k}" for k, v in sorted(cls.items(), key=lambda x: -x[1])))
print("in it:", sum(1 for op in body if op == "ds_read_b32"), "ds_read_b32 +", sum(1 for op in body if op == "v_readfirstlane_b32"), "v_readfirstlane_b32 (primary and sub-table look-ups of the two codes),",
      sum(1 for op in body if op == "v_readlane_b32"), "v_readlane_b32 (input window), tokens to their lanes:", sum(1 for op in body if op.startswith("v_cndmask")), "v_cndmask_b32,",
      sum(1 for op in body if op.startswith("ds_write")), "LDS writes")
PY
  echo "## copies at any alignment are single instructions:"
  grep -c "global_load_dwordx2" $T/inf.s | sed 's/^/global_load_dwordx2: /'; grep -c "global_store_dwordx2" $T/inf.s | sed 's/^/global_store_dwordx2: /'
  echo "## fences of the copy rounds (workgroup scope): s_waitcnt vmcnt(0) count in the kernel: $(grep -c 's_waitcnt vmcnt(0)' $T/inf.s); cache invalidates (buffer_inv): $(grep -c buffer_inv $T/inf.s)"
} > "$OUT"
rm -rf $T
cat "$OUT"
