# Timing experiment: the same kernel binaries at fewer workgroups per CU (unused dynamic LDS, GHIP_DBG_EXTRA_LDS).
cd "$(dirname "$0")/../galah_amd/csrc"
for V in ${VARIANTS:-base noappend}; do
  for E in ${EXTRAS:-0 6000 12000 18000 30000 50000}; do
    echo "== $V extra_lds=$E"; GHIP_DBG_EXTRA_LDS=$E GHIP_LIB_OVERRIDE=$PWD/build/dbg_$V/libgalah_hip.so python ../../scripts/sketch_bench.py 400 2>&1 | grep "minhash\|fused" | sed 's/k-mer pass.*//'
  done
done
