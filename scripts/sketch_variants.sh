# Timing experiment: builds debug variants of the sketch kernel (hash / table reads removed) next to the
# product library and times each with scripts/sketch_bench.py.  Run the build part here, the timing on the GPU box.
cd "$(dirname "$0")/../galah_amd/csrc"
for V in nohash nolds nohash_nolds noappend; do
  D=build/dbg_$V; mkdir -p $D
  FLAGS=""; case $V in nohash) FLAGS="-DGHIP_DBG_NOHASH";; nolds) FLAGS="-DGHIP_DBG_NOLDS";; nohash_nolds) FLAGS="-DGHIP_DBG_NOHASH -DGHIP_DBG_NOLDS";; noappend) FLAGS="-DGHIP_DBG_NOAPPEND";; esac
  if [ "$1" = build ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c sketch.hip -o $D/sketch.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgalah_hip.so build/api.o build/ingest.o build/cluster.o build/comm.o $D/sketch.o build/pairs.o build/pairs_probe.o build/pairs_join.o build/ani.o -lz -lpthread -ldl
  else
    echo "== $V"; GHIP_LIB_OVERRIDE=$PWD/$D/libgalah_hip.so python ../../scripts/sketch_bench.py 400 2>&1 | grep -v amdgpu.ids | grep "minhash\|fused"
  fi
done
