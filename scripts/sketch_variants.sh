# Timing experiment: builds debug variants of the sketch kernel (hash / table reads / seed appends removed, ...) next to
# the product library and times each with scripts/sketch_bench.py.  Run the build part here (`build`), the timing on the
# GPU box.  `old` = the kernel sources of the last commit, for an A/B on the same box.
cd "$(dirname "$0")/../galah_amd/csrc"
VARIANTS="${VARIANTS:-old base ifcvt noappend nohash nolds}"
for V in $VARIANTS; do
  D=build/dbg_$V; mkdir -p $D
  FLAGS=""; case $V in nohash) FLAGS="-DGHIP_DBG_NOHASH";; nolds) FLAGS="-DGHIP_DBG_NOLDS";; nohash_nolds) FLAGS="-DGHIP_DBG_NOHASH -DGHIP_DBG_NOLDS";; noappend) FLAGS="-DGHIP_DBG_NOAPPEND";;
     ifcvt) FLAGS="-DGHIP_DBG_IFCVT";; esac
  if [ "$1" = build ]; then
    SRC=.
    if [ $V = old ]; then
      SRC=$D/src; mkdir -p $SRC
      for f in sketch.hip ani.hip seed_common.h murmur21_asm.h ghip_internal.h; do git show HEAD:galah_amd/csrc/$f > $SRC/$f; done
      sed -i 's#"../../include/galah_hip.h"#"../../../../../include/galah_hip.h"#' $SRC/ghip_internal.h
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c $SRC/ani.hip -o $D/ani.o 2>/dev/null
    else cp build/ani.o $D/ani.o; fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -I../../include -c $SRC/sketch.hip -o $D/sketch.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgalah_hip.so build/api.o build/ingest.o build/cluster.o build/comm.o $D/sketch.o build/pairs.o build/pairs_probe.o build/pairs_join.o $D/ani.o -lz -lpthread -ldl
  else
    echo "== $V"; GHIP_LIB_OVERRIDE=$PWD/$D/libgalah_hip.so python ../../scripts/sketch_bench.py 400 2>&1 | grep -v amdgpu.ids | grep "minhash\|fused"
  fi
done
