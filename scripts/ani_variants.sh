# Timing experiment: debug variants of ani_pairs (parts of the join removed; wrong results) timed on the bench workload's
# 4 500 precluster pairs.  `build` here, run on the GPU box.
cd "$(dirname "$0")/../galah_amd/csrc"
for V in ${VARIANTS:-base norvotes nocompare all_off}; do
  D=build/dbg_ani_$V; mkdir -p $D
  case $V in base) F="";; norvotes) F="-DGHIP_DBG_ANI_NORVOTES";; nocompare) F="-DGHIP_DBG_ANI_NOCOMPARE";;
    all_off) F="-DGHIP_DBG_ANI_NORVOTES -DGHIP_DBG_ANI_NOCOMPARE";; esac
  if [ "$1" = build ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $F -c ani.hip -o $D/ani.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgalah_hip.so build/api.o build/ingest.o build/cluster.o build/comm.o build/sketch.o build/pairs.o build/pairs_probe.o build/pairs_join.o $D/ani.o -lz -lpthread -ldl
  else
    echo -n "== $V: "; GHIP_LIB_OVERRIDE=$PWD/$D/libgalah_hip.so python ../../scripts/ani_pairs_bench.py 2>&1 | grep "ani_pairs"
  fi
done
