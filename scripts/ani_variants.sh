# Timing experiment: debug variants of ani_pairs (parts of the join removed; wrong results) timed on the bench workload's
# 4 500 precluster pairs.  `build` here, run on the GPU box.
cd "$(dirname "$0")/../galah_amd/csrc"
for V in ${VARIANTS:-base norvotes nocompare all_off}; do
  D=build/dbg_ani_$V; mkdir -p $D
  case $V in base|old) F="";; norvotes) F="-DGHIP_DBG_ANI_NORVOTES";; nocompare) F="-DGHIP_DBG_ANI_NOCOMPARE";; phases) F="-DGHIP_DBG_ANI_PHASES";;
    all_off) F="-DGHIP_DBG_ANI_NORVOTES -DGHIP_DBG_ANI_NOCOMPARE";; esac
  if [ "$1" = build ]; then
    SRC=ani.hip
    if [ $V = old ]; then git show HEAD:galah_amd/csrc/ani.hip > ani_dbg_old.hip; SRC=ani_dbg_old.hip; fi   # the last commit's kernel, for an A/B on one box
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $F -c $SRC -o $D/ani.o 2>/dev/null; rm -f ani_dbg_old.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgalah_hip.so build/api.o build/ingest.o build/cluster.o build/comm.o build/sketch.o build/pairs.o build/pairs_probe.o build/pairs_join.o $D/ani.o -lz -lpthread -ldl
  else
    echo -n "== $V: "; GHIP_LIB_OVERRIDE=$PWD/$D/libgalah_hip.so python ../../scripts/ani_pairs_bench.py 2>&1 | grep "ani_pairs" | tail -3
  fi
done
