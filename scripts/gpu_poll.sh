#!/bin/bash
# Offers one gpurun call until a box answers (the GPU was closed to this repository for rounds 4-5; "refused" costs nothing).
# usage: scripts/gpu_poll.sh <timeout_s> <stages...>   -- exits 0 after the first call that actually ran; log: /tmp/gpu_poll.log
# An empty file /tmp/gpu_hold makes the poller wait (set it while the tree is mid-edit).
T=$1; shift
STAGES="$*"
n=0
while true; do
  if [ -e /tmp/gpu_hold ]; then sleep 30; continue; fi
  n=$((n+1))
  /usr/local/graft/bin/gpurun --timeout $T -- "bash scripts/r06_validate.sh $STAGES" > /tmp/gpu_poll_last.txt 2>&1
  rc=$?
  st=$(python3 -c "import json;print(json.load(open('/root/repo/gpurun_out/.last_call.json')).get('status'))" 2>/dev/null)
  echo "$(date -u +%FT%TZ) offer $n stages=[$STAGES] rc=$rc status=$st" >> /tmp/gpu_poll.log
  if [ "$st" != "refused" ] && [ "$rc" != "2" ] && [ "$rc" != "3" ]; then exit 0; fi
  sleep 240
done
