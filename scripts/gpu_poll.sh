#!/bin/bash
# Offers gpurun calls until a box answers (the GPU was closed to this repository for rounds 4-6; "refused" costs nothing).
# usage: scripts/gpu_poll.sh <timeout_s> "<stages of call 1>" ["<stages of call 2>" ...]
#   every argument after the timeout is one call of scripts/r06_validate.sh; the next one is offered once the one before it has RUN
#   (whatever its outcome: its output is under gpurun_out/r06v/).  Log: /tmp/gpu_poll.log.  An empty file /tmp/gpu_hold makes the
#   poller wait (set it while the tree is mid-edit: a call snapshots /root/repo as it is).
T=$1; shift
n=0
for STAGES in "$@"; do
  while true; do
    if [ -e /tmp/gpu_hold ]; then sleep 30; continue; fi
    n=$((n+1))
    /usr/local/graft/bin/gpurun --timeout $T -- "bash scripts/r06_validate.sh $STAGES" > /tmp/gpu_poll_last.txt 2>&1
    rc=$?
    st=$(python3 -c "import json;print(json.load(open('/root/repo/gpurun_out/.last_call.json')).get('status'))" 2>/dev/null)
    echo "$(date -u +%FT%TZ) offer $n stages=[$STAGES] rc=$rc status=$st" >> /tmp/gpu_poll.log
    if [ "$st" != "refused" ] && [ "$rc" != "2" ] && [ "$rc" != "3" ]; then cp /tmp/gpu_poll_last.txt "/tmp/gpu_poll_ran_$n.txt"; break; fi
    sleep 240
  done
done
