#!/bin/bash
# Runs the GPU steps listed in a file, one per line: "<seconds> <name> <command...>".
# Each step runs under its own time limit with its output in gpurun_out/<name>.log.  A step
# that fails with an ordinary error (exit 1..123) is reported and the next one runs; a time
# limit, abort, segmentation fault or kill (exit >= 124) ends the session: nothing more is
# started on the GPU after it.
steps=${1:?steps file}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
worst=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  secs=${line%% *}; rest=${line#* }
  name=${rest%% *}; cmd=${rest#* }
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== stopping: $name ended with $rc"; exit $rc; fi
  [ $rc -ne 0 ] && worst=$rc
done < "$steps"
exit $worst
