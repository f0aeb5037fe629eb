#!/bin/bash
# Runs the GPU steps given as arguments ("name|seconds|command") one after another from the
# repository root; output of each into gpurun_out/<name>.log.  An ordinary failure (exit status 1
# to 5: a failed assertion, a bad argument) is reported and the next step still runs; a time
# limit, an abort, a segmentation fault or any other death by signal ends the script there, so
# that nothing more is started on a GPU that may be in trouble.
mkdir -p gpurun_out
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"
  tail -n 4 "gpurun_out/$name.log"
  if [ $rc -gt 5 ]; then echo "== stopping after $name"; exit $rc; fi
done
exit 0
