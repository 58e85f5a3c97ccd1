#!/bin/bash
# Runs GPU steps one after another on the box: "name|seconds|command" per argument.  A step that
# ends with anything but 0 stops the chain: after a fault, an abort, a time limit or a failed test
# (which a GPU fault can look like) nothing more is started on the GPU in the same call.
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "[$name] rc=$rc"
  tail -n 4 "gpurun_out/$name.log"
  if [ $rc -ne 0 ]; then echo "stopping after $name (rc=$rc)"; exit $rc; fi
done
