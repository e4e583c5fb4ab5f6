#!/bin/bash
# Run GPU steps one after another on the gpurun box; stop at the first step that was KILLED (timeout / signal) --
# an assertion failure does not stop the sequence.  usage: gpu_steps.sh "<secs> <logname> <command...>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    set -- $spec
    secs=$1; name=$2; shift 2
    echo "=== $name: $*" | tee -a gpurun_out/steps.log
    timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    rc=$?
    echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log
    tail -n 3 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "killed: stopping"; exit $rc; fi
done
exit 0
