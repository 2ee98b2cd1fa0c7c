#!/bin/bash
# same-box A/B of alternative builds (tools/_lib_*.so) on tools/k1bench.py: usage ab_k1.sh "M..." tag:lib ...
Ms=$1; shift
for round in 1 2; do
  for tl in "$@"; do
    tag=${tl%%:*}; lib=${tl#*:}
    if [ "$lib" = "default" ]; then python tools/k1bench.py $tag $Ms 2>&1 | grep k1bench
    else VLPET_LIB=$lib python tools/k1bench.py $tag $Ms 2>&1 | grep k1bench; fi
  done
done
