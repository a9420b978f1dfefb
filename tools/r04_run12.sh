#!/bin/bash
export TMPDIR=/tmp
for nc in 0 1; do
  echo "nn_cache=$nc assoc-only at the converged pose:"; timeout 300 python bench.py --assoc-only 300 --assoc-after 10 --opt nn_cache=$nc --no-cpu-baseline --no-extras 2>/dev/null | tail -1
done
