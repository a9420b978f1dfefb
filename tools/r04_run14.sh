#!/bin/bash
export TMPDIR=/tmp
for lib in ${LIBS:-v1 v2 v3}; do
  echo -n "lib=$lib nn_cache=0: "; LILI_HIP_LIBRARY=$PWD/tools/_probe/libs/lib_$lib.so timeout 300 python bench.py --assoc-only 300 --assoc-after 10 --opt nn_cache=0 --no-cpu-baseline --no-extras 2>/dev/null | tail -1
done
