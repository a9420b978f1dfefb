#!/bin/bash
# round 3, GPU run 8: local-map commit after the dense centroid / small tiles / no super rows for small maps; full suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03h; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -8 $OUT/pytest_all.log
sed -n '/lm_loop.py <<PY/,/^PY$/p' tools/r03_run7.sh | sed '1d;$d' > /tmp/lm_loop.py
for inc in 1 0; do python /tmp/lm_loop.py $inc; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lmloop -- python /tmp/lm_loop.py 1 > /dev/null 2> $OUT/lmloop.err
python tools/kstats.py $OUT/lmloop_kernel_stats.csv | head -24
