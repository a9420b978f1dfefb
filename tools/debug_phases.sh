for d in 0 1 2 3; do
LILI_DEBUG=$d python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('DEBUG $d', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"
done
