# quick A/B: GPU parity subset + bench line (usage: bash tools/quick_bench.sh [bench args])
python -m pytest tests/test_s2m_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['final_pose']['t'])"
