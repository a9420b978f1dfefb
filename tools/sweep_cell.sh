mkdir -p gpurun_out
for cfg in "1 0" "2 55" "2 60" "2 65" "2 70" "2 75" "2 80"; do set -- $cfg
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --reach $1 --cell-pct $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('reach $1 pct $2:', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['final_pose']['t'])"
done
