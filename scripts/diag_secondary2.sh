for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-scoring --no-drop-in 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print('bench run: headline %.3f secondary graph %.3f eager %.3f' % (d['ms_per_step'], s['graph_ms_per_step'], s['eager_ms_per_step']))"
done
