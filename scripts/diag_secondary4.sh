for i in 1 2; do
python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print('default bench run: headline %.3f secondary graph %.3f eager %.3f scoring %.1f M' % (d['ms_per_step'], s['graph_ms_per_step'], s['eager_ms_per_step'], d['scoring']['value']/1e6))"
done
