# The driver's short run (bench.py --steps 20 --warmup 5) under different group plans (experiment hooks of tr_scene.cpp)
run() { python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['kernel_us_per_frame'].get('k_tile'))
"; }
for rep in 1 2; do
for sg in 1 2 3 4 6; do TR_SHORT_GROUPS=$sg run "short_factor=$sg"; done
for g in 5 7 10 20; do TR_GROUP=$g run "fixed_group=$g"; done
done
