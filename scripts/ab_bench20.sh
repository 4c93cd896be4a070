# same-box A/B of the driver's short run (bench.py --steps 20 --warmup 5): this build against round 3's library
for rep in 1 2 3; do
for v in main r03; do
  unset TR_LIBRARY
  [ $v = r03 ] && export TR_LIBRARY=scratch/r03/libtiny_renderer.so
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$v', 'ms_per_step', d['ms_per_step'], 'latency', d['latency_us'], 'per_frame', d['per_frame_protocol']['ms_per_step'], 'unfused', d['per_frame_protocol']['unfused_ms_per_step'], d['kernel_us_per_frame'])
"
done
done
