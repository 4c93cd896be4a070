# same-box A/B of the bench's long run: this build against scratch/<name> builds (scripts/make_variant.sh)
# usage: bash scripts/ab_variant.sh "bench args" name [name ...]
args="$1"; shift
for rep in 1 2; do
for v in main "$@"; do
  unset TR_LIBRARY
  [ $v != main ] && export TR_LIBRARY=scratch/$v/tiny_renderer_amd/lib/libtiny_renderer.so
  [ $v = r03 ] && export TR_LIBRARY=scratch/r03/libtiny_renderer.so
  python bench.py $args --no-cpu --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$v', 'ms_per_step', d['ms_per_step'], 'parity', d['parity_vs_oracle']['ok'], d['kernel_us_per_frame'])
"
done
done
