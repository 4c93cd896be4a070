# same-box A/B of any command: this build against scratch/<name> builds (scripts/make_variant.sh)
# usage: bash scripts/ab_cmd.sh "command" name [name ...]     (prints the command's last two lines per build, two repetitions)
cmd="$1"; shift
for rep in 1 2; do
for v in main "$@"; do
  unset TR_LIBRARY
  [ $v != main ] && export TR_LIBRARY=scratch/$v/tiny_renderer_amd/lib/libtiny_renderer.so
  echo "== $v"
  bash -c "$cmd" 2>&1 | tail -n 2
done
done
