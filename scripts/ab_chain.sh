# same-box A/B of the chain in front of a tile kernel: this build (two setup streams / one) against round 3's library
for v in main one_stream r03; do
  unset TR_LIBRARY TR_SETUP_STREAMS
  case $v in
    one_stream) export TR_SETUP_STREAMS=1;;
    r03) export TR_LIBRARY=scratch/r03/libtiny_renderer.so;;
  esac
  echo "== $v"
  python scripts/probe_latency.py 4096 phong 2>&1 | tail -1
  python scripts/probe_latency.py 800 default african_head 2>&1 | tail -1
  python scripts/probe_unfused.py 4096 phong 2>&1 | tail -2
  python scripts/probe_unfused.py 4096 shadow 2>&1 | tail -2
  python scripts/probe_unfused.py 2048 phong 2>&1 | tail -2
done
