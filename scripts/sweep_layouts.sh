# Tile layouts (waves per tile x column / shared resolve) under frame groups, per workload: which one the automatic
# choice (tile_layout, tr_scene.cpp) should make.  usage: bash scripts/sweep_layouts.sh
run() { # size grid model pipe
  for wm in "0 0" "4 1" "4 2" "8 1" "8 2"; do set -- $1 $2 $3 $4 $wm
    echo -n "WAVES=$5 MODE=$6 : "; WAVES=$5 MODE=$6 FRAMES=800 python scripts/probe_groups.py $1 $2 $3 $4 2>&1 | grep "groups of" | sed 's/  */ /g'
  done; }
run 800 1 african_head default
run 2048 1 diablo phong
run 4096 1 diablo phong
run 1024 1 diablo phong
