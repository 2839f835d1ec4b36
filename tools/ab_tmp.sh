cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do
  for t in "5,8,12,32,2" "4,8,12,32,2" "6,8,12,32,2" "3,8,12,32,2" "5,8,12,64,2" "5,8,12,32,1" "5,8,12,16,2" "5,12,16,32,2" "5,6,8,32,2"; do
    MI_RAYLIB_TUNE=$t,16,4,48,6,5,1 python bench.py --spp 300 --steps 4 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tune $t', round(d['value']/1e9,3), round(d['ms_per_step'],1))"
  done
done
