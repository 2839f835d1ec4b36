cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do
  for t in "12,2,0" "12,2,1"; do
    MI_RAYLIB_TUNE=5,8,12,32,2,16,4,48,$t python bench.py --spp 300 --steps 4 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbl,maxExtra,leafThenNode $t', round(d['value']/1e9,3), round(d['ms_per_step'],1))"
  done
done
