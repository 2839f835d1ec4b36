cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
cp ipu_ray_lib_amd/libmi_raylib.so /tmp/new.so
for r in 1 2 3 4; do
  for v in prev new; do
    if [ $v = prev ]; then cp ipu_ray_lib_amd/libmi_raylib_prev.so ipu_ray_lib_amd/libmi_raylib.so; else cp /tmp/new.so ipu_ray_lib_amd/libmi_raylib.so; fi
    python bench.py --spp 300 --steps 4 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e9,3), round(d['ms_per_step'],1))"
  done
done
cp /tmp/new.so ipu_ray_lib_amd/libmi_raylib.so
