cd $GRAFT_REPO_ROOT
cp ipu_ray_lib_amd/libmi_raylib.so /tmp/new.so
for r in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then cp ipu_ray_lib_amd/libmi_raylib_prev.so ipu_ray_lib_amd/libmi_raylib.so; else cp /tmp/new.so ipu_ray_lib_amd/libmi_raylib.so; fi
    python bench.py --spp 400 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/new.so ipu_ray_lib_amd/libmi_raylib.so
