#!/bin/bash
# The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool):
# libmi_scene_host.so (scene plumbing, BVH builder, glb / Collada readers, scene blob, NIF assets, band dealing) built with
# -fsanitize=address,undefined and the CPU test modules that drive it run against that build.
#   tools/sanitize_host.sh [pytest args...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/san
H=$R/ipu_ray_lib_amd/csrc/host
g++ -std=c++17 -O1 -g -ffp-contract=off -fno-fast-math -fPIC -Wall -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -o $R/build/san/libmi_scene_host.so $H/bvh_sah.cpp $H/glb_reader.cpp $H/dae_reader.cpp $H/scene_builtin.cpp $H/scene_api.cpp $H/nif_assets.cpp -ldl
ln -sf $R/ipu_ray_lib_amd/libmi_nif_h5.so $R/build/san/libmi_nif_h5.so       # the HDF5 plugin is looked up beside the host library
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  MI_SCENE_HOST_LIB=$R/build/san/libmi_scene_host.so python -m pytest tests/test_host_and_abi.py tests/test_scene_blob.py tests/test_nif_assets.py tests/test_importers.py tests/test_sharding_gloo.py -x -q -m "not gpu" "$@"
