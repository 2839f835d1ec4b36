timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "test_path_trace_bit_exact and 4" > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
for t in "4,48" "1,48" "8,48" "4,32" "4,64" "8,32" "2,24"; do echo "qPush,qServe=$t"; MI_RAYLIB_KERNEL=4 MI_RAYLIB_TUNE=5,8,12,32,2,16,$t timeout -k 10 120 python bench.py --spp 200 --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
MI_RAYLIB_KERNEL=4 timeout -k 10 120 python tests/phase_probe.py
