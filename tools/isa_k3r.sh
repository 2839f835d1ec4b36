#!/bin/bash
# Compile raylib.hip keeping the ISA of K3r (nif_regs_kernel<10, 8, 2, D>) under build/isa/k8.s and print its resource lines.
#   tools/isa_k3r.sh [extra hipcc flags]
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/isa && cd $R/build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function -I $R/include -DMI_RAYLIB_VARIANTS=1 -DMI_NIF_REGS_ONLY10 -mllvm -pragma-unroll-threshold=10000000 "$@" -save-temps=obj -o libtest.so $R/ipu_ray_lib_amd/csrc/raylib.hip 2>&1 | grep -E "error" | head
S=raylib-hip-amdgcn-amd-amdhsa-gfx950.s
k="ILj10ELj8ELj2ELj4ELb${STG:-0}E"
a=$(grep -n "^_ZN2mi15nif_regs_kernel$k" $S | head -1 | cut -d: -f1); b=$(grep -n "\.amdhsa_kernel _ZN2mi15nif_regs_kernel$k" $S | head -1 | cut -d: -f1)
sed -n "${a},${b}p" $S > k8.s
awk -v a=$b 'NR>a && NR<a+120' $S | grep -E "ScratchSize|codeLen|next_free_vgpr|next_free_sgpr" | tr '\n' ' '; echo
echo "vector state (s_mov m0, v): $(grep -c "s_mov_b32 m0, v" k8.s)"; echo "scratch ops: $(grep -c scratch_ k8.s)  mfma: $(grep -c v_mfma k8.s)  readlane/writelane: $(grep -c 'v_readlane\|v_writelane' k8.s)"
[ -n "$KEEP" ] || rm -f *.bc *.hipi *.o *.out libtest.so
