# usage: asm2lib.sh <device.s> <tag> [opt]: assembles the device code, bundles it with a host object of the same source -> libfrag_<tag>.so
s=$1; tag=$2; opt=${3:--O2}
B=/opt/rocm/lib/llvm/bin
F="--offload-arch=gfx950 -std=c++17 -fPIC -I../../include -I../../fsae-mpc_amd/csrc -DQP_ONLY_T=${FRAG_T:-5} -DQP_FRAGILE -Wno-everything"
L=../../fsae-mpc_amd/lib
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $s -o a_$tag.o || exit 1
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o a_$tag.out a_$tag.o || exit 1
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=a_$tag.out -output=a_$tag.hipfb || exit 1
hipcc $F $opt --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang a_$tag.hipfb -c ../../fsae-mpc_amd/csrc/qp_solver.hip -o f_$tag.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o libfrag_$tag.so f_$tag.o $L/qp_wg_1_5.o $L/qp_wg_6_6.o $L/qp_wg_7_7.o $L/qp_wg_8_8.o $L/qp_wg_9_9.o $L/qp_wg_10_10.o $L/qp_wg_11_11.o $L/qp_wg_12_12.o $L/track.o $L/ltv_build.o $L/reference.o $L/plant.o $L/capi.o || exit 1
rm -f a_$tag.o a_$tag.out a_$tag.hipfb f_$tag.o
