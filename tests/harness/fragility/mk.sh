# usage: mk.sh <opt> <tag> [extra flags...]  -> libfrag_<tag>.so
opt=$1; tag=$2; shift 2
F="--offload-arch=gfx950 -std=c++17 -fPIC -I../../include -I../../fsae-mpc_amd/csrc -DQP_ONLY_T=${FRAG_T:-5} -DQP_FRAGILE -Wno-everything"
L=../../fsae-mpc_amd/lib
hipcc $F $opt "$@" -c ../../fsae-mpc_amd/csrc/qp_solver.hip -o f_$tag.o 2> f_$tag.err || { echo "compile FAIL $tag"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o libfrag_$tag.so f_$tag.o $L/qp_wg_1_5.o $L/qp_wg_6_6.o $L/qp_wg_7_7.o $L/qp_wg_8_8.o $L/qp_wg_9_9.o $L/qp_wg_10_10.o $L/qp_wg_11_11.o $L/qp_wg_12_12.o $L/track.o $L/ltv_build.o $L/reference.o $L/plant.o $L/capi.o || echo "link FAIL $tag"
rm -f f_$tag.o
