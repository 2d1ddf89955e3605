# development: one-unit build of qp_solver.hip for a single tile count (default T = 5) through the checked pipeline, linked with the
# shipped objects of everything else -> fsae-mpc_amd/lib/libfsaempc_exp.so (FSAEMPC_LIB selects it).  usage: tools/build_exp.sh [T] [extra flags]
set -e
T=${1:-5}; shift || true
L=fsae-mpc_amd/lib
HIPCC=hipcc tools/hipcc_checked.sh $L/exp_qp_solver.o fsae-mpc_amd/csrc/qp_solver.hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ifsae-mpc_amd/csrc -Wno-unused-function -DQP_ONLY_T=$T "$@"
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfsaempc_exp.so $L/exp_qp_solver.o $L/qp_wg_1_5.o $L/qp_wg_6_6.o $L/qp_wg_7_7.o $L/qp_wg_8_8.o $L/qp_wg_9_9.o $L/qp_wg_10_10.o $L/qp_wg_11_11.o $L/qp_wg_12_12.o $L/track.o $L/ltv_build.o $L/reference.o $L/plant.o $L/capi.o
