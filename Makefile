# Builds the product library (HIP, gfx950 only) and the CPU oracle (test infrastructure).
HIPCC   ?= hipcc
ARCH    ?= gfx950
CSRC    := fsae-mpc_amd/csrc
LIBDIR  := fsae-mpc_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function

all: $(LIBDIR)/libfsaempc.so oracle o1 dbg

# Every device translation unit is compiled through tools/hipcc_checked.sh: device code via assembly text, which is checked (and,
# if need be, repaired) for the compiler defect of DESIGN.md "Build-variant fragility: root cause" before it is assembled; the
# report of each unit stays next to its object (*.isa.log; `make isa-report` prints them).
CC_CHECKED := HIPCC=$(HIPCC) tools/hipcc_checked.sh
CHECKDEPS := tools/hipcc_checked.sh tools/check_isa_exec_prologue.py

$(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/qp_solver.h $(CSRC)/ltv_build.h $(CSRC)/reference.h $(CSRC)/plant.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(HIPFLAGS)

# WGFLAGS (both solve kernels): without machine LICM and with sinking-to-avoid-spills the T = 8 workgroup kernel spills 20 VGPRs
# instead of 166 (256 are its budget at two waves per SIMD) and the one-wavefront kernels need no scratch at all (T = 5: 140..504
# bytes per lane before): hoisted address arithmetic no longer lives across the whole iteration loop.
WGFLAGS := -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills=1

# qp_solver.hip is compiled as six translation units (see the note in the file): main + the tile counts T = 1..4, 5, 6, 7
QPOBJ := $(LIBDIR)/qp_solver_tu0.o $(LIBDIR)/qp_solver_tu1.o $(LIBDIR)/qp_solver_tu2.o $(LIBDIR)/qp_solver_tu3.o $(LIBDIR)/qp_solver_tu4.o
$(LIBDIR)/qp_solver_tu%.o: $(CSRC)/qp_solver.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(HIPFLAGS) $(WGFLAGS) -DQP_TU=$*

# qp_wg.hip (workgroup-per-QP solve kernel, tile counts T = 1..12) is compiled once per range of tile counts (lo_hi)
WGOBJ := $(LIBDIR)/qp_wg_1_5.o $(LIBDIR)/qp_wg_6_6.o $(LIBDIR)/qp_wg_7_7.o $(LIBDIR)/qp_wg_8_8.o $(LIBDIR)/qp_wg_9_9.o $(LIBDIR)/qp_wg_10_10.o $(LIBDIR)/qp_wg_11_11.o $(LIBDIR)/qp_wg_12_12.o
$(LIBDIR)/qp_wg_%.o: $(CSRC)/qp_wg.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(HIPFLAGS) $(WGFLAGS) -DQP_WG_TLO=$(word 1,$(subst _, ,$*)) -DQP_WG_THI=$(word 2,$(subst _, ,$*))

# host-side track pipeline (no device code)
$(LIBDIR)/track.o: $(CSRC)/track.cpp include/fsaempc.h
	@mkdir -p $(LIBDIR)
	g++ -O2 -std=c++17 -fPIC -Wall -Iinclude -c $< -o $@

COMMON := $(LIBDIR)/ltv_build.o $(LIBDIR)/reference.o $(LIBDIR)/plant.o $(LIBDIR)/capi.o $(LIBDIR)/track.o
$(LIBDIR)/libfsaempc.so: $(QPOBJ) $(WGOBJ) $(COMMON)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

# development builds of the workgroup kernel: ONE instantiation (T = 8 + slack border: BASELINE configs[2], dynamic N = 60) linked with the
# shipped objects of everything else; plain and with phase stamps.  Select with FSAEMPC_LIB.
WGDEVFLAGS := $(WGFLAGS) -DQP_WG_TLO=8 -DQP_WG_THI=8 -DQP_WG_ONLY_NB=4
$(LIBDIR)/wgdev_qp_wg.o: $(CSRC)/qp_wg.hip $(CSRC)/qp_solver.h include/fsaempc.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(WGDEVFLAGS) -c $< -o $@
$(LIBDIR)/wgdevst_qp_wg.o: $(CSRC)/qp_wg.hip $(CSRC)/qp_solver.h include/fsaempc.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(WGDEVFLAGS) -DQP_STAMPS=1 -c $< -o $@
WGDEVREST := $(QPOBJ) $(filter-out $(LIBDIR)/qp_wg_8_8.o,$(WGOBJ)) $(COMMON)
wgdev: $(LIBDIR)/libfsaempc_wgdev.so $(LIBDIR)/libfsaempc_wgdevst.so
$(LIBDIR)/libfsaempc_wgdev.so: $(LIBDIR)/wgdev_qp_wg.o $(WGDEVREST)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^
$(LIBDIR)/libfsaempc_wgdevst.so: $(LIBDIR)/wgdevst_qp_wg.o $(WGDEVREST)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

# diagnostic library with the in-kernel dump hooks of both kernels (tests/test_gpu_parity.py::test_01_normal_matrix_dump_matches_numpy); T <= 5
DBGOBJ := $(LIBDIR)/dbg_qp_solver_tu0.o $(LIBDIR)/dbg_qp_solver_tu1.o $(LIBDIR)/dbg_qp_solver_tu2.o $(LIBDIR)/dbg_qp_wg_1_5.o
$(LIBDIR)/dbg_qp_wg_%.o: $(CSRC)/qp_wg.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(HIPFLAGS) $(WGFLAGS) -DQP_DEBUG_DUMP -DQP_WG_TLO=$(word 1,$(subst _, ,$*)) -DQP_WG_THI=$(word 2,$(subst _, ,$*))
$(LIBDIR)/dbg_qp_solver_tu%.o: $(CSRC)/qp_solver.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(HIPFLAGS) $(WGFLAGS) -DQP_DEBUG_DUMP -DQP_TU=$*
dbg: $(LIBDIR)/libfsaempc_dbg.so
$(LIBDIR)/libfsaempc_dbg.so: $(DBGOBJ) $(LIBDIR)/qp_solver_tu3.o $(LIBDIR)/qp_solver_tu4.o $(filter-out $(LIBDIR)/qp_wg_1_5.o,$(WGOBJ)) $(COMMON)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

# guard build: the solver sources at -O1, every instantiated (T, NB) (tests/test_gpu_parity.py::test_shipped_build_matches_O1_build
# compares the two builds on the GPU; see DESIGN.md "Build-variant fragility")
O1FLAGS := --offload-arch=$(ARCH) -O1 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wno-unused-function $(WGFLAGS)
O1OBJ := $(LIBDIR)/o1_qp_solver_tu0.o $(LIBDIR)/o1_qp_solver_tu1.o $(LIBDIR)/o1_qp_solver_tu2.o $(LIBDIR)/o1_qp_solver_tu3.o $(LIBDIR)/o1_qp_solver_tu4.o \
         $(LIBDIR)/o1_qp_wg_1_5.o $(LIBDIR)/o1_qp_wg_6_6.o $(LIBDIR)/o1_qp_wg_7_7.o $(LIBDIR)/o1_qp_wg_8_8.o $(LIBDIR)/o1_qp_wg_9_9.o $(LIBDIR)/o1_qp_wg_10_10.o \
         $(LIBDIR)/o1_qp_wg_11_11.o $(LIBDIR)/o1_qp_wg_12_12.o
$(LIBDIR)/o1_qp_solver_tu%.o: $(CSRC)/qp_solver.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(O1FLAGS) -DQP_TU=$*
# (the guard's workgroup kernels take the unpipelined variant of pass 1, -DQP_WG_NOPIPE: an independent code path for the check,
#  and the -O1 build of the pipelined one walks wrong iterates on kinematic N = 64 -- DESIGN.md 5c, open)
$(LIBDIR)/o1_qp_wg_%.o: $(CSRC)/qp_wg.hip $(CSRC)/qp_solver.h include/fsaempc.h $(CHECKDEPS)
	@mkdir -p $(LIBDIR)
	$(CC_CHECKED) $@ $< $(O1FLAGS) -DQP_WG_NOPIPE -DQP_WG_TLO=$(word 1,$(subst _, ,$*)) -DQP_WG_THI=$(word 2,$(subst _, ,$*))
o1: $(LIBDIR)/libfsaempc_O1.so
$(LIBDIR)/libfsaempc_O1.so: $(O1OBJ) $(COMMON)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

isa-report:
	@cat $(LIBDIR)/*.isa.log

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR)/*.o $(LIBDIR)/*.so $(LIBDIR)/*.isa.log $(LIBDIR)/*.s
	$(MAKE) -C oracle clean
.PHONY: all oracle clean o1 wgdev dbg isa-report
