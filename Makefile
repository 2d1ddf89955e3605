# Builds the product library (HIP, gfx950 only) and the CPU oracle (test infrastructure).
HIPCC   ?= hipcc
ARCH    ?= gfx950
CSRC    := fsae-mpc_amd/csrc
LIBDIR  := fsae-mpc_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function

all: $(LIBDIR)/libfsaempc.so oracle

$(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/qp_solver.h $(CSRC)/ltv_build.h include/fsaempc.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libfsaempc.so: $(LIBDIR)/qp_solver.o $(LIBDIR)/ltv_build.o $(LIBDIR)/capi.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR)/*.o $(LIBDIR)/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
