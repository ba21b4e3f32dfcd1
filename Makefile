# Build libalmpc.so (gfx950 HIP kernels + C ABI) and the oracle's C restatement.
# No cmake: hipcc and gcc directly.  `make` builds both; `make lib` / `make oracle` build one; `make -j8` compiles the library's
# translation units in parallel (one per kernel family + the C ABI: csrc/almpc_tu_*.hip, csrc/almpc_api.hip).
# `make unity` builds the same library from ONE translation unit (lib/libalmpc_unity.so; tools/gen_instances.py reads the
# instantiation lists off it), `make stamps` the diagnostic -DALMPC_STAMPS build (lib/libalmpc_stamps.so, unity as well).
PKG      := automationlabsmodelpredictivecontrol.jl_amd
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CS       := $(PKG)/csrc
LIB      := $(PKG)/lib/libalmpc.so
OBJDIR   := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-cuda-compat
ORACLE   := oracle/_build/libalmpc_oracle.so  # generic name; oracle/c_oracle.py builds a per-CPU copy itself

TUS      := api tu_step tu_polish_gen tu_instance tu_design_a tu_design_b tu_sdual_a tu_sdual_b tu_sdual_c
OBJS     := $(patsubst %,$(OBJDIR)/almpc_%.o,$(TUS))
H_K      := $(CS)/almpc_kernels.hip.h
H_ALL    := $(wildcard $(CS)/*.h) $(wildcard $(CS)/instances/*.inc) include/almpc.h

all: lib oracle
lib: $(LIB)
oracle: $(ORACLE)

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined -o $@ $(OBJS)

# every translation unit depends on exactly the headers it reads
$(OBJDIR)/almpc_api.o:           $(H_ALL)
$(OBJDIR)/almpc_tu_step.o:       $(H_K) $(CS)/instances/step.inc
$(OBJDIR)/almpc_tu_polish_gen.o: $(H_K) $(CS)/almpc_polish_gen.hip.h $(CS)/instances/polish_gen.inc
$(OBJDIR)/almpc_tu_instance.o:   $(H_K) $(CS)/almpc_instance.hip.h $(CS)/almpc_design.hip.h $(CS)/almpc_fnn.hip.h $(CS)/instances/instance.inc
$(OBJDIR)/almpc_tu_design_a.o:   $(H_K) $(CS)/almpc_instance.hip.h $(CS)/almpc_design.hip.h $(CS)/almpc_fnn.hip.h $(CS)/instances/design_a.inc
$(OBJDIR)/almpc_tu_design_b.o:   $(H_K) $(CS)/almpc_design.hip.h $(CS)/almpc_riccati.hip.h $(CS)/almpc_fnn.hip.h $(CS)/instances/design_b.inc
$(OBJDIR)/almpc_tu_sdual_a.o:    $(H_K) $(CS)/almpc_riccati.hip.h $(CS)/almpc_sdual.hip.h $(CS)/instances/sdual_a.inc
$(OBJDIR)/almpc_tu_sdual_b.o:    $(H_K) $(CS)/almpc_riccati.hip.h $(CS)/almpc_sdual.hip.h $(CS)/instances/sdual_b.inc
$(OBJDIR)/almpc_tu_sdual_c.o:    $(H_K) $(CS)/almpc_riccati.hip.h $(CS)/almpc_sdual.hip.h $(CS)/instances/sdual_c.inc

$(OBJDIR)/almpc_%.o: $(CS)/almpc_%.hip
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

unity: $(PKG)/lib/libalmpc_unity.so
$(PKG)/lib/libalmpc_unity.so: $(CS)/almpc_api.hip $(H_ALL)
	$(HIPCC) $(HIPFLAGS) -shared -DALMPC_UNITY -o $@ $<
stamps: $(PKG)/lib/libalmpc_stamps.so
$(PKG)/lib/libalmpc_stamps.so: $(CS)/almpc_api.hip $(H_ALL)
	$(HIPCC) $(HIPFLAGS) -shared -DALMPC_UNITY -DALMPC_STAMPS -o $@ $<

$(ORACLE): oracle/almpc_oracle.c
	@mkdir -p $(dir $@)
	gcc -O3 -march=native -fopenmp -fPIC -shared -Wall -o $@ $< -lm

clean:
	rm -rf $(LIB) $(ORACLE) $(OBJDIR)
.PHONY: all lib oracle unity stamps clean
