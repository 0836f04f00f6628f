# Build of the MI355X (gfx950) wavefront-alignment library and the CPU oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
# wfa_duo_kernel's translation unit: LLVM's atomic optimizer off (wfa_amd/csrc/wfa_duo_cfg.hpp says why)
DUOFLAGS ?= -mllvm -amdgpu-atomic-optimizer-strategy=None
LIB     := wfa_amd/lib/libwfahip.so
SRC     := wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp
DUOSRC  := wfa_amd/csrc/wfa_duo.hip
DUOOBJ  := build/obj/wfa_duo.o
HDR     := $(wildcard wfa_amd/csrc/*.hpp) include/wfa_hip.h

all: $(LIB) oracle

OBJDIR  := build/obj
OBJS    := $(OBJDIR)/wfa_host.o $(OBJDIR)/wfa_gen.o $(OBJDIR)/wfa_multi.o $(DUOOBJ)

$(DUOOBJ): $(DUOSRC) $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(DUOFLAGS) -c -o $@ $(DUOSRC)

$(OBJDIR)/wfa_host.o: wfa_amd/csrc/wfa_host.hip $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJDIR)/wfa_gen.o: wfa_amd/csrc/wfa_gen.cpp $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJDIR)/wfa_multi.o: wfa_amd/csrc/wfa_multi.cpp $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	@mkdir -p wfa_amd/lib
	$(HIPCC) -fPIC --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle -s

asm: $(SRC) $(DUOSRC) $(HDR)
	@mkdir -p build/asm
	$(HIPCC) $(HIPFLAGS) -save-temps=obj -c -o build/asm/wfa_host.o wfa_amd/csrc/wfa_host.hip -Rpass-analysis=kernel-resource-usage 2> build/asm/resource_usage.txt || true
	$(HIPCC) $(HIPFLAGS) $(DUOFLAGS) -save-temps=obj -c -o build/asm/wfa_duo.o wfa_amd/csrc/wfa_duo.hip -Rpass-analysis=kernel-resource-usage 2>> build/asm/resource_usage.txt || true

clean:
	rm -rf $(LIB) build
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean
