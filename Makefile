# Build of the MI355X (gfx950) wavefront-alignment library and the CPU oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
# wfa_duo_kernel's translation unit: LLVM's atomic optimizer off (wfa_amd/csrc/wfa_duo_cfg.hpp says why)
DUOFLAGS ?= -mllvm -amdgpu-atomic-optimizer-strategy=None
LIB     := wfa_amd/lib/libwfahip.so
CSRC    := wfa_amd/csrc
HDR     := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inc) include/wfa_hip.h
OBJDIR  := build/obj
# one translation unit per penalty shape of the sub-wave forward kernels (wfa_fwd.hpp), one for wfa_duo_kernel, one for the
# long-pair kernels, three for the host side (router, host entries, debug aids: wfa_ctx.hpp): they compile side by side (make -j)
SHAPES  := s24 s13 s12 s23 s22 s33
UNITS   := wfa_host wfa_entry wfa_debug wfa_long wfa_duo $(addprefix wfa_fwd_,$(SHAPES))
OBJS    := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(UNITS))) $(OBJDIR)/wfa_gen.o $(OBJDIR)/wfa_multi.o

all: $(LIB) oracle

$(OBJDIR)/wfa_duo.o: $(CSRC)/wfa_duo.hip $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(DUOFLAGS) -c -o $@ $<

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	@mkdir -p wfa_amd/lib
	$(HIPCC) -fPIC --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle -s

asm: $(HDR)
	@mkdir -p build/asm
	@rm -f build/asm/resource_usage.txt
	for u in $(UNITS); do \
	  fl=""; [ $$u = wfa_duo ] && fl="$(DUOFLAGS)"; \
	  $(HIPCC) $(HIPFLAGS) $$fl -save-temps=obj -c -o build/asm/$$u.o $(CSRC)/$$u.hip -Rpass-analysis=kernel-resource-usage 2>> build/asm/resource_usage.txt || true; \
	done

clean:
	rm -rf $(LIB) build
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean
