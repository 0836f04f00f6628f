# Build of the MI355X (gfx950) wavefront-alignment library and the CPU oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
LIB     := wfa_amd/lib/libwfahip.so
SRC     := wfa_amd/csrc/wfa_host.hip wfa_amd/csrc/wfa_gen.cpp wfa_amd/csrc/wfa_multi.cpp
HDR     := $(wildcard wfa_amd/csrc/*.hpp) include/wfa_hip.h

all: $(LIB) oracle

$(LIB): $(SRC) $(HDR)
	@mkdir -p wfa_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle -s

asm: $(SRC) $(HDR)
	@mkdir -p build/asm
	$(HIPCC) $(HIPFLAGS) -save-temps=obj -c -o build/asm/wfa_host.o wfa_amd/csrc/wfa_host.hip -Rpass-analysis=kernel-resource-usage 2> build/asm/resource_usage.txt || true

clean:
	rm -rf $(LIB) build
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean
