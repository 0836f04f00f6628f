# Build of the MI355X (gfx950) wavefront-alignment library and the CPU oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
# -amdgpu-atomic-optimizer-strategy=None: the queue atomics of the kernels are issued by one lane already; the optimizer's
# wave-aggregated form waits for the result on the spot, which wfa_duo_kernel's prefetch must not (wfa_duo.hpp)
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None
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
