# liblasr.so: hand-written HIP for gfx950 (MI355X).  `make` builds the library in-tree.
HIPCC ?= hipcc
ARCH ?= gfx950
SRC := $(wildcard lightning_asr_amd/csrc/*.hip)
OBJ := $(patsubst lightning_asr_amd/csrc/%.hip,build/%.o,$(SRC))
CXXFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function

lightning_asr_amd/liblasr.so: $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

# -MMD: every header a source includes (common.h, gemm.h, ctc_lattice.h, include/lasr.h ...) is a prerequisite of its object
build/%.o: lightning_asr_amd/csrc/%.hip
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -MMD -MP -MF build/$*.d -c $< -o $@

-include $(OBJ:.o=.d)

clean:
	rm -rf build lightning_asr_amd/liblasr.so
