# liblasr.so: hand-written HIP for gfx950 (MI355X).  `make` builds the library in-tree.
HIPCC ?= hipcc
ARCH ?= gfx950
SRC := $(wildcard lightning_asr_amd/csrc/*.hip)
OBJ := $(patsubst lightning_asr_amd/csrc/%.hip,build/%.o,$(SRC))
CXXFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function

all: lightning_asr_amd/liblasr.so tests/stub_rccl/libstubrccl.so

# test infrastructure: a stand-in librccl for several ranks sharing one GPU (tests/test_gpu_dp.py, via LASR_RCCL_PATH)
tests/stub_rccl/libstubrccl.so: tests/stub_rccl/stub_rccl.hip
	$(HIPCC) -O2 --offload-arch=$(ARCH) -fPIC -shared -std=c++17 -o $@ $< -lrt

lightning_asr_amd/liblasr.so: $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

# -MMD: every header a source includes (common.h, gemm.h, ctc_lattice.h, include/lasr.h ...) is a prerequisite of its object
build/%.o: lightning_asr_amd/csrc/%.hip
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -MMD -MP -MF build/$*.d -c $< -o $@

-include $(OBJ:.o=.d)

clean:
	rm -rf build lightning_asr_amd/liblasr.so tests/stub_rccl/libstubrccl.so

.PHONY: all clean
