# Builds, in-tree:
#   openbnsl_amd/lib/libobnsl_engine.so   HIP kernels + C ABI (gfx950)
#   openbnsl_amd/openbnsllib*.so          pybind11 module with the reference's Python surface
#   oracle/_build, oracle/_ref            test-only checkers (see oracle/Makefile)
PY ?= python3
EXT := $(shell $(PY) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC := $(shell $(PY) -m pybind11 --includes)
HIPCC ?= hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
            -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude

ENGINE := openbnsl_amd/lib/libobnsl_engine.so
PYMOD := openbnsl_amd/openbnsllib$(EXT)
HOST_SRC := $(wildcard openbnsl_amd/host/*.cpp)
HOST_HDR := $(wildcard openbnsl_amd/host/*.h) include/obnsl_engine.h

ifneq ($(HOST_SRC),)
all: $(ENGINE) $(PYMOD) oracle
else
all: $(ENGINE) oracle
endif

CSRC := openbnsl_amd/csrc
ENGINE_SRC := $(CSRC)/abi.hip $(CSRC)/planner.hip $(CSRC)/passes.hip $(CSRC)/gemm_mfma.hip $(CSRC)/gemm_valu.hip \
              $(CSRC)/kernels_count.hip $(CSRC)/kernels_score.hip $(CSRC)/kernels_search.hip \
              $(CSRC)/group.hip $(CSRC)/env.cpp $(CSRC)/anneal.cpp $(CSRC)/score_file.cpp
ENGINE_HDR := $(CSRC)/engine_internal.h $(CSRC)/engine_types.h $(CSRC)/engine_state.h include/obnsl_engine.h
ENGINE_OBJ := $(patsubst $(CSRC)/%,build/engine/%.o,$(ENGINE_SRC))
PROBE_OBJ := $(patsubst $(CSRC)/%,build/probes/%.o,$(ENGINE_SRC))
# the same library with the timing probes compiled in (wrong counts by construction when one
# is switched on): loaded only by tools/, through OBNSL_ENGINE_LIB
PROBE_LIB := tools/bin/libobnsl_engine_probes.so

build/engine/%.o: $(CSRC)/% $(ENGINE_HDR)
	mkdir -p build/engine
	$(HIPCC) $(HIPFLAGS) -I$(CSRC) -pthread -c $< -o $@

build/probes/%.o: $(CSRC)/% $(ENGINE_HDR)
	mkdir -p build/probes
	$(HIPCC) $(HIPFLAGS) -DOBNSL_PROBES -I$(CSRC) -pthread -c $< -o $@

$(ENGINE): $(ENGINE_OBJ) $(CSRC)/exports.map
	mkdir -p openbnsl_amd/lib
	$(HIPCC) --offload-arch=gfx950 -fPIC -pthread -shared $(ENGINE_OBJ) -ldl -Wl,--version-script=$(CSRC)/exports.map -o $@

$(PROBE_LIB): $(PROBE_OBJ)
	mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -fPIC -pthread -shared $(PROBE_OBJ) -ldl -Wl,--version-script=$(CSRC)/exports.map -o $@

probes: $(PROBE_LIB)

$(PYMOD): $(HOST_SRC) $(HOST_HDR) $(ENGINE)
	g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Iinclude -Iopenbnsl_amd/host \
	    $(PYINC) $(HOST_SRC) -o $@ -Lopenbnsl_amd/lib -lobnsl_engine \
	    -Wl,-rpath,'$$ORIGIN/lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(ENGINE) $(PYMOD) $(PROBE_LIB)
	rm -rf build/engine build/probes
	$(MAKE) -C oracle clean

.PHONY: all oracle clean probes
