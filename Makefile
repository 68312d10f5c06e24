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

ENGINE_SRC := openbnsl_amd/csrc/engine.hip openbnsl_amd/csrc/group.hip openbnsl_amd/csrc/anneal.cpp openbnsl_amd/csrc/score_file.cpp
$(ENGINE): $(ENGINE_SRC) openbnsl_amd/csrc/engine_internal.h include/obnsl_engine.h
	mkdir -p openbnsl_amd/lib
	$(HIPCC) $(HIPFLAGS) -Iopenbnsl_amd/csrc -pthread -shared $(ENGINE_SRC) -ldl -o $@

$(PYMOD): $(HOST_SRC) $(HOST_HDR) $(ENGINE)
	g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Iinclude -Iopenbnsl_amd/host \
	    $(PYINC) $(HOST_SRC) -o $@ -Lopenbnsl_amd/lib -lobnsl_engine \
	    -Wl,-rpath,'$$ORIGIN/lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(ENGINE) $(PYMOD)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
