# Builds libfs2hip.so (hand-written HIP kernels for gfx950 behind a C ABI) and the oracle helpers.
# hipcc cross-compiles gfx950 without a GPU present.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := fastspeech2_amd/csrc
OBJ   := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -I$(CSRC) -Iinclude -Wno-unused-result
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS)) $(OBJ)/fs2_api.o
LIB   := fastspeech2_amd/libfs2hip.so

AIDLIB := tests/aids/libfs2_testaid.so

all: $(LIB) $(AIDLIB)

# host-only test aids over the kernels' own schedule source (include/fs2hip_testaid.h); not part of the product library
$(AIDLIB): tests/aids/fs2_testaid.cpp $(CSRC)/fs2_sched.h include/fs2hip_testaid.h
	g++ -O2 -std=c++17 -shared -fPIC -I$(CSRC) -Iinclude -o $@ tests/aids/fs2_testaid.cpp

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/fs2_common.h $(CSRC)/fs2_gemm.h $(CSRC)/fs2_sched.h $(CSRC)/fs2_wgrad.h $(CSRC)/fs2_gemm_epi.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/fs2_api.o: $(CSRC)/fs2_api.cpp
	@mkdir -p $(OBJ)
	$(HIPCC) -O2 -std=c++17 -fPIC -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB) $(AIDLIB)

.PHONY: all clean

# Development build: the same sources with -DFS2_DEV (environment-driven ablation / forced-variant switches compiled IN).
# Never loaded by the product: tools/ scripts select it with FS2_LIB_PATH=fastspeech2_amd/libfs2hip_dev.so.
DEVOBJ := build/obj_dev
DEVOBJS := $(patsubst $(CSRC)/%.hip,$(DEVOBJ)/%.o,$(SRCS)) $(DEVOBJ)/fs2_api.o
DEVLIB := fastspeech2_amd/libfs2hip_dev.so
$(DEVOBJ)/%.o: $(CSRC)/%.hip $(CSRC)/fs2_common.h $(CSRC)/fs2_gemm.h $(CSRC)/fs2_sched.h $(CSRC)/fs2_wgrad.h $(CSRC)/fs2_gemm_epi.h
	@mkdir -p $(DEVOBJ)
	$(HIPCC) $(HIPFLAGS) -DFS2_DEV -c $< -o $@
$(DEVOBJ)/fs2_api.o: $(CSRC)/fs2_api.cpp
	@mkdir -p $(DEVOBJ)
	$(HIPCC) -O2 -std=c++17 -fPIC -c $< -o $@
$(DEVLIB): $(DEVOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DEVOBJS)
dev: $(DEVLIB)
.PHONY: dev
