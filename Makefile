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

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/fs2_common.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/fs2_api.o: $(CSRC)/fs2_api.cpp
	@mkdir -p $(OBJ)
	$(HIPCC) -O2 -std=c++17 -fPIC -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
