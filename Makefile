# Builds libevmi_hip.so (gfx950 code object + host runtime) and the oracle's C pieces.
# No cmake: plain hipcc / gcc, outputs stay in-tree so they travel with gpurun snapshots.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := everyvoice_amd/csrc
BUILD      := build/evmi
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
SRCS       := $(wildcard $(CSRC)/*.hip)
OBJS       := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
LIB        := everyvoice_amd/libevmi_hip.so

all: $(LIB)

# The inference convolution kernels' epilogues are VALU-bound at 32 / 64 channels, and fmaxf(f, f * slope) costs THREE vector ops when
# NaNs are honoured (the compiler canonicalises f with a self-max first).  These two files hold no NaN test; fmaxf already returns the
# other operand for a quiet NaN, so results do not change.
NONAN      := resblock_pair conv_tc_mfma
$(foreach f,$(NONAN),$(eval $(BUILD)/$(f).o: HIPFLAGS += -fno-honor-nans))

$(BUILD)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/evmi.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# tuning variants of the inference convolution (tools/sweep_conv.py): NOT part of the product library
BENCHLIB   := tools/microbench/libevmi_bench.so
bench-kernels: $(LIB)
	$(HIPCC) $(HIPFLAGS) -Itools/microbench -shared tools/microbench/bench_kernels.hip -Leveryvoice_amd -levmi_hip -Wl,-rpath,'$$ORIGIN/../../everyvoice_amd' -o $(BENCHLIB)

clean:
	rm -rf $(BUILD) $(LIB)

.PHONY: all clean bench-kernels
