# Builds the C-ABI engine (HIP, gfx950) and the oracle's C restatement.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := directtrajopt.jl_amd/csrc
# TUNING=1 compiles the A/B switches (environment variables DTO_*) into a SECOND library, libdto_engine_t.so, from objects of
# its own (*.t.o); the product build reads no environment variable
LIB   := directtrajopt.jl_amd/libdto_engine$(if $(TUNING),_t,).so
O     := $(if $(TUNING),t.o,o)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(if $(TUNING),-DDTO_TUNING,)
OBJS  := $(addprefix $(CSRC)/,dto_kernels.$(O) dto_small.$(O) dto_sweep_fused.$(O) dto_sweep_gs.$(O) dto_chain64.$(O) dto_tdb.$(O) dto_hostxfer.$(O) dto_comm.$(O) dto_engine.$(O))

all: $(LIB)

$(CSRC)/dto_kernels.$(O): $(CSRC)/dto_kernels.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h $(CSRC)/dto_gemm_ring.hip.h $(CSRC)/dto_hostxfer.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_small.$(O): $(CSRC)/dto_small.hip $(CSRC)/dto_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_tdb.$(O): $(CSRC)/dto_tdb.hip $(CSRC)/dto_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_sweep_fused.$(O): $(CSRC)/dto_sweep_fused.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_sweep_gs.$(O): $(CSRC)/dto_sweep_gs.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_chain64.$(O): $(CSRC)/dto_chain64.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_hostxfer.$(O): $(CSRC)/dto_hostxfer.cpp $(CSRC)/dto_hostxfer.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/dto_comm.$(O): $(CSRC)/dto_comm.cpp $(CSRC)/dto_comm.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/dto_engine.$(O): $(CSRC)/dto_engine.cpp $(CSRC)/dto_kernels.h $(CSRC)/dto_hostxfer.h $(CSRC)/dto_comm.h include/dto_engine.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread -ldl

clean:
	rm -f $(CSRC)/*.o directtrajopt.jl_amd/libdto_engine.so directtrajopt.jl_amd/libdto_engine_t.so

.PHONY: all clean
