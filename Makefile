# Builds the C-ABI engine (HIP, gfx950) and the oracle's C restatement.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := directtrajopt.jl_amd/csrc
LIB   := directtrajopt.jl_amd/libdto_engine.so
# TUNING=1 compiles the A/B switches (environment variables DTO_*) into the library; the product build reads none
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(if $(TUNING),-DDTO_TUNING,)

all: $(LIB)

$(CSRC)/dto_kernels.o: $(CSRC)/dto_kernels.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h $(CSRC)/dto_hostxfer.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_small.o: $(CSRC)/dto_small.hip $(CSRC)/dto_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_tdb.o: $(CSRC)/dto_tdb.hip $(CSRC)/dto_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_sweep_fused.o: $(CSRC)/dto_sweep_fused.hip $(CSRC)/dto_kernels.h $(CSRC)/dto_gemm.hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/dto_hostxfer.o: $(CSRC)/dto_hostxfer.cpp $(CSRC)/dto_hostxfer.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/dto_comm.o: $(CSRC)/dto_comm.cpp $(CSRC)/dto_comm.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/dto_engine.o: $(CSRC)/dto_engine.cpp $(CSRC)/dto_kernels.h $(CSRC)/dto_hostxfer.h $(CSRC)/dto_comm.h include/dto_engine.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(CSRC)/dto_kernels.o $(CSRC)/dto_small.o $(CSRC)/dto_sweep_fused.o $(CSRC)/dto_tdb.o $(CSRC)/dto_hostxfer.o $(CSRC)/dto_comm.o $(CSRC)/dto_engine.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread -ldl

clean:
	rm -f $(CSRC)/*.o $(LIB)

.PHONY: all clean
