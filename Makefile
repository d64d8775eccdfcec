# Top-level build: HIP engine (gfx950 only), host front end, oracle (test infrastructure).
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CLANGXX  ?= /opt/rocm/lib/llvm/bin/clang++   # acq_math.hpp uses clang vector types (ext_vector_type)
ARCH     := gfx950
PKG      := gnss-gps-sdr_amd
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
LIBDIR   := $(PKG)/lib
BINDIR   := $(PKG)/bin
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $(HIPFLAGS_EXTRA)
HOSTFLAGS:= -O2 -std=c++17 -fPIC -Wall -Wno-unknown-pragmas -ffp-contract=off

all: lib host oracle emul

lib: $(LIBDIR)/libgpsacq.so
$(LIBDIR)/libgpsacq.so: $(CSRC)/acq_kernels.hip $(CSRC)/key_kernels.hip $(CSRC)/iq_kernels.hip $(CSRC)/gen_kernels.hip $(CSRC)/gpsacq_engine.cpp $(CSRC)/gpsacq_multi.cpp $(CSRC)/*.hpp include/gpsacq.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -c $(CSRC)/gpsacq_engine.cpp -o $(LIBDIR)/gpsacq_engine.o
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/gpsacq_multi.cpp -o $(LIBDIR)/gpsacq_multi.o
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/acq_kernels.hip -o $(LIBDIR)/acq_kernels.o
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/key_kernels.hip -o $(LIBDIR)/key_kernels.o
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/iq_kernels.hip -o $(LIBDIR)/iq_kernels.o
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/gen_kernels.hip -o $(LIBDIR)/gen_kernels.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(LIBDIR)/acq_kernels.o $(LIBDIR)/key_kernels.o $(LIBDIR)/iq_kernels.o $(LIBDIR)/gen_kernels.o $(LIBDIR)/gpsacq_engine.o $(LIBDIR)/gpsacq_multi.o -ldl -pthread

host: $(LIBDIR)/libgps_search.so $(BINDIR)/gps_test $(BINDIR)/hip_floor $(BINDIR)/pk_fma_stream
# measurement aid of bench.py's roofline: the rate of a pure v_pk_fma_f32 stream on this box (roofline.pk_fma_stream_TF)
$(BINDIR)/pk_fma_stream: tools/ubench/pk_fma_stream.hip
	@mkdir -p $(BINDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ $<
# measurement aid of bench.py's e2e_cli leg: the wall clock of a HIP process that does nothing (runtime start-up floor)
$(BINDIR)/hip_floor: tools/ubench/hip_floor.hip
	@mkdir -p $(BINDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ $<
$(LIBDIR)/libgps_search.so: $(HOST)/search_api.cpp include/gps_search.h include/gpsacq.h $(LIBDIR)/libgpsacq.so
	$(CXX) $(HOSTFLAGS) -pthread -shared -o $@ $(HOST)/search_api.cpp -L$(LIBDIR) -lgpsacq -Wl,-rpath,'$$ORIGIN'
$(BINDIR)/gps_test: $(HOST)/gps_test.cpp include/gps_search.h $(LIBDIR)/libgps_search.so
	@mkdir -p $(BINDIR)
	$(CXX) $(HOSTFLAGS) -pthread -o $@ $(HOST)/gps_test.cpp -L$(LIBDIR) -lgps_search -lgpsacq -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

emul: tests/emul/libemul_acq.so
tests/emul/libemul_acq.so: tests/emul/emul_acq.cpp $(CSRC)/*.hpp
	$(CLANGXX) -x c++ $(HOSTFLAGS) -shared -o $@ tests/emul/emul_acq.cpp

# Drop-in check (authoring container only): the reference's own front end, compiled from where it lies and never copied,
# linked against our SearchInit/SearchTask.  The binary goes to oracle/_ref/ (git-ignored).  It travels to the GPU box for ONE
# consumer: tests/test_gpu_parity.py::test_reference_main_against_our_library, the only thing in the tree that opens it
# (`grep -rn gps_test_refmain`: that test, this target) -- no product path, bench.py and smoke() never touch it.
# It holds the reference's main() only (argument handling + the two calls); every search symbol it calls is ours.
dropin-check: $(LIBDIR)/libgps_search.so
	@mkdir -p oracle/_ref
	$(CXX) -O2 -I/root/reference/c /root/reference/c/test_search_offline.cpp -o oracle/_ref/gps_test_refmain \
	    -L$(LIBDIR) -lgps_search -lgpsacq -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)'

clean:
	rm -rf $(LIBDIR) $(BINDIR) build tests/emul/libemul_acq.so
	$(MAKE) -C oracle clean
.PHONY: all lib host oracle emul clean dropin-check
