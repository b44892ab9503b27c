# Builds every native piece in-tree:
#   pathtracer_amd/lib/libtkhip.so   HIP kernels + C ABI (gfx950)
#   pathtracer_amd/lib/libtkhost.so  C++ loader, tracer classes, C view for Python
#   pathtracer_amd/lib/raytracer     headless CLI (raytracer <file.scene>)
#   oracle/liboracle.so              CPU oracle (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
LIBDIR = pathtracer_amd/lib
CSRC = pathtracer_amd/csrc
HOST = pathtracer_amd/host

# -ffp-contract=off: products and sums round separately, as in the CPU oracle,
# so that distance tests, hit tests and the pixel update are bit-comparable.
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
           -Iinclude -I$(CSRC) -Wall -Wno-unused-function
CXXFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(HOST) -Wall

HIP_SRCS = $(wildcard $(CSRC)/*.hip)
HIP_HDRS = $(wildcard $(CSRC)/*.h) include/tkhip.h
HOST_SRCS = $(HOST)/scene_loader.cpp $(HOST)/tkhost_api.cpp $(HOST)/sppm_tracer.cpp
HOST_HDRS = $(wildcard $(HOST)/*.h) include/tkhost.h include/tkhip.h

all: $(LIBDIR)/libtkhip.so $(LIBDIR)/libtkhost.so $(LIBDIR)/raytracer oracle

# one object per translation unit (under build/, git-ignored), so that a change recompiles one file
HIP_OBJS = $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))
EXTRA_HIPFLAGS ?=

build/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -c -o $@ $<

$(LIBDIR)/libtkhip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(HIP_OBJS)

$(LIBDIR)/libtkhost.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/libtkhip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRCS) -L$(LIBDIR) -ltkhip -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/raytracer: $(HOST)/main.cpp $(LIBDIR)/libtkhost.so
	$(CXX) $(CXXFLAGS) -o $@ $(HOST)/main.cpp -L$(LIBDIR) -ltkhost -ltkhip -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIBDIR)/*.so $(LIBDIR)/raytracer build/*.o
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
