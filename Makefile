# Top-level build: libntt_mi355x.so (HIP, gfx950), the CPU oracle and the CPU
# emulation of the kernel templates used by the tests.
#
#   make            product library only
#   make all-test   + oracle, reference oracle (if /root/reference exists), emulator
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG      = optimized-number-theoretic-transform-implementations_amd
CSRC     = $(PKG)/csrc
LIB      = $(PKG)/libntt_mi355x.so
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
            -Wall -Wextra -Wno-unused-parameter -Iinclude -Iinclude/internal -I$(CSRC) $(EXTRA_HIPFLAGS)
# (longest translation units first: `make -j` starts them in this order, and the four FP64 block-kernel units take two minutes each)
OBJS     = $(CSRC)/inst_f64k0.o $(CSRC)/inst_f64k1.o $(CSRC)/inst_f64k18.o $(CSRC)/inst_f64w.o \
           $(CSRC)/inst_dm_u64x_k0.o $(CSRC)/inst_dm_u64x_k1.o $(CSRC)/inst_dm_u64x_k3.o $(CSRC)/inst_u64x_k0.o $(CSRC)/inst_u64x_k1.o $(CSRC)/inst_u64x_k3.o \
           $(CSRC)/inst_u64.o $(CSRC)/inst_u64r4.o \
           $(CSRC)/inst_dot_f64k0.o $(CSRC)/inst_dot_f64k1.o $(CSRC)/inst_dot_f64k18.o $(CSRC)/inst_dot_f64w.o $(CSRC)/inst_dot_u64.o \
           $(CSRC)/inst_mul_f64k0.o $(CSRC)/inst_mul_f64k1.o $(CSRC)/inst_mul_f64k18.o $(CSRC)/inst_mul_f64w.o $(CSRC)/inst_mul_u64.o \
           $(CSRC)/inst_team_f64k0.o $(CSRC)/inst_team_f64k1.o $(CSRC)/inst_team_f64k18.o $(CSRC)/inst_team_f64w.o $(CSRC)/ntt_host.o
# the kernel translation units see the kernel headers only; the host layer also the public headers
KHDRS    = $(wildcard $(CSRC)/*.h)
HDRS     = $(KHDRS) $(wildcard $(CSRC)/host/*.inc) $(wildcard include/*.h) $(wildcard include/internal/*.h)

lib: $(LIB)

$(CSRC)/ntt_host.o: $(CSRC)/ntt_host.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/%.o: $(CSRC)/%.hip $(KHDRS)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle all
	if [ -d /root/reference/src ]; then $(MAKE) -C oracle ref; fi

emu:
	$(MAKE) -C tests/emu -j8

all-test: lib oracle emu

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean
	$(MAKE) -C tests/emu clean

.PHONY: lib oracle emu all-test clean

# instruction-level micro-benchmarks (run on the GPU box: build/ubench)
ubench: build/ubench
build/ubench: tools/ubench.hip $(HDRS)
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -ffp-contract=off -I$(CSRC) -o $@ tools/ubench.hip
.PHONY: ubench

# data-movement skeletons of the 2^14 kernel (run on the GPU box: build/skel [GiB] [launches])
skel: build/skel
build/skel: tools/skel.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -ffp-contract=off -o $@ tools/skel.hip
.PHONY: skel

# ASAN + UBSAN run of the CPU-side code (oracle, host table/pass planning, kernel templates in the emulator)
sanitize:
	$(MAKE) -C tests/sanitize run
.PHONY: sanitize

# memory skeleton of the XCD-local four-step transform for N = 2^16 / 2^17 (run on the GPU box: build/skel4 [GiB] [reps] [filter])
skel4: build/skel4
build/skel4: tools/skel4.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -ffp-contract=off -o $@ tools/skel4.hip
.PHONY: skel4

# second skeleton of the XCD-local two-pass transform: barrier-free items, class queues (build/skel5 [GiB] [reps] [filter])
skel5: build/skel5
build/skel5: tools/skel5.hip
	mkdir -p build
	$(HIPCC) -O3 --offload-arch=$(ARCH) -std=c++17 -ffp-contract=off -o $@ tools/skel5.hip
.PHONY: skel5
