# Makefile -- native build of the MI355X IGD search path (gfx950 only).
#
#   make            -> igd_amd/lib/libigd_hip.so   HIP engine (include/igd_hip.h)
#                      igd_amd/lib/libigd.so       CLI/libigd flavour of the reference ABI
#                      igd_amd/lib/libigd_py.so    handle flavour (reference src_py wrapper)
#                      igd_amd/lib/libigdr.so      R flavour, .C entry points (+ .Call when R headers exist)
#                      igd_amd/lib/libigd_synth.so synthetic data generator (bench/tests)
#                      bin/igd, bin/igd_synth
#   make oracle     -> oracle/_build/* (+ oracle/_ref/igd when /root/reference is present)
# hipcc cross-compiles for gfx950 without a GPU present.
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CFLAGS  ?= -O2 -g -std=gnu99 -Wall -Wextra -Wno-unused-parameter -fPIC
HIPFLAGS?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function $(EXTRA)
INC     := -Iinclude -Iigd_amd/csrc -Itools
SRC     := igd_amd/csrc
LIB     ?= igd_amd/lib
RPATH   := -Wl,-rpath,'$$ORIGIN' -Wl,-Bsymbolic-functions

all: $(LIB)/libigd_hip.so $(LIB)/libigd.so $(LIB)/libigd_py.so $(LIB)/libigdr.so \
     $(LIB)/libigd_synth.so bin/igd bin/igd_synth

$(LIB) bin:
	mkdir -p $@

$(LIB)/libigd_hip.so: $(SRC)/igd_hip.hip $(wildcard $(SRC)/engine/*.hpp) $(SRC)/igd_create.hip $(SRC)/igd_sortscan.hpp include/igd_hip.h | $(LIB)
	$(HIPCC) $(HIPFLAGS) $(INC) -shared -o $@ $(SRC)/igd_hip.hip $(SRC)/igd_create.hip -lpthread -Wl,-Bsymbolic-functions

CORE_SRC := $(SRC)/igd_core.c $(SRC)/igd_hostpath.c $(SRC)/igd_create.c
# the three host flavours map libigd_hip.so (and with it the HIP runtime) at the first engine call, not at process start
# (igd_hip_lazy.c: 13 ms of every start otherwise -- twice the reference's whole `search -q` of 10^3 queries)
LAZY_SRC := $(SRC)/igd_hip_lazy.c
CORE_HDR := $(SRC)/igd_hip_lazy.c $(SRC)/igd_hostpath.c $(SRC)/igd_core.h $(SRC)/igd_create_host.h include/igd_create.h include/igd_hip.h

$(LIB)/libigd.so: $(SRC)/igd_cli_abi.c $(CORE_SRC) $(CORE_HDR) include/igd_search.h include/igd_base.h $(LIB)/libigd_hip.so
	$(CC) $(CFLAGS) $(INC) -shared -o $@ $(SRC)/igd_cli_abi.c $(CORE_SRC) $(LAZY_SRC) -lz -lpthread -ldl $(RPATH)

$(LIB)/libigd_py.so: $(SRC)/igd_py_abi.c $(CORE_SRC) $(CORE_HDR) include/igd_py_abi.h $(LIB)/libigd_hip.so
	$(CC) $(CFLAGS) $(INC) -shared -o $@ $(SRC)/igd_py_abi.c $(CORE_SRC) $(LAZY_SRC) -lz -lpthread -ldl $(RPATH)

# R flavour: the .C / plain-C entry points always build; the .Call (SEXP) ones need R's headers
R_INC := $(shell R RHOME >/dev/null 2>&1 && echo "-DIGDR_HAVE_R -I`R RHOME`/include")
$(LIB)/libigdr.so: $(SRC)/igdr_abi.c $(CORE_SRC) $(CORE_HDR) include/igdr_abi.h $(LIB)/libigd_hip.so
	$(CC) $(CFLAGS) $(INC) $(R_INC) -shared -o $@ $(SRC)/igdr_abi.c $(CORE_SRC) $(LAZY_SRC) -lz -lpthread -ldl $(RPATH)

$(LIB)/libigd_synth.so: tools/igd_synth.c tools/igd_synth_writer.c tools/igd_synth_writer.h $(SRC)/igd_core.c $(CORE_HDR) $(LIB)/libigd_hip.so
	$(CC) $(CFLAGS) $(INC) -shared -o $@ tools/igd_synth.c tools/igd_synth_writer.c $(SRC)/igd_core.c -L$(LIB) -ligd_hip -lz -lpthread $(RPATH)

bin/igd: $(SRC)/igd_main.c $(LIB)/libigd.so | bin
	$(CC) $(CFLAGS) $(INC) -o $@ $(SRC)/igd_main.c -L$(LIB) -ligd -Wl,-rpath,'$$ORIGIN/../igd_amd/lib'

bin/igd_synth: tools/igd_synth.c tools/igd_synth_writer.c $(SRC)/igd_core.c $(CORE_HDR) $(LIB)/libigd_hip.so | bin
	$(CC) $(CFLAGS) $(INC) -DIGD_SYNTH_MAIN -o $@ tools/igd_synth.c tools/igd_synth_writer.c $(SRC)/igd_core.c -L$(LIB) -ligd_hip -lz -lpthread -Wl,-rpath,'$$ORIGIN/../igd_amd/lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) bin
.PHONY: all oracle clean
