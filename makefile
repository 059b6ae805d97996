# `make bindings` mirrors the reference's build entry (reference makefile:1-4): it produces the Python
# module `bindings` — here a thin shim over the C-ABI library of hand-written gfx950 kernels.
PYTHON ?= python3

bindings:
	$(PYTHON) -m million_amd.build

install: bindings   # reference makefile:1-4 installs the module; here: a .pth file in the user site-packages (what `pip install -e .` does)
	$(PYTHON) tools/install_pth.py

uninstall:
	$(PYTHON) tools/install_pth.py --uninstall

debug-ids:         # diagnostic variant: page ids bounds-checked in the decode-attention kernels (MILLION_HIP_LIB=million_amd/libmillion_hip_dbgids.so)
	$(PYTHON) -m million_amd.build --debug-ids

# A plain-C caller of the C ABI (no Python, no torch; reference: scripts/modeldb/bindings/Kernel_Test/main.cu:59-226):
# prepares a codebook, encodes a prompt into pages, replays decode launches, checks codes and outputs on the host.
ROCM ?= /opt/rocm
cabi-bench: bindings
	mkdir -p build
	gcc -std=c99 -O2 -Wall -Wextra -ffp-contract=off -Iinclude -I$(ROCM)/include tools/cabi_bench.c -o build/cabi_bench \
	    -Lmillion_amd -lmillion_hip -L$(ROCM)/lib -lamdhip64 -lm -Wl,-rpath,'$$ORIGIN/../million_amd' -Wl,-rpath,$(ROCM)/lib

oracle:
	$(MAKE) -C oracle

golden:            # build container only: needs /root/reference (tools/gen_golden.py)
	$(PYTHON) tools/gen_golden.py

test:
	$(PYTHON) -m pytest tests -x -q -m "not gpu"

test-gpu:
	$(PYTHON) -m pytest tests -x -q -m gpu

bench:
	$(PYTHON) bench.py

clean:
	rm -f million_amd/libmillion_hip.so million_amd/libmillion_hip_dbgids.so oracle/libpq_oracle.so build/cabi_bench

.PHONY: cabi-bench bindings install uninstall debug-ids oracle golden test test-gpu bench clean
