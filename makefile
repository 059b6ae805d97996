# `make bindings` mirrors the reference's build entry (reference makefile:1-4): it produces the Python
# module `bindings` — here a thin shim over the C-ABI library of hand-written gfx950 kernels.
PYTHON ?= python3

bindings:
	$(PYTHON) -m million_amd.build

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
	rm -f million_amd/libmillion_hip.so oracle/libpq_oracle.so

.PHONY: bindings oracle golden test test-gpu bench clean
