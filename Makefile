# Top-level conveniences.  `make` builds everything __graft_entry__.build() builds; `make check` is the gate in front of every
# commit: the CPU suite (oracle against the golden vectors, host logic, ABI/documentation guards, gloo world-size-2 runs).
PY ?= python

all:
	$(PY) __graft_entry__.py

check: all
	$(PY) -m pytest tests/ -x -q -m "not gpu"

# experiment build of the library (probes of profiles/micro; never the product): build/exp/liblsf_x.so
exp:
	@mkdir -p build/exp
	$(MAKE) -C levelsetfortran_amd/csrc OUT=../../build/exp/liblsf_x.so EXTRA=-DLSF_EXPERIMENTS

.PHONY: all check exp
