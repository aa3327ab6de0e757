# Convenience targets; the canonical entry points are __graft_entry__.build()/smoke(), pytest and bench.py.
PY ?= python

.PHONY: build test gpu-test smoke bench clean
build:            ## hipcc --offload-arch=gfx950: libyy_hip.so in-tree, plus the CPU oracle (test infrastructure)
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build       ## CPU suite: oracle vs reference goldens, host logic, C-ABI symbols, gloo world-size-2
	$(PY) -m pytest tests -x -q -m "not gpu"
gpu-test: build   ## parity suite proper, through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -x -q -m gpu
smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"
bench: build      ## one JSON line: BASELINE config 2 on one GPU
	$(PY) bench.py
clean:
	rm -f yinyang-game-alphazero_amd/csrc/libyy_hip.so oracle/libyy_oracle.so
