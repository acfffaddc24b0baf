# Convenience targets; the contract entry points are build.sh, bench.py and __graft_entry__.py.
.PHONY: build test gpu-test bench smoke goldens clean

build:            ## hipcc --offload-arch=gfx950, in-tree: detikzify_amd/lib/libdtk_hip.so
	bash build.sh

test:             ## CPU: oracle vs goldens, host logic vs the reference's own code, C-ABI symbols, gloo ranks
	python -m pytest tests -x -q -m "not gpu"

gpu-test:         ## parity on an MI355X
	python -m pytest tests -x -q -m gpu

smoke:
	python -c "import __graft_entry__ as g; g.smoke()"

bench:            ## one JSON line: tokens/s, roofline, cpu_baseline
	python bench.py

goldens:          ## build container only (reads /root/reference); deterministic: regenerating changes no byte
	python tests/golden/make_golden.py

clean:
	rm -rf build detikzify_amd/lib/libdtk_hip.so
