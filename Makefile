# Convenience targets; the driver uses __graft_entry__.build(), pytest and bench.py directly.
.PHONY: build test test-gpu bench clean golden

build:
	python -c "import __graft_entry__ as g; g.build()"

test: build
	python -m pytest tests -x -q -m "not gpu"

test-gpu: build
	python -m pytest tests -x -q -m gpu

bench: build
	python bench.py

golden:            # needs /root/reference (build container only)
	python tests/golden/make_golden.py

clean:
	$(MAKE) -C reflectance_filtering_amd/csrc clean
	$(MAKE) -C oracle clean
