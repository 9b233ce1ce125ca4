# Convenience targets; the driver uses __graft_entry__.build(), pytest and bench.py directly.
.PHONY: all build test test-gpu bench clean
all: build
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	$(MAKE) -C datok_amd/csrc clean
	$(MAKE) -C oracle clean
