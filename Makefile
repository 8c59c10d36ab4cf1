# convenience targets; everything here is also reachable as plain python commands (README.md)
.PHONY: build oracle test test-gpu bench smoke clean
build:
	python -m hrbffusion3d_amd.build
oracle:
	$(MAKE) -C oracle
test: build oracle
	python -m pytest tests -q -m "not gpu"
test-gpu: build oracle
	python -m pytest tests -q -m gpu
bench: build oracle
	python bench.py
smoke:
	python -c "import __graft_entry__ as g; g.build(); g.smoke()"
clean:
	rm -rf hrbffusion3d_amd/libhrbf_mi355.so hrbffusion3d_amd/_build oracle/_build
