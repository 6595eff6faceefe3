# Top-level conveniences.  `make` builds the HIP library; `make test_launch` is the reference's own
# plumbing check (its Makefile target of the same name): the bundled read against the DXZ1* monomers,
# final TSV compared with the reference's golden file.  Needs an MI355X for test_launch.
ROOT := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
TD   := $(ROOT)tests/golden/test_data
OUT  ?= /tmp/sd_test_launch

build:
	$(MAKE) -C $(ROOT)stringdecomposer_amd/csrc

oracle:
	$(MAKE) -C $(ROOT)oracle all

test_launch: build
	python3 $(ROOT)bin/stringdecomposer $(TD)/read.fa $(TD)/DXZ1_star_monomers.fa -o $(OUT) --second-best
	grep -q "Thank you for using StringDecomposer!" $(OUT)/stringdecomposer.log
	diff -q $(TD)/final_decomposition_fc89af8.tsv $(OUT)/final_decomposition.tsv

test:
	python3 -m pytest $(ROOT)tests -q -m "not gpu"

test_gpu:
	python3 -m pytest $(ROOT)tests -q -m gpu

clean:
	$(MAKE) -C $(ROOT)stringdecomposer_amd/csrc clean
	rm -rf $(OUT)

.PHONY: build oracle test_launch test test_gpu clean
