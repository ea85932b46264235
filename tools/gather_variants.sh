#!/bin/bash
python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in ${VARIANTS:-4 6}; do echo "== variant $v"; CPM_GATHER_VARIANT=$v python tools/gather_exp.py 2>&1 | grep gather; done
CPM_GATHER_VARIANT=${STAMP:-6} python tools/gather_stamps.py 2>&1 | tail -24
