#!/bin/bash
# The -m gpu suite once per option set (PNR_TEST_OPTIONS, tests/conftest.py): kernel forms and scheduler settings must not change a
# single byte of any result.   bash scripts/options_matrix.sh > gpurun_out/options_matrix.txt
for o in "cube_copy=0" "gauss_march=0" "sums_deep=0" "sums_deep=1" "max_split=3,split_x10=12" "groups=3,poll=2,lag=1" "groups=1,target=8" "tentative=0,concentrate=0"; do
  echo "== PNR_TEST_OPTIONS=$o"
  PNR_TEST_OPTIONS=$o timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
done
