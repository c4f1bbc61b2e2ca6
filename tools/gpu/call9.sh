#!/bin/bash
# round 4, GPU call 9: the tests added since call 8
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c9; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -q -k "beyond_the_lds or without_a_spectrum_cache or multi_auto or independent_pass or up_to_the_3073 or config3_tap_count" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
