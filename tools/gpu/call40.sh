#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end or cfar" 2>&1 | tail -1
