#!/bin/bash
mkdir -p gpurun_out/r04_c29; O=gpurun_out/r04_c29
timeout 600 python tools/cumask_probe.py 1024 > $O/cumask.txt 2>&1; tail -8 $O/cumask.txt
