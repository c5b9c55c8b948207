#!/bin/bash
set -u
timeout 600 python -m pytest tests -m gpu -q -x -k "extract or harness or driver or golden or gaussian or front or remap" 2>&1 | tail -2
for i in 1 2; do
echo "graph   : $(python tools/driver_call_times.py 40 2>&1 | grep 'remap_extract\|^track')"
echo "no graph: $(CMS_NO_FRAME_GRAPH=1 python tools/driver_call_times.py 40 2>&1 | grep 'remap_extract\|^track')"
done
