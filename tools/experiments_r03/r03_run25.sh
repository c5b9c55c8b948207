#!/bin/bash
# k_fast_cells: time up to each phase boundary (CMS_DBG_FAST_STOP: 9 launch only, 1 staging, 2 compass pre-test, 3 refinement, 4 ring score, 0 all)
for st in 9 1 2 3 4 0; do
echo "stop=$st: $(CMS_DBG_FAST_STOP=$st python tools/prof_frames.py 256 550 6 2>&1 | tail -1 | cut -c1-120)"
done
