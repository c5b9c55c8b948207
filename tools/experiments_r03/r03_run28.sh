#!/bin/bash
set -u
timeout 600 python -m pytest tests -m gpu -q -x -k "pose or harness or driver or extract or smoke or golden" 2>&1 | tail -2
python tools/driver_call_times.py 40 2>&1 | tail -12
python tools/prof_pose.py 64 300 2>&1 | grep -v oracle
