#!/bin/bash
# the first thing to run on a box with >= 2 GPUs: see tools/scale_first_run.py
exec python "$(dirname "$0")/scale_first_run.py" "$@"
