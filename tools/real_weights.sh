#!/bin/bash
# One command for the day real weights exist:   tools/real_weights.sh <snapshot_dir> [--steps audit,known,census,e2e] [--out DIR]
# (audit of config.json / tensors, the reference's known answer notebooks/examples.ipynb:296, arg-max census on trained weights,
# the reference's command line end to end).  Without a directory every step SKIPs and the exit code is 0.  See tools/real_weights.py.
cd "$(dirname "$0")/.." && exec python3 tools/real_weights.py "$@"
