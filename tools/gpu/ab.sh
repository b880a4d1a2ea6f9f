#!/bin/bash
# same-box A/B of a library knob: bash tools/gpu/ab.sh KNOB [N] [resample_fn]
timeout -k 10 300 python tools/ab_knob.py "$@" 2>&1 | grep -v amdgpu.ids
