#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16s.py -x -q -s -k "wide_rows or trained_norm1 or device_side_scale" 2>&1 | grep -v '^$' | tail -25
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_backward.py -x -q -k "trained_like or tap_with_one or shared_parameter or g11" 2>&1 | tail -15
