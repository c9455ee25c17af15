#!/bin/bash
BDS_VERBOSE=1 timeout 900 python -m pytest tests/test_acq_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -30
