#!/bin/bash
# round 3, GPU call 1: full GPU test suite + A/B of the GELU-table / statistics variants + the two-stream overlap probe
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log )
tail -5 gpurun_out/r03a_pytest.log
AB_STEPS=40 AB_WARMUP=20 bash tools/ab_run.sh base:TMPNN_LIB=thermompnn_amd/libtmpnn_base.so lut:TMPNN_LIB=thermompnn_amd/libtmpnn_lut.so lutstat noslp:TMPNN_LIB=thermompnn_amd/libtmpnn_noslp.so base2:TMPNN_LIB=thermompnn_amd/libtmpnn_base.so lutstat2 2>&1 | tee gpurun_out/r03a_ab.log
timeout 300 python tools/overlap_probe.py 2>&1 | tail -2 | tee gpurun_out/r03a_overlap.log
