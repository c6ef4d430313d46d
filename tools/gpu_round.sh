#!/bin/bash
# One gpurun call of the round's standard checks; each step logs under gpurun_out/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/gpu_tests.log
