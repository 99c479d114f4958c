#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r02_a
python -m pytest tests -m gpu -x -q > gpurun_out/r02_a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_a/pytest.log
bash tools/pmc_sq.sh r02_a --step-mode two_streams
