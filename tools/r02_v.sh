#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_v.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_v.log
bash tools/ab_libs.sh "csrc_base csrc" 3
