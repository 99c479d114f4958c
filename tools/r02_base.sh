#!/bin/bash
# round-2 baseline on the GPU box: headline bench, then the SQ counter pass
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_base
mkdir -p $OUT
cd $ROOT
python bench.py --bank-cache /tmp/bank > $OUT/bench.json 2> $OUT/bench.err && echo "bench: $(cut -c1-160 $OUT/bench.json)" && bash tools/pmc_sq.sh r02_base
