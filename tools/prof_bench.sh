#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (the numbers DESIGN.md / profiles/ quote), then the plain bench line.
# usage (GPU box): bash tools/prof_bench.sh r01g
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o $tag --output-format csv -- python3 bench.py > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.log
python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -1 gpurun_out/$tag/bench.json
