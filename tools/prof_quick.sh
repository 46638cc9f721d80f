#!/bin/bash
# kernel-trace statistics of the default bench command only (the quick look while iterating): gpurun_out/<tag>_kernel_stats.csv
tag=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag --output-format csv -- python3 bench.py --cpu-seconds 0 ${@:2} > $out/bench_under_rocprof.json 2> $out/rocprof_trace.log
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
head -14 gpurun_out/${tag}_kernel_stats.csv
