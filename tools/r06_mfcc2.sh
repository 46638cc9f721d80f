#!/bin/bash
# k_mfcc_frames variants (GPU box): bash tools/r06_mfcc2.sh   -- kernel time by rocprofv3 per variant (build defs | runtime switches)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
for defs in "" "-DMFCC_BREV_READ=0"; do
  touch htk_amd/csrc/mfcc.hip
  if [ -n "$defs" ]; then HTKAMD_MFCC_DEFS="$defs" python3 -m htk_amd.build > /dev/null 2>&1; else python3 -m htk_amd.build > /dev/null 2>&1; fi
  python tools/mfcc_diag.py 2>&1 | tail -1
  for v in "HTKAMD_MFCC_WPC=100000" "HTKAMD_MFCC_ONE_FRAME=1 HTKAMD_MFCC_WPC=100000" "HTKAMD_MFCC_WPC=16"; do
    out=gpurun_out/mfccv; rm -rf $out; mkdir -p $out
    env $v rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 tools/mfcc_bench.py > $out/o.txt 2>/dev/null
    echo "[$defs] $v: $(grep k_mfcc_frames $out/p_kernel_stats.csv | cut -d, -f1-4)  | $(tail -1 $out/o.txt | cut -c1-80)"
  done
done
touch htk_amd/csrc/mfcc.hip; python3 -m htk_amd.build > /dev/null 2>&1
