#!/bin/bash
# round 6: stand-alone SSRN (whole chip) under a kernel trace: the per-dispatch table, default build and with NO_FUSED_CONVT_LN
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r06
rm -rf /tmp/ssrn6; rocprofv3 --kernel-trace --output-format csv -d /tmp/ssrn6 -o s -- python $R/profiles/r03_ssrn_layers.py > /dev/null 2>&1
python $R/profiles/r03_ssrn_layers.py --summarize $(find /tmp/ssrn6 -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r06/ssrn_table.txt
tail -40 $R/gpurun_out/r06/ssrn_table.txt
