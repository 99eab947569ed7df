#!/bin/bash
# flake hunt: the whole GPU suite twice more, as the driver runs it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r06/suite_x$i.log 2>&1; echo "run $i rc=$?"; tail -1 gpurun_out/r06/suite_x$i.log; done
