cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
OPH_TRACE=1 OPH_RUN_STAMPS=1 timeout 300 python profiles/r03_probe.py 3 > gpurun_out/r05/stamps_aql.txt 2>&1; echo rc=$?
grep -h "batch\|stamped step\|hc_fused level\|step 100\|cone of step 10" gpurun_out/r05/stamps_aql.txt | tail -22
