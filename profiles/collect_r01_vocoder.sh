#!/bin/bash
# Profile collection for the Griffin-Lim vocoder (round 1), run on the GPU box through gpurun from the repo root:
#   gpurun --timeout 1200 -- 'bash profiles/collect_r01_vocoder.sh'
# Same layout as collect_r01.sh (tag r01_vocoder); summarise with: python profiles/summarize.py r01_vocoder
set -x
T=r01_vocoder
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/profiles/vocoder_time.py"
echo "python profiles/vocoder_time.py   (16 utterances x 800 frames, n_iter 50; 3 calls on the hipFFT path, 3 on the fused path)" > $O/command.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $T -- $CMD > $O/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o $T -- $CMD > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o $T -- $CMD > $O/pmc_write.log 2>&1
# LDS pressure of the fused kernel instead of the MFMA counters (no MFMA in this path)
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o $T -- $CMD > $O/pmc_mfma.log 2>&1
ls -R $O | head -30
cd $R && timeout 120 python profiles/vocoder_time.py > $O/timing.txt 2>&1; cat $O/timing.txt
