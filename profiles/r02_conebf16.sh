#!/bin/bash
# experiment: the two many-row cone contractions on the split-bf16 kernel (OPH_CONE_BF16X3=1) -- speed and deviation
cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r03o}; mkdir -p $out
for v in 0 1; do
  if [ $v = 1 ]; then export OPH_CONE_BF16X3=1; else unset OPH_CONE_BF16X3; fi
  r=$(timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2))")
  echo "cone bf16x3=$v: $r"
done
unset OPH_CONE_BF16X3
python - <<'PY'
import os, subprocess, sys, numpy as np
sys.path.insert(0, "tests")
import test_gpu_decode_modes as T
import pathlib, tempfile
tmp = pathlib.Path(tempfile.mkdtemp())
ref = T._run(tmp, "loop", {}, 200, 16, 1)
got = T._run(tmp, "bf16", {"OPH_CONE_BF16X3": "1"}, 200, 16, 1)
same = np.array_equal(got["al"].argmax(1), ref["al"].argmax(1))
print("attention trace identical:", same, " max-abs Y %.3e  align %.3e" % (np.abs(got["Y"] - ref["Y"]).max(), np.abs(got["al"] - ref["al"]).max()))
if not same:
    d = (got["al"].argmax(1) != ref["al"].argmax(1))
    print("utterances with a different trace:", int(d.any(1).sum()), "of", d.shape[0], "; first differing step per utterance:", [int(np.argmax(r)) if r.any() else -1 for r in d])
PY
