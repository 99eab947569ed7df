#!/bin/bash
# experiment: the cone's tail (layers k >= OPH_CONE_SPLIT) on a second stream with the same CU mask (a FOURTH masked stream)
cd $GRAFT_REPO_ROOT
for v in 0 2 1; do
  if [ $v = 0 ]; then unset OPH_CONE_SPLIT; else export OPH_CONE_SPLIT=$v; fi
  r=$(timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), 'seq', round(d['config']['sequential_ms_per_step'],2))")
  echo "cone split at $v: $r"
done
unset OPH_CONE_SPLIT
python - <<'PY'
import sys, numpy as np, pathlib, tempfile
sys.path.insert(0, "tests")
import test_gpu_decode_modes as T
tmp = pathlib.Path(tempfile.mkdtemp())
for sm, mt, B in ((1, 200, 16), (0, 120, 5)):
    ref = T._run(tmp, "loop", {}, mt, B, sm)
    got = T._run(tmp, "split", {"OPH_CONE_SPLIT": "2"}, mt, B, sm)
    print("stop_mode", sm, "steps", int(got["steps"]), int(ref["steps"]), "trace identical:", np.array_equal(got["al"].argmax(1), ref["al"].argmax(1)),
          " max-abs Y %.3e  align %.3e" % (np.abs(got["Y"] - ref["Y"]).max(), np.abs(got["al"] - ref["al"]).max()))
PY
