#!/bin/bash
# the command-line synthesis (10-line transcript, early stop, Griffin-Lim) over and over under a watchdog that dumps the Python
# stacks of a stuck run; usage: r04_cli_stress.sh <runs>
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os
G = "tests/golden"
snap = json.load(open(os.path.join(G, "config_snapshot.json")))["lj_test.cfg"]
os.makedirs("/tmp/cli", exist_ok=True)
with open("/tmp/cli/lj_test.cfg", "w") as f:
    for k, v in sorted(snap.items()):
        f.write("%s = %r\n" % (k, v))
    f.write("test_transcript = %r\nsampledir = %r\n" % (os.path.abspath(os.path.join(G, "test_transcript_lj_test.csv")), "/tmp/cli/synth"))
    for k in ("topworkdir", "voicedir", "logdir", "datadir", "waveforms", "coarse_audio_dir", "full_audio_dir", "full_mel_dir", "attention_guide_dir"):
        f.write("%s = %r\n" % (k, "/tmp/cli/work/" + k))
    f.write("transcript = %r\n" % "/tmp/cli/work/transcript.csv")
PY
for i in $(seq 1 ${1:-30}); do
  OPH_HANG_DUMP_S=40 OPH_TRACE=1 PYTHONPATH=$GRAFT_REPO_ROOT timeout -s KILL 80 python -m ophelia_amd.synthesize -c /tmp/cli/lj_test.cfg -N 10 -odir /tmp/cli/out -random_init $((i % 7)) > /tmp/cli/run.log 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "^\[oph\]" /tmp/cli/run.log | tail -40; grep "^\[oph\]" /tmp/cli/run.log | tail -12; fi
done
