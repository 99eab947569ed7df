cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-vocoder --no-profile > gpurun_out/r05/base.json 2> gpurun_out/r05/base.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r05/base.json
bash profiles/r05_stamps.sh
