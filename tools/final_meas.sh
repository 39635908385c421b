set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -1 gpurun_out/final/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/final/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/final/rocprof.err
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/collect_pmc.py r01 256 > gpurun_out/final/pmc.log 2>&1
tail -1 gpurun_out/final/pmc.log
for w in stereo2000 fhd4000 euroc_bow; do timeout -k 10 600 python bench.py --workload $w --steps 10 --cpu-frames 0 2> gpurun_out/final/$w.err | tail -1 > gpurun_out/final/$w.json; cat gpurun_out/final/$w.json | cut -c1-400; done
