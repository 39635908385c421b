# Regenerates the per-round measurement artifacts in ONE gpurun call:  gpurun -- 'bash tools/final_meas.sh r02'
# -> gpurun_out/final_<tag>/ (bench lines, rocprofv3 kernel-trace stats, PMC traffic, SQ counters); tools/publish_profiles.py
# then copies the summaries into profiles/.
set -e
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-frames 0 --extras 0 --no-verify > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/collect_pmc.py $TAG 256 > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log
timeout -k 10 900 python3 tools/collect_sq.py $TAG > $OUT/sq.log 2>&1
tail -3 $OUT/sq.log | cut -c1-300
for w in stereo2000 fhd4000 euroc_bow; do timeout -k 10 600 python bench.py --workload $w --steps 10 --cpu-frames 0 2> $OUT/$w.err | tail -1 > $OUT/$w.json; cat $OUT/$w.json | cut -c1-300; done
