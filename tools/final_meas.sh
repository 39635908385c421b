# Regenerates the per-round measurement artifacts in ONE gpurun call:  gpurun -- 'bash tools/final_meas.sh r03'
# -> gpurun_out/final_<tag>/ (bench lines, rocprofv3 kernel-trace stats, PMC traffic, SQ counters, the other kernels' trace, the B = 1
# launch chain, the matcher stamps, the host-fed C client); tools/publish_profiles.py then copies the summaries into profiles/.
set -e
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-frames 0 --extras 0 --no-verify > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_others -- python3 $GRAFT_REPO_ROOT/tools/profile_others.py > $GRAFT_REPO_ROOT/$OUT/others.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_b1 -- python3 $GRAFT_REPO_ROOT/tools/bench_b1.py 1 100 > $GRAFT_REPO_ROOT/$OUT/b1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_matchers -- python3 $GRAFT_REPO_ROOT/tools/matcher_bench.py > $GRAFT_REPO_ROOT/$OUT/matchers_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $OUT/prof_b1 > $OUT/b1_chain.txt
python tools/trace_by_grid.py $OUT/prof_matchers k_match > $OUT/matcher_kernels.txt
python tools/bench_b1.py 1 300 >> $OUT/b1_chain.txt
timeout -k 10 900 python3 tools/collect_pmc.py $TAG 256 > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log
timeout -k 10 900 python3 tools/collect_pmc.py $TAG 32 euroc_bow > $OUT/pmc_bow.log 2>&1
tail -1 $OUT/pmc_bow.log
timeout -k 10 900 python3 tools/collect_sq.py $TAG > $OUT/sq.log 2>&1
tail -3 $OUT/sq.log | cut -c1-300
for w in stereo2000 fhd4000 euroc_bow; do timeout -k 10 600 python bench.py --workload $w --steps 10 2> $OUT/$w.err | tail -1 > $OUT/$w.json; cat $OUT/$w.json | cut -c1-300; done
python tools/matcher_bench.py 2>/dev/null > $OUT/matchers.json
python tools/diag_match_stamps.py 2>&1 | grep total > $OUT/match_stamps.txt || true
for n in 1 2 4; do ./examples/stereo_stream --streams $n --frames 3000 --nfeat 1000 | tail -1; done > $OUT/hostfed_c.txt
cat $OUT/hostfed_c.txt | cut -c1-160
python tools/pcie_bw.py > $OUT/pcie_bw.txt 2>&1; python tools/hostfed_batched_diag.py 128 30 >> $OUT/pcie_bw.txt 2>&1; tail -12 $OUT/pcie_bw.txt
# the N-rank line started WITHOUT a launcher (two ranks share the one GPU of the box: gloo rendezvous; the driver's 8-GPU run uses nccl = RCCL)
ORBX_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 2 --batch 64 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; cut -c1-200 $OUT/bench_2ranks_gloo.json
# the pair form of k_fast (opt-in) beside the default on the same box: the negative result of round 4
for p in 0 1; do ORBX_FAST_PAIR=$p python tools/ab_fast_only.py; done > $OUT/fast_pair_ab.txt 2>&1; cat $OUT/fast_pair_ab.txt
bash tools/sq_bow.sh > $OUT/sq_bow.txt 2>&1; cat $OUT/sq_bow.txt | cut -c1-300
