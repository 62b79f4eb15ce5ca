# Profiling session on the GPU box (see tools/rocprof_summary.py): bash tools/profile.sh <name>  ->  gpurun_out/<name>/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py > $O/bench_target.json 2> $O/bench_target.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt.json 2> $O/kt.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline > $O/pmc_sq.json 2> $O/pmc_sq.err || true
cd $R
python3 tools/rocprof_summary.py --steps 200 --kernel-stats $O/kt --pmc $O/pmc_fetch $O/pmc_write $O/pmc_sq --out-stats $O/kernel_stats.csv --out-pmc $O/pmc_summary.json
find $O -name "*.csv" -size +2M -delete
du -sh $O
