#!/bin/bash
# Evidence run for profiles/ (on the GPU box: gpurun -- tools/prof_all.sh): the default bench line, rocprofv3 kernel-trace stats of the
# same command, and the two HBM traffic counter passes; results land in gpurun_out/final/ (see profiles/README.md)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/bench_steps20.json
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_steps5_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/kbench.py bf16 > $O/pmc_fetch.out 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/kbench.py bf16 > $O/pmc_write.out 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
echo "write done"
cd $R
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/bench_steps5_kernel_stats.csv; done
for f in $(find $O/pmc_fetch -name "*counter_collection.csv"); do cp $f $O/pmc_fetch_counter_collection.csv; done
for f in $(find $O/pmc_write -name "*counter_collection.csv"); do cp $f $O/pmc_write_counter_collection.csv; done
find $O -maxdepth 3 | head -30
ls $O
# keep the merged-back payload small
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
du -sh $O
