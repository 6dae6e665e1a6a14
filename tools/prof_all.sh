#!/bin/bash
# Evidence run for profiles/ (on the GPU box: gpurun -- tools/prof_all.sh [TAG]): the default bench line, the train-mode line,
# rocprofv3 kernel-trace stats of the same bench command, the two HBM traffic counter passes (FETCH_SIZE / WRITE_SIZE: separate
# --pmc passes with kernel-trace only, as the MI355X guide prescribes), the MFMA-busy / clock pass and the conv3x3_v6 stamps.
# Results land in gpurun_out/final/; tools/prof_collect.py then writes profiles/<TAG>_* from them (run here, commit).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/bench_steps20.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
python bench.py --mode train --steps 5 > $O/bench_train.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python bench.py --mode train --precision bf16x3 --steps 5 > $O/bench_train_x3.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "train bench done"
python bench.py --precision bf16x3 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_x3_steps10.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python tools/kbench.py bf16x3 > $O/kbench_x3.txt 2>&1 || { tail -5 $O/kbench_x3.txt; exit 1; }
echo "bf16x3 done"
python tools/kbench.py bf16 32 32 512 > $O/kbench_c5.txt 2>&1 || { tail -5 $O/kbench_c5.txt; exit 1; }
echo "c5 done"
if [ -f scratch/x/v6_stamp/lib.so ]; then HRNET_HIP_LIB=scratch/x/v6_stamp/lib.so python tools/stamps/read_v6.py > $O/v6_stamps.txt 2>&1 || true; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_steps5_under_rocprof.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x3 -o s -- python3 $R/bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/stats_x3.err || { tail -5 $O/stats_x3.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tr -o s -- python3 $R/bench.py --mode train --steps 3 --warmup 1 > /dev/null 2> $O/stats_tr.err || { tail -5 $O/stats_tr.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_trx3 -o s -- python3 $R/bench.py --mode train --precision bf16x3 --steps 3 --warmup 1 > /dev/null 2> $O/stats_trx3.err || { tail -5 $O/stats_trx3.err; exit 1; }
echo "more stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/kbench.py bf16 > $O/pmc_fetch.out 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/kbench.py bf16 > $O/pmc_write.out 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_busy -o t -- python3 $R/tools/kbench.py bf16 > $O/pmc_busy.out 2> $O/pmc_busy.err || { tail -5 $O/pmc_busy.err; exit 1; }
echo "busy done"
cd $R
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/bench_steps5_kernel_stats.csv; done
for f in $(find $O/stats_x3 -name "*kernel_stats.csv"); do cp $f $O/x3_kernel_stats.csv; done
for f in $(find $O/stats_tr -name "*kernel_stats.csv"); do cp $f $O/train_kernel_stats.csv; done
for f in $(find $O/stats_trx3 -name "*kernel_stats.csv"); do cp $f $O/train_x3_kernel_stats.csv; done
for f in $(find $O/pmc_fetch -name "*counter_collection.csv"); do cp $f $O/pmc_fetch_counter_collection.csv; done
for f in $(find $O/pmc_write -name "*counter_collection.csv"); do cp $f $O/pmc_write_counter_collection.csv; done
for f in $(find $O/pmc_busy -name "*counter_collection.csv"); do cp $f $O/pmc_busy_cc.csv; done
for f in $(find $O/pmc_busy -name "*kernel_trace.csv"); do cp $f $O/pmc_busy_kt.csv; done
rm -rf $O/stats $O/stats_x3 $O/stats_tr $O/stats_trx3 $O/pmc_fetch $O/pmc_write $O/pmc_busy
ls $O; du -sh $O
