#!/bin/bash
# Everything profiles/rNN/ holds, in one go on a GPU box: tools/collect_profiles.sh OUTDIR   (run from the repo root)
#   bench.json               the default bench.py line
#   bench_under_rocprof.json the same measurement under the kernel trace (no CPU / config legs: they launch no timed kernels)
#   bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of that run
#   pmc/                     FETCH_SIZE, WRITE_SIZE and SQ counters of five training-gradient calls, one --pmc pass per group
out=$1
root=$PWD
mkdir -p $root/$out
python bench.py --gpus 1 --steps 20 --warmup 5 > $root/$out/bench.json 2> $root/$out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b --output-format csv -- python $root/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-configs \
    > $root/$out/bench_under_rocprof.json 2> $root/$out/bench_under_rocprof.err || exit 1
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $root/$out/bench_kernel_stats.csv
cd $root
bash tools/pmc_hbm.sh $out/pmc -- python tools/prof_train.py 2 || exit 1
python tools/pmc_json.py $out/pmc $out/pmc_summary_x6.json > /dev/null
rm -rf $out/pmc/*/*/*.db 2>/dev/null
echo done
