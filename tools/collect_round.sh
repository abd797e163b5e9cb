#!/bin/bash
# Collects one round's evidence on the GPU box into gpurun_out/$1/ (run through gpurun from the repository root):
#   GPU test suite, default bench line (+ CPU baseline), extract-only / hipGraph / other-size lines, per-layer table,
#   rocprofv3 kernel stats of the bench command, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy) of the same command.
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
python tools/determinism_check.py 2000 > $out/determinism.txt 2>&1
python bench.py --steps 20 --warmup 5 --dump-layers > $out/bench.json 2> $out/layer_table.txt
python bench.py --steps 20 --warmup 5 --extract-only --no-cpu-baseline > $out/bench_extract_only.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-graphs --no-cpu-baseline --no-strict > $out/bench_eager.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-strict > $out/bench_streams1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --size 1024x1024 --no-cpu-baseline > $out/bench_1024x1024.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --size 1024x768 --no-cpu-baseline > $out/bench_1024x768.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --size 640x480 --no-cpu-baseline --dump-layers > $out/bench_640x480.json 2> $out/layer_table_640x480.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --sustain 0"
rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- $B > $out/prof_bench.json 2> $out/prof.err
python $R/tools/rocprof_summary.py $out/prof/*/${tag}_results.db > $out/rocprof_kernel_stats.txt 2>> $out/prof.err || python $R/tools/rocprof_summary.py $out/prof/${tag}_results.db > $out/rocprof_kernel_stats.txt 2>> $out/prof.err
S1="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --sustain 0 --no-graphs --streams 1"
rocprofv3 --kernel-trace --stats -d $out/prof1 -o ${tag}s1 -- $S1 > $out/prof_bench_streams1.json 2> $out/prof1.err
python $R/tools/rocprof_summary.py $out/prof1/*/${tag}s1_results.db > $out/rocprof_kernel_stats_streams1.txt 2>> $out/prof1.err || python $R/tools/rocprof_summary.py $out/prof1/${tag}s1_results.db > $out/rocprof_kernel_stats_streams1.txt 2>> $out/prof1.err
rm -rf $out/prof1/*/*.db $out/prof1/*.db
S="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict --sustain 0 --no-graphs --streams 1"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc1 -o x --output-format csv -- $S > /dev/null 2> $out/pmc1.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmc2 -o x --output-format csv -- $S > /dev/null 2> $out/pmc2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc3 -o x --output-format csv -- $S > /dev/null 2> $out/pmc3.err
python $R/tools/pmc_summary.py $out/pmc1 $out/pmc2 $out/pmc3 > $out/pmc_summary.txt 2> $out/pmc_summary.err
rm -rf $out/prof/*/*.db $out/prof/*.db   # the database is large; the summary is what gets committed
ls $out
