#!/bin/bash
# Collects one round's evidence on the GPU box into gpurun_out/$1/ (run through gpurun from the repository root):
#   GPU test suite, smoke, race screen, default bench line (+ CPU baseline), the other bench legs, per-layer tables,
#   rocprofv3 kernel stats of the bench command, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy) of the same command.
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
rm -f gpurun_out/f16c_parity_measured.txt gpurun_out/strict_parity_measured.txt gpurun_out/f16c_conditioning_measured.txt gpurun_out/margin_measured.txt
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $out/pytest_gpu.txt
cp gpurun_out/margin_measured.txt $out/margin_measured.txt 2>/dev/null
cp gpurun_out/f16c_parity_measured.txt $out/f16c_parity_measured.txt 2>/dev/null
cp gpurun_out/strict_parity_measured.txt $out/strict_parity_measured.txt 2>/dev/null
cp gpurun_out/f16c_conditioning_measured.txt $out/f16c_conditioning_measured.txt 2>/dev/null
[ -x build/probe/mfma_peak ] && { ./build/probe/mfma_peak > $out/mfma_peak.txt 2>&1; ./build/probe/mfma_peak mixed > $out/mfma_mixed.txt 2>&1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
python tools/determinism_check.py 1500 f16c > $out/determinism.txt 2>&1
python tools/determinism_check.py 500 f16 >> $out/determinism.txt 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --streams 1 --no-graphs --no-cpu-baseline --no-strict --no-pipeline --dump-layers > $out/bench_streams1_eager.json 2> $out/layer_table.txt
python bench.py --extract-only --no-cpu-baseline --no-strict > $out/bench_extract_only.json 2>/dev/null
python bench.py --no-graphs --no-cpu-baseline --no-strict --no-pipeline > $out/bench_eager.json 2>/dev/null
python bench.py --comp-rb 0 --no-cpu-baseline --no-strict --no-pipeline > $out/bench_comp_rb0.json 2>/dev/null
python bench.py --precision f16 --no-cpu-baseline --no-strict > $out/bench_f16.json 2>/dev/null
python bench.py --mix --no-cpu-baseline --no-strict --no-pipeline > $out/bench_mix.json 2>/dev/null
python bench.py --size 1024x1024 --no-cpu-baseline --no-strict > $out/bench_1024x1024.json 2>/dev/null
python bench.py --size 1024x768 --no-cpu-baseline --no-strict > $out/bench_1024x768.json 2>/dev/null
python bench.py --size 640x480 --no-cpu-baseline --no-strict --steps 20 --warmup 5 --streams 1 --no-graphs --dump-layers > $out/bench_640x480_streams1.json 2> $out/layer_table_640x480.txt
python tools/strict_layers.py f16x3 > $out/strict_layers_f16x3.txt 2>&1
python tools/strict_layers.py f16x3d > $out/strict_layers_f16x3d.txt 2>&1
[ -x build/probe/valu_cost ] && ./build/probe/valu_cost > $out/valu_cost.txt 2>&1
python tools/pipeline_bench.py > $out/pipeline_bench.json 2> $out/pipeline_bench.err
python tools/pipeline_bench.py --workers 8 --precision f16c > $out/pipeline_bench_w8.json 2>> $out/pipeline_bench.err
python tools/match_gap_stats.py > $out/match_gap_stats.json 2> $out/match_gap_stats.err
python tools/host_soak.py --ranks 8 > $out/host_soak_8ranks.json 2> $out/host_soak.err
python tools/relax_probe.py 3 > $out/relax_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-pipeline --sustain 0"
rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- $B > $out/prof_bench.json 2> $out/prof.err
python $R/tools/rocprof_summary.py $out/prof/*/${tag}_results.db > $out/rocprof_kernel_stats.txt 2>> $out/prof.err || python $R/tools/rocprof_summary.py $out/prof/${tag}_results.db > $out/rocprof_kernel_stats.txt 2>> $out/prof.err
S1="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-pipeline --sustain 0 --no-graphs --streams 1"
rocprofv3 --kernel-trace --stats -d $out/prof1 -o ${tag}s1 -- $S1 > $out/prof_bench_streams1.json 2> $out/prof1.err
python $R/tools/rocprof_summary.py $out/prof1/*/${tag}s1_results.db > $out/rocprof_kernel_stats_streams1.txt 2>> $out/prof1.err || python $R/tools/rocprof_summary.py $out/prof1/${tag}s1_results.db > $out/rocprof_kernel_stats_streams1.txt 2>> $out/prof1.err
rm -rf $out/prof1/*/*.db $out/prof1/*.db
S="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strict --no-pipeline --sustain 0 --no-graphs --streams 1"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc1 -o x --output-format csv -- $S > /dev/null 2> $out/pmc1.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmc2 -o x --output-format csv -- $S > /dev/null 2> $out/pmc2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc3 -o x --output-format csv -- $S > /dev/null 2> $out/pmc3.err
python $R/tools/pmc_summary.py $out/pmc1 $out/pmc2 $out/pmc3 > $out/pmc_summary.txt 2> $out/pmc_summary.err
python $R/tools/pmc_to_json.py $tag $out/pmc_summary.txt > $out/pmc_traffic.json 2>> $out/pmc_summary.err
bash $R/tools/collect_strict.sh $tag
cd /tmp
rm -rf $out/prof/*/*.db $out/prof/*.db $out/pmc1 $out/pmc2 $out/pmc3   # the databases / raw CSVs are large; the summaries are what gets committed
ls $out
