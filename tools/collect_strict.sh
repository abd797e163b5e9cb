#!/bin/bash
# rocprofv3 kernel stats and HBM-side PMC passes of the strict mode (f16x3) on its throughput path, into gpurun_out/$1/
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
X="python $R/tools/strict_layers.py f16x3"
$X > $out/strict_layers_f16x3.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/profx -o ${tag}x -- $X > /dev/null 2> $out/profx.err
python $R/tools/rocprof_summary.py $out/profx/*/${tag}x_results.db > $out/rocprof_kernel_stats_strict_f16x3.txt 2>> $out/profx.err || python $R/tools/rocprof_summary.py $out/profx/${tag}x_results.db > $out/rocprof_kernel_stats_strict_f16x3.txt 2>> $out/profx.err
rocprofv3 --pmc FETCH_SIZE -d $out/pmcx1 -o x --output-format csv -- $X > /dev/null 2> $out/pmcx1.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmcx2 -o x --output-format csv -- $X > /dev/null 2> $out/pmcx2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $out/pmcx3 -o x --output-format csv -- $X > /dev/null 2> $out/pmcx3.err
python $R/tools/pmc_summary.py $out/pmcx1 $out/pmcx2 $out/pmcx3 > $out/pmc_summary_strict_f16x3.txt 2>> $out/pmc_summary.err
rm -rf $out/profx/*/*.db $out/profx/*.db $out/pmcx1 $out/pmcx2 $out/pmcx3
