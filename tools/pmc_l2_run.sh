#!/bin/bash
# L2 / L1 request counters and SQ wait counters of one bench pass (run through gpurun): per-kernel summary on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmcl2
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
S="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-strict --sustain 0 --no-graphs --streams 1"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/a -o x --output-format csv -- $S > /dev/null 2> $out/a.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum -d $out/b -o x --output-format csv -- $S > /dev/null 2> $out/b.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES -d $out/c -o x --output-format csv -- $S > /dev/null 2> $out/c.err
python $R/tools/pmc_summary.py $out/a $out/b $out/c
tail -3 $out/a.err $out/b.err $out/c.err | grep -i -E "error|invalid|not" | head
