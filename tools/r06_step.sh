#!/bin/bash
# round 6 work steps on the GPU box: bash tools/r06_step.sh <tag> <what...>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
B="python bench.py --no-cpu-baseline --no-strict --no-pipeline --no-configs"
for what in "$@"; do
case $what in
  relax_probe) python tools/relax_probe.py 3 > $out/relax_probe.txt 2>&1 ;;
  f16c_tests) python -m pytest tests/test_gpu_f16c.py tests/test_gpu_margin.py tests/test_gpu_f16c_conditioning.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -600 > $out/f16c_tests.txt ;;
  all_tests) python -m pytest tests -m gpu -q 2>&1 | tail -400 > $out/pytest_gpu.txt ;;
  ab_c3b)
    for r in 1 2; do
      $B --opt c3b_plain=0 > $out/bench_c3b0_$r.json 2> $out/bench_c3b0_$r.err
      $B > $out/bench_auto_$r.json 2> $out/bench_auto_$r.err
      SFD2_C3A_KEEP_CORR=1 $B > $out/bench_auto_keepcorr_$r.json 2> $out/bench_auto_keepcorr_$r.err
    done
    $B --steps 20 --warmup 5 --streams 1 --no-graphs --dump-layers > $out/bench_streams1_eager.json 2> $out/layer_table.txt
    $B --steps 20 --warmup 5 --streams 1 --no-graphs --dump-layers --opt c3b_plain=0 > $out/bench_streams1_eager_c3b0.json 2> $out/layer_table_c3b0.txt
    ;;
  layers) $B --steps 20 --warmup 5 --streams 1 --no-graphs --dump-layers > $out/bench_streams1_eager.json 2> $out/layer_table.txt ;;
  bench) python bench.py > $out/bench.json 2> $out/bench.err ;;
  bench_quick) python bench.py --no-cpu-baseline --no-pipeline > $out/bench_quick.json 2> $out/bench_quick.err ;;
  smoke) python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1 ;;
  soak) python tools/host_soak.py --ranks 8 > $out/host_soak_8ranks.json 2> $out/host_soak.err ;;
  *) echo "unknown step $what" ;;
esac
done
ls $out
