#!/bin/bash
# One GPU-box visit (through gpurun): full -m gpu suite (log kept), smoke, bench line, rocprofv3 kernel stats + PMC passes.
# usage: scripts/gpu_round.sh <tag> [tests|notests] [pmc|nopmc] [neural]      (BENCH_ARGS="--workload c2": every bench.py call gets them;
#        name such a tag <round>_c2 so that bench.py finds the workload's counter files again)
set -u
TAG=${1:-r02a}
DO_TESTS=${2:-tests}
DO_PMC=${3:-pmc}
BA=${BENCH_ARGS:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export PYTHONDONTWRITEBYTECODE=1
if [ "$DO_TESTS" = "tests" ]; then
  (timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -150) > $OUT/gpu_tests.log
  tail -5 $OUT/gpu_tests.log
  (timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v Warning | tail -5) > $OUT/smoke.log
  tail -2 $OUT/smoke.log
fi
(timeout 600 python bench.py $BA 2>$OUT/bench.stderr | tail -1) > $OUT/bench.json
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $REPO/bench.py $BA --steps 6 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe > $OUT/trace_bench.json 2>$OUT/trace.stderr
python $REPO/scripts/step_timeline.py $OUT/trace/bench_kernel_trace.csv 20 > $OUT/step_timeline.txt 2>&1
python $REPO/scripts/frame_overlap_report.py $OUT/trace/bench_kernel_trace.csv > $OUT/frame_overlap.txt 2>&1
# the exchange's plumbing on this one-GPU box: two ranks share cuda:0 and talk through gloo (the numbers mean nothing)
(BDS_BENCH_SHARE_GPU=1 timeout 300 python $REPO/bench.py $BA --gpus 2 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe 2>$OUT/share2.stderr | tail -1) > $OUT/bench_share2.json
if [ "$DO_PMC" = "pmc" ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- \
      python $REPO/bench.py $BA --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe > /dev/null 2>$OUT/pmc_fetch.stderr
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- \
      python $REPO/bench.py $BA --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe > /dev/null 2>$OUT/pmc_write.stderr
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU \
      --output-format csv -d $OUT/pmc_sq -o bench -- \
      python $REPO/bench.py $BA --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe > /dev/null 2>$OUT/pmc_sq.stderr
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --output-format csv -d $OUT/pmc_sq2 -o bench -- \
      python $REPO/bench.py $BA --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline --no-pair-stats --no-api-path --no-random-views --no-train-cadence --no-exchange-probe > /dev/null 2>$OUT/pmc_sq2.stderr
fi
# optional 4th argument "neural": the neural bilateral variants (one-kernel transform) -- timing, kernel stats and, with pmc, the MFMA
# busy cycles of the head / fused-image kernels (SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 64 per v_mfma_f32_32x32x2_f32)
if [ "${4:-}" = "neural" ]; then
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/neural_trace -o nm -- \
      python $REPO/scripts/neural_module_time.py > $OUT/neural_module_time.txt 2>$OUT/neural_trace.stderr
  grep "transform fwd" $OUT/neural_module_time.txt
  timeout 120 python $REPO/scripts/mlp_head_time.py > $OUT/mlp_head_time.txt 2>&1
  if [ "$DO_PMC" = "pmc" ]; then
    timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
        --output-format csv -d $OUT/neural_pmc -o nm -- python $REPO/scripts/neural_module_time.py > /dev/null 2>$OUT/neural_pmc.stderr
  fi
fi
# keep what travels back small (gpurun merges <= 64 MiB): stats + counter CSVs only
find $OUT -type f ! -name "*.csv" ! -name "*.json" ! -name "*.log" ! -name "*.txt" ! -name "*.stderr" -delete
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -type f -size +20M -delete
du -sh $OUT/* | sort -h | tail -4
find $OUT -name "*.csv" | head -20
du -sh $OUT
