set -u
# counters are collected with one band stream (bench.py --streams 1): with two concurrent band pipelines the per-dispatch counters of overlapping kernels mix
# Usage: [BENCH_ARGS="--scene 3d_hallways.json --max-depth 12"] [PMC_DIR=pmc_hallways] tools/pmc_passes.sh
R=$GRAFT_REPO_ROOT
PMC_DIR=${PMC_DIR:-pmc}
rm -rf $R/gpurun_out/$PMC_DIR
mkdir -p $R/gpurun_out/$PMC_DIR
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/$PMC_DIR/counters_list.txt 2>&1
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$PMC_DIR/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --repeats 1 --no-alone --no-other-configs --streams 1 --frames-in-flight 1 ${BENCH_ARGS:-} > $R/gpurun_out/$PMC_DIR/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU
run c FETCH_SIZE
run d WRITE_SIZE
run e GRBM_GUI_ACTIVE
